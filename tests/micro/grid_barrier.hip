// What does a grid-wide barrier cost on MI355X?  (Sizing a persistent LM decode kernel: 122 launch boundaries per token today.)
// G resident workgroups of 256 threads meet `rounds` times at a monotonic arrival counter in global memory (agent scope);
// every spin is bounded, so a workgroup that is not resident cannot hang the others: the kernel gives up and says so.
// Stand-alone; not part of the library.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(256) void barrier_k(unsigned* counter, unsigned* gave_up, float* sink, int rounds, int work) {
    const unsigned G = gridDim.x;
    float v = threadIdx.x;
    for (int r = 1; r <= rounds; ++r) {
        for (int i = 0; i < work; ++i) v = v * 1.0001f + 0.5f;          // stand-in for a phase's work
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = G * (unsigned)r;
            long spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 2000000) { atomicExch(gave_up, 1u); break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (__hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    sink[blockIdx.x * 256 + threadIdx.x] = v;
}

int main() {
    unsigned *counter, *gave_up; float* sink;
    hipMalloc(&counter, 4); hipMalloc(&gave_up, 4); hipMalloc(&sink, 1024 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int work : {0, 2000}) {
        for (int G : {16, 64, 128, 256, 512}) {
            float us[2];
            for (int k = 0; k < 2; ++k) {
                const int rounds = k ? 1220 : 20;           // difference = 1200 barrier rounds net of launch and ramp-up
                hipMemset(counter, 0, 4); hipMemset(gave_up, 0, 4);
                hipEventRecord(a, 0);
                hipLaunchKernelGGL(barrier_k, dim3(G), dim3(256), 0, 0, counter, gave_up, sink, rounds, work);
                hipEventRecord(b, 0); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); us[k] = ms * 1e3f;
            }
            unsigned g = 0; hipMemcpy(&g, gave_up, 4, hipMemcpyDeviceToHost);
            printf("work %4d  G %3d: %6.2f us per round%s\n", work, G, (us[1] - us[0]) / 1200.f, g ? "  (GAVE UP: not all workgroups resident)" : "");
        }
    }
    return 0;
}
