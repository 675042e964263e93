// Latency anatomy of a small kernel on gfx950 (ablation ladder), stand-alone.
#include <hip/hip_runtime.h>
#include <functional>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int LEVEL>
__global__ __launch_bounds__(256) void ladder_k(const float* __restrict__ x, const uint16_t* __restrict__ W, float* __restrict__ y, int K) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    uint16_t* xs = (uint16_t*)sm;
    float* red = (float*)(sm + 24 * 1032 * 2);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, lr = lane & 31, kh = lane >> 5;
    const int n32 = blockIdx.x, K16 = K / 16;
    frag_ab b[16];
    if (LEVEL >= 2) {
#pragma unroll
        for (int u = 0; u < 16; ++u) { int kk = wid + u * 4; if (kk < K16) b[u] = *(const frag_ab*)(W + ((long)n32 * K16 + kk) * 512 + lane * 8); }
    }
    if (LEVEL >= 1) {
        const int r = tid >> 5, q = tid & 31;
        float4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] = make_float4(0, 0, 0, 0); if (i * 128 < K) v[i] = *(const float4*)(x + (long)r * K + (q + 32 * i) * 4); }
#pragma unroll
        for (int i = 0; i < 8; ++i) if (i * 128 < K) {
            int c4 = (q + 32 * i) * 4;
            uint32_t h0 = __float_as_uint(v[i].x) >> 16, h1 = __float_as_uint(v[i].y) >> 16, h2 = __float_as_uint(v[i].z) >> 16, h3 = __float_as_uint(v[i].w) >> 16;
            for (int pl = 0; pl < 3; ++pl) *(uint2*)(xs + (pl * 8 + r) * 1032 + c4) = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
        }
        __syncthreads();
    }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (LEVEL >= 2) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            int kk = wid + u * 4;
            if (kk < K16) {
                frag_ab af = *(const frag_ab*)(xs + min(lr, 23) * 1032 + kk * 16 + kh * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b[u], acc, 0, 0, 0);
            }
        }
    }
    float p = acc[0] + acc[4] + acc[8];
    if (LEVEL >= 3) {
        red[wid * 64 + lane] = p;
        __syncthreads();
        if (wid) return;
        p += red[64 + lane] + red[128 + lane] + red[192 + lane];
    } else if (wid) return;
    if (LEVEL >= 4) p += y[(long)kh * 4 * 1024 + n32 * 32 + lr];
    y[(long)kh * 4 * 1024 + n32 * 32 + lr] = p;
}

static float time_loop(hipStream_t st, int iters, const std::function<void()>& f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(a, st);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}

int main() {
    hipStream_t st; hipStreamCreate(&st);
    const int K = 896, N = 896;
    float *x, *y; uint16_t* W;
    hipMalloc(&x, 8 * 1024 * 4); hipMemset(x, 0, 8 * 1024 * 4);
    hipMalloc(&y, 16 * 1024 * 4); hipMemset(y, 0, 16 * 1024 * 4);
    const size_t welems = (size_t)304 * 56 * 512 + (size_t)24 * N * K + 4096;
    hipMalloc(&W, welems * 2); hipMemset(W, 0, welems * 2);
    size_t lds = 24 * 1032 * 2 + 1024 + 64;
#define RUN(L, G, T) { hipFuncSetAttribute((const void*)ladder_k<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); int i = 0; \
    printf("level %d grid %3d threads %d: %.2f us\n", L, G, T, time_loop(st, 300, [&] { hipLaunchKernelGGL(ladder_k<L>, dim3(G), dim3(T), lds, st, x, W + (size_t)(i++ % 24) * N * K, y, K); })); }
    RUN(0, 28, 256) RUN(1, 28, 256) RUN(2, 28, 256) RUN(3, 28, 256) RUN(4, 28, 256)
    RUN(0, 304, 256) RUN(4, 304, 256) RUN(0, 1, 256) RUN(4, 1, 256)
    // graph replay of 10 launches of level 4
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(ladder_k<4>, dim3(28), dim3(256), lds, st, x, W + (size_t)i * N * K, y, K);
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    printf("graph of 10 x level 4: %.2f us per kernel\n", time_loop(st, 50, [&] { hipGraphLaunch(ge, st); }) / 10);
    return 0;
}
