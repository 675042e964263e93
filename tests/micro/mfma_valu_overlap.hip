// Does vector-ALU work issue under a running MFMA?  (not part of the library)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize tests/micro/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
// A wave runs groups of {one v_mfma_f32_32x32x16_bf16, NV pairs of (v_fma_f32, v_exp_f32)} on four rotating accumulators, with the
// accumulators in the vector file (MFMA's VGPR form) or pinned to the accumulator file (AGPR form), at one or two waves per SIMD.
// Printed: cycles of the 100 MHz clock converted with the measured kernel time -> ns per group and, for a 2.4 GHz clock, cycles.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <bool AG, int NV, bool MF>
__global__ void k(float* out, int iters) {
    f32x16 acc[4];
    for (int g = 0; g < 4; ++g) for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    frag_ab a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (threadIdx.x + j)); b[j] = (__bf16)(0.002f * (threadIdx.x - j)); }
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = 0.01f * (threadIdx.x + j);
    const float c = 0.999f, d = -0.001f;
    if constexpr (AG) { asm volatile("" : "+a"(acc[0]), "+a"(acc[1])); asm volatile("" : "+a"(acc[2]), "+a"(acc[3])); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (MF) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[g], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                x[2 * v] = fmaf(x[2 * v], c, d);
                x[2 * v + 1] = __builtin_amdgcn_exp2f(x[2 * v + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (AG) { asm volatile("" : "+a"(acc[0]), "+a"(acc[1])); asm volatile("" : "+a"(acc[2]), "+a"(acc[3])); }
    }
    float s = 0.f;
    for (int g = 0; g < 4; ++g) for (int r = 0; r < 16; ++r) s += acc[g][r];
    for (int j = 0; j < 8; ++j) s += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool AG, int NV, bool MF>
static void run(const char* what, int threads, float* out) {
    const int iters = 2000, wgs = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<AG, NV, MF>), dim3(wgs), dim3(threads), 0, 0, out, 10);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<AG, NV, MF>), dim3(wgs), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double groups = (double)iters * 4, ns = ms * 1e6 / groups;
    printf("%-58s %d waves/SIMD: %6.2f ns per group of a wave's stream = %5.1f cycles at 2.4 GHz (per SIMD: %5.1f cycles per group)\n", what, threads / 256, ns, ns * 2.4, ns * 2.4 / (threads / 256));
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    for (int threads : {256, 512}) {
        run<false, 0, true>("MFMA only, accumulators in VGPRs", threads, out);
        run<true, 0, true>("MFMA only, accumulators in AGPRs", threads, out);
        run<false, 2, false>("2 x (fma + exp2) only", threads, out);
        run<false, 2, true>("MFMA + 2 x (fma + exp2), accumulators in VGPRs", threads, out);
        run<true, 2, true>("MFMA + 2 x (fma + exp2), accumulators in AGPRs", threads, out);
        run<false, 4, false>("4 x (fma + exp2) only", threads, out);
        run<false, 4, true>("MFMA + 4 x (fma + exp2), accumulators in VGPRs", threads, out);
        run<true, 4, true>("MFMA + 4 x (fma + exp2), accumulators in AGPRs", threads, out);
    }
    return 0;
}
