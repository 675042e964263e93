// How fast can a FEW CUs stream a token step's weights through LDS - and what does that cost a DiT GEMM that runs beside them?
// (the decode step that does not hold most of the chip.)  Not part of the library.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I fangyan_tts_amd/csrc -I include tests/micro/stream_few_cus.hip \
//         fangyan_tts_amd/csrc/gemm.hip fangyan_tts_amd/csrc/runtime.hip -o tests/micro/stream_few_cus_bench
// G workgroups (one per CU: the ring takes most of the LDS) of 8 waves each stream a contiguous share of a 728 MB buffer:
// every wave issues its part of a slot by LDS-DMA (non-temporal), SLOTS-1 slots ahead, waits with a counted vmcnt, meets the
// others at one raw s_barrier per slot, then reads its fragments of the slot (ds_read_b128) and runs three 16x16x32 MFMAs per
// fragment, as the decode products do (three bf16 planes of the activations against each weight fragment).
#include "gemm.h"
#include "runtime.h"
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <thread>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int SLOT_KB, int SLOTS>
__global__ __launch_bounds__(512) void stream_k(const char* __restrict__ w, long bytes_per_wg, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char ring[];
    constexpr int SLOT = SLOT_KB * 1024, PER = SLOT / (8 * 1024);          // DMA instructions (1 KB each) per wave and slot
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const char* src = w + (long)blockIdx.x * bytes_per_wg;
    const long nslot = bytes_per_wg / SLOT;
    auto issue = [&](long t) {
        char* dst = ring + (t % SLOTS) * SLOT;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int j = i * 8 + wid;
            __builtin_amdgcn_global_load_lds((const void*)(src + t * SLOT + j * 1024 + lane * 16), (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 2);
        }
    };
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    frag_ab a[3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 8; ++i) a[p][i] = (__bf16)(0.001f * (lane + p + i));
    for (long t = 0; t < SLOTS - 1 && t < nslot; ++t) issue(t);
    for (long t = 0; t < nslot; ++t) {
        const long ahead = std::min<long>(nslot - 1 - t, SLOTS - 2);
        // this wave's DMAs of slot t have landed once only the later slots' are outstanding (PER per slot, in order)
        if (ahead >= 4) __builtin_amdgcn_s_waitcnt(0x0F70 | ((4 * PER) & 15) | ((((4 * PER) >> 4) & 3) << 14));
        else if (ahead == 3) __builtin_amdgcn_s_waitcnt(0x0F70 | ((3 * PER) & 15) | ((((3 * PER) >> 4) & 3) << 14));
        else if (ahead == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * PER) & 15) | ((((2 * PER) >> 4) & 3) << 14));
        else if (ahead == 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (PER & 15));
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
        if (t + SLOTS - 1 < nslot) issue(t + SLOTS - 1);
        const char* s = ring + (t % SLOTS) * SLOT;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const frag_ab b = *reinterpret_cast<const frag_ab*>(s + (i * 8 + wid) * 1024 + lane * 16);
#pragma unroll
            for (int p = 0; p < 3; ++p) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[p], b, acc, 0, 0, 0);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) sink[blockIdx.x * 512 + tid] = acc[0];
}

template <int SLOT_KB, int SLOTS>
static float run(const char* w, size_t total, int G, float* sink, hipStream_t st, int iters) {
    const size_t lds = (size_t)SLOT_KB * 1024 * SLOTS;
    auto k = stream_k<SLOT_KB, SLOTS>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const long per = (long)(total / G / (SLOT_KB * 1024)) * (SLOT_KB * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(G), dim3(512), lds, st, w, per, sink);
    hipEventRecord(a, st);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(G), dim3(512), lds, st, w, per, sink);
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}

int main() {
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const size_t total = 728ull << 20;
    char* w; hipMalloc(&w, total);
    { std::vector<unsigned short> h(total / 2); for (auto& v : h) v = 0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15); hipMemcpy(w, h.data(), total, hipMemcpyHostToDevice); }
    float* sink; hipMalloc(&sink, 256 * 512 * 4);
    printf("728 MB through G workgroups (one per CU), LDS-DMA ring, 3 MFMAs per 1 KB fragment; us per pass, GB/s per CU, TB/s\n");
    for (int G : {24, 32, 38, 48, 64, 96, 152, 256}) {
        const float u32 = run<32, 4>(w, total, G, sink, s1, 10), u16 = run<16, 8>(w, total, G, sink, s1, 10), u16s = run<16, 6>(w, total, G, sink, s1, 10);
        printf("G %3d: 4 slots x 32 KB %8.1f us (%5.1f GB/s per CU, %.2f TB/s) | 8 x 16 KB %8.1f us | 6 x 16 KB %8.1f us\n", G, u32, total / G / u32 * 1e-3, total / u32 * 1e-6, u16, u16s);
    }
    // beside a DiT product: the stream loops on one stream while the qkv / ff1 / out GEMMs run on another
    const int M = 6400, Nmax = 3072, Kmax = 2048;
    bf16_t *A, *W, *O; float* bias;
    hipMalloc(&A, (size_t)M * Kmax * 2); hipMalloc(&W, (size_t)24 * Nmax * Kmax * 2); hipMalloc(&O, (size_t)M * Nmax * 2); hipMalloc(&bias, Nmax * 4);
    hipMemcpy(A, w, (size_t)M * Kmax * 2, hipMemcpyDeviceToDevice);
    for (int i = 0; i < 24; ++i) hipMemcpy(W + (size_t)i * Nmax * Kmax, w + (size_t)i * 4099 * 2, (size_t)Nmax * Kmax * 2, hipMemcpyDeviceToDevice);
    hipMemset(bias, 0, Nmax * 4);
    struct S { int N, K; const char* name; } shapes[] = {{3072, 1024, "qkv"}, {2048, 1024, "ff1"}, {1024, 1024, "out"}};
    auto gemm_loop = [&](int iters) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, s2);
        for (int i = 0; i < iters; ++i)
            for (auto& s : shapes) {
                GemmEpi e; e.bias = bias; e.out = O; e.out_bf16 = 1; e.ldc = s.N;
                gemm_bf16(A, s.K, W + (size_t)(i % 24) * Nmax * Kmax, M, s.N, s.K, e, s2);
            }
        hipEventRecord(b, s2); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        return ms * 1e3f / iters;
    };
    gemm_loop(5);
    const float alone = gemm_loop(40);
    printf("qkv + ff1 + out products alone: %.1f us per round\n", alone);
    for (int G : {32, 38, 48, 64}) {
        volatile bool stop = false;
        float stream_us = 0;
        std::thread th([&] {
            int n = 0; float acc = 0;
            while (!stop) { acc += run<32, 4>(w, total, G, sink, s1, 4); ++n; }
            stream_us = acc / std::max(n, 1);
        });
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        const float beside = gemm_loop(40);
        stop = true; th.join();
        printf("beside the stream on %2d CUs: products %.1f us per round (x %.2f), a 728 MB pass %.1f us (%.2f TB/s)\n", G, beside, beside / alone, stream_us, total / stream_us * 1e-6);
    }
    return 0;
}
