// How does the cost of a DiT block's four products per 400-row sequence depend on the number of sequences in the launch?
// (tile counts against 256 CUs: DESIGN.md section 10).  Not part of the library.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -I fangyan_tts_amd/csrc tests/micro/gemm_msweep.hip fangyan_tts_amd/csrc/gemm.hip fangyan_tts_amd/csrc/runtime.hip -o tests/micro/gemm_msweep_bench
#include "gemm.h"
#include "runtime.h"
#include <functional>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

static float time_loop(hipStream_t st, int iters, const std::function<void()>& f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a, st);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}

int main() {
    hipStream_t st; hipStreamCreate(&st);
    const int Mmax = 16000, Nmax = 3072, Kmax = 2048, NW = 24;
    bf16_t *A, *W, *O; float *R, *bias;
    hipMalloc(&A, (size_t)Mmax * Kmax * 2); hipMalloc(&W, (size_t)NW * Nmax * Kmax * 2); hipMalloc(&O, (size_t)Mmax * Nmax * 2);
    hipMalloc(&R, (size_t)Mmax * Nmax * 4); hipMalloc(&bias, Nmax * 4);
    {
        std::vector<bf16_t> h((size_t)Mmax * Kmax);
        for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15);
        hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        for (int i = 0; i < NW; ++i) hipMemcpy(W + (size_t)i * Nmax * Kmax, h.data() + (size_t)i * 4099, (size_t)Nmax * Kmax * 2, hipMemcpyHostToDevice);
    }
    hipMemset(R, 0, (size_t)Mmax * Nmax * 4); hipMemset(bias, 0, Nmax * 4);
    float2* rope; hipMalloc(&rope, 400 * 32 * sizeof(float2)); hipMemset(rope, 0, 400 * 32 * sizeof(float2));
    extern int gemm_tile_override;
    struct P { int N, K, mode; const char* name; } prods[] = {{3072, 1024, 0, "qkv"}, {1024, 1024, 1, "out"}, {2048, 1024, 0, "ff1"}, {1024, 2048, 1, "ff2"}};
    printf("%6s %5s | %-28s | %-28s | %-28s | %-28s | block best, per sequence\n", "M", "seqs", "qkv auto/2002/2003/1320", "out", "ff1", "ff2");
    for (int seqs : {8, 10, 12, 16, 20, 24, 32, 40}) {
        const int M = seqs * 400;
        float best_sum = 0.f;
        printf("%6d %5d |", M, seqs);
        for (auto& p : prods) {
            float best = 1e9f;
            for (int tile : {0, 2002, 2003, 1320}) {
                if (tile == 1320 && p.N % 256) { printf("   -  "); continue; }
                gemm_tile_override = tile;
                int wi = 0;
                float us = time_loop(st, 48, [&] {
                    GemmEpi e; e.bias = bias; e.ldc = p.N;
                    if (p.mode == 1) { e.mode = EPI_GATE_RESID; e.resid = R; e.gate = bias; }
                    else { e.out = O; e.out_bf16 = 1; e.act = p.N == 2048 ? ACT_GELU_TANH : ACT_NONE; if (p.N == 3072) { e.rope = rope; e.rope_T = 400; e.rope_half = 32; e.rope_stride = 1024; } }
                    gemm_bf16(A, p.K, W + (size_t)(wi++ % NW) * Nmax * Kmax, M, p.N, p.K, e, st);
                });
                printf(" %6.1f", us);
                if (us < best) best = us;
            }
            printf(" |");
            best_sum += best;
        }
        printf(" %7.1f us, %6.2f us/seq\n", best_sum, best_sum / seqs);
    }
    return 0;
}
