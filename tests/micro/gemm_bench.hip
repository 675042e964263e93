// Stand-alone GEMM microbenchmark (DiT shapes).  Not part of the library.
#include "gemm.h"
#include "runtime.h"
#include <functional>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

static int g_warm = 3;
static float time_loop(hipStream_t st, int iters, const std::function<void()>& f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < g_warm; ++i) f();
    hipEventRecord(a, st);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
    const bool pmc = argc > 1;        // any argument: auto kernel choice only, 1 warm-up + 2 timed launches per shape (for rocprofv3 --pmc)
    hipStream_t st; hipStreamCreate(&st);
    const int Mmax = 12800, Nmax = 3072, Kmax = 2048;
    bf16_t *A, *W, *O; float *R, *bias;
    hipMalloc(&A, (size_t)Mmax * Kmax * 2); hipMalloc(&W, (size_t)Nmax * Kmax * 2); hipMalloc(&O, (size_t)Mmax * Nmax * 2);
    hipMalloc(&R, (size_t)Mmax * Nmax * 4); hipMalloc(&bias, Nmax * 4);
    {   // random-ish bf16 data (clock behaviour differs on zeros)
        std::vector<bf16_t> h((size_t)Mmax * Kmax);
        for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15);
        hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(W, h.data(), (size_t)Nmax * Kmax * 2, hipMemcpyHostToDevice);
    }
    hipMemset(R, 0, (size_t)Mmax * Nmax * 4); hipMemset(bias, 0, Nmax * 4);
    struct S { int M, N, K; int mode; const char* name; } shapes[] = {
        {6400, 3072, 1024, 0, "qkv  bf16 out"}, {6400, 1024, 1024, 1, "out  gate-resid"}, {6400, 2048, 1024, 0, "ff1  gelu bf16"},
        {6400, 1024, 2048, 1, "ff2  gate-resid"}, {6400, 1024, 1024, 0, "out-shape bf16 out"}, {1280, 1024, 1024, 0, "M=1280 bf16 out"},
        {12800, 1024, 1024, 0, "M=12800 bf16 out"}, {12800, 3072, 1024, 0, "M=12800 qkv"}, {12800, 2048, 1024, 0, "M=12800 ff1"}, {3200, 3072, 1024, 0, "M=3200 qkv"}, {3200, 1024, 1024, 1, "M=3200 out"}, {3200, 1024, 2048, 1, "M=3200 ff2"}, {6400, 1024, 320, 2, "in-proj f32 out"}};
    extern int gemm_tile_override;
    if (pmc) g_warm = 1;
    for (int tile : {128, 0, 2, 3}) {
        if (pmc && tile) continue; gemm_tile_override = tile; printf("--- tile override %d (128: register-staged 128x128x64 only; 0: automatic choice; 2 / 3: LDS-DMA ring with 256x128 / 128x128 tiles wherever it applies)\n", tile);
    for (auto& s : shapes) {
        GemmEpi e;
        e.bias = bias;
        if (s.mode == 1) { e.mode = EPI_GATE_RESID; e.resid = R; e.gate = bias; e.ldc = s.N; }
        else { e.out = s.mode == 2 ? (void*)R : (void*)O; e.out_bf16 = s.mode != 2; e.ldc = s.N; e.act = s.mode == 0 && s.N == 2048 ? ACT_GELU_TANH : ACT_NONE; }
        float us = time_loop(st, pmc ? 2 : 30, [&] { gemm_bf16(A, s.K, W, s.M, s.N, s.K, e, st); });
        double gf = 2.0 * s.M * s.N * s.K / 1e9;
        printf("%-20s M %5d N %4d K %4d : %8.2f us  %7.1f TFLOP/s\n", s.name, s.M, s.N, s.K, us, gf / us);
    } }
    return 0;
}
