// Stand-alone GEMM microbenchmark (DiT shapes).  Not part of the library.
#include "gemm.h"
#include "runtime.h"
#include <algorithm>
#include <functional>
#include <math.h>
#include <string.h>
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
    // the flow decoder walks 22 blocks x 4 products of different weights (352 MB per estimator call: more than the 256 MB
    // Infinity Cache), so a product's weights come from HBM every time: the timed loop cycles through NW weight buffers
    const int NW = pmc ? 1 : 24;
    bf16_t *A, *W, *O; float *R, *bias;
    hipMalloc(&A, (size_t)Mmax * Kmax * 2); hipMalloc(&W, (size_t)NW * Nmax * Kmax * 2); hipMalloc(&O, (size_t)Mmax * Nmax * 2);
    hipMalloc(&R, (size_t)Mmax * Nmax * 4); hipMalloc(&bias, Nmax * 4);
    {   // random-ish bf16 data (clock behaviour differs on zeros)
        std::vector<bf16_t> h((size_t)Mmax * Kmax);
        for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15);
        hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        for (int i = 0; i < NW; ++i) hipMemcpy(W + (size_t)i * Nmax * Kmax, h.data() + (size_t)i * 4099, (size_t)Nmax * Kmax * 2, hipMemcpyHostToDevice);
    }
    hipMemset(R, 0, (size_t)Mmax * Nmax * 4); hipMemset(bias, 0, Nmax * 4);
    float2* rope; hipMalloc(&rope, 400 * 32 * sizeof(float2));                    // the flow decoder's qkv product rotates head 0 of q and k (T = 400)
    { std::vector<float2> t(400 * 32); for (size_t i = 0; i < t.size(); ++i) t[i] = make_float2(cosf(0.01f * i), sinf(0.01f * i)); hipMemcpy(rope, t.data(), t.size() * sizeof(float2), hipMemcpyHostToDevice); }
    struct S { int M, N, K; int mode; const char* name; } shapes[] = {
        {6400, 3072, 1024, 0, "qkv  bf16 out"}, {6400, 1024, 1024, 1, "out  gate-resid"}, {6400, 2048, 1024, 0, "ff1  gelu bf16"},
        {6400, 1024, 2048, 1, "ff2  gate-resid"}, {6400, 1024, 1024, 0, "out-shape bf16 out"}, {1280, 1024, 1024, 0, "M=1280 bf16 out"},
        {12800, 1024, 1024, 0, "M=12800 bf16 out"}, {12800, 3072, 1024, 0, "M=12800 qkv"}, {12800, 2048, 1024, 0, "M=12800 ff1"}, {3200, 3072, 1024, 0, "M=3200 qkv"}, {3200, 1024, 1024, 1, "M=3200 out"}, {3200, 1024, 2048, 1, "M=3200 ff2"}, {6400, 1024, 320, 2, "in-proj f32 out"}};
    extern int gemm_tile_override;
    if (argc > 1 && !strcmp(argv[1], "small")) {
        // stream=True sizes: 100 new rows of an incremental chunk, 612 rows of a first chunk (5 s prompt).  Tile 3: the three-stage ring of
        // 128x128 tiles; 8128: the same with eight stages (what the automatic choice takes below one tile per CU).  Bits must agree.
        struct S2 { int M, N, K, mode; const char* name; } sm[] = {
            {100, 3072, 1024, 0, "qkv"}, {100, 1024, 1024, 1, "out"}, {100, 2048, 1024, 0, "ff1"}, {100, 1024, 2048, 1, "ff2"},
            {612, 3072, 1024, 0, "qkv"}, {612, 1024, 1024, 1, "out"}, {612, 2048, 1024, 0, "ff1"}, {612, 1024, 2048, 1, "ff2"},
            {1600, 3072, 1024, 0, "qkv"}, {1600, 1024, 2048, 1, "ff2"}};
        for (auto& q : sm) {
            std::vector<uint32_t> keep;
            for (int tile : {3, 8128, 0, 6400, 6408, 6432, 6412}) {      // 64xx: the 64-deep register-staged kernel (6412: its 256x128 form for the gated residual products)
                if ((q.mode == 1) != (tile == 6412) && tile >= 6400) continue;
                gemm_tile_override = tile;
                GemmEpi e; e.bias = bias;
                if (q.mode == 1) { e.mode = EPI_GATE_RESID; e.resid = R; e.gate = bias; e.ldc = q.N; }
                else { e.out = O; e.out_bf16 = 1; e.ldc = q.N; e.act = q.N == 2048 ? ACT_GELU_TANH : ACT_NONE; }
                hipMemsetAsync(R, 0, (size_t)q.M * q.N * 4, st); hipMemsetAsync(O, 0, (size_t)q.M * q.N * 2, st);
                gemm_bf16(A, q.K, W, q.M, q.N, q.K, e, st);
                hipStreamSynchronize(st);
                std::vector<uint32_t> got((size_t)q.M * q.N * (q.mode == 1 ? 4 : 2) / 4);
                hipMemcpy(got.data(), q.mode == 1 ? (void*)R : (void*)O, got.size() * 4, hipMemcpyDeviceToHost);
                size_t bad = 0;
                if (keep.empty()) keep = got; else for (size_t i = 0; i < got.size(); ++i) bad += got[i] != keep[i];
                int turn = 0;
                float us = time_loop(st, 48, [&] { gemm_bf16(A, q.K, W + (size_t)(turn++ % NW) * Nmax * Kmax, q.M, q.N, q.K, e, st); });
                printf("%-4s M %5d N %4d K %4d tile %4d : %7.2f us   %zu words differ from tile 3%s\n", q.name, q.M, q.N, q.K, tile, us, bad, bad ? "  <-- MISMATCH" : "");
            }
        }
        return 0;
    }
    if (pmc) g_warm = 1;
    if (!pmc) {   // every tiling accumulates K in the same order: the outputs must be bit-identical to the automatic choice's
        for (auto& s : shapes) {
            if (s.mode != 0) continue;
            std::vector<bf16_t> ref((size_t)s.M * s.N), got(ref.size());
            for (int tile : {0, 256, 320, 1256, 1320, 2002, 2003, 2256, 3320, 3256, 2, 3, 128, 6400, 6408, 6432}) {     // 6400 / 6432: gemm64_k, 256x256 / 320x256 (run with FY_GEMM64=0 so that tile 0 is the 32-deep ring)
                if (tile >= 256 && tile < 2000 && s.N % 256) continue;
                gemm_tile_override = tile;
                GemmEpi e; e.bias = bias; e.out = O; e.out_bf16 = 1; e.ldc = s.N; e.act = s.N == 2048 ? ACT_GELU_TANH : ACT_NONE;
                if (s.N == 3072 && s.M % 400 == 0) { e.rope = rope; e.rope_T = 400; e.rope_half = 32; e.rope_stride = 1024; }
                hipMemsetAsync(O, 0xff, ref.size() * 2, st);
                gemm_bf16(A, s.K, W, s.M, s.N, s.K, e, st);
                hipStreamSynchronize(st);
                hipMemcpy(tile ? got.data() : ref.data(), O, ref.size() * 2, hipMemcpyDeviceToHost);
                if (tile) {
                    size_t bad = 0; double maxd = 0;
                    auto f = [](bf16_t b) { uint32_t u = (uint32_t)b << 16; float x; memcpy(&x, &u, 4); return x; };
                    for (size_t i = 0; i < ref.size(); ++i) { bad += ref[i] != got[i]; maxd = std::max(maxd, (double)fabsf(f(ref[i]) - f(got[i]))); }
                    if (tile >= 2000 && tile != 3320 && tile != 3256 && tile != 2256 && tile != 6432 && tile != 6400 && tile != 6408) { printf("check %-20s tile %4d (16x16x32 MFMA: another summation order inside the instruction): %zu of %zu outputs differ by at most %.3g\n", s.name, tile, bad, ref.size(), maxd); continue; }
                    printf("check %-20s M %5d N %4d K %4d tile %3d: %zu of %zu outputs differ from the automatic tiling%s\n", s.name, s.M, s.N, s.K, tile, bad, ref.size(), bad ? "  <-- MISMATCH" : "");
                }
            }
        }
    }
    if (!pmc) {   // the gated fp32 residual products: resid += gate * (acc + bias) from zero, every tiling against the automatic choice, bit for bit
        std::vector<float> gv(Nmax);
        for (int i = 0; i < Nmax; ++i) gv[i] = 0.25f + 0.001f * (i % 97);
        float* gate; hipMalloc(&gate, Nmax * 4); hipMemcpy(gate, gv.data(), Nmax * 4, hipMemcpyHostToDevice);
        for (auto& s : shapes) {
            if (s.mode != 1) continue;
            std::vector<float> ref((size_t)s.M * s.N), got(ref.size());
            for (int tile : {0, 3, 2003, 6412}) {
                gemm_tile_override = tile;
                GemmEpi e; e.bias = gate; e.mode = EPI_GATE_RESID; e.resid = R; e.gate = gate; e.ldc = s.N;
                hipMemsetAsync(R, 0, ref.size() * 4, st);
                gemm_bf16(A, s.K, W, s.M, s.N, s.K, e, st);
                gemm_bf16(A, s.K, W + (size_t)Nmax * Kmax, s.M, s.N, s.K, e, st);          // a second accumulation on top: the read-modify-write path
                hipStreamSynchronize(st);
                hipMemcpy(tile ? got.data() : ref.data(), R, ref.size() * 4, hipMemcpyDeviceToHost);
                if (tile) {
                    size_t bad = 0;
                    for (size_t i = 0; i < ref.size(); ++i) bad += memcmp(&ref[i], &got[i], 4) != 0;
                    printf("check %-20s M %5d N %4d K %4d tile %4d: %zu of %zu fp32 outputs differ from the automatic tiling%s\n", s.name, s.M, s.N, s.K, tile, bad, ref.size(), bad ? "  <-- MISMATCH" : "");
                }
            }
        }
        hipMemsetAsync(R, 0, (size_t)Mmax * Nmax * 4, st);
        hipStreamSynchronize(st);
    }
    for (int tile : {0, 6400, 6408, 6432, 6412, 2002, 2003, 1320, 3320, 2256, 3256}) {   // + 3320 / 3256: staggered 320x256 / 256x256 on 16x16x32; 2256: 256x256 plain
        if (pmc && tile) continue; gemm_tile_override = tile; printf("--- tile override %d (0: automatic choice; 2 / 3: LDS-DMA ring with 256x128 / 128x128 tiles; 2002 / 2003: the same on 16x16x32 MFMAs; 1320: 320x256 tiles, staggered wave groups)\n", tile);
    for (auto& s : shapes) {
        GemmEpi e;
        e.bias = bias;
        if (s.mode == 1) { e.mode = EPI_GATE_RESID; e.resid = R; e.gate = bias; e.ldc = s.N; }
        else { e.out = s.mode == 2 ? (void*)R : (void*)O; e.out_bf16 = s.mode != 2; e.ldc = s.N; e.act = s.mode == 0 && s.N == 2048 ? ACT_GELU_TANH : ACT_NONE; }
        if (s.N == 3072 && s.M % 400 == 0) { e.rope = rope; e.rope_T = 400; e.rope_half = 32; e.rope_stride = 1024; }
        int turn = 0;
        float us = time_loop(st, pmc ? 2 : 48, [&] { gemm_bf16(A, s.K, W + (size_t)(turn++ % NW) * Nmax * Kmax, s.M, s.N, s.K, e, st); });
        double gf = 2.0 * s.M * s.N * s.K / 1e9;
        printf("%-20s M %5d N %4d K %4d : %8.2f us  %7.1f TFLOP/s\n", s.name, s.M, s.N, s.K, us, gf / us);
    } }
    return 0;
}
