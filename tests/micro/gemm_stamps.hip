// Where a DiT block's four products spend their time: per-workgroup 100 MHz stamps (entry, first stage landed, K loop done,
// epilogue done) of the ring GEMM, launched back to back as the flow decoder does.  Not part of the library.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DFY_GEMM_STAMPS -I fangyan_tts_amd/csrc -I include tests/micro/gemm_stamps.hip \
//         fangyan_tts_amd/csrc/gemm.hip fangyan_tts_amd/csrc/runtime.hip -o tests/micro/gemm_stamps
#include "gemm.h"
#include "runtime.h"
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

int main(int argc, char** argv) {
    hipStream_t st; hipStreamCreate(&st);
    const int M = argc > 1 ? atoi(argv[1]) : 6400, Nmax = 3072, Kmax = 2048;
    bf16_t *A, *W, *O; float *R, *bias;
    hipMalloc(&A, (size_t)M * Kmax * 2); hipMalloc(&W, (size_t)Nmax * Kmax * 2); hipMalloc(&O, (size_t)M * Nmax * 2);
    hipMalloc(&R, (size_t)M * Nmax * 4); hipMalloc(&bias, Nmax * 4);
    {
        std::vector<bf16_t> h((size_t)std::max(M, Nmax) * Kmax);
        for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15);
        hipMemcpy(A, h.data(), (size_t)M * Kmax * 2, hipMemcpyHostToDevice);
        hipMemcpy(W, h.data(), (size_t)Nmax * Kmax * 2, hipMemcpyHostToDevice);
    }
    hipMemset(R, 0, (size_t)M * Nmax * 4); hipMemset(bias, 0, Nmax * 4);
    unsigned long long* sb; const size_t SB = 4 * 4096 * 4;
    hipMalloc(&sb, SB * 8); hipMemset(sb, 0, SB * 8);
    struct S { int N, K; int mode; const char* name; } shapes[] = {{3072, 1024, 0, "qkv"}, {1024, 1024, 1, "out"}, {2048, 1024, 0, "ff1"}, {1024, 2048, 1, "ff2"}};
    extern int gemm_tile_override;
    for (int tile : {0, 6400, 6412, 6432, 3320}) {      // run with FY_GEMM64=0: tile 0 = the 32-deep ring, 6400 = gemm64_k
        gemm_tile_override = tile;
        auto block = [&](bool stamp) {
            int slot = 0;
            for (auto& s : shapes) {
                GemmEpi e; e.bias = bias; e.stamp_slot = slot++;
                if (s.mode == 1) { e.mode = EPI_GATE_RESID; e.resid = R; e.gate = bias; e.ldc = s.N; }
                else { e.out = O; e.out_bf16 = 1; e.ldc = s.N; e.act = s.N == 2048 ? ACT_GELU_TANH : ACT_NONE; }
                gemm_bf16(A, s.K, W, M, s.N, s.K, e, st);
            }
        };
        gemm_set_stamps(nullptr);
        for (int i = 0; i < 5; ++i) block(false);
        hipStreamSynchronize(st);
        gemm_set_stamps(sb);
        hipMemsetAsync(sb, 0, SB * 8, st);
        block(true); block(true);                       // the second pass overwrites the first: warm state
        hipStreamSynchronize(st);
        gemm_set_stamps(nullptr);
        std::vector<unsigned long long> h(SB);
        hipMemcpy(h.data(), sb, SB * 8, hipMemcpyDeviceToHost);
        printf("--- tile override %d, M %d (us relative to the first workgroup of the block's first product; min / median / max over workgroups)\n", tile, M);
        unsigned long long t00 = ~0ULL;
        for (int w = 0; w < 4096; ++w) if (h[w * 4]) t00 = std::min(t00, h[w * 4]);
        for (int k = 0; k < 4; ++k) {
            std::vector<double> v[4];
            for (int w = 0; w < 4096; ++w) {
                const unsigned long long* p = &h[((size_t)k * 4096 + w) * 4];
                if (!p[0]) continue;
                for (int i = 0; i < 4; ++i) v[i].push_back((double)(long long)(p[i] - t00) * 0.01);
            }
            printf("%s: %zu workgroups\n", shapes[k].name, v[0].size());
            const char* names[4] = {"entry", "first stage landed", "K loop done", "epilogue done"};
            for (int i = 0; i < 4; ++i) {
                std::sort(v[i].begin(), v[i].end());
                printf("   %-20s %8.2f %8.2f %8.2f\n", names[i], v[i].front(), v[i][v[i].size() / 2], v[i].back());
            }
            // per-workgroup durations
            std::vector<double> d[3];
            for (int w = 0; w < 4096; ++w) {
                const unsigned long long* p = &h[((size_t)k * 4096 + w) * 4];
                if (!p[0]) continue;
                for (int i = 0; i < 3; ++i) d[i].push_back((double)(long long)(p[i + 1] - p[i]) * 0.01);
            }
            const char* dn[3] = {"prologue", "K loop", "epilogue"};
            for (int i = 0; i < 3; ++i) {
                std::sort(d[i].begin(), d[i].end());
                printf("   %-20s %8.2f %8.2f %8.2f\n", dn[i], d[i].front(), d[i][d[i].size() / 2], d[i].back());
            }
        }
    }
    return 0;
}
