// Stand-alone microbenchmark of the 32-row decode products (not part of the library).
// hipcc -O3 --offload-arch=gfx950 -I fangyan_tts_amd/csrc tests/micro/gemv32_bench.hip fangyan_tts_amd/csrc/gemv32.hip fangyan_tts_amd/csrc/gemm.hip fangyan_tts_amd/csrc/runtime.hip -o tests/micro/gemv32_bench
// Launch shapes are chosen by FY_GV32_KS / FY_GV32_NT2_FROM / FY_GV32_NW8_BELOW (read once per process).
#include "gemm.h"
#include "gemv32.h"
#include "runtime.h"
#include <functional>
#include <vector>
#include <stdio.h>

static float time_loop(hipStream_t st, int iters, const std::function<void()>& f) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(a, st);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b, st);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
    hipStream_t st; hipStreamCreate(&st);
    const int H = 896, I = 4864;
    struct Shape { int N, K; const char* name; int mode; bool norm; } shapes[] = {
        {1152, H, "qkv+norm", GV32_STORE, true}, {H, H, "o_proj", GV32_ADD_IMG, false}, {2 * I, H, "gate/up+norm", GV32_SWIGLU_IMG, true},
        {H, I, "down", GV32_ADD_IMG, false}, {6761, H, "head+norm", GV32_STORE, true}};
    for (int R : {8, 32, 64}) {
        bf16_t *img_h, *img_a;
        hipMalloc(&img_h, gv32_image_elems(R, H) * 2); hipMemset(img_h, 0, gv32_image_elems(R, H) * 2);
        hipMalloc(&img_a, gv32_image_elems(R, I) * 2); hipMemset(img_a, 0, gv32_image_elems(R, I) * 2);
        float *y, *ln, *ssq, *part; int* cnt;
        hipMalloc(&y, (size_t)R * 9728 * 4); hipMemset(y, 0, (size_t)R * 9728 * 4);
        hipMalloc(&ln, 9728 * 4); hipMemset(ln, 0, 9728 * 4);
        hipMalloc(&ssq, gv32_ssq_floats(R, H) * 4); hipMemset(ssq, 0, gv32_ssq_floats(R, H) * 4);
        hipMalloc(&part, gv32_partial_floats(R, H, I) * 4 + 64);
        hipMalloc(&cnt, gv32_counter_ints(R, H, I) * 4 + 64); hipMemset(cnt, 0, gv32_counter_ints(R, H, I) * 4 + 64);
        float total = 0.f;
        for (auto& s : shapes) {
            std::vector<bf16_t*> Ws(24);      // 24 distinct weight copies so every launch streams from HBM like a real 24-layer step
            for (auto& w : Ws) { hipMalloc(&w, gemv_packed_elems(s.N, s.K) * 2); hipMemset(w, 0, gemv_packed_elems(s.N, s.K) * 2); }
            int i = 0;
            float us = time_loop(st, 240, [&] {
                Gv32Args a; a.W = Ws[i++ % 24]; a.R = R; a.N = s.N; a.K = s.K; a.mode = s.mode;
                a.img = s.K == H ? img_h : img_a;
                if (s.norm) { a.ssq = ssq; a.n_ssq = H / 32; a.eps = 1e-6f; }
                if (s.mode == GV32_STORE) { a.y = y; a.ldy = s.N; }
                if (s.mode == GV32_ADD_IMG) { a.y = y; a.ldy = s.N; a.ln_next = ln; a.img_out = img_h; a.ssq_out = ssq; }
                if (s.mode == GV32_SWIGLU_IMG) a.img_out = img_a;
                if (s.K > 2048) { a.partial = part; a.counters = cnt; }
                if (gemv32(a, st) != 0) { printf("gemv32 failed: %s\n", fy_last_error()); exit(1); }
            });
            double mb = (double)s.N * s.K * 2 / 1e6;
            printf("R %2d %-14s N %5d K %5d : %7.2f us/launch  %6.1f MB  %7.1f GB/s\n", R, s.name, s.N, s.K, us, mb, mb / us * 1e3 / 1e3);
            total += us * (s.N == 6761 ? 1.f / 24.f : 1.f);
            for (auto& w : Ws) hipFree(w);
        }
        printf("R %2d: products of one layer + head/24: %.1f us -> x24 = %.0f us per token step (attention, sampler not included)\n", R, total, total * 24);
        hipFree(img_h); hipFree(img_a); hipFree(y); hipFree(ln); hipFree(ssq); hipFree(part); hipFree(cnt);
    }
    return 0;
}
