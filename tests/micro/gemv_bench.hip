// Stand-alone microbenchmark of the decode GEMV (not part of the library).
// hipcc -O3 --offload-arch=gfx950 -I fangyan_tts_amd/csrc tests/micro/gemv_bench.hip fangyan_tts_amd/csrc/gemm.hip fangyan_tts_amd/csrc/runtime.hip -o /tmp/gemv_bench
#include "gemm.h"
#include "runtime.h"
#include <vector>
#include <stdio.h>

__global__ void empty_k() {}
__global__ void touch_k(float* p) { p[threadIdx.x] += 1.f; }

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

int main() {
    hipStream_t st; hipStreamCreate(&st);
    struct Shape { int N, K; const char* name; int mode; bool norm; } shapes[] = {
        {1152, 896, "qkv+norm", GV_STORE, true}, {896, 896, "o_proj", GV_ADD, false}, {9728, 896, "gate/up+norm", GV_SWIGLU, true},
        {896, 4864, "down", GV_ADD, false}, {6761, 896, "head+norm", GV_STORE, true}};
    float* x; hipMalloc(&x, 512 * 1024); hipMemset(x, 0, 512 * 1024);
    float* y; hipMalloc(&y, 8 * 9728 * 4); hipMemset(y, 0, 8 * 9728 * 4);
    float* nw; hipMalloc(&nw, 1024 * 4); hipMemset(nw, 0, 4096);
    float* part; hipMalloc(&part, gemv_partial_floats(8, 896, 4864) * 4 + 64);
    int* cnt; hipMalloc(&cnt, gemv_counter_ints(8, 896, 4864) * 4 + 64); hipMemset(cnt, 0, gemv_counter_ints(8, 896, 4864) * 4 + 64);
    printf("empty kernel: %.2f us/launch\n", time_loop(st, 200, [&] { hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, st); }));
    printf("touch kernel: %.2f us/launch\n", time_loop(st, 200, [&] { hipLaunchKernelGGL(touch_k, dim3(1), dim3(64), 0, st, y); }));
    for (auto& s : shapes) {
        // 24 distinct weight copies so every launch streams from HBM like a real 24-layer step
        std::vector<bf16_t*> Ws(24);
        for (auto& w : Ws) { hipMalloc(&w, gemv_packed_elems(s.N, s.K) * 2); hipMemset(w, 0, gemv_packed_elems(s.N, s.K) * 2); }
        int i = 0;
        float us = time_loop(st, 240, [&] {
            GemvArgs a; a.W = Ws[i++ % 24]; a.ldx = s.K; a.R = 8; a.N = s.N; a.K = s.K; a.y = y; a.ldy = s.N; a.mode = s.mode;
            if (s.K > 1024) { a.x_split = (const bf16_t*)x; a.partial = part; a.counters = cnt; } else a.x = x;
            if (s.mode == GV_SWIGLU) { a.mode = GV_SWIGLU_SPLIT; a.y_split = (bf16_t*)y; a.ldy = s.N / 2; }
            if (s.norm) { a.norm_w = nw; a.eps = 1e-6f; }
            gemv_bf16w(a, st);
        });
        int fixed = 0;
        float us_warm = time_loop(st, 240, [&] {
            GemvArgs a; a.W = Ws[fixed]; a.ldx = s.K; a.R = 8; a.N = s.N; a.K = s.K; a.y = y; a.ldy = s.N; a.mode = s.mode;
            if (s.K > 1024) { a.x_split = (const bf16_t*)x; a.partial = part; a.counters = cnt; } else a.x = x;
            if (s.mode == GV_SWIGLU) { a.mode = GV_SWIGLU_SPLIT; a.y_split = (bf16_t*)y; a.ldy = s.N / 2; }
            if (s.norm) { a.norm_w = nw; a.eps = 1e-6f; }
            gemv_bf16w(a, st);
        });
        printf("   same weights every launch (cache-warm): %7.2f us/launch\n", us_warm);
        double mb = (double)s.N * s.K * 2 / 1e6;
        printf("%-14s N %5d K %5d : %7.2f us/launch  %6.1f MB  %7.1f GB/s\n", s.name, s.N, s.K, us, mb, mb / us * 1e3 / 1e3);
        for (auto& w : Ws) hipFree(w);
    }
    return 0;
}
