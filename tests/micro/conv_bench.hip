// Stand-alone microbenchmark of the HiFT resblock convolutions (one launch per shape).  Not part of the library.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DFY_CONV_STAMPS] -I fangyan_tts_amd/csrc -I include tests/micro/conv_bench.hip \
//         fangyan_tts_amd/csrc/{conv,runtime}.hip -o tests/micro/conv_bench ;  conv_bench [1]   (1 = bf16 activation streams)
#include "conv.h"
#include "runtime.h"
#include <functional>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
extern "C" const char* fy_last_error(void);
#ifdef FY_CONV_STAMPS
void conv_dbg_read(unsigned long long* out, bool reset);
#endif

static float time_loop(hipStream_t st, int iters, const std::function<void()>& f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) f();
    hipEventRecord(a, st);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
    hipStream_t st; hipStreamCreate(&st);
    struct S { int C, KW, dil, B, L, resid; const char* name; } shapes[] = {
        {64, 11, 1, 8, 600000, 0, "s3 k11 conv1"}, {64, 11, 1, 8, 600000, 1, "s3 k11 conv2+res"}, {64, 3, 5, 8, 600000, 0, "s3 k3 d5 conv1"},
        {64, 7, 3, 8, 600000, 1, "s3 k7 d3 conv2+res"}, {128, 7, 3, 8, 200000, 0, "s2 k7 conv1"}, {128, 11, 1, 8, 200000, 1, "s2 k11 conv2+res"},
        {256, 7, 1, 8, 40000, 0, "s1 k7 conv1"}, {256, 11, 5, 8, 40000, 1, "s1 k11 conv2+res"}};
    size_t maxel = 0;
    for (auto& s : shapes) maxel = std::max(maxel, (size_t)s.B * s.L * s.C);
    float *x, *y, *r, *alpha, *bias, *wv;
    hipMalloc(&x, maxel * 4); hipMalloc(&y, maxel * 4); hipMalloc(&r, maxel * 4); hipMalloc(&alpha, 1024 * 4); hipMalloc(&bias, 1024 * 4);
    hipMalloc(&wv, 256 * 256 * 16 * 4);
    {
        std::vector<float> h(maxel);
        for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
        hipMemcpy(x, h.data(), maxel * 4, hipMemcpyHostToDevice);
        hipMemcpy(r, h.data(), maxel * 4, hipMemcpyHostToDevice);
        hipMemcpy(wv, h.data(), 256 * 256 * 16 * 4, hipMemcpyHostToDevice);
        for (int i = 0; i < 1024; ++i) h[i] = 1.0f + 0.001f * i;
        hipMemcpy(alpha, h.data(), 4096, hipMemcpyHostToDevice);
        hipMemset(bias, 0, 4096);
    }
    for (auto& s : shapes) {
        ConvW w;
        if (conv_pack(w, wv, nullptr, bias, s.C, s.C, s.KW, 1, false, true, st)) { printf("pack failed\n"); return 1; }
        ConvDesc d = {};
        d.x = x; d.x_bs = (long)s.L * s.C; d.x_ld = s.C; d.L_in = s.L;
        d.y = y; d.y_bs = (long)s.L * s.C; d.y_ld = s.C; d.L_out = s.L;
        d.resid = r; d.r_bs = d.y_bs; d.r_ld = s.C; d.bias = w.bias; d.alpha = alpha;
        d.B = s.B; d.Cin = d.Cout = s.C; d.KW = s.KW; d.dil = s.dil; d.stride = 1; d.up = 1; d.pad_left = (s.KW - 1) * s.dil; d.groups = 1;
        d.pre_act = ACT_SNAKE; d.add_resid = s.resid; d.out_scale = 1.f;
        const int streams = argc > 1 ? atoi(argv[1]) : 0;     // 1: bf16 streams as inside a ResBlock (conv1: bf16 in/out; conv2: bf16 in, fp32 RMW + bf16 out)
        if (streams) {
            d.x_act = reinterpret_cast<const bf16_t*>(x); d.alpha_out = alpha;
            d.y_act = reinterpret_cast<bf16_t*>(y) + (size_t)s.B * s.L * s.C;
            if (!s.resid) d.y = nullptr;
        }
        float us = time_loop(st, 5, [&] { if (conv1d_bf16_mfma(d, w, false, st)) { printf("err %s\n", fy_last_error()); exit(1); } });
        double el = (double)s.B * s.L * s.C;
        double gb = (streams ? el * (s.resid ? 2 + 4 + 4 + 2 : 4) : el * 4 * (2 + s.resid)) / 1e9, tf = 2.0 * el * s.C * s.KW / 1e12;
        printf("%-20s C %3d k %2d : %9.1f us  %6.2f TB/s (moved)  %6.1f TFLOP/s\n", s.name, s.C, s.KW, us, gb / us * 1e3, tf / us * 1e6);
#ifdef FY_CONV_STAMPS
        { unsigned long long t[8]; hipDeviceSynchronize(); conv_dbg_read(t, true);
          double tot = 0; for (int i = 0; i < 6; ++i) tot += t[i];
          printf("    wave0 cycles: pre-sync %.1f%%  stage %.1f%%  sync %.1f%%  taps %.1f%%  sync %.1f%%  epilogue %.1f%%   (%.0f cycles / WG)\n",
                 100 * t[0] / tot, 100 * t[1] / tot, 100 * t[2] / tot, 100 * t[3] / tot, 100 * t[4] / tot, 100 * t[5] / tot,
                 tot / (7.0 * s.B * ((s.L + (s.C >= 128 ? 127 : 255)) / (s.C >= 128 ? 128 : 256)) * (s.C >= 128 ? s.C / 128 : 1))); }
#endif
        conv_free(w);
    }
    return 0;
}
