// Stand-alone microbenchmark of the DiT attention forms (not part of the library).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I fangyan_tts_amd/csrc -I include tests/micro/attn_bench.hip \
//         fangyan_tts_amd/csrc/attn.hip fangyan_tts_amd/csrc/attn_dit.hip fangyan_tts_amd/csrc/gemv32.hip fangyan_tts_amd/csrc/runtime.hip -o tests/micro/attn_bench
//   (attn_dit.hip wants -fno-slp-vectorize: compile it to an object first, as tests/micro/prof_attn.sh does)
//   FY_ATTN_V1=1: the round-2 kernels.  The first lines check sampled outputs (ragged lengths, plain and with a chunk mask) against a double-precision sum.
//   FY_ATTN_WAVES=4 (the 4-wave form) | unset (one workgroup per (sequence, head) where that covers the chip)   ./attn_bench [T] [nseq]
#include "attn.h"
#include "runtime.h"
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

#ifdef AT2_STAMPS
void at2_set_stamps(unsigned long long* p);
#endif

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 400, nseq = argc > 2 ? atoi(argv[2]) : 16, H = 16;
    hipStream_t st; hipStreamCreate(&st);
    const size_t M = (size_t)nseq * T;
    std::vector<unsigned short> h(M * 3 * H * 64);
    for (auto& v : h) { const float f = ((rand() & 0xffff) / 32768.f - 1.f); unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    bf16_t *qkv, *out; int* len;
    hipMalloc(&qkv, h.size() * 2); hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&out, M * H * 64 * 2);
    std::vector<int> l(nseq, T);
    hipMalloc(&len, nseq * 4); hipMemcpy(len, l.data(), nseq * 4, hipMemcpyHostToDevice);
    // sampled rows against a double-precision reference: ragged lengths, plain and with the streaming chunk mask
    {
        std::vector<int> lr(nseq);
        for (int i = 0; i < nseq; ++i) lr[i] = std::max(1, T - 29 * i);
        hipMemcpy(len, lr.data(), nseq * 4, hipMemcpyHostToDevice);
        std::vector<unsigned short> ho(M * H * 64);
        auto bf = [](unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return (double)f; };
        for (int chunk : {0, 50}) {
            hipMemset(out, 0, M * H * 64 * 2);
            if (dit_attention(qkv, out, len, nseq, T, H, chunk, st) != 0) { printf("failed: %s\n", fy_last_error()); return 1; }
            hipStreamSynchronize(st);
            hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost);
            double worst = 0, big = 0;
            int bad = 0, n = 0;
            for (int k = 0; k < 96; ++k) {
                const int sq = (k * 7) % nseq, hh = (k * 5) % H, L = lr[sq];
                const int q = k < 8 ? std::min(L - 1, k * 37) : (k < 16 ? L - 1 - (k - 8) % L : (k * 131 + 17) % L);
                const int vis = chunk > 0 ? std::min(L, (q / chunk + 1) * chunk) : L;
                const size_t ld = 3 * H * 64;
                const unsigned short* qr = h.data() + ((size_t)sq * T + q) * ld + hh * 64;
                std::vector<double> sc(vis);
                double mx = -1e300;
                for (int j = 0; j < vis; ++j) {
                    const unsigned short* kr = h.data() + ((size_t)sq * T + j) * ld + H * 64 + hh * 64;
                    double a = 0;
                    for (int d = 0; d < 64; ++d) a += bf(qr[d]) * bf(kr[d]);
                    sc[j] = a * 0.125;
                    mx = std::max(mx, sc[j]);
                }
                double l = 0;
                for (int j = 0; j < vis; ++j) { sc[j] = exp(sc[j] - mx); l += sc[j]; }
                for (int d = 0; d < 64; ++d) {
                    double o = 0;
                    for (int j = 0; j < vis; ++j) o += sc[j] * bf(h[((size_t)sq * T + j) * ld + 2 * H * 64 + hh * 64 + d]);
                    o /= l;
                    const double g = bf(ho[((size_t)sq * T + q) * H * 64 + hh * 64 + d]);
                    worst = std::max(worst, fabs(g - o));
                    big = std::max(big, fabs(o));
                    if (!(fabs(g - o) <= 0.02 * std::max(1.0, fabs(o)) * 0.5 + 4e-3)) ++bad;
                    ++n;
                }
            }
            printf("check chunk %d: %d sampled values, max |got - ref| %.3e (max |ref| %.3f), %d outside tolerance\n", chunk, n, worst, big, bad);
        }
        hipMemcpy(len, l.data(), nseq * 4, hipMemcpyHostToDevice);
    }
    // between two calls the qkv product of the real decoder rewrites the operand: stream 64 MB through the caches
    char* junk; hipMalloc(&junk, 256u << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float tot = 0.f;
    const int iters = 20;
    for (int i = 0; i < iters + 3; ++i) {
        hipMemsetAsync(junk, i, 256u << 20, st);
        hipEventRecord(a, st);
        if (dit_attention(qkv, out, len, nseq, T, H, 0, st) != 0) { printf("failed: %s\n", fy_last_error()); return 1; }
        hipEventRecord(b, st); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (i >= 3) tot += ms;
    }
#ifdef AT2_STAMPS
    {   // per-workgroup phase stamps of one call (100 MHz clock): entry, first tiles staged, key loop done, stores issued
        const int nwg = ((T + 511) / 512) * H * nseq;
        unsigned long long* sb; hipMalloc(&sb, nwg * 32); hipMemset(sb, 0, nwg * 32);
        at2_set_stamps(sb);
        hipMemsetAsync(junk, 1, 256u << 20, st);
        dit_attention(qkv, out, len, nseq, T, H, 0, st);
        hipStreamSynchronize(st);
        at2_set_stamps(nullptr);
        std::vector<unsigned long long> hs(nwg * 4);
        hipMemcpy(hs.data(), sb, nwg * 32, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t3 = 0;
        double ph[3] = {0, 0, 0}, phmax[3] = {0, 0, 0}, entry_max = 0;
        for (int i = 0; i < nwg; ++i) { t0 = std::min(t0, hs[i * 4]); t3 = std::max(t3, hs[i * 4 + 3]); }
        for (int i = 0; i < nwg; ++i) {
            entry_max = std::max(entry_max, (double)(hs[i * 4] - t0));
            for (int k = 0; k < 3; ++k) { const double d = (double)(hs[i * 4 + k + 1] - hs[i * 4 + k]); ph[k] += d / nwg; phmax[k] = std::max(phmax[k], d); }
        }
        printf("stamps over %d workgroups (us): first entry -> last exit %.2f; last entry at %.2f; staging %.2f (max %.2f), key loop %.2f (max %.2f), drain + stores %.2f (max %.2f)\n",
               nwg, (t3 - t0) / 100.0, entry_max / 100.0, ph[0] / 100.0, phmax[0] / 100.0, ph[1] / 100.0, phmax[1] / 100.0, ph[2] / 100.0, phmax[2] / 100.0);
    }
#endif
    const double flops = 4.0 * T * T * 64 * H * nseq;
    printf("T %d, %d sequences: %.1f us per call (cold caches), %.0f TFLOP/s\n", T, nseq, tot * 1e3 / iters, flops / (tot * 1e-3 / iters) / 1e12);
    hipEventRecord(a, st);
    for (int i = 0; i < iters; ++i) dit_attention(qkv, out, len, nseq, T, H, 0, st);
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("back to back: %.1f us per call\n", ms * 1e3 / iters);
    return 0;
}
