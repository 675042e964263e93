// Stand-alone microbenchmark of the DiT attention forms (not part of the library).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I fangyan_tts_amd/csrc -I include tests/micro/attn_bench.hip \
//         fangyan_tts_amd/csrc/attn.hip fangyan_tts_amd/csrc/runtime.hip -o tests/micro/attn_bench
//   FY_ATTN_WAVES=4 (the 4-wave form) | unset (one workgroup per (sequence, head) where that covers the chip)   ./attn_bench [T] [nseq]
#include "attn.h"
#include "runtime.h"
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

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
    const double flops = 4.0 * T * T * 64 * H * nseq;
    printf("T %d, %d sequences: %.1f us per call (cold caches), %.0f TFLOP/s\n", T, nseq, tot * 1e3 / iters, flops / (tot * 1e-3 / iters) / 1e12);
    hipEventRecord(a, st);
    for (int i = 0; i < iters; ++i) dit_attention(qkv, out, len, nseq, T, H, 0, st);
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("back to back: %.1f us per call\n", ms * 1e3 / iters);
    return 0;
}
