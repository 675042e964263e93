// Stand-alone driver of the HiFT engine through the C ABI (for rocprofv3 --pmc, which cannot sit under torch).
//   hift_bench weights.bin [batch] [frames] [iters] [flags]
#include "../../include/fy_cosy3.h"
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: hift_bench weights.bin [batch] [frames] [iters] [flags]\n"); return 2; }
    const int B = argc > 2 ? atoi(argv[2]) : 4, F = argc > 3 ? atoi(argv[3]) : 2000, iters = argc > 4 ? atoi(argv[4]) : 3;
    const unsigned flags = argc > 5 ? (unsigned)atoi(argv[5]) : 0;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 1; }
    int n = 0;
    if (fread(&n, 4, 1, f) != 1) return 1;
    std::vector<std::string> names(n);
    std::vector<fy_tensor> t(n);
    for (int i = 0; i < n; ++i) {
        int len = 0, nd = 0;
        if (fread(&len, 4, 1, f) != 1) return 1;
        names[i].resize(len);
        if (fread(&names[i][0], 1, len, f) != (size_t)len || fread(&nd, 4, 1, f) != 1) return 1;
        long long sh[4] = {1, 1, 1, 1};
        if (fread(sh, 8, nd, f) != (size_t)nd) return 1;
        size_t cnt = 1;
        for (int k = 0; k < nd; ++k) cnt *= (size_t)sh[k];
        std::vector<float> h(cnt);
        if (fread(h.data(), 4, cnt, f) != cnt) return 1;
        void* d = nullptr;
        hipMalloc(&d, cnt * 4);
        hipMemcpy(d, h.data(), cnt * 4, hipMemcpyHostToDevice);
        t[i].data = d; t[i].ndim = nd;
        for (int k = 0; k < 4; ++k) t[i].shape[k] = k < nd ? sh[k] : 0;
    }
    for (int i = 0; i < n; ++i) t[i].name = names[i].c_str();
    fclose(f);
    fy_hift* h = nullptr;
    if (fy_hift_create(&h, nullptr, t.data(), n, B, F, nullptr)) { fprintf(stderr, "create: %s\n", fy_last_error()); return 1; }
    const size_t S = (size_t)F * 480;
    float *mel, *ri, *sn, *wav;
    hipMalloc(&mel, (size_t)B * 80 * F * 4); hipMalloc(&ri, 64); hipMalloc(&sn, S * 9 * 4); hipMalloc(&wav, (size_t)B * S * 4);
    {
        std::vector<float> hm((size_t)B * 80 * F), hs(S * 9);
        for (auto& v : hm) v = (float)rand() / RAND_MAX;
        for (auto& v : hs) v = (float)rand() / RAND_MAX;
        hipMemcpy(mel, hm.data(), hm.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(sn, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
        hipMemset(ri, 0, 64);
    }
    std::vector<int> frames(B, F);
    for (int it = 0; it < iters + 1; ++it) {
        auto t0 = std::chrono::steady_clock::now();
        if (fy_hift_infer(h, mel, frames.data(), B, F, ri, sn, wav, nullptr, flags, nullptr)) { fprintf(stderr, "infer: %s\n", fy_last_error()); return 1; }
        hipDeviceSynchronize();
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (it) printf("iter %d: %.2f ms  (%.0f frames, %.1f x real time)\n", it, ms, (double)B * F, (double)B * F * 0.02 / (ms * 1e-3));
    }
    fy_hift_destroy(h);
    return 0;
}
