// How many kernel launches per second can T host threads push, each into its own stream?  (The pipelined benchmark
// issues ~145k launches/s from four threads.)  Stand-alone; not part of the library.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <stdio.h>
#include <thread>
#include <vector>

__global__ void tiny_k(float* y, int spin) {
    float v = y[threadIdx.x];
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    y[threadIdx.x] = v;
}

int main() {
    float* buf; hipMalloc(&buf, 64 * 1024 * 4); hipMemset(buf, 0, 64 * 1024 * 4);
    std::vector<hipStream_t> all(8);
    for (auto& s : all) hipStreamCreate(&s);         // created once: the stream -> hardware queue assignment stays fixed
    for (auto& s : all) { hipLaunchKernelGGL(tiny_k, dim3(1), dim3(64), 0, s, buf, 0); hipStreamSynchronize(s); }   // every stream has been used once
    hipLaunchKernelGGL(tiny_k, dim3(1), dim3(64), 0, 0, buf, 0); hipDeviceSynchronize();                           // and the null stream
    for (int spin : {600}) {                      // 0: empty kernel; 600: ~10 us of dependent FMAs, like a decode kernel
        for (int T : {2, 3, 4, 5}) {
            std::vector<hipStream_t> st(all.begin(), all.begin() + T);
            const int N = spin ? 4000 : 20000;
            auto work = [&](int t) {
                hipSetDevice(0);
                for (int i = 0; i < N; ++i) {
                    hipLaunchKernelGGL(tiny_k, dim3(28), dim3(256), 0, st[t], buf + t * 4096, spin);
                    if ((i & 1023) == 1023) hipStreamSynchronize(st[t]);
                }
                hipStreamSynchronize(st[t]);
            };
            work(0);                                  // warm
            for (int extra : {0, 1}) {
            std::atomic<bool> stop{false};
            std::thread occ;
            if (extra) occ = std::thread([&] {          // a further stream that launches only now and then (every ~200 us), like a main thread
                hipSetDevice(0);
                while (!stop) { hipLaunchKernelGGL(tiny_k, dim3(1), dim3(64), 0, all[7], buf + 60000, 0); hipStreamSynchronize(all[7]); std::this_thread::sleep_for(std::chrono::microseconds(200)); }
            });
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back(work, t);
            for (auto& x : th) x.join();
            double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("spin %4d  threads %d%s: %7.1f k launches/s in total, %.2f us per launch per thread\n", spin, T, extra ? " + occasional stream" : "", T * N / s / 1e3, s / N * 1e6);
            stop = true; if (extra) occ.join(); }
        }
    }
    return 0;
}
