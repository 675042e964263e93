#include "runtime.h"
#include <stdarg.h>

static thread_local char g_err[1024] = "";

void fy_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* fy_last_error(void) { return g_err; }
extern "C" int fy_version(void) { return 100; }

int Weights::init(const fy_tensor* w, int n) {
    FY_CHECK(w != nullptr && n > 0, FY_ERR_ARG, "no weights given");
    for (int i = 0; i < n; ++i) {
        FY_CHECK(w[i].name && w[i].data && w[i].ndim >= 1 && w[i].ndim <= 4, FY_ERR_WEIGHT, "weight #%d is malformed", i);
        m[w[i].name] = &w[i];
    }
    return FY_OK;
}

const float* Weights::get(const std::string& name, std::initializer_list<long> shape) const {
    auto it = m.find(name);
    if (it == m.end()) {
        fy_set_error("missing weight '%s'", name.c_str());
        return nullptr;
    }
    const fy_tensor* t = it->second;
    bool ok = (size_t)t->ndim == shape.size();
    int i = 0;
    for (long s : shape) {
        if (ok && t->shape[i] != s) ok = false;
        ++i;
    }
    if (!ok) {
        std::string want, got;
        for (long s : shape) want += std::to_string(s) + ",";
        for (int k = 0; k < t->ndim; ++k) got += std::to_string((long)t->shape[k]) + ",";
        fy_set_error("weight '%s' has shape (%s), expected (%s)", name.c_str(), got.c_str(), want.c_str());
        return nullptr;
    }
    return (const float*)t->data;
}

// 32x32 LDS-tiled transposes
__global__ void tr_bcl_blc_k(const float* __restrict__ src, float* __restrict__ dst, int C, int L, long src_bs, long dst_bs, int dst_ld) {
    __shared__ float tile[32][33];
    int b = blockIdx.z, c0 = blockIdx.y * 32, l0 = blockIdx.x * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
    for (int i = ty; i < 32; i += 8) {
        int c = c0 + i, l = l0 + tx;
        tile[i][tx] = (c < C && l < L) ? src[b * src_bs + (long)c * L + l] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int l = l0 + i, c = c0 + tx;
        if (c < C && l < L) dst[b * dst_bs + (long)l * dst_ld + c] = tile[tx][i];
    }
}
__global__ void tr_blc_bcl_k(const float* __restrict__ src, float* __restrict__ dst, int C, int L, long src_bs, int src_ld, long dst_bs, int dst_ld) {
    __shared__ float tile[32][33];
    int b = blockIdx.z, c0 = blockIdx.y * 32, l0 = blockIdx.x * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        int l = l0 + i, c = c0 + tx;
        tile[i][tx] = (c < C && l < L) ? src[b * src_bs + (long)l * src_ld + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int c = c0 + i, l = l0 + tx;
        if (c < C && l < L) dst[b * dst_bs + (long)c * dst_ld + l] = tile[tx][i];
    }
}

int transpose_bcl_to_blc(const float* src, float* dst, int B, int C, int L, long src_bs, long dst_bs, int dst_ld, hipStream_t st) {
    dim3 grid(cdiv(L, 32), cdiv(C, 32), B);
    hipLaunchKernelGGL(tr_bcl_blc_k, grid, dim3(256), 0, st, src, dst, C, L, src_bs, dst_bs, dst_ld);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}
int transpose_blc_to_bcl(const float* src, float* dst, int B, int C, int L, long src_bs, int src_ld, long dst_bs, int dst_ld, hipStream_t st) {
    dim3 grid(cdiv(L, 32), cdiv(C, 32), B);
    hipLaunchKernelGGL(tr_blc_bcl_k, grid, dim3(256), 0, st, src, dst, C, L, src_bs, src_ld, dst_bs, dst_ld);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// ---------------------------------------------------------------------------------------------
#include <algorithm>
#include <map>
#include <mutex>
struct ProfRec { std::string name; double work; hipEvent_t a, b; };
static bool g_prof_on = false;
static std::string g_prof_only;                  // when non-empty, only scopes of this name are recorded
static std::vector<ProfRec> g_prof;
static std::mutex g_prof_mu;

ProfScope::ProfScope(const char* name, double work, hipStream_t s) : st(s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_only.empty() && ("," + g_prof_only + ",").find(std::string(",") + name + ",") == std::string::npos) return;      // a comma-separated list
    ProfRec r;
    r.name = name; r.work = work;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, st);
    g_prof.push_back(r);
    slot = (int)g_prof.size() - 1;
}
ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_prof[slot].b, st);
}

extern "C" void fy_prof_enable(int on) { g_prof_on = on != 0; }

extern "C" void fy_prof_only(const char* name) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_only = name ? name : "";
}

extern "C" void fy_prof_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
}

// sums over all recorded launches called `name`: milliseconds, algorithmic work (flops or bytes), launches
extern "C" int fy_prof_get(const char* name, double* total_ms, double* work, int64_t* count) {
    FY_CHECK(name && total_ms && work && count, FY_ERR_ARG, "fy_prof_get: null argument");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    *total_ms = 0; *work = 0; *count = 0;
    for (auto& r : g_prof) {
        if (r.name != name) continue;
        HIP_TRY(hipEventSynchronize(r.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        *total_ms += ms; *work += r.work; *count += 1;
    }
    return FY_OK;
}

// The time during which AT LEAST ONE recorded launch called `name` was running: with the flow decoders of two steps side by side
// (tts_pipeline, flow_workers = 2) two launches share the chip and each one's own duration says little about the chip's rate.
extern "C" int fy_prof_union(const char* name, double* union_ms) {
    FY_CHECK(name && union_ms, FY_ERR_ARG, "fy_prof_union: null argument");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    *union_ms = 0;
    std::vector<std::pair<double, double>> iv;
    const ProfRec* base = nullptr;
    for (auto& r : g_prof) {
        if (r.name != name) continue;
        HIP_TRY(hipEventSynchronize(r.b));
        if (!base) base = &r;
        float t0 = 0.f, t1 = 0.f;
        if (&r != base) HIP_TRY(hipEventElapsedTime(&t0, base->a, r.a));      // negative when r began before the first record (another stream)
        HIP_TRY(hipEventElapsedTime(&t1, base->a, r.b));
        iv.emplace_back((double)t0, (double)t1);
    }
    std::sort(iv.begin(), iv.end());
    double end = -1e300;
    for (auto& v : iv) {
        if (v.first > end) { *union_ms += v.second - v.first; end = v.second; }
        else if (v.second > end) { *union_ms += v.second - end; end = v.second; }
    }
    return FY_OK;
}

// ---- which streams can really run side by side? --------------------------------------------------------------------
// A HIP stream is served by a hardware queue and the hardware queues by the four pipes of the compute micro-engine; a pipe
// waits for a stream-ordered kernel to finish before it looks at the next packet, so two busy streams that land on one
// queue - or on two queues of one pipe - take turns instead of overlapping (measured on MI355X: chains of 10 us kernels
// from 2 threads take 1.0x the time of one chain on distinct pipes, 2.0x on a shared one; the pipelined benchmark loses
// 15-60 % when two of its four streams collide).  Which queue a stream gets depends on how many streams the process has
// created before, so it is measured: chains of short dependent kernels from two host threads, pair by pair.
#include <chrono>
#include <thread>
#include <vector>
__global__ void spin_k(float* y, int spin) {
    float v = y[threadIdx.x];
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    y[threadIdx.x] = v;
}

extern "C" int fy_stream_overlap(void* const* streams, int32_t n, float* ratio) {
    FY_CHECK(streams && ratio && n >= 1 && n <= 32, FY_ERR_ARG, "fy_stream_overlap: bad arguments");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    float* buf = nullptr;
    HIP_TRY(hipMalloc(&buf, (size_t)n * 64 * sizeof(float)));
    (void)hipMemset(buf, 0, (size_t)n * 64 * sizeof(float));
    const int reps = 200, spin = 600;                            // ~10 us per kernel, ~2.4 ms per chain
    auto chain = [&](int i) {
        (void)hipSetDevice(dev);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(spin_k, dim3(1), dim3(64), 0, (hipStream_t)streams[i], buf + i * 64, spin);
        (void)hipStreamSynchronize((hipStream_t)streams[i]);
    };
    auto timed = [&](int i, int j) {
        auto t0 = std::chrono::steady_clock::now();
        if (j < 0) chain(i);
        else {
            std::thread other(chain, j);
            chain(i);
            other.join();
        }
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    timed(0, -1);                                                // warm
    double single = 1e30;
    for (int r = 0; r < 3; ++r) single = std::min(single, timed(0, -1));
    for (int i = 0; i < n; ++i) {
        ratio[i * n + i] = 1.f;
        for (int j = i + 1; j < n; ++j) {
            const double t = std::min(timed(i, j), timed(i, j));
            ratio[i * n + j] = ratio[j * n + i] = (float)(t / single);
        }
    }
    (void)hipFree(buf);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// A stream whose kernels may run only on the compute units set in `mask` (bit i of word i / 32 = CU i), created by the HIP
// runtime this library is bound to - the one torch uses in the same process (a second runtime loaded by name from Python
// could be a different copy).
extern "C" int fy_stream_create_masked(void** out, const uint32_t* mask, int32_t n_words) {
    FY_CHECK(out && mask && n_words >= 1, FY_ERR_ARG, "fy_stream_create_masked: bad arguments");
    hipStream_t st = nullptr;
    HIP_TRY(hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask));
    *out = (void*)st;
    return FY_OK;
}

extern "C" int fy_stream_destroy(void* stream) {
    if (stream) HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return FY_OK;
}

// ---- synthetic tensors (fangyan_tts_amd/synth.py; SURVEY 8c "Synthetic weights") on the device -----------------------------------
// dst[i] = float(lo + (hi - lo) * u(seed, start + i)), u = (splitmix64 finaliser of seed + (index + 1) * golden) >> 11 scaled by
// 2^-53 - bit for bit what synth.uniform computes in numpy (float64 arithmetic, no contraction, one rounding to float32);
// flags bit 0: round to the nearest bf16-representable value (ties to even, synth.bf16_round); bit 1: dst = 1.0f + that (fp32 add).
// Lets bench.py and the tests fill the 0.86 G parameters without a single torch elementwise launch.
__global__ void synth_uniform_k(float* __restrict__ dst, long n, unsigned long long seed, long start, double lo, double hi, unsigned flags) {
#pragma clang fp contract(off)
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        unsigned long long z = ((unsigned long long)(start + i) + 1ull) * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0);
        const double span = hi - lo;
        const double prod = span * u;
        float x = (float)(lo + prod);
        if (flags & 1u) {
            unsigned b = __float_as_uint(x);
            b = (b + 0x7FFFu + ((b >> 16) & 1u)) & 0xFFFF0000u;
            x = __uint_as_float(b);
        }
        if (flags & 2u) x = 1.0f + x;
        dst[i] = x;
    }
}

extern "C" int fy_synth_uniform(float* dst, int64_t n, uint64_t seed, int64_t start, double lo, double hi, uint32_t flags, void* stream) {
    FY_CHECK(dst && n >= 1 && start >= 0, FY_ERR_ARG, "fy_synth_uniform: bad arguments");
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(synth_uniform_k, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, (hipStream_t)stream, dst, (long)n,
                       (unsigned long long)seed, (long)start, lo, hi, flags);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// ---- small integer tables by kernel argument (runtime.h) -------------------------------------------------------------------------
struct IntPack { int v[256]; };
__global__ void upload_ints_k(int* __restrict__ dst, IntPack p, int n) {
    const int i = threadIdx.x;
    if (i < n) dst[i] = p.v[i];
}
int upload_ints(int* dst, const int* src, int n, hipStream_t st) {
    FY_CHECK(dst && src && n >= 0, FY_ERR_ARG, "upload_ints: bad arguments");
    for (int o = 0; o < n; o += 256) {
        IntPack p;
        const int m = n - o < 256 ? n - o : 256;
        memcpy(p.v, src + o, (size_t)m * sizeof(int));
        hipLaunchKernelGGL(upload_ints_k, dim3(1), dim3(256), 0, st, dst + o, p, m);
    }
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

int current_device_slot() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev < FY_MAX_DEVICES ? dev : FY_MAX_DEVICES - 1;
}

PersistentChain& persistent_chain() {
    static PersistentChain chains[FY_MAX_DEVICES];
    return chains[current_device_slot()];
}

// How the library's host threads wait for a stream in its hot paths (fy_llm_step's look at the stop flags every 8 token steps, the
// prefill's uploads).  hipStreamSynchronize spins on a core under HIP's default schedule - and on this stack (ROCm 7.0 runtime under
// torch 2.10) hipDeviceScheduleBlockingSync changes nothing about that (tests/micro/blocking_sync_probe.py: 1.0 core either way,
// and the process then hangs at exit).  A pipelined rank has an LM thread, flow workers and a consumer waiting most of the time:
// 2.7 cores per rank.  mode 1: poll hipStreamQuery and sleep `sleep_us` between polls - a waiting thread costs ~nothing and a wait
// ends at most sleep_us late.  Process-wide.
#include <atomic>
#include <unistd.h>
static std::atomic<int> g_wait_mode{0}, g_wait_sleep_us{200};
extern "C" int fy_set_host_wait(int32_t mode, int32_t sleep_us) {
    FY_CHECK((mode == 0 || mode == 1) && sleep_us >= 1 && sleep_us <= 100000, FY_ERR_ARG, "fy_set_host_wait: mode 0 (spin) or 1 (poll + sleep), 1 <= sleep_us <= 100000");
    g_wait_mode.store(mode);
    g_wait_sleep_us.store(sleep_us);
    return FY_OK;
}
hipError_t stream_wait(hipStream_t st) {
    if (g_wait_mode.load(std::memory_order_relaxed) == 0) return hipStreamSynchronize(st);
    const int us = g_wait_sleep_us.load(std::memory_order_relaxed);
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        usleep(us);
    }
}
