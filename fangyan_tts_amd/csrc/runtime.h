// Host-side plumbing shared by the three engines: error string, named-weight
// lookup with shape checks, device buffers, layout transposes.
#pragma once
#include "common.h"
#include "../../include/fy_cosy3.h"
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

struct Weights {
    std::unordered_map<std::string, const fy_tensor*> m;
    int init(const fy_tensor* w, int n);
    // device pointer of `name` after checking its shape; nullptr (+ error set) when absent or mis-shaped
    const float* get(const std::string& name, std::initializer_list<long> shape) const;
    bool has(const std::string& name) const { return m.count(name) != 0; }
};

// owns device allocations, frees them in the destructor
struct DevPool {
    std::vector<void*> ptrs;
    size_t bytes = 0;
    template <typename T>
    int alloc(T** p, size_t n) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, n * sizeof(T) + 256);
        if (e != hipSuccess) {
            fy_set_error("hipMalloc(%zu B) failed: %s", n * sizeof(T), hipGetErrorString(e));
            return FY_ERR_HIP;
        }
        ptrs.push_back(q);
        bytes += n * sizeof(T);
        *p = (T*)q;
        return FY_OK;
    }
    ~DevPool() {
        for (void* p : ptrs) (void)hipFree(p);
    }
};

// Two persistent grids (llm_decode_k: 152 workgroups, llm_decode32_k: 76) that are each only partly resident would wait for each
// other's CUs until the spin bound, so the launches of ALL handles of a device - of both kernels - are chained through ONE event:
// a launch waits on the GPU for the previous persistent launch.  Usage: lock mu; wait ev if set (else create); launch; record ev.
struct PersistentChain {
    std::mutex mu;
    hipEvent_t ev = nullptr;
};
PersistentChain& persistent_chain();        // of the calling thread's current device
// the calling thread's current device ordinal, clamped to [0, FY_MAX_DEVICES): the key of per-device one-time state
#define FY_MAX_DEVICES 64
int current_device_slot();
// One-time setup that belongs to a DEVICE (a function attribute, __constant__ tables): `static PerDeviceOnce once;`, tested and set at
// done[current_device_slot()] with acquire / release - safe from several threads (two threads may both do the setup: it is idempotent).
#include <atomic>
struct PerDeviceOnce {
    std::atomic<bool> done[FY_MAX_DEVICES];
};

// hipStreamSynchronize, or - after fy_set_host_wait(1, us) - hipStreamQuery polled with sleeps in between (runtime.hip)
hipError_t stream_wait(hipStream_t st);

// n host ints -> device memory WITHOUT a copy engine or a stream synchronisation: the values ride in the kernel arguments of tiny
// launches (256 per launch), so the host buffer may die at once and the caller's stream never waits for the host - a call
// (fy_flow_infer, fy_hift_infer) can be enqueued behind a previous call that is still running (tts_pipeline does that).
int upload_ints(int* dst, const int* src, int n, hipStream_t st);

// (B, C, L) -> (B, L, C) with per-batch strides in elements
int transpose_bcl_to_blc(const float* src, float* dst, int B, int C, int L, long src_bs, long dst_bs, int dst_ld, hipStream_t st);
int transpose_blc_to_bcl(const float* src, float* dst, int B, int C, int L, long src_bs, int src_ld, long dst_bs, int dst_ld, hipStream_t st);

// ---- opt-in launch profiler (bench.py's roofline leg): brackets a launch with HIP events on the
// ---- launch stream and accumulates per-name time and algorithmic work.  Off by default.
struct ProfScope {
    int slot = -1;
    hipStream_t st;
    ProfScope(const char* name, double work, hipStream_t st);
    ~ProfScope();
};
