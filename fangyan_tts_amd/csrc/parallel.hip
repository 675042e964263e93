// The one exchange of the data-parallel path as a C entry (fy_allgather_audio): every rank's finished audio and its
// lengths to every rank in ONE fused, fixed-size all-gather over RCCL (xGMI inside a node).  The reference's multi-GPU inference
// gathers nothing - each rank writes its own files (CosyVoice/runtime/triton_trtllm/offline_inference.py:312-322); north_star
// asks for the all-gather of the finished audio.  Python hosts use fangyan_tts_amd/parallel.py:gather_audio over
// torch.distributed (the process group is theirs); this entry is for a host that owns an RCCL communicator itself.
//
// The library does NOT link RCCL: a process must hold ONE copy of it (and of the HIP runtime under it), and which copy is the
// host's choice (torch bundles its own).  The entry resolves ncclAllGather from whatever RCCL the process has loaded with global
// symbol visibility (dlsym(RTLD_DEFAULT)); without one it fails with FY_ERR_STATE and says so.
#include "runtime.h"
#include <algorithm>
#include <atomic>
#include <dlfcn.h>
#include <rccl/rccl.h>

// record of one rank: b_max rows of s_max samples, then b_max + 1 int32 (count, lengths) bit-cast to float
__global__ void ag_pack_k(const float* __restrict__ wav, long wav_ld, const int* __restrict__ n_samples, int b, int b_max, int s_max,
                          float* __restrict__ rec) {
    const long n = (long)b_max * s_max;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int r = (int)(i / s_max), c = (int)(i % s_max);
        rec[i] = (r < b && c < n_samples[r]) ? wav[(long)r * wav_ld + c] : 0.f;
    }
    // the header publishes what the record holds: a row longer than s_max is cut, and so is its published length (a consumer
    // that reads n_all[r][1 + i] samples of a (.., s_max) row never leaves the row)
    if (blockIdx.x == 0 && threadIdx.x <= b_max) {
        const int v = threadIdx.x == 0 ? b : (threadIdx.x - 1 < b ? min(max(n_samples[threadIdx.x - 1], 0), s_max) : 0);
        rec[n + threadIdx.x] = __int_as_float(v);
    }
}
// [world][rec] -> wav_all (world * b_max, s_max) and n_all (world, b_max + 1) = count, lengths per rank
__global__ void ag_unpack_k(const float* __restrict__ recs, int world, int b_max, int s_max, float* __restrict__ wav_all, int* __restrict__ n_all) {
    const long per = (long)b_max * s_max, rec = per + b_max + 1, n = (long)world * per;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / per;
        wav_all[i] = recs[r * rec + (i - r * per)];
    }
    for (long i = blockIdx.x * 256L + threadIdx.x; i < (long)world * (b_max + 1); i += (long)gridDim.x * 256) {
        const long r = i / (b_max + 1);
        n_all[i] = __float_as_int(recs[r * rec + per + (i - r * (b_max + 1))]);
    }
}

// The same record for a host whose exchange is not the library's (fangyan_tts_amd/parallel.py gathers with torch.distributed): the
// lengths are HOST ints and ride in the kernel arguments, so nothing is copied and the stream never waits for the host.
struct AgLens { int v[256]; };
__global__ void ag_pack_host_lens_k(const float* __restrict__ wav, long wav_ld, AgLens lens, int b, int b_max, int s_max, float* __restrict__ rec) {
    const long n = (long)b_max * s_max;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int r = (int)(i / s_max), c = (int)(i % s_max);
        rec[i] = (r < b && c < lens.v[r]) ? wav[(long)r * wav_ld + c] : 0.f;
    }
    if (blockIdx.x == 0 && threadIdx.x <= b_max) {
        const int v = threadIdx.x == 0 ? b : (threadIdx.x - 1 < b ? min(max(lens.v[threadIdx.x - 1], 0), s_max) : 0);
        rec[n + threadIdx.x] = __int_as_float(v);
    }
}
extern "C" int fy_audio_record_pack(const float* wav, int64_t wav_ld, const int32_t* n_samples_host, int32_t b, int32_t b_max, int32_t s_max,
                                    float* rec, void* stream) {
    FY_CHECK(wav && n_samples_host && rec && b >= 0 && b <= b_max && b_max >= 1 && b_max <= 255 && s_max >= 1 && wav_ld >= 1, FY_ERR_ARG,
             "fy_audio_record_pack: bad arguments (b %d, b_max %d <= 255, s_max %d)", b, b_max, s_max);
    AgLens lens;
    memset(&lens, 0, sizeof(lens));
    for (int i = 0; i < b; ++i) {
        FY_CHECK(n_samples_host[i] >= 0 && n_samples_host[i] <= wav_ld, FY_ERR_ARG, "fy_audio_record_pack: utterance %d has %d samples in rows of %lld", i,
                 n_samples_host[i], (long long)wav_ld);
        lens.v[i] = n_samples_host[i];
    }
    const long n = (long)b_max * s_max;
    hipLaunchKernelGGL(ag_pack_host_lens_k, dim3((unsigned)std::min<long>(4096, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wav, (long)wav_ld, lens, b,
                       b_max, s_max, rec);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

extern "C" size_t fy_allgather_audio_scratch_floats(int32_t world, int32_t b_max, int32_t s_max) {
    return (size_t)(world + 1) * ((size_t)b_max * s_max + b_max + 1);
}

extern "C" int fy_allgather_audio(void* rccl_comm, int32_t world, const float* wav, int64_t wav_ld, const int32_t* n_samples, int32_t b,
                                  int32_t b_max, int32_t s_max, float* scratch, float* wav_all, int32_t* n_all, void* stream) {
    FY_CHECK(rccl_comm && wav && n_samples && scratch && wav_all && n_all && world >= 1 && b >= 0 && b <= b_max && b_max >= 1 && b_max <= 255 &&
                 s_max >= 1 && wav_ld >= 1, FY_ERR_ARG, "fy_allgather_audio: bad arguments (b %d, b_max %d <= 255, s_max %d)", b, b_max, s_max);
    typedef decltype(&ncclAllGather) allgather_t;
    typedef decltype(&ncclCommCount) commcount_t;
    // only a FOUND symbol is kept: a host that loads RCCL after a first (failed) call succeeds on the next one
    static std::atomic<allgather_t> ag_cached{nullptr};
    static std::atomic<commcount_t> cc_cached{nullptr};
    allgather_t ag = ag_cached.load(std::memory_order_acquire);
    if (!ag) {
        ag = (allgather_t)dlsym(RTLD_DEFAULT, "ncclAllGather");
        if (ag) ag_cached.store(ag, std::memory_order_release);
    }
    FY_CHECK(ag != nullptr, FY_ERR_STATE, "fy_allgather_audio: no RCCL in this process (ncclAllGather not found: load librccl.so with global symbol "
             "visibility - the copy the communicator was made with - before the call)");
    commcount_t cc = cc_cached.load(std::memory_order_acquire);
    if (!cc) {
        cc = (commcount_t)dlsym(RTLD_DEFAULT, "ncclCommCount");
        if (cc) cc_cached.store(cc, std::memory_order_release);
    }
    // the gather writes nranks records into a scratch sized from the caller's `world`: the two must agree
    if (cc) {
        int nranks = 0;
        const ncclResult_t rcc = cc((ncclComm_t)rccl_comm, &nranks);
        FY_CHECK(rcc == ncclSuccess, FY_ERR_HIP, "fy_allgather_audio: ncclCommCount failed (%d)", (int)rcc);
        FY_CHECK(nranks == world, FY_ERR_ARG, "fy_allgather_audio: world = %d but the communicator has %d ranks", world, nranks);
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t rec = (size_t)b_max * s_max + b_max + 1;
    float* mine = scratch;                                   // [rec]
    float* all = scratch + rec;                              // [world][rec]
    const long n = (long)b_max * s_max;
    hipLaunchKernelGGL(ag_pack_k, dim3((unsigned)std::min<long>(4096, (n + 255) / 256)), dim3(256), 0, st, wav, (long)wav_ld, n_samples, b, b_max, s_max, mine);
    HIP_TRY(hipGetLastError());
    const ncclResult_t rc = ag(mine, all, rec, ncclFloat, (ncclComm_t)rccl_comm, st);
    FY_CHECK(rc == ncclSuccess, FY_ERR_HIP, "fy_allgather_audio: ncclAllGather failed (%d)", (int)rc);
    hipLaunchKernelGGL(ag_unpack_k, dim3((unsigned)std::min<long>(4096, ((long)world * n + 255) / 256)), dim3(256), 0, st, all, world, b_max, s_max, wav_all, n_all);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}
