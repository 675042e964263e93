// Persistent decode step of the speech-token LM for up to 32 sequences on a few compute units (llm_decode32.h).
//
//   * G = inter / (16 TG) workgroups of 8 waves, one per CU (38 for CosyVoice3-0.5B at TG = 8), resident for the whole token step.
//     Workgroup g owns, per layer: qkv column tile g, attention units g and g + G of the R x Hk (row, kv head) pairs, o-proj tile
//     g, the TG gate/up tiles [TG g, TG g + TG) = SwiGLU columns [16 TG g, ...) = K fragments [TG g, ...) of the down projection
//     (so gate/up and its K slice of `down` run back to back through LDS with no hand-off between them), and the fixed-order
//     reduction of down-projection tile g over the G partial tiles.  Five grid-wide hand-offs per layer.
//   * Products as in gemv32.hip: the operand is the gv32 A image (the exact 3-way bf16 split, written by the producer), read into
//     REGISTERS once per phase (K = hidden: 7 fragments x 3 planes per wave) and kept there for all of the workgroup's tiles;
//     weights straight from HBM into registers (non-temporal, a tile ahead); three v_mfma_f32_32x32x16_bf16 per 1 KiB weight
//     fragment; the eight waves split K and meet in LDS in wave order.  With bf16-exact weights every product is exact, sums
//     are fp32 in a fixed order: the ids equal the per-operation path's and the reference fixtures'.
//   * Attention: one (row, kv head) unit at a time for the whole query-head group (attn.hip: llm_attention_step_gqa_k's body).
//   * Hand-off = llm_decode.hip's: every byte another workgroup reads is stored write-through (sc1) and read with sc1 loads; all
//     waves drain, the workgroup raises its own flag word, workgroup 0 polls the flag array and raises one "go" word.  Every spin
//     is bounded by the 100 MHz clock (1 s); a time-out sets the status word and every workgroup leaves.
#include "llm_decode32.h"
#include "gemv32.h"
#include "runtime.h"
#include <mutex>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 d32_frag;
typedef uint32_t d32_u4 __attribute__((ext_vector_type(4)));
#define D32_SC1 16
#define D32_GO 32                          // the "go" word sits this many words behind the flag array
#define D32_RLX __ATOMIC_RELAXED
#define D32_AGT __HIP_MEMORY_SCOPE_AGENT
#define D32_CH 128                         // positions of the value cache staged per pass of the attention

struct D32Args {
    const Dec32Layer* layers;
    int n_layers;
    const bf16_t* w_head;
    const bf16_t* w_head_lo;               // exact-weights mode (XW): the head's lo plane
    const float* norm_w;
    int R, G, TG, H, I, QKV, Hq, Hk, NS, max_ctx;
    float eps;
    float* h;
    bf16_t* img_h;
    float* ssq;
    float* qkv;
    bf16_t* img_ao;
    float* part;                           // [G][H/32 tiles][4][64][4] partial tiles of the down projection, accumulator layout
    float* logits;
    const int* st;
    const float* inv_freq;
    unsigned* flags;
    unsigned epoch0;
    unsigned* status;
    unsigned long long* stamps;
};

// LDS map (bytes)
#define D32_RED 0                          // [2][8 waves][16][64] floats: the waves' accumulators, double-buffered (64 KB);
                                           // the attention's scratch and the reduction's staging live here in their phases
#define D32_SLICE 65536                    // [TG <= 8][3][64][8] bf16: the SwiGLU slice as A fragments of the down projection
#define D32_RSTD (65536 + 24576)           // [32] floats
#define D32_MISC (D32_RSTD + 128)          // fail flag
#define D32_LDS (D32_MISC + 64)

__device__ __forceinline__ d32_frag d32_ld_nt(const bf16_t* p) {
    d32_u4 r = __builtin_nontemporal_load(reinterpret_cast<const d32_u4*>(p));
    return __builtin_bit_cast(d32_frag, r);
}
__device__ __forceinline__ float d32_ld(const float* p) { return __hip_atomic_load(p, D32_RLX, D32_AGT); }
__device__ __forceinline__ void d32_st(float* p, float v) { __hip_atomic_store(p, v, D32_RLX, D32_AGT); }
// 16-byte write-through accesses: a wave-uniform descriptor over the whole buffer + this lane's byte offset
typedef __amdgpu_buffer_rsrc_t d32_rsrc;
__device__ __forceinline__ d32_rsrc d32_desc(const void* base, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 d32_ld4(d32_rsrc rs, long float_index) {
    const d32_u4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(float_index * 4), 0, D32_SC1);
    return make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
}
__device__ __forceinline__ void d32_st4(d32_rsrc rs, long float_index, const float4& v) {
    const d32_u4 r = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(r, rs, (int)(float_index * 4), 0, D32_SC1);
}
// four consecutive columns of one row -> the three planes of a gv32 image, write-through (8 bytes each)
__device__ __forceinline__ void d32_put4(bf16_t* img, int K16, int row, int col0, float x0, float x1, float x2, float x3) {
    unsigned h[4], m[4], l[4];
    gv32_split3(x0, h[0], m[0], l[0]); gv32_split3(x1, h[1], m[1], l[1]);
    gv32_split3(x2, h[2], m[2], l[2]); gv32_split3(x3, h[3], m[3], l[3]);
    char* base = reinterpret_cast<char*>(img);
    auto pack = [](unsigned a, unsigned b, unsigned c, unsigned d) { return (unsigned long long)(a | (b << 16)) | ((unsigned long long)(c | (d << 16)) << 32); };
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(base + gv32_off(K16, row, col0, 0)), pack(h[0], h[1], h[2], h[3]), D32_RLX, D32_AGT);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(base + gv32_off(K16, row, col0, 1)), pack(m[0], m[1], m[2], m[3]), D32_RLX, D32_AGT);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(base + gv32_off(K16, row, col0, 2)), pack(l[0], l[1], l[2], l[3]), D32_RLX, D32_AGT);
}

// ---- grid-wide hand-off (llm_decode.hip's protocol) ---------------------------------------------------------------------
__device__ __forceinline__ void d32_signal(const D32Args& a, int g, int lane, unsigned epoch) {
    if (lane == 0) __hip_atomic_store(a.flags + g, epoch, D32_RLX, D32_AGT);
}
__device__ __forceinline__ bool d32_grid_wait(const D32Args& a, int g, int lane, unsigned epoch) {
    unsigned* go = a.flags + ((a.G + 3) / 4) * 4 + D32_GO;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (g != 0) {
        for (;;) {
            if ((int)(__hip_atomic_load(go, D32_RLX, D32_AGT) - epoch) >= 0) return true;
            if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) return false;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(a.flags, 0, ((a.G + 3) / 4) * 16, 0x00020000);
    for (;;) {
        bool ok = true;
        if (lane * 4 < a.G) {
            const d32_u4 f = __builtin_amdgcn_raw_buffer_load_b128(rf, lane * 16, 0, D32_SC1);
#pragma unroll
            for (int i = 0; i < 4; ++i) ok = ok && (lane * 4 + i >= a.G || (int)(f[i] - epoch) >= 0);
        }
        if (__all(ok)) {
            if (lane == 0) __hip_atomic_store(go, epoch, D32_RLX, D32_AGT);
            return true;
        }
        if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) return false;
    }
}
// end of a phase: every wave drains its write-through stores, the workgroup meets, wave 0 raises the flag; then EVERY wave requests
// what the next phase needs that does not depend on the hand-off (its first weight fragments: `prefetch`), so those loads fly
// while wave 0 waits for the grid (its poll comes back behind its own requests: vector-memory operations return in order, and
// the round trip of the hand-off is longer than theirs); the workgroup meets again.
template <class Pf>
__device__ __forceinline__ bool d32_handoff(const D32Args& a, char* smem, int g, int wid, int lane, unsigned& epoch, Pf prefetch) {
    volatile int* fail = reinterpret_cast<volatile int*>(smem + D32_MISC);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++epoch;
    if (wid == 0) d32_signal(a, g, lane, epoch);
    prefetch();
    if (wid == 0 && !d32_grid_wait(a, g, lane, epoch)) {
        if (lane == 0) { *fail = 1; __hip_atomic_store(a.status, 1u, D32_RLX, D32_AGT); }
    }
    __syncthreads();
    return *fail == 0;
}

// ---- products -----------------------------------------------------------------------------------------------------------
// this wave's A fragments of a K = hidden operand image (fragments kk = wid + 8 u), write-through data: sc1 loads
template <int FW>
__device__ __forceinline__ void d32_load_image(const bf16_t* img, int K16, d32_frag (&A)[FW][3], int wid, int lane) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(img), 0, K16 * 3072, 0x00020000);
#pragma unroll
    for (int u = 0; u < FW; ++u)
#pragma unroll
        for (int p = 0; p < 3; ++p)
            A[u][p] = __builtin_bit_cast(d32_frag, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, ((wid + 8 * u) * 3 + p) * 1024, D32_SC1));
}
template <int FW>
__device__ __forceinline__ void d32_load_w(const bf16_t* W, long tile, int K16, d32_frag (&b)[FW], int wid, int lane) {
#pragma unroll
    for (int u = 0; u < FW; ++u) b[u] = d32_ld_nt(W + (tile * K16 + wid + 8 * u) * 512 + lane * 8);
}
template <int FW>
__device__ __forceinline__ void d32_mfma(const d32_frag (&A)[FW][3], const d32_frag (&b)[FW], f32x16& acc) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int u = 0; u < FW; ++u)
#pragma unroll
        for (int p = 0; p < 3; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[u][p], b[u], acc, 0, 0, 0);
}
// exact-weights mode: the lo-plane fragments of the same tile meet the two leading operand planes and continue the accumulator
template <int FW>
__device__ __forceinline__ void d32_mfma_lo(const d32_frag (&A)[FW][3], const d32_frag (&b)[FW], f32x16& acc) {
#pragma unroll
    for (int u = 0; u < FW; ++u)
#pragma unroll
        for (int p = 0; p < 2; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[u][p], b[u], acc, 0, 0, 0);
}
__device__ __forceinline__ void d32_park(float* red, int buf, int wid, int lane, const f32x16& acc) {
    float* p = red + ((buf * 8 + wid) * 16) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) p[i * 64] = acc[i];
}
// item (row r, columns c0 .. c0+3) of a parked tile: the eight waves' partial sums in wave order (threads 0 .. 255)
__device__ __forceinline__ float4 d32_combine(const float* red, int buf, int it) {
    const int r = it >> 3, c0 = (it & 7) * 4;
    const int i = (r & 3) + 4 * (r >> 3), kh = (r >> 2) & 1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const float4 x = *reinterpret_cast<const float4*>(red + ((buf * 8 + w) * 16 + i) * 64 + 32 * kh + c0);
        v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
    }
    return v;
}
// 1 / rms of the 32 rows from the NP per-tile sums of squares [NP][32] (gemv32_k's order of the sums)
__device__ __forceinline__ void d32_rstd(const float* ssq, int NP, float eps, int H, float* rstd_s, int tid) {
    if (tid < 256) {
        const int r = tid >> 3, part = tid & 7;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tt = part + 8 * i;
            v[i] = tt < NP ? d32_ld(ssq + (long)tt * 32 + r) : 0.f;
        }
        float s = (v[0] + v[1]) + (v[2] + v[3]);
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (part == 0) rstd_s[r] = rsqrtf(s / H + eps);
    }
}
// residual epilogue of item `it` of tile `tile`: h += x, the A image of ln_next x h, the tile's sums of squares (all write-through)
// (y0: the residual row piece, requested by the caller before the products)
__device__ __forceinline__ void d32_add_img(const D32Args& a, d32_rsrc rs_h, int tile, int it, const float4& x, const float4& y0, const float* ln_next) {
    const int r = it >> 3, c0 = (it & 7) * 4, n = tile * 32 + c0;
    float hn[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < a.R) {
        hn[0] = y0.x + x.x; hn[1] = y0.y + x.y; hn[2] = y0.z + x.z; hn[3] = y0.w + x.w;
        d32_st4(rs_h, (long)r * a.H + n, make_float4(hn[0], hn[1], hn[2], hn[3]));
    }
    const float4 w4 = *reinterpret_cast<const float4*>(ln_next + n);
    d32_put4(a.img_h, a.H / 16, r, n, hn[0] * w4.x, hn[1] * w4.y, hn[2] * w4.z, hn[3] * w4.w);
    float s = (hn[0] * hn[0] + hn[1] * hn[1]) + (hn[2] * hn[2] + hn[3] * hn[3]);
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if ((it & 7) == 0) d32_st(a.ssq + (long)tile * 32 + r, s);
}

// ---- attention: one (row, kv head) unit for the whole query-head group ----------------------------------------------------
// The group's cached keys AND values are staged through LDS in chunks of D32_CH positions - one coalesced burst, requested together
// with the step's q / k / v rows - and wave h then does everything for query head h by itself: the scores (a lane per position: the
// 64 products in the per-head kernel's order), the softmax, the weighted sum of the values (a lane per dimension).  Scores of wave
// h are written and read by wave h only, so a unit of up to D32_CH positions costs two workgroup barriers.
#define D32_KP 68                          // floats per staged key row: 16-byte aligned rows whose b128 reads are conflict-free
__device__ __forceinline__ void d32_attention_unit(const D32Args& a, const Dec32Layer& ly, char* smem, int r, int hk, int tid, int wid, int lane) {
    float* sh = reinterpret_cast<float*>(smem + D32_RED);            // the scratch spans the accumulator and slice regions (88 KB)
    float* qs = sh;                               // [8][64] rotated queries
    float* knew = sh + 512;                       // this step's rotated key and its value
    float* vnew = sh + 576;
    float* sc = sh + 640;                         // [grp][max_ctx] scores, then probabilities
    const int grp = a.Hq / a.Hk, max_ctx = a.max_ctx;
    float* Kt = sc + ((grp * max_ctx + 3) & ~3);  // [D32_CH][D32_KP]
    float* Vt = Kt + D32_CH * D32_KP;             // [D32_CH][64]
    bf16_t* psb = reinterpret_cast<bf16_t*>(Vt + D32_CH * 64);       // [8 waves][192]
    const int pos = a.st[r];
    const float* row = a.qkv + (long)r * a.QKV;
    const float* Kb = ly.Kc + ((long)r * a.Hk + hk) * max_ctx * 64;
    const float* Vb = ly.Vc + ((long)r * a.Hk + hk) * max_ctx * 64;
    // requests first: this step's rows (write-through data of the qkv phase), then the first chunk of the cache
    float qv = 0.f, kv = 0.f, vv = 0.f;
    if (wid < grp) qv = d32_ld(row + (hk * grp + wid) * 64 + lane);
    if (wid == 7) { kv = d32_ld(row + (a.Hq + hk) * 64 + lane); vv = d32_ld(row + (a.Hq + a.Hk + hk) * 64 + lane); }
    constexpr int NL = D32_CH * 16 / 512;          // float4s per thread and chunk (of K, and of V)
    float4 k4[NL], v4[NL];
    const bool one = pos <= D32_CH;                // the whole context in one chunk: keys and values staged together
    auto load_chunk = [&](const float* src, int c0, float4 (&d)[NL]) {
        const int n = min(D32_CH, pos - c0);
        const float4* p = reinterpret_cast<const float4*>(src + (long)c0 * 64);
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int e = tid + i * 512;
            d[i] = e < n * 16 ? p[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load_chunk(Kb, 0, k4);
    if (one) load_chunk(Vb, 0, v4);
    const float ang = (float)pos * a.inv_freq[lane & 31];
    float sn, cs;
    sincosf(ang, &sn, &cs);
    __syncthreads();                                      // the previous unit / phase has left the scratch
    if (wid < grp) {
        const float qo = __shfl_xor(qv, 32, 64);
        qs[wid * 64 + lane] = lane < 32 ? qv * cs - qo * sn : qv * cs + qo * sn;
    }
    if (wid == 7) {
        const float ko = __shfl_xor(kv, 32, 64);
        const float kr = lane < 32 ? kv * cs - ko * sn : kv * cs + ko * sn;
        ly.Kc[((long)r * a.Hk + hk) * max_ctx * 64 + (long)pos * 64 + lane] = kr;
        ly.Vc[((long)r * a.Hk + hk) * max_ctx * 64 + (long)pos * 64 + lane] = vv;
        knew[lane] = kr;
        vnew[lane] = vv;
    }
    auto store_k = [&]() {
#pragma unroll
        for (int i = 0; i < NL; ++i) { const int e = tid + i * 512; *reinterpret_cast<float4*>(Kt + (e >> 4) * D32_KP + (e & 15) * 4) = k4[i]; }
    };
    auto store_v = [&]() {
#pragma unroll
        for (int i = 0; i < NL; ++i) reinterpret_cast<float4*>(Vt)[tid + i * 512] = v4[i];
    };
    float* sch = sc + wid * max_ctx;
    for (int c0 = 0; c0 < pos; c0 += D32_CH) {
        const int n = min(D32_CH, pos - c0);
        if (c0 > 0) { __syncthreads(); load_chunk(Kb, c0, k4); }
        store_k();
        if (one) store_v();
        __syncthreads();
        if (wid < grp) {
            const float4* q4 = reinterpret_cast<const float4*>(qs + wid * 64);
            for (int j = lane; j < n; j += 64) {
                const float4* kr4 = reinterpret_cast<const float4*>(Kt + j * D32_KP);
                float s = 0.f;
#pragma unroll
                for (int d4 = 0; d4 < 16; ++d4) {
                    const float4 q = q4[d4], k = kr4[d4];
                    s = fmaf(q.x, k.x, s); s = fmaf(q.y, k.y, s); s = fmaf(q.z, k.z, s); s = fmaf(q.w, k.w, s);
                }
                sch[c0 + j] = s * 0.125f;
            }
        }
    }
    if (pos == 0) __syncthreads();                        // knew / vnew / qs are read below
    float p_new = 0.f, sum = 1.f, acc = 0.f;
    if (wid < grp) {
        const float s_new = wave_sum(qs[wid * 64 + lane] * knew[lane]) * 0.125f;
        float mx = s_new;
        for (int j = lane; j < pos; j += 64) mx = fmaxf(mx, sch[j]);
        mx = wave_max(mx);
        float su = 0.f;
        for (int j = lane; j < pos; j += 64) {
            const float p = expf(sch[j] - mx);
            sch[j] = p;
            su += p;
        }
        p_new = expf(s_new - mx);
        sum = wave_sum(su) + p_new;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the probabilities of the other lanes are read next (same wave)
    }
    for (int c0 = 0; c0 < pos; c0 += D32_CH) {
        const int n = min(D32_CH, pos - c0);
        if (!one) {
            __syncthreads();
            load_chunk(Vb, c0, v4);
            store_v();
            __syncthreads();
        }
        if (wid < grp)
            for (int j = 0; j < n; ++j) acc = fmaf(sch[c0 + j], Vt[j * 64 + lane], acc);
    }
    if (wid < grp) {
        const int hq = hk * grp + wid;
        const float o = (acc + p_new * vnew[lane]) / sum;
        unsigned h, m, l;
        gv32_split3(o, h, m, l);
        bf16_t* ps = psb + wid * 192;
        ps[lane] = (bf16_t)h; ps[64 + lane] = (bf16_t)m; ps[128 + lane] = (bf16_t)l;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 24) {
            const int plane = lane >> 3, pc = lane & 7, col0 = hq * 64 + 8 * pc;
            const d32_u4 piece = *reinterpret_cast<const d32_u4*>(ps + plane * 64 + pc * 8);
            __builtin_amdgcn_raw_buffer_store_b128(piece, d32_desc(a.img_ao, (long)(a.H / 16) * 3072), (int)gv32_off(a.H / 16, r, col0, plane), 0, D32_SC1);
        }
    }
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------
// Per-phase copies of the thread coordinates that the compiler cannot see through: left alone it hoists every phase's address
// arithmetic out of the layer loop and keeps ~200 registers of it alive across all phases (179 spilled registers, measured).
#define D32_FRESH() int tid_p = tid, lane_p = lane, wid_p = wid; asm volatile("" : "+v"(tid_p), "+v"(lane_p), "+s"(wid_p))

// One K = hidden product phase over the column tiles first, first + stride, ... < n_tiles of weight matrix W: the operand image in
// registers, the weights a tile ahead (b0 arrives already requested: the hand-off's prefetch), the eight waves' partial tiles through
// LDS; pre(tile, item) may request what the epilogue needs (the residual) before the products; epi(tile, item, sum, pre's value)
// runs on threads 0 .. 255.
// XW (exact-weights mode): a tile's weights are W + W_lo; the two planes take turns in the SAME registers - the hi fragments' products
// run while the lo fragments fly, the lo fragments' while the next tile's hi fragments fly - so the register budget is the one-plane form's.
template <int FW, bool XW, class Pre, class Epi>
__device__ __forceinline__ void d32_product(const bf16_t* W, const bf16_t* W_lo, const bf16_t* img, int K16, int first, int stride, int n_tiles, float* red, int tid, int wid, int lane,
                                            d32_frag (&b0)[FW], Pre pre, Epi epi) {
    if (first >= n_tiles) return;
    d32_frag A[FW][3], b1[FW];
    f32x16 acc;
    d32_load_image<FW>(img, K16, A, wid, lane);
    int buf = 0;
    for (int t = first; t < n_tiles; t += stride, buf ^= 1) {
        if constexpr (XW) d32_load_w<FW>(W_lo, t, K16, b1, wid, lane);
        else d32_load_w<FW>(W, min(t + stride, n_tiles - 1), K16, b1, wid, lane);        // unconditional (clamped): no copies at a join
        float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < 256) pv = pre(t, tid);
        d32_mfma<FW>(A, b0, acc);
        if constexpr (XW) {
#pragma unroll
            for (int u = 0; u < FW; ++u) b0[u] = b1[u];
            d32_load_w<FW>(W, min(t + stride, n_tiles - 1), K16, b1, wid, lane);
            d32_mfma_lo<FW>(A, b0, acc);
        }
        d32_park(red, buf, wid, lane, acc);
        __syncthreads();
        if (tid < 256) epi(t, tid, d32_combine(red, buf, tid), pv);
#pragma unroll
        for (int u = 0; u < FW; ++u) b0[u] = b1[u];
    }
}

template <int FW, int TG, bool XW = false>
__global__ __launch_bounds__(512) void llm_decode32_k(const D32Args a) {
    extern __shared__ __attribute__((aligned(16))) char d32_smem[];
    char* smem = d32_smem;
    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = blockIdx.x;
    float* red = reinterpret_cast<float*>(smem + D32_RED);
    bf16_t* slice = reinterpret_cast<bf16_t*>(smem + D32_SLICE);
    float* rstd_s = reinterpret_cast<float*>(smem + D32_RSTD);
    if (tid == 0) *reinterpret_cast<volatile int*>(smem + D32_MISC) = __hip_atomic_load(a.status, D32_RLX, D32_AGT) != 0;
    __syncthreads();
    if (*reinterpret_cast<volatile int*>(smem + D32_MISC)) return;
    const int G = a.G, H = a.H, K16 = H / 16, K16I = a.I / 16;
    const int NTQ = a.QKV / 32, NTO = H / 32, NTH = (a.NS + 31) / 32, NP = H / 32;
    unsigned epoch = a.epoch0;
    int n_stamp = 0;
    auto stamp = [&]() { if (a.stamps && g == 0 && tid == 0) a.stamps[n_stamp++] = __builtin_amdgcn_s_memrealtime(); };
    auto none4 = [](int, int) { return make_float4(0.f, 0.f, 0.f, 0.f); };
    stamp();
    d32_frag bn[FW];                                      // the next phase's first weight fragments, requested inside the hand-off
    d32_load_w<FW>(a.layers[0].wqkv, min(g, NTQ - 1), K16, bn, wid, lane);
#pragma clang loop unroll(disable)
    for (int L = 0; L < a.n_layers; ++L) {
        // ================= P1: qkv = Wqkv (ln1 x h) * rstd + b =================
        {
            D32_FRESH();
            const Dec32Layer ly = a.layers[L];
            const d32_rsrc rs_qkv = d32_desc(a.qkv, (long)32 * a.QKV * 4);
            if (g < NTQ) d32_rstd(a.ssq, NP, a.eps, H, rstd_s, tid_p);
            d32_product<FW, XW>(ly.wqkv, ly.wqkv_lo, a.img_h, K16, g, G, NTQ, red, tid_p, wid_p, lane_p, bn, none4, [&](int t, int it, const float4& v, const float4&) {
                const int r = it >> 3, n = t * 32 + (it & 7) * 4;
                if (r < a.R) {
                    const float rs = rstd_s[r];
                    const float4 bq = *reinterpret_cast<const float4*>(ly.bqkv + n);
                    d32_st4(rs_qkv, (long)r * a.QKV + n, make_float4(v.x * rs + bq.x, v.y * rs + bq.y, v.z * rs + bq.z, v.w * rs + bq.w));
                }
            });
        }
        stamp();
        if (!d32_handoff(a, smem, g, wid, lane, epoch, [&]() {})) return;
        stamp();
        // ================= P2: attention, (row, kv head) units =================
        {
            D32_FRESH();
            const Dec32Layer ly = a.layers[L];
            for (int u = g; u < a.R * a.Hk; u += G) d32_attention_unit(a, ly, smem, u / a.Hk, u % a.Hk, tid_p, wid_p, lane_p);
        }
        stamp();
        if (!d32_handoff(a, smem, g, wid, lane, epoch, [&]() { d32_load_w<FW>(a.layers[L].wo, min(g, NTO - 1), K16, bn, wid, lane); })) return;
        stamp();
        // ================= P3: h += Wo ao; image of ln2 x h; sums of squares =================
        {
            D32_FRESH();
            const Dec32Layer ly = a.layers[L];
            const d32_rsrc rs_h = d32_desc(a.h, (long)32 * H * 4);
            d32_product<FW, XW>(ly.wo, ly.wo_lo, a.img_ao, K16, g, G, NTO, red, tid_p, wid_p, lane_p, bn,
                            [&](int t, int it) { return (it >> 3) < a.R ? d32_ld4(rs_h, (long)(it >> 3) * H + t * 32 + (it & 7) * 4) : make_float4(0.f, 0.f, 0.f, 0.f); },
                            [&](int t, int it, const float4& v, const float4& y0) { d32_add_img(a, rs_h, t, it, v, y0, ly.ln2); });
        }
        stamp();
        if (!d32_handoff(a, smem, g, wid, lane, epoch, [&]() { d32_load_w<FW>(a.layers[L].wgu, (long)TG * g, K16, bn, wid, lane); })) return;
        stamp();
        // ================= P4: gate/up tiles [TG g, TG g + TG) -> SwiGLU slice in LDS -> K slice of the down projection =================
        {
            // wave w takes the down projection's column tiles w, w + 8, ... (at most four: hidden <= 1024), NB at a time with all of their
            // weight fragments in flight; the first batch is requested before the last gate/up tile's products
            // (exact-weights mode: both planes of a batch are in flight together, so a batch is half as many tiles)
            constexpr int NB = (TG == 4 ? 4 : 2) / (XW ? 2 : 1);
            D32_FRESH();
            const Dec32Layer ly = a.layers[L];
            d32_rstd(a.ssq, NP, a.eps, H, rstd_s, tid_p);
            d32_frag wd[NB][TG], wdl[XW ? NB : 1][TG];
            auto load_wd = [&](int i0) {
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int t = min(wid_p + 8 * (i0 + i), NTO - 1);
#pragma unroll
                    for (int u = 0; u < TG; ++u) wd[i][u] = d32_ld_nt(ly.wd + ((long)t * K16I + (long)TG * g + u) * 512 + lane_p * 8);
                    if constexpr (XW) {
#pragma unroll
                        for (int u = 0; u < TG; ++u) wdl[i][u] = d32_ld_nt(ly.wd_lo + ((long)t * K16I + (long)TG * g + u) * 512 + lane_p * 8);
                    }
                }
            };
            {
                d32_frag A[FW][3];
                f32x16 acc;
                d32_load_image<FW>(a.img_h, K16, A, wid_p, lane_p);
                // one gate/up tile: products on bn, the waves' sums through LDS, SwiGLU of the 16 column pairs into the slice
                // (exact-weights mode: the gate/up weights are a stream of 2 TG half tiles - hi plane, lo plane, hi plane of the next
                // tile, ... - through the same registers; an even half starts the accumulator, an odd one continues and finishes it)
                auto tile = [&](int v) {
                    const int tl = XW ? v >> 1 : v;
                    if (XW && (v & 1)) d32_mfma_lo<FW>(A, bn, acc);
                    else d32_mfma<FW>(A, bn, acc);
                    if (XW && !(v & 1)) return;
                    d32_park(red, tl & 1, wid_p, lane_p, acc);
                    __syncthreads();
                    // waves 0-3 finish the even tiles, waves 4-7 the odd ones: the half that has no epilogue is already in the next tile's products
                    if ((tid_p >> 8) == (tl & 1)) {
                        const int it = tid_p & 255;
                        const float4 v = d32_combine(red, tl & 1, it);
                        const int r = it >> 3, c0 = (it & 7) * 4;
                        float o0 = 0.f, o1 = 0.f;
                        if (r < a.R) {
                            const float rs = rstd_s[r];
                            o0 = act_silu(v.x * rs) * (v.y * rs);
                            o1 = act_silu(v.z * rs) * (v.w * rs);
                        }
                        unsigned h0, m0, l0, h1, m1, l1;
                        gv32_split3(o0, h0, m0, l0); gv32_split3(o1, h1, m1, l1);
                        const int k = tl * 16 + (c0 >> 1);           // SwiGLU column inside the workgroup's slice (even)
                        char* base = reinterpret_cast<char*>(slice);
                        *reinterpret_cast<unsigned*>(base + gv32_off(TG, r, k, 0)) = h0 | (h1 << 16);
                        *reinterpret_cast<unsigned*>(base + gv32_off(TG, r, k, 1)) = m0 | (m1 << 16);
                        *reinterpret_cast<unsigned*>(base + gv32_off(TG, r, k, 2)) = l0 | (l1 << 16);
                    }
                };
                // weights two tiles ahead: a tile's turn (products, the trip through LDS, SwiGLU) is shorter than a trip to HBM
                d32_frag b1[FW], b2[FW];
                constexpr int NV = XW ? 2 * TG : TG;
                auto vload = [&](int v, d32_frag (&b)[FW]) {
                    if constexpr (XW) d32_load_w<FW>((v & 1) ? ly.wgu_lo : ly.wgu, (long)TG * g + (v >> 1), K16, b, wid_p, lane_p);
                    else d32_load_w<FW>(ly.wgu, (long)TG * g + v, K16, b, wid_p, lane_p);
                };
                vload(1, b1);
#pragma unroll 1
                for (int v = 0; v + 2 < NV; ++v) {
                    vload(v + 2, b2);
                    tile(v);
#pragma unroll
                    for (int u = 0; u < FW; ++u) { bn[u] = b1[u]; b1[u] = b2[u]; }
                }
                tile(NV - 2);
#pragma unroll
                for (int u = 0; u < FW; ++u) bn[u] = b1[u];
                load_wd(0);                               // the down projection's first batch flies under the last tile
                tile(NV - 1);
            }
            __syncthreads();                              // the slice is complete
            stamp();
            const d32_rsrc rs_part = d32_desc(a.part, (long)G * NTO * 4096);
#pragma unroll 1
            for (int i0 = 0; i0 < 4; i0 += NB) {
                if (wid_p + 8 * i0 >= NTO) break;
                if (i0 > 0) load_wd(i0);
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int t = wid_p + 8 * (i0 + i);
                    f32x16 acc;
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
                    for (int u = 0; u < TG; ++u)
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            const d32_frag af = *reinterpret_cast<const d32_frag*>(slice + ((u * 3 + p) * 64 + lane_p) * 8);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wd[i][u], acc, 0, 0, 0);
                            if constexpr (XW) { if (p < 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wdl[i][u], acc, 0, 0, 0); }
                        }
                    if (t < NTO) {
                        const long dst = ((long)g * NTO + t) * 1024 + lane_p * 4;
#pragma unroll
                        for (int q = 0; q < 4; ++q) d32_st4(rs_part, dst + q * 256, make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]));
                    }
                }
            }
        }
        stamp();
        if (!d32_handoff(a, smem, g, wid, lane, epoch, [&]() {
                // the next layer's qkv tile, or the head's first tile
                if (L + 1 < a.n_layers) d32_load_w<FW>(a.layers[L + 1].wqkv, min(g, NTQ - 1), K16, bn, wid, lane);
                else d32_load_w<FW>(a.w_head, min(g, NTH - 1), K16, bn, wid, lane);
            })) return;
        stamp();
        // ================= P5: h += sum over the G partial tiles (fixed order); image under the next norm weight =================
        {
            D32_FRESH();
            const float* lnn = L + 1 < a.n_layers ? a.layers[L + 1].ln1 : a.norm_w;
            const d32_rsrc rs_h = d32_desc(a.h, (long)32 * H * 4), rs_part = d32_desc(a.part, (long)G * NTO * 4096);
            float4* comb = reinterpret_cast<float4*>(red);                 // [2][256]
            float* T = red + 2048;                                         // [32][33]
            for (int t = g; t < NTO; t += G) {
                __syncthreads();
                const int half = tid_p >> 8, idx = tid_p & 255;
                const int g0 = half ? G / 2 : 0, g1 = half ? G : G / 2;
                float4 y0 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (tid_p < 256 && (tid_p >> 3) < a.R) y0 = d32_ld4(rs_h, (long)(tid_p >> 3) * H + t * 32 + (tid_p & 7) * 4);
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int gg = g0; gg < g1; gg += 10) {
                    float4 v[10];
#pragma unroll
                    for (int i = 0; i < 10; ++i) v[i] = gg + i < g1 ? d32_ld4(rs_part, ((long)(gg + i) * NTO + t) * 1024 + idx * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < 10; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
                }
                comb[half * 256 + idx] = s;
                __syncthreads();
                if (tid_p < 256) {
                    const float4 x = comb[tid_p], y = comb[256 + tid_p];
                    const int q = tid_p >> 6, l = tid_p & 63, r0 = 8 * q + 4 * (l >> 5), c = l & 31;       // accumulator layout: rows r0 .. r0+3 of column c
                    T[(r0 + 0) * 33 + c] = x.x + y.x; T[(r0 + 1) * 33 + c] = x.y + y.y;
                    T[(r0 + 2) * 33 + c] = x.z + y.z; T[(r0 + 3) * 33 + c] = x.w + y.w;
                }
                __syncthreads();
                if (tid_p < 256) {
                    const int r = tid_p >> 3, c0 = (tid_p & 7) * 4;
                    d32_add_img(a, rs_h, t, tid_p, make_float4(T[r * 33 + c0], T[r * 33 + c0 + 1], T[r * 33 + c0 + 2], T[r * 33 + c0 + 3]), y0, lnn);
                }
            }
        }
        stamp();
        if (!d32_handoff(a, smem, g, wid, lane, epoch, [&]() {})) return;
        stamp();
    }
    // ================= head: logits = W_head (norm_w x h) * rstd =================
    if (g < NTH) d32_rstd(a.ssq, NP, a.eps, H, rstd_s, tid);
    d32_product<FW, XW>(a.w_head, a.w_head_lo, a.img_h, K16, g, G, NTH, red, tid, wid, lane, bn, none4, [&](int t, int it, const float4& v, const float4&) {
        const int r = it >> 3, n = t * 32 + (it & 7) * 4;
        if (r < a.R) {
            const float rs = rstd_s[r];
            const float x[4] = {v.x * rs, v.y * rs, v.z * rs, v.w * rs};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n + j < a.NS) a.logits[(long)r * a.NS + n + j] = x[j];
        }
    });
    stamp();
}

// ---- host ---------------------------------------------------------------------------------------------------------------
struct Dec32Plan {
    Dec32Shape s;
    int G = 0, TG = 8, kind = 0;                 // kind = FW: 7 (hidden 896) or 2 (hidden 256)
    DevPool pool;
    Dec32Layer* layers = nullptr;
    const bf16_t *w_head = nullptr, *w_head_lo = nullptr;
    const float* norm_w = nullptr;
    float* part = nullptr;
    unsigned *flags = nullptr, *status = nullptr;
    unsigned epoch = 0;
    unsigned long long* stamps = nullptr;
};

static int d32_env(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

static int d32_tg(const Dec32Shape& s) {
    // TG = gate/up tiles per workgroup: 4 -> G = inter / 64 workgroups (76), 8 -> inter / 128 (38).  Measured at 32 rows, CosyVoice3-0.5B
    // (profiles/r04_llm_decode32_stamps.txt): 44.6 us per layer = 1.08 ms per token step on 76 CUs, 56.7 us = 1.37 ms on 38; the pipelined
    // benchmark step is the same for both (what the LM costs the flow decoder is its weight stream, not its CUs), so the faster one
    const int tg = d32_env("FY_LLM_DEC32_TG", 4);
    return (tg == 4 || tg == 8) && s.I % (16 * tg) == 0 ? tg : 0;
}

bool decode32_supported(const Dec32Shape& s) {
    if (!d32_env("FY_LLM_PERSISTENT32", 1)) return false;
    if (s.Hq * 64 != s.H || s.Hk < 1 || s.Hq % s.Hk != 0 || s.Hq / s.Hk > 7) return false;       // wave 7 rotates the new key
    if (!(s.H == 896 || s.H == 256) || d32_tg(s) == 0 || s.mb < 1) return false;
    const int grp = s.Hq / s.Hk;
    if ((size_t)(640 + ((grp * s.max_ctx + 3) & ~3) + D32_CH * (D32_KP + 64)) * 4 + 8 * 384 > (size_t)D32_RSTD) return false;       // the attention scratch spans `red` and `slice`
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    return s.I / (16 * d32_tg(s)) <= cus;
}

int decode32_groups(const Dec32Plan* p) { return p ? p->G : 0; }

int decode32_create(Dec32Plan** out, const Dec32Shape& s, const Dec32Layer* layers, const bf16_t* w_head, const float* norm_w, hipStream_t st,
                    const bf16_t* w_head_lo) {
    FY_CHECK(out && layers && w_head && norm_w && decode32_supported(s), FY_ERR_ARG, "decode32_create: unsupported shape");
    for (int i = 0; i < s.layers; ++i) {
        const bool lo = layers[i].wqkv_lo && layers[i].wo_lo && layers[i].wgu_lo && layers[i].wd_lo;
        const bool none = !layers[i].wqkv_lo && !layers[i].wo_lo && !layers[i].wgu_lo && !layers[i].wd_lo;
        FY_CHECK(w_head_lo ? lo : none, FY_ERR_ARG, "decode32_create: layer %d: the lo planes must be given for every matrix and the head, or for none", i);
    }
    Dec32Plan* p = new Dec32Plan();
    p->w_head_lo = w_head_lo;
    p->s = s; p->TG = d32_tg(s); p->G = s.I / (16 * p->TG); p->kind = s.H / 16 / 8;
    p->w_head = w_head; p->norm_w = norm_w;
    auto fail = [&](int rc) { delete p; return rc; };
#define TRYP(e) do { int _r = (e); if (_r) return fail(_r); } while (0)
    TRYP(p->pool.alloc(&p->layers, (size_t)s.layers));
    TRYP(p->pool.alloc(&p->part, (size_t)p->G * (s.H / 32) * 1024));
    const size_t n_flag_words = (size_t)((p->G + 3) / 4) * 4 + D32_GO + 4;
    TRYP(p->pool.alloc(&p->flags, n_flag_words)); TRYP(p->pool.alloc(&p->status, (size_t)4));
#undef TRYP
    if (hipMemcpyAsync(p->layers, layers, (size_t)s.layers * sizeof(Dec32Layer), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemsetAsync(p->flags, 0, n_flag_words * 4, st) != hipSuccess || hipMemsetAsync(p->status, 0, 16, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        fy_set_error("decode32_create: upload failed");
        return fail(FY_ERR_HIP);
    }
    for (const void* fn : {(const void*)llm_decode32_k<7, 8>, (const void*)llm_decode32_k<7, 4>, (const void*)llm_decode32_k<2, 8>, (const void*)llm_decode32_k<2, 4>,
                           (const void*)llm_decode32_k<7, 8, true>, (const void*)llm_decode32_k<7, 4, true>, (const void*)llm_decode32_k<2, 8, true>, (const void*)llm_decode32_k<2, 4, true>})
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)D32_LDS) != hipSuccess) {
            fy_set_error("decode32_create: %d bytes of LDS refused", (int)D32_LDS);
            return fail(FY_ERR_HIP);
        }
    *out = p;
    return FY_OK;
}

void decode32_destroy(Dec32Plan* p) { delete p; }

// like the 8-row persistent step - and behind the SAME event (runtime.h: persistent_chain) - launches of ALL handles are chained:
// two partly resident grids, of either kernel, would wait for each other

int decode32_step(Dec32Plan* p, int B, float* h, bf16_t* img_h, float* ssq, float* qkv, bf16_t* img_ao, const int* st_block,
                  const float* inv_freq, float* logits, hipStream_t stream) {
    FY_CHECK(p && B >= 1 && B <= 32 && h && img_h && ssq && qkv && img_ao && st_block && inv_freq && logits, FY_ERR_ARG, "decode32_step: bad arguments");
    const Dec32Shape& s = p->s;
    D32Args a;
    a.layers = p->layers; a.n_layers = s.layers; a.w_head = p->w_head; a.w_head_lo = p->w_head_lo; a.norm_w = p->norm_w;
    a.R = B; a.G = p->G; a.TG = p->TG; a.H = s.H; a.I = s.I; a.QKV = s.qkv(); a.Hq = s.Hq; a.Hk = s.Hk; a.NS = s.NS; a.max_ctx = s.max_ctx; a.eps = s.eps;
    a.h = h; a.img_h = img_h; a.ssq = ssq; a.qkv = qkv; a.img_ao = img_ao; a.part = p->part; a.logits = logits;
    a.st = st_block; a.inv_freq = inv_freq; a.flags = p->flags; a.epoch0 = p->epoch; a.status = p->status; a.stamps = p->stamps;
    p->epoch += 5u * (unsigned)s.layers;
    const double wbytes = (p->w_head_lo ? 4.0 : 2.0) * ((double)s.layers * ((double)s.qkv() * s.H + (double)s.H * s.H + 3.0 * s.I * s.H) + (double)s.NS * s.H);
    ProfScope prof("llm_decode32", wbytes, stream);
    PersistentChain& chain = persistent_chain();
    std::lock_guard<std::mutex> lk(chain.mu);
    if (!chain.ev) HIP_TRY(hipEventCreateWithFlags(&chain.ev, hipEventDisableTiming));
    else HIP_TRY(hipStreamWaitEvent(stream, chain.ev, 0));
    if (p->w_head_lo) {
        if (p->kind == 7 && p->TG == 8) hipLaunchKernelGGL((llm_decode32_k<7, 8, true>), dim3(p->G), dim3(512), D32_LDS, stream, a);
        else if (p->kind == 7) hipLaunchKernelGGL((llm_decode32_k<7, 4, true>), dim3(p->G), dim3(512), D32_LDS, stream, a);
        else if (p->TG == 8) hipLaunchKernelGGL((llm_decode32_k<2, 8, true>), dim3(p->G), dim3(512), D32_LDS, stream, a);
        else hipLaunchKernelGGL((llm_decode32_k<2, 4, true>), dim3(p->G), dim3(512), D32_LDS, stream, a);
    } else
    if (p->kind == 7 && p->TG == 8) hipLaunchKernelGGL((llm_decode32_k<7, 8>), dim3(p->G), dim3(512), D32_LDS, stream, a);
    else if (p->kind == 7) hipLaunchKernelGGL((llm_decode32_k<7, 4>), dim3(p->G), dim3(512), D32_LDS, stream, a);
    else if (p->TG == 8) hipLaunchKernelGGL((llm_decode32_k<2, 8>), dim3(p->G), dim3(512), D32_LDS, stream, a);
    else hipLaunchKernelGGL((llm_decode32_k<2, 4>), dim3(p->G), dim3(512), D32_LDS, stream, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(chain.ev, stream));
    return FY_OK;
}

int decode32_status(Dec32Plan* p, unsigned* out, hipStream_t stream) {
    HIP_TRY(hipMemcpyAsync(out, p->status, 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemsetAsync(p->status, 0, 4, stream));
    return FY_OK;
}

int decode32_stamps(Dec32Plan* p, unsigned long long* out, int n, hipStream_t stream) {
    const int cap = 2 + 12 * p->s.layers + 8;
    if (n <= 0) { p->stamps = nullptr; return FY_OK; }
    if (!p->stamps) {
        unsigned long long* d = nullptr;
        FY_TRY(p->pool.alloc(&d, (size_t)cap));
        HIP_TRY(hipMemsetAsync(d, 0, (size_t)cap * 8, stream));
        p->stamps = d;
    }
    HIP_TRY(hipMemcpyAsync(out, p->stamps, (size_t)(n < cap ? n : cap) * 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return FY_OK;
}
