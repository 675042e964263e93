// Persistent decode step of the speech-token LM (Qwen2Encoder.forward_one_step + llm_decoder for up to 8 sequences,
// CosyVoice/cosyvoice/llm/llm.py:246-258, 518): ONE launch runs the 24 layers and the head of a token step, where
// round 1 needed 121 dependent launches.  A decode step streams 728 MB of bf16 weights but its 122 phases are each a few
// microseconds of latency chain, so the design attacks the chain, not the bandwidth:
//
//   * G = inter/32 workgroups (152 for CosyVoice3-0.5B), one per CU, 8 waves each, in two roles.
//     W waves (0-3) own the weights: every weight fragment a wave will need in layer L+1 is requested (non-temporal
//       16-byte loads into registers) right after its last use in layer L, a whole layer ahead of the dependency chain, and
//       those waves touch global memory for nothing else (vector-memory operations of a wave complete in order: a wave
//       with a deep prefetch queue cannot also do a timely load).  They read the activation operand from LDS and run the MFMAs.
//     S waves (4-7) own the chain: the grid-wide hand-offs, staging the activation operand into LDS (LDS-DMA), attention,
//       and every epilogue (bias, RMSNorm scale, residual, SwiGLU, the exact 3-way bf16 split of what the next product reads).
//   * fp32-faithful products on v_mfma_f32_16x16x32_bf16: the 8 activation rows are split exactly x = hi + mid + lo
//     (3 x 8 mantissa bits) and the three planes meet each weight fragment in three MFMAs that accumulate into ONE
//     accumulator (rows 8-15 of the A operand are zero); with bf16-exact weights every product is exact and accumulation
//     is fp32, so greedy ids track the fp32 reference (same argument as gemm.hip's GEMV).
//     Producers write the split operand ("A image") in MFMA fragment order, so staging is a linear global -> LDS copy
//     and the W waves' ds_read_b128 are conflict-free.
//   * Phases of a layer: qkv (72 column tiles of 16) | attention (112 (sequence, head) units) | o-proj (56 tiles, + residual,
//     next A image) | gate/up (4 tiles per workgroup) FUSED with the down projection's K-slice of the same 32 SwiGLU columns
//     (no hand-off between them; per-workgroup partial sums of the 896 outputs) | fixed-order reduction of the G partials
//     (+ residual, next A image).  Five grid-wide hand-offs per layer.
//   * Hand-off = write-through (sc1) payload stores, drained; the workgroup raises ITS OWN flag word (sc1 store; no atomic
//     read-modify-write, no fence); one wave polls the whole flag array (G words = one 16-byte-per-lane wave load, sc1);
//     consumers read the payload with sc1 loads (MI355X_MICROARCH.md, "Valid forms").  Every spin is bounded by the real-time clock.
//     tests/micro/flag_barrier.hip prices it.
//   * All sums are in a fixed order: results do not depend on timing or placement.
#include "llm_decode.h"
#include "runtime.h"
#include <mutex>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
#define SC1 16
#define DEC_GO_WORD 32                     // the "go" word sits this many words behind the flag array (decode_create sizes and clears it)
#define PER_MAX 10                         // partials per summing group in the reduction phase: G <= 16 * PER_MAX workgroups
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
#define RLX __ATOMIC_RELAXED
#define AGT __HIP_MEMORY_SCOPE_AGENT

// Layer L's tensors are base + L * stride (one allocation per kind): pure scalar arithmetic.  A table of pointers in
// memory is read with VECTOR loads (the compiler cannot prove the table constant), and a W wave that waits for one
// waits for vmcnt(0) - its whole prefetch queue.
struct DecLayer {
    const bf16_t *wq, *wo, *wgu, *wdn;
    const float *bq, *ln1, *ln2;
    float *Kc, *Vc;
};
struct DecLayers {
    DecLayer l0;                                    // layer 0
    long s_wq, s_wo, s_wgu, s_wdn, s_bq, s_ln, s_cache;   // strides in elements
    __device__ __forceinline__ DecLayer operator[](int L) const {
        DecLayer d;
        d.wq = l0.wq + L * s_wq; d.wo = l0.wo + L * s_wo; d.wgu = l0.wgu + L * s_wgu; d.wdn = l0.wdn + L * s_wdn;
        d.bq = l0.bq + L * s_bq; d.ln1 = l0.ln1 + L * s_ln; d.ln2 = l0.ln2 + L * s_ln;
        d.Kc = l0.Kc + L * s_cache; d.Vc = l0.Vc + L * s_cache;
        return d;
    }
};

struct DecArgs {
    DecLayers layers;
    int n_layers;
    const bf16_t* w_head;
    const float* norm_w;
    int B, G, QKV, Hq, Hk, NS, NTQ, NTO, NTH, max_ctx, mb;
    float eps;
    float* hres;          // [8][H]    residual stream (in: the rows the sampler wrote)
    bf16_t* img_h;        // A image of norm_weight * h (qkv / gate-up / head operand)
    bf16_t* img_ao;       // A image of the attention output
    float* ssq_u;         // [H/8][8]  partial sums of squares per 8-column unit (P0 / P5 producers)
    float* ssq_o;         // [H/16][8] per o-proj tile
    unsigned long long* qkv;   // [8][QKV] granules {tag = epoch of this layer's qkv phase, fp32 bits}: attention polls its own inputs
    float* part;          // [G][H/8 units][8 rows][8 columns] down-projection partial sums
    float* logits;        // [B][NS]
    const int* st;        // handle state block: row 0 = positions
    const float* inv_freq;
    unsigned* flags;      // [G rounded up to 4]
    unsigned epoch0;
    unsigned* status;     // != 0: a grid hand-off timed out
    unsigned long long* stamps;   // diagnostic (null in normal use): workgroup 0 records the 100 MHz clock at every phase point
};

template <int KF_, int TQ_, int TO_, bool QO_SHARED_>
struct DecCfg {
    static constexpr int KF = KF_, TQ = TQ_, TO = TO_;
    static constexpr bool QO_SHARED = QO_SHARED_;
    static constexpr int H = KF * 32;
    static constexpr int FPT = KF / 4;                  // fragments per tile per W wave
    static constexpr int DNW = KF / 2;                  // down-projection column tiles per W wave (H/16 tiles over 4 waves)
    static constexpr int TS = TQ > TO ? TQ : TO;        // tile slots of the small-product weight buffer in LDS
    static constexpr int IMG = KF * 1536;               // bytes of one A image: per 32-deep K fragment three planes (hi, mid, lo) of 8 rows x 32 k = 512 B
    // LDS: A image (the down projection's 8 x H output block reuses it once the gate/up products have read it) | the
    // one-fragment A image of the SwiGLU slice | the W waves' partial sums (attention scratch in its place during P2) |
    // the weights of this workgroup's qkv tile (or o-proj tile) for the coming phase (LDS-DMA by the S waves) | small state
    static constexpr int OFF_IMG = 0, OFF_Y = 0, OFF_IMGD = IMG, OFF_RED = IMG + 1536, OFF_SMALL = OFF_RED + 32768, OFF_MISC = OFF_SMALL + TS * KF * 1024;
    static constexpr int LDS = OFF_MISC + 2048;
    static_assert(KF % 4 == 0, "hidden must be a multiple of 128");
};

__device__ __forceinline__ void wgb() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// the refill of a register set must stay behind the MFMAs that read it: otherwise the compiler renames it into fresh
// registers to start the loads early, and two copies of the weights do not fit the register file
__device__ __forceinline__ void w_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, RLX, AGT); }
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, RLX, AGT); }
__device__ __forceinline__ frag_ab ldw_nt(const bf16_t* p) {
    u32x4_t r = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return __builtin_bit_cast(frag_ab, r);
}
__device__ __forceinline__ void split3d(float f, bf16_t& h, bf16_t& m, bf16_t& l) {
    __bf16 hb = (__bf16)f;
    float r1 = f - (float)hb;
    __bf16 mb = (__bf16)r1;
    float r2 = r1 - (float)mb;
    __bf16 lb = (__bf16)r2;
    h = __builtin_bit_cast(unsigned short, hb);
    m = __builtin_bit_cast(unsigned short, mb);
    l = __builtin_bit_cast(unsigned short, lb);
}
// byte offset inside an A image of the 16-byte piece (plane, row s, columns col0 .. col0+7), col0 % 8 == 0
__device__ __forceinline__ int img_off(int plane, int s, int col0) {
    const int kk = col0 >> 5, q = (col0 & 31) >> 3;
    return kk * 1536 + plane * 512 + (q * 8 + s) * 16;          // = the MFMA A-operand lane (row s, k octet q) of that fragment
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }

// ---- W waves ---------------------------------------------------------------------------------------------------------
// T column tiles of one product: this wave's K fragments (kk = wid, wid+4, ...) against the staged A image; per tile the
// two accumulator blocks go to red[wave][tile][block][lane][4] for the S waves to combine.
template <int KF, int FPT, int T, int BASE, int N>
__device__ __forceinline__ void w_tiles(const frag_ab (&w)[N], const bool (&has)[T], const char* img, float* red, int wid, int lane) {
    frag_ab zf;
#pragma unroll
    for (int e = 0; e < 8; ++e) zf[e] = (__bf16)0.f;
    const bool row_ok = (lane & 15) < 8;                 // A rows 8-15 are zero
    const char* pa = img + wid * 1536 + ((lane >> 4) * 8 + (lane & 7)) * 16;
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // K fragment outermost: each A fragment is read from LDS once and meets every tile's weights.  The reads run ONE
    // fragment ahead of the MFMAs and no further (scheduling barriers): left alone the compiler hoists all of them to the
    // top - 84 registers the weights need.
    frag_ab A[2][3];
    auto rd = [&](int u, frag_ab (&dst)[3]) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            dst[p] = *reinterpret_cast<const frag_ab*>(pa + u * 4 * 1536 + p * 512);
            if (!row_ok) dst[p] = zf;
        }
    };
    rd(0, A[0]);
#pragma unroll
    for (int u = 0; u < FPT; ++u) {
        __builtin_amdgcn_sched_barrier(0);
        if (u + 1 < FPT) rd(u + 1, A[(u + 1) & 1]);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[u & 1][p], w[BASE + t * FPT + u], acc[t], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    float* rw = red + wid * 1024 + lane * 4;             // red[wave][tile][lane][4]; rows 0-7 live in lanes 0-31
#pragma unroll
    for (int t = 0; t < T; ++t)
        if (has[t] && lane < 32) *reinterpret_cast<f32x4*>(rw + t * 256) = acc[t];
}

// the same with the weights in LDS (small[tile slot][kk][1 KiB]): the qkv and o-proj products, whose few tiles per
// workgroup the S waves fetch one phase ahead
template <int KF, int FPT, int T>
__device__ __forceinline__ void w_tiles_lds(const char* small, const bool (&has)[T], const char* img, float* red, int wid, int lane) {
    frag_ab zf;
#pragma unroll
    for (int e = 0; e < 8; ++e) zf[e] = (__bf16)0.f;
    const bool row_ok = (lane & 15) < 8;
    const char* pa = img + wid * 1536 + ((lane >> 4) * 8 + (lane & 7)) * 16;
    const char* pw = small + wid * 1024 + lane * 16;
    float* rw = red + wid * 1024 + lane * 4;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        if (has[t]) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            frag_ab w[FPT];                                  // all of the tile's weight fragments in flight together
#pragma unroll
            for (int u = 0; u < FPT; ++u) w[u] = *reinterpret_cast<const frag_ab*>(pw + (t * KF + 4 * u) * 1024);
#pragma unroll
            for (int u = 0; u < FPT; ++u) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    frag_ab A = *reinterpret_cast<const frag_ab*>(pa + u * 4 * 1536 + p * 512);
                    if (!row_ok) A = zf;
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, w[u], acc, 0, 0, 0);
                }
            }
            if (lane < 32) *reinterpret_cast<f32x4*>(rw + t * 256) = acc;
        }
    }
}

template <class C>
__device__ __forceinline__ void w_main(const DecArgs& a, char* smem, const int g, const int wid, const int lane) {
    constexpr int KF = C::KF, FPT = C::FPT, TQ = C::TQ, TO = C::TO, DNW = C::DNW;
    const int G = a.G, NTO = a.NTO;
    frag_ab wgu[4 * FPT], wdn[DNW];
    bool hasq[TQ], haso[TO], hasg[4] = {true, true, true, true};
#pragma unroll
    for (int t = 0; t < TQ; ++t) hasq[t] = g + t * G < a.NTQ;
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        const int ot = C::QO_SHARED ? g - a.NTQ : g + t * G;
        haso[t] = ot >= 0 && ot < NTO;
    }
    // Weight requests: buffer loads with the layer's tensor as the (wave-uniform) descriptor, ONE lane-offset register and a
    // scalar fragment offset - no per-load 64-bit address registers (with flat addresses the compiler precomputed and
    // spilled one address pair per fragment, and every reload sat behind an s_waitcnt vmcnt(0)).
    const int voff = lane * 16;
    // the descriptor words are forced into scalar registers: a descriptor the compiler holds in vector registers turns
    // every load into a readfirstlane "waterfall" loop
    auto rsrc_of = [](const bf16_t* p, long frags) {
        const unsigned long long pu = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)pu), hi = __builtin_amdgcn_readfirstlane((unsigned)(pu >> 32));
        const int bytes = __builtin_amdgcn_readfirstlane((int)(frags * 1024));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
    };
    auto ldf = [&](__amdgpu_buffer_rsrc_t r, int frag) {
        return __builtin_bit_cast(frag_ab, __builtin_amdgcn_raw_buffer_load_b128(r, voff, __builtin_amdgcn_readfirstlane(frag * 1024), 2 /* nt */));
    };
    // Every refill is UNCONDITIONAL (tile indices clamped to valid ones, the layer index clamped to the last layer): a
    // conditionally refilled register array costs copies at every join, and the register file has none to spare.
    // `pace` > 0 spreads the requests out (s_sleep between them): a CU's share of a layer is 168 KB, and requested in one
    // burst it sits in the CU's memory pipeline in front of the S waves' stores and loads - the dependency chain - for
    // microseconds (the partial-sum stores of P4 took 4 us to ISSUE behind it).  The deadline is a whole layer away.
    const int hmax = a.NTH - 1;
    bool hash[4];
    auto pause = [&](int pace) {
        if (pace > 0) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_sleep(5); __builtin_amdgcn_sched_barrier(0); }
    };
    // gate/up tiles of layer L, or (L == n_layers) the head's tile slots 4 chunk .. 4 chunk + 3
    auto load_gu = [&](int L, int chunk, int pace) {
        const bool head = L >= a.n_layers;
        const auto r = rsrc_of(head ? a.w_head : a.layers[head ? 0 : L].wgu, head ? (long)a.NTH * KF : (long)4 * G * KF);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ht = g + (chunk * 4 + t) * G;
            hash[t] = ht < a.NTH;
            const int tile = head ? min(ht, hmax) : 4 * g + t;
#pragma unroll
            for (int u = 0; u < FPT; ++u) { wgu[t * FPT + u] = ldf(r, tile * KF + wid + 4 * u); pause(pace); }
        }
    };
    // this workgroup's K-slice of the down projection: H/16 column tiles x one fragment
    auto load_dn = [&](int L, int pace) {
        const auto r = rsrc_of(a.layers[L].wdn, (long)G * NTO);
#pragma unroll
        for (int u = 0; u < DNW; ++u) { wdn[u] = ldf(r, g * NTO + 4 * u + wid); pause(pace); }
    };
    const char* img = smem + C::OFF_IMG;
    const char* imgd = smem + C::OFF_IMGD;
    float* red = reinterpret_cast<float*>(smem + C::OFF_RED);
    float* ybuf = reinterpret_cast<float*>(smem + C::OFF_Y);
    // (an LDS word read through a generic pointer becomes a FLAT load, and a flat load waits for vmcnt(0): the whole
    // prefetch queue)
    const volatile __attribute__((address_space(3))) int* fail = (const volatile __attribute__((address_space(3))) int*)(smem + C::OFF_MISC + 32);
    const char* small = smem + C::OFF_SMALL;
    const volatile __attribute__((address_space(3))) int* drained = (const volatile __attribute__((address_space(3))) int*)(smem + C::OFF_MISC + 40);
    load_gu(0, 0, 0);
    load_dn(0, 0);
    frag_ab zf;
#pragma unroll
    for (int e = 0; e < 8; ++e) zf[e] = (__bf16)0.f;
    for (int L = 0; L < a.n_layers; ++L) {
        const int Ln = L + 1 < a.n_layers ? L + 1 : L;
        // ---- P1: qkv ----
        wgb();                                              // b1: A image staged
        if (*fail) return;
        w_tiles_lds<KF, FPT, TQ>(small, hasq, img, red, wid, lane);
        wgb();                                              // b2
        // ---- P2: attention (S waves) ----
        // ---- P3: o-proj ----
        wgb();
        if (*fail) return;
        w_tiles_lds<KF, FPT, TO>(small, haso, img, red, wid, lane);
        wgb();
        // ---- P4: gate/up, SwiGLU, this workgroup's K-slice of the down projection ----
        wgb();
        if (*fail) return;
        w_tiles<KF, FPT, 4, 0, 4 * FPT>(wgu, hasg, img, red, wid, lane);
        wgb();                                              // b2: partial sums in red
        wgb();                                              // b2a: S wrote the SwiGLU slice as a one-fragment A image
        {
            const bool row_ok = (lane & 15) < 8;
            const char* pd = imgd + ((lane >> 4) * 8 + (lane & 7)) * 16;
            frag_ab A[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                A[p] = *reinterpret_cast<const frag_ab*>(pd + p * 512);
                if (!row_ok) A[p] = zf;
            }
            float* yw = ybuf + (4 * (lane >> 4)) * C::H + wid * 16 + (lane & 15);     // row 4(l>>4)+i, column tile 4u + wid
#pragma unroll
            for (int u = 0; u < DNW; ++u) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int p = 0; p < 3; ++p) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[p], wdn[u], acc, 0, 0, 0);
                if (lane < 32) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) yw[i * C::H + u * 64] = acc[i];
                }
            }
        }
        wgb();                                              // b2b: ybuf complete
        w_fence();
        // the refill waits until the S waves' partial-sum stores have drained (an LDS word they set): the 4.3 MB of
        // partials are the one bandwidth-sensitive hand-off of the layer
        while (*drained - (L + 1) < 0) __builtin_amdgcn_s_sleep(2);
        load_gu(L + 1, 0, 1);                               // the next layer's gate/up tiles, or the head's first chunk
        load_dn(Ln, 1);
        // ---- P5: reduction of the partials (S waves) ----
    }
    // ---- head: llm_decoder over the final-norm image ----
    const int TH = (a.NTH + G - 1) / G, nch = (TH + 3) / 4;
    for (int c = 0; c < nch; ++c) {
        if (c > 0) load_gu(a.n_layers, c, 0);
        wgb();
        if (*fail) return;
        w_tiles<KF, FPT, 4, 0, 4 * FPT>(wgu, hash, img, red, wid, lane);
        wgb();
    }
}

// ---- S waves ---------------------------------------------------------------------------------------------------------
struct SCtx {
    int g, G, stid, sw, lane, gen;
    volatile __attribute__((address_space(3))) int* cnt;        // arrival counter of the S-wave barrier
    __amdgpu_buffer_rsrc_t r_flags, r_imgh, r_imgao, r_part, r_ssq_u, r_ssq_o;
};

// S-wave barrier (the four S waves only; the W waves are not held up): arrivals counted in an LDS word
__device__ __forceinline__ void sbar(volatile __attribute__((address_space(3))) int* cnt, int& gen, int lane) {
    gen += 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add((__attribute__((address_space(3))) int*)cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (*cnt - gen < 0) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

// Grid-wide hand-off, wave sw == 0 of every workgroup: raise this workgroup's flag; workgroup 0 polls the flag array (G
// words = one 16-byte-per-lane load) and then raises ONE "go" word that all the others poll - two hops, but a fifth of the
// polling traffic of everybody reading every flag, and measurably faster (tests/micro/flag_barrier.hip: 1.5 against 2.6 us
// at 152 workgroups).  Bounded by the real-time clock; false = timed out.
__device__ __forceinline__ bool grid_sync(const DecArgs& a, const SCtx& c, unsigned epoch) {
    if (c.lane == 0) __hip_atomic_store(a.flags + c.g, epoch, RLX, AGT);
    unsigned* go = a.flags + ((c.G + 3) / 4) * 4 + DEC_GO_WORD;     // a line of its own
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (c.g != 0) {
        for (;;) {
            if ((int)(__hip_atomic_load(go, RLX, AGT) - epoch) >= 0) return true;
            if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) return false;      // 1 s of the 100 MHz clock
        }
    }
    for (;;) {
        bool ok = true;
        if (c.lane * 4 < c.G) {
            const u32x4_t f = __builtin_amdgcn_raw_buffer_load_b128(c.r_flags, c.lane * 16, 0, SC1);
#pragma unroll
            for (int i = 0; i < 4; ++i) ok = ok && (c.lane * 4 + i >= c.G || (int)(f[i] - epoch) >= 0);
        }
        if (__all(ok)) {
            if (c.lane == 0) __hip_atomic_store(go, epoch, RLX, AGT);
            return true;
        }
        if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) return false;
    }
}

// end of a phase: every S wave drains its write-through stores, the S waves meet, wave 0 does the grid hand-off, they meet
// again.  On a time-out the W waves (waiting at their next barrier) are released and told to leave.
template <class C>
__device__ __forceinline__ bool s_handoff(const DecArgs& a, SCtx& c, char* smem, unsigned& epoch) {
    volatile __attribute__((address_space(3))) int* fail = (volatile __attribute__((address_space(3))) int*)(smem + C::OFF_MISC + 32);
    drain_vm();
    sbar(c.cnt, c.gen, c.lane);
    ++epoch;
    if (c.sw == 0 && !grid_sync(a, c, epoch)) {
        if (c.lane == 0) { *fail = 1; __hip_atomic_store(a.status, 1u, RLX, AGT); }
    }
    sbar(c.cnt, c.gen, c.lane);
    if (*fail) { wgb(); return false; }
    return true;
}

template <class C>
__device__ __forceinline__ void s_stage(const bf16_t* src, char* dst, const SCtx& c) {
    constexpr int NCH = C::IMG / 1024;
    for (int j = c.sw; j < NCH; j += 4)
        __builtin_amdgcn_global_load_lds((const void*)(reinterpret_cast<const char*>(src) + j * 1024 + c.lane * 16), LDSP(dst + j * 1024), 16, 0, SC1);
}

// the weights of this workgroup's tiles of a small product (qkv or o-proj), global -> LDS by DMA, one phase before the
// product: small[slot][kk][1 KiB]; 4 S waves share the KiB pieces.  Non-temporal: each piece is read once per step.
template <int KF, int T>
__device__ __forceinline__ void s_small_dma(const bf16_t* w, const int (&tile)[T], const bool (&has)[T], char* small, const SCtx& c) {
#pragma unroll
    for (int t = 0; t < T; ++t)
        if (has[t])
            for (int kk = c.sw; kk < KF; kk += 4)
                __builtin_amdgcn_global_load_lds((const void*)(w + ((long)tile[t] * KF + kk) * 512 + c.lane * 8), LDSP(small + (t * KF + kk) * 1024), 16, 0, 2);
}

// RMSNorm scale of the 8 rows from NP partial sums of squares [NP][8]: one 16-byte load per thread into LDS (all in flight
// together: chained 4-byte loads cost a memory round trip each), summed in a fixed order by 8 threads after the barrier
__device__ __forceinline__ void s_ssq_fetch(__amdgpu_buffer_rsrc_t r, int NP, float* tmp, const SCtx& c) {
    const int p = c.stid >> 1, half = c.stid & 1;
    if (p < NP) *reinterpret_cast<u32x4_t*>(tmp + p * 8 + 4 * half) = __builtin_amdgcn_raw_buffer_load_b128(r, (p * 8 + 4 * half) * 4, 0, SC1);
}
__device__ __forceinline__ void s_rstd_finish(const float* tmp, int NP, float eps, int H, float* rstd, const SCtx& c) {
    if (c.stid < 8) {
        float acc = 0.f;
        for (int p = 0; p < NP; ++p) acc += tmp[p * 8 + c.stid];
        rstd[c.stid] = rsqrtf(acc / H + eps);
    }
}

// sum of the four W waves' partial results (K split over the waves), fixed order, for output (tile slot t, row s, column c)
__device__ __forceinline__ float s_combine(const float* red, int t, int s, int col) {
    const float* b = red + t * 256 + (col + 16 * (s >> 2)) * 4 + (s & 3);
    return ((b[0] + b[1024]) + b[2048]) + b[3072];
}

// this lane's value x -> three bf16 planes in the wave's scratch (ps[plane*64 + lane]); lanes 0..23 then each own one
// 16-byte piece (plane = j / 8, the 8 consecutive lanes 8k .. 8k+7, k = j % 8) and return it
__device__ __forceinline__ u32x4_t s_piece(bf16_t* ps, float x, int lane) {
    bf16_t h, m, l;
    split3d(x, h, m, l);
    ps[lane] = h; ps[64 + lane] = m; ps[128 + lane] = l;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int j = lane < 24 ? lane : 0;
    return *reinterpret_cast<const u32x4_t*>(ps + (j >> 3) * 64 + (j & 7) * 8);
}

template <class C>
__device__ __forceinline__ void s_main(const DecArgs& a, char* smem, const int g, const int sw, const int lane) {
    constexpr int H = C::H, TQ = C::TQ, TO = C::TO;
    const int G = a.G, QKV = a.QKV, B = a.B, NTO = a.NTO;
    SCtx c;
    c.g = g; c.G = G; c.sw = sw; c.lane = lane; c.stid = sw * 64 + lane;
    c.r_flags = __builtin_amdgcn_make_buffer_rsrc(a.flags, 0, ((G + 3) / 4) * 16, 0x00020000);
    c.r_imgh = __builtin_amdgcn_make_buffer_rsrc(a.img_h, 0, C::IMG, 0x00020000);
    c.r_imgao = __builtin_amdgcn_make_buffer_rsrc(a.img_ao, 0, C::IMG, 0x00020000);
    c.r_part = __builtin_amdgcn_make_buffer_rsrc(a.part, 0, G * 8 * H * 4, 0x00020000);
    c.r_ssq_u = __builtin_amdgcn_make_buffer_rsrc(a.ssq_u, 0, (H / 8) * 32, 0x00020000);
    c.r_ssq_o = __builtin_amdgcn_make_buffer_rsrc(a.ssq_o, 0, NTO * 32, 0x00020000);
    c.gen = 0;
    c.cnt = (volatile __attribute__((address_space(3))) int*)(smem + C::OFF_MISC + 36);
    float* ssq_tmp = reinterpret_cast<float*>(smem + C::OFF_RED + 16384);   // [<= 128][8] floats, behind the W waves' 16 KiB of partial sums
    const int stid = c.stid;
    char* img = smem + C::OFF_IMG;
    bf16_t* imgd = reinterpret_cast<bf16_t*>(smem + C::OFF_IMGD);
    float* red = reinterpret_cast<float*>(smem + C::OFF_RED);
    float* ybuf = reinterpret_cast<float*>(smem + C::OFF_Y);
    float* rstd = reinterpret_cast<float*>(smem + C::OFF_MISC);
    bf16_t* ps = reinterpret_cast<bf16_t*>(smem + C::OFF_MISC + 64 + sw * 384);
    unsigned epoch = a.epoch0;
    int n_stamp = 0;
    auto stamp = [&]() {
        if (a.stamps && g == 0 && stid == 0) a.stamps[n_stamp++] = __builtin_amdgcn_s_memrealtime();
    };
    stamp();
    int qt[TQ], ot[TO];
    bool hasq[TQ], haso[TO];
#pragma unroll
    for (int t = 0; t < TQ; ++t) { qt[t] = g + t * G; hasq[t] = qt[t] < a.NTQ; }
#pragma unroll
    for (int t = 0; t < TO; ++t) {
        ot[t] = C::QO_SHARED ? g - a.NTQ : g + t * G;
        haso[t] = ot[t] >= 0 && ot[t] < NTO;
    }
    const int NU = H / 8;
    char* small = smem + C::OFF_SMALL;
    s_small_dma<C::KF, TQ>(a.layers[0].wq, qt, hasq, small, c);       // layer 0's qkv tile(s); landed by the drain before P1's barrier
    // ---- P0: h (as the sampler or the prefill left it) -> A image under layer 0's input norm + partial sums of squares ----
    if (sw == 0) {
        const float* ln = a.layers[0].ln1;
        for (int u = g; u < NU; u += G) {
            const int s = lane >> 3, col = 8 * u + (lane & 7);
            const float v = ld_sc1(a.hres + s * H + col);
            float q = v * v;
            q += __shfl_xor(q, 4, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 1, 64);
            if ((lane & 7) == 0) st_sc1(a.ssq_u + u * 8 + s, q);
            const u32x4_t pc = s_piece(ps, v * ln[col], lane);
            if (lane < 24) __builtin_amdgcn_raw_buffer_store_b128(pc, c.r_imgh, img_off(lane >> 3, lane & 7, 8 * u), 0, SC1);
        }
    }
    if (!s_handoff<C>(a, c, smem, epoch)) return;
    stamp();                                                // 1: P0 done
    const int grp = a.Hq / a.Hk;
    for (int L = 0; L < a.n_layers; ++L) {
        const DecLayer ly = a.layers[L];
        // ================= P1: qkv = Wqkv (ln1 * h) * rstd + b =================
        s_stage<C>(a.img_h, img, c);
        s_ssq_fetch(c.r_ssq_u, NU, ssq_tmp, c);
        drain_vm();
        wgb();                                              // b1
        stamp();                                            // P1 staged
        s_rstd_finish(ssq_tmp, NU, a.eps, H, rstd, c);
        wgb();                                              // b2: W waves' partial sums in red
        stamp();                                            // P1 products done
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
            if (hasq[t] && stid < 128) {
                const int s = stid >> 4, col = stid & 15, n = qt[t] * 16 + col;
                const float v = s_combine(red, t, s, col) * rstd[s] + ly.bq[n];
                __hip_atomic_store(a.qkv + s * QKV + n, ((unsigned long long)(epoch + 1) << 32) | __float_as_uint(v), RLX, AGT);
            }
        }
        stamp();                                            // P1 epilogue issued
        // No grid hand-off here: an attention unit needs the q / k / v of ONE head - 12 producer tiles - so the values travel
        // as 8-byte {tag, value} granules written by one write-through store each (the data is the flag,
        // MI355X_MICROARCH.md R2) and the unit polls its own 192 of them; no drain, no flag round trip.  The epoch still
        // advances: it is the tag.  (The S waves meet once: the partial-sum region becomes the attention scratch.)
        ++epoch;
        sbar(c.cnt, c.gen, lane);
        stamp();                                            // P1 -> P2
        // ================= P2: attention over the cache, one (sequence, head) per iteration =================
        s_small_dma<C::KF, TO>(ly.wo, ot, haso, small, c);  // this layer's o-proj tile(s): the qkv product has read the buffer (b2)
        {
            float* qs = red;                                // [64]
            float* part = red + 64;                         // [4][64]
            float* redm = red + 320;                        // [8]
            float* sc = red + 328;                          // [max_ctx]
            const int NUNIT = 8 * a.Hq;
            for (int u = g; u < NUNIT; u += G) {
                const int s = u / a.Hq, hq = u - s * a.Hq, hk = hq / grp;
                const bool act = s < B;
                const int pos = act ? a.st[s] : 0;
                // what does not depend on this step's q / k / v goes first: the rotation angles
                const float ang = (float)pos * a.inv_freq[lane & 31];
                float sn, cs;
                sincosf(ang, &sn, &cs);
                float qv = 0.f, kv = 0.f, vv = 0.f;
                if (act) {
                    const unsigned long long* row = a.qkv + (long)s * QKV;
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    for (;;) {
                        const unsigned long long gq = __hip_atomic_load(row + hq * 64 + lane, RLX, AGT);
                        const unsigned long long gk = __hip_atomic_load(row + (a.Hq + hk) * 64 + lane, RLX, AGT);
                        const unsigned long long gv = __hip_atomic_load(row + (a.Hq + a.Hk + hk) * 64 + lane, RLX, AGT);
                        qv = __uint_as_float((unsigned)gq); kv = __uint_as_float((unsigned)gk); vv = __uint_as_float((unsigned)gv);
                        if (__all((unsigned)(gq >> 32) == epoch && (unsigned)(gk >> 32) == epoch && (unsigned)(gv >> 32) == epoch)) break;
                        if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) {      // 1 s: a producer never arrived
                            if (lane == 0) {
                                *(volatile __attribute__((address_space(3))) int*)(smem + C::OFF_MISC + 32) = 1;
                                __hip_atomic_store(a.status, 2u, RLX, AGT);
                            }
                            break;
                        }
                    }
                }
                const float qo = __shfl_xor(qv, 32, 64), ko = __shfl_xor(kv, 32, 64);
                const float qr = lane < 32 ? qv * cs - qo * sn : qv * cs + qo * sn;
                const float kr = lane < 32 ? kv * cs - ko * sn : kv * cs + ko * sn;
                float* Kb = ly.Kc + ((long)s * a.Hk + hk) * a.max_ctx * 64;
                float* Vb = ly.Vc + ((long)s * a.Hk + hk) * a.max_ctx * 64;
                if (sw == 0) {
                    if (act && hq % grp == 0) {
                        Kb[(long)pos * 64 + lane] = kr;
                        Vb[(long)pos * 64 + lane] = vv;
                    }
                    qs[lane] = qr;
                }
                sbar(c.cnt, c.gen, lane);                   // a1
                // Coalesced cache reads: a wave load covers 4 positions x 64 dims (1 KiB contiguous): lane = (position
                // in the group ks, dim quad dq); wave sw takes the groups sw, sw+4, ...; 16 groups (256 positions over
                // the 4 waves) are in flight per pass.  Each lane's 16 partial dot products (one per group) are summed
                // over the 16 dim-quad lanes by a reduce-scatter butterfly (15 exchanges instead of 64): lane dq ends
                // up with the score of group dq.
                const float scale = 0.125f;
                float sq = qr * kr;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
                const float s_new = sq * scale;
                float mx = s_new;
                const int ks = lane >> 4, dq = lane & 15;
                const float4 q4 = *reinterpret_cast<const float4*>(qs + dq * 4);
#pragma nounroll
                for (int p0 = 0; p0 < pos; p0 += 256) {
                    float d[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int j = p0 + (i * 4 + sw) * 4 + ks;
                        float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (j < pos) k4 = *reinterpret_cast<const float4*>(Kb + (long)j * 64 + dq * 4);
                        d[i] = fmaf(q4.w, k4.w, fmaf(q4.z, k4.z, fmaf(q4.y, k4.y, q4.x * k4.x)));
                    }
#pragma unroll
                    for (int m = 8; m > 0; m >>= 1) {
                        const bool up = (dq & m) != 0;
#pragma unroll
                        for (int k = 0; k < m; ++k) {
                            const float send = up ? d[k] : d[k + m];
                            const float keep = up ? d[k + m] : d[k];
                            d[k] = keep + __shfl_xor(send, m, 64);
                        }
                    }
                    const int j = p0 + (dq * 4 + sw) * 4 + ks;
                    if (j < pos) {
                        const float sv = d[0] * scale;
                        sc[j] = sv;
                        mx = fmaxf(mx, sv);
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                if (lane == 0) redm[sw] = mx;
                sbar(c.cnt, c.gen, lane);                   // a2
                mx = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
                float sum = 0.f;
#pragma nounroll
                for (int j = stid; j < pos; j += 256) {
                    const float p = expf(sc[j] - mx);
                    sc[j] = p;
                    sum += p;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
                if (lane == 0) redm[4 + sw] = sum;
                sbar(c.cnt, c.gen, lane);                   // a3
                const float p_new = expf(s_new - mx);
                sum = ((redm[4] + redm[5]) + (redm[6] + redm[7])) + p_new;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma nounroll
                for (int p0 = 0; p0 < pos; p0 += 256) {
                    float4 v4[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int j = p0 + (i * 4 + sw) * 4 + ks;
                        v4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (j < pos) v4[i] = *reinterpret_cast<const float4*>(Vb + (long)j * 64 + dq * 4);
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int j = p0 + (i * 4 + sw) * 4 + ks;
                        const float pj = j < pos ? sc[j] : 0.f;
                        acc.x = fmaf(pj, v4[i].x, acc.x); acc.y = fmaf(pj, v4[i].y, acc.y);
                        acc.z = fmaf(pj, v4[i].z, acc.z); acc.w = fmaf(pj, v4[i].w, acc.w);
                    }
                }
#pragma unroll
                for (int o = 16; o < 64; o <<= 1) {
                    acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
                    acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
                }
                if (ks == 0) *reinterpret_cast<float4*>(part + sw * 64 + dq * 4) = acc;
                sbar(c.cnt, c.gen, lane);                   // a4
                if (sw == 0) {
                    const float o = (((part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane])) + p_new * vv) / sum;
                    const u32x4_t pc = s_piece(ps, o, lane);
                    if (act && lane < 24) __builtin_amdgcn_raw_buffer_store_b128(pc, c.r_imgao, img_off(lane >> 3, s, hq * 64 + 8 * (lane & 7)), 0, SC1);
                }
            }
        }
        stamp();                                            // P2 attention done
        if (!s_handoff<C>(a, c, smem, epoch)) return;
        stamp();                                            // P2 hand-off done
        // ================= P3: h += Wo ao; A image of ln2 * h; partial sums of squares per tile =================
        s_stage<C>(a.img_ao, img, c);
        drain_vm();
        wgb();                                              // b1
        stamp();                                            // P3 staged
        float hold[TO];
#pragma unroll
        for (int t = 0; t < TO; ++t) {                      // the residual is requested while the W waves multiply
            hold[t] = 0.f;
            if (haso[t] && t * 128 <= stid && stid < t * 128 + 128) {
                const int i = stid - t * 128;
                hold[t] = ld_sc1(a.hres + (i >> 4) * H + ot[t] * 16 + (i & 15));
            }
        }
        wgb();                                              // b2
        stamp();                                            // P3 products done
#pragma unroll
        for (int t = 0; t < TO; ++t) {
            // 128 outputs per tile: threads [128 t', 128 t' + 128) of the 256, tile slots two at a time
            if (haso[t] && (t & 1) * 128 <= stid && stid < (t & 1) * 128 + 128) {
                const int i = stid - (t & 1) * 128, s = i >> 4, col = i & 15, n = ot[t] * 16 + col;
                const float hn = hold[t] + s_combine(red, t, s, col);
                st_sc1(a.hres + s * H + n, hn);
                float q = hn * hn;
                q += __shfl_xor(q, 8, 64); q += __shfl_xor(q, 4, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 1, 64);
                if (col == 0) st_sc1(a.ssq_o + ot[t] * 8 + s, q);
                const u32x4_t pc = s_piece(ps, hn * ly.ln2[n], lane);
                // this wave's 64 lanes = rows s0 .. s0+3 (s0 = s of lane 0) x 16 columns: piece k covers lanes 8k..8k+7
                const int s0 = (i - lane) >> 4;
                if (lane < 24) __builtin_amdgcn_raw_buffer_store_b128(pc, c.r_imgh, img_off(lane >> 3, s0 + ((lane & 7) >> 1), ot[t] * 16 + 8 * (lane & 1)), 0, SC1);
            }
        }
        stamp();                                            // P3 epilogue issued
        if (!s_handoff<C>(a, c, smem, epoch)) return;
        stamp();                                            // P3 hand-off done
        // ================= P4: gate/up -> SwiGLU slice -> K-slice of the down projection -> partial sums =================
        s_stage<C>(a.img_h, img, c);
        s_ssq_fetch(c.r_ssq_o, NTO, ssq_tmp, c);
        drain_vm();
        wgb();                                              // b1
        stamp();                                            // P4 staged
        s_rstd_finish(ssq_tmp, NTO, a.eps, H, rstd, c);
        wgb();                                              // b2
        stamp();                                            // P4 gate/up products done
        {
            const int t = stid >> 6, s = (stid & 63) >> 3, j = stid & 7;
            const float gt = s_combine(red, t, s, 2 * j) * rstd[s], up = s_combine(red, t, s, 2 * j + 1) * rstd[s];
            bf16_t h, m, l;
            split3d(silu_f(gt) * up, h, m, l);
            // one K fragment: column t*8 + j -> octet t, element j
            imgd[(t * 8 + s) * 8 + j] = h;
            imgd[256 + (t * 8 + s) * 8 + j] = m;
            imgd[512 + (t * 8 + s) * 8 + j] = l;
        }
        wgb();                                              // b2a
        stamp();                                            // P4 SwiGLU slice written
        wgb();                                              // b2b: ybuf = this workgroup's partial of the 8 x H outputs
        stamp();                                            // P4 down products done
        {
            // stored per 8-column unit: part[g][unit][8 rows][8 columns], so that the reduction reads whole 128-byte lines
            // (with rows of H floats a reducer's 32-byte pieces sat in 40 different lines per wave load)
            for (int i = stid; i < 8 * H / 4; i += 256) {
                const int u = i >> 4, r = (i & 15) >> 1, half = i & 1;
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(ybuf + r * H + 8 * u + 4 * half);
                __builtin_amdgcn_raw_buffer_store_b128(v, c.r_part, (g * 8 * H + i * 4) * 4, 0, SC1);
            }
        }
        stamp();                                            // P4 partial stores issued
        drain_vm();
        if (lane == 0) *(volatile __attribute__((address_space(3))) int*)(smem + C::OFF_MISC + 40) = L + 1;      // the W waves may start their refill
        if (!s_handoff<C>(a, c, smem, epoch)) return;
        stamp();                                            // P4 hand-off done
        // ================= P5: h += sum over workgroups of the partials (fixed order); next A image =================
        if (L + 1 < a.n_layers) s_small_dma<C::KF, TQ>(a.layers[L + 1].wq, qt, hasq, small, c);      // the o-proj product has read the buffer
        {
            float* psum = red;                              // [16 groups][16 positions][4]
            const float* lnn = L + 1 < a.n_layers ? a.layers[L + 1].ln1 : a.norm_w;
            const int per = (G + 15) / 16;
            for (int u = g; u < NU; u += G) {
                const int pg = stid >> 4, p16 = stid & 15, s = p16 >> 1, half = p16 & 1;
                const int g0 = pg * per, g1 = min(G, g0 + per);
                f32x4 v[PER_MAX];
#pragma unroll
                for (int i = 0; i < PER_MAX; ++i)
                    if (i < per && g0 + i < g1)
                        v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(c.r_part, ((g0 + i) * 8 * H + u * 64 + s * 8 + 4 * half) * 4, 0, SC1));
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < PER_MAX; ++i)
                    if (i < per && g0 + i < g1) acc += v[i];
                *reinterpret_cast<f32x4*>(psum + (pg * 16 + p16) * 4) = acc;
                float hprev = 0.f;
                if (sw == 0) hprev = ld_sc1(a.hres + (lane >> 3) * H + 8 * u + (lane & 7));
                sbar(c.cnt, c.gen, lane);                   // c1
                if (sw == 0) {
                    const int s2 = lane >> 3, cc = lane & 7;
                    float tot = 0.f;
#pragma unroll
                    for (int k = 0; k < 16; ++k) tot += psum[(k * 16 + s2 * 2 + (cc >> 2)) * 4 + (cc & 3)];
                    const float hn = hprev + tot;
                    st_sc1(a.hres + s2 * H + 8 * u + cc, hn);
                    float q = hn * hn;
                    q += __shfl_xor(q, 4, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 1, 64);
                    if (cc == 0) st_sc1(a.ssq_u + u * 8 + s2, q);
                    const u32x4_t pc = s_piece(ps, hn * lnn[8 * u + cc], lane);
                    if (lane < 24) __builtin_amdgcn_raw_buffer_store_b128(pc, c.r_imgh, img_off(lane >> 3, lane & 7, 8 * u), 0, SC1);
                }
                sbar(c.cnt, c.gen, lane);                   // c2
            }
        }
        stamp();                                            // P5 reduction done
        if (!s_handoff<C>(a, c, smem, epoch)) return;
        stamp();                                            // P5 hand-off done
    }
    // ================= head: logits = W_head (norm_w * h) * rstd =================
    const int TH = (a.NTH + G - 1) / G, nch = (TH + 3) / 4;
    for (int ch = 0; ch < nch; ++ch) {
        if (ch == 0) {
            s_stage<C>(a.img_h, img, c);
            s_ssq_fetch(c.r_ssq_u, NU, ssq_tmp, c);
            drain_vm();
        }
        wgb();
        if (ch == 0) s_rstd_finish(ssq_tmp, NU, a.eps, H, rstd, c);
        wgb();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int tile = g + (ch * 4 + t) * G;
            if (tile < a.NTH && (t & 1) * 128 <= stid && stid < (t & 1) * 128 + 128) {
                const int i = stid - (t & 1) * 128, s = i >> 4, col = i & 15, n = tile * 16 + col;
                if (s < B && n < a.NS) a.logits[(long)s * a.NS + n] = s_combine(red, t, s, col) * rstd[s];
            }
        }
        stamp();                                            // head chunk done
    }
}

template <class C>
__global__ __launch_bounds__(512, 2) void llm_decode_k(const DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) char dec_smem[];
    const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = blockIdx.x;
    if (tid == 0) {
        // a time-out of an earlier launch that nobody has read yet (fy_llm_step looks every 8 steps): leave at once instead
        // of waiting out the spin bound again
        *(volatile __attribute__((address_space(3))) int*)(dec_smem + C::OFF_MISC + 32) = __hip_atomic_load(a.status, RLX, AGT) != 0;       // fail flag
        *(volatile __attribute__((address_space(3))) int*)(dec_smem + C::OFF_MISC + 36) = 0;       // S-wave barrier counter
        *(volatile __attribute__((address_space(3))) int*)(dec_smem + C::OFF_MISC + 40) = 0;       // layers whose partial sums have drained
    }
    __syncthreads();
    if (*(volatile __attribute__((address_space(3))) int*)(dec_smem + C::OFF_MISC + 32)) return;
    if (wid < 4) w_main<C>(a, dec_smem, g, wid, lane);
    else s_main<C>(a, dec_smem, g, wid - 4, lane);
}

typedef DecCfg<28, 1, 1, true> CfgFull;          // CosyVoice3-0.5B: hidden 896, 72 qkv tiles + 56 o tiles <= 152 workgroups
typedef DecCfg<8, 2, 1, false> CfgTiny;          // the reduced-size twin of the tests: hidden 256, 16 workgroups

// ---- packing: fp32 [N][K] -> B fragments of v_mfma_f32_16x16x32_bf16: lane l, element j = W[16 tile + (l & 15)][32 kk + 8 (l >> 4) + j].
// tile-major [tile][kk][512] (a tile's K fragments contiguous) or K-major [kk][tile][512] (a K slice's column tiles contiguous)
__global__ void dec_pack_k(const float* __restrict__ src, bf16_t* __restrict__ dst, int N, int K, int kmajor) {
    const int KF = K / 32, NT = (N + 15) / 16;
    const long total = (long)NT * KF * 512;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int j = i & 7, l = (i >> 3) & 63;
        const long f = i >> 9;
        const int kk = kmajor ? (int)(f / NT) : (int)(f % KF), tile = kmajor ? (int)(f % NT) : (int)(f / KF);
        const int n = tile * 16 + (l & 15), k = kk * 32 + 8 * (l >> 4) + j;
        dst[i] = n < N ? f32_to_bf16(src[(long)n * K + k]) : (bf16_t)0;
    }
}

struct DecodePlan {
    DecodeShape s;
    int G = 0, NTQ = 0, NTO = 0, NTH = 0, kind = 0;         // kind 1 full, 2 tiny
    DevPool pool;
    DecLayers layers;
    bf16_t* w_head = nullptr;
    float* norm_w = nullptr;
    bf16_t *img_h = nullptr, *img_ao = nullptr;
    float *ssq_u = nullptr, *ssq_o = nullptr, *part = nullptr;
    unsigned long long* qkv = nullptr;
    unsigned *flags = nullptr, *status = nullptr;
    unsigned epoch = 0;
    size_t lds = 0;
    unsigned long long* stamps = nullptr;                   // allocated by decode_stamps on first use
};

static int env_flag(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

static int plan_kind(const DecodeShape& s) {
    if (s.Hq * 64 != s.H || s.I % 32 != 0 || s.Hk < 1 || s.Hq % s.Hk != 0) return 0;
    const int G = s.I / 32, NTQ = s.qkv() / 16, NTO = s.H / 16;
    if (G > 16 * PER_MAX) return 0;
    if (s.H == CfgFull::H && NTQ + NTO <= G) return 1;
    if (s.H == CfgTiny::H && NTQ <= CfgTiny::TQ * G && NTO <= CfgTiny::TO * G) return 2;
    return 0;
}

bool decode_supported(const DecodeShape& s) {
    if (!env_flag("FY_LLM_PERSISTENT", 1)) return false;
    if (plan_kind(s) == 0) return false;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    return s.I / 32 <= cus;                                  // one workgroup per CU, all resident at once
}

size_t decode_bytes(const DecodePlan* p) { return p ? p->pool.bytes : 0; }

int decode_create(DecodePlan** out, const DecodeShape& s, hipStream_t st) {
    DecodePlan* p = new DecodePlan();
    p->s = s;
    p->kind = plan_kind(s);
    p->G = s.I / 32; p->NTQ = s.qkv() / 16; p->NTO = s.H / 16; p->NTH = (s.NS + 15) / 16;
    auto fail = [&](int rc) { delete p; return rc; };
    if (p->kind == 0) { fy_set_error("decode_create: unsupported shape"); return fail(FY_ERR_ARG); }
    const int KF = s.H / 32, IMG = KF * 1536;
    p->lds = p->kind == 1 ? CfgFull::LDS : CfgTiny::LDS;
    const size_t att = (size_t)(328 + s.max_ctx) * 4;       // attention scratch lives in the partial-sum region
    if (att > (size_t)32768) { fy_set_error("decode_create: context %d too long for the score buffer", s.max_ctx); return fail(FY_ERR_ARG); }
#define TRYP(e) do { int _r = (e); if (_r) return fail(_r); } while (0)
    {
        DecLayers& ly = p->layers;
        ly.s_wq = (long)p->NTQ * KF * 512; ly.s_wo = (long)p->NTO * KF * 512; ly.s_wgu = (long)(2 * s.I / 16) * KF * 512;
        ly.s_wdn = (long)p->NTO * (s.I / 32) * 512; ly.s_bq = s.qkv(); ly.s_ln = s.H; ly.s_cache = 0;
        bf16_t *wq, *wo, *wgu, *wdn;
        float *bq, *l1, *l2;
        TRYP(p->pool.alloc(&wq, (size_t)ly.s_wq * s.layers)); TRYP(p->pool.alloc(&wo, (size_t)ly.s_wo * s.layers));
        TRYP(p->pool.alloc(&wgu, (size_t)ly.s_wgu * s.layers)); TRYP(p->pool.alloc(&wdn, (size_t)ly.s_wdn * s.layers));
        TRYP(p->pool.alloc(&bq, (size_t)ly.s_bq * s.layers)); TRYP(p->pool.alloc(&l1, (size_t)ly.s_ln * s.layers)); TRYP(p->pool.alloc(&l2, (size_t)ly.s_ln * s.layers));
        ly.l0.wq = wq; ly.l0.wo = wo; ly.l0.wgu = wgu; ly.l0.wdn = wdn; ly.l0.bq = bq; ly.l0.ln1 = l1; ly.l0.ln2 = l2; ly.l0.Kc = nullptr; ly.l0.Vc = nullptr;
    }
    TRYP(p->pool.alloc(&p->w_head, (size_t)p->NTH * KF * 512));
    TRYP(p->pool.alloc(&p->norm_w, (size_t)s.H));
    TRYP(p->pool.alloc(&p->img_h, (size_t)IMG / 2)); TRYP(p->pool.alloc(&p->img_ao, (size_t)IMG / 2));
    TRYP(p->pool.alloc(&p->ssq_u, (size_t)(s.H / 8) * 8)); TRYP(p->pool.alloc(&p->ssq_o, (size_t)p->NTO * 8));
    TRYP(p->pool.alloc(&p->qkv, (size_t)8 * s.qkv())); TRYP(p->pool.alloc(&p->part, (size_t)p->G * 8 * s.H));
    // the flag words (G rounded up to a 16-byte load) and, a line further on, the "go" word grid_sync polls: all of it zeroed
    // (hipMalloc does not clear, and a stale "go" of a destroyed plan would let every workgroup run through its hand-offs)
    const size_t n_flag_words = (size_t)((p->G + 3) / 4) * 4 + DEC_GO_WORD + 4;
    TRYP(p->pool.alloc(&p->flags, n_flag_words)); TRYP(p->pool.alloc(&p->status, (size_t)4));
#undef TRYP
    if (hipMemsetAsync(p->flags, 0, n_flag_words * 4, st) != hipSuccess || hipMemsetAsync(p->status, 0, 16, st) != hipSuccess ||
        hipMemsetAsync(p->img_h, 0, IMG, st) != hipSuccess || hipMemsetAsync(p->img_ao, 0, IMG, st) != hipSuccess ||
        hipMemsetAsync(p->part, 0, (size_t)p->G * 8 * s.H * 4, st) != hipSuccess || hipMemsetAsync(p->qkv, 0, (size_t)8 * s.qkv() * 8, st) != hipSuccess) {
        fy_set_error("decode_create: memset failed");
        return fail(FY_ERR_HIP);
    }
    const void* fn = p->kind == 1 ? (const void*)llm_decode_k<CfgFull> : (const void*)llm_decode_k<CfgTiny>;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds) != hipSuccess) {
        fy_set_error("decode_create: %zu bytes of LDS refused", p->lds);
        return fail(FY_ERR_HIP);
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 512, p->lds) != hipSuccess || per_cu < 1) {
        fy_set_error("decode_create: the decode kernel does not fit a CU");
        return fail(FY_ERR_HIP);
    }
    *out = p;
    return FY_OK;
}

void decode_destroy(DecodePlan* p) { delete p; }

static int pack(const float* src, const bf16_t* dst, int N, int K, int kmajor, hipStream_t st) {
    const size_t total = (size_t)((N + 15) / 16) * (K / 32) * 512, blocks = (total + 255) / 256;
    hipLaunchKernelGGL(dec_pack_k, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, st, src, const_cast<bf16_t*>(dst), N, K, kmajor);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

int decode_pack_layer(DecodePlan* p, int i, const DecodeLayerSrc& src, hipStream_t st) {
    FY_CHECK(p && i >= 0 && i < p->s.layers, FY_ERR_ARG, "decode_pack_layer: bad layer");
    const DecodeShape& s = p->s;
    DecLayers& ly = p->layers;
    FY_TRY(pack(src.wqkv, ly.l0.wq + i * ly.s_wq, s.qkv(), s.H, 0, st));
    FY_TRY(pack(src.wo, ly.l0.wo + i * ly.s_wo, s.H, s.H, 0, st));
    FY_TRY(pack(src.wgu, ly.l0.wgu + i * ly.s_wgu, 2 * s.I, s.H, 0, st));
    FY_TRY(pack(src.wd, ly.l0.wdn + i * ly.s_wdn, s.H, s.I, 1, st));
    HIP_TRY(hipMemcpyAsync(const_cast<float*>(ly.l0.bq + i * ly.s_bq), src.bqkv, (size_t)s.qkv() * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(const_cast<float*>(ly.l0.ln1 + i * ly.s_ln), src.ln1, (size_t)s.H * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(const_cast<float*>(ly.l0.ln2 + i * ly.s_ln), src.ln2, (size_t)s.H * 4, hipMemcpyDeviceToDevice, st));
    // the caches of the layers are one allocation with a fixed stride (llm.hip)
    if (i == 0) { ly.l0.Kc = src.Kc; ly.l0.Vc = src.Vc; }
    else if (i == 1) ly.s_cache = (long)(src.Kc - ly.l0.Kc);
    FY_CHECK(i < 2 || (src.Kc == ly.l0.Kc + i * ly.s_cache && src.Vc == ly.l0.Vc + i * ly.s_cache), FY_ERR_ARG, "decode_pack_layer: caches must be equally spaced");
    return FY_OK;
}

int decode_pack_head(DecodePlan* p, const float* w_head, const float* norm_w, hipStream_t st) {
    FY_CHECK(p && w_head && norm_w, FY_ERR_ARG, "decode_pack_head: null argument");
    FY_TRY(pack(w_head, p->w_head, p->s.NS, p->s.H, 0, st));
    HIP_TRY(hipMemcpyAsync(p->norm_w, norm_w, (size_t)p->s.H * 4, hipMemcpyDeviceToDevice, st));
    return FY_OK;
}

// Two persistent grids that are each only partly resident would wait for each other's CUs forever (until the spin bound),
// so persistent launches of ALL handles of a device - this kernel's and llm_decode32_k's - are chained (runtime.h:
// persistent_chain): a launch waits (on the GPU) for the previous one.

int decode_step(DecodePlan* p, int B, float* h, const int* st_block, const float* inv_freq, float* logits, hipStream_t stream) {
    FY_CHECK(p && B >= 1 && B <= 8 && h && st_block && inv_freq && logits, FY_ERR_ARG, "decode_step: bad arguments");
    const DecodeShape& s = p->s;
    DecArgs a;
    a.layers = p->layers; a.n_layers = s.layers; a.w_head = p->w_head; a.norm_w = p->norm_w;
    a.B = B; a.G = p->G; a.QKV = s.qkv(); a.Hq = s.Hq; a.Hk = s.Hk; a.NS = s.NS; a.NTQ = p->NTQ; a.NTO = p->NTO; a.NTH = p->NTH;
    a.max_ctx = s.max_ctx; a.mb = s.mb; a.eps = s.eps;
    a.hres = h; a.img_h = p->img_h; a.img_ao = p->img_ao; a.ssq_u = p->ssq_u; a.ssq_o = p->ssq_o; a.qkv = p->qkv; a.part = p->part;
    a.logits = logits; a.st = st_block; a.inv_freq = inv_freq; a.flags = p->flags; a.epoch0 = p->epoch; a.status = p->status; a.stamps = p->stamps;
    p->epoch += 1u + 5u * (unsigned)s.layers;               // hand-offs per launch: P0 + five per layer
    // bench.py's roofline leg: algorithmic bytes of a token step = every bf16 weight once (the launch streams nothing twice)
    const double wbytes = 2.0 * ((double)s.layers * ((double)s.qkv() * s.H + (double)s.H * s.H + 3.0 * s.I * s.H) + (double)s.NS * s.H);
    ProfScope prof("llm_decode", wbytes, stream);
    PersistentChain& chain = persistent_chain();
    std::lock_guard<std::mutex> lk(chain.mu);
    if (!chain.ev) HIP_TRY(hipEventCreateWithFlags(&chain.ev, hipEventDisableTiming));
    else HIP_TRY(hipStreamWaitEvent(stream, chain.ev, 0));
    if (p->kind == 1) hipLaunchKernelGGL(llm_decode_k<CfgFull>, dim3(p->G), dim3(512), p->lds, stream, a);
    else hipLaunchKernelGGL(llm_decode_k<CfgTiny>, dim3(p->G), dim3(512), p->lds, stream, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(chain.ev, stream));
    return FY_OK;
}

// != 0 once a hand-off of some launch on this plan has timed out since the last call (read after a stream synchronisation);
// the word is cleared behind the copy, so one time-out is reported once and the handle stays usable on either decode path
int decode_status(DecodePlan* p, unsigned* out, hipStream_t stream) {
    HIP_TRY(hipMemcpyAsync(out, p->status, 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemsetAsync(p->status, 0, 4, stream));
    return FY_OK;
}

// diagnostic: n > 0 arms the recording of phase time stamps (workgroup 0, 100 MHz clock) for the following launches and
// copies the stamps of the last launch into `out` (18 per layer + 2 + head chunks); n == 0 disarms
int decode_stamps(DecodePlan* p, unsigned long long* out, int n, hipStream_t stream) {
    const int cap = 2 + 18 * p->s.layers + 64;
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
