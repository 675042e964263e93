// Attention kernels.
//   dit_attention   : non-causal multi-head attention of the DiT estimator (16 heads x 64, bf16 in,
//                     fp32 online softmax, bf16 out) with a key-padding mask and the optional
//                     block-causal chunk mask of streaming inference
//                     (CosyVoice/cosyvoice/flow/DiT/modules.py:349-407; utils/mask.py:127-158).
//   llm_attention   : fp32 grouped-query causal attention over the KV cache, one wave per
//                     (query row, query head); serves prefill rows and decode steps alike
//                     (transformers Qwen2Attention as called from llm/llm.py:246-258).
#include "attn.h"
#include "runtime.h"
#include "gemv32.h"
#include <algorithm>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;

#define AT_D 64
#define AT_KP 72             // bf16 elements per LDS row of the K tile (64 + 8 pad: conflict-free ds_read_b128 rows)
#define AT_VP 96             // V tile pitch: 48 dwords, so the four rows of a transposed 4x16 block sit 16 banks apart

typedef __attribute__((ext_vector_type(4))) short s16x4;

// The scores are computed TRANSPOSED, S^T = K Q^T (v_mfma_f32_32x32x16_bf16: A = K rows out of LDS, B = Q^T straight
// from global memory), so a lane owns one query (its column) and 32 of the tile's 64 keys (its registers; the other
// 32 are on lane^32): softmax statistics are per-lane scalars with one cross-half exchange, and the probabilities,
// rounded to bf16 in place, ARE the B operand of O^T = V^T P^T -- no LDS round trip for P.  The k order of that second
// product follows the accumulator's register order (key 16s + 8(j>>2) + 4h + (j&3) in element j of lane half h); the
// A operand V^T is read to match, transposed by the LDS itself (ds_read_b64_tr_b16 on the row-major V tile).
// qkv: bf16 [nseq*Tmax][3*H*64] as [q | k | v]; out: bf16 [nseq*Tmax][H*64]
template <int AT_NW, int WPE>
__global__ __launch_bounds__(AT_NW * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void dit_attention_k(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                              const int* __restrict__ seq_len, int Tmax, int H, int chunk, float scale_log2, int q_begin, int seq_rows, int row_step) {
    __shared__ __attribute__((aligned(16))) bf16_t Kbuf[2][64 * AT_KP];      // two tiles: the next one is written while this one is read,
    __shared__ __attribute__((aligned(16))) bf16_t Vbuf[2][64 * AT_VP];      // one barrier per key tile
    // the many-wave form runs at 4 waves per SIMD (128 registers): the query fragments wait in LDS instead of 16 registers
    constexpr bool QL = AT_NW > 4;
    __shared__ __attribute__((aligned(16))) bf16_t Qbuf[QL ? AT_NW * 32 * AT_KP : 8];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int s = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * (AT_NW * 32);
    const int len = seq_len[s];
    if (q0 >= len || q0 + AT_NW * 32 <= q_begin) return;     // q_begin: only the queries from that row on are wanted (incremental streaming)
    const int ld = 3 * H * AT_D;
    // row t of sequence s is row s * seq_rows + t * row_step of the tensors (s Tmax + t, or 2 t + s: the sequences interleaved)
    const long ldr = (long)ld * row_step;
    const bf16_t* base = qkv + (long)s * seq_rows * ld;
    const int lr = lane & 31, hf = lane >> 5;
    const int qrow = q0 + wid * 32 + lr;                     // this lane's query
    frag_ab qf[4];                                           // B[k = 16ks + 8hf + j][col = query]
    bf16_t* Qs = Qbuf + (QL ? (wid * 32 + lr) * AT_KP + hf * 8 : 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = *reinterpret_cast<const frag_ab*>(base + (long)min(qrow, len - 1) * ldr + h * AT_D + ks * 16 + hf * 8);
        if (QL) *reinterpret_cast<frag_ab*>(Qs + ks * 16) = qf[ks];      // read back by the same lane only
    }
    f32x16 o[2];                                             // O^T[d = 32dt + (r&3) + 8(r>>2) + 4hf][query]
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    // keys this block may need: everything (non-causal) or up to the end of the last query's chunk
    int kend = len;
    if (chunk > 0) kend = min(len, ((min(q0 + AT_NW * 32 - 1, len - 1) / chunk) + 1) * chunk);
    const int lim = chunk > 0 ? min(len, ((min(qrow, len - 1) / chunk) + 1) * chunk) : len;   // keys < lim are visible to this query
    // K/V tiles are fetched one tile ahead into registers: the loads of tile t+1 fly under the MFMAs of tile t.  A tile is
    // 2 x 512 16-byte chunks (64 keys x 8 chunks of K, then of V); chunk c = tid + i * threads
    constexpr int NT = AT_NW * 64, NLD = (1024 + NT - 1) / NT;
    uint4 kvreg[NLD];
    auto load_kv = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * NT, idx = c & 511, key = idx >> 3, dc = (idx & 7) * 8;
            if (1024 % NT == 0 || c < 1024) {
                const bool ok = k0 + key < len;
                const bf16_t* p = base + (long)min(k0 + key, len - 1) * ldr + h * AT_D + dc + (c < 512 ? 1 : 2) * H * AT_D;
                kvreg[i] = *reinterpret_cast<const uint4*>(p);
                if (!ok) kvreg[i] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    // transposed V read: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4-key x 16-d block and
    // receives column (lane & 15) of its four keys
    const int gi = lane >> 4, li = lane & 15;
    const int v_off = (4 * hf + (li >> 2)) * AT_VP + 16 * (gi & 1) + 4 * (li & 3);
    auto store_kv = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * NT, idx = c & 511, key = idx >> 3, dc = (idx & 7) * 8;
            if (1024 % NT == 0 || c < 1024) {
                if (c < 512) *reinterpret_cast<uint4*>(Kbuf[buf] + key * AT_KP + dc) = kvreg[i];
                else *reinterpret_cast<uint4*>(Vbuf[buf] + key * AT_VP + dc) = kvreg[i];
            }
        }
    };
    const bool wave_live = q0 + wid * 32 < len;              // a wave past the sequence end only helps staging
    load_kv(0);
    store_kv(0);
    __syncthreads();
    for (int k0 = 0, it = 0; k0 < kend; k0 += 64, ++it) {
        const bf16_t* Ks = Kbuf[it & 1];
        const bf16_t* Vs = Vbuf[it & 1];
        const bool more = k0 + 64 < kend;
        if (more) load_kv(k0 + 64);
        if (wave_live) {
        // S^T: sc[kt][r] = score of key k0 + 32kt + (r&3) + 8(r>>2) + 4hf against this lane's query
        f32x16 sc[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                frag_ab kf = *reinterpret_cast<const frag_ab*>(Ks + (kt * 32 + lr) * AT_KP + ks * 16 + hf * 8);
                const frag_ab q = QL ? *reinterpret_cast<const frag_ab*>(Qs + ks * 16) : qf[ks];
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, q, sc[kt], 0, 0, 0);
            }
        }
        frag_ab pf[2][2];                                    // P^T as the B operand: [kt][k-step s] = registers 8s .. 8s+7
        float alpha, ps = 0.f;
        if (chunk == 0 && k0 + 64 <= len) {
            // every key of the tile is visible to every query: no masks, the scale rides in the exponent's fma
            float mx = sc[0][0];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * scale_log2;
            const float mnew = fmaxf(m_run, mx);
            alpha = __builtin_amdgcn_exp2f(m_run - mnew);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], scale_log2, -mnew));
                    ps += p;
                    pf[kt][r >> 3][r & 7] = (__bf16)p;
                }
            m_run = mnew;
        } else {
            float mx = -1e30f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
                    const float v = key < lim ? sc[kt][r] * scale_log2 : -1e30f;
                    sc[kt][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mnew = fmaxf(m_run, mx);
            alpha = __builtin_amdgcn_exp2f(m_run - mnew);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = sc[kt][r] > -1e29f ? __builtin_amdgcn_exp2f(sc[kt][r] - mnew) : 0.f;
                    ps += p;
                    pf[kt][r >> 3][r & 7] = (__bf16)p;
                }
            m_run = mnew;
        }
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {     // the running maximum settles after a few tiles
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
        // O^T += V^T P^T
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const bf16_t* vb = Vs + (kt * 32 + st * 16) * AT_VP + dt * 32 + v_off;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + 8 * AT_VP));
                    bf16x8 vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(frag_ab, vv), pf[kt][st], o[dt], 0, 0, 0);
                }
        }
        if (more) store_kv((it + 1) & 1);                    // that buffer was last read before the previous barrier
        __syncthreads();
    }
    if (qrow >= len || qrow < q_begin) return;
    const float inv = 1.f / l_run;
    bf16_t* op = out + ((long)s * seq_rows + (long)qrow * row_step) * (H * AT_D) + h * AT_D;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(o[dt][rb * 4 + 0] * inv) | ((uint32_t)f32_to_bf16(o[dt][rb * 4 + 1] * inv) << 16);
            pk.y = (uint32_t)f32_to_bf16(o[dt][rb * 4 + 2] * inv) | ((uint32_t)f32_to_bf16(o[dt][rb * 4 + 3] * inv) << 16);
            *reinterpret_cast<uint2*>(op + dt * 32 + rb * 8 + hf * 4) = pk;
        }
}


int dit_attention(const bf16_t* qkv, bf16_t* out, const int* seq_len, int nseq, int Tmax, int H, int chunk, hipStream_t st, int q_begin, bool interleaved) {
    FY_CHECK(qkv && out && seq_len && nseq >= 1 && Tmax >= 1 && H >= 1, FY_ERR_ARG, "dit_attention: bad arguments");
    const float sl2 = 0.125f * 1.4426950408889634f;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
    }
    static const int force = getenv("FY_ATTN_WAVES") ? atoi(getenv("FY_ATTN_WAVES")) : 0;   // experiments and the equality test
    // Queries are cut into 32-row waves; the waves of a workgroup share the K/V tiles of their (sequence, head).  With 4-wave
    // workgroups (3 per CU) T = 400 gives 4 x H x nseq workgroups, a quarter of them with ONE live wave, and 1024 of them on 768
    // places take two rounds: 33.9 us per call at batch 8.  When a (sequence, head) fits one workgroup of <= 16 waves and those
    // workgroups cover at least half the CUs, one workgroup takes all of a pair's queries (4 waves per SIMD, Q fragments in LDS):
    // K/V are staged once per pair and the grid is a single round - 27.5 us (rocprofv3 over bench.py --no-pipeline).  From 8 waves
    // on the form needs no spill at 128 registers.
    const int seq_rows = interleaved ? 1 : Tmax, row_step = interleaved ? nseq : 1;
    static const int v1 = getenv("FY_ATTN_V1") ? atoi(getenv("FY_ATTN_V1")) : 0;            // the round-2 kernels (A/B measurements)
    if (!v1) {
        return dit_attention2(qkv, out, seq_len, nseq, Tmax, H, chunk, st, q_begin, seq_rows, row_step);
    }
    const int nw = cdiv(Tmax, 32);
    int W = 4;                                                   // waves per workgroup
    if (force) W = force;
    else if (nw >= 8 && nw <= 16 && H * nseq >= cus / 2 && q_begin == 0) W = nw;
    if (W <= 4) {
        dim3 grid(cdiv(Tmax, 4 * 32), H, nseq);
        hipLaunchKernelGGL((dit_attention_k<4, 3>), grid, dim3(4 * 64), 0, st, qkv, out, seq_len, Tmax, H, chunk, sl2, q_begin, seq_rows, row_step);
    } else {
        dim3 grid(cdiv(nw, W), H, nseq);
        // (9-12 waves do not fit four per SIMD without spilling - the compiler said so on every build - and never ran at more than three)
#define FY_ATT_CASE(N) case N: hipLaunchKernelGGL((dit_attention_k<N, (N >= 9 && N <= 12) ? 3 : 4>), grid, dim3(N * 64), 0, st, qkv, out, seq_len, Tmax, H, chunk, sl2, q_begin, seq_rows, row_step); break;
        switch (W) {
            FY_ATT_CASE(8) FY_ATT_CASE(9) FY_ATT_CASE(10) FY_ATT_CASE(11) FY_ATT_CASE(12)
            FY_ATT_CASE(13) FY_ATT_CASE(14) FY_ATT_CASE(15) FY_ATT_CASE(16)
            default: FY_CHECK(false, FY_ERR_ARG, "dit_attention: %d waves per workgroup", W);
        }
#undef FY_ATT_CASE
    }
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// -----------------------------------------------------------------------------------------------
// The split-operand form for the fp32-class flow decoder (FY_PRECISE): q, k, v arrive as x = hi + lo, two bf16 planes written by
// the projection's epilogue (16 mantissa bits), and every product keeps the three leading terms of (a_hi + a_lo)(b_hi + b_lo):
//   S^T = K_hi Q_hi^T + K_hi Q_lo^T + K_lo Q_hi^T;   P = softmax(S) in fp32, split again p = p_hi + p_lo;
//   O^T = V_hi^T P_hi^T + V_lo^T P_hi^T + V_hi^T P_lo^T            (the dropped lo x lo terms are 2^-16 of a term each)
// - three MFMAs where the bf16 kernel has one, instead of the fp32 VALU kernel this replaces (which re-read a sequence's keys
// from L2 per query: 104-155 us per call against ~30).  Same tiling as dit_attention_k: scores transposed so a lane owns one
// query, the probabilities in registers are the B operand of the second product, V^T by ds_read_b64_tr_b16; K/V tiles of both
// planes double-buffered in LDS (86 KB: one workgroup of up to 8 waves per CU, two waves per SIMD, Q fragments in registers).
// qkv_hi / qkv_lo: bf16 [nseq*Tmax][3*H*64] as [q | k | v]; out_hi / out_lo: bf16 [nseq*Tmax][H*64]
template <int AT_NW>
__global__ __launch_bounds__(AT_NW * 64) void dit_attention_split_k(const bf16_t* __restrict__ qkv_hi, const bf16_t* __restrict__ qkv_lo,
                                                                     bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                                                     const int* __restrict__ seq_len, int Tmax, int H, int chunk, float scale_log2) {
    extern __shared__ __attribute__((aligned(16))) bf16_t at_smem[];
    constexpr int KSZ = 64 * AT_KP, VSZ = 64 * AT_VP, BUF = 2 * KSZ + 2 * VSZ;       // one buffer: K_hi, K_lo, V_hi, V_lo
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int s = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * (AT_NW * 32);
    const int len = seq_len[s];
    if (q0 >= len) return;
    const int ld = 3 * H * AT_D;
    const long sbase = (long)s * Tmax * ld;
    const int lr = lane & 31, hf = lane >> 5;
    const int qrow = q0 + wid * 32 + lr;                     // this lane's query
    frag_ab qh[4], ql[4];                                    // B[k = 16ks + 8hf + j][col = query], both planes
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const long o = sbase + (long)min(qrow, len - 1) * ld + h * AT_D + ks * 16 + hf * 8;
        qh[ks] = *reinterpret_cast<const frag_ab*>(qkv_hi + o);
        ql[ks] = *reinterpret_cast<const frag_ab*>(qkv_lo + o);
    }
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    int kend = len;
    if (chunk > 0) kend = min(len, ((min(q0 + AT_NW * 32 - 1, len - 1) / chunk) + 1) * chunk);
    const int lim = chunk > 0 ? min(len, ((min(qrow, len - 1) / chunk) + 1) * chunk) : len;   // keys < lim are visible to this query
    // a tile = 4 planes x 512 16-byte chunks (64 keys x 8 chunks): K_hi, K_lo, V_hi, V_lo; chunk c = tid + i * threads
    constexpr int NT = AT_NW * 64, NLD = (2048 + NT - 1) / NT;
    uint4 kvreg[NLD];
    auto load_kv = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * NT, pl = c >> 9, idx = c & 511, key = idx >> 3, dc = (idx & 7) * 8;
            if (2048 % NT == 0 || c < 2048) {
                const bool ok = k0 + key < len;
                const bf16_t* src = (pl & 1) ? qkv_lo : qkv_hi;
                kvreg[i] = *reinterpret_cast<const uint4*>(src + sbase + (long)min(k0 + key, len - 1) * ld + h * AT_D + dc + (pl < 2 ? 1 : 2) * H * AT_D);
                if (!ok) kvreg[i] = make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_kv = [&](int buf) {
        bf16_t* b = at_smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * NT, pl = c >> 9, idx = c & 511, key = idx >> 3, dc = (idx & 7) * 8;
            if (2048 % NT == 0 || c < 2048) {
                if (pl < 2) *reinterpret_cast<uint4*>(b + pl * KSZ + key * AT_KP + dc) = kvreg[i];
                else *reinterpret_cast<uint4*>(b + 2 * KSZ + (pl - 2) * VSZ + key * AT_VP + dc) = kvreg[i];
            }
        }
    };
    const int gi = lane >> 4, li = lane & 15;
    const int v_off = (4 * hf + (li >> 2)) * AT_VP + 16 * (gi & 1) + 4 * (li & 3);
    const bool wave_live = q0 + wid * 32 < len;              // a wave past the sequence end only helps staging
    load_kv(0);
    store_kv(0);
    __syncthreads();
    for (int k0 = 0, it = 0; k0 < kend; k0 += 64, ++it) {
        const bf16_t* Kh = at_smem + (it & 1) * BUF;
        const bf16_t* Kl = Kh + KSZ;
        const bf16_t* Vh = Kh + 2 * KSZ;
        const bf16_t* Vl = Vh + VSZ;
        const bool more = k0 + 64 < kend;
        if (more) load_kv(k0 + 64);
        if (wave_live) {
        f32x16 sc[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[kt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ko = (kt * 32 + lr) * AT_KP + ks * 16 + hf * 8;
                const frag_ab kh = *reinterpret_cast<const frag_ab*>(Kh + ko);
                const frag_ab kl = *reinterpret_cast<const frag_ab*>(Kl + ko);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[ks], sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[ks], sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[ks], sc[kt], 0, 0, 0);
            }
        }
        frag_ab ph[2][2], pl[2][2];                          // P^T = p_hi + p_lo as B operands: [kt][k-step s] = registers 8s .. 8s+7
        float mx = -1e30f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
                const float v = key < lim ? sc[kt][r] * scale_log2 : -1e30f;
                sc[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(m_run, mx);
        const float alpha = exp2f(m_run - mnew);
        float ps = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = sc[kt][r] > -1e29f ? exp2f(sc[kt][r] - mnew) : 0.f;
                ps += p;
                const __bf16 hb = (__bf16)p;
                ph[kt][r >> 3][r & 7] = hb;
                pl[kt][r >> 3][r & 7] = (__bf16)(p - (float)hb);
            }
        m_run = mnew;
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int vo = (kt * 32 + st * 16) * AT_VP + dt * 32 + v_off;
                    s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vh + vo));
                    s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vh + vo + 8 * AT_VP));
                    s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vl + vo));
                    s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vl + vo + 8 * AT_VP));
                    const bf16x8 vh = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    const bf16x8 vl = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(frag_ab, vh), ph[kt][st], o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(frag_ab, vl), ph[kt][st], o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(frag_ab, vh), pl[kt][st], o[dt], 0, 0, 0);
                }
        }
        if (more) store_kv((it + 1) & 1);                    // that buffer was last read before the previous barrier
        __syncthreads();
    }
    if (qrow >= len) return;
    const float inv = 1.f / l_run;
    const long oo = ((long)s * Tmax + qrow) * (H * AT_D) + h * AT_D;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            float v[4];
            bf16_t hb[4], lb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = o[dt][rb * 4 + j] * inv;
                hb[j] = f32_to_bf16(v[j]);
                lb[j] = f32_to_bf16(v[j] - bf16_to_f32(hb[j]));
            }
            uint2 a, b;
            a.x = (uint32_t)hb[0] | ((uint32_t)hb[1] << 16); a.y = (uint32_t)hb[2] | ((uint32_t)hb[3] << 16);
            b.x = (uint32_t)lb[0] | ((uint32_t)lb[1] << 16); b.y = (uint32_t)lb[2] | ((uint32_t)lb[3] << 16);
            *reinterpret_cast<uint2*>(out_hi + oo + dt * 32 + rb * 8 + hf * 4) = a;
            *reinterpret_cast<uint2*>(out_lo + oo + dt * 32 + rb * 8 + hf * 4) = b;
        }
}

template <int W>
static int dit_attention_split_launch(const bf16_t* qh, const bf16_t* ql, bf16_t* oh, bf16_t* ol, const int* seq_len, int nseq, int Tmax, int H, int chunk,
                                      dim3 grid, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * (2 * 64 * AT_KP + 2 * 64 * AT_VP) * sizeof(bf16_t);      // 86 016 B: above the 64 KB default
    // set once per device and kernel (several flow handles may make their first call at once: the setup is idempotent)
    static PerDeviceOnce attr_once;
    const int dslot = current_device_slot();
    if (!attr_once.done[dslot].load(std::memory_order_acquire)) {
        HIP_TRY(hipFuncSetAttribute((const void*)dit_attention_split_k<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_once.done[dslot].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((dit_attention_split_k<W>), grid, dim3(W * 64), lds, st, qh, ql, oh, ol, seq_len, Tmax, H, chunk, 0.125f * 1.4426950408889634f);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

int dit_attention_split(const bf16_t* qkv_hi, const bf16_t* qkv_lo, bf16_t* out_hi, bf16_t* out_lo, const int* seq_len, int nseq, int Tmax, int H, int chunk,
                        hipStream_t st) {
    FY_CHECK(qkv_hi && qkv_lo && out_hi && out_lo && seq_len && nseq >= 1 && Tmax >= 1 && H >= 1, FY_ERR_ARG, "dit_attention_split: bad arguments");
    // 32 queries per wave, up to 8 waves per workgroup, the waves spread evenly over the workgroups of a (sequence, head); at least 4
    // (waves past the sequence end help staging the 4-plane K/V tiles)
    const int nw = cdiv(Tmax, 32), nblk = cdiv(nw, 8), W = std::max(4, cdiv(nw, nblk));
    dim3 grid(cdiv(nw, W), H, nseq);
    switch (W) {
        case 4: return dit_attention_split_launch<4>(qkv_hi, qkv_lo, out_hi, out_lo, seq_len, nseq, Tmax, H, chunk, grid, st);
        case 5: return dit_attention_split_launch<5>(qkv_hi, qkv_lo, out_hi, out_lo, seq_len, nseq, Tmax, H, chunk, grid, st);
        case 6: return dit_attention_split_launch<6>(qkv_hi, qkv_lo, out_hi, out_lo, seq_len, nseq, Tmax, H, chunk, grid, st);
        case 7: return dit_attention_split_launch<7>(qkv_hi, qkv_lo, out_hi, out_lo, seq_len, nseq, Tmax, H, chunk, grid, st);
        default: return dit_attention_split_launch<8>(qkv_hi, qkv_lo, out_hi, out_lo, seq_len, nseq, Tmax, H, chunk, grid, st);
    }
}

// -----------------------------------------------------------------------------------------------
// q: fp32 [R][q_ld] (head hq at column hq*64); cache K/V: fp32 [seq][kvh][max_ctx][64];
// row r attends positions 0..row_pos[r] of sequence row_seq[r]; out: fp32 [R][o_ld]
__global__ __launch_bounds__(64) void llm_attention_k(const float* __restrict__ q, int q_ld, const float* __restrict__ Kc,
                                                      const float* __restrict__ Vc, const int* __restrict__ row_seq,
                                                      const int* __restrict__ row_pos, float* __restrict__ out, int o_ld,
                                                      int Hq, int Hk, int max_ctx, float scale) {
    extern __shared__ float sh[];             // [64] q + [max_ctx] scores
    float* qs = sh;
    float* sc = sh + 64;
    const int r = blockIdx.x, hq = blockIdx.y, lane = threadIdx.x;
    const int seq = row_seq[r], n = row_pos[r] + 1;
    const int hk = hq / (Hq / Hk);
    qs[lane] = q[(long)r * q_ld + hq * 64 + lane];
    __syncthreads();
    const float* Kb = Kc + ((long)seq * Hk + hk) * max_ctx * 64;
    const float* Vb = Vc + ((long)seq * Hk + hk) * max_ctx * 64;
    float mx = -1e30f;
    for (int j = lane; j < n; j += 64) {
        const float4* kr = reinterpret_cast<const float4*>(Kb + (long)j * 64);
        float s = 0.f;
#pragma unroll
        for (int d4 = 0; d4 < 16; ++d4) {
            float4 kv = kr[d4];
            s = fmaf(qs[d4 * 4 + 0], kv.x, s);
            s = fmaf(qs[d4 * 4 + 1], kv.y, s);
            s = fmaf(qs[d4 * 4 + 2], kv.z, s);
            s = fmaf(qs[d4 * 4 + 3], kv.w, s);
        }
        s *= scale;
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) {
        float p = expf(sc[j] - mx);
        sc[j] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    __syncthreads();
    float acc = 0.f;
    for (int j = 0; j < n; ++j) acc = fmaf(sc[j], Vb[(long)j * 64 + lane], acc);
    out[(long)r * o_ld + hq * 64 + lane] = acc / sum;
}

int llm_attention(const float* q, int q_ld, const float* Kc, const float* Vc, const int* row_seq, const int* row_pos, float* out,
                  int o_ld, int R, int Hq, int Hk, int max_ctx, hipStream_t st) {
    FY_CHECK(q && Kc && Vc && row_seq && row_pos && out && R >= 1 && Hq % Hk == 0, FY_ERR_ARG, "llm_attention: bad arguments");
    size_t lds = (64 + (size_t)max_ctx) * sizeof(float);
    FY_CHECK(lds <= 64 * 1024, FY_ERR_ARG, "llm_attention: context %d too long for the score buffer", max_ctx);
    hipLaunchKernelGGL(llm_attention_k, dim3(R, Hq), dim3(64), lds, st, q, q_ld, Kc, Vc, row_seq, row_pos, out, o_ld, Hq, Hk, max_ctx, 0.125f);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// -----------------------------------------------------------------------------------------------
// Decode step: split-half RoPE of this row's q and new k, the cache append, and attention over
// positions 0..pos in one launch.  qkv: fp32 [R][(Hq+2Hk)*64] raw projections of the current token;
// the new key/value never round-trip through the cache inside the step (each block rotates its own
// copy; the first query head of a kv group writes it for later steps).
__global__ __launch_bounds__(256) void llm_attention_step_k(const float* __restrict__ qkv, float* __restrict__ Kc, float* __restrict__ Vc,
                                                            const int* __restrict__ row_seq, const int* __restrict__ row_pos,
                                                            const float* __restrict__ inv_freq, float* __restrict__ out, int o_ld,
                                                            int Hq, int Hk, int max_ctx, float scale, bf16_t* __restrict__ img) {
    // 4 waves per (row, query head): every wave rotates q / the new k itself (64 lanes = 64 dims); the cached keys
    // are spread one per thread, the value rows in four contiguous quarters, partial sums meet in LDS.
    extern __shared__ float sh[];             // [64] q + [max_ctx] scores + [4][64] partial outputs + [8] reductions
    float* qs = sh;
    float* sc = sh + 64;
    float* part = sc + max_ctx;
    float* redm = part + 256;
    const int r = blockIdx.x, hq = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int seq = row_seq[r], pos = row_pos[r];
    const int grp = Hq / Hk, hk = hq / grp;
    const int ld = (Hq + 2 * Hk) * 64;
    const float* row = qkv + (long)r * ld;
    const float qv = row[hq * 64 + lane], kv = row[(Hq + hk) * 64 + lane], vv = row[(Hq + Hk + hk) * 64 + lane];
    const float ang = (float)pos * inv_freq[lane & 31];
    float sn, cs;
    sincosf(ang, &sn, &cs);
    const float qo = __shfl_xor(qv, 32, 64), ko = __shfl_xor(kv, 32, 64);
    const float qr = lane < 32 ? qv * cs - qo * sn : qv * cs + qo * sn;
    const float kr = lane < 32 ? kv * cs - ko * sn : kv * cs + ko * sn;
    float* Kb = Kc + ((long)seq * Hk + hk) * max_ctx * 64;
    float* Vb = Vc + ((long)seq * Hk + hk) * max_ctx * 64;
    if (wid == 0) {
        if (hq % grp == 0) {
            Kb[(long)pos * 64 + lane] = kr;
            Vb[(long)pos * 64 + lane] = vv;
        }
        qs[lane] = qr;
    }
    __syncthreads();
    const float s_new = wave_sum(qr * kr) * scale;
    float mx = s_new;
    for (int j = tid; j < pos; j += 256) {
        const float4* kp = reinterpret_cast<const float4*>(Kb + (long)j * 64);
        float4 k4[16];
#pragma unroll
        for (int d4 = 0; d4 < 16; ++d4) k4[d4] = kp[d4];
        float s = 0.f;
#pragma unroll
        for (int d4 = 0; d4 < 16; ++d4) {
            s = fmaf(qs[d4 * 4 + 0], k4[d4].x, s);
            s = fmaf(qs[d4 * 4 + 1], k4[d4].y, s);
            s = fmaf(qs[d4 * 4 + 2], k4[d4].z, s);
            s = fmaf(qs[d4 * 4 + 3], k4[d4].w, s);
        }
        s *= scale;
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if (lane == 0) redm[wid] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(redm[0], redm[1]), fmaxf(redm[2], redm[3]));
    float sum = 0.f;
    for (int j = tid; j < pos; j += 256) {
        float p = expf(sc[j] - mx);
        sc[j] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) redm[4 + wid] = sum;
    __syncthreads();
    const float p_new = expf(s_new - mx);
    sum = ((redm[4] + redm[5]) + (redm[6] + redm[7])) + p_new;
    // values: wave w owns positions [w*q4, (w+1)*q4)
    const int q4 = (pos + 3) >> 2, j0 = wid * q4, j1 = min(pos, j0 + q4);
    float acc = 0.f;
    int j = j0;
    for (; j + 8 <= j1; j += 8) {             // eight independent row loads in flight per trip
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = Vb[(long)(j + u) * 64 + lane];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = fmaf(sc[j + u], v[u], acc);
    }
    for (; j < j1; ++j) acc = fmaf(sc[j], Vb[(long)j * 64 + lane], acc);
    part[wid * 64 + lane] = acc;
    __syncthreads();
    if (wid == 0) {
        float o = ((part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane])) + p_new * vv;
        o /= sum;
        if (!img) { out[(long)r * o_ld + hq * 64 + lane] = o; return; }
        // the o-proj product of the 32-row path reads an A image (gemv32.h): the exact 3-way bf16 split of this head's 64
        // values as 8 pieces of 16 bytes per plane; lanes 0-23 each assemble and store one
        unsigned h, m, l;
        gv32_split3(o, h, m, l);
        bf16_t* ps = reinterpret_cast<bf16_t*>(part);           // every lane has read its partial sums above (one wave, in order)
        ps[lane] = (bf16_t)h; ps[64 + lane] = (bf16_t)m; ps[128 + lane] = (bf16_t)l;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 24) {
            const int plane = lane >> 3, pc = lane & 7, col0 = hq * 64 + 8 * pc;
            const uint4 piece = *reinterpret_cast<const uint4*>(ps + plane * 64 + pc * 8);
            *reinterpret_cast<uint4*>(reinterpret_cast<char*>(img) + gv32_off(Hq * 4, r, col0, plane)) = piece;
        }
    }
}

// The same step with ONE block per (row, kv head) serving all Hq / Hk query heads of the group (<= 8): the group's cached keys
// are read once (a thread per position: its 64-dim dot product against every head's rotated query out of LDS), the values are
// staged through LDS in chunks of GQ_CH positions (one coalesced burst) and every head's wave walks them there.  R x Hk blocks
// where the per-head kernel above launches R x Hq - beside the flow decoder an LM block displaces a GEMM workgroup from its CU for
// as long as it lives, and this kernel's 448 blocks x 6 us per layer were a third of the LM's block-seconds.
#define GQ_CH 128
__global__ __launch_bounds__(512) void llm_attention_step_gqa_k(const float* __restrict__ qkv, float* __restrict__ Kc, float* __restrict__ Vc,
                                                                const int* __restrict__ row_seq, const int* __restrict__ row_pos,
                                                                const float* __restrict__ inv_freq, float* __restrict__ out, int o_ld,
                                                                int Hq, int Hk, int max_ctx, float scale, bf16_t* __restrict__ img) {
    extern __shared__ float sh[];
    float* qs = sh;                               // [8][64] rotated queries of the group
    float* knew = sh + 512;                       // [64] this step's rotated key, [64] its value
    float* vnew = sh + 576;
    float* sc = sh + 640;                         // [grp][max_ctx] scores, then probabilities
    const int r = blockIdx.x, hk = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int grp = Hq / Hk;
    float* vbuf = sc + ((grp * max_ctx + 3) & ~3);   // [GQ_CH][64] staged values (16-byte aligned)
    bf16_t* psb = reinterpret_cast<bf16_t*>(vbuf + GQ_CH * 64);       // [8 waves][192] split scratch of the image writer
    const int seq = row_seq[r], pos = row_pos[r];
    const int ld = (Hq + 2 * Hk) * 64;
    const float* row = qkv + (long)r * ld;
    const float ang = (float)pos * inv_freq[lane & 31];
    float sn, cs;
    sincosf(ang, &sn, &cs);
    float* Kb = Kc + ((long)seq * Hk + hk) * max_ctx * 64;
    float* Vb = Vc + ((long)seq * Hk + hk) * max_ctx * 64;
    if (wid < grp) {
        const float qv = row[(hk * grp + wid) * 64 + lane];
        const float qo = __shfl_xor(qv, 32, 64);
        qs[wid * 64 + lane] = lane < 32 ? qv * cs - qo * sn : qv * cs + qo * sn;
    }
    if (wid == 0) {
        const float kv = row[(Hq + hk) * 64 + lane], vv = row[(Hq + Hk + hk) * 64 + lane];
        const float ko = __shfl_xor(kv, 32, 64);
        const float kr = lane < 32 ? kv * cs - ko * sn : kv * cs + ko * sn;
        Kb[(long)pos * 64 + lane] = kr;
        Vb[(long)pos * 64 + lane] = vv;
        knew[lane] = kr;
        vnew[lane] = vv;
    }
    __syncthreads();
    // scores of the cached positions: one position per thread, its key row against every head's query (the per-head kernel's order
    // of the 64 products)
    for (int j = tid; j < pos; j += 512) {
        const float4* kp = reinterpret_cast<const float4*>(Kb + (long)j * 64);
        float4 k4[16];
#pragma unroll
        for (int d4 = 0; d4 < 16; ++d4) k4[d4] = kp[d4];
        for (int h = 0; h < grp; ++h) {
            const float* q = qs + h * 64;
            float s = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < 16; ++d4) {
                s = fmaf(q[d4 * 4 + 0], k4[d4].x, s);
                s = fmaf(q[d4 * 4 + 1], k4[d4].y, s);
                s = fmaf(q[d4 * 4 + 2], k4[d4].z, s);
                s = fmaf(q[d4 * 4 + 3], k4[d4].w, s);
            }
            sc[h * max_ctx + j] = s * scale;
        }
    }
    __syncthreads();
    // softmax: wave h owns head h
    float p_new = 0.f, sum = 1.f;
    if (wid < grp) {
        float* sch = sc + wid * max_ctx;
        const float s_new = wave_sum(qs[wid * 64 + lane] * knew[lane]) * scale;
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
    }
    // values: GQ_CH positions at a time through LDS; thread (head = wave, dim = lane) accumulates in position order
    float acc = 0.f;
    for (int c0 = 0; c0 < pos; c0 += GQ_CH) {
        const int n = min(GQ_CH, pos - c0);
        __syncthreads();                                  // the probabilities are written / the previous chunk has been read
        const float4* vsrc = reinterpret_cast<const float4*>(Vb + (long)c0 * 64);
        float4 v4[GQ_CH * 16 / 512];
#pragma unroll
        for (int i = 0; i < GQ_CH * 16 / 512; ++i) {
            const int e = tid + i * 512;
            v4[i] = e < n * 16 ? vsrc[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < GQ_CH * 16 / 512; ++i) reinterpret_cast<float4*>(vbuf)[tid + i * 512] = v4[i];
        __syncthreads();
        if (wid < grp) {
            const float* sch = sc + wid * max_ctx + c0;
            for (int j = 0; j < n; ++j) acc = fmaf(sch[j], vbuf[j * 64 + lane], acc);
        }
    }
    if (wid >= grp) return;
    const int hq = hk * grp + wid;
    const float o = (acc + p_new * vnew[lane]) / sum;
    if (!img) { out[(long)r * o_ld + hq * 64 + lane] = o; return; }
    unsigned h, m, l;
    gv32_split3(o, h, m, l);
    bf16_t* ps = psb + wid * 192;
    ps[lane] = (bf16_t)h; ps[64 + lane] = (bf16_t)m; ps[128 + lane] = (bf16_t)l;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < 24) {
        const int plane = lane >> 3, pc = lane & 7, col0 = hq * 64 + 8 * pc;
        const uint4 piece = *reinterpret_cast<const uint4*>(ps + plane * 64 + pc * 8);
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(img) + gv32_off(Hq * 4, r, col0, plane)) = piece;
    }
}

int llm_attention_step(const float* qkv, float* Kc, float* Vc, const int* row_seq, const int* row_pos, const float* inv_freq,
                       float* out, int o_ld, int R, int Hq, int Hk, int max_ctx, hipStream_t st, bf16_t* img) {
    FY_CHECK(qkv && Kc && Vc && row_seq && row_pos && inv_freq && (out || img) && R >= 1 && Hq % Hk == 0, FY_ERR_ARG, "llm_attention_step: bad arguments");
    {
        // one block per (row, kv head) when the group fits a block's waves (FY_LLM_ATTN_GQA=0: the per-head kernel)
        static const int gqa = getenv("FY_LLM_ATTN_GQA") ? atoi(getenv("FY_LLM_ATTN_GQA")) : 1;
        const int grp = Hq / Hk;
        const size_t lds_g = (640 + (((size_t)grp * max_ctx + 3) & ~(size_t)3) + GQ_CH * 64) * sizeof(float) + 8 * 192 * sizeof(bf16_t);
        if (gqa && grp <= 8 && lds_g <= 64 * 1024) {
            hipLaunchKernelGGL(llm_attention_step_gqa_k, dim3(R, Hk), dim3(512), lds_g, st, qkv, Kc, Vc, row_seq, row_pos, inv_freq, out, o_ld, Hq, Hk,
                               max_ctx, 0.125f, img);
            HIP_TRY(hipGetLastError());
            return FY_OK;
        }
    }
    size_t lds = (64 + (size_t)max_ctx + 256 + 8) * sizeof(float);
    FY_CHECK(lds <= 64 * 1024, FY_ERR_ARG, "llm_attention_step: context %d too long for the score buffer", max_ctx);
    hipLaunchKernelGGL(llm_attention_step_k, dim3(R, Hq), dim3(256), lds, st, qkv, Kc, Vc, row_seq, row_pos, inv_freq, out, o_ld, Hq, Hk,
                       max_ctx, 0.125f, img);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}
