// The DiT attention kernel of round 5 (dit_attention's default; attn.hip keeps the round-2 kernels for FY_ATTN_V1=1, the cross-check
// and the FY_PRECISE split form).  Non-causal multi-head attention of the DiT estimator (16 heads x 64, bf16 in, fp32 online softmax,
// bf16 out) with a key-padding mask and the optional block-causal chunk mask of streaming inference
// (CosyVoice/cosyvoice/flow/DiT/modules.py:349-407; utils/mask.py:127-158).
// Compiled with -fno-slp-vectorize (build.py): packed f32 VALU forms beside MFMAs cost more than the two scalar ones they replace.
#include "attn.h"
#include "runtime.h"
#include <algorithm>
#include <stdlib.h>
#include <type_traits>
#include <utility>

typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define AT_D 64
#define AT_KP 72             // bf16 elements per LDS row of the K tile (64 + 8 pad: conflict-free ds_read_b128 rows)
#define AT_VP 96             // V tile pitch: 48 dwords, so the four rows of a transposed 4x16 block sit 16 banks apart

// The scores are computed TRANSPOSED, S^T = K Q^T (v_mfma_f32_32x32x16_bf16: A = K rows out of LDS, B = Q^T straight
// from global memory), so a lane owns one query (its column) and 16 of a half tile's 32 keys (its registers; the other
// 16 are on lane^32): softmax statistics are per-lane scalars with one cross-half exchange, and the probabilities,
// rounded to bf16 in place, ARE the B operand of O^T = V^T P^T -- no LDS round trip for P.  The k order of that second
// product follows the accumulator's register order (key 8(j>>2) + 4h + (j&3) in element j of lane half h); the
// A operand V^T is read to match, transposed by the LDS itself (ds_read_b64_tr_b16 on the row-major V tile).
//
// What bounded dit_attention_k (26.5 us per call at 16 x 400 frames, MFMA busy 20 %): a wave ran QK^T, then the softmax of the same
// tile, then PV - and the waves of a workgroup meet at one barrier per key tile, so all of them are in the same phase at once: the
// matrix cores idle while every wave is in its exp block and the vector unit idles under the MFMAs (the three pipes' times simply
// add: ~7 us of MFMA + ~11 of VALU issue + the LDS fragment reads).  This form gives ONE wave two independent instruction streams:
//   * a workgroup is four waves of NQ = 2 32-query blocks each (256 queries), two workgroups per CU = two waves per SIMD at <= 256
//     registers; a wave's blocks share every K and V^T fragment read (one read of 4 + 8 fragments per 32-key half tile).  (NQ = 4 with
//     one wave per SIMD and the whole 512-register file was built first: its sixteen 16-register accumulator tuples did not survive
//     the register allocator - hundreds of spills and 80 copies between the vector and accumulator files per stage.)
//   * the key axis is walked in 32-key half tiles, and the blocks ROTATE through a stage pipeline.  Stage G = h NQ + q runs, on the
//     vector unit, the exponents of block q for half tile h (S -> P: exp2, bf16 pack) and the row maxima of the block two stages
//     back; on the matrix cores, for the block ONE stage back (whose P is complete), O += V(h)^T P^T and at once the scores of its
//     next half tile, S = K(h+1) Q^T, into the registers its exponents just freed.  A stage is ten MFMAs (ONES) each followed by a
//     slice of the vector work (two exp2 + their fma + one cvt_pk = 28 issue cycles beside the MFMA's 32), scheduling barriers
//     between the groups - no MFMA waits for the vector work issued beside it, and a block's scores exist once;
//   * the control flow inside the key loop is ONE body: masks (sequence end, streaming chunks) are a rarely taken branch that
//     rewrites a block's scores before anything reads them, and the rescale is a rarely taken branch after the maxima - with a body
//     per mask / parity case the allocator shuffled the accumulators between the paths;
//   * ONES: the row sums l = sum_k P come from the matrix cores too (an all-ones A operand: 2 more MFMAs per block and half tile
//     instead of 16 v_add_f32 - the vector unit is the busier pipe at head dimension 64), over the SAME bf16 P the numerator uses;
//   * K tiles in a ring of three (scores run a half tile ahead of PV, so K(t+1) is read while V(t) still is), V tiles in a ring of two,
//     one barrier per 64 keys.
// A query's result depends on its own row and the sequence's keys only - not on the batch, the launch grid or q_begin: the
// incremental and batched paths stay bit-identical to the plain ones (tests/test_flow_gpu.py).
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

#define SB() __builtin_amdgcn_sched_barrier(0)
#ifndef AT2_ABL
#define AT2_ABL 0            // ablations for tests/micro/prof_attn.sh (never in the library): 1 no exp2, 2 no MFMAs, 3 no K/V loads in the loop, 4 = 3 + no barrier, 5 no maxima
#endif
// -DAT2_STAMPS (tests/micro/attn_bench.hip): wave 0 of every workgroup records the 100 MHz clock at entry, when the first tiles are staged,
// after the key loop and after the stores
#ifdef AT2_STAMPS
__device__ unsigned long long* at2_stamp_buf;
void at2_set_stamps(unsigned long long* p) { hipMemcpyToSymbol(HIP_SYMBOL(at2_stamp_buf), &p, sizeof(p)); }
#define AT2_STAMP(i) do { if (threadIdx.x == 0 && at2_stamp_buf) at2_stamp_buf[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define AT2_STAMP(i) do { } while (0)
#endif
template <int NW, bool ONES>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void dit_attention2_k(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, const int* __restrict__ seq_len, int Tmax, int H,
                                                        int chunk, float scale_log2, int q_begin, int seq_rows, int row_step) {
    constexpr int NQ = 2, NT = NW * 64, NLD = 512 / NT;                               // blocks per wave; threads; 16-byte chunks of a 64-key tile per thread
    constexpr int KSZ = 64 * AT_KP, VSZ = 64 * AT_VP, QW = NQ * 32, QB = NW * QW;     // queries per wave, per workgroup
    static_assert(NW == 4 || NW == 8, "four or eight waves");
    __shared__ __attribute__((aligned(16))) bf16_t at2_smem[3 * KSZ + 2 * VSZ];      // K ring, V ring; after the key loop: the output rows, per wave
    bf16_t* const Kbuf = at2_smem;
    bf16_t* const Vbuf = at2_smem + 3 * KSZ;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int s = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QB;
    const int len = seq_len[s];
    if (q0 >= len || q0 + QB <= q_begin) return;
    AT2_STAMP(0);
    const int ld = 3 * H * AT_D;
    const long ldr = (long)ld * row_step;
    const bf16_t* base = qkv + (long)s * seq_rows * ld;
    const int lr = lane & 31, hf = lane >> 5;
    const int qrow0 = q0 + wid * QW + lr;                    // this lane's queries: qrow0 + 32 q
    frag_ab qf[NQ][4];                                       // B[k = 16ks + 8hf + j][col = query]
    f32x16 o[NQ][2];                                         // O^T[d = 32dt + (r&3) + 8(r>>2) + 4hf][query], per query block
    f32x16 ol[NQ];                                           // ONES: every element = the row sum of this lane's query
    f32x16 sc[NQ];                                           // a block's scores: of half tile h until its exponents are taken, then of h + 1
    float m_run[NQ], l_run[NQ];
    int lim[NQ];                                             // keys < lim are visible to the query
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qf[q][ks] = *reinterpret_cast<const frag_ab*>(base + (long)min(qrow0 + 32 * q, len - 1) * ldr + h * AT_D + ks * 16 + hf * 8);
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[q][0][r] = 0.f; o[q][1][r] = 0.f; ol[q][r] = 0.f; }
        m_run[q] = -1e30f;
        l_run[q] = 0.f;
        lim[q] = chunk > 0 ? min(len, ((min(qrow0 + 32 * q, len - 1) / chunk) + 1) * chunk) : len;
    }
    int kend = len;
    if (chunk > 0) kend = min(len, ((min(q0 + QB - 1, len - 1) / chunk) + 1) * chunk);
    const int NH = (kend + 31) >> 5, NTL = (kend + 63) >> 6;
    uint4 kreg[NLD], vreg[NLD];                              // a 64-key tile = 512 16-byte chunks
    auto load_tile = [&](int k0, int which, uint4 (&reg)[NLD]) {          // which: 1 K, 2 V
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * NT, key = c >> 3, dc = (c & 7) * 8;
            const bool ok = k0 + key < len;
            reg[i] = *reinterpret_cast<const uint4*>(base + (long)min(k0 + key, len - 1) * ldr + h * AT_D + dc + which * H * AT_D);
            if (!ok) reg[i] = make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&](bf16_t* dst, int pitch, const uint4 (&reg)[NLD]) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int c = tid + i * NT, key = c >> 3, dc = (c & 7) * 8;
            *reinterpret_cast<uint4*>(dst + key * pitch + dc) = reg[i];
        }
    };
    const int gi = lane >> 4, li = lane & 15;
    const int v_off = (4 * hf + (li >> 2)) * AT_VP + 16 * (gi & 1) + 4 * (li & 3);
    const int k_off = lr * AT_KP + hf * 8;
    const bool wave_live = q0 + wid * QW < len;              // a wave past the sequence end only helps staging
    const bf16x8 ones8 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    frag_ab ones = __builtin_bit_cast(frag_ab, ones8);
    asm volatile("" : "+v"(ones));                         // (opaque: held in four registers, not rebuilt from literals in every stage)
    auto xhalf = [&](float v, float& own_lo, float& own_hi) {  // the two half-waves' values of a lane pair
        const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
        own_lo = __builtin_bit_cast(float, r[0]);
        own_hi = __builtin_bit_cast(float, r[1]);
    };
    frag_ab kf[4];                                           // K fragments of the half tile whose scores are being computed
    frag_ab vf[2][2];                                        // V^T fragments of the half tile being accumulated: [st][dt]
    auto read_k = [&](const bf16_t* Kh) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const frag_ab*>(Kh + k_off + ks * 16);
    };
    auto read_v = [&](const bf16_t* Vh) {
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16_t* vb = Vh + st * 16 * AT_VP + dt * 32 + v_off;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vb + 8 * AT_VP));
                const bf16x8 vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                vf[st][dt] = __builtin_bit_cast(frag_ab, vv);
            }
    };
    // a half tile with invisible keys (past the sequence end, or past the query's chunk in streaming): their scores become -1e30
    // before anything reads them (exp2 gives exactly 0 for them; the tile's maximum ignores them)
    auto mask_scores = [&](auto qc, int kb) {
        constexpr int q = decltype(qc)::value;
        const int visn = lim[q] - kb - 4 * hf;               // register r holds a visible key while (r&3) + 8(r>>2) < visn
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[q][r] = (r & 3) + 8 * (r >> 2) < visn ? sc[q][r] : -1e30f;
    };
    // Row maxima of block q's fresh scores in two pieces (both ride behind the stage's first MFMAs, which belong to the OTHER block):
    // the lane's maximum as a tree of v_max3 (depth 3: a serial chain of 15 cost its latency before every stage), then the exchange
    // with the other half-wave; the running maximum moves, and everything summed so far shrinks by alpha (rarely: the maxima settle
    // after a few tiles)
    auto row_max_lane = [&](auto qc) -> float {
        constexpr int q = decltype(qc)::value;
        const f32x16& v = sc[q];
        auto m3 = [](float a, float b, float c) { return fmaxf(fmaxf(a, b), c); };
        const float t0 = m3(v[0], v[1], v[2]), t1 = m3(v[3], v[4], v[5]), t2 = m3(v[6], v[7], v[8]), t3 = m3(v[9], v[10], v[11]), t4 = m3(v[12], v[13], v[14]);
        return m3(m3(t0, t1, t2), m3(t3, t4, v[15]), t0);
    };
    auto row_max_update = [&](auto qc, float mx) {
        constexpr int q = decltype(qc)::value;
        float a, b;
        xhalf(mx, a, b);
        mx = fmaxf(a, b) * scale_log2;
        const float mnew = fmaxf(m_run[q], mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run[q] - mnew);
        m_run[q] = mnew;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[q][0][r] *= alpha; o[q][1][r] *= alpha; }
            if constexpr (ONES) ol[q][0] *= alpha;                                       // (the only element read at the end)
            else l_run[q] *= alpha;
        }
    };
    const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    frag_ab pf[2][2];                                        // P^T of the block in flight as the B operand: [stage parity][k-step st]
    // the ten (ONES) or eight MFMAs of a stage for block q: O += V^T P^T, the row sums, and the block's next scores - the two
    // chains interleaved so that no MFMA follows one it depends on; valu(i) is the vector slice issued behind MFMA i
    auto mfma_stage = [&](auto qc, auto parc, auto with_qk_c, auto&& valu) {
        constexpr int q = decltype(qc)::value, par = decltype(parc)::value;
        constexpr bool with_qk = decltype(with_qk_c)::value;
        constexpr int NPV = ONES ? 6 : 4;
        static_for<NPV + 4>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            // order: PV, QK, PV, QK, PV, QK, PV, QK, PV ... (the 4 QK MFMAs at the odd places 1, 3, 5, 7)
            constexpr bool is_qk = (i & 1) && i < 8;
            if constexpr (is_qk) {
                constexpr int ks = i >> 1;
                if constexpr (with_qk && AT2_ABL != 2) sc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[q][ks], ks == 0 ? z16 : sc[q], 0, 0, 0);
            } else {
                constexpr int n = i < 8 ? i >> 1 : i - 4;                // the n-th PV MFMA: st = n / (NPV / 2), w = n % (NPV / 2)
                constexpr int st = n / (NPV / 2), w = n % (NPV / 2);
                if constexpr (AT2_ABL == 2) { if constexpr (w < 2) o[q][w][0] += (float)pf[par][st][0] + (float)vf[st][w][0]; else ol[q][0] += (float)pf[par][st][1] + 1.f; }
                else if constexpr (w < 2) o[q][w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[st][w], pf[par][st], o[q][w], 0, 0, 0);
                else ol[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf[par][st], ol[q], 0, 0, 0);
            }
            valu(ic);
            SB();
        });
    };
    // the body of half tile hh (keys kb .. kb + 31): Kn = the K rows of half tile hh + 1, Vh = the V rows of half tile hh.
    // One instantiation serves every half tile but the first (whose first stage has no MFMAs behind it); masks enter through
    // mask_scores, a rarely taken branch on the scores only, so the exponent slices and the maxima never test a key's visibility.
    auto body = [&](auto first_c, bool mask_next, bool mask_cur, const bf16_t* Kn, const bf16_t* Vh, int kb) {
        constexpr bool first = decltype(first_c)::value;
        static_for<NQ>([&](auto sc_) {
            constexpr int st_ = decltype(sc_)::value;                   // stage: block st_'s exponents, block st_ - 1's MFMAs, block st_ - 2's maxima
            constexpr int qa = (st_ + NQ - 1) % NQ, qb = (st_ + NQ - 2) % NQ, par = st_ & 1;
            // (the first two stages of the first body have no maxima to take: the prologue took them)
            constexpr bool maxima = !(first && st_ < 2) && AT2_ABL != 5;
            float mq = m_run[st_], mx_lane = 0.f, ps = 0.f;
            // slice e of the exponent work: two probabilities, registers 2 e and 2 e + 1
            auto e2 = [&](auto ic) {
                constexpr int e = decltype(ic)::value;
#pragma unroll
                for (int r = 2 * e; r < 2 * e + 2; ++r) {
                    const float p = AT2_ABL == 1 ? fmaf(sc[st_][r], scale_log2, -mq) : __builtin_amdgcn_exp2f(fmaf(sc[st_][r], scale_log2, -mq));
                    if constexpr (!ONES) ps += p;
                    pf[par][r >> 3][r & 7] = (__bf16)p;
                }
            };
            // the vector work behind MFMA i of the stage (the MFMAs are block qa's; everything here is block st_'s): the maxima of its
            // fresh scores behind the first two - block st_'s scores are of half tile hh + 1 when its MFMAs ran in this body (stages
            // 2 ..), of half tile hh otherwise - then the eight exponent slices, one per MFMA
            auto valu = [&](auto ic) {
                constexpr int i = decltype(ic)::value, n_mfma = (ONES ? 6 : 4) + 4;
                if constexpr (i == 0) {
                    if constexpr (maxima) {
                        if (st_ >= 2 ? mask_next : mask_cur) mask_scores(std::integral_constant<int, qb>{}, st_ >= 2 ? kb + 32 : kb);
                        mx_lane = row_max_lane(std::integral_constant<int, qb>{});
                    }
                } else if constexpr (i == 1) {
                    if constexpr (maxima) row_max_update(std::integral_constant<int, qb>{}, mx_lane);
                    mq = m_run[st_];
                    asm volatile("" : "+v"(mq));             // (opaque: the exponents' arguments are computed in their slices, not a block early)
                } else {
                    e2(std::integral_constant<int, i - 2>{});
                    if constexpr (i == n_mfma - 1) static_for<8 - (n_mfma - 2)>([&](auto jc) { e2(std::integral_constant<int, n_mfma - 2 + decltype(jc)::value>{}); });
                }
            };
            static_assert(qb == st_ % NQ || NQ != 2, "two blocks per wave: the maxima are the block's whose exponents follow");
            if constexpr (first && st_ == 0) {
                asm volatile("" : "+v"(mq));
                static_for<8>(e2);
            } else {
                mfma_stage(std::integral_constant<int, qa>{}, std::integral_constant<int, par ^ 1>{}, std::true_type{}, valu);
            }
            // (P is complete HERE: without a use the compiler sinks the exponent slices into the next stage, beside ITS slices, and this
            // stage's MFMAs run bare)
            asm volatile("" : "+v"(pf[par][0]), "+v"(pf[par][1]));
            if constexpr (!ONES) { float a, b; xhalf(ps, a, b); l_run[st_] += a + b; }
            if constexpr (st_ == 0) { read_k(Kn); read_v(Vh); }          // the fragments every block of this body shares (stage 0 used the previous body's)
            SB();
        });
    };
    // prologue: K(0), K(1), V(0) staged; the scores of half tile 0 and their maxima
    {
        uint4 kreg1[NLD];                                    // (all three tiles requested before the first is stored)
        load_tile(0, 1, kreg);
        load_tile(0, 2, vreg);
        if (NTL > 1) load_tile(64, 1, kreg1);
        store_tile(Kbuf, AT_KP, kreg);
        store_tile(Vbuf, AT_VP, vreg);
        if (NTL > 1) store_tile(Kbuf + KSZ, AT_KP, kreg1);
    }
    __syncthreads();
    AT2_STAMP(1);
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (wave_live) {
        read_k(Kbuf);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int q = 0; q < NQ; ++q) sc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[q][ks], ks == 0 ? z16 : sc[q], 0, 0, 0);
        static_for<NQ>([&](auto qc) { mask_scores(qc, 0); row_max_update(qc, row_max_lane(qc)); });
    }
    int kslot = 0;                                           // ring slot of K(t)
    for (int t = 0; t < NTL; ++t) {
        const int k0 = t * 64;
        const bool ldk = t + 2 < NTL && AT2_ABL != 3 && AT2_ABL != 4, ldv = t + 1 < NTL && AT2_ABL != 3 && AT2_ABL != 4;
        if (ldk) load_tile(k0 + 128, 1, kreg);
        if (ldv) load_tile(k0 + 64, 2, vreg);
        const int kslot1 = kslot == 2 ? 0 : kslot + 1, kslot2 = kslot1 == 2 ? 0 : kslot1 + 1;
        if (wave_live) {
            const bf16_t* Kt = Kbuf + kslot * KSZ;
            const bf16_t* Kt1 = Kbuf + kslot1 * KSZ;
            const bf16_t* Vt = Vbuf + (t & 1) * VSZ;
            // which half tiles hold invisible keys (streaming: any; otherwise those past the sequence end - past the last half tile
            // the scores are computed on stale rows and masked away whole)
            const bool m0 = chunk > 0 || k0 + 32 > len, m1 = chunk > 0 || k0 + 64 > len, m2 = chunk > 0 || k0 + 96 > len;
            if (t == 0) body(T_{}, m1, m0, Kt + 32 * AT_KP, Vt, k0);
            else body(F_{}, m1, m0, Kt + 32 * AT_KP, Vt, k0);
            if (2 * t + 1 < NH) body(F_{}, m2, m1, Kt1, Vt + 32 * AT_VP, k0 + 32);
        }
        if (ldk) store_tile(Kbuf + kslot2 * KSZ, AT_KP, kreg);
        if (ldv) store_tile(Vbuf + ((t + 1) & 1) * VSZ, AT_VP, vreg);
        kslot = kslot1;
        if (AT2_ABL != 4) __syncthreads();
    }
    AT2_STAMP(2);
    // drain: the last block's P of the last half tile
    if (wave_live) mfma_stage(std::integral_constant<int, NQ - 1>{}, std::integral_constant<int, (NQ - 1) & 1>{}, std::false_type{}, [](auto) {});
    // Output: a lane holds 4-element pieces of its query's row (O^T's register order); stored from there, a wave instruction touches
    // 32 rows x 16 bytes - the launch's last 2.7 us were this write path (tests/micro/prof_attn.sh stamps).  Through LDS instead (the
    // K / V rings are free behind the loop's last barrier; 32 rows x 144 B per wave): 8 lanes then store one row's 128 bytes.
    if (!wave_live) return;
    bf16_t* const orow = at2_smem + wid * (32 * AT_KP);
    static_assert(NW * 32 * AT_KP <= 3 * KSZ + 2 * VSZ, "output staging fits the rings");
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float inv = 1.f / (ONES ? ol[q][0] : l_run[q]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                typedef __attribute__((ext_vector_type(4))) __bf16 bf16v4;
                const bf16v4 pk = {(__bf16)(o[q][dt][rb * 4 + 0] * inv), (__bf16)(o[q][dt][rb * 4 + 1] * inv), (__bf16)(o[q][dt][rb * 4 + 2] * inv), (__bf16)(o[q][dt][rb * 4 + 3] * inv)};
                *reinterpret_cast<bf16v4*>(orow + lr * AT_KP + dt * 32 + rb * 8 + hf * 4) = pk;
            }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int ps_ = 0; ps_ < 4; ++ps_) {
            const int row = ps_ * 8 + (lane >> 3), qrow = q0 + wid * QW + 32 * q + row;
            const uint4 v = *reinterpret_cast<const uint4*>(orow + row * AT_KP + (lane & 7) * 8);
            if (qrow < len && qrow >= q_begin)
                *reinterpret_cast<uint4*>(out + ((long)s * seq_rows + (long)qrow * row_step) * (H * AT_D) + h * AT_D + (lane & 7) * 8) = v;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    AT2_STAMP(3);
}
#undef SB

int dit_attention2(const bf16_t* qkv, bf16_t* out, const int* seq_len, int nseq, int Tmax, int H, int chunk, hipStream_t st, int q_begin, int seq_rows, int row_step) {
    const float sl2 = 0.125f * 1.4426950408889634f;
    // 64 queries per wave.  Up to 256 frames: four waves per workgroup; longer sequences: eight (512 queries, one workgroup per CU) -
    // at the benchmark's 400 frames a (sequence, head)'s keys and values are then fetched once, not once per 256 queries
    // (52 MB instead of 78 per call: at 16 x 400 x 16 heads the call moves as long as it computes)
    // Which: the tiling whose last workgroup of a sequence is fuller (650 frames = 21 blocks: 2 x 16 slots of the eight-wave form
    // leave every second CU idle two thirds of the launch; 3 x 8 slots do not), the eight-wave form on a tie.
    const int nqb = cdiv(Tmax, 32), g8 = cdiv(nqb, 16), g4 = cdiv(nqb, 8);
    const bool four = Tmax <= 256 || 10 * nqb * (2 * g8 - g4) > g4 * g8 * 16;        // nqb / (8 g4) > nqb / (16 g8) + 0.1
    if (four) {
        dim3 grid(g4, H, nseq);
        hipLaunchKernelGGL((dit_attention2_k<4, true>), grid, dim3(256), 0, st, qkv, out, seq_len, Tmax, H, chunk, sl2, q_begin, seq_rows, row_step);
    } else {
        dim3 grid(g8, H, nseq);
        hipLaunchKernelGGL((dit_attention2_k<8, true>), grid, dim3(512), 0, st, qkv, out, seq_len, Tmax, H, chunk, sl2, q_begin, seq_rows, row_step);
    }
    HIP_TRY(hipGetLastError());
    return FY_OK;
}
