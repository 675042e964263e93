// See gemm.h.
#include "gemm.h"
#include "runtime.h"
#include <algorithm>
#include <atomic>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;

#define GM_BM 128
int gemm_tile_override = 0;        // microbenchmarks: 0 auto; 128 / 64 register-staged kernel with that M tile; ring kernel: 2 / 3 = 256x128 / 128x128 tiles, 2002 / 2003 the same on 16x16x32 MFMAs, 256 / 320 = 256x256 / 320x256, 1256 / 1320 those with staggered wave groups
#define GM_BN 128
#define GM_BK 64
#define GM_PITCH 72          // bf16 elements per LDS row: 64 + 8 pad (144 B: 16-B aligned, spreads ds_read_b128 over banks)

__device__ __forceinline__ void split8(const float4& a, const float4& b, uint4& hi, uint4& lo) {
    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t h[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        bf16_t hb = f32_to_bf16(f[i]);
        h[i] = hb;
        l[i] = f32_to_bf16(f[i] - bf16_to_f32(hb));
    }
    hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}

// the exact three-way split x = hi + mid + lo (3 x 8 mantissa bits; hardware round-to-nearest-even converts)
__device__ __forceinline__ void split8_3(const float4& a, const float4& b, uint4& hi, uint4& mid, uint4& lo) {
    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 hb = (__bf16)f[i];
        const float r1 = f[i] - (float)hb;
        const __bf16 mb = (__bf16)r1;
        const float r2 = r1 - (float)mb;
        const __bf16 lb = (__bf16)r2;
        h[i] = __builtin_bit_cast(unsigned short, hb);
        m[i] = __builtin_bit_cast(unsigned short, mb);
        l[i] = __builtin_bit_cast(unsigned short, lb);
    }
    hi = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    mid = make_uint4(m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16));
    lo = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}

// The rotation of two interleaved pairs, v = (a, b, c, d), t = (cos0, sin0, cos1, sin1): out = t * cos + rotate_half(t) * sin as the
// reference computes it (x-transformers apply_rotary_pos_emb: two multiplies and an add, each rounded).  Floating-point contraction is
// OFF here: left to the compiler, one instantiation of an epilogue fused a multiply-add where another did not, and tilings that
// accumulate K in the same order came out 3-6 bf16 roundings apart in 20 M outputs - big calls and small calls of the estimator must
// agree bit for bit (tests/test_flow_gpu.py::test_bench_sized_estimator_equals_pairs, the incremental stream's chunks).
__device__ __forceinline__ void rope_rotate(float4& v, const float4& t) {
#pragma clang fp contract(off)
    const float a = v.x, b = v.y, c = v.z, d = v.w;
    const float ac = a * t.x, bs = b * t.y, bc = b * t.x, as = a * t.y, cc = c * t.z, ds = d * t.w, dc = d * t.z, cs = c * t.w;
    v.x = ac - bs; v.y = bc + as;
    v.z = cc - ds; v.w = dc + cs;
}

// Sum over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15), the same value and the same order of additions in every lane: pairs,
// quads (quad permutes), the two quads of a half (half-row mirror), the two halves (row mirror) - vector-ALU data movement, no trip
// through the LDS crossbar that __shfl_xor takes (64 of those per 32-row block made the gated-residual epilogue 7 us slower)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));      // quad_perm [1, 0, 3, 2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));      // quad_perm [2, 3, 0, 1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));     // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));     // row_mirror
    return v;
}

// The folded LayerNorm-modulate (GemmEpi::ln_rows): out = rstd (acc - mean u) + b, every operation rounded on its own so that all
// epilogue forms agree bit for bit
__device__ __forceinline__ float ln_fold1(float acc, float mean, float rstd, float u, float b) {
#pragma clang fp contract(off)
    const float t = mean * u;
    const float d = acc - t;
    const float y = rstd * d;
    return y + b;
}
__device__ __forceinline__ void ln_fold4(float4& v, const float2& st, const float4& u, const float4& b) {
    v.x = ln_fold1(v.x, st.x, st.y, u.x, b.x); v.y = ln_fold1(v.y, st.x, st.y, u.y, b.y);
    v.z = ln_fold1(v.z, st.x, st.y, u.z, b.z); v.w = ln_fold1(v.w, st.x, st.y, u.w, b.w);
}

// x-transformers apply_rotary_pos_emb on four consecutive output columns starting at n (two interleaved pairs)
__device__ __forceinline__ void epi_rope(float4& v, const GemmEpi& e, int m, int n) {
    int nn = n >= e.rope_stride ? n - e.rope_stride : n;
    if (n >= 2 * e.rope_stride || nn >= 2 * e.rope_half) return;
    const float2* tab = e.rope + (long)((m / e.rope_div) % e.rope_T) * e.rope_half + (nn >> 1);
    const float2 c0 = tab[0], c1 = tab[1];
    rope_rotate(v, make_float4(c0.x, c0.y, c1.x, c1.y));
}

// Epilogue shared by the GEMM kernels, through LDS: the accumulator layout (column on the lane, rows in registers)
// would store 2-4 bytes per lane; each wave instead parks its 64x64 tile in LDS (the operand buffers are free by now)
// and reads it back row-wise, so every lane moves 4 consecutive columns: 16 wide stores per lane instead of 64 narrow ones.
template <int EPI>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[2][2], bf16_t* gm_smem, const GemmEpi& e, int M, int N, int m0, int n0,
                                              int wm, int wn, int wid, int lane) {
    const int lr = lane & 31, kh = lane >> 5;
    constexpr int EP = 68;                                   // fp32 pitch of the per-wave tile
    float* et = reinterpret_cast<float*>(gm_smem) + wid * 64 * EP;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) et[(mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * EP + ni * 32 + lr] = acc[mi][ni][r];
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): the wave reads back only its own tile
    __builtin_amdgcn_wave_barrier();
    const int c4 = (lane & 15) * 4, rsub = lane >> 4;
    const int n = n0 + wn * 64 + c4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), gv = bv;
    if (n < N) {                                             // N % 4 == 0 is checked on the host
        if (e.bias) bv = *reinterpret_cast<const float4*>(e.bias + n);
        if (EPI == 3) gv = *reinterpret_cast<const float4*>(e.gate + n);
    }
    // all LDS reads come first: in a kernel that also uses the LDS DMA the compiler fences every later LDS read with
    // vmcnt(0), which would make each row wait for the previous row's global store
    float4 vr[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) vr[it] = *reinterpret_cast<const float4*>(et + (it * 4 + rsub) * EP + c4);
    float4 old[16];
    if (EPI == 3) {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int m = m0 + wm * 64 + it * 4 + rsub;
            old[it] = (n < N && m < M) ? *reinterpret_cast<const float4*>(e.resid + (long)m * e.ldc + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int row = it * 4 + rsub, m = m0 + wm * 64 + row;
        float4 v = vr[it];
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (n >= N || m >= M) continue;
        if (EPI == 3) {
            float4 o = old[it];
            o.x = fmaf(gv.x, v.x, o.x); o.y = fmaf(gv.y, v.y, o.y); o.z = fmaf(gv.z, v.z, o.z); o.w = fmaf(gv.w, v.w, o.w);
            *reinterpret_cast<float4*>(e.resid + (long)m * e.ldc + n) = o;
        } else if (EPI == 2) {
            // fp32 output; the fp32-class mode of the flow decoder also wants the activation and the rotary embedding here,
            // with the exact tanh (the bf16 path's hardware-exp form is only good to bf16 precision)
            if (e.act == ACT_GELU_TANH) { v.x = act_gelu_tanh(v.x); v.y = act_gelu_tanh(v.y); v.z = act_gelu_tanh(v.z); v.w = act_gelu_tanh(v.w); }
            if (e.rope) epi_rope(v, e, m, n);
            *reinterpret_cast<float4*>((float*)e.out + (long)m * e.ldc + n) = v;
        } else {
            if (EPI == 1) { v.x = act_gelu_tanh_fast(v.x); v.y = act_gelu_tanh_fast(v.y); v.z = act_gelu_tanh_fast(v.z); v.w = act_gelu_tanh_fast(v.w); }
            if (EPI == 0 && e.rope) epi_rope(v, e, m, n);
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
            pk.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
            *reinterpret_cast<uint2*>((bf16_t*)e.out + (long)m * e.ldc + n) = pk;
        }
    }
}

// C[M,N] = A[M,K] W[N,K]^T.  128x128 block tile, 4 waves as 2x2, each wave 64x64 = 2x2 accumulators of
// v_mfma_f32_32x32x16_bf16; BK = 64 with two LDS buffers: the global loads of tile t+1 are issued before
// the 16 MFMAs of tile t and land in the other buffer after them - one barrier per K step.
// EPI: 0 store bf16, 1 store bf16 after GELU(tanh), 2 store fp32, 3 gated residual (resid += gate * (acc + bias))
// PRECISE: 0 = bf16 A operand; 2 = fp32 A split exactly into bf16 hi + lo on the fly (2 MFMAs per fragment: fp32-class);
// 3 = hi + mid + lo (3 MFMAs: every product exact with bf16-exact weights - the LM prefill, whose ids must track fp32)
template <int PRECISE, int EPI, int BM>
__global__ __launch_bounds__(BM * 2) void gemm_bf16_k(const void* __restrict__ Av, int lda, const bf16_t* __restrict__ W, int M, int N, int K, GemmEpi e) {
    extern __shared__ __attribute__((aligned(16))) bf16_t gm_smem[];
    constexpr int NT = BM * 2;                               // threads: (BM/64) x 2 waves, each a 64x64 tile
    constexpr int PL = PRECISE == 4 ? 2 : (PRECISE ? PRECISE : 1);      // planes of the A operand in LDS (4: hi + lo arrive as two bf16 planes)
    constexpr int A_ELEMS = PL * BM * GM_PITCH, B_ELEMS = GM_BN * GM_PITCH, BUF = A_ELEMS + B_ELEMS;
    constexpr int A_LOADS = BM * 8 / NT, B_LOADS = GM_BN * 8 / NT;      // 16-B chunks per thread per tile
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1, lr = lane & 31, kh = lane >> 5;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MiB L2), so
    // workgroup `orig` is given logical tile xcd*chunk + orig/8: every XCD walks a contiguous run of tiles,
    // N-tiles fastest, and the A rows it is working on are fetched into its L2 once instead of by all eight.
    const int ntn = (N + GM_BN - 1) / GM_BN, nwg = gridDim.x;
    const int orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * GM_BN;
    const bf16_t* Ab = (const bf16_t*)Av;
    const float* Af = (const float*)Av;

    uint4 ra[A_LOADS], ral[A_LOADS], ram[PRECISE == 3 ? A_LOADS : 1], rb[B_LOADS];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            int idx = tid + i * NT, row = idx >> 3, kc = (idx & 7) * 8;
            int gm = m0 + row;
            if (PRECISE == 4) {
                ra[i] = gm < M ? *reinterpret_cast<const uint4*>(Ab + (long)gm * lda + k0 + kc) : make_uint4(0, 0, 0, 0);
                ral[i] = gm < M ? *reinterpret_cast<const uint4*>(e.a_lo + (long)gm * lda + k0 + kc) : make_uint4(0, 0, 0, 0);
            } else if (PRECISE) {
                float4 x0 = make_float4(0, 0, 0, 0), x1 = x0;
                if (gm < M) {
                    const float* p = Af + (long)gm * lda + k0 + kc;
                    x0 = *reinterpret_cast<const float4*>(p);
                    x1 = *reinterpret_cast<const float4*>(p + 4);
                }
                if (PRECISE == 3) split8_3(x0, x1, ra[i], ram[i], ral[i]); else split8(x0, x1, ra[i], ral[i]);
            } else {
                ra[i] = gm < M ? *reinterpret_cast<const uint4*>(Ab + (long)gm * lda + k0 + kc) : make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            int idx = tid + i * NT, row = idx >> 3, kc = (idx & 7) * 8;
            int gn = n0 + row;
            rb[i] = gn < N ? *reinterpret_cast<const uint4*>(W + (long)gn * K + k0 + kc) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&](bf16_t* buf) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            int idx = tid + i * NT, row = idx >> 3, kc = (idx & 7) * 8;
            *reinterpret_cast<uint4*>(buf + row * GM_PITCH + kc) = ra[i];
            if (PRECISE) *reinterpret_cast<uint4*>(buf + BM * GM_PITCH + row * GM_PITCH + kc) = ral[i];
            if (PRECISE == 3) *reinterpret_cast<uint4*>(buf + 2 * BM * GM_PITCH + row * GM_PITCH + kc) = ram[i];
        }
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i) {
            int idx = tid + i * NT, row = idx >> 3, kc = (idx & 7) * 8;
            *reinterpret_cast<uint4*>(buf + A_ELEMS + row * GM_PITCH + kc) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    load_tile(0);
    store_tile(gm_smem);
    __syncthreads();
    const int nt = K / GM_BK;
    for (int t = 0; t < nt; ++t) {
        const bool more = t + 1 < nt;
        if (more) load_tile((t + 1) * GM_BK);
        const bf16_t* As = gm_smem + (t & 1) * BUF;
        const bf16_t* Bs = As + A_ELEMS;
#pragma unroll
        for (int ks = 0; ks < GM_BK / 16; ++ks) {
            frag_ab a[2], al[2], am[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                int off = (wm * 64 + mi * 32 + lr) * GM_PITCH + ks * 16 + kh * 8;
                a[mi] = *reinterpret_cast<const frag_ab*>(As + off);
                if (PRECISE) al[mi] = *reinterpret_cast<const frag_ab*>(As + BM * GM_PITCH + off);
                if (PRECISE == 3) am[mi] = *reinterpret_cast<const frag_ab*>(As + 2 * BM * GM_PITCH + off);
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                b[ni] = *reinterpret_cast<const frag_ab*>(Bs + (wn * 64 + ni * 32 + lr) * GM_PITCH + ks * 16 + kh * 8);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
                    if (PRECISE == 3) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[mi], b[ni], acc[mi][ni], 0, 0, 0);
                    if (PRECISE) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], b[ni], acc[mi][ni], 0, 0, 0);
                }
        }
        if (more) store_tile(gm_smem + ((t + 1) & 1) * BUF);
        __syncthreads();
    }

    gemm_epilogue<EPI>(acc, gm_smem, e, M, N, m0, n0, wm, wn, wid, lane);
}

// 256x256x32 tile, 8 waves as 2 (M) x 4 (N), each wave 128x64 = 4x2 accumulators of v_mfma_f32_32x32x16_bf16,
// operands copied global -> LDS by the DMA path (global_load_lds, 16 B per lane, no staging registers) into a ring of
// three 32 KB stages that runs two K steps ahead (a fourth stage measured no faster end to end and leaves less LDS for
// the LM's kernels that share the CUs in the pipelined benchmark).
// Why this shape (measured on MI355X with tests/micro/gemm_bench at the DiT shapes, M = 6400): operand fetches take
// ~1.5 us to land while every CU streams, so a CU's MFMA rate is (bytes it keeps in flight) x (flops per byte of tile)
// / latency.  The 128x128x64 tile with one step of lookahead keeps 32 KB in flight per workgroup (K step 2.7k cycles
// against 512 of MFMA); this ring keeps 64 KB in flight at twice the flops per byte.
// The DMA writes a wave-instruction's 64 x 16 bytes linearly (16 rows of 64 B), so rows carry no padding; bank conflicts
// are avoided by an XOR swizzle applied to the SOURCE chunk each lane fetches and to the fragment reads: chunk c of
// row r lives at slot c ^ ((r >> 2) & 3).  Counted s_waitcnt vmcnt(4) + raw s_barrier keep a stage in flight across
// the barrier (__syncthreads() would drain them).
#define G2_BK 32
// chunk c (16 bytes) of tile row r lives at slot c ^ G2_SW(r) of the row's 64 bytes: conflict-free for the fragment reads of both
// MFMA shapes (32x32x16: 32 rows x one chunk per half-wave; 16x16x32: 16 rows x all four chunks)
#define G2_SW(row) ((0 - ((row) >> 2)) & 3)
// -DFY_GEMM_STAMPS (tests/micro/gemm_bench.hip): wave 0 of every workgroup records the 100 MHz clock at entry, when the first
// stage has landed, after the K loop and after the epilogue
#ifdef FY_GEMM_STAMPS
__device__ unsigned long long* gemm_stamp_buf;
void gemm_set_stamps(unsigned long long* p) { hipMemcpyToSymbol(HIP_SYMBOL(gemm_stamp_buf), &p, sizeof(p)); }
#define G2_STAMP(i) do { if (tid == 0 && gemm_stamp_buf) gemm_stamp_buf[(e.stamp_slot * 4096 + blockIdx.x) * 4 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define G2_STAMP(i) do { } while (0)
#endif
#ifndef G2_INTERLEAVE
#define G2_INTERLEAVE 1
#endif
#ifndef G2_DIRECT
#define G2_DIRECT 1
#endif
#define G2_RING_BYTES (96 * 1024)                            // BN 256: three 32 KB stages; BN 128: three 24 KB stages = 72 KB, so that two workgroups share a CU

// The folded LayerNorm's row statistics (gemm.h): thread r of the workgroup turns the producer's slots of tile row r into (mean, rstd)
// at smem + ln_off - slots summed in slot order, in double (E[x^2] - mean^2 wants it), eps as the flow decoder's LayerNorm.  Called
// before the K loop (whose barriers make the values visible to the epilogue); the loads fly under the first stage's.
__device__ __forceinline__ void ring_ln_rows(char* smem, int ln_off, const GemmEpi& e, int M, int m0, int BM, int tid, int nthreads) {
    const int n_slots = e.ln_dim >> 6;
    for (int r = tid; r < BM; r += nthreads) {
        const float2* sp = e.ln_rows_slots + (long)min(m0 + r, M - 1) * n_slots;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
        for (int k = 0; k < n_slots; ++k) { const float2 v = sp[k]; s1 += (double)v.x; s2 += (double)v.y; }
        // (one double reciprocal per thread; the root as v_rsq_f32 + one Newton step: a thread's statistics sit before its workgroup's K
        // loop, and the two double divisions + double root this replaces cost every qkv / ff1 launch 2-3 us)
        const double inv = 1.0 / (double)e.ln_dim, mean = s1 * inv, var = fmax(s2 * inv - mean * mean, 0.0);
        const float v = (float)(var + 1e-6);
        float rs = __builtin_amdgcn_rsqf(v);
        rs = rs * (1.5f - 0.5f * v * rs * rs);
        reinterpret_cast<float2*>(smem + ln_off)[r] = make_float2((float)mean, rs);
    }
}

// The ring kernels' epilogue (gemm256_k, gemm64_k): from the accumulators - acc (32x32x16 form) or acc4 (16x16x32 form; DE: the
// transposed products, a lane holds 16 consecutive columns of one row per 16-row tile) - to memory, with the bias, the rotary
// embedding, GELU, the gated residual or the split planes.  Called by every wave of the workgroup after the K loop.
// FM = 16-row tiles per wave (the wave covers FM * 16 rows x 64 columns); AH: the epilogue may fetch residual / table rows a tile ahead
// (the one-workgroup-per-CU tilings of 8 waves have the registers for it)
template <int EPI, int BN, int FM, int MF, bool DE, bool AH>
__device__ __forceinline__ void ring_epilogue(char* smem, f32x16 (&acc)[(FM + 1) / 2][2], f32x4 (&acc4)[MF ? FM : 1][4], const GemmEpi& e, int M, int N,
                                              int m0, int n0, int wm, int wn, int wid, int lane, int ln_off = 0) {
    // (ln_off: where in LDS the caller's ring_ln_rows left the (mean, rstd) of the tile's rows, when e.ln_rows_slots is set)
    // the thread's coordinates pass through an empty asm: the compiler then cannot compute the epilogue's addresses (bias, table and
    // residual rows, ...) ABOVE the caller's K loop and carry them through it - in the 128-register kernels it did, and spilled them
    asm volatile("" : "+v"(lane), "+v"(wm), "+v"(wn), "+v"(wid));
    const int tid = threadIdx.x, lr = lane & 31, kh = lane >> 5;
    (void)tid; (void)lr; (void)kh;
    constexpr int MI = FM / 2;                                      // 32-row blocks (the LDS epilogue walks them: FM even there)
    static_assert(DE || FM % 2 == 0, "the LDS epilogue parks 32-row blocks");
    __syncthreads();                                                // all waves are done with the ring: the epilogue parks tiles in it
    G2_STAMP(2);

    constexpr bool RE = EPI == 0 || EPI == 4;                       // epilogues that may carry the rotary embedding
    if constexpr (DE) {
        // Direct epilogue: for 16-row tile i the lane holds row (lane & 15), columns nq .. nq + 15 as acc4[i][j][r] = column 4 j + r.
        // The residual rows (EPI 3) or the rotary table rows (RE) of tile i + 1 are fetched before tile i is stored.
        const int nq = n0 + wn * 64 + (lane >> 4) * 16, mw = m0 + wm * (FM * 16) + (lane & 15);
        float4 bq[4], gq[EPI == 3 ? 4 : 1], uq[EPI == 0 || EPI == 1 ? 4 : 1];
        const bool lnf = (EPI == 0 || EPI == 1) && e.ln_rows_slots != nullptr;    // the folded LayerNorm-modulate (gemm.h)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bq[c] = e.bias ? *reinterpret_cast<const float4*>(e.bias + nq + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (EPI == 3) gq[c] = *reinterpret_cast<const float4*>(e.gate + nq + 4 * c);
            if (EPI == 0 || EPI == 1) uq[c] = lnf ? *reinterpret_cast<const float4*>(e.ln_u + nq + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        int rope_c = -1;                                            // float2 index of the lane's first pair inside a table row
        if (RE && e.rope) {
            const int nn = nq >= e.rope_stride ? nq - e.rope_stride : nq;
            if (nq < 2 * e.rope_stride && nn < 2 * e.rope_half) rope_c = nn >> 1;
        }
        constexpr bool PF = EPI == 3 || RE, AHEAD = AH;      // the one-per-CU tiles have the registers to fetch a tile ahead, and nobody else to hide the trip
        float4 cur[PF ? 4 : 1], nxt[PF && AHEAD ? 4 : 1];
        auto fetch = [&](int i, auto& t) {
            const int m = min(mw + i * 16, M - 1);
            if constexpr (EPI == 3) {
#pragma unroll
                for (int c = 0; c < 4; ++c) t[c] = *reinterpret_cast<const float4*>(e.resid + (long)m * e.ldc + nq + 4 * c);
            } else if constexpr (RE) {
                if (rope_c >= 0) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) t[c] = *reinterpret_cast<const float4*>(e.rope + (long)((m / e.rope_div) % e.rope_T) * e.rope_half + rope_c + 2 * c);
                }
            }
        };
        // the folded LayerNorm's (mean, rstd) of this lane's FM rows: requested together, ahead of the tile loop
        float2 lrow[(EPI == 0 || EPI == 1) ? FM : 1];
        if constexpr (EPI == 0 || EPI == 1) {
            if (lnf) {
#pragma unroll
                for (int i = 0; i < FM; ++i) lrow[i] = reinterpret_cast<const float2*>(smem + ln_off)[wm * (FM * 16) + (lane & 15) + i * 16];
            }
        }
        if constexpr (PF && AHEAD) fetch(0, cur);
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            if constexpr (PF && AHEAD) { if (i + 1 < FM) fetch(i + 1, nxt); }
            if constexpr (PF && !AHEAD) fetch(i, cur);
            const int m = mw + i * 16;
            if (m < M) {
                uint32_t pk[8], pl[8];
                float2 lst = make_float2(0.f, 1.f);
                if constexpr (EPI == 0 || EPI == 1) { if (lnf) lst = lrow[i]; }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float4 v = make_float4(acc4[i][c][0], acc4[i][c][1], acc4[i][c][2], acc4[i][c][3]);
                    if ((EPI == 0 || EPI == 1) && lnf) ln_fold4(v, lst, uq[c], bq[c]);
                    else { v.x += bq[c].x; v.y += bq[c].y; v.z += bq[c].z; v.w += bq[c].w; }
                    if constexpr (EPI == 3) {
                        float4 o = cur[c];
                        o.x = fmaf(gq[c].x, v.x, o.x); o.y = fmaf(gq[c].y, v.y, o.y); o.z = fmaf(gq[c].z, v.z, o.z); o.w = fmaf(gq[c].w, v.w, o.w);
                        *reinterpret_cast<float4*>(e.resid + (long)m * e.ldc + nq + 4 * c) = o;
                    } else if constexpr (EPI == 2) {
                        *reinterpret_cast<float4*>((float*)e.out + (long)m * e.ldc + nq + 4 * c) = v;
                    } else {
                        if (EPI == 5) { v.x = act_gelu_tanh(v.x); v.y = act_gelu_tanh(v.y); v.z = act_gelu_tanh(v.z); v.w = act_gelu_tanh(v.w); }
                        if (EPI == 1) { v.x = act_gelu_tanh_fast(v.x); v.y = act_gelu_tanh_fast(v.y); v.z = act_gelu_tanh_fast(v.z); v.w = act_gelu_tanh_fast(v.w); }
                        if constexpr (RE) {
                            if (rope_c >= 0) {                       // x-transformers apply_rotary_pos_emb on two interleaved pairs (epi_rope's arithmetic)
                                rope_rotate(v, cur[c]);
                            }
                        }
                        const bf16_t h0 = f32_to_bf16(v.x), h1 = f32_to_bf16(v.y), h2 = f32_to_bf16(v.z), h3 = f32_to_bf16(v.w);
                        pk[2 * c] = (uint32_t)h0 | ((uint32_t)h1 << 16);
                        pk[2 * c + 1] = (uint32_t)h2 | ((uint32_t)h3 << 16);
                        if constexpr (EPI == 4 || EPI == 5) {        // x = hi + lo as two bf16 planes for the next split-operand product
                            pl[2 * c] = (uint32_t)f32_to_bf16(v.x - bf16_to_f32(h0)) | ((uint32_t)f32_to_bf16(v.y - bf16_to_f32(h1)) << 16);
                            pl[2 * c + 1] = (uint32_t)f32_to_bf16(v.z - bf16_to_f32(h2)) | ((uint32_t)f32_to_bf16(v.w - bf16_to_f32(h3)) << 16);
                        }
                    }
                }
                if constexpr (EPI != 3 && EPI != 2) {
                    bf16_t* o = (bf16_t*)e.out + (long)m * e.ldc + nq;
                    *reinterpret_cast<uint4*>(o) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                    *reinterpret_cast<uint4*>(o + 8) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
                    if constexpr (EPI == 4 || EPI == 5) {
                        bf16_t* ol = (bf16_t*)e.out_lo + (long)m * e.ldc + nq;
                        *reinterpret_cast<uint4*>(ol) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
                        *reinterpret_cast<uint4*>(ol + 8) = make_uint4(pl[4], pl[5], pl[6], pl[7]);
                    }
                }
            }
            if constexpr (PF && AHEAD) {
#pragma unroll
                for (int c = 0; c < 4; ++c) cur[c] = nxt[c];
            }
        }
        G2_STAMP(3);
        return;
    }
    // epilogue through LDS, a quarter of the wave's tile (32 rows x 64 columns) at a time: see gemm_epilogue
    float* park = reinterpret_cast<float*>(smem) + wid * 32 * 68;
    const int c4 = (lane & 15) * 4, rsub = lane >> 4;
    const int n = n0 + wn * 64 + c4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), gv = bv, uv = bv;
    if (e.bias) bv = *reinterpret_cast<const float4*>(e.bias + n);
    if (EPI == 3) gv = *reinterpret_cast<const float4*>(e.gate + n);
    const bool lnf = (EPI == 0 || EPI == 1) && e.ln_rows_slots != nullptr;        // the folded LayerNorm-modulate (gemm.h)
    if (lnf) uv = *reinterpret_cast<const float4*>(e.ln_u + n);
    // Rotary embedding (EPI 0): only the lanes whose four columns lie in a rotated range fetch (cos, sin) pairs, and they fetch
    // a 32-row block's eight table rows in one batch, one block AHEAD of their use (under the previous block's stores) - fetched
    // where they are used, each row's 16 bytes cost a trip to L2 in the middle of the store loop (qkv at M = 6400: +11 us on the
    // 320x256 tile, whose epilogue nothing hides).
    int rope_col = -1;                                              // float2 index of this lane's first pair inside a table row
    if (RE && e.rope) {
        const int nn = n >= e.rope_stride ? n - e.rope_stride : n;
        if (n < 2 * e.rope_stride && nn < 2 * e.rope_half) rope_col = nn >> 1;
    }
    constexpr bool RAHEAD = AH;                             // the one-per-CU tiles have the registers for it; at 128 they would cost a workgroup per CU
    float4 rt[8], rn[RAHEAD ? 8 : 1];
    auto rope_fetch = [&](int mi, float4 (&t)[8]) {
        if (rope_col < 0) return;
        const int m00 = m0 + wm * (FM * 16) + mi * 32 + rsub;
#pragma unroll
        for (int it = 0; it < 8; ++it)
            t[it] = *reinterpret_cast<const float4*>(e.rope + (long)((min(m00 + it * 4, M - 1) / e.rope_div) % e.rope_T) * e.rope_half + rope_col);
    };
    if (RE && RAHEAD) rope_fetch(0, rt);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        if (RE && !RAHEAD) rope_fetch(mi, rt);                   // before the block's trip through LDS
        if constexpr (MF) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) park[(i * 16 + 4 * (lane >> 4) + r) * 68 + j * 16 + (lane & 15)] = acc4[2 * mi + i][j][r];
        } else
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) park[((r & 3) + 8 * (r >> 2) + 4 * kh) * 68 + ni * 32 + lr] = acc[mi][ni][r];
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): the wave reads back only its own quarter tile
        __builtin_amdgcn_wave_barrier();
        // the read-back is inline asm: in a kernel that uses the LDS DMA the compiler fences every LDS read it can see with
        // vmcnt(0), which would make each quarter wait for the previous quarter's global stores to retire
        f32x4 vq[8];
        const uint32_t pa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)(park + rsub * 68 + c4);
#pragma unroll
        for (int it = 0; it < 8; ++it) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(vq[it]) : "v"(pa), "n"(it * 4 * 68 * 4) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(vq[0]), "+v"(vq[1]), "+v"(vq[2]), "+v"(vq[3]), "+v"(vq[4]), "+v"(vq[5]), "+v"(vq[6]), "+v"(vq[7]) : : "memory");
        float4 vr[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) vr[it] = make_float4(vq[it][0], vq[it][1], vq[it][2], vq[it][3]);
        __builtin_amdgcn_wave_barrier();                         // the next quarter overwrites the park region
        if constexpr (RE && RAHEAD) { if (mi + 1 < MI) rope_fetch(mi + 1, rn); }     // travels under this block's stores
        const int mb = m0 + wm * (FM * 16) + mi * 32 + rsub;
        float4 old[8];
        if (EPI == 3) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int m = min(mb + it * 4, M - 1);
                old[it] = *reinterpret_cast<const float4*>(e.resid + (long)m * e.ldc + n);
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int m = mb + it * 4;
            float4 v = vr[it];
            if (lnf) ln_fold4(v, reinterpret_cast<const float2*>(smem + ln_off)[m - m0], uv, bv);
            else { v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
            if (m >= M) continue;
            if (EPI == 3) {
                float4 o = old[it];
                o.x = fmaf(gv.x, v.x, o.x); o.y = fmaf(gv.y, v.y, o.y); o.z = fmaf(gv.z, v.z, o.z); o.w = fmaf(gv.w, v.w, o.w);
                *reinterpret_cast<float4*>(e.resid + (long)m * e.ldc + n) = o;
                if (e.h_bf16) {
                    // what the NEXT product reads when LayerNorm-modulate is folded across it (gemm.h): bf16 of the new residual row piece,
                    // and this wave's 64 columns' (sum, sum of squares) of the row - the 16 lanes of a row meet by shuffles in a fixed order
                    uint2 hb;
                    hb.x = (uint32_t)f32_to_bf16(o.x) | ((uint32_t)f32_to_bf16(o.y) << 16);
                    hb.y = (uint32_t)f32_to_bf16(o.z) | ((uint32_t)f32_to_bf16(o.w) << 16);
                    *reinterpret_cast<uint2*>(e.h_bf16 + (long)m * e.ldc + n) = hb;
                    float s1 = (o.x + o.y) + (o.z + o.w), s2 = (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
                    s1 = row16_sum(s1); s2 = row16_sum(s2);
                    if ((lane & 15) == 0) e.ln_slots[(long)m * (N >> 6) + ((n0 + wn * 64) >> 6)] = make_float2(s1, s2);
                }
            } else if (EPI == 2) {
                *reinterpret_cast<float4*>((float*)e.out + (long)m * e.ldc + n) = v;
            } else if (EPI == 4 || EPI == 5) {
                // fp32-class result for the next split-operand product: exact-tanh GELU (5) or the rotary embedding (4) on the fp32
                // sums, then x = hi + lo as two bf16 planes
                if (EPI == 5) { v.x = act_gelu_tanh(v.x); v.y = act_gelu_tanh(v.y); v.z = act_gelu_tanh(v.z); v.w = act_gelu_tanh(v.w); }
                if (EPI == 4 && rope_col >= 0) {
                    rope_rotate(v, rt[it]);
                }
                const bf16_t h0 = f32_to_bf16(v.x), h1 = f32_to_bf16(v.y), h2 = f32_to_bf16(v.z), h3 = f32_to_bf16(v.w);
                uint2 ph, pl;
                ph.x = (uint32_t)h0 | ((uint32_t)h1 << 16);
                ph.y = (uint32_t)h2 | ((uint32_t)h3 << 16);
                pl.x = (uint32_t)f32_to_bf16(v.x - bf16_to_f32(h0)) | ((uint32_t)f32_to_bf16(v.y - bf16_to_f32(h1)) << 16);
                pl.y = (uint32_t)f32_to_bf16(v.z - bf16_to_f32(h2)) | ((uint32_t)f32_to_bf16(v.w - bf16_to_f32(h3)) << 16);
                *reinterpret_cast<uint2*>((bf16_t*)e.out + (long)m * e.ldc + n) = ph;
                *reinterpret_cast<uint2*>((bf16_t*)e.out_lo + (long)m * e.ldc + n) = pl;
            } else {
                if (EPI == 1) { v.x = act_gelu_tanh_fast(v.x); v.y = act_gelu_tanh_fast(v.y); v.z = act_gelu_tanh_fast(v.z); v.w = act_gelu_tanh_fast(v.w); }
                if (EPI == 0 && rope_col >= 0) {                 // x-transformers apply_rotary_pos_emb on two interleaved pairs (epi_rope's arithmetic)
                    rope_rotate(v, rt[it]);
                }
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
                pk.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
                *reinterpret_cast<uint2*>((bf16_t*)e.out + (long)m * e.ldc + n) = pk;
            }
        }
        if constexpr (RE && RAHEAD) {
            if (rope_col >= 0) {
#pragma unroll
                for (int it = 0; it < 8; ++it) rt[it] = rn[it];
            }
        }
    }
    G2_STAMP(3);
}

// BN = 256: 8 waves as 2 (M) x 4 (N), 128x64 (BM 256) or 160x64 (BM 320) per wave.  BN = 128: 4 x 2 waves of 64x64, <= 128 VGPRs
// and a 72 KB ring, so two workgroups are resident per CU: one's prologue / epilogue bursts run under the other's K loop (the
// four DiT products of a block: 184 -> 168 us), and the N = 1024 products get twice the tiles.
// STAG: the two wave groups of a one-per-CU workgroup run half a K step apart (comment at the loop).  MF = 1: the products on
// v_mfma_f32_16x16x32_bf16 instead of 32x32x16 - the same cycles per flop, the same bits out, but the chip holds a higher clock
// on it (MI355X_MICROARCH.md, DVFS give-back item 7): 5-19 % less time at two or more workgroups per CU.
// SP = 1: the split-operand form (gemm_split): the stage holds TWO A tiles, the hi and the lo plane of x = hi + lo, and every
// fragment pair meets its B fragment in two MFMAs that accumulate into the same registers (MF = 1 only).
// NST: stages of the ring.  3 at the sizes above (two or three workgroups per CU hide each other's waits).  A launch with fewer tiles
// than CUs (stream=True chunks: a hundred to a few hundred rows) is bound by memory latency x the bytes ONE workgroup keeps in flight
// (two 16 KB stages: 512 KB of operands in ~15 us); NST = 8 keeps seven stages in flight in the LDS nobody else on the CU wants.
// Same K order, same bits.
template <int EPI, int BN, int BM = 256, int STAG = 0, int MF = 0, int SP = 0, int NST = 3>
__global__ __launch_bounds__(512) void gemm256_k(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int M, int N, int K, GemmEpi e) {
    extern __shared__ __attribute__((aligned(16))) bf16_t gm_smem[];
    char* smem = reinterpret_cast<char*>(gm_smem);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int NWAVE = BM == 128 ? 4 : 8;                             // BM 128 (with BN 128): 2 x 2 waves of 64x64, 48 KB ring, three workgroups per CU
    constexpr int WN = BN / 64, WM = NWAVE / WN, MI = BM / WM / 32;       // waves along N / M, 32-row tiles per wave
    static_assert(!SP || (MF && !STAG), "the split-operand forms exist for the 16x16x32 products without staggered wave groups");
    constexpr int APL = SP + 1;                                           // A planes per stage: 1, 2 (hi + lo: gemm_split) or 3 (hi + mid + lo: gemm_exact3)
    // DE: the direct epilogue of the 16x16x32 form (at the end of the kernel).  The products are taken TRANSPOSED - the W fragment
    // as the instruction's first operand - so a lane ends up with one output row's values, and W rows are dealt to the four
    // column tiles j as local column (c >> 2) * 16 + 4 j + (c & 3) for instruction index c, so those values are 16 CONSECUTIVE
    // columns: stored from the registers in full 128-byte lines, no trip through LDS.  Same products, same K order: same bits.
    // Its B fragment reads 4 rows out of every 16, so rows 16 apart must sit in different slots: the swizzle gets a (row >> 4) term.
    constexpr bool DE = MF && G2_DIRECT && EPI != 2 && EPI != 3;      // fp32 outputs (64-byte runs per row from here) stay on the LDS epilogue's full lines: out-proj 24 -> 32 us with this one
    auto SW = [](int row) -> int { return DE ? (G2_SW(row) ^ ((0 - (row >> 4)) & 3)) : G2_SW(row); };
    constexpr int STAGE = (APL * BM + BN) * G2_BK * 2, STAGES = NST, NB = BN / (NWAVE * 16);   // NB: B DMAs per wave and stage
    static_assert(NST == 3 || (!STAG && !(MF && G2_INTERLEAVE) && NST <= 9), "deeper rings: the plain loop only");
    constexpr int NA_ALL = APL * BM / 16, NA = (NA_ALL + NWAVE - 1) / NWAVE, NA_LAST = NA_ALL - (NA - 1) * NWAVE;   // A DMAs: wave-instruction j = i * NWAVE + wid fills rows [16 j, 16 j + 16) (SP: of plane j / (BM / 16)); the last round only on waves < NA_LAST
    constexpr int BOFF = APL * BM * G2_BK * 2;                            // the B tile follows the A tile(s) inside a stage
    const int wm = wid / WN, wn = wid % WN, lr = lane & 31, kh = lane >> 5;
    // XCD-aware tile order (see gemm_bf16_k)
    const int ntn = N / BN, nwg = gridDim.x;
    const int orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;
    G2_STAMP(0);
    constexpr int LN_OFF = NST * ((SP + 1) * BM + BN) * G2_BK * 2;       // behind the ring (the launcher adds BM x 8 bytes for it)
    if constexpr (EPI == 0 || EPI == 1) { if (e.ln_rows_slots) ring_ln_rows(smem, LN_OFF, e, M, m0, BM, tid, NWAVE * 64); }

    // K steps: K / 32 - or, for the split-operand form with a second weight plane (e.w_lo), twice that: steps [ntk, 2 ntk) walk
    // the same A columns again against the lo plane of W (same rows, same pitch: one pointer difference)
    const int ntk = K / G2_BK;
    const bool w2 = SP == 1 && e.w_lo != nullptr;
    const int nt = w2 ? 2 * ntk : ntk;
    const long wdelta = w2 ? (long)(e.w_lo - W) : 0L;
    auto kofs = [&](int t) -> int { if constexpr (SP == 1) return (t >= ntk ? t - ntk : t) * G2_BK; else return t * G2_BK; };
    auto wofs = [&](int t) -> long { if constexpr (SP == 1) return t >= ntk ? wdelta : 0L; else return 0L; };
    // wave-instruction j fills stage bytes [j * 1024, +1024) of the A (and B) tile:
    // row = j*16 + lane/4, slot = lane%4  ->  logical chunk = slot ^ ((row >> 2) & 3)
    const bf16_t* a_src[NA];
    const bf16_t* b_src[NB];
    const bool a_last = NA_LAST == NWAVE || wid < NA_LAST;           // wave-uniform: this wave takes part in the last round of A DMAs
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int j = i * NWAVE + wid;
        const int row = (SP ? j % (BM / 16) : j) * 16 + (lane >> 2), c = (lane & 3) ^ SW(row);
        const int pl = SP ? j / (BM / 16) : 0;                         // wave-uniform: the plane this wave-instruction fills
        const bf16_t* Ap = pl == 0 ? A : (pl == 1 ? e.a_lo : e.a_lo2);
        a_src[i] = Ap + (long)min(m0 + row, M - 1) * lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int row = (wid * NB + i) * 16 + (lane >> 2), c = (lane & 3) ^ SW(row);
        b_src[i] = W + (long)min(n0 + row, N - 1) * K + c * 8;
    }
    // DMA d of a stage (0 .. NA-1: A rows, NA .. NA+NB-1: B rows) on its own: the K loop of the 16x16x32 form issues a stage's DMAs
    // one at a time between its MFMA groups (below)
    auto issue_one = [&](int t, int d) {
        char* st = smem + (t % STAGES) * STAGE;
        if (d < NA) {
            if (d + 1 < NA || a_last)
                __builtin_amdgcn_global_load_lds((const void*)(a_src[d] + kofs(t)), (__attribute__((address_space(3))) void*)(st + (d * NWAVE + wid) * 1024), 16, 0, 0);
        } else
            __builtin_amdgcn_global_load_lds((const void*)(b_src[d - NA] + wofs(t) + kofs(t)), (__attribute__((address_space(3))) void*)(st + BOFF + wid * NB * 1024 + (d - NA) * 1024), 16, 0, 0);
    };
    auto issue = [&](int t) {
        char* st = smem + (t % STAGES) * STAGE;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if (i + 1 < NA || a_last)
                __builtin_amdgcn_global_load_lds((const void*)(a_src[i] + kofs(t)), (__attribute__((address_space(3))) void*)(st + (i * NWAVE + wid) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_global_load_lds((const void*)(b_src[i] + wofs(t) + kofs(t)), (__attribute__((address_space(3))) void*)(st + BOFF + wid * NB * 1024 + i * 1024), 16, 0, 0);
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // fragment read offsets inside a stage (bytes): row*64 + ((2ks + kh) ^ ((row>>2)&3))*16
    int a_off[MI], b_off[2], a_sw[MI], b_sw[2];
#pragma unroll
    for (int i = 0; i < MI; ++i) { const int ra = wm * (MI * 32) + i * 32 + lr; a_off[i] = ra * 64; a_sw[i] = G2_SW(ra); }
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int rb = wn * 64 + i * 32 + lr; b_off[i] = BOFF + rb * 64; b_sw[i] = G2_SW(rb); }
    // the 16x16x32 form: fragment i covers 16 rows, lane l reads row (l & 15), chunk (l >> 4): one read per fragment and K step
    int a16[MF ? 2 * MI : 1], b16[MF ? 4 : 1];
    f32x4 acc4[MF ? 2 * MI : 1][4];
    if constexpr (MF) {
#pragma unroll
        for (int i = 0; i < 2 * MI; ++i) { const int ra = wm * (MI * 32) + i * 16 + (lane & 15); a16[i] = ra * 64 + (((lane >> 4) ^ SW(ra)) << 4); }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rb = wn * 64 + (DE ? ((lane & 15) >> 2) * 16 + i * 4 + (lane & 3) : i * 16 + (lane & 15));
            b16[i] = BOFF + rb * 64 + (((lane >> 4) ^ SW(rb)) << 4);
        }
#pragma unroll
        for (int i = 0; i < 2 * MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < nt) issue(t);
    if constexpr (STAG) {
        // One workgroup per CU: its two waves on a SIMD (wid and wid + 4) would read fragments together and run MFMAs together,
        // each waiting while the other kind of work could run.  Here the K step is cut into a read interval and an MFMA interval
        // with a barrier after each, and waves 4-7 run one interval behind waves 0-3: on every SIMD one wave reads while the
        // other computes.  Interval numbering I_k (between barriers k-1 and k), K step t:
        //   waves 0-3: I_2t   issue(t+2), read fragments(t)              I_2t+1  MFMAs(t), wait for own DMAs of stage t+1
        //   waves 4-7: I_2t+1 read fragments(t), wait for stage t+1      I_2t+2  issue(t+3), MFMAs(t)
        // Stage t+2 (or t+3) goes into the slot of stage t-1 (t), which both groups have read before barrier 2t-1 (2t+1);
        // every wave has waited for its DMAs of stage t before barrier 2t-1, one barrier before the first read of it.
        constexpr int D = NA + NB;
        auto wait_stage = [&](int t) {
            if (t >= nt) return;
            if (t + 1 >= nt) __builtin_amdgcn_s_waitcnt(0x0F70);
            else if (a_last) __builtin_amdgcn_s_waitcnt(0x0F70 | D);
            else __builtin_amdgcn_s_waitcnt(0x0F70 | (D - 1));
        };
        frag_ab fa[2][MI], fb[2][2];
        auto reads = [&](int t) {
            const char* sb = smem + (t % STAGES) * STAGE;
            if constexpr (MF) {
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j >> 1][j & 1] = *reinterpret_cast<const frag_ab*>(sb + b16[j]);
#pragma unroll
                for (int i = 0; i < 2 * MI; ++i) fa[i & 1][i >> 1] = *reinterpret_cast<const frag_ab*>(sb + a16[i]);
            } else
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) fb[ks][ni] = *reinterpret_cast<const frag_ab*>(sb + b_off[ni] + (((2 * ks + kh) ^ b_sw[ni]) << 4));
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) fa[ks][mi] = *reinterpret_cast<const frag_ab*>(sb + a_off[mi] + (((2 * ks + kh) ^ a_sw[mi]) << 4));
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);                     // lgkmcnt(0): the slot may be refilled after the next barrier
        };
        // it >= 0: stage `it` is requested during the MFMA interval, one DMA after every few MFMAs - a DMA's issue waits for the
        // CU's address path, and in the read interval (five in a row, beside 14 fragment reads) that wait was the interval
        auto mfmas = [&](int it) {
            constexpr int NM = MF ? 2 * MI * 4 : 2 * MI * 2, EV = NM / (D + 1);      // MFMAs of the interval; one DMA after every EV of them
            int q = 0;
            __builtin_amdgcn_s_setprio(1);
            auto between = [&]() {
                ++q;
                if constexpr (G2_INTERLEAVE) {
                    if (q % EV == 0 && q / EV - 1 < D) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (it >= 0) issue_one(it, q / EV - 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            if constexpr (MF) {
#pragma unroll
                for (int i = 0; i < 2 * MI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc4[i][j] = DE ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j >> 1][j & 1], fa[i & 1][i >> 1], acc4[i][j], 0, 0, 0)
                                        : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i & 1][i >> 1], fb[j >> 1][j & 1], acc4[i][j], 0, 0, 0);
                        between();
                    }
            } else
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][mi], fb[ks][ni], acc[mi][ni], 0, 0, 0);
                        between();
                    }
            __builtin_amdgcn_s_setprio(0);
        };
        wait_stage(0);
        __builtin_amdgcn_s_barrier();                               // barrier -1
        G2_STAMP(1);
        if (wid < NWAVE / 2) {
            for (int t = 0; t < nt; ++t) {
                if (!G2_INTERLEAVE && t + 2 < nt) issue(t + 2);
                reads(t);
                __builtin_amdgcn_s_barrier();                       // 2t
                __builtin_amdgcn_sched_barrier(0);
                mfmas(G2_INTERLEAVE && t + 2 < nt ? t + 2 : -1);
                wait_stage(t + 1);
                __builtin_amdgcn_s_barrier();                       // 2t + 1
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_barrier();                           // 2 nt: the other group's last interval
        } else {
            if (2 < nt) issue(2);
            __builtin_amdgcn_s_barrier();                           // 0
            for (int t = 0; t < nt; ++t) {
                reads(t);
                wait_stage(t + 1);
                __builtin_amdgcn_s_barrier();                       // 2t + 1
                __builtin_amdgcn_sched_barrier(0);
                if (!G2_INTERLEAVE && t + 3 < nt) issue(t + 3);
                mfmas(G2_INTERLEAVE && t + 3 < nt ? t + 3 : -1);
                __builtin_amdgcn_s_barrier();                       // 2t + 2
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else
    for (int t = 0; t < nt; ++t) {
        // stage t has landed once at most the stages issued after it are still outstanding (2 + NB DMAs per stage per wave)
        const int ahead = min(nt - 1 - t, STAGES - 2);
        constexpr int D = NA + NB;                                       // DMAs per stage of a wave that takes part in every round (2 D <= 15: the count fits the low four bits)
        if constexpr (NST == 3) {
        if (ahead == 0) __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0)
        else if (a_last) { if (ahead >= 2) __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * D)); else __builtin_amdgcn_s_waitcnt(0x0F70 | D); }
        else { if (ahead >= 2) __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * (D - 1))); else __builtin_amdgcn_s_waitcnt(0x0F70 | (D - 1)); }
        } else {
            // vmcnt(ahead x the wave's DMAs per stage): six bits, the high two in bits 15:14 of the immediate
            static_assert(NA_LAST == NWAVE && (NST - 2) * D <= 63, "every wave issues D DMAs per stage; the count fits vmcnt");
#define G2_VM(n) (0x0F70 | ((n) & 15) | (((n) >> 4) << 14))
            switch (ahead) {
                case 0: __builtin_amdgcn_s_waitcnt(G2_VM(0)); break;
                case 1: __builtin_amdgcn_s_waitcnt(G2_VM(1 * D)); break;
                case 2: __builtin_amdgcn_s_waitcnt(G2_VM(2 * D)); break;
                case 3: __builtin_amdgcn_s_waitcnt(G2_VM(3 * D)); break;
                case 4: __builtin_amdgcn_s_waitcnt(G2_VM(4 * D)); break;
                case 5: __builtin_amdgcn_s_waitcnt(G2_VM(5 * D)); break;
                case 6: __builtin_amdgcn_s_waitcnt(G2_VM(6 * D)); break;
                default: __builtin_amdgcn_s_waitcnt(G2_VM(7 * D)); break;
            }
#undef G2_VM
        }
        __builtin_amdgcn_s_barrier();                               // everyone's part of stage t is in LDS; the slot of stage t-1 is free
        if (t == 0) G2_STAMP(1);
        const bool refill = t + STAGES - 1 < nt;
        if (!(MF && G2_INTERLEAVE) && refill) issue(t + STAGES - 1);
        const char* sb = smem + (t % STAGES) * STAGE;
        frag_ab fa[2][MI], fb[2][2];
        if constexpr (MF) {
            // The refill of the slot freed by the barrier goes out one DMA at a time BETWEEN the MFMA groups, after this stage's
            // fragment reads: a DMA's issue waits for the CU's one address path (1 KB at <= 64 B per clock, shared by every wave of
            // the CU), and issued as a burst after the barrier - every wave of the workgroup in step - nothing computes meanwhile.
            frag_ab fl[SP ? SP : 1][SP ? 2 : 1][SP ? MI : 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j >> 1][j & 1] = *reinterpret_cast<const frag_ab*>(sb + b16[j]);
#pragma unroll
            for (int i = 0; i < 2 * MI; ++i) fa[i & 1][i >> 1] = *reinterpret_cast<const frag_ab*>(sb + a16[i]);
            if constexpr (SP) {
#pragma unroll
                for (int p = 0; p < SP; ++p)
#pragma unroll
                    for (int i = 0; i < 2 * MI; ++i) fl[p][i & 1][i >> 1] = *reinterpret_cast<const frag_ab*>(sb + a16[i] + (p + 1) * BM * G2_BK * 2);
            }
            constexpr int GR = APL * 2 * MI, ND = NA + NB;               // MFMA groups (four products each) and DMAs per stage
#pragma unroll
            for (int g = 0; g < GR; ++g) {
                const int i = g % (2 * MI);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const frag_ab fx = g < 2 * MI ? fa[i & 1][i >> 1] : fl[SP ? g / (2 * MI) - 1 : 0][i & 1][i >> 1];
                    acc4[i][j] = DE ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j >> 1][j & 1], fx, acc4[i][j], 0, 0, 0)
                                    : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx, fb[j >> 1][j & 1], acc4[i][j], 0, 0, 0);
                }
                if constexpr (G2_INTERLEAVE) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (refill) {
#pragma unroll
                        for (int d = 0; d < ND; ++d)
                            if (d * GR / ND == g) issue_one(t + STAGES - 1, d);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            continue;
        }
        auto load_frags = [&](int ks, frag_ab (&a)[MI], frag_ab (&b)[2]) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[ni] = *reinterpret_cast<const frag_ab*>(sb + b_off[ni] + (((2 * ks + kh) ^ b_sw[ni]) << 4));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const frag_ab*>(sb + a_off[mi] + (((2 * ks + kh) ^ a_sw[mi]) << 4));
        };
        load_frags(0, fa[0], fb[0]);
#pragma unroll
        for (int ks = 0; ks < G2_BK / 16; ++ks) {
            if (ks + 1 < G2_BK / 16) load_frags(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][mi], fb[ks & 1][ni], acc[mi][ni], 0, 0, 0);
        }
    }
    ring_epilogue<EPI, BN, 2 * MI, MF, DE, BN == 256>(smem, acc, acc4, e, M, N, m0, n0, wm, wn, wid, lane, LN_OFF);
}

template <int EPI, int BN, int BM = 256, int STAG = 0, int MF = 0, int SP = 0, int NST = 3>
static int gemm_launch_256(const bf16_t* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st) {
    static std::atomic<bool> attr_set[FY_MAX_DEVICES];       // per device: the attribute belongs to the device's copy of the function
    const size_t lds = (size_t)NST * ((SP + 1) * BM + BN) * G2_BK * 2 + ((EPI == 0 || EPI == 1) ? (size_t)BM * 8 : 0);   // (+ the folded LayerNorm's row statistics)  256x256: 96 KB; 320x256: 108 KB (one workgroup per CU); 256x128: 72 KB (two); 128x128: 48 KB (three); split operand: 256x128 120 KB (one), 128x128 72 KB (two)
    const int dev_slot = current_device_slot();
    if (!attr_set[dev_slot].load(std::memory_order_acquire)) {
        HIP_TRY(hipFuncSetAttribute((const void*)gemm256_k<EPI, BN, BM, STAG, MF, SP, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev_slot].store(true, std::memory_order_release);
    }
    dim3 grid((N / BN) * cdiv(M, BM));
    hipLaunchKernelGGL((gemm256_k<EPI, BN, BM, STAG, MF, SP, NST>), grid, dim3(BM == 128 ? 256 : 512), lds, st, A, lda, W, M, N, K, epi);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// ---- the 64-deep, register-staged kernel (round 5) ----------------------------------------------------------------------------
// What bounds gemm256_k at the DiT shapes is its OPERAND FETCH, not the matrix cores (tests/micro/fetch_rows.hip, profiles/
// r05_gemm_fetch_bound.txt): its staging traffic alone - global -> LDS by LDS-DMA, no MFMAs - takes about as long as the whole kernel.
// The LDS-DMA path of a CU saturates near 18-24 bytes per clock (10-14 TB/s chip-wide) whether the lines hit L2 or not, whatever the
// ring depth; plain 16-byte loads to REGISTERS of the same lines run at 17.3 TB/s with one workgroup per CU.  So this kernel
//   (1) stages through registers: a thread loads its 64 bytes of stage t + 2 (global_load_dwordx4 x 4, whole 128-byte lines per
//       8 lanes) while stage t is being multiplied, and writes them to the other LDS buffer (ds_write_b128) one stage later - two
//       stages of lookahead, one of them in registers, where two LDS-DMA stages of this size were all 160 KB hold;
//   (2) uses 64-deep stages of a 256 x 256 tile: a line is requested once (the 32-deep ring asks for its halves in two stages) and
//       ff1 stages 210 MB where two 256 x 128 tiles per CU stage 315;
//   (3) runs SIXTEEN waves (4 x 4, each 64 rows x 64 columns, <= 128 registers) as the one workgroup of its CU: four waves per SIMD
//       hide each other's fragment reads and barrier skew as two workgroups per CU did, without a second copy of the operands.
//   * LDS: two buffers of [BM + BN rows][128 B]; chunk c (16 B) of row r sits at slot c ^ SW(r): conflict-free for the 16x16x32
//     fragment reads (16 rows x one chunk per 16 lanes; the direct epilogue's 4-rows-of-every-16 B pattern) and for the stores
//     (8 lanes = one row's 128 bytes).
//   * K order per accumulator: 32-deep sub-steps in ascending k, exactly gemm256_k's - outputs are bit-identical to its tilings'.
#ifndef G64_SPREAD
#define G64_SPREAD 0          // staging spread over the MFMA groups instead of in one block between the sub-steps: measured 1-5 % slower (ff1 35.1-36.1 against 34.3-34.7 us)
#endif
#ifndef G64_ABL
#define G64_ABL 0          // ablations for tests/micro/gemm_bench (never in the library): 1 no MFMAs, 2 no global loads, 3 no fragment reads
#endif
// NWAVE = 16: 4 x 4 waves of (BM / 4) x 64 (BM = 256: the 128-register form above).  NWAVE = 8: 2 x 4 waves of (BM / 2) x 64 with up to
// 256 registers - for BM = 320 (qkv at M = 6400: 240 tiles = one round where 256-row tiles need a second one), whose 80-row wave
// tiles do not fit 128 registers.
// BN = 128 (the N = 1024 products, out-projection and ff2, at M = 6400: 200 tiles of 256 x 128): 8 x 2 waves of 32 x 64, the gated
// fp32 residual through the LDS epilogue (its 256-byte row runs; the LDS allocation is the epilogue's 136 KB of parked tiles).
template <int EPI, int BM, int BN, int NWAVE>
__global__ __launch_bounds__(NWAVE * 64) void gemm64_k(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int M, int N, int K, GemmEpi e) {
    extern __shared__ __attribute__((aligned(16))) bf16_t gm_smem[];
    char* smem = reinterpret_cast<char*>(gm_smem);
    constexpr int NT = NWAVE * 64;
    static_assert((NWAVE == 16 && BM == 256 && (BN == 256 || BN == 128)) || (NWAVE == 8 && BN == 256 && (BM == 256 || BM == 320)),
                  "gemm64_k: 256x256 (8 or 16 waves), 256x128 (16 waves) and 320x256 (8 waves) tiles");
    constexpr int WN = BN / 64, WM = NWAVE / WN, FM = BM / WM / 16;       // FM 16-row tiles x 4 column tiles per wave
    constexpr bool DE = EPI != 2 && EPI != 3;                             // bf16 outputs: the direct epilogue (transposed products)
    constexpr bool SPREAD = G64_SPREAD && NWAVE == 16;                   // (the 320-row form has no register to spare for it)
    constexpr int STAGE = (BM + BN) * 128, BOFF = BM * 128;
    constexpr int NLA = BM * 8 / NT, NLB = BN * 8 / NT, NL = NLA + NLB;      // 16-byte loads per thread and stage: A rows, W rows
    static_assert(NLA * NT == BM * 8, "whole loads");
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    auto SW = [](int row) -> int { return ((row >> 1) ^ ((row >> 4) << 1)) & 7; };
    const int ntn = N / BN, nwg = gridDim.x;
    const int orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;
    G2_STAMP(0);
    constexpr int LN_OFF = 2 * STAGE;                                   // behind the two buffers (the launcher adds BM x 8 bytes for it)
    // staging: load i of a thread is chunk (tid & 7) of row (tid >> 3) + 128 i of the stage (rows 0 .. BM-1: A, then W).  Sources as a
    // wave-uniform base + one 32-bit byte offset per load (64-bit pointers would not fit the 128 registers), destination offset in LDS
    const char* const Abase = reinterpret_cast<const char*>(A + (long)m0 * lda);
    const char* const Wbase = reinterpret_cast<const char*>(W + (long)n0 * K);
    uint32_t g_off[NL], l_off[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int row = (tid >> 3) + (NT / 8) * i, c = tid & 7;
        if (i < NLA) g_off[i] = (uint32_t)(min(row, M - 1 - m0) * lda + c * 8) * 2u;
        else g_off[i] = (uint32_t)(min(row - BM, N - 1 - n0) * K + c * 8) * 2u;
        l_off[i] = row * 128 + ((c ^ SW(row)) << 4);
    }
    static_assert(NL <= 9, "nine staging registers");
    uint4 rs0, rs1, rs2, rs3, rs4, rs5, rs6, rs7, rs8;                   // (named: as an array, however it was indexed, this ended up in scratch memory)
    // (the empty asm keeps base + offset from being hoisted out of the K loop as 64-bit pairs, which would not fit the registers)
#define G64_L(k, t) if constexpr (NL > k) { uint32_t o_ = g_off[k]; asm volatile("" : "+v"(o_)); rs##k = *reinterpret_cast<const uint4*>((k < NLA ? Abase : Wbase) + (t) * 128 + (size_t)o_); }
#define G64_S(k, t) if constexpr (NL > k) { *reinterpret_cast<uint4*>(smem + ((t) & 1) * STAGE + l_off[k]) = rs##k; }
#define G64_LOAD_STAGE(t) G64_L(0, t) G64_L(1, t) G64_L(2, t) G64_L(3, t) G64_L(4, t) G64_L(5, t) G64_L(6, t) G64_L(7, t) G64_L(8, t)
#define G64_STORE_STAGE(t) G64_S(0, t) G64_S(1, t) G64_S(2, t) G64_S(3, t) G64_S(4, t) G64_S(5, t) G64_S(6, t) G64_S(7, t) G64_S(8, t)
    f32x4 acc4[FM][4];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Fragment read addresses, ONE register per operand (the 128-register cap again).  A fragment i: row ra = wm FM 16 + 16 i + (lane & 15),
    // chunk (lane >> 4) + 4 ks, slot = chunk ^ SW(ra), and SW(ra) = ((lane & 15) >> 1) ^ (((wm FM + i) << 1) & 7): the lane's part and
    // a wave-uniform part, so address = [la: the lane's part] ^ [xa(i) ^ 64 ks: uniform, low 7 bits] + 2048 i.  B fragment j (the direct
    // epilogue's dealing of W rows to the column tiles): row rb = wn 64 + 16 ((lane & 15) >> 2) + 4 j + (lane & 3), SW(rb) = (2 j) ^ [lane's
    // part], address = [lb] ^ (32 j ^ 64 ks) + 512 j.  (All of it modulo a 128-byte aligned LDS base: checked below.)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    if (lds0 & 127) __builtin_trap();
    const int q16 = lane >> 4, l15 = lane & 15;
    const uint32_t la = lds0 + (wm * (FM * 16) + l15) * 128 + ((q16 ^ (l15 >> 1)) << 4);
    // (without the direct epilogue the B fragments are plain 16-row tiles like A's: row wn 64 + 16 j + (lane & 15), address = [lb] ^ (32 j ^ 64 ks) + 2048 j)
    const uint32_t lb = DE ? lds0 + BOFF + (wn * 64 + (l15 >> 2) * 16 + (lane & 3)) * 128 + ((q16 ^ ((lane >> 1) & 1) ^ ((l15 >> 2) << 1)) << 4)
                           : lds0 + BOFF + (wn * 64 + l15) * 128 + ((q16 ^ (l15 >> 1)) << 4);
    auto xa = [&](int i) -> uint32_t { return (uint32_t)((((wm * FM + i) << 1) & 7) << 4); };       // wave-uniform (a constant when FM = 4)
    const int nt = K / 64;
    // one 32-deep sub-step of stage t: the fragment reads are inline asm issued two A fragments ahead of their use (24 fragment
    // registers live instead of 32); reads return in order, so group g waits until at most one read is outstanding
    auto sub_step = [&](int t, int ks) {
        const uint32_t va = la + (t & 1) * STAGE, vb = lb + (t & 1) * STAGE;
        f32x4 rb[4], ra[FM];
#pragma unroll
        for (int j = 0; j < 4; ++j) { if (G64_ABL == 3) rb[j] = f32x4{1.f, 1.f, 1.f, 1.f}; else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb[j]) : "v"(vb ^ (uint32_t)((j * 32) ^ (ks * 64))), "n"(DE ? j * 512 : j * 2048)); }
#pragma unroll
        for (int i = 0; i < 2; ++i) { if (G64_ABL == 3) ra[i] = f32x4{1.f, 1.f, 1.f, 1.f}; else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ra[i]) : "v"(va ^ (xa(i) ^ (uint32_t)(ks * 64))), "n"(i * 2048)); }
#pragma unroll
        for (int g = 0; g < FM; ++g) {
            if (g + 1 < FM) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(ra[g]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(ra[g]));
            const frag_ab fa_g = __builtin_bit_cast(frag_ab, ra[g]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { if (G64_ABL == 1) acc4[g][j][0] += rb[j][0] + ra[g][0]; else acc4[g][j] = DE ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(frag_ab, rb[j]), fa_g, acc4[g][j], 0, 0, 0)
                                                                                                           : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_g, __builtin_bit_cast(frag_ab, rb[j]), acc4[g][j], 0, 0, 0); }
            // (the read two groups ahead goes out BEHIND this group's MFMAs, into the registers fragment g just left: 8 live, not 12)
            if (g + 2 < FM) { if (G64_ABL == 3) ra[g + 2] = f32x4{1.f, 1.f, 1.f, 1.f}; else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ra[g + 2]) : "v"(va ^ (xa(g + 2) ^ (uint32_t)(ks * 64))), "n"((g + 2) * 2048)); }
            // Staging, one register per MFMA group of the FIRST sub-step (G64_SPREAD): register k holds its 16 bytes of stage t + 1
            // (requested a stage ago) - into the other buffer (last read during stage t - 1, which every wave has left: the barrier),
            // and its load of stage t + 2 goes out at once.  Spread over the groups, the 16 waves' stores and loads do not arrive
            // at the LDS and the memory pipe as one burst.
            if (SPREAD && ks == 0) {
                const bool st_ok = t + 1 < nt;
                const int tl = min(t + 2, nt - 1);
                switch (g) {
                    case 0: if (st_ok) { G64_S(0, t + 1) } if (G64_ABL != 2) { G64_L(0, tl) } if (FM == 5 || FM == 10) { if (st_ok) { G64_S(8, t + 1) } if (G64_ABL != 2) { G64_L(8, tl) } } break;
                    case 1: if (st_ok) { G64_S(1, t + 1) } if (G64_ABL != 2) { G64_L(1, tl) } break;
                    case 2: if (st_ok) { G64_S(2, t + 1) } if (G64_ABL != 2) { G64_L(2, tl) } break;
                    case 3: if (st_ok) { G64_S(3, t + 1) } if (G64_ABL != 2) { G64_L(3, tl) } break;
                    case 4: if (st_ok) { G64_S(4, t + 1) } if (G64_ABL != 2) { G64_L(4, tl) } break;
                    case 5: if (st_ok) { G64_S(5, t + 1) } if (G64_ABL != 2) { G64_L(5, tl) } break;
                    case 6: if (st_ok) { G64_S(6, t + 1) } if (G64_ABL != 2) { G64_L(6, tl) } break;
                    case 7: if (st_ok) { G64_S(7, t + 1) } if (G64_ABL != 2) { G64_L(7, tl) } break;
                    default: break;
                }
            }
        }
    };
    { G64_LOAD_STAGE(0) }
    // the folded LayerNorm's row statistics: their loads go out BEHIND the first stage's and land with it (in front of them they were a
    // second memory latency at the head of every workgroup)
    if constexpr (EPI == 0 || EPI == 1) { if (e.ln_rows_slots) ring_ln_rows(smem, LN_OFF, e, M, m0, BM, tid, NT); }
    { G64_STORE_STAGE(0) }
    { G64_LOAD_STAGE(min(1, nt - 1)) }
    __syncthreads();
    G2_STAMP(1);
    for (int t = 0; t < nt; ++t) {
        sub_step(t, 0);
        // the registers hold stage t + 1 (requested a stage ago): into the other buffer - last read during stage t - 1, which every wave
        // has left (the barrier below) - and the loads of stage t + 2 go out at once (unconditional, clamped: no copies at a join)
        if (!SPREAD) {
            if (t + 1 < nt) { G64_STORE_STAGE(t + 1) }
            if (G64_ABL != 2) { G64_LOAD_STAGE(min(t + 2, nt - 1)) }
        }
        sub_step(t, 1);
        __syncthreads();
    }
#undef G64_L
#undef G64_S
#undef G64_LOAD_STAGE
#undef G64_STORE_STAGE
    f32x16 acc_unused[(FM + 1) / 2][2];
    // the epilogue gets the thread's coordinates derived AGAIN from the thread id, behind an empty asm: nothing of the epilogue's
    // address arithmetic can then be live through the K loop (at the 128-register cap every such value was a spill)
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, wid_e = tid_e >> 6;
    ring_epilogue<EPI, BN, FM, 1, DE, NWAVE == 8>(smem, acc_unused, acc4, e, M, N, m0, n0, wid_e / WN, wid_e % WN, wid_e, lane_e, LN_OFF);
}

template <int EPI, int BM, int BN, int NWAVE>
static int gemm_launch_64(const bf16_t* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st) {
    static PerDeviceOnce attr_once;
    // two buffers: 2 x 64 KB (256x256), 2 x 72 KB (320x256), 2 x 48 KB (256x128) - or what the LDS epilogue parks (a 32 x 64 fp32 tile per wave)
    const size_t lds = std::max((size_t)2 * (BM + BN) * 128 + ((EPI == 0 || EPI == 1) ? (size_t)BM * 8 : 0), (EPI == 2 || EPI == 3) ? (size_t)NWAVE * 32 * 68 * 4 : (size_t)0);
    const int dslot = current_device_slot();
    if (!attr_once.done[dslot].load(std::memory_order_acquire)) {
        HIP_TRY(hipFuncSetAttribute((const void*)gemm64_k<EPI, BM, BN, NWAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_once.done[dslot].store(true, std::memory_order_release);
    }
    dim3 grid((N / BN) * cdiv(M, BM));
    hipLaunchKernelGGL((gemm64_k<EPI, BM, BN, NWAVE>), grid, dim3(NWAVE * 64), lds, st, A, lda, W, M, N, K, epi);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

template <int PRECISE, int EPI, int BM>
static int gemm_launch3(const void* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st) {
    static std::atomic<bool> attr_set[FY_MAX_DEVICES];
    const size_t lds = (size_t)2 * ((PRECISE == 4 ? 2 : (PRECISE ? PRECISE : 1)) * BM + GM_BN) * GM_PITCH * sizeof(bf16_t);
    const int dev_slot = current_device_slot();
    if (!attr_set[dev_slot].load(std::memory_order_acquire)) {
        HIP_TRY(hipFuncSetAttribute((const void*)gemm_bf16_k<PRECISE, EPI, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev_slot].store(true, std::memory_order_release);
    }
    dim3 grid(cdiv(N, GM_BN) * cdiv(M, BM));
    hipLaunchKernelGGL((gemm_bf16_k<PRECISE, EPI, BM>), grid, dim3(BM * 2), lds, st, A, lda, W, M, N, K, epi);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// Which kernel: the LDS-DMA ring - 256x128 tiles when they cover most of the chip, 128x128 tiles for smaller grids; the
// register-staged 128x128x64 kernel for N % 128 != 0 and for the split operand of the precise form.
template <int PRECISE, int EPI>
static int gemm_launch2(const void* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st) {
    if (gemm_tile_override == 64) return gemm_launch3<PRECISE, EPI, 64>(A, lda, W, M, N, K, epi, st);
    // Measured on MI355X with tests/micro/gemm_bench (round 2; us, M = 6400 unless noted; tests/micro/gemm_stamps.hip gives the
    // per-workgroup timeline).  What the timeline says: a K step runs at ~80 % of what the MFMAs allow at the clock the chip
    // holds under this load (1.5-1.7 GHz, not 2.4); the rest of a launch is the tile count against the CU count, ~6 us of
    // memory-bound epilogue that every workgroup enters at the same time, and 1-3 us of prologue and launch gap.
    //   tiling (workgroups per CU)         qkv N 3072   out N 1024   ff1 N 2048   ff2 N 1024 K 2048
    //   256x128 (2), 32x32x16 MFMA             58.9         27.9         43.1         43.5
    //   256x128 (2), 16x16x32 MFMA             56.1         27.5         40.7         43.3       the chip holds a higher clock on this shape
    //   128x128 (3), 32x32x16 / 16x16x32    58.7 / 56.1  27.6 / 24.3  51.5 / 48.8  42.3 / 37.0   (M 3200: 18.8 / 19.8 and 29.4 / 31.3)
    //   256x256 (1), plain / staggered      72.9 / 70.1  38.3 / 36.4  45.1 / 42.2  62.6 / 58.8   300 tiles: two rounds
    //   320x256 (1), staggered                 47.5         44.0         49.1         70.1       240 tiles: one round
    // All tilings give bit-identical outputs (K is accumulated in the same order; checked by gemm_bench).
    // the ring kernels' direct epilogue stores 16 consecutive columns per lane in 16-byte pieces
    const bool ring_ok = (((uintptr_t)epi.out | (uintptr_t)epi.out_lo | (uintptr_t)epi.resid) & 15) == 0 && epi.ldc % 8 == 0 &&
                         (!epi.rope || (epi.rope_half % 8 == 0 && epi.rope_stride % 16 == 0));
    if (!PRECISE && gemm_tile_override != 128 && N % 128 == 0 && K % G2_BK == 0 && ring_ok) {
        static int cus = 0;
        if (!cus) {
            int dev = 0;
            hipDeviceProp_t p;
            cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
        }
        const bf16_t* Ab = (const bf16_t*)A;
        static const int env_ov = getenv("FY_GEMM_TILE") ? atoi(getenv("FY_GEMM_TILE")) : 0;   // experiments: a tiling for every product
        const int ov = gemm_tile_override ? gemm_tile_override : env_ov;
        if (ov == 256 && N % 256 == 0) return gemm_launch_256<EPI, 256>(Ab, lda, W, M, N, K, epi, st);
        if (ov == 320 && N % 256 == 0) return gemm_launch_256<EPI, 256, 320>(Ab, lda, W, M, N, K, epi, st);
        if (ov == 1256 && N % 256 == 0) return gemm_launch_256<EPI, 256, 256, 1>(Ab, lda, W, M, N, K, epi, st);
        if (ov == 3320 && N % 256 == 0) return gemm_launch_256<EPI, 256, 320, 1, 1>(Ab, lda, W, M, N, K, epi, st);
        if (ov == 3256 && N % 256 == 0) return gemm_launch_256<EPI, 256, 256, 1, 1>(Ab, lda, W, M, N, K, epi, st);
        if (ov == 2256 && N % 256 == 0) return gemm_launch_256<EPI, 256, 256, 0, 1>(Ab, lda, W, M, N, K, epi, st);
        if (ov == 2) return gemm_launch_256<EPI, 128>(Ab, lda, W, M, N, K, epi, st);
        if (ov == 3) return gemm_launch_256<EPI, 128, 128>(Ab, lda, W, M, N, K, epi, st);
        const int t128 = (N / 128) * cdiv(M, 256), t64 = (N / 128) * cdiv(M, 128), t320 = N % 256 ? 0 : (N / 256) * cdiv(M, 320);
        // Round 5: the 64-deep ring with ONE 256x256 or 320x256 tile of sixteen waves per CU for the bf16-output products whose tiles
        // cover the chip in one round (ff1 at M = 6400: 200 tiles; qkv: 240) - gemm64_k's comment says why.  FY_GEMM64=0: off.
        static const int g64 = getenv("FY_GEMM64") ? atoi(getenv("FY_GEMM64")) : 1;
        if constexpr (EPI == 0 || EPI == 1) {
            if ((ov == 6400 || ov == 6408 || ov == 6432 || (ov == 0 && g64)) && N % 256 == 0 && K % 64 == 0) {
                const int t256 = (N / 256) * cdiv(M, 256);
                // (the rotary epilogue's 16-wave instantiation does not fit 128 registers without spilling inside the K loop - and a spill of
                // a register that an inline-asm ds_read is still filling stores garbage - so the rotary products take the 8-wave form)
                if (ov == 6400 || (ov == 0 && t256 <= cus && 20 * t256 >= 15 * cus)) return gemm_launch_64<EPI, 256, 256, EPI == 1 ? 16 : 8>(Ab, lda, W, M, N, K, epi, st);
                if (ov == 6408) return gemm_launch_64<EPI, 256, 256, 8>(Ab, lda, W, M, N, K, epi, st);
                if (ov == 6432 || (ov == 0 && t320 <= cus && 20 * t320 >= 15 * cus)) return gemm_launch_64<EPI, 320, 256, 8>(Ab, lda, W, M, N, K, epi, st);
            }
        }
        if constexpr (EPI == 3) {
            // the gated fp32 residual products (out-projection, ff2) as 256 x 128 tiles of 16 waves, one per CU: measured, NOT chosen
            // (override 6412 only).  Its K loop takes what the 128 x 128 LDS-DMA tiles' takes (15.6 against 15.4 us for the
            // out-projection, 31.9 against 30.6 for ff2): sixteen 32 x 64 wave tiles read 0.75 fragments per MFMA and, with the staging
            // stores, the LDS array is busier than the matrix cores - and one tile per CU has nothing to run under its epilogue.
            if (ov == 6412 && K % 64 == 0) return gemm_launch_64<EPI, 256, 128, 16>(Ab, lda, W, M, N, K, epi, st);
        }
        // 320x256 tiles, one workgroup per CU, when they cover the chip in ONE round where 256x128 tiles would need a second one
        if (ov == 1320 && N % 256 == 0) return gemm_launch_256<EPI, 256, 320, 1>(Ab, lda, W, M, N, K, epi, st);
        if (ov == 0 && t128 > 2 * cus && t320 <= cus && 20 * t320 >= 17 * cus) return gemm_launch_256<EPI, 256, 320, 1, 1>(Ab, lda, W, M, N, K, epi, st);
        // 256x128 tiles (two workgroups per CU: one's prologue / epilogue bursts run under the other's K loop) from one tile per CU on
        if (ov == 2002 || (ov == 0 && t128 >= cus)) return gemm_launch_256<EPI, 128, 256, 0, 1>(Ab, lda, W, M, N, K, epi, st);
        // fewer tiles than that: 128x128 tiles (4 waves, 48 KB, three workgroups per CU); the 16x16x32 form reads a K step's
        // fragments in one go, which costs a workgroup that is alone on its CU more than the clock gains
        if (ov == 2003 || (ov == 0 && 2 * t64 >= 3 * cus)) return gemm_launch_256<EPI, 128, 128, 0, 1>(Ab, lda, W, M, N, K, epi, st);
        // fewer tiles than CUs (stream=True chunks): one workgroup per CU at most - an eight-stage ring (gemm256_k's comment).  FY_GEMM_DEEP=0: off
        static const int deep = getenv("FY_GEMM_DEEP") ? atoi(getenv("FY_GEMM_DEEP")) : 1;
        if (ov == 8128 || (ov == 0 && deep && t64 <= cus && K >= 8 * G2_BK)) return gemm_launch_256<EPI, 128, 128, 0, 0, 0, 8>(Ab, lda, W, M, N, K, epi, st);
        return gemm_launch_256<EPI, 128, 128>(Ab, lda, W, M, N, K, epi, st);
    }
    return gemm_launch3<PRECISE, EPI, 128>(A, lda, W, M, N, K, epi, st);
}

template <int PRECISE>
static int gemm_launch(const void* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st) {
    if (epi.mode == EPI_GATE_RESID) return gemm_launch2<PRECISE, 3>(A, lda, W, M, N, K, epi, st);
    if (!epi.out_bf16) {
        FY_CHECK(epi.act == ACT_NONE || epi.act == ACT_GELU_TANH, FY_ERR_ARG, "gemm: only GELU(tanh) is fused");
        return gemm_launch2<PRECISE, 2>(A, lda, W, M, N, K, epi, st);
    }
    if (epi.act == ACT_GELU_TANH) return gemm_launch2<PRECISE, 1>(A, lda, W, M, N, K, epi, st);
    FY_CHECK(epi.act == ACT_NONE, FY_ERR_ARG, "gemm: only GELU(tanh) is fused");
    return gemm_launch2<PRECISE, 0>(A, lda, W, M, N, K, epi, st);
}

static int gemm_check(const void* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& e, int a_elem) {
    FY_CHECK(A && W && M >= 1 && N >= 1 && K >= GM_BK && K % GM_BK == 0, FY_ERR_ARG, "gemm: bad shape M %d N %d K %d", M, N, K);
    FY_CHECK(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && (lda * a_elem) % 16 == 0, FY_ERR_ARG, "gemm: operands must be 16-B aligned");
    FY_CHECK(e.ldc >= N && ((e.mode == EPI_STORE && e.out) || (e.mode == EPI_GATE_RESID && e.resid && e.gate)), FY_ERR_ARG, "gemm: bad epilogue");
    FY_CHECK(N % 4 == 0 && e.ldc % 4 == 0, FY_ERR_ARG, "gemm: N and the output pitch must be multiples of 4 (N %d, ldc %d)", N, e.ldc);
    FY_CHECK(!e.rope || (e.mode == EPI_STORE && e.act == ACT_NONE && e.rope_T >= 1 && e.rope_div >= 1 && e.rope_half >= 2 && e.rope_half % 2 == 0 &&
                         e.rope_stride >= 2 * e.rope_half), FY_ERR_ARG, "gemm: bad rotary epilogue");
    return FY_OK;
}

int gemm_bf16(const bf16_t* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st) {
    FY_TRY(gemm_check(A, lda, W, M, N, K, epi, 2));
    if (epi.h_bf16 || epi.ln_rows_slots) {
        // the folded LayerNorm-modulate lives in the ring kernels' epilogues only
        FY_CHECK(N % 128 == 0 && K % G2_BK == 0 && gemm_tile_override != 128 && gemm_tile_override != 64, FY_ERR_ARG, "gemm: the folded LayerNorm needs a ring-kernel shape (N %d, K %d)", N, K);
        FY_CHECK(!epi.h_bf16 || (epi.mode == EPI_GATE_RESID && epi.ln_slots && ((uintptr_t)epi.h_bf16 & 15) == 0), FY_ERR_ARG, "gemm: h_bf16 goes with the gated residual and ln_slots");
        FY_CHECK(!epi.ln_rows_slots || (epi.mode == EPI_STORE && epi.out_bf16 && epi.ln_u && !epi.out_lo && epi.ln_dim == K && K % 64 == 0), FY_ERR_ARG, "gemm: ln_rows_slots goes with a bf16 output, ln_u and ln_dim = K");
    }
    ProfScope prof("gemm_bf16", 2.0 * M * N * K, st);
    return gemm_launch<0>(A, lda, W, M, N, K, epi, st);
}

int gemm_f32a_precise(const float* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st) {
    FY_TRY(gemm_check(A, lda, W, M, N, K, epi, 4));
    return gemm_launch<2>(A, lda, W, M, N, K, epi, st);
}

// x = hi + lo as two bf16 planes (gemm.h): the ring kernel with two A tiles per stage; the register-staged kernel for N % 128 != 0
template <int EPI>
static int gemm_split_launch(const bf16_t* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st) {
    if (N % 128 == 0 && K % G2_BK == 0) {
        static const int tile = getenv("FY_GEMM_SPLIT_TILE") ? atoi(getenv("FY_GEMM_SPLIT_TILE")) : 0;   // experiments: 256 = 256x128 tiles, one workgroup per CU
        if (tile == 256) return gemm_launch_256<EPI, 128, 256, 0, 1, 1>(A, lda, W, M, N, K, epi, st);
        return gemm_launch_256<EPI, 128, 128, 0, 1, 1>(A, lda, W, M, N, K, epi, st);
    }
    if constexpr (EPI == 2 || EPI == 3) {
        // the register-staged kernel knows one weight plane: the lo plane is a second launch that adds into the first one's output
        // (both epilogues are linear in the product), through the gated residual form - epi.gate_ones: a vector of N ones
        FY_TRY((gemm_launch3<4, EPI, 128>(A, lda, W, M, N, K, epi, st)));
        if (!epi.w_lo) return FY_OK;
        FY_CHECK(epi.gate_ones, FY_ERR_ARG, "gemm_split: a second weight plane on the register-staged kernel needs epi.gate_ones");
        GemmEpi e2;
        e2.mode = EPI_GATE_RESID; e2.gate = epi.gate_ones; e2.ldc = epi.ldc; e2.a_lo = epi.a_lo;
        e2.resid = EPI == 3 ? epi.resid : (float*)epi.out;
        return gemm_launch3<4, 3, 128>(A, lda, epi.w_lo, M, N, K, e2, st);
    }
    FY_CHECK(false, FY_ERR_ARG, "gemm_split: a split-plane output needs N %% 128 == 0 and K %% 32 == 0 (N %d, K %d)", N, K);
}

int gemm_split(const bf16_t* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st) {
    FY_TRY(gemm_check(A, lda, W, M, N, K, epi, 2));
    FY_CHECK(epi.a_lo && ((uintptr_t)epi.a_lo & 15) == 0, FY_ERR_ARG, "gemm_split: the lo plane of A is missing or misaligned");
    FY_CHECK(!epi.w_lo || ((uintptr_t)epi.w_lo & 15) == 0, FY_ERR_ARG, "gemm_split: the lo plane of W is misaligned");
    ProfScope prof("gemm_split", (epi.w_lo ? 8.0 : 4.0) * M * N * K, st);
    if (epi.mode == EPI_GATE_RESID) return gemm_split_launch<3>(A, lda, W, M, N, K, epi, st);
    if (epi.out_lo) {
        FY_CHECK(epi.act == ACT_NONE || (epi.act == ACT_GELU_TANH && !epi.rope), FY_ERR_ARG, "gemm_split: only GELU(tanh) or the rotary embedding is fused");
        return epi.act == ACT_GELU_TANH ? gemm_split_launch<5>(A, lda, W, M, N, K, epi, st) : gemm_split_launch<4>(A, lda, W, M, N, K, epi, st);
    }
    FY_CHECK(!epi.out_bf16 && epi.act == ACT_NONE && !epi.rope, FY_ERR_ARG, "gemm_split: outputs are fp32, the gated residual, or split bf16 planes");
    return gemm_split_launch<2>(A, lda, W, M, N, K, epi, st);
}

bool gemm_exact3_supported(int N, int K) { return N % 128 == 0 && K % G2_BK == 0; }

int gemm_exact3(const bf16_t* A_hi, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st) {
    FY_TRY(gemm_check(A_hi, lda, W, M, N, K, epi, 2));
    FY_CHECK(gemm_exact3_supported(N, K), FY_ERR_ARG, "gemm_exact3: N %d must be a multiple of 128 and K %d of 32", N, K);
    FY_CHECK(epi.a_lo && epi.a_lo2 && (((uintptr_t)epi.a_lo | (uintptr_t)epi.a_lo2) & 15) == 0, FY_ERR_ARG, "gemm_exact3: a plane of A is missing or misaligned");
    FY_CHECK((epi.mode == EPI_GATE_RESID || !epi.out_bf16) && !epi.out_lo && epi.act == ACT_NONE && !epi.rope, FY_ERR_ARG, "gemm_exact3: outputs are fp32 or the gated residual");
    FY_CHECK((((uintptr_t)epi.out | (uintptr_t)epi.resid) & 15) == 0 && epi.ldc % 4 == 0, FY_ERR_ARG, "gemm_exact3: misaligned output");
    ProfScope prof("gemm_exact", 2.0 * M * N * K, st);
    // 128x128 tiles, 96 KB of ring (one workgroup per CU): the LM prefill's products have 49-532 tiles
    if (epi.mode == EPI_GATE_RESID) return gemm_launch_256<3, 128, 128, 0, 1, 2>(A_hi, lda, W, M, N, K, epi, st);
    return gemm_launch_256<2, 128, 128, 0, 1, 2>(A_hi, lda, W, M, N, K, epi, st);
}

__global__ void split3_planes_k(const float* __restrict__ src, int ld_src, int rows, int cols, bf16_t* __restrict__ hi, bf16_t* __restrict__ mid, bf16_t* __restrict__ lo) {
    const long n4 = (long)rows * (cols / 4);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long r = i / (cols / 4);
        const int c = (int)(i % (cols / 4)) * 4;
        const float4 x = *reinterpret_cast<const float4*>(src + r * ld_src + c);
        const float f[4] = {x.x, x.y, x.z, x.w};
        uint32_t h[4], m[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                   // split8_3's arithmetic
            const __bf16 hb = (__bf16)f[j];
            const float r1 = f[j] - (float)hb;
            const __bf16 mb = (__bf16)r1;
            const float r2 = r1 - (float)mb;
            const __bf16 lb = (__bf16)r2;
            h[j] = __builtin_bit_cast(unsigned short, hb); m[j] = __builtin_bit_cast(unsigned short, mb); l[j] = __builtin_bit_cast(unsigned short, lb);
        }
        const long o = r * cols + c;
        *reinterpret_cast<uint2*>(hi + o) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
        *reinterpret_cast<uint2*>(mid + o) = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
        *reinterpret_cast<uint2*>(lo + o) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
    }
}
int split3_planes(const float* src, int ld_src, int rows, int cols, bf16_t* hi, bf16_t* mid, bf16_t* lo, hipStream_t st) {
    FY_CHECK(src && hi && mid && lo && rows >= 1 && cols >= 4 && cols % 4 == 0 && ld_src % 4 == 0, FY_ERR_ARG, "split3_planes: bad arguments");
    const long n4 = (long)rows * (cols / 4);
    hipLaunchKernelGGL(split3_planes_k, dim3((unsigned)std::min<long>(4096, (n4 + 255) / 256)), dim3(256), 0, st, src, ld_src, rows, cols, hi, mid, lo);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// test hook (tests/test_flow_gpu.py): out [M][N] bf16 = act(A [M][K] bf16 x W [N][K] bf16 + bias) with one tiling forced for every shape
// (gemm_launch2's override codes; 0 = the automatic choice) - every tiling and both epilogues must give the same bits
extern "C" int fy_debug_gemm_bf16(const void* A, const void* W, int32_t M, int32_t N, int32_t K, const float* bias, void* out, int32_t gelu,
                                  int32_t tile, const float* rope, int32_t rope_T, void* stream) {
    GemmEpi e;
    e.bias = bias; e.out = out; e.out_bf16 = 1; e.ldc = N; e.act = gelu ? ACT_GELU_TANH : ACT_NONE;
    if (rope) { e.rope = reinterpret_cast<const float2*>(rope); e.rope_T = rope_T; e.rope_half = 32; e.rope_stride = N / 3; }   // the DiT's qkv: head 0 of q and of k
    const int was = gemm_tile_override;
    gemm_tile_override = tile;
    const int rc = gemm_bf16((const bf16_t*)A, K, (const bf16_t*)W, M, N, K, e, (hipStream_t)stream);
    gemm_tile_override = was;
    return rc;
}

// test hook (tests/test_llm_gpu.py): out [M][N] fp32 = A [M][K] fp32 x W [N][K] bf16 (+ bias) through the exact three-way split, on the
// ring kernel (ring != 0; planes = scratch of 3 M K bf16) or on the register-staged kernel
extern "C" int fy_debug_gemm_exact(const float* A, const void* W, int32_t M, int32_t N, int32_t K, const float* bias, float* out, int32_t ring,
                                   void* planes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    GemmEpi e;
    e.bias = bias; e.out = out; e.out_bf16 = 0; e.ldc = N;
    if (!ring) return gemm_f32a_exact(A, K, (const bf16_t*)W, M, N, K, e, st);
    FY_CHECK(planes, FY_ERR_ARG, "fy_debug_gemm_exact: the ring form needs the planes scratch");
    bf16_t *p0 = (bf16_t*)planes, *p1 = p0 + (size_t)M * K, *p2 = p1 + (size_t)M * K;
    FY_TRY(split3_planes(A, K, M, K, p0, p1, p2, st));
    e.a_lo = p1; e.a_lo2 = p2;
    return gemm_exact3(p0, K, (const bf16_t*)W, M, N, K, e, st);
}

int gemm_f32a_exact(const float* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st) {
    FY_TRY(gemm_check(A, lda, W, M, N, K, epi, 4));
    ProfScope prof("gemm_exact", 2.0 * M * N * K, st);
    return gemm_launch<3>(A, lda, W, M, N, K, epi, st);
}

__global__ void cast_f32_bf16_k(const float* __restrict__ s, bf16_t* __restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * 256UL + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = f32_to_bf16(s[i]);
}
int cast_f32_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t st) {
    size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(cast_f32_bf16_k, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, st, src, dst, n);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// =============================================================================
// decode GEMV on the matrix cores: y[r][n] = sum_k W[n][k] x[r][k] for up to 8 fp32 activation
// rows per z-slice, at fp32 fidelity.
//   * each activation is split exactly into three bf16 terms x = hi + mid + lo (3 x 8 mantissa bits);
//     the 8 rows become 24 rows of the A operand of v_mfma_f32_32x32x16_bf16 (rows 0-7 hi, 8-15 mid,
//     16-23 lo, 24-31 zero) and the three partial products are summed in the epilogue.  With
//     bf16-representable weights every product is exact and the accumulation is fp32, so the result
//     tracks an fp32 reference - which is what keeps greedy token ids bit-exact;
//   * weights are pre-packed in B-fragment order [N/32][K/16][64 lanes][8], so a wave streams 1 KiB
//     contiguous per MFMA with non-temporal loads and eight fragments in flight;
//   * a block owns one 32-column tile; its waves split K and combine through LDS in a fixed order
//     (deterministic: no atomics);  an optional RMSNorm is folded in (activations times the norm
//     weight before the split, 1/rms on the finished sum).
// =============================================================================
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
#define GV_SLICE 1024                      // K staged per pass
#define GV_PITCH (GV_SLICE + 8)            // bf16 elements per LDS row
#define GV_BATCH 8                         // weight fragments in flight per wave

__device__ __forceinline__ frag_ab ld_frag_nt(const bf16_t* p) {
    u32x4_t r = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return __builtin_bit_cast(frag_ab, r);
}

// split one fp32 value exactly into three bf16 terms (hardware round-to-nearest-even converts)
__device__ __forceinline__ void split3(float f, uint32_t& h, uint32_t& m, uint32_t& l) {
    __bf16 hb = (__bf16)f;
    float r1 = f - (float)hb;
    __bf16 mb = (__bf16)r1;
    float r2 = r1 - (float)mb;
    __bf16 lb = (__bf16)r2;
    h = __builtin_bit_cast(unsigned short, hb);
    m = __builtin_bit_cast(unsigned short, mb);
    l = __builtin_bit_cast(unsigned short, lb);
}

// shared epilogue of the two GEMV kernels (wave 0 of the block): p[i] is the finished sum for
// activation row r = 4*kh + i, output column n.  yres[i]: the residual prefetched at kernel start.
__device__ __forceinline__ void gv_epilogue(const GemvArgs& a, const float (&p)[4], const float (&yres)[4], const float* rstd_s,
                                            int n, int kh, int lr, int R) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * kh + i;
        float tot = p[i];
        if (rstd_s) tot *= rstd_s[r];
        if (a.bias && n < a.N) tot += a.bias[n];
        const long rr = (long)blockIdx.z * 8 + r;
        if (a.mode == GV_SWIGLU || a.mode == GV_SWIGLU_SPLIT) {
            float other = __shfl_xor(tot, 1, 64);          // rows interleaved (gate_i, up_i)
            if ((lr & 1) == 0 && r < R && n < a.N) {
                float v = act_silu(tot) * other;
                if (a.mode == GV_SWIGLU) a.y[rr * a.ldy + (n >> 1)] = v;
                else {                                      // three bf16 planes for the direct-A consumer
                    uint32_t h, m, l;
                    split3(v, h, m, l);
                    bf16_t* o = a.y_split + (long)blockIdx.z * 24 * a.ldy + (n >> 1);
                    o[(long)r * a.ldy] = (bf16_t)h;
                    o[(long)(8 + r) * a.ldy] = (bf16_t)m;
                    o[(long)(16 + r) * a.ldy] = (bf16_t)l;
                }
            }
        } else if (r < R && n < a.N) {
            if (a.mode == GV_ADD) a.y[rr * a.ldy + n] = yres[i] + tot; else a.y[rr * a.ldy + n] = tot;
        }
    }
}

// (A) K <= 1024: activations staged (and split) through LDS; 4 waves split K.
__global__ __launch_bounds__(256) void gemv_lds_k(GemvArgs a) {
    constexpr int NW = 4, MAXF = GV_SLICE / 16 / NW;                        // <= 16 fragments per wave
    extern __shared__ __attribute__((aligned(16))) char gv_smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(gv_smem);                       // [24][GV_PITCH]
    float* red = reinterpret_cast<float*>(gv_smem + 24 * GV_PITCH * 2);     // [NW][64][4]
    float* rstd_s = red + NW * 256;                                         // [8]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 31, kh = lane >> 5;
    const int n32 = blockIdx.x, K16 = a.K / 16, n = n32 * 32 + lr;
    const float* xg = a.x + (long)blockIdx.z * 8 * a.ldx;
    const int R = min(8, a.R - blockIdx.z * 8);
    const bf16_t* wt = a.W + (long)n32 * K16 * 512 + lane * 8;
    // everything that does not depend on the activations is requested first: this wave's weight
    // fragments (kk = wid, wid+4, ...) and, for y += Wx, the residual
    frag_ab b[MAXF];
#pragma unroll
    for (int u = 0; u < MAXF; ++u) {
        const int kk = wid + u * NW;
        if (kk < K16) b[u] = ld_frag_nt(wt + (long)kk * 512);
    }
    float yres[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.mode == GV_ADD && wid == 0 && n < a.N) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * kh + i < R) yres[i] = a.y[((long)blockIdx.z * 8 + 4 * kh + i) * a.ldy + n];
    }
    {   // stage + split: 32 threads per activation row, each a fixed set of float4 columns (deterministic sums)
        const int r = tid >> 5, q = tid & 31;
        float4 v[GV_SLICE / 128];
#pragma unroll
        for (int i = 0; i < GV_SLICE / 128; ++i) {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i * 128 < a.K && r < R) v[i] = *reinterpret_cast<const float4*>(xg + (long)r * a.ldx + (q + 32 * i) * 4);
        }
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < GV_SLICE / 128; ++i) {
            if (i * 128 < a.K) {
                const int c4 = (q + 32 * i) * 4;
                float4 x = v[i];
                if (a.norm_w) {
                    float4 w4 = *reinterpret_cast<const float4*>(a.norm_w + c4);
                    ss += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
                    x.x *= w4.x; x.y *= w4.y; x.z *= w4.z; x.w *= w4.w;
                }
                uint32_t h[4], m[4], l[4];
                split3(x.x, h[0], m[0], l[0]); split3(x.y, h[1], m[1], l[1]);
                split3(x.z, h[2], m[2], l[2]); split3(x.w, h[3], m[3], l[3]);
                *reinterpret_cast<uint2*>(xs + r * GV_PITCH + c4) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
                *reinterpret_cast<uint2*>(xs + (8 + r) * GV_PITCH + c4) = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
                *reinterpret_cast<uint2*>(xs + (16 + r) * GV_PITCH + c4) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
            }
        }
        if (a.norm_w) {
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
            if (q == 0) rstd_s[r] = rsqrtf(ss / a.K + a.eps);
        }
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    frag_ab zero_frag;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero_frag[e] = (__bf16)0.f;
#pragma unroll
    for (int u = 0; u < MAXF; ++u) {
        const int kk = wid + u * NW;
        if (kk < K16) {
            frag_ab af = zero_frag;
            if (lr < 24) af = *reinterpret_cast<const frag_ab*>(xs + lr * GV_PITCH + kk * 16 + kh * 8);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b[u], acc, 0, 0, 0);
        }
    }
    // D rows: reg i -> row (i&3) + 8*(i>>2) + 4*kh.  hi rows 0-7, mid 8-15, lo 16-23: activation row r = 4*kh + i
    float p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = (acc[i] + acc[4 + i]) + acc[8 + i];
    *reinterpret_cast<float4*>(red + (wid * 64 + lane) * 4) = make_float4(p[0], p[1], p[2], p[3]);
    __syncthreads();
    if (wid != 0) return;
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        float4 q = *reinterpret_cast<const float4*>(red + (w * 64 + lane) * 4);
        p[0] += q.x; p[1] += q.y; p[2] += q.z; p[3] += q.w;
    }
    gv_epilogue(a, p, yres, a.norm_w ? rstd_s : nullptr, n, kh, lr, R);
}

// (B) any K: the A operand comes pre-split from global memory (x_split: bf16 [z][24][lda], written by a
// GV_SWIGLU_SPLIT epilogue); no LDS staging, no slices.  K is split over gridDim.y blocks (more CUs
// streaming the same few output tiles) and over the 16 waves of a block; the K-slice blocks of a tile
// hand their partial sums over in HBM and the last one to arrive adds them in slice order, so the
// result does not depend on arrival order (agent-scope release / acquire around an arrival ticket).
__global__ __launch_bounds__(1024) void gemv_direct_k(GemvArgs a) {
    constexpr int NW = 16;
    __shared__ __attribute__((aligned(16))) float red[NW * 256];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 31, kh = lane >> 5;
    const int n32 = blockIdx.x, K16 = a.K / 16, n = n32 * 32 + lr;
    const int KS = gridDim.y, ks = blockIdx.y;
    const int kper = (K16 + KS - 1) / KS, kbeg = ks * kper, kend = min(K16, kbeg + kper);
    const int R = min(8, a.R - blockIdx.z * 8);
    const bf16_t* wt = a.W + (long)n32 * K16 * 512 + lane * 8;
    const bf16_t* xa = a.x_split + ((long)blockIdx.z * 24 + min(lr, 23)) * a.ldx + kh * 8;
    float yres[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.mode == GV_ADD && wid == 0 && n < a.N) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * kh + i < R) yres[i] = a.y[((long)blockIdx.z * 8 + 4 * kh + i) * a.ldy + n];
    }
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    frag_ab zero_frag;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero_frag[e] = (__bf16)0.f;
    constexpr int DB = 8;                                  // fragments in flight per wave
    for (int kb = kbeg + wid; kb < kend; kb += NW * DB) {
        frag_ab b[DB], af[DB];
#pragma unroll
        for (int u = 0; u < DB; ++u) {
            const int kk = kb + u * NW;
            if (kk < kend) {
                b[u] = ld_frag_nt(wt + (long)kk * 512);
                af[u] = *reinterpret_cast<const frag_ab*>(xa + kk * 16);
            }
        }
#pragma unroll
        for (int u = 0; u < DB; ++u) {
            const int kk = kb + u * NW;
            if (kk < kend) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lr < 24 ? af[u] : zero_frag, b[u], acc, 0, 0, 0);
        }
    }
    float p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = (acc[i] + acc[4 + i]) + acc[8 + i];
    *reinterpret_cast<float4*>(red + (wid * 64 + lane) * 4) = make_float4(p[0], p[1], p[2], p[3]);
    __syncthreads();
    if (wid != 0) return;
#pragma unroll
    for (int w = 1; w < NW; ++w) {
        float4 q = *reinterpret_cast<const float4*>(red + (w * 64 + lane) * 4);
        p[0] += q.x; p[1] += q.y; p[2] += q.z; p[3] += q.w;
    }
    if (KS > 1) {
        const long tile = (long)blockIdx.z * gridDim.x + n32;
        float* slab = a.partial + (tile * KS) * 256;
        *reinterpret_cast<float4*>(slab + (long)ks * 256 + lane * 4) = make_float4(p[0], p[1], p[2], p[3]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int ticket = 0;
        if (lane == 0) ticket = __hip_atomic_fetch_add(a.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ticket = __shfl(ticket, 0, 64);
        if (ticket != KS - 1) return;                       // not the last slice of this tile
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) a.counters[tile] = 0;                // ready for the next launch
        p[0] = p[1] = p[2] = p[3] = 0.f;
        for (int s2 = 0; s2 < KS; ++s2) {                   // fixed order: independent of who arrived when
            float4 q = *reinterpret_cast<const float4*>(slab + (long)s2 * 256 + lane * 4);
            p[0] += q.x; p[1] += q.y; p[2] += q.z; p[3] += q.w;
        }
    }
    gv_epilogue(a, p, yres, nullptr, n, kh, lr, R);
}

// pack fp32 [N][K] (row-major, torch Linear layout) into B-fragment order, zero-padding N to 32
__global__ void gemv_pack_k(const float* __restrict__ src, bf16_t* __restrict__ dst, int N, int K) {
    const int K16 = K / 16;
    const long total = (long)((N + 31) / 32) * K16 * 512;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int j = i & 7, l = (i >> 3) & 63;
        const long f = i >> 9;
        const int k16 = (int)(f % K16), n32 = (int)(f / K16);
        const int n = n32 * 32 + (l & 31), k = k16 * 16 + 8 * (l >> 5) + j;
        dst[i] = n < N ? f32_to_bf16(src[(long)n * K + k]) : (bf16_t)0;
    }
}

size_t gemv_packed_elems(int N, int K) { return (size_t)((N + 31) / 32) * (K / 16) * 512; }

int gemv_pack(const float* src, bf16_t* dst, int N, int K, hipStream_t st) {
    FY_CHECK(src && dst && N >= 1 && K >= 16 && K % 16 == 0, FY_ERR_ARG, "gemv_pack: bad shape N %d K %d", N, K);
    size_t blocks = (gemv_packed_elems(N, K) + 255) / 256;
    hipLaunchKernelGGL(gemv_pack_k, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, st, src, dst, N, K);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// split-K workspace of the direct form: [tiles][4 slices][64 lanes][4] floats and one arrival counter per tile
size_t gemv_partial_floats(int R, int N, int K) { return K >= 2048 ? (size_t)cdiv(R, 8) * cdiv(N, 32) * 4 * 256 : 0; }
size_t gemv_counter_ints(int R, int N, int K) { return K >= 2048 ? (size_t)cdiv(R, 8) * cdiv(N, 32) : 0; }

int gemv_bf16w(const GemvArgs& a, hipStream_t st) {
    FY_CHECK(a.W && (a.x || a.x_split) && (a.y || a.y_split) && a.R >= 1 && a.N >= 1 && a.K >= 16 && a.K % 16 == 0, FY_ERR_ARG,
             "gemv: bad arguments R %d N %d K %d", a.R, a.N, a.K);
    FY_CHECK(a.mode != GV_SWIGLU_SPLIT || a.y_split, FY_ERR_ARG, "gemv: GV_SWIGLU_SPLIT needs y_split");
    FY_CHECK((a.mode != GV_SWIGLU && a.mode != GV_SWIGLU_SPLIT) || a.N % 2 == 0, FY_ERR_ARG, "gemv: SwiGLU rows must come in interleaved pairs");
    FY_CHECK(((uintptr_t)a.W & 15) == 0, FY_ERR_ARG, "gemv: weights must be 16-B aligned");
    dim3 grid(cdiv(a.N, 32), 1, cdiv(a.R, 8));
    ProfScope prof("gemv", 2.0 * a.N * a.K, st);                 // work = the product's bf16 weight bytes (= flops per row)
    if (a.x_split) {
        FY_CHECK(!a.norm_w && a.ldx % 8 == 0 && ((uintptr_t)a.x_split & 15) == 0, FY_ERR_ARG, "gemv: bad pre-split operand");
        int KS = (a.K >= 2048 && a.partial && a.counters) ? 4 : 1;
        grid.y = KS;
        hipLaunchKernelGGL(gemv_direct_k, grid, dim3(1024), 0, st, a);
    } else {
        FY_CHECK(a.K <= GV_SLICE && a.K % 128 == 0 && a.ldx % 4 == 0 && ((uintptr_t)a.x & 15) == 0, FY_ERR_ARG,
                 "gemv: the LDS-staged form needs K <= %d, K %% 128 == 0 (K = %d)", GV_SLICE, a.K);
        static PerDeviceOnce attr_once;
        const int dslot = current_device_slot();
        const size_t lds = (size_t)24 * GV_PITCH * 2 + 4 * 256 * 4 + 8 * 4;
        if (!attr_once.done[dslot].load(std::memory_order_acquire)) {
            HIP_TRY(hipFuncSetAttribute((const void*)gemv_lds_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_once.done[dslot].store(true, std::memory_order_release);
        }
        hipLaunchKernelGGL(gemv_lds_k, grid, dim3(256), lds, st, a);
    }
    HIP_TRY(hipGetLastError());
    return FY_OK;
}
