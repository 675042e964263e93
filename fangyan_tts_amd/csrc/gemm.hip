// See gemm.h.
#include "gemm.h"
#include "runtime.h"
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;

#define GM_BM 128
int gemm_tile_override = 0;        // microbenchmarks: 0 auto; 128 / 64 register-staged kernel with that M tile; 2 / 3 / 256 ring kernel with 256x128 / 128x128 / 256x256 tiles
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

// x-transformers apply_rotary_pos_emb on four consecutive output columns starting at n (two interleaved pairs)
__device__ __forceinline__ void epi_rope(float4& v, const GemmEpi& e, int m, int n) {
    int nn = n >= e.rope_stride ? n - e.rope_stride : n;
    if (n >= 2 * e.rope_stride || nn >= 2 * e.rope_half) return;
    const float2* tab = e.rope + (long)(m % e.rope_T) * e.rope_half + (nn >> 1);
    const float2 c0 = tab[0], c1 = tab[1];
    const float a = v.x, b = v.y, c = v.z, d = v.w;
    v.x = a * c0.x - b * c0.y; v.y = b * c0.x + a * c0.y;
    v.z = c * c1.x - d * c1.y; v.w = d * c1.x + c * c1.y;
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
    constexpr int A_ELEMS = (PRECISE ? PRECISE : 1) * BM * GM_PITCH, B_ELEMS = GM_BN * GM_PITCH, BUF = A_ELEMS + B_ELEMS;
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
            if (PRECISE) {
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
#define G2_RING_BYTES (96 * 1024)                            // BN 256: three 32 KB stages; BN 128: three 24 KB stages = 72 KB, so that two workgroups share a CU

// BN = 256: 8 waves as 2 (M) x 4 (N), 128x64 per wave.  BN = 128: 4 x 2 waves of 64x64, <= 128 VGPRs and a 72 KB ring, so
// two workgroups are resident per CU: one's prologue / epilogue bursts run under the other's K loop (the four DiT
// products of a block: 184 -> 168 us), and the N = 1024 products get twice the tiles.
template <int EPI, int BN, int BM = 256>
__global__ __launch_bounds__(512) void gemm256_k(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W, int M, int N, int K, GemmEpi e) {
    extern __shared__ __attribute__((aligned(16))) bf16_t gm_smem[];
    char* smem = reinterpret_cast<char*>(gm_smem);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    constexpr int NWAVE = BM == 128 ? 4 : 8;                             // BM 128 (with BN 128): 2 x 2 waves of 64x64, 48 KB ring, three workgroups per CU
    constexpr int WN = BN / 64, WM = NWAVE / WN, MI = BM / WM / 32;       // waves along N / M, 32-row tiles per wave
    constexpr int STAGE = (BM + BN) * G2_BK * 2, STAGES = (BN == 128 ? 3 : G2_RING_BYTES / STAGE), NB = BN / (NWAVE * 16);   // NB: B DMAs per wave and stage
    constexpr int BOFF = BM * G2_BK * 2;                                  // the B tile follows the A tile inside a stage
    const int wm = wid / WN, wn = wid % WN, lr = lane & 31, kh = lane >> 5;
    // XCD-aware tile order (see gemm_bf16_k)
    const int ntn = N / BN, nwg = gridDim.x;
    const int orig = blockIdx.x, xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int m0 = (wg / ntn) * BM, n0 = (wg % ntn) * BN;

    // wave-instruction i (of 2) of wave wid fills stage bytes [(wid*2+i)*1024, +1024) of the A (and B) tile:
    // row = (wid*2+i)*16 + lane/4, slot = lane%4  ->  logical chunk = slot ^ ((row >> 2) & 3)
    const bf16_t* a_src[2];
    const bf16_t* b_src[NB];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wid * 2 + i) * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
        a_src[i] = A + (long)min(m0 + row, M - 1) * lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int row = (wid * NB + i) * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
        b_src[i] = W + (long)min(n0 + row, N - 1) * K + c * 8;
    }
    auto issue = [&](int t) {
        char* st = smem + (t % STAGES) * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const void*)(a_src[i] + t * G2_BK), (__attribute__((address_space(3))) void*)(st + wid * 2048 + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_global_load_lds((const void*)(b_src[i] + t * G2_BK), (__attribute__((address_space(3))) void*)(st + BOFF + wid * NB * 1024 + i * 1024), 16, 0, 0);
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
    for (int i = 0; i < MI; ++i) { const int ra = wm * (MI * 32) + i * 32 + lr; a_off[i] = ra * 64; a_sw[i] = (ra >> 2) & 3; }
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int rb = wn * 64 + i * 32 + lr; b_off[i] = BOFF + rb * 64; b_sw[i] = (rb >> 2) & 3; }

    const int nt = K / G2_BK;
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < nt) issue(t);
    for (int t = 0; t < nt; ++t) {
        // stage t has landed once at most the stages issued after it are still outstanding (2 + NB DMAs per stage per wave)
        const int ahead = min(nt - 1 - t, STAGES - 2);
        constexpr int D = 2 + NB;
        if (ahead >= 2) __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * D));   // vmcnt(2 D): D <= 4, so the count fits the low four bits
        else if (ahead == 1) __builtin_amdgcn_s_waitcnt(0x0F70 | D);
        else __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0)
        __builtin_amdgcn_s_barrier();                               // everyone's part of stage t is in LDS; the slot of stage t-1 is free
        if (t + STAGES - 1 < nt) issue(t + STAGES - 1);
        const char* sb = smem + (t % STAGES) * STAGE;
        frag_ab fa[2][MI], fb[2][2];
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
    __syncthreads();                                                // all waves are done with the ring: the epilogue parks tiles in it

    // epilogue through LDS, a quarter of the wave's tile (32 rows x 64 columns) at a time: see gemm_epilogue
    float* park = reinterpret_cast<float*>(smem) + wid * 32 * 68;
    const int c4 = (lane & 15) * 4, rsub = lane >> 4;
    const int n = n0 + wn * 64 + c4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), gv = bv;
    if (e.bias) bv = *reinterpret_cast<const float4*>(e.bias + n);
    if (EPI == 3) gv = *reinterpret_cast<const float4*>(e.gate + n);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
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
        const int mb = m0 + wm * (MI * 32) + mi * 32 + rsub;
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
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (m >= M) continue;
            if (EPI == 3) {
                float4 o = old[it];
                o.x = fmaf(gv.x, v.x, o.x); o.y = fmaf(gv.y, v.y, o.y); o.z = fmaf(gv.z, v.z, o.z); o.w = fmaf(gv.w, v.w, o.w);
                *reinterpret_cast<float4*>(e.resid + (long)m * e.ldc + n) = o;
            } else if (EPI == 2) {
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
}

template <int EPI, int BN, int BM = 256>
static int gemm_launch_256(const bf16_t* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st) {
    static bool attr_set = false;
    const size_t lds = BN == 128 ? (size_t)3 * (BM + BN) * G2_BK * 2 : (size_t)G2_RING_BYTES;   // BN 128: 72 KB (two workgroups per CU) or 48 KB (three)
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)gemm256_k<EPI, BN, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid((N / BN) * cdiv(M, BM));
    hipLaunchKernelGGL((gemm256_k<EPI, BN, BM>), grid, dim3(BM == 128 ? 256 : 512), lds, st, A, lda, W, M, N, K, epi);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

template <int PRECISE, int EPI, int BM>
static int gemm_launch3(const void* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st) {
    static bool attr_set = false;
    const size_t lds = (size_t)2 * ((PRECISE ? PRECISE : 1) * BM + GM_BN) * GM_PITCH * sizeof(bf16_t);
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void*)gemm_bf16_k<PRECISE, EPI, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
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
    // Measured on MI355X with tests/micro/gemm_bench (us; register-staged 128x128x64 / ring 256x256, one workgroup per CU /
    // ring 256x128, two per CU):
    //   M 6400:  N 3072 K 1024: 68 / 70 / 57    N 2048 K 1024: 58 / 42 / 42    N 1024 K 1024: 32 / 37 / 27    N 1024 K 2048: 48 / 62 / 42
    //   M 12800: N 3072 K 1024: 119 / 115 / 97  N 2048 K 1024: 95 / 87 / 83    N 1024 K 1024: 47 / 38 / 38
    //   M 3200:  N 3072 K 1024: 41 / 39 / 37    N 1024 K 1024: 21.6 / 36 / 22.8   N 1024 K 2048: 34 / 61 / 37   (104 tiles: too few)
    // The 256x256 tile is ~15 % faster inside the K loop but never ahead overall: every tile pays ~10 us of prologue /
    // epilogue bursts, which only a second resident workgroup hides; it stays selectable for the microbenchmark.
    if (!PRECISE && gemm_tile_override != 128 && N % 128 == 0 && K % G2_BK == 0) {
        static int cus = 0;
        if (!cus) {
            int dev = 0;
            hipDeviceProp_t p;
            cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
        }
        if (gemm_tile_override == 256 && N % 256 == 0) return gemm_launch_256<EPI, 256>((const bf16_t*)A, lda, W, M, N, K, epi, st);
        const int t128 = (N / 128) * cdiv(M, 256);
        if (gemm_tile_override == 2 || (gemm_tile_override == 0 && 5 * t128 >= 3 * cus)) return gemm_launch_256<EPI, 128>((const bf16_t*)A, lda, W, M, N, K, epi, st);
        // fewer tiles than that: the same ring with 128x128 tiles (4 waves, 48 KB, three workgroups per CU) - at M = 3200
        // 18 / 29 / 36 us for the out / ff2 / qkv shapes against 21 / 34 / 41 with the register-staged kernel
        return gemm_launch_256<EPI, 128, 128>((const bf16_t*)A, lda, W, M, N, K, epi, st);
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
    FY_CHECK(!e.rope || (e.mode == EPI_STORE && e.act == ACT_NONE && e.rope_T >= 1 && e.rope_half >= 2 && e.rope_half % 2 == 0 &&
                         e.rope_stride >= 2 * e.rope_half), FY_ERR_ARG, "gemm: bad rotary epilogue");
    return FY_OK;
}

int gemm_bf16(const bf16_t* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st) {
    FY_TRY(gemm_check(A, lda, W, M, N, K, epi, 2));
    ProfScope prof("gemm_bf16", 2.0 * M * N * K, st);
    return gemm_launch<0>(A, lda, W, M, N, K, epi, st);
}

int gemm_f32a_precise(const float* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st) {
    FY_TRY(gemm_check(A, lda, W, M, N, K, epi, 4));
    return gemm_launch<2>(A, lda, W, M, N, K, epi, st);
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
        static bool attr_set = false;
        const size_t lds = (size_t)24 * GV_PITCH * 2 + 4 * 256 * 4 + 8 * 4;
        if (!attr_set) {
            HIP_TRY(hipFuncSetAttribute((const void*)gemv_lds_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        hipLaunchKernelGGL(gemv_lds_k, grid, dim3(256), lds, st, a);
    }
    HIP_TRY(hipGetLastError());
    return FY_OK;
}
