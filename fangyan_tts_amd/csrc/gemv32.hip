// Decode-side products of the speech-token LM for up to 32 rows per weight pass (gemv32.h): the per-operation path of
// Qwen2Encoder.forward_one_step (CosyVoice/cosyvoice/llm/llm.py:246-258) when more than 8 sequences decode together -
// tts_pipeline's LM call over the batches of several steps - and of the prefill below the tiled-GEMM threshold.
//
//   * grid = (column tiles / NT, K slices, 32-row slices); a block owns NT 32-column tiles, its NW waves split the K
//     fragments of the block's K slice; per fragment a wave loads 1 KiB of weights per tile (non-temporal: read once per
//     pass) and the three 1 KiB planes of the operand's A image (from L2: every block reads the same image), then runs
//     3 NT v_mfma_f32_32x32x16_bf16.  DB fragments are in flight per wave; nothing is staged through LDS and no barrier
//     precedes the MFMAs.
//   * the waves' accumulators meet in LDS and are summed in wave order; K slices (the down projection, K = inter) hand
//     their partial tiles over in HBM and the last block to arrive adds them in slice order: fixed summation order, the
//     result does not depend on timing.
//   * epilogues write what the NEXT product reads: the residual stream, the A image of (next norm weight x new residual)
//     and the tile's partial sum of squares for the fused RMSNorm (GV32_ADD_IMG), or the A image of the SwiGLU output
//     (GV32_SWIGLU_IMG).  The consumer turns the partial sums into 1/rms per row and applies it to the finished sums.
#include "gemv32.h"
#include "runtime.h"
#include <algorithm>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 gv_frag;
typedef uint32_t gv_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ gv_frag gv32_ld_nt(const bf16_t* p) {
    gv_u32x4 r = __builtin_nontemporal_load(reinterpret_cast<const gv_u32x4*>(p));
    return __builtin_bit_cast(gv_frag, r);
}
__device__ __forceinline__ gv_frag gv32_ld(const bf16_t* p) {
    gv_u32x4 r = *reinterpret_cast<const gv_u32x4*>(p);
    return __builtin_bit_cast(gv_frag, r);
}

template <int NT, int NW, int DB, bool XW = false>
__global__ __launch_bounds__(NW * 64) void gemv32_k(const Gv32Args a) {
    extern __shared__ __attribute__((aligned(16))) float red[];          // [NW][NT][16][64]: the waves' accumulators (dynamic: 64 KB at NT = 4)
    __shared__ float rstd_s[32];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int tile0 = blockIdx.x * NT, KS = gridDim.y, ks = blockIdx.y, z = blockIdx.z;
    const int K16 = a.K / 16, NT32 = (a.N + 31) / 32, Rpad = 32 * gridDim.z;
    const int kper = (K16 + KS - 1) / KS, kbeg = ks * kper, kend = min(K16, kbeg + kper);
    // partial sums of squares of the operand rows (fused RMSNorm): requested first, summed after the products
    float ssq_p = 0.f;
    float ssq_v[4] = {0.f, 0.f, 0.f, 0.f};
    const bool do_rstd = a.ssq != nullptr && tid < 256;
    if (do_rstd) {
        const int r = tid >> 3, part = tid & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tt = part + 8 * i;
            if (tt < a.n_ssq) ssq_v[i] = a.ssq[(long)tt * Rpad + z * 32 + r];
        }
    }
    const bf16_t* ap = a.img + (long)z * K16 * 1536 + lane * 8;
    const bf16_t* wt[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wt[t] = a.W + (long)min(tile0 + t, NT32 - 1) * K16 * 512 + lane * 8;
    const long w_lo = XW ? a.W_lo - a.W : 0;                 // the lo plane's fragments sit at the same offsets
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    for (int kb = kbeg + wid; kb < kend; kb += NW * DB) {
        gv_frag b[NT][DB], b2[XW ? NT : 1][DB], af[DB][3];
#pragma unroll
        for (int u = 0; u < DB; ++u) {
            const int kk = kb + u * NW;
            if (kk < kend) {
#pragma unroll
                for (int t = 0; t < NT; ++t) b[t][u] = gv32_ld_nt(wt[t] + (long)kk * 512);
                if constexpr (XW) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) b2[t][u] = gv32_ld_nt(wt[t] + w_lo + (long)kk * 512);
                }
#pragma unroll
                for (int p = 0; p < 3; ++p) af[u][p] = gv32_ld(ap + ((long)kk * 3 + p) * 512);
            }
        }
#pragma unroll
        for (int u = 0; u < DB; ++u) {
            const int kk = kb + u * NW;
            if (kk < kend) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[u][p], b[t][u], acc[t], 0, 0, 0);
                if constexpr (XW) {
#pragma unroll
                    for (int p = 0; p < 2; ++p)
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[u][p], b2[t][u], acc[t], 0, 0, 0);
                }
            }
        }
    }
    // D: register i of lane l = row (i & 3) + 8 (i >> 2) + 4 (l >> 5), column l & 31
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[((wid * NT + t) * 16 + i) * 64 + lane] = acc[t][i];
    if (do_rstd) {
        // more than 32 partials per row (K > 1024) do not occur: the products with a fused norm have K = hidden
        ssq_p = (ssq_v[0] + ssq_v[1]) + (ssq_v[2] + ssq_v[3]);
        ssq_p += __shfl_xor(ssq_p, 1, 64); ssq_p += __shfl_xor(ssq_p, 2, 64); ssq_p += __shfl_xor(ssq_p, 4, 64);
        if ((tid & 7) == 0) rstd_s[tid >> 3] = rsqrtf(ssq_p / a.K + a.eps);
    }
    __syncthreads();
    // item = (tile slot t, row r, columns c0 .. c0+3): NT * 256 of them over the block's threads, eight consecutive threads per row
    constexpr int ITEMS = (NT * 256 + NW * 64 - 1) / (NW * 64);
    float4 v[ITEMS];
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
        const int it = tid + q * NW * 64;
        v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (it < NT * 256) {
            const int t = it >> 8, e = it & 255, r = e >> 3, c0 = (e & 7) * 4;
            const int i = (r & 3) + 4 * (r >> 3), kh = (r >> 2) & 1;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float4 x = *reinterpret_cast<const float4*>(red + ((w * NT + t) * 16 + i) * 64 + 32 * kh + c0);
                v[q].x += x.x; v[q].y += x.y; v[q].z += x.z; v[q].w += x.w;
            }
        }
    }
    if (KS > 1) {
        // K slices: the partial tiles go to HBM, the last block to arrive adds them in slice order (producer: stores, every
        // storing wave drains, barrier, one lane releases at agent scope and takes a ticket; consumer: acquire, barrier, loads)
        const long tile = (long)z * gridDim.x + blockIdx.x;
        float* slab = a.partial + tile * KS * (NT * 1024);
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            const int it = tid + q * NW * 64;
            if (it < NT * 256) *reinterpret_cast<float4*>(slab + (long)ks * (NT * 1024) + it * 4) = v[q];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int ticket = __hip_atomic_fetch_add(a.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = ticket == KS - 1;
            if (ticket == KS - 1) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                a.counters[tile] = 0;                        // ready for the next launch
            }
        }
        __syncthreads();
        if (!s_last) return;
#pragma unroll
        for (int q = 0; q < ITEMS; ++q) {
            const int it = tid + q * NW * 64;
            if (it < NT * 256) {
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int s2 = 0; s2 < KS; ++s2) {
                    const float4 x = *reinterpret_cast<const float4*>(slab + (long)s2 * (NT * 1024) + it * 4);
                    s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
                }
                v[q] = s;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
        const int it = tid + q * NW * 64;
        if (it >= NT * 256) continue;                        // whole waves: 8 threads per row, 64 | 256
        const int t = it >> 8, e = it & 255, r = e >> 3, c0 = (e & 7) * 4;
        const int row = z * 32 + r, n = (tile0 + t) * 32 + c0;
        const bool tile_ok = tile0 + t < NT32, row_ok = row < a.R;
        if (!tile_ok) continue;                              // uniform per wave (a wave's 64 items share t)
        const float rs = a.ssq ? rstd_s[r] : 1.f;
        float x[4] = {v[q].x * rs, v[q].y * rs, v[q].z * rs, v[q].w * rs};
        if (a.mode == GV32_STORE) {
            if (row_ok) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < a.N) a.y[(long)row * a.ldy + n + j] = x[j] + (a.bias ? a.bias[n + j] : 0.f);
            }
        } else if (a.mode == GV32_ADD_IMG) {
            // N % 32 == 0 (checked by the launcher): whole tiles
            float hn[4] = {0.f, 0.f, 0.f, 0.f};
            if (row_ok) {
                const float4 y0 = *reinterpret_cast<const float4*>(a.y + (long)row * a.ldy + n);
                hn[0] = y0.x + x[0]; hn[1] = y0.y + x[1]; hn[2] = y0.z + x[2]; hn[3] = y0.w + x[3];
                *reinterpret_cast<float4*>(a.y + (long)row * a.ldy + n) = make_float4(hn[0], hn[1], hn[2], hn[3]);
            }
            if (a.img_out) {
                const float4 w4 = *reinterpret_cast<const float4*>(a.ln_next + n);
                gv32_put4(a.img_out, a.N / 16, row, n, hn[0] * w4.x, hn[1] * w4.y, hn[2] * w4.z, hn[3] * w4.w);
                float s = (hn[0] * hn[0] + hn[1] * hn[1]) + (hn[2] * hn[2] + hn[3] * hn[3]);
                s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
                if ((e & 7) == 0) a.ssq_out[(long)(tile0 + t) * Rpad + row] = s;
            }
        } else {                                             // GV32_SWIGLU_IMG: columns (gate_i, up_i) interleaved, N % 32 == 0
            float o0 = 0.f, o1 = 0.f;
            if (row_ok) { o0 = act_silu(x[0]) * x[1]; o1 = act_silu(x[2]) * x[3]; }
            unsigned h0, m0, l0, h1, m1, l1;
            gv32_split3(o0, h0, m0, l0); gv32_split3(o1, h1, m1, l1);
            const int k = n >> 1;                            // even: a 4-byte pair inside one 16-byte piece
            char* base = reinterpret_cast<char*>(a.img_out);
            const int K16o = a.N / 32;
            *reinterpret_cast<unsigned*>(base + gv32_off(K16o, row, k, 0)) = h0 | (h1 << 16);
            *reinterpret_cast<unsigned*>(base + gv32_off(K16o, row, k, 1)) = m0 | (m1 << 16);
            *reinterpret_cast<unsigned*>(base + gv32_off(K16o, row, k, 2)) = l0 | (l1 << 16);
        }
    }
}

// x fp32 [R][ldx] -> A image of (ln * x), per-tile sums of x^2.  grid (K/32, slices), 256 threads: thread = (row, 4 columns)
__global__ __launch_bounds__(256) void gv32_split_rows_k(const float* __restrict__ x, int ldx, int R, int K, const float* __restrict__ ln,
                                                         bf16_t* __restrict__ img, float* __restrict__ ssq) {
    const int e = threadIdx.x, r = e >> 3, c0 = (e & 7) * 4;
    const int row = blockIdx.y * 32 + r, n = blockIdx.x * 32 + c0, Rpad = 32 * gridDim.y;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < R) v = *reinterpret_cast<const float4*>(x + (long)row * ldx + n);
    float4 w = make_float4(1.f, 1.f, 1.f, 1.f);
    if (ln) w = *reinterpret_cast<const float4*>(ln + n);
    gv32_put4(img, K / 16, row, n, v.x * w.x, v.y * w.y, v.z * w.z, v.w * w.w);
    if (ssq) {
        float s = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if ((e & 7) == 0) ssq[(long)blockIdx.x * Rpad + row] = s;
    }
}

int gv32_split_rows(const float* x, int ldx, int R, int K, const float* ln, bf16_t* img, float* ssq, hipStream_t st) {
    FY_CHECK(x && img && R >= 1 && K >= 32 && K % 32 == 0 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0, FY_ERR_ARG, "gv32_split_rows: bad arguments R %d K %d", R, K);
    hipLaunchKernelGGL(gv32_split_rows_k, dim3(K / 32, cdiv(R, 32)), dim3(256), 0, st, x, ldx, R, K, ln, img, ssq);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// split-K (K >= 2048: the down projection): [row slices][tiles][KS][32 x 32] floats and one arrival counter per (slice, tile)
static constexpr int GV32_KS_MAX = 16;
size_t gv32_partial_floats(int R, int N, int K) { return K >= 2048 ? (size_t)cdiv(R, 32) * cdiv(N, 32) * GV32_KS_MAX * 1024 : 0; }
size_t gv32_counter_ints(int R, int N, int K) { return K >= 2048 ? (size_t)cdiv(R, 32) * cdiv(N, 32) : 0; }

static int gv32_env(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

int gemv32(const Gv32Args& a, hipStream_t st) {
    FY_CHECK(a.W && a.img && a.R >= 1 && a.N >= 1 && a.K >= 16 && a.K % 16 == 0, FY_ERR_ARG, "gemv32: bad arguments R %d N %d K %d", a.R, a.N, a.K);
    FY_CHECK((((uintptr_t)a.W | (uintptr_t)a.W_lo | (uintptr_t)a.img) & 15) == 0, FY_ERR_ARG, "gemv32: weights and image must be 16-B aligned");
    FY_CHECK(!a.ssq || (a.n_ssq >= 1 && a.n_ssq <= 32), FY_ERR_ARG, "gemv32: %d partial sums of squares per row (1..32)", a.n_ssq);
    if (a.mode == GV32_STORE) FY_CHECK(a.y && a.ldy >= a.N, FY_ERR_ARG, "gemv32: GV32_STORE needs y");
    if (a.mode == GV32_ADD_IMG)
        FY_CHECK(a.y && a.N % 32 == 0 && a.ldy % 4 == 0 && ((uintptr_t)a.y & 15) == 0 && (!a.img_out || (a.ln_next && a.ssq_out)), FY_ERR_ARG,
                 "gemv32: GV32_ADD_IMG needs y (16-B aligned rows), N %% 32 == 0, and ln_next + ssq_out with img_out");
    if (a.mode == GV32_SWIGLU_IMG) FY_CHECK(a.img_out && a.N % 32 == 0, FY_ERR_ARG, "gemv32: GV32_SWIGLU_IMG needs img_out and N %% 32 == 0");
    const int tiles = cdiv(a.N, 32), Z = cdiv(a.R, 32);
    ProfScope prof("gemv", (a.W_lo ? 4.0 : 2.0) * a.N * a.K, st);                 // work = the product's bf16 weight bytes
    // Launch shapes (tests/micro/gemv32_bench.hip): K slices only for the long K of the down projection; two tiles per block
    // where one tile per block would need a second round of blocks (gate/up: 304 tiles); 8 waves where few tiles exist
    static const int ks_down = gv32_env("FY_GV32_KS", 8), nt2_from = gv32_env("FY_GV32_NT2_FROM", 257), nw8_below = gv32_env("FY_GV32_NW8_BELOW", 128);
    int KS = 1;
    if (a.K >= 2048 && a.partial && a.counters) KS = std::min(std::max(1, ks_down), GV32_KS_MAX);
    auto lds_of = [](int nt, int nw) { return (size_t)nw * nt * 16 * 64 * sizeof(float); };
    // (four tiles per block - the A image read once per 128 columns - measured slower: 14.0 / 29.1 us against 11.5 / 12.9 for
    // gate/up and down alone, 63.0 against 61.6 ms per pipelined step)
    if (a.W_lo) {
        // two weight planes per fragment: fewer fragments in flight per wave so that both planes' registers fit
        if (tiles >= nt2_from && KS == 1) hipLaunchKernelGGL((gemv32_k<2, 4, 3, true>), dim3(cdiv(tiles, 2), 1, Z), dim3(256), lds_of(2, 4), st, a);
        else if (tiles < nw8_below && KS == 1) hipLaunchKernelGGL((gemv32_k<1, 8, 4, true>), dim3(tiles, 1, Z), dim3(512), lds_of(1, 8), st, a);
        else hipLaunchKernelGGL((gemv32_k<1, 4, 5, true>), dim3(tiles, KS, Z), dim3(256), lds_of(1, 4), st, a);
    } else if (tiles >= nt2_from && KS == 1) {
        hipLaunchKernelGGL((gemv32_k<2, 4, 7>), dim3(cdiv(tiles, 2), 1, Z), dim3(256), lds_of(2, 4), st, a);
    } else if (tiles < nw8_below && KS == 1) {
        hipLaunchKernelGGL((gemv32_k<1, 8, 7>), dim3(tiles, 1, Z), dim3(512), lds_of(1, 8), st, a);
    } else {
        hipLaunchKernelGGL((gemv32_k<1, 4, 7>), dim3(tiles, KS, Z), dim3(256), lds_of(1, 4), st, a);
    }
    HIP_TRY(hipGetLastError());
    return FY_OK;
}
