// Causal 1-D convolutions for the HiFT vocoder, the f0 predictor, PreLookahead
// and the DiT convolutional position embedding.  See conv.h.
//
// Reference semantics: CausalConv1d / CausalConv1dDownSample / CausalConv1dUpsample,
// CosyVoice/cosyvoice/transformer/convolution.py:150-258 (zero pad on the causal
// side, nearest-repeat upsample before the conv), ResBlock / Snake fusion points
// from hifigan/generator.py:110-117, 682-700.
#include "conv.h"
#include "runtime.h"
#include <algorithm>
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;
// -DFY_CONV_STAMPS: wave 0 of every workgroup adds the cycles it spent in each phase of the MFMA kernel to fy_dbg[]
// (tests/micro/conv_bench.hip prints the split); off in the library
#ifdef FY_CONV_STAMPS
__device__ unsigned long long fy_dbg[8];
void conv_dbg_read(unsigned long long* out, bool reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(fy_dbg), sizeof(fy_dbg));
    if (reset) { unsigned long long z[8] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(fy_dbg), z, sizeof(z)); }
}
#define DBG_T(i) do { long long _n = clock64(); if (tid == 0) atomicAdd(&fy_dbg[i], (unsigned long long)(_n - _t)); _t = _n; } while (0)
#else
#define DBG_T(i)
#endif

__device__ __forceinline__ int floordiv(int a, int b) {   // b > 0
    return a >= 0 ? a / b : -((-a + b - 1) / b);
}

__device__ __forceinline__ float apply_pre(float v, int act, float slope, float alpha) {
    if (act == ACT_LEAKY) return act_leaky(v, slope);
    if (act == ACT_SNAKE) return act_snake(v, alpha);
    return v;
}
__device__ __forceinline__ float apply_post(float v, int act, float slope) {
    if (act == ACT_ELU) return act_elu(v);
    if (act == ACT_LEAKY) return act_leaky(v, slope);
    if (act == ACT_MISH) return act_mish(v);
    return v;
}

__device__ __forceinline__ void conv_store(const ConvDesc& d, int b, int p, int co, float v) {
    // epilogue shared by both kernels: bias already added by the caller
    v = apply_post(v, d.post_act, d.post_slope);
    int q = d.reflect1 ? p + 1 : p;
    long yo = (long)b * d.y_bs + (long)q * d.y_ld + co;
    float r = v;
    if (d.add_resid) r += d.resid[(long)b * d.r_bs + (long)q * d.r_ld + co];
    r *= d.out_scale;
    if (d.accumulate) d.y[yo] += r; else d.y[yo] = r;
    if (d.reflect1 && p == 1) {                      // q = 0 mirrors conv row 1
        long y0 = (long)b * d.y_bs + co;
        float r0 = v;
        if (d.add_resid) r0 += d.resid[(long)b * d.r_bs + co];
        r0 *= d.out_scale;
        if (d.accumulate) d.y[y0] += r0; else d.y[y0] = r0;
    }
}

// =============================================================================
// fp32 direct kernel: block = 64 positions x 64 output channels, 4x4 per thread
// =============================================================================
#define DIR_TP 64
#define DIR_TCO 64
#define DIR_CC 16

__global__ __launch_bounds__(256) void conv1d_f32_direct_k(ConvDesc d, const float* __restrict__ w, int cout_pad, int nrows_max) {
    extern __shared__ __attribute__((aligned(16))) float xs[];      // [nrows_max][DIR_CC]
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int b = blockIdx.z;
    const int Cin_g = d.Cin / d.groups, Cout_g = d.Cout / d.groups;
    const int tiles_per_g = (Cout_g + DIR_TCO - 1) / DIR_TCO;
    const int g = blockIdx.y / tiles_per_g;
    const int co_in_g = (blockIdx.y % tiles_per_g) * DIR_TCO + tx * 4;
    const int co = g * Cout_g + co_in_g;
    const int ci_base = g * Cin_g;
    const int p0 = blockIdx.x * DIR_TP;
    const int n_out = d.out_len ? d.out_len[b] : d.L_out;
    if (p0 >= n_out) return;
    const int n_in = d.in_len ? d.in_len[b] : d.L_in;
    const int row_lo = floordiv(p0 * d.stride - d.pad_left, d.up);
    const int row_hi = floordiv((p0 + DIR_TP - 1) * d.stride + (d.KW - 1) * d.dil - d.pad_left, d.up);
    const int nrows = row_hi - row_lo + 1;
    const float* xb = d.x + (long)b * d.x_bs;

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int ci0 = 0; ci0 < Cin_g; ci0 += DIR_CC) {
        const int ccn = min(DIR_CC, Cin_g - ci0);
        __syncthreads();
        for (int idx = tid; idx < nrows * DIR_CC; idx += 256) {
            int r = idx / DIR_CC, c = idx % DIR_CC;
            int row = row_lo + r;
            float v = 0.f;
            if (row >= 0 && row < n_in && c < ccn) {
                int ci = ci_base + ci0 + c;
                v = xb[(long)row * d.x_ld + ci];
                v = apply_pre(v, d.pre_act, d.pre_slope, d.pre_act == ACT_SNAKE ? d.alpha[ci] : 0.f);
            }
            xs[idx] = v;
        }
        __syncthreads();
        if (co_in_g < Cout_g) {
            for (int t = 0; t < d.KW; ++t) {
                int rr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int s = (p0 + ty * 4 + i) * d.stride + t * d.dil - d.pad_left;
                    rr[i] = ((d.up == 1) ? s : floordiv(s, d.up)) - row_lo;
                }
                const float* wt = w + ((long)(t * Cin_g + ci0)) * cout_pad + co;
                for (int c = 0; c < ccn; ++c) {
                    float4 wv = *reinterpret_cast<const float4*>(wt + (long)c * cout_pad);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float xv = xs[rr[i] * DIR_CC + c];
                        acc[i][0] = fmaf(xv, wv.x, acc[i][0]);
                        acc[i][1] = fmaf(xv, wv.y, acc[i][1]);
                        acc[i][2] = fmaf(xv, wv.z, acc[i][2]);
                        acc[i][3] = fmaf(xv, wv.w, acc[i][3]);
                    }
                }
            }
        }
    }
    if (co_in_g >= Cout_g) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int p = p0 + ty * 4 + i;
        if (p >= n_out) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int c = co + j;
            if (co_in_g + j < Cout_g && c < d.Cout) conv_store(d, b, p, c, acc[i][j] + (d.bias ? d.bias[c] : 0.f));
        }
    }
}

int conv1d_f32_direct(const ConvDesc& d, const ConvW& w, hipStream_t st) {
    FY_CHECK(w.w_dir != nullptr, FY_ERR_STATE, "conv1d_f32_direct: weights not packed for the direct kernel");
    FY_CHECK(d.Cin == w.Cin && d.Cout == w.Cout && d.KW == w.KW && d.groups == w.groups, FY_ERR_ARG,
             "conv1d_f32_direct: descriptor (%d,%d,%d,g%d) != weights (%d,%d,%d,g%d)", d.Cin, d.Cout, d.KW, d.groups,
             w.Cin, w.Cout, w.KW, w.groups);
    FY_CHECK(d.up >= 1 && d.stride >= 1 && d.dil >= 1 && d.B >= 1 && d.L_out >= 1, FY_ERR_ARG, "conv1d_f32_direct: bad geometry");
    FY_CHECK((d.Cout / d.groups) % 4 == 0 || d.groups == 1, FY_ERR_ARG, "conv1d_f32_direct: Cout/groups must be a multiple of 4");
    int nrows_max = ((DIR_TP - 1) * d.stride + (d.KW - 1) * d.dil) / d.up + 3;
    size_t lds = (size_t)nrows_max * DIR_CC * sizeof(float);
    FY_CHECK(lds <= 160 * 1024, FY_ERR_ARG, "conv1d_f32_direct: input tile needs %zu B of LDS", lds);
    int Cout_g = d.Cout / d.groups;
    dim3 grid(cdiv(d.L_out, DIR_TP), d.groups * cdiv(Cout_g, DIR_TCO), d.B);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)conv1d_f32_direct_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(conv1d_f32_direct_k, grid, dim3(256), lds, st, d, w.w_dir, w.cout_pad4(), nrows_max);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// =============================================================================
// bf16 MFMA implicit-GEMM kernel
//   block = WAVES_P x WAVES_C waves, each wave a 64-position x 64-channel tile
//   (2x2 v_mfma_f32_32x32x16_bf16 accumulators).  D[pos][co] = sum_{t,ci}
//   X[pos + t*dil - pad][ci] * W[t][ci][co]; A = activations out of LDS,
//   B = weights in fragment order straight from L2.
// =============================================================================
#define MF_CC 128                      // input channels staged per pass
// LDS row pitch in bytes: the staged channels as bf16 + a 16-B pad against bank conflicts
static __host__ __device__ inline int mf_rowb(int cin_gp) { return (cin_gp < MF_CC ? cin_gp : MF_CC) * 2 + 16; }

// NK: 16-wide k steps per staged channel chunk when that is the same for every chunk of the launch (4, 6, 8);
// 0 = decided at run time (odd channel counts of reduced-size models).  A compile-time NK and clamped addresses
// keep the hot loops free of branches, so the compiler batches their loads instead of waiting on each one.
template <int WAVES_P, int WAVES_C, bool PRECISE, int NK, bool INB>
__global__ __launch_bounds__(WAVES_P* WAVES_C * 64) void conv1d_bf16_mfma_k(ConvDesc d, const bf16_t* __restrict__ wp, int nrows_max) {
    constexpr int TP = WAVES_P * 64, TCO = WAVES_C * 64, NT = WAVES_P * WAVES_C * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];    // hi tile [nrows_max][MF_ROWB] (+ lo tile when PRECISE)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wpi = wid / WAVES_C, wci = wid % WAVES_C;
    const int b = blockIdx.z;
    const int Cin_g = d.Cin / d.groups, Cout_g = d.Cout / d.groups;
    const int Cin_gp = ((Cin_g + 15) / 16) * 16;
    const int C16 = Cin_gp / 16, N32 = (Cout_g + 31) / 32;   // a ragged last 32-column tile has zero weights (groups == 1 only)
    const int MF_ROWB = mf_rowb(Cin_gp);
    const int tiles_per_g = (Cout_g + TCO - 1) / TCO;
    const int g = blockIdx.y / tiles_per_g;
    const int n32_base = (blockIdx.y % tiles_per_g) * (TCO / 32) + wci * 2;   // this wave's first 32-column tile in the group
    const int ci_base = g * Cin_g;
    const int p0 = blockIdx.x * TP;
    const int n_out = d.out_len ? d.out_len[b] : d.L_out;
    if (p0 >= n_out) return;
    const int n_in = d.in_len ? d.in_len[b] : d.L_in;
    const int row_lo = floordiv(p0 - d.pad_left, d.up);
    const int row_hi = floordiv(p0 + TP - 1 + (d.KW - 1) * d.dil - d.pad_left, d.up);
    const int nrows = row_hi - row_lo + 1;
    const float* xb = d.x + (long)b * d.x_bs;
    char* lo_tile = smem + (size_t)nrows_max * MF_ROWB;
#ifdef FY_CONV_STAMPS
    long long _t = clock64();
#endif
    const bool wave_live = n32_base < N32;          // whole wave beyond the group's channels: only helps staging
    const bool tile1_live = n32_base + 1 < N32;

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int kh = lane >> 5, lr = lane & 31;
    for (int ci0 = 0; ci0 < Cin_gp; ci0 += MF_CC) {
        const int ccn = min(MF_CC, Cin_gp - ci0);          // multiple of 16
        const int q4 = ccn / 4;
        __syncthreads();
        DBG_T(0);
        // branch-free and unrolled: out-of-range rows / channels read a clamped address and are zeroed by a select,
        // and the activation kind is a compile-time constant of the loop body, so several loads stay in flight
        const int total = nrows * q4;
        const bool q4_pow2 = (q4 & (q4 - 1)) == 0;
        const int q4_sh = 31 - __clz(q4);
        const int row_max = max(n_in - 1, 0), ci_max = max(Cin_g - 4, 0);
        auto stage = [&](auto act_tag) {
            constexpr int ACT = decltype(act_tag)::value;
            constexpr int SB = 4;                              // items per batch: SB (x2 with Snake) loads in flight per thread
            for (int base = tid; base < total; base += NT * SB) {
                float4 vv[SB], aa[SB];
                int rr[SB], cc[SB];
                bool okk[SB];
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int idx = min(base + u * NT, total - 1);
                    const int r = q4_pow2 ? (idx >> q4_sh) : idx / q4;
                    const int c4 = (idx - r * q4) * 4;
                    const int row = row_lo + r, ci = ci0 + c4;
                    rr[u] = r; cc[u] = c4;
                    okk[u] = row >= 0 && row < n_in && ci < Cin_g;
                    vv[u] = *reinterpret_cast<const float4*>(xb + (long)min(max(row, 0), row_max) * d.x_ld + ci_base + min(ci, ci_max));
                    aa[u] = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (ACT == ACT_SNAKE) aa[u] = *reinterpret_cast<const float4*>(d.alpha + ci_base + min(ci, ci_max));
                }
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    if (base + u * NT >= total) break;
                    float4 v = vv[u];
                    const float4 a = aa[u];
                    const int r = rr[u], c4 = cc[u];
                    if (!okk[u]) v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ACT == ACT_LEAKY) {
                        v.x = act_leaky(v.x, d.pre_slope); v.y = act_leaky(v.y, d.pre_slope);
                        v.z = act_leaky(v.z, d.pre_slope); v.w = act_leaky(v.w, d.pre_slope);
                    } else if (ACT == ACT_SNAKE) {
                        if (PRECISE) {
                            v.x = act_snake(v.x, a.x); v.y = act_snake(v.y, a.y);
                            v.z = act_snake(v.z, a.z); v.w = act_snake(v.w, a.w);
                        } else {
                            // the operand is rounded to bf16 next: hardware sine / reciprocal (~1e-6) are ample
                            v.x = snake_fast(v.x, a.x); v.y = snake_fast(v.y, a.y);
                            v.z = snake_fast(v.z, a.z); v.w = snake_fast(v.w, a.w);
                        }
                    }
                    bf16_t h0 = f32_to_bf16(v.x), h1 = f32_to_bf16(v.y), h2 = f32_to_bf16(v.z), h3 = f32_to_bf16(v.w);
                    uint2 pk;
                    pk.x = (uint32_t)h0 | ((uint32_t)h1 << 16);
                    pk.y = (uint32_t)h2 | ((uint32_t)h3 << 16);
                    *reinterpret_cast<uint2*>(smem + (size_t)r * MF_ROWB + c4 * 2) = pk;
                    if (PRECISE) {
                        bf16_t l0 = f32_to_bf16(v.x - bf16_to_f32(h0)), l1 = f32_to_bf16(v.y - bf16_to_f32(h1));
                        bf16_t l2 = f32_to_bf16(v.z - bf16_to_f32(h2)), l3 = f32_to_bf16(v.w - bf16_to_f32(h3));
                        pk.x = (uint32_t)l0 | ((uint32_t)l1 << 16);
                        pk.y = (uint32_t)l2 | ((uint32_t)l3 << 16);
                        *reinterpret_cast<uint2*>(lo_tile + (size_t)r * MF_ROWB + c4 * 2) = pk;
                    }
                }
            }
        };
        if (INB) {
            // the producer already activated and rounded this operand: a straight copy, every load of the tile in
            // flight at once (<= SBB 16-byte chunks per thread), zero rows selected in registers
            constexpr int SBB = 12;
            const int q8 = ccn / 8, total8 = nrows * q8;
            const bool q8_pow2 = (q8 & (q8 - 1)) == 0;
            const int q8_sh = 31 - __clz(q8);
            const bf16_t* xa = d.x_act + (long)b * d.x_bs;
            for (int base = tid; base < total8; base += NT * SBB) {
                uint4 vv[SBB];
                int off[SBB];
#pragma unroll
                for (int u = 0; u < SBB; ++u) {
                    const int idx = min(base + u * NT, total8 - 1);
                    const int r = q8_pow2 ? (idx >> q8_sh) : idx / q8;
                    const int c8 = (idx - r * q8) * 8;
                    const int row = row_lo + r;
                    vv[u] = *reinterpret_cast<const uint4*>(xa + (long)min(max(row, 0), row_max) * d.x_ld + ci_base + ci0 + c8);
                    off[u] = (row >= 0 && row < n_in) ? r * MF_ROWB + c8 * 2 : -1 - (r * MF_ROWB + c8 * 2);
                }
#pragma unroll
                for (int u = 0; u < SBB; ++u) {
                    if (base + u * NT >= total8) break;
                    const bool ok = off[u] >= 0;
                    *reinterpret_cast<uint4*>(smem + (ok ? off[u] : -1 - off[u])) = ok ? vv[u] : make_uint4(0, 0, 0, 0);
                }
            }
        } else if (d.pre_act == ACT_SNAKE) stage(std::integral_constant<int, ACT_SNAKE>{});
        else if (d.pre_act == ACT_LEAKY) stage(std::integral_constant<int, ACT_LEAKY>{});
        else stage(std::integral_constant<int, ACT_NONE>{});
        DBG_T(1);
        __syncthreads();
        DBG_T(2);
        if (wave_live) {
        const int c16_0 = ci0 / 16, nk = NK ? NK : ccn / 16;  // nk <= MF_CC/16 = 8
        constexpr int KMAX = NK ? NK : MF_CC / 16;
        const int t1off = tile1_live ? 512 : 0;               // a dead second tile re-reads the first: no branch
        // weights stream from L2 in half-tap batches (KH k-steps x two 32-column tiles), one batch ahead of the
        // MFMAs that use them; two batches of registers only, so two workgroups still fit a CU
        constexpr int KH = KMAX / 2;
        auto load_b = [&](int t, int half, frag_ab (&bb)[KH][2]) {
            const bf16_t* wt = wp + ((((long)g * d.KW + t) * C16 + c16_0 + half * KH) * N32 + n32_base) * 512 + lane * 8;
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) {
                if (NK || half * KH + kk < nk) {
                    bb[kk][0] = *reinterpret_cast<const frag_ab*>(wt + (long)kk * N32 * 512);
                    bb[kk][1] = *reinterpret_cast<const frag_ab*>(wt + (long)kk * N32 * 512 + t1off);
                }
            }
        };
        auto tap_half = [&](int t, int half, const frag_ab (&bb)[KH][2]) {
            int arow[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                int s = p0 + wpi * 64 + mi * 32 + lr + t * d.dil - d.pad_left;
                arow[mi] = ((d.up == 1) ? s : floordiv(s, d.up)) - row_lo;
            }
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) {
                if (NK || half * KH + kk < nk) {
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi) {
                        size_t off = (size_t)arow[mi] * MF_ROWB + ((half * KH + kk) * 16 + kh * 8) * 2;
                        frag_ab a = *reinterpret_cast<const frag_ab*>(smem + off);
                        acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb[kk][0], acc[mi][0], 0, 0, 0);
                        acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb[kk][1], acc[mi][1], 0, 0, 0);
                        if (PRECISE) {
                            frag_ab al = *reinterpret_cast<const frag_ab*>(lo_tile + off);
                            acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bb[kk][0], acc[mi][0], 0, 0, 0);
                            acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bb[kk][1], acc[mi][1], 0, 0, 0);
                        }
                    }
                }
            }
        };
        frag_ab h0[KH][2], h1[KH][2];
        load_b(0, 0, h0);
        for (int t = 0; t < d.KW; ++t) {
            load_b(t, 1, h1);
            tap_half(t, 0, h0);
            if (t + 1 < d.KW) load_b(t + 1, 0, h0);
            tap_half(t, 1, h1);
        }
        }
    }
    // epilogue through LDS: the accumulator layout (channel on the lane, positions in registers) would move 4 bytes
    // per lane and store; each wave parks half of its tile (32 positions x 64 channels) in the input tile's LDS,
    // reads it back position-major and moves 16 bytes per lane: residual / accumulate operands are requested as a
    // batch before the first store.
    DBG_T(3);
    __syncthreads();                                         // every wave is done with the input tile
    DBG_T(4);
    if (!wave_live) return;
    constexpr int EP = 68;                                   // fp32 pitch of the parked half tile
    float* et = reinterpret_cast<float*>(smem) + wid * 32 * EP;
    const float osc = d.out_scale;
    const int qoff = d.reflect1 ? 1 : 0;
    float* yb = d.y ? d.y + (long)b * d.y_bs : nullptr;
    bf16_t* ya = d.y_act ? d.y_act + (long)b * d.y_bs : nullptr;
    const float* rb = d.add_resid ? d.resid + (long)b * d.r_bs : nullptr;
    const int c4 = (lane & 15) * 4, rsub = lane >> 4;
    const bool col_live = (tile1_live || c4 < 32) && n32_base * 32 + c4 < ((Cout_g + 3) & ~3);
    const int co = g * Cout_g + n32_base * 32 + (col_live ? c4 : 0);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), av = make_float4(1.f, 1.f, 1.f, 1.f);
    if (d.bias) bv = *reinterpret_cast<const float4*>(d.bias + co);
    const int ca = d.alpha_mod ? co % d.alpha_mod : co;        // (a float4 of channels never straddles a phase: alpha_mod % 32 == 0)
    if (ya) av = *reinterpret_cast<const float4*>(d.alpha_out + ca);
    bf16_t* ya2 = ya && d.y_act2 ? d.y_act2 + (long)b * d.y_bs : nullptr;
    bf16_t* ya3 = ya && d.y_act3 ? d.y_act3 + (long)b * d.y_bs : nullptr;
    float4 av2 = av, av3 = av;
    if (ya2) av2 = *reinterpret_cast<const float4*>(d.alpha_out2 + ca);
    if (ya3) av3 = *reinterpret_cast<const float4*>(d.alpha_out3 + ca);
    // three instantiations: the plain one (no output activation, no reflect pad: every ResBlock conv) and the Mish one stay small --
    // the generic one carries the transcendental code of every activation kind at each of its 64 elements
    auto epilogue = [&](auto kind_tag) {
        constexpr int KIND = decltype(kind_tag)::value;      // 0 plain, 1 Mish only (the DiT position conv), 2 anything
        constexpr bool GENERIC = KIND == 2;
    #pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
    #pragma unroll
            for (int ni = 0; ni < 2; ++ni)
    #pragma unroll
                for (int r = 0; r < 16; ++r) et[((r & 3) + 8 * (r >> 2) + 4 * kh) * EP + ni * 32 + lr] = acc[mi][ni][r];
            __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): the wave reads back only its own half tile
            __builtin_amdgcn_wave_barrier();
            const int pbase = p0 + wpi * 64 + mi * 32 + rsub;
            float4 rv[8], ov[8];
    #pragma unroll
            for (int it = 0; it < 8; ++it) { rv[it] = make_float4(0.f, 0.f, 0.f, 0.f); ov[it] = rv[it]; }
            if (rb) {
    #pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const long q = min(pbase + it * 4, n_out - 1) + qoff;
                    rv[it] = *reinterpret_cast<const float4*>(rb + q * d.r_ld + co);
                }
            }
            if (d.accumulate && yb) {
    #pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const long q = min(pbase + it * 4, n_out - 1) + qoff;
                    ov[it] = *reinterpret_cast<const float4*>(yb + q * d.y_ld + co);
                }
            }
    #pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int p = pbase + it * 4;
                float4 v = *reinterpret_cast<const float4*>(et + (it * 4 + rsub) * EP + c4);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                if (GENERIC) {
                    v.x = apply_post(v.x, d.post_act, d.post_slope); v.y = apply_post(v.y, d.post_act, d.post_slope);
                    v.z = apply_post(v.z, d.post_act, d.post_slope); v.w = apply_post(v.w, d.post_act, d.post_slope);
                }
                if (KIND == 1) {
                    if (PRECISE) { v.x = act_mish(v.x); v.y = act_mish(v.y); v.z = act_mish(v.z); v.w = act_mish(v.w); }
                    else { v.x = act_mish_fast(v.x); v.y = act_mish_fast(v.y); v.z = act_mish_fast(v.z); v.w = act_mish_fast(v.w); }
                }
                if (p >= n_out || !col_live) continue;
                float4 o;
                o.x = fmaf(v.x + rv[it].x, osc, ov[it].x); o.y = fmaf(v.y + rv[it].y, osc, ov[it].y);
                o.z = fmaf(v.z + rv[it].z, osc, ov[it].z); o.w = fmaf(v.w + rv[it].w, osc, ov[it].w);
                if (yb) *reinterpret_cast<float4*>(yb + (long)(p + qoff) * d.y_ld + co) = o;
                if (ya) {                                        // what the consumer conv would stage from this value
                    auto put = [&](bf16_t* dst, const float4& a) {
                        const float s0 = snake_fast(o.x, a.x), s1 = snake_fast(o.y, a.y), s2 = snake_fast(o.z, a.z), s3 = snake_fast(o.w, a.w);
                        uint2 pk;
                        pk.x = (uint32_t)f32_to_bf16(s0) | ((uint32_t)f32_to_bf16(s1) << 16);
                        pk.y = (uint32_t)f32_to_bf16(s2) | ((uint32_t)f32_to_bf16(s3) << 16);
                        *reinterpret_cast<uint2*>(dst + (long)(p + qoff) * d.y_ld + co) = pk;
                    };
                    put(ya, av);
                    if (ya2) put(ya2, av2);
                    if (ya3) put(ya3, av3);
                }
                if (GENERIC && mi == 0 && it == 0 && d.reflect1 && p == 1) {                      // ReflectionPad1d((1,0)): row 0 mirrors conv row 1
                    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), o0 = r0;
                    if (rb) r0 = *reinterpret_cast<const float4*>(rb + co);
                    if (d.accumulate) o0 = *reinterpret_cast<const float4*>(yb + co);
                    o0.x = fmaf(v.x + r0.x, osc, o0.x); o0.y = fmaf(v.y + r0.y, osc, o0.y);
                    o0.z = fmaf(v.z + r0.z, osc, o0.z); o0.w = fmaf(v.w + r0.w, osc, o0.w);
                    *reinterpret_cast<float4*>(yb + co) = o0;
                }
            }
            __builtin_amdgcn_wave_barrier();                     // the next half overwrites the parked tile
        }
    };
    if (d.post_act == ACT_NONE && !d.reflect1) epilogue(std::integral_constant<int, 0>{});
    else if (d.post_act == ACT_MISH && !d.reflect1) epilogue(std::integral_constant<int, 1>{});
    else epilogue(std::integral_constant<int, 2>{});
    DBG_T(5);
}

template <int WP, int WC, bool PR, int NK, bool INB>
static int launch_mfma2(const ConvDesc& d, const ConvW& w, int nrows_max, size_t lds, hipStream_t st) {
    int Cout_g = d.Cout / d.groups;
    dim3 grid(cdiv(d.L_out, WP * 64), d.groups * cdiv(Cout_g, WC * 64), d.B);
    auto kern = conv1d_bf16_mfma_k<WP, WC, PR, NK, INB>;
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(WP * WC * 64), lds, st, d, w.w_mfma, nrows_max);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

template <int WP, int WC, bool PR, bool INB>
static int launch_mfma(const ConvDesc& d, const ConvW& w, hipStream_t st) {
    constexpr int TP = WP * 64;
    int nrows_max = (TP - 1 + (d.KW - 1) * d.dil) / d.up + 3;
    const int Cin_gp = ((d.Cin / d.groups + 15) / 16) * 16;
    size_t lds = (size_t)nrows_max * mf_rowb(Cin_gp) * (PR ? 2 : 1);
    lds = std::max(lds, (size_t)WP * WC * 32 * 68 * sizeof(float));     // the epilogue parks half tiles there
    FY_CHECK(lds <= 160 * 1024, FY_ERR_ARG, "conv1d_bf16_mfma: input tile needs %zu B of LDS", lds);
    ProfScope prof("conv_mfma", 2.0 * d.B * d.L_out * (double)d.Cout * (d.Cin / d.groups) * d.KW, st);
    // every staged chunk has the same number of k steps when Cin_gp <= 128 or Cin_gp is a multiple of 128
    int nk = 0;
    if (Cin_gp <= MF_CC) nk = Cin_gp / 16; else if (Cin_gp % MF_CC == 0) nk = MF_CC / 16;
    switch (nk) {
        case 2: return launch_mfma2<WP, WC, PR, 2, INB>(d, w, nrows_max, lds, st);
        case 4: return launch_mfma2<WP, WC, PR, 4, INB>(d, w, nrows_max, lds, st);
        case 6: return launch_mfma2<WP, WC, PR, 6, INB>(d, w, nrows_max, lds, st);
        case 8: return launch_mfma2<WP, WC, PR, 8, INB>(d, w, nrows_max, lds, st);
        default: return launch_mfma2<WP, WC, PR, 0, INB>(d, w, nrows_max, lds, st);
    }
}

int conv1d_bf16_mfma(const ConvDesc& d, const ConvW& w, bool precise, hipStream_t st) {
    FY_CHECK(w.w_mfma != nullptr, FY_ERR_STATE, "conv1d_bf16_mfma: weights not packed for the MFMA kernel");
    FY_CHECK(d.Cin == w.Cin && d.Cout == w.Cout && d.KW == w.KW && d.groups == w.groups, FY_ERR_ARG,
             "conv1d_bf16_mfma: descriptor (%d,%d,%d,g%d) != weights (%d,%d,%d,g%d)", d.Cin, d.Cout, d.KW, d.groups,
             w.Cin, w.Cout, w.KW, w.groups);
    FY_CHECK(d.stride == 1 && d.up >= 1 && d.dil >= 1 && d.B >= 1 && d.L_out >= 1, FY_ERR_ARG, "conv1d_bf16_mfma: bad geometry");
    int Cin_g = d.Cin / d.groups, Cout_g = d.Cout / d.groups;
    FY_CHECK((Cout_g % 32 == 0 || (d.groups == 1 && !d.y_act && d.y_ld >= ((Cout_g + 3) & ~3))) && Cin_g % 4 == 0 && d.x_ld % 4 == 0, FY_ERR_ARG,
             "conv1d_bf16_mfma: Cout/groups %% 32 (or one group with an output pitch rounded up to 4), Cin/groups %% 4 and the input pitch %% 4 must be 0");
    FY_CHECK(d.y || d.y_act, FY_ERR_ARG, "conv1d_bf16_mfma: no output");
    FY_CHECK(d.y_ld % 4 == 0 && (d.y_bs % 4) == 0, FY_ERR_ARG, "conv1d_bf16_mfma: output pitch must be a multiple of 4");
    FY_CHECK(!d.y || ((uintptr_t)d.y & 15) == 0, FY_ERR_ARG, "conv1d_bf16_mfma: output must be 16-B aligned");
    FY_CHECK(!d.y_act || (((uintptr_t)d.y_act & 7) == 0 && d.alpha_out && !d.reflect1), FY_ERR_ARG,
             "conv1d_bf16_mfma: the bf16 output stream needs 8-B alignment, alpha_out and no reflect pad");
    FY_CHECK((!d.y_act2 || (d.y_act && d.alpha_out2 && ((uintptr_t)d.y_act2 & 7) == 0)) && (!d.y_act3 || (d.y_act && d.alpha_out3 && ((uintptr_t)d.y_act3 & 7) == 0)) &&
             d.alpha_mod % 32 == 0, FY_ERR_ARG, "conv1d_bf16_mfma: further output streams need the first one, their alphas and 8-B alignment; alpha_mod %% 32 == 0");
    FY_CHECK(d.y || !(d.accumulate || d.reflect1), FY_ERR_ARG, "conv1d_bf16_mfma: accumulate / reflect need the fp32 output");
    FY_CHECK(!d.add_resid || (((uintptr_t)d.resid & 15) == 0 && (d.r_bs % 4) == 0 && d.r_ld % 4 == 0), FY_ERR_ARG,
             "conv1d_bf16_mfma: residual must be 16-B aligned");
    if (d.x_act) {
        FY_CHECK(!precise, FY_ERR_ARG, "conv1d_bf16_mfma: the bf16 input stream has no split (precise) form");
        FY_CHECK(((uintptr_t)d.x_act & 15) == 0 && d.x_bs % 8 == 0 && d.x_ld % 8 == 0 && Cin_g % 8 == 0 && d.up == 1, FY_ERR_ARG,
                 "conv1d_bf16_mfma: the bf16 input stream needs 16-B aligned rows, Cin/groups %% 8 == 0 and no upsampling");
        if (Cout_g >= 128) return launch_mfma<2, 2, false, true>(d, w, st);
        return launch_mfma<4, 1, false, true>(d, w, st);
    }
    FY_CHECK(d.x && ((uintptr_t)d.x & 15) == 0 && (d.x_bs % 4) == 0, FY_ERR_ARG, "conv1d_bf16_mfma: input must be 16-B aligned");
    if (Cout_g >= 128) {
        return precise ? launch_mfma<2, 2, true, false>(d, w, st) : launch_mfma<2, 2, false, false>(d, w, st);
    }
    return precise ? launch_mfma<4, 1, true, false>(d, w, st) : launch_mfma<4, 1, false, false>(d, w, st);
}

// =============================================================================
// One ResBlock iteration in one launch (conv.h: ResIterDesc).  WP x WC waves: WP*64 rows of conv1's output (= conv2's input,
// kept in LDS as the activated bf16 values) x WC*64 = C channels - a workgroup owns ALL channels of its rows, which is what
// lets conv2 follow without leaving the chip.  conv2 produces the last TP - (KW-1) of those rows.  The MFMA order of every
// accumulator is the one conv1d_bf16_mfma_k uses (taps outer, 16-channel steps inner), so both forms give the same bits.
// =============================================================================
template <int WP, int WC, bool INB>
__global__ __launch_bounds__(WP* WC * 64) void resblock_iter_k(ResIterDesc d, const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2) {
    constexpr int TP = WP * 64, C = WC * 64, NT = WP * WC * 64;
    // input channels are walked in chunks of <= 128 (the order conv1d_bf16_mfma_k accumulates in): NCH chunks of NK 16-channel steps
    constexpr int ROWB = C * 2 + 16, C16 = C / 16, NCH = C > 128 ? C / 128 : 1, NK = C16 / NCH, N32 = C / 32, KH = NK / 2, EP = 68;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: conv1's input rows X first; once every wave has read them, conv2's input rows Y [TP] take their place and the
    // epilogues park their tiles behind Y
    char* X = smem;
    char* Y = smem;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wpi = wid / WC, wci = wid % WC, kh = lane >> 5, lr = lane & 31;
    const int b = blockIdx.z;
    const int h2 = d.KW - 1, h1 = (d.KW - 1) * d.dil, P2 = TP - h2;
    const int p0 = blockIdx.x * P2;                          // first output row of this workgroup
    const int n = d.len ? d.len[b] : d.L;
    if (p0 >= n) return;
    const int g0 = p0 - h2 - h1;                             // global row of X row 0
    const int nrows = TP + h1, row_max = max(n - 1, 0);
    // ---- stage conv1's input ----
    if (INB) {
        constexpr int SBB = 12, q8 = C / 8;
        const int total8 = nrows * q8;
        const bf16_t* xa = d.x_act + (long)b * d.bs;
        for (int base = tid; base < total8; base += NT * SBB) {
            uint4 vv[SBB];
            int off[SBB];
#pragma unroll
            for (int u = 0; u < SBB; ++u) {
                const int idx = min(base + u * NT, total8 - 1);
                const int r = idx / q8, c8 = (idx - r * q8) * 8;
                const int row = g0 + r;
                vv[u] = *reinterpret_cast<const uint4*>(xa + (long)min(max(row, 0), row_max) * d.ld + c8);
                off[u] = (row >= 0 && row < n) ? r * ROWB + c8 * 2 : -1 - (r * ROWB + c8 * 2);
            }
#pragma unroll
            for (int u = 0; u < SBB; ++u) {
                if (base + u * NT >= total8) break;
                const bool ok = off[u] >= 0;
                *reinterpret_cast<uint4*>(X + (ok ? off[u] : -1 - off[u])) = ok ? vv[u] : make_uint4(0, 0, 0, 0);
            }
        }
    } else {
        constexpr int SB = 4, q4 = C / 4;
        const int total = nrows * q4;
        const float* xb = d.x + (long)b * d.bs;
        for (int base = tid; base < total; base += NT * SB) {
            float4 vv[SB], aa[SB];
            int rr[SB], cc[SB];
            bool okk[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = min(base + u * NT, total - 1);
                const int r = idx / q4, c4 = (idx - r * q4) * 4;
                const int row = g0 + r;
                rr[u] = r; cc[u] = c4;
                okk[u] = row >= 0 && row < n;
                vv[u] = *reinterpret_cast<const float4*>(xb + (long)min(max(row, 0), row_max) * d.ld + c4);
                aa[u] = *reinterpret_cast<const float4*>(d.alpha1 + c4);
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                if (base + u * NT >= total) break;
                float4 v = vv[u];
                const float4 a = aa[u];
                if (!okk[u]) v = make_float4(0.f, 0.f, 0.f, 0.f);
                v.x = snake_fast(v.x, a.x); v.y = snake_fast(v.y, a.y); v.z = snake_fast(v.z, a.z); v.w = snake_fast(v.w, a.w);
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
                pk.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
                *reinterpret_cast<uint2*>(X + (size_t)rr[u] * ROWB + cc[u] * 2) = pk;
            }
        }
    }
    __syncthreads();

    f32x16 acc[2][2];
    const int n32_base = wci * 2;
    // acc += sum over taps t and 16-channel steps of  src[row0 + wpi*64 + mi*32 + lr + t*dil][.] x W[t][.][this wave's 64 columns]
    auto product = [&](const char* src, const bf16_t* wp, int dil, int last_row) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        auto load_b = [&](int ch, int t, int half, frag_ab (&bb)[KH][2]) {
            const bf16_t* wt = wp + (((long)t * C16 + ch * NK + half * KH) * N32 + n32_base) * 512 + lane * 8;
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) {
                bb[kk][0] = *reinterpret_cast<const frag_ab*>(wt + (long)kk * N32 * 512);
                bb[kk][1] = *reinterpret_cast<const frag_ab*>(wt + (long)kk * N32 * 512 + 512);
            }
        };
        auto tap_half = [&](int ch, int t, int half, const frag_ab (&bb)[KH][2]) {
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const size_t off = (size_t)min(wpi * 64 + mi * 32 + lr + t * dil, last_row) * ROWB + ((ch * NK + half * KH + kk) * 16 + kh * 8) * 2;
                    const frag_ab a = *reinterpret_cast<const frag_ab*>(src + off);
                    acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb[kk][0], acc[mi][0], 0, 0, 0);
                    acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb[kk][1], acc[mi][1], 0, 0, 0);
                }
            }
        };
        frag_ab h0[KH][2], h1[KH][2];
        for (int ch = 0; ch < NCH; ++ch) {
            load_b(ch, 0, 0, h0);
            for (int t = 0; t < d.KW; ++t) {
                load_b(ch, t, 1, h1);
                tap_half(ch, t, 0, h0);
                if (t + 1 < d.KW) load_b(ch, t + 1, 0, h0);
                tap_half(ch, t, 1, h1);
            }
        }
    };
    product(X, w1, d.dil, nrows - 1);
    __syncthreads();                                         // every wave is done with X

    float* et = reinterpret_cast<float*>(smem + (size_t)TP * ROWB) + wid * 32 * EP;
    const int c4 = (lane & 15) * 4, rsub = lane >> 4;
    const int co = wci * 64 + c4;
    {   // conv1's epilogue: bias, snake(alpha2), bf16 -> Y; rows before the sequence start are conv2's zero padding
        const float4 bv = d.bias1 ? *reinterpret_cast<const float4*>(d.bias1 + co) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 av = *reinterpret_cast<const float4*>(d.alpha2 + co);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) et[((r & 3) + 8 * (r >> 2) + 4 * kh) * EP + ni * 32 + lr] = acc[mi][ni][r];
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = wpi * 64 + mi * 32 + it * 4 + rsub;
                float4 v = *reinterpret_cast<const float4*>(et + (it * 4 + rsub) * EP + c4);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                v.x = snake_fast(v.x, av.x); v.y = snake_fast(v.y, av.y); v.z = snake_fast(v.z, av.z); v.w = snake_fast(v.w, av.w);
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
                pk.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
                if (p0 - h2 + r < 0) pk = make_uint2(0, 0);
                *reinterpret_cast<uint2*>(Y + (size_t)r * ROWB + co * 2) = pk;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    product(Y, w2, 1, TP - 1);                               // rows past the tile only feed the KW - 1 masked outputs

    // conv2's epilogue (the plain form of conv1d_bf16_mfma_k's): bias, + x, scale / accumulate, fp32 out and the next conv1's stream
    float* yb = d.y ? d.y + (long)b * d.bs : nullptr;
    bf16_t* ya = d.y_act ? d.y_act + (long)b * d.bs : nullptr;
    const float* rb = d.resid + (long)b * d.bs;
    const float osc = d.out_scale;
    const float4 bv = d.bias2 ? *reinterpret_cast<const float4*>(d.bias2 + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 av = make_float4(1.f, 1.f, 1.f, 1.f);
    if (ya) av = *reinterpret_cast<const float4*>(d.alpha_out + co);
    const int n_out = min(n, p0 + P2);                       // rows this workgroup writes: [p0, n_out)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) et[((r & 3) + 8 * (r >> 2) + 4 * kh) * EP + ni * 32 + lr] = acc[mi][ni][r];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        const int pbase = p0 + wpi * 64 + mi * 32 + rsub;
        float4 rv[8], ov[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const long q = min(pbase + it * 4, n_out - 1);
            rv[it] = *reinterpret_cast<const float4*>(rb + q * d.ld + co);
            ov[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (d.accumulate && yb) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const long q = min(pbase + it * 4, n_out - 1);
                ov[it] = *reinterpret_cast<const float4*>(yb + q * d.ld + co);
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int p = pbase + it * 4;
            float4 v = *reinterpret_cast<const float4*>(et + (it * 4 + rsub) * EP + c4);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (p >= n_out) continue;
            float4 o;
            o.x = fmaf(v.x + rv[it].x, osc, ov[it].x); o.y = fmaf(v.y + rv[it].y, osc, ov[it].y);
            o.z = fmaf(v.z + rv[it].z, osc, ov[it].z); o.w = fmaf(v.w + rv[it].w, osc, ov[it].w);
            if (yb) *reinterpret_cast<float4*>(yb + (long)p * d.ld + co) = o;
            if (ya) {
                o.x = snake_fast(o.x, av.x); o.y = snake_fast(o.y, av.y); o.z = snake_fast(o.z, av.z); o.w = snake_fast(o.w, av.w);
                uint2 pk;
                pk.x = (uint32_t)f32_to_bf16(o.x) | ((uint32_t)f32_to_bf16(o.y) << 16);
                pk.y = (uint32_t)f32_to_bf16(o.z) | ((uint32_t)f32_to_bf16(o.w) << 16);
                *reinterpret_cast<uint2*>(ya + (long)p * d.ld + co) = pk;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

bool conv_resblock_iter_supported(int C, int KW, int dil) {
    if (C != 64 && C != 128 && C != 256) return false;
    // 256 channels: a 64-row tile is what fits two workgroups per CU, and its halo (114 input rows and 64 conv1 rows for 54
    // outputs) costs more than the fusion saves: 361 against 357 ms at BASELINE config 5 - built, bit-identical, off by default
    static const bool fuse256 = getenv("FY_HIFT_FUSE256") && atoi(getenv("FY_HIFT_FUSE256")) != 0;
    if (C == 256 && !fuse256) return false;
    const int TP = C == 64 ? 256 : C == 128 ? 128 : 64, h1 = (KW - 1) * dil, h2 = KW - 1;
    return KW >= 1 && dil >= 1 && h2 < TP / 2 && (size_t)(TP + h1) * (C * 2 + 16) <= 80 * 1024;
}

template <int WP, int WC, bool INB>
static int launch_resblock_iter(const ResIterDesc& d, const ConvW& w1, const ConvW& w2, hipStream_t st) {
    constexpr int TP = WP * 64, C = WC * 64, ROWB = C * 2 + 16;
    const int h1 = (d.KW - 1) * d.dil, h2 = d.KW - 1, P2 = TP - h2;
    // X, then Y in its place with the parked half tiles behind it: <= 80 KB, so two workgroups share a CU and one's staging and
    // epilogues run under the other's MFMAs
    const size_t lds = std::max((size_t)(TP + h1) * ROWB, (size_t)TP * ROWB + (size_t)WP * WC * 32 * 68 * sizeof(float));
    FY_CHECK(lds <= 160 * 1024, FY_ERR_ARG, "conv_resblock_iter: needs %zu B of LDS", lds);
    auto kern = resblock_iter_k<WP, WC, INB>;
    static PerDeviceOnce attr_once;                                  // per device, not per process
    const int dslot = current_device_slot();
    if (!attr_once.done[dslot].load(std::memory_order_acquire)) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.done[dslot].store(true, std::memory_order_release);
    }
    ProfScope prof("conv_mfma", 2.0 * 2.0 * d.B * d.L * (double)C * C * d.KW, st);
    dim3 grid(cdiv(d.L, P2), 1, d.B);
    hipLaunchKernelGGL(kern, grid, dim3(WP * WC * 64), lds, st, d, w1.w_mfma, w2.w_mfma);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

int conv_resblock_iter(const ResIterDesc& d, const ConvW& w1, const ConvW& w2, hipStream_t st) {
    FY_CHECK(conv_resblock_iter_supported(d.C, d.KW, d.dil), FY_ERR_ARG, "conv_resblock_iter: C %d, k %d, dilation %d not supported", d.C, d.KW, d.dil);
    FY_CHECK(w1.w_mfma && w2.w_mfma && w1.Cin == d.C && w1.Cout == d.C && w2.Cin == d.C && w2.Cout == d.C && w1.KW == d.KW && w2.KW == d.KW &&
             w1.groups == 1 && w2.groups == 1, FY_ERR_ARG, "conv_resblock_iter: weights do not match the descriptor");
    FY_CHECK((d.x_act || (d.x && d.alpha1)) && d.resid && d.alpha2 && (d.y || d.y_act) && (!d.y_act || d.alpha_out) && (d.y || !d.accumulate),
             FY_ERR_ARG, "conv_resblock_iter: bad descriptor");
    FY_CHECK(d.ld == d.C && d.bs % 8 == 0 && d.B >= 1 && d.L >= 1, FY_ERR_ARG, "conv_resblock_iter: tensors must be dense channels-last");
    FY_CHECK((((uintptr_t)d.x | (uintptr_t)d.x_act | (uintptr_t)d.resid | (uintptr_t)d.y | (uintptr_t)d.y_act) & 15) == 0, FY_ERR_ARG,
             "conv_resblock_iter: tensors must be 16-B aligned");
    if (d.C == 64) return d.x_act ? launch_resblock_iter<4, 1, true>(d, w1, w2, st) : launch_resblock_iter<4, 1, false>(d, w1, w2, st);
    if (d.C == 256) return d.x_act ? launch_resblock_iter<1, 4, true>(d, w1, w2, st) : launch_resblock_iter<1, 4, false>(d, w1, w2, st);
    static const bool wide = getenv("FY_HIFT_FUSE_WIDE") && atoi(getenv("FY_HIFT_FUSE_WIDE"));      // experiment: 256-row tiles, one workgroup per CU
    if (wide) return d.x_act ? launch_resblock_iter<4, 2, true>(d, w1, w2, st) : launch_resblock_iter<4, 2, false>(d, w1, w2, st);
    return d.x_act ? launch_resblock_iter<2, 2, true>(d, w1, w2, st) : launch_resblock_iter<2, 2, false>(d, w1, w2, st);
}

// =============================================================================
// fp32 MFMA implicit-GEMM kernel (v_mfma_f32_32x32x2_f32): exact fp32 products and sums on the matrix cores, for the
// layers that must stay fp32 (the f0 predictor: its output is integrated into the harmonic source's phase).
//   block = 2 x 2 waves, each a 64-position x 64-channel tile; the input tile (32 channels at a time) sits in LDS
//   as fp32, weights ([KW][Cin][Cout_pad] fp32, the direct kernel's layout) stream from L2 one tap ahead.
//   stride 1, no up-sampling, one group.
// =============================================================================
#define FM_TP 128
#define FM_TC 128
#define FM_CC 32
#define FM_PITCH 33

template <int POST>
__global__ __launch_bounds__(256) void conv1d_f32_mfma_k(ConvDesc d, const float* __restrict__ w, int cout_pad) {
    extern __shared__ __attribute__((aligned(16))) float fxs[];      // [FM_TP + (KW-1) dil][FM_PITCH]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wpi = wid >> 1, wci = wid & 1, lr = lane & 31, kh = lane >> 5;
    const int b = blockIdx.z, p0 = blockIdx.x * FM_TP, co0 = blockIdx.y * FM_TC + wci * 64;
    const int n_out = d.out_len ? d.out_len[b] : d.L_out;
    if (p0 >= n_out) return;
    const int n_in = d.in_len ? d.in_len[b] : d.L_in;
    const int row_lo = p0 - d.pad_left, nrows = FM_TP + (d.KW - 1) * d.dil;
    const float* xb = d.x + (long)b * d.x_bs;
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const int cob[2] = {min(co0 + lr, d.Cout - 1), min(co0 + 32 + lr, d.Cout - 1)};
    for (int ci0 = 0; ci0 < d.Cin; ci0 += FM_CC) {
        __syncthreads();
        {   // clamped addresses + selects: the loads of a batch are all in flight before the first LDS write
            constexpr int SB = 5;
            const int total = nrows * (FM_CC / 4), row_max = max(n_in - 1, 0), ci_max = max(d.Cin - 4, 0);
            for (int base = tid; base < total; base += 256 * SB) {
                float4 vv[SB];
                int off[SB];
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int idx = min(base + u * 256, total - 1);
                    const int r = idx >> 3, c4 = (idx & 7) * 4, row = row_lo + r, ci = ci0 + c4;
                    vv[u] = *reinterpret_cast<const float4*>(xb + (long)min(max(row, 0), row_max) * d.x_ld + min(ci, ci_max));
                    off[u] = (row >= 0 && row < n_in && ci < d.Cin) ? r * FM_PITCH + c4 : -1 - (r * FM_PITCH + c4);
                }
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    if (base + u * 256 >= total) break;
                    const bool ok = off[u] >= 0;
                    float* dst = fxs + (ok ? off[u] : -1 - off[u]);
                    dst[0] = ok ? apply_pre(vv[u].x, d.pre_act, d.pre_slope, 0.f) : 0.f;
                    dst[1] = ok ? apply_pre(vv[u].y, d.pre_act, d.pre_slope, 0.f) : 0.f;
                    dst[2] = ok ? apply_pre(vv[u].z, d.pre_act, d.pre_slope, 0.f) : 0.f;
                    dst[3] = ok ? apply_pre(vv[u].w, d.pre_act, d.pre_slope, 0.f) : 0.f;
                }
            }
        }
        __syncthreads();
        float b0[FM_CC / 2][2], b1[FM_CC / 2][2];
        auto load_w = [&](int t, float (&bb)[FM_CC / 2][2]) {
#pragma unroll
            for (int kk = 0; kk < FM_CC / 2; ++kk) {
                const int ci = ci0 + 2 * kk + kh;
                const float* wr = w + ((long)t * d.Cin + min(ci, d.Cin - 1)) * cout_pad;
                // no guards: channels past Cin meet zero activations in LDS, columns past Cout are never stored
                bb[kk][0] = wr[cob[0]];
                bb[kk][1] = wr[cob[1]];
            }
        };
        auto tap = [&](int t, const float (&bb)[FM_CC / 2][2]) {
            const float* x0 = fxs + (wpi * 64 + lr + t * d.dil) * FM_PITCH + kh;
#pragma unroll
            for (int kk = 0; kk < FM_CC / 2; ++kk) {
                const float a0 = x0[2 * kk], a1 = x0[32 * FM_PITCH + 2 * kk];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bb[kk][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bb[kk][1], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bb[kk][0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bb[kk][1], acc[1][1], 0, 0, 0);
            }
        };
        load_w(0, b0);
        for (int t = 0; t < d.KW; t += 2) {
            load_w(min(t + 1, d.KW - 1), b1);
            tap(t, b0);
            load_w(min(t + 2, d.KW - 1), b0);
            if (t + 1 < d.KW) tap(t + 1, b1);
        }
    }
    float* yb = d.y + (long)b * d.y_bs;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int co = co0 + ni * 32 + lr;
        if (co >= d.Cout) continue;
        const float bv = d.bias ? d.bias[co] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + wpi * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (p >= n_out) continue;
                float v = acc[mi][ni][r] + bv;
                if (POST == ACT_ELU) v = act_elu(v);
                if (POST == ACT_LEAKY) v = act_leaky(v, d.post_slope);
                if (d.add_resid) v += d.resid[(long)b * d.r_bs + (long)p * d.r_ld + co];
                yb[(long)p * d.y_ld + co] = v;
            }
    }
}

// Short sequences (a few hundred frames: the benchmark's 3 s utterances, streaming chunks) give the kernel above a few dozen
// workgroups that each walk all Cin x KW serially.  This form spends the four waves of a workgroup on ONE 64 x 64 output
// tile instead: wave w takes the channel chunks w, w+4, ... (its own LDS tile, no block barrier inside the loop) and the
// four partial sums are added in wave order at the end - 4x more workgroups, a 4x shorter chain, still deterministic.
template <int POST>
__global__ __launch_bounds__(256) void conv1d_f32_mfma_ks_k(ConvDesc d, const float* __restrict__ w, int cout_pad, int tile_floats) {
    extern __shared__ __attribute__((aligned(16))) float fks[];      // 4 x [64 + (KW-1) dil][FM_PITCH], then 3 x [64][64] partial sums
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int lr = lane & 31, kh = lane >> 5;
    const int b = blockIdx.z, p0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
    const int n_out = d.out_len ? d.out_len[b] : d.L_out;
    if (p0 >= n_out) return;
    const int n_in = d.in_len ? d.in_len[b] : d.L_in;
    const int row_lo = p0 - d.pad_left, nrows = 64 + (d.KW - 1) * d.dil;
    const float* xb = d.x + (long)b * d.x_bs;
    float* xs = fks + wid * tile_floats;
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    const int cob[2] = {min(co0 + lr, d.Cout - 1), min(co0 + 32 + lr, d.Cout - 1)};
    const int row_max = max(n_in - 1, 0), ci_max = max(d.Cin - 4, 0);
    for (int ci0 = wid * FM_CC; ci0 < d.Cin; ci0 += 4 * FM_CC) {
        __builtin_amdgcn_wave_barrier();                     // this wave is done reading its previous tile
        constexpr int SB = 5;
        const int total = nrows * (FM_CC / 4);
        for (int base = lane; base < total; base += 64 * SB) {
            float4 vv[SB];
            int off[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int idx = min(base + u * 64, total - 1);
                const int r = idx >> 3, c4 = (idx & 7) * 4, row = row_lo + r, ci = ci0 + c4;
                vv[u] = *reinterpret_cast<const float4*>(xb + (long)min(max(row, 0), row_max) * d.x_ld + min(ci, ci_max));
                off[u] = (row >= 0 && row < n_in && ci < d.Cin) ? r * FM_PITCH + c4 : -1 - (r * FM_PITCH + c4);
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                if (base + u * 64 >= total) break;
                const bool ok = off[u] >= 0;
                float* dst = xs + (ok ? off[u] : -1 - off[u]);
                dst[0] = ok ? apply_pre(vv[u].x, d.pre_act, d.pre_slope, 0.f) : 0.f;
                dst[1] = ok ? apply_pre(vv[u].y, d.pre_act, d.pre_slope, 0.f) : 0.f;
                dst[2] = ok ? apply_pre(vv[u].z, d.pre_act, d.pre_slope, 0.f) : 0.f;
                dst[3] = ok ? apply_pre(vv[u].w, d.pre_act, d.pre_slope, 0.f) : 0.f;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): the tile is wave-private
        __builtin_amdgcn_wave_barrier();
        for (int t = 0; t < d.KW; ++t) {
            float bb[FM_CC / 2][2];
#pragma unroll
            for (int kk = 0; kk < FM_CC / 2; ++kk) {
                const float* wr = w + ((long)t * d.Cin + min(ci0 + 2 * kk + kh, d.Cin - 1)) * cout_pad;
                bb[kk][0] = wr[cob[0]];
                bb[kk][1] = wr[cob[1]];
            }
            const float* x0 = xs + (lr + t * d.dil) * FM_PITCH + kh;
#pragma unroll
            for (int kk = 0; kk < FM_CC / 2; ++kk) {
                const float a0 = x0[2 * kk], a1 = x0[32 * FM_PITCH + 2 * kk];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bb[kk][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bb[kk][1], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bb[kk][0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bb[kk][1], acc[1][1], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                         // every wave is done with its input tile
    float* ps = fks;                                         // [3][64 lanes][64]
    if (wid > 0) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) ps[((wid - 1) * 64 + (mi * 2 + ni) * 16 + r) * 64 + lane] = acc[mi][ni][r];
    }
    __syncthreads();
    if (wid != 0) return;
#pragma unroll
    for (int ww = 0; ww < 3; ++ww)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] += ps[(ww * 64 + (mi * 2 + ni) * 16 + r) * 64 + lane];
    float* yb = d.y + (long)b * d.y_bs;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int co = co0 + ni * 32 + lr;
        if (co >= d.Cout) continue;
        const float bv = d.bias ? d.bias[co] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (p >= n_out) continue;
                float v = acc[mi][ni][r] + bv;
                if (POST == ACT_ELU) v = act_elu(v);
                if (POST == ACT_LEAKY) v = act_leaky(v, d.post_slope);
                if (d.add_resid) v += d.resid[(long)b * d.r_bs + (long)p * d.r_ld + co];
                yb[(long)p * d.y_ld + co] = v;
            }
    }
}

template <int POST>
static void launch_f32_mfma(const ConvDesc& d, const ConvW& w, hipStream_t st) {
    // the choice depends on the sequence length only - never on the batch - so a batch and its utterances alone agree bit for bit
    if (d.L_out <= 4096) {
        const int tile_floats = (64 + (d.KW - 1) * d.dil) * FM_PITCH;
        const size_t lds = std::max((size_t)4 * tile_floats, (size_t)3 * 64 * 64) * sizeof(float);
        dim3 grid(cdiv(d.L_out, 64), cdiv(d.Cout, 64), d.B);
        hipLaunchKernelGGL(conv1d_f32_mfma_ks_k<POST>, grid, dim3(256), lds, st, d, w.w_dir, w.cout_pad4(), tile_floats);
    } else {
        const size_t lds = (size_t)(FM_TP + (d.KW - 1) * d.dil) * FM_PITCH * sizeof(float);
        dim3 grid(cdiv(d.L_out, FM_TP), cdiv(d.Cout, FM_TC), d.B);
        hipLaunchKernelGGL(conv1d_f32_mfma_k<POST>, grid, dim3(256), lds, st, d, w.w_dir, w.cout_pad4());
    }
}

int conv1d_f32_mfma(const ConvDesc& d, const ConvW& w, hipStream_t st) {
    FY_CHECK(w.w_dir != nullptr, FY_ERR_STATE, "conv1d_f32_mfma: weights not packed in the fp32 layout");
    FY_CHECK(d.Cin == w.Cin && d.Cout == w.Cout && d.KW == w.KW && d.groups == 1 && w.groups == 1, FY_ERR_ARG, "conv1d_f32_mfma: descriptor != weights");
    FY_CHECK(d.stride == 1 && d.up == 1 && d.dil >= 1 && d.B >= 1 && d.L_out >= 1 && !d.accumulate && !d.reflect1 && d.out_scale == 1.f &&
             d.pre_act != ACT_SNAKE && (d.post_act == ACT_NONE || d.post_act == ACT_ELU || d.post_act == ACT_LEAKY) && d.Cin % 4 == 0 && d.x_ld % 4 == 0 && d.x_bs % 4 == 0 &&
             ((uintptr_t)d.x & 15) == 0, FY_ERR_ARG, "conv1d_f32_mfma: unsupported fusion / geometry / alignment");
    const size_t lds = (size_t)(FM_TP + (d.KW - 1) * d.dil) * FM_PITCH * sizeof(float);
    FY_CHECK(lds <= 64 * 1024, FY_ERR_ARG, "conv1d_f32_mfma: input tile needs %zu B of LDS", lds);
    if (d.post_act == ACT_ELU) launch_f32_mfma<ACT_ELU>(d, w, st);
    else if (d.post_act == ACT_LEAKY) launch_f32_mfma<ACT_LEAKY>(d, w, st);
    else launch_f32_mfma<ACT_NONE>(d, w, st);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// =============================================================================
// weight packing (load time)
// =============================================================================
__global__ void wn_scale_k(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ scale, int per_co) {
    // scale[co] = g[co] / ||v[co]||  (torch._weight_norm over dims != 0); 1 when g is null
    __shared__ double red[256];
    int co = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < per_co; i += 256) {
        double x = v[(long)co * per_co + i];
        s += x * x;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) scale[co] = g ? (float)((double)g[co] / sqrt(red[0])) : 1.0f;
}

__global__ void pack_dir_k(const float* __restrict__ v, const float* __restrict__ scale, float* __restrict__ out,
                           int Cout, int Cin_g, int KW, int cout_pad) {
    long n = (long)KW * Cin_g * cout_pad;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        int co = i % cout_pad;
        long r = i / cout_pad;
        int ci = r % Cin_g, t = r / Cin_g;
        out[i] = co < Cout ? v[((long)co * Cin_g + ci) * KW + t] * scale[co] : 0.f;
    }
}

__global__ void pack_mfma_k(const float* __restrict__ v, const float* __restrict__ scale, bf16_t* __restrict__ out,
                            int Cout, int Cin_g, int KW, int groups) {
    const int Cout_g = Cout / groups, Cin_gp = ((Cin_g + 15) / 16) * 16, C16 = Cin_gp / 16, N32 = (Cout_g + 31) / 32;
    long n = (long)groups * KW * C16 * N32 * 512;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        int j = i & 7, l = (i >> 3) & 63;
        long r = i >> 9;
        int n32 = r % N32; r /= N32;
        int c16 = r % C16; r /= C16;
        int t = r % KW;
        int g = r / KW;
        int ci = c16 * 16 + (l >> 5) * 8 + j;
        int co = g * Cout_g + n32 * 32 + (l & 31);
        float x = (ci < Cin_g && n32 * 32 + (l & 31) < Cout_g) ? v[((long)co * Cin_g + ci) * KW + t] * scale[co] : 0.f;
        out[i] = f32_to_bf16(x);
    }
}

int conv_pack(ConvW& cw, const float* v, const float* g, const float* bias, int Cout, int Cin, int KW, int groups,
              bool want_direct, bool want_mfma, hipStream_t st) {
    FY_CHECK(Cout > 0 && Cin > 0 && KW > 0 && groups > 0 && Cin % groups == 0 && Cout % groups == 0, FY_ERR_ARG,
             "conv_pack: bad shape (%d,%d,%d,g%d)", Cout, Cin, KW, groups);
    cw.Cin = Cin; cw.Cout = Cout; cw.KW = KW; cw.groups = groups;
    const int Cin_g = Cin / groups;
    float* scale = nullptr;
    HIP_TRY(hipMalloc(&scale, Cout * sizeof(float)));
    hipLaunchKernelGGL(wn_scale_k, dim3(Cout), dim3(256), 0, st, v, g, scale, Cin_g * KW);
    if (bias) {                                              // padded: the MFMA epilogue reads whole 32-column tiles
        const size_t nb = ((size_t)Cout + 31) / 32 * 32;
        HIP_TRY(hipMalloc(&cw.bias, nb * sizeof(float)));
        HIP_TRY(hipMemsetAsync(cw.bias, 0, nb * sizeof(float), st));
        HIP_TRY(hipMemcpyAsync(cw.bias, bias, Cout * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    if (want_direct) {
        long n = (long)KW * Cin_g * cw.cout_pad4();
        HIP_TRY(hipMalloc(&cw.w_dir, n * sizeof(float)));
        hipLaunchKernelGGL(pack_dir_k, dim3((int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, st, v, scale,
                           cw.w_dir, Cout, Cin_g, KW, cw.cout_pad4());
    }
    if (want_mfma) {
        FY_CHECK((Cout / groups) % 32 == 0 || groups == 1, FY_ERR_ARG,
                 "conv_pack: MFMA layout needs Cout/groups %% 32 == 0 when grouped (got %d)", Cout / groups);
        long n = (long)groups * KW * (cw.cin_g_pad() / 16) * ((Cout / groups + 31) / 32) * 512;
        HIP_TRY(hipMalloc(&cw.w_mfma, n * sizeof(bf16_t)));
        hipLaunchKernelGGL(pack_mfma_k, dim3((int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, st, v, scale,
                           cw.w_mfma, Cout, Cin_g, KW, groups);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipFree(scale));
    return FY_OK;
}

// Nearest-repeat up-sampling (x u) followed by a causal k-tap conv, as ONE stride-1 conv on the un-repeated input
// with u*Cout output channels: output row u n + phi reads repeated rows u n + phi + t - (k-1), i.e. input rows
// n + floor((phi + t - k + 1) / u) -- only K' = ceil((k-1)/u) + 1 distinct ones -- so the taps that land on one
// input row are summed beforehand: W'[phi Cout + co][ci][t'] = sum_{t : floor((phi+t-k+1)/u) = t'-(K'-1)} W[co][ci][t].
// The result rows [n][phi Cout + co] ARE the channels-last up-sampled tensor [u n + phi][co].  k/K' fewer MFMAs.
__global__ void pack_mfma_poly_k(const float* __restrict__ v, const float* __restrict__ scale, bf16_t* __restrict__ out,
                                 int Cout, int Cin, int KW, int up, int K2) {
    const int Cout2 = Cout * up, Cin_p = ((Cin + 15) / 16) * 16, C16 = Cin_p / 16, N32 = (Cout2 + 31) / 32;
    long n = (long)K2 * C16 * N32 * 512;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        int j = i & 7, l = (i >> 3) & 63;
        long r = i >> 9;
        int n32 = r % N32; r /= N32;
        int c16 = r % C16;
        int t2 = r / C16;
        int ci = c16 * 16 + (l >> 5) * 8 + j;
        int co2 = n32 * 32 + (l & 31);
        float x = 0.f;
        if (ci < Cin && co2 < Cout2) {
            const int phi = co2 / Cout, co = co2 % Cout, want = t2 - (K2 - 1);
            for (int t = 0; t < KW; ++t) {
                const int a = phi + t - KW + 1;
                const int fl = a >= 0 ? a / up : -((-a + up - 1) / up);
                if (fl == want) x += v[((long)co * Cin + ci) * KW + t] * scale[co];
            }
        }
        out[i] = f32_to_bf16(x);
    }
}

int conv_pack_polyphase(ConvW& cw, const float* v, const float* g, const float* bias, int Cout, int Cin, int KW, int up, hipStream_t st) {
    FY_CHECK(Cout > 0 && Cin > 0 && KW > 0 && up > 1 && (Cout * up) % 32 == 0, FY_ERR_ARG, "conv_pack_polyphase: bad shape (%d,%d,%d,x%d)", Cout, Cin, KW, up);
    const int K2 = (KW - 1 + up - 1) / up + 1, Cout2 = Cout * up;
    cw.Cin = Cin; cw.Cout = Cout2; cw.KW = K2; cw.groups = 1;
    float* scale = nullptr;
    HIP_TRY(hipMalloc(&scale, Cout * sizeof(float)));
    hipLaunchKernelGGL(wn_scale_k, dim3(Cout), dim3(256), 0, st, v, g, scale, Cin * KW);
    if (bias) {
        HIP_TRY(hipMalloc(&cw.bias, (size_t)Cout2 * sizeof(float)));
        for (int p = 0; p < up; ++p)
            HIP_TRY(hipMemcpyAsync(cw.bias + (size_t)p * Cout, bias, Cout * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    long n = (long)K2 * (cw.cin_g_pad() / 16) * (Cout2 / 32) * 512;
    HIP_TRY(hipMalloc(&cw.w_mfma, n * sizeof(bf16_t)));
    hipLaunchKernelGGL(pack_mfma_poly_k, dim3((int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, st, v, scale, cw.w_mfma,
                       Cout, Cin, KW, up, K2);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipFree(scale));
    return FY_OK;
}

void conv_free(ConvW& cw) {
    if (cw.w_dir) (void)hipFree(cw.w_dir);
    if (cw.w_mfma) (void)hipFree(cw.w_mfma);
    if (cw.bias) (void)hipFree(cw.bias);
    cw = ConvW();
}
