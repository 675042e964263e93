// Dense products for the DiT estimator and the Qwen2 speech-token LM.
//   gemm_bf16   : C[M,N] = A[M,K] * W[N,K]^T on v_mfma_f32_32x32x16_bf16, A and W both K-contiguous
//                 (torch Linear layout), fused epilogues; 256x128x32 tiles fed by an LDS-DMA ring, two workgroups
//                 per CU, for grids that fill the chip, 128x128x64 register-staged tiles otherwise (gemm.hip says why).
//   gemv_bf16w  : y[r,n] = sum_k W[n,k] x[r,k] for r <= 8 rows of fp32 activations against bf16 weights,
//                 fp32-faithful on the matrix cores (exact 3-way bf16 split of the activations) - the
//                 LLM decode product.
#pragma once
#include "common.h"

enum { EPI_STORE = 0, EPI_GATE_RESID = 1 };

struct GemmEpi {
    int mode = EPI_STORE;
    const float* bias = nullptr;       // [N]
    int act = ACT_NONE;                // applied to acc + bias (EPI_STORE)
    void* out = nullptr;               // EPI_STORE: bf16 or fp32 [M][ldc]
    int out_bf16 = 1;
    int ldc = 0;
    // EPI_GATE_RESID: resid[m][n] += gate[n] * (acc + bias[n]) for rows with row_ok[m] != 0 (null = all)
    float* resid = nullptr;
    const float* gate = nullptr;
    // EPI_STORE to bf16, optional: rotary embedding of the interleaved pairs of columns [0, 2*rope_half) and
    // [rope_stride, rope_stride + 2*rope_half) at position (row % rope_T), table [rope_T][rope_half] of (cos, sin)
    const float2* rope = nullptr;
    int rope_T = 0, rope_half = 0, rope_stride = 0;
    int rope_div = 1;                  // position of output row m = (m / rope_div) % rope_T (2: the two sequences' rows interleaved)
    // split-operand (fp32-class) products, gemm_split: the A operand arrives as TWO bf16 planes, x = hi + lo with hi = bf16(x) and
    // lo = bf16(x - hi) - a_lo is the lo plane ([M][lda] like A); out_lo != null stores the result the same way (out = hi plane,
    // out_lo = lo plane, both bf16 [M][ldc]) for the next product, after the exact-tanh GELU (act) or the rotary embedding (rope)
    const bf16_t* a_lo = nullptr;
    const bf16_t* a_lo2 = nullptr;     // the third plane of an EXACT three-way split x = hi + mid (a_lo) + lo (a_lo2): gemm_exact3
    // gemm_split only: the lo plane of the WEIGHTS, W_true = W + w_lo with W = bf16(w), w_lo = bf16(w - W), [N][K] like W (general
    // fp32 checkpoints; null = the weights are bf16-representable).  The K loop then runs a second pass over w_lo.
    const bf16_t* w_lo = nullptr;
    // LayerNorm-modulate folded across a product (flow decoder, default mode; flow.hip says when).  The producer of the residual
    // stream (EPI_GATE_RESID, ring kernels) also leaves bf16(new resid) in h_bf16 [M][ldc] and, per (row, 64-column tile), the
    // (sum, sum of squares) of the new values in ln_slots [M][N / 64].  The consumer (EPI_STORE, bf16 output) multiplies bf16(resid)
    // itself by W' = W (1 + scale) and finishes out = rstd (acc - mean ln_u[n]) + bias[n]; every workgroup first turns the slots of
    // ITS rows (ln_rows_slots [M][ln_dim / 64], summed in slot order, in double) into (mean, rstd) in LDS - no launch and no
    // hand-off between workgroups for that (ln_u[n] = sum_k W'[n][k]; `bias` then carries bias + W shift; ln_dim = the row length
    // the statistics are of = this product's K).
    bf16_t* h_bf16 = nullptr;
    float2* ln_slots = nullptr;
    const float2* ln_rows_slots = nullptr;
    int ln_dim = 0;
    const float* ln_u = nullptr;
    const float* gate_ones = nullptr;  // N ones: needed with w_lo only where gemm_split falls back to the register-staged kernel (N % 128 != 0)
    void* out_lo = nullptr;
#ifdef FY_GEMM_STAMPS
    int stamp_slot = 0;                // tests/micro/gemm_stamps.hip: which 4096-workgroup slot of the stamp buffer this launch writes
#endif
};
#ifdef FY_GEMM_STAMPS
void gemm_set_stamps(unsigned long long* p);
#endif

extern int gemm_tile_override;     // microbenchmarks only: 0 = automatic choice (gemm.hip lists the forms)
// A: bf16 [M][lda]; W: bf16 [N][K] (K % 64 == 0); M, N arbitrary
int gemm_bf16(const bf16_t* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st);
// A: fp32 [M][lda], split into bf16 hi + lo on the fly (fp32-class accuracy when W is bf16-exact)
int gemm_f32a_precise(const float* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st);

// A = A_hi + epi.a_lo, two bf16 planes [M][lda] written by the producer (16 mantissa bits): two MFMAs per fragment into one fp32
// accumulator on the LDS-DMA ring kernel - the fp32-class (FY_PRECISE) flow decoder at 2x the MFMA work of gemm_bf16, not the
// register-staged fp32 path's 5x.  Outputs: fp32 store, gated residual, or split bf16 planes (epi.out_lo).  With epi.w_lo the
// weights are two planes as well (4x the MFMA work).
int gemm_split(const bf16_t* A_hi, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st);

// A: fp32 [M][lda], split EXACTLY into bf16 hi + mid + lo (3 MFMAs per fragment): with bf16-exact W every product is exact and
// the accumulation is fp32 - the LM prefill over many rows (same fidelity as gemv_bf16w, whose 8-row form re-streams the weights)
int gemm_f32a_exact(const float* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st);

// The same product with A already split: A_hi + epi.a_lo + epi.a_lo2, three bf16 planes [M][lda] (split3_planes), on the LDS-DMA
// ring kernel - three A tiles per stage, three MFMAs per fragment pair into one accumulator (N % 128 == 0, K % 32 == 0).
int gemm_exact3(const bf16_t* A_hi, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st);
bool gemm_exact3_supported(int N, int K);
// fp32 [rows][ld_src] (cols used) -> three bf16 planes [rows][cols]: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)
int split3_planes(const float* src, int ld_src, int rows, int cols, bf16_t* hi, bf16_t* mid, bf16_t* lo, hipStream_t st);

// fp32 [n] -> bf16 [n]
int cast_f32_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t st);

// ---- decode-side products -----------------------------------------------------------
enum { GV_STORE = 0, GV_ADD = 1, GV_SWIGLU = 2, GV_SWIGLU_SPLIT = 3 };
struct GemvArgs {
    const bf16_t* W = nullptr;   // packed by gemv_pack (B-fragment order of v_mfma_f32_32x32x16_bf16)
    const float* x = nullptr;    // [R][ldx] fp32, K <= 1024 (split in LDS) ...
    const bf16_t* x_split = nullptr; // ... or pre-split bf16 [ceil(R/8)][24][ldx] (hi / mid / lo planes of 8 rows), any K
    int ldx = 0;
    int R = 0, N = 0, K = 0;
    const float* bias = nullptr; // [N]
    float* y = nullptr;          // [R][ldy]
    bf16_t* y_split = nullptr;   // GV_SWIGLU_SPLIT: [ceil(R/8)][24][ldy] planes for a following x_split product
    int ldy = 0;
    int mode = GV_STORE;         // GV_ADD: y += ; GV_SWIGLU: rows interleaved (gate_i, up_i) -> y[r][i] = silu(g)*u, N counts both
    const float* norm_w = nullptr; // fused RMSNorm: y = W (norm_w * x) * rsqrt(mean(x^2) + eps), K <= 1024
    float eps = 0.f;
    float* partial = nullptr;    // split-K workspace of the pre-split form (gemv_partial_floats), with ...
    int* counters = nullptr;     // ... zero-initialised arrival counters (gemv_counter_ints); both optional
};
int gemv_bf16w(const GemvArgs& a, hipStream_t st);
// fp32 [N][K] row-major (torch Linear layout) -> bf16 fragment order; dst holds gemv_packed_elems(N, K) elements
size_t gemv_packed_elems(int N, int K);
int gemv_pack(const float* src, bf16_t* dst, int N, int K, hipStream_t st);
size_t gemv_partial_floats(int R, int N, int K);
size_t gemv_counter_ints(int R, int N, int K);
