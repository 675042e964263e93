// Dense products for the DiT estimator and the Qwen2 speech-token LM.
//   gemm_bf16   : C[M,N] = A[M,K] * W[N,K]^T on v_mfma_f32_32x32x16_bf16, 128x128x32 block tile,
//                 A and W both K-contiguous (torch Linear layout), fused epilogues.
//   gemv_bf16w  : y[r,n] = sum_k W[n,k] x[r,k] for r < 8 rows of fp32 activations against bf16
//                 weights in fp32 VALU arithmetic - the HBM-bound LLM decode product.
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
};

// A: bf16 [M][lda]; W: bf16 [N][K] (K % 32 == 0); M, N arbitrary
int gemm_bf16(const bf16_t* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st);
// A: fp32 [M][lda], split into bf16 hi + lo on the fly (fp32-class accuracy when W is bf16-exact)
int gemm_f32a_precise(const float* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st);

// fp32 [n] -> bf16 [n]
int cast_f32_bf16(const float* src, bf16_t* dst, size_t n, hipStream_t st);

// ---- decode-side products -----------------------------------------------------------
enum { GV_STORE = 0, GV_ADD = 1, GV_SWIGLU = 2 };
struct GemvArgs {
    const bf16_t* W = nullptr;   // [N][K] row-major
    const float* x = nullptr;    // [R][ldx]
    int ldx = 0;
    int R = 0, N = 0, K = 0;
    const float* bias = nullptr; // [N]
    float* y = nullptr;          // [R][ldy]
    int ldy = 0;
    int mode = GV_STORE;         // GV_ADD: y += ; GV_SWIGLU: rows interleaved (gate_i, up_i) -> y[r][i] = silu(g)*u, N counts both
    float* partial = nullptr;    // workspace [ksplit][R][N] when K is split
};
int gemv_bf16w(const GemvArgs& a, hipStream_t st);
size_t gemv_partial_floats(int R, int N, int K);
