// Decode-side products of the speech-token LM for up to 32 activation rows per weight pass (gemv32.hip).
//
// The 8-row products of gemm.h (gemv_bf16w) stream every weight matrix once per 8 sequences; a pipeline that has 32
// utterances in flight (cli/model.py:tts_pipeline) then reads the 728 MB of LM weights four times per token step.  Here one
// pass serves 32 rows: the fp32 activations travel between the products as an "A image" - the exact 3-way bf16 split
// x = hi + mid + lo, already in the A-fragment order of v_mfma_f32_32x32x16_bf16 - written ONCE by the producing kernel's
// epilogue (o-proj / down projection: residual add + next RMSNorm weight; gate/up: SwiGLU; attention; the sampler), and
// every consumer block reads its fragments straight from L2 into registers: no staging, no split arithmetic, no barrier
// in front of the MFMAs.  Three MFMAs (hi, mid, lo planes of 32 rows) per 1 KiB weight fragment accumulate into one
// 32 x 32 accumulator; with bf16-exact weights every product is exact and the sum is fp32, as in gemv_bf16w.
#pragma once
#include "common.h"

// A image of R rows x K columns (K % 16 == 0): [ceil(R/32)][K/16][3 planes][64 lanes][8] bf16;
// lane l of fragment kk holds row (l & 31), columns 16 kk + 8 (l >> 5) .. + 7.  Rows beyond R are zero.
inline size_t gv32_image_elems(int R, int K) { return (size_t)((R + 31) / 32) * (K / 16) * 3 * 512; }
// RMSNorm partial sums of squares, one per 32-column tile of the producer: [K/32][32 ceil(R/32)] floats
inline size_t gv32_ssq_floats(int R, int K) { return (size_t)(K / 32) * 32 * ((R + 31) / 32); }
size_t gv32_partial_floats(int R, int N, int K);
size_t gv32_counter_ints(int R, int N, int K);

enum { GV32_STORE = 0, GV32_ADD_IMG = 1, GV32_SWIGLU_IMG = 2 };
struct Gv32Args {
    const bf16_t* W = nullptr;       // gemv_pack's fragment order [N/32][K/16][64][8]
    // exact-weights mode (general fp32 checkpoints): the matrix is W + W_lo, W = bf16(w), W_lo = bf16(w - W), same order; per
    // fragment the hi plane meets all three operand planes and the lo plane the two leading ones (the dropped lo x lo term is
    // 2^-26 of the product) - five MFMAs instead of three.  null: one plane (weights that are bf16-representable)
    const bf16_t* W_lo = nullptr;
    const bf16_t* img = nullptr;     // A image of the R x K operand
    int R = 0, N = 0, K = 0;
    // fused RMSNorm of the operand (the image holds norm_w * x): acc *= rsqrt(sum of the n_ssq partials of the row / K + eps)
    const float* ssq = nullptr;
    int n_ssq = 0;
    float eps = 0.f;
    const float* bias = nullptr;     // [N], GV32_STORE
    int mode = GV32_STORE;
    float* y = nullptr;              // GV32_STORE: y[r][n] = acc * rstd + bias;  GV32_ADD_IMG: y[r][n] += acc (the residual stream)
    int ldy = 0;
    // GV32_ADD_IMG: img_out = image of (ln_next * y_new) with N columns, ssq_out[N/32][rows] = per-tile sums of y_new^2
    // GV32_SWIGLU_IMG: W's rows are interleaved (gate_i, up_i); img_out = image of silu(gate) * up with N/2 columns
    const float* ln_next = nullptr;
    bf16_t* img_out = nullptr;
    float* ssq_out = nullptr;
    float* partial = nullptr;        // split-K workspace (gv32_partial_floats) and zero-initialised arrival counters
    int* counters = nullptr;         // (gv32_counter_ints); both or neither
};
int gemv32(const Gv32Args& a, hipStream_t st);

// x fp32 [R][ldx] (K columns) -> A image of (ln * x) (ln null: of x) and, if ssq is given, the per-tile sums of x^2
int gv32_split_rows(const float* x, int ldx, int R, int K, const float* ln, bf16_t* img, float* ssq, hipStream_t st);

// device helpers shared with the kernels that write images themselves (attention step, sampler)
#ifdef __HIPCC__
// byte offset inside an image with K16 = K/16 fragments per slice of the 8-byte half piece (plane, row, columns col0 .. col0+3), col0 % 4 == 0
__device__ __forceinline__ long gv32_off(int K16, int row, int col0, int plane) {
    const int z = row >> 5, r = row & 31, kk = col0 >> 4, half = (col0 >> 3) & 1, j = col0 & 7;
    return ((((long)z * K16 + kk) * 3 + plane) * 64 + half * 32 + r) * 16 + j * 2;
}
__device__ __forceinline__ void gv32_split3(float f, unsigned& h, unsigned& m, unsigned& l) {
    __bf16 hb = (__bf16)f;
    float r1 = f - (float)hb;
    __bf16 mb = (__bf16)r1;
    float r2 = r1 - (float)mb;
    __bf16 lb = (__bf16)r2;
    h = __builtin_bit_cast(unsigned short, hb);
    m = __builtin_bit_cast(unsigned short, mb);
    l = __builtin_bit_cast(unsigned short, lb);
}
// four consecutive columns of one row -> the three planes (8 bytes each)
__device__ __forceinline__ void gv32_put4(bf16_t* img, int K16, int row, int col0, float x0, float x1, float x2, float x3) {
    unsigned h[4], m[4], l[4];
    gv32_split3(x0, h[0], m[0], l[0]); gv32_split3(x1, h[1], m[1], l[1]);
    gv32_split3(x2, h[2], m[2], l[2]); gv32_split3(x3, h[3], m[3], l[3]);
    char* base = reinterpret_cast<char*>(img);
    *reinterpret_cast<uint2*>(base + gv32_off(K16, row, col0, 0)) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    *reinterpret_cast<uint2*>(base + gv32_off(K16, row, col0, 1)) = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
    *reinterpret_cast<uint2*>(base + gv32_off(K16, row, col0, 2)) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
}
#endif
