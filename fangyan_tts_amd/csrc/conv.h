// 1-D causal convolution kernels over channels-last fp32 activations.
//   conv1d_f32_direct : exact fp32 VALU kernel, any shape: the FY_DIRECT cross-check
//                       of every conv and the layers of reduced-size models that
//                       the MFMA tilings do not fit.
//   conv1d_f32_mfma   : the same arithmetic on the fp32 matrix instruction (f0
//                       predictor, PreLookahead - the layers that must stay fp32).
//   conv1d_bf16_mfma  : implicit GEMM on v_mfma_f32_32x32x16_bf16; the input tile
//                       (+ causal halo) is activated once and staged in LDS as
//                       bf16, the taps walk LDS rows, weights stream from L2 in
//                       fragment order.  fp32 in HBM on both sides, or the bf16
//                       activation streams x_act / y_act between the two convs of
//                       a ResBlock iteration (same values the LDS staging would
//                       have produced, half the bytes, no staging arithmetic).
#pragma once
#include "common.h"

// the snake every bf16 stream and every LDS staging of the MFMA convs applies (one definition: a stream holds the very values the
// consumer's staging would have computed)
__device__ __forceinline__ float snake_fast(float x, float a) {
    float s = __sinf(x * a);
    return fmaf(__builtin_amdgcn_rcpf(a + 1e-9f), s * s, x);
}

struct ConvDesc {
    // input: element (b, row, c) at x + b*x_bs + row*x_ld + c   (fp32)
    const float* x; long x_bs; int x_ld;
    int L_in; const int* in_len;          // rows >= in_len[b] (or L_in) and rows < 0 read as zero
    // output, same addressing
    float* y; long y_bs; int y_ld;
    int L_out; const int* out_len;        // conv output rows computed for batch b
    const float* resid; long r_bs; int r_ld;
    const float* bias;                    // [Cout] or null
    const float* alpha;                   // snake alpha [Cin] (pre_act == ACT_SNAKE)
    int B, Cin, Cout, KW, dil, stride, up, pad_left, groups;
    int pre_act; float pre_slope;
    int post_act; float post_slope;
    int add_resid;                        // v += resid[q]
    int accumulate;                       // y[q] += v*scale  instead of  y[q] = v*scale
    float out_scale;
    int reflect1;                         // ReflectionPad1d((1,0)) of the conv output: row p -> q = p+1, q=0 <- p=1
    // bf16 activation streams of the MFMA kernel (conv1d_bf16_mfma only; both optional):
    //   x_act : the input, ALREADY activated and rounded by its producer, addressed like x (x, pre_act are ignored)
    //   y_act : snake(result; alpha_out) rounded to bf16, addressed like y -- what the consumer conv would have
    //           staged from y; y itself may then be null when nobody else reads it
    //   y_act2 / y_act3 : further streams of the same result under other alphas (alpha_out2 / alpha_out3) - the three ResBlocks of a
    //           HiFT stage all start from the up-conv's output, each with its own snake; need y_act
    //   alpha_mod : the alphas are indexed by (output channel % alpha_mod) when non-zero (the polyphase up-conv's channels are
    //           [phase][C]: every phase of a channel takes that channel's alpha)
    const bf16_t* x_act;
    bf16_t* y_act;
    const float* alpha_out;
    bf16_t* y_act2; const float* alpha_out2;
    bf16_t* y_act3; const float* alpha_out3;
    int alpha_mod;
};

// weights of one conv in both kernel layouts (either may be null)
struct ConvW {
    float* w_dir = nullptr;    // [KW][Cin_g][Cout_pad4] fp32
    bf16_t* w_mfma = nullptr;  // [g][KW][Cin_g_pad/16][Cout_g/32][512] bf16 (B-fragment order)
    float* bias = nullptr;     // [Cout]
    int Cin = 0, Cout = 0, KW = 0, groups = 1;
    int cout_pad4() const { return (Cout + 3) & ~3; }
    int cin_g_pad() const { return ((Cin / groups + 15) / 16) * 16; }
};

// pack from the reference layout (Cout, Cin/groups, KW) fp32; g != null folds weight_norm:
// w = g * v / ||v|| per output channel (torch._weight_norm, dim 0)
int conv_pack(ConvW& cw, const float* v, const float* g, const float* bias, int Cout, int Cin, int KW, int groups,
              bool want_direct, bool want_mfma, hipStream_t st);
// nearest-repeat x`up` + causal conv as a stride-1 conv with up*Cout output channels on the un-repeated input (MFMA layout only)
int conv_pack_polyphase(ConvW& cw, const float* v, const float* g, const float* bias, int Cout, int Cin, int KW, int up, hipStream_t st);
void conv_free(ConvW& cw);

int conv1d_f32_direct(const ConvDesc& d, const ConvW& w, hipStream_t st);
int conv1d_bf16_mfma(const ConvDesc& d, const ConvW& w, bool precise, hipStream_t st);
// exact fp32 on v_mfma_f32_32x32x2_f32 (w.w_dir layout); stride 1, one group; bias, ELU / leaky-relu, residual add
int conv1d_f32_mfma(const ConvDesc& d, const ConvW& w, hipStream_t st);

// One ResBlock iteration of the HiFT generator in ONE launch (hifigan/generator.py:110-117):
//     out = conv2(snake(conv1(snake(x; a1)); a2)) + x,        conv1: k taps, dilation d; conv2: k taps, dilation 1; both causal
// The inner activation never leaves the chip: a workgroup computes conv1 on its rows plus the k-1 rows conv2 needs to the left,
// rounds snake(.) to bf16 into LDS (the very values conv1's bf16 output stream carried) and runs conv2 from there.  Same MFMA
// order per output as the two-launch form, so the results are identical bit for bit.  C = 64 or 128 channels.
struct ResIterDesc {
    const float* x;            // fp32 input of conv1 (activated with alpha1 while staged) ...
    const bf16_t* x_act;       // ... or its bf16 stream, already activated by the producer (x_act != null: x is not read)
    const float* resid;        // the iteration's x, fp32: added to conv2's result
    long bs; int ld;           // every tensor: element (b, row, c) at base + b*bs + row*ld + c, rows [0, L), ld == C
    const float* alpha1;       // snake of conv1's input (fp32 input only)
    const float* alpha2;       // snake between the convs
    const float* bias1; const float* bias2;
    int KW, dil;               // taps of both convs, dilation of conv1
    float* y;                  // fp32 result: y = (conv2 + resid) * out_scale, or y += ... (accumulate); may be null when only y_act is wanted
    bf16_t* y_act;             // optional: snake(result; alpha_out) rounded to bf16 - the next iteration's conv1 input stream
    const float* alpha_out;
    float out_scale; int accumulate;
    int B, L, C; const int* len;   // rows >= len[b] (null: L) are neither read nor written
};
int conv_resblock_iter(const ResIterDesc& d, const ConvW& w1, const ConvW& w2, hipStream_t st);
bool conv_resblock_iter_supported(int C, int KW, int dil);
