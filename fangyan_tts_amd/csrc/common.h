// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// Wave = 64 lanes; MFMA fragments follow the gfx950 lane maps
// (v_mfma_f32_32x32x16_bf16: A[row l&31][k 8(l>>5)+j], B[k 8(l>>5)+j][col l&31],
//  D[row (r&3)+8(r>>2)+4(l>>5)][col l&31]).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <string.h>
#include <stdio.h>

typedef unsigned short bf16_t;                                   // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;        // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16;       // 32x32 accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define FY_OK 0
#define FY_ERR_ARG (-1)
#define FY_ERR_HIP (-2)
#define FY_ERR_WEIGHT (-3)
#define FY_ERR_STATE (-4)

void fy_set_error(const char* fmt, ...);

#define HIP_TRY(expr)                                                                 \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            fy_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return FY_ERR_HIP;                                                        \
        }                                                                             \
    } while (0)

#define FY_TRY(expr)                 \
    do {                             \
        int _r = (expr);             \
        if (_r != FY_OK) return _r;  \
    } while (0)

#define FY_CHECK(cond, code, ...)    \
    do {                             \
        if (!(cond)) {               \
            fy_set_error(__VA_ARGS__); \
            return (code);           \
        }                            \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- bf16 <-> fp32 -----------------------------------------------------------
__host__ __device__ inline float bf16_to_f32(bf16_t h) {
    union { uint32_t u; float f; } v;
    v.u = ((uint32_t)h) << 16;
    return v.f;
}
// round to nearest even; inputs here are finite activations / weights
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
    union { uint32_t u; float f; } v;
    v.f = f;
    uint32_t u = v.u;
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// ---- activations -----------------------------------------------------------------
enum { ACT_NONE = 0, ACT_LEAKY = 1, ACT_SNAKE = 2, ACT_ELU = 3, ACT_MISH = 4, ACT_GELU_TANH = 5, ACT_SILU = 6 };

__device__ inline float act_leaky(float x, float slope) { return x >= 0.f ? x : x * slope; }
// Snake.forward, CosyVoice/cosyvoice/transformer/activation.py:73-84
__device__ inline float act_snake(float x, float a) {
    float s = sinf(x * a);
    return x + (1.0f / (a + 1e-9f)) * (s * s);
}
__device__ inline float act_elu(float x) { return x > 0.f ? x : expm1f(x); }
__device__ inline float act_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ inline float act_mish(float x) { return x * tanhf(act_softplus(x)); }
// mish through one exponential: tanh(log(1 + e)) = n / (n + 2) with n = e (e + 2), e = exp(x)  (an identity; for results that
// are rounded to bf16 next, hardware exp / rcp are ample)
__device__ inline float act_mish_fast(float x) {
    const float e = __expf(fminf(x, 20.f));
    const float n = e * (e + 2.f);
    return x * n * __builtin_amdgcn_rcpf(n + 2.f);
}
__device__ inline float act_silu(float x) { return x / (1.0f + expf(-x)); }
// tanh-GELU on values that are rounded to bf16 next: tanh(u) = 1 - 2 / (exp(2u) + 1) with the hardware exp
__device__ inline float act_gelu_tanh_fast(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float u = k0 * (x + k1 * x * x * x);
    float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * u) + 1.0f);
    return 0.5f * x * (1.0f + t);
}
__device__ inline float act_gelu_tanh(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    return 0.5f * x * (1.0f + tanhf(k0 * (x + k1 * x * x * x)));
}

// ---- wave reductions (64 lanes) ------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
