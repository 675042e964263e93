// See gemm.h.
#include "gemm.h"
#include "runtime.h"

typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;

#define GM_BM 128
#define GM_BN 128
#define GM_BK 32
#define GM_PITCH 40          // bf16 elements per LDS row: 32 + 8 pad (80 B, keeps 16-B alignment)

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

template <bool PRECISE>
__global__ __launch_bounds__(256) void gemm_bf16_k(const void* __restrict__ Av, int lda, const bf16_t* __restrict__ W, int M, int N, int K, GemmEpi e) {
    __shared__ __attribute__((aligned(16))) bf16_t As[(PRECISE ? 2 : 1) * GM_BM * GM_PITCH];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[GM_BN * GM_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1, lr = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * GM_BM, n0 = blockIdx.x * GM_BN;
    const bf16_t* Ab = (const bf16_t*)Av;
    const float* Af = (const float*)Av;

    uint4 ra[2], ral[2], rb[2];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int idx = tid + i * 256, row = idx >> 2, kc = (idx & 3) * 8;
            int gm = m0 + row, gn = n0 + row;
            if (PRECISE) {
                float4 x0 = make_float4(0, 0, 0, 0), x1 = x0;
                if (gm < M) {
                    const float* p = Af + (long)gm * lda + k0 + kc;
                    x0 = *reinterpret_cast<const float4*>(p);
                    x1 = *reinterpret_cast<const float4*>(p + 4);
                }
                split8(x0, x1, ra[i], ral[i]);
            } else {
                ra[i] = gm < M ? *reinterpret_cast<const uint4*>(Ab + (long)gm * lda + k0 + kc) : make_uint4(0, 0, 0, 0);
            }
            rb[i] = gn < N ? *reinterpret_cast<const uint4*>(W + (long)gn * K + k0 + kc) : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int idx = tid + i * 256, row = idx >> 2, kc = (idx & 3) * 8;
            *reinterpret_cast<uint4*>(As + row * GM_PITCH + kc) = ra[i];
            if (PRECISE) *reinterpret_cast<uint4*>(As + GM_BM * GM_PITCH + row * GM_PITCH + kc) = ral[i];
            *reinterpret_cast<uint4*>(Bs + row * GM_PITCH + kc) = rb[i];
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
    store_tile();
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += GM_BK) {
        const bool more = k0 + GM_BK < K;
        if (more) load_tile(k0 + GM_BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag_ab a[2], al[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                int off = (wm * 64 + mi * 32 + lr) * GM_PITCH + ks * 16 + kh * 8;
                a[mi] = *reinterpret_cast<const frag_ab*>(As + off);
                if (PRECISE) al[mi] = *reinterpret_cast<const frag_ab*>(As + GM_BM * GM_PITCH + off);
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                b[ni] = *reinterpret_cast<const frag_ab*>(Bs + (wn * 64 + ni * 32 + lr) * GM_PITCH + ks * 16 + kh * 8);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
                    if (PRECISE) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], b[ni], acc[mi][ni], 0, 0, 0);
                }
        }
        __syncthreads();
        if (more) {
            store_tile();
            __syncthreads();
        }
    }

#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + wn * 64 + ni * 32 + lr;
        if (n >= N) continue;
        const float bv = e.bias ? e.bias[n] : 0.f;
        const float gv = e.mode == EPI_GATE_RESID ? e.gate[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (m >= M) continue;
                float v = acc[mi][ni][r] + bv;
                if (e.mode == EPI_GATE_RESID) {
                    float* p = e.resid + (long)m * e.ldc + n;
                    *p = *p + gv * v;
                } else {
                    if (e.act == ACT_GELU_TANH) v = act_gelu_tanh(v);
                    else if (e.act == ACT_SILU) v = act_silu(v);
                    else if (e.act == ACT_MISH) v = act_mish(v);
                    if (e.out_bf16) ((bf16_t*)e.out)[(long)m * e.ldc + n] = f32_to_bf16(v);
                    else ((float*)e.out)[(long)m * e.ldc + n] = v;
                }
            }
        }
    }
}

static int gemm_check(const void* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& e, int a_elem) {
    FY_CHECK(A && W && M >= 1 && N >= 1 && K >= GM_BK && K % GM_BK == 0, FY_ERR_ARG, "gemm: bad shape M %d N %d K %d", M, N, K);
    FY_CHECK(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && (lda * a_elem) % 16 == 0, FY_ERR_ARG, "gemm: operands must be 16-B aligned");
    FY_CHECK(e.ldc >= N && ((e.mode == EPI_STORE && e.out) || (e.mode == EPI_GATE_RESID && e.resid && e.gate)), FY_ERR_ARG, "gemm: bad epilogue");
    return FY_OK;
}

int gemm_bf16(const bf16_t* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st) {
    FY_TRY(gemm_check(A, lda, W, M, N, K, epi, 2));
    dim3 grid(cdiv(N, GM_BN), cdiv(M, GM_BM));
    ProfScope prof("gemm_bf16", 2.0 * M * N * K, st);
    hipLaunchKernelGGL(gemm_bf16_k<false>, grid, dim3(256), 0, st, (const void*)A, lda, W, M, N, K, epi);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

int gemm_f32a_precise(const float* A, int lda, const bf16_t* W, int M, int N, int K, const GemmEpi& epi, hipStream_t st) {
    FY_TRY(gemm_check(A, lda, W, M, N, K, epi, 4));
    dim3 grid(cdiv(N, GM_BN), cdiv(M, GM_BM));
    hipLaunchKernelGGL(gemm_bf16_k<true>, grid, dim3(256), 0, st, (const void*)A, lda, W, M, N, K, epi);
    HIP_TRY(hipGetLastError());
    return FY_OK;
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
// decode GEMV: block = 4 waves x 4 output rows; lanes split K in 8-element chunks;
// 8 activation rows staged in LDS as fp32; fp32 FMA on bf16 weights widened in registers.
// =============================================================================
#define GV_ROWS 4
#define GV_KSLICE 1280       // activations staged per block: 8 x 1280 fp32 = 40 KB

__global__ __launch_bounds__(256) void gemv_bf16w_k(GemvArgs a, int kslice, int ksplit) {
    extern __shared__ __attribute__((aligned(16))) float xs[];       // [8][kslice]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ks = blockIdx.y;
    const int k0 = ks * kslice;
    const int kn = min(kslice, a.K - k0);                              // multiple of 8
    const float* xg = a.x + (long)blockIdx.z * 8 * a.ldx;
    const int R = min(8, a.R - blockIdx.z * 8);
    for (int i = tid; i < 8 * (kn / 4); i += 256) {
        int r = i / (kn / 4), c4 = (i % (kn / 4)) * 4;
        float4 v = make_float4(0, 0, 0, 0);
        if (r < R) v = *reinterpret_cast<const float4*>(xg + (long)r * a.ldx + k0 + c4);
        *reinterpret_cast<float4*>(xs + r * kslice + c4) = v;
    }
    __syncthreads();
    const int n_first = (blockIdx.x * 4 + wid) * GV_ROWS;
    if (n_first >= a.N) return;
    float acc[GV_ROWS * 8];
#pragma unroll
    for (int i = 0; i < GV_ROWS * 8; ++i) acc[i] = 0.f;
    const int nchunk = kn / 8;
    for (int c = lane; c < nchunk; c += 64) {
        uint4 w[GV_ROWS];
#pragma unroll
        for (int j = 0; j < GV_ROWS; ++j) {
            int n = min(n_first + j, a.N - 1);
            w[j] = *reinterpret_cast<const uint4*>(a.W + (long)n * a.K + k0 + c * 8);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float4 x0 = *reinterpret_cast<const float4*>(xs + r * kslice + c * 8);
            float4 x1 = *reinterpret_cast<const float4*>(xs + r * kslice + c * 8 + 4);
#pragma unroll
            for (int j = 0; j < GV_ROWS; ++j) {
                float s = acc[j * 8 + r];
                s = fmaf(__uint_as_float(w[j].x << 16), x0.x, s);
                s = fmaf(__uint_as_float(w[j].x & 0xFFFF0000u), x0.y, s);
                s = fmaf(__uint_as_float(w[j].y << 16), x0.z, s);
                s = fmaf(__uint_as_float(w[j].y & 0xFFFF0000u), x0.w, s);
                s = fmaf(__uint_as_float(w[j].z << 16), x1.x, s);
                s = fmaf(__uint_as_float(w[j].z & 0xFFFF0000u), x1.y, s);
                s = fmaf(__uint_as_float(w[j].w << 16), x1.z, s);
                s = fmaf(__uint_as_float(w[j].w & 0xFFFF0000u), x1.w, s);
                acc[j * 8 + r] = s;
            }
        }
    }
    // transpose-reduce the 32 per-lane partials over the 64 lanes: 32 shuffles instead of 32 x 6
#pragma unroll
    for (int half = 16, bit = 32; half >= 1; half >>= 1, bit >>= 1) {
        const bool up = (lane & bit) != 0;
#pragma unroll
        for (int i = 0; i < half; ++i) {
            float keep = up ? acc[i + half] : acc[i];
            float send = up ? acc[i] : acc[i + half];
            acc[i] = keep + __shfl_xor(send, bit, 64);
        }
    }
    float tot = acc[0] + __shfl_xor(acc[0], 1, 64);
    if (lane & 1) return;
    const int v = lane >> 1, j = v >> 3, r = v & 7;
    const int n = n_first + j;
    if (n >= a.N || r >= R) return;
    const long rr = (long)blockIdx.z * 8 + r;
    if (ksplit > 1) {
        a.partial[((long)ks * a.R + rr) * a.N + n] = tot;
        return;
    }
    if (a.bias) tot += a.bias[n];
    if (a.mode == GV_SWIGLU) {
        // rows interleaved (gate_i, up_i): the pair sits in lanes v and v + 8 (j even / j + 1)
        float other = __shfl(tot, lane + 16, 64);
        if ((j & 1) == 0) a.y[rr * a.ldy + (n >> 1)] = act_silu(tot) * other;
    } else if (a.mode == GV_ADD) {
        a.y[rr * a.ldy + n] += tot;
    } else {
        a.y[rr * a.ldy + n] = tot;
    }
}

__global__ void gemv_reduce_k(GemvArgs a, int ksplit) {
    long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= (long)a.R * a.N) return;
    int n = (int)(i % a.N);
    long r = i / a.N;
    float s = 0.f;
    for (int k = 0; k < ksplit; ++k) s += a.partial[((long)k * a.R + r) * a.N + n];
    if (a.bias) s += a.bias[n];
    if (a.mode == GV_ADD) a.y[r * a.ldy + n] += s; else a.y[r * a.ldy + n] = s;
}

static int gv_ksplit(int K) { return (K + GV_KSLICE - 1) / GV_KSLICE; }
size_t gemv_partial_floats(int R, int N, int K) { return gv_ksplit(K) > 1 ? (size_t)gv_ksplit(K) * R * N : 0; }

int gemv_bf16w(const GemvArgs& a, hipStream_t st) {
    FY_CHECK(a.W && a.x && a.y && a.R >= 1 && a.N >= 1 && a.K >= 8 && a.K % 8 == 0 && a.ldx % 4 == 0, FY_ERR_ARG,
             "gemv: bad arguments R %d N %d K %d", a.R, a.N, a.K);
    int ksplit = gv_ksplit(a.K);
    int kslice = ksplit == 1 ? a.K : ((a.K / ksplit + 7) / 8) * 8;
    while (kslice * ksplit < a.K) kslice += 8;
    FY_CHECK(ksplit == 1 || (a.partial && a.mode != GV_SWIGLU), FY_ERR_ARG, "gemv: split-K needs a workspace and a plain epilogue");
    FY_CHECK(a.mode != GV_SWIGLU || a.N % 8 == 0, FY_ERR_ARG, "gemv: SwiGLU rows must come in interleaved pairs");
    dim3 grid(cdiv(a.N, 4 * GV_ROWS), ksplit, cdiv(a.R, 8));
    ProfScope prof("gemv", 2.0 * a.N * a.K, st);
    hipLaunchKernelGGL(gemv_bf16w_k, grid, dim3(256), (size_t)8 * kslice * sizeof(float), st, a, kslice, ksplit);
    if (ksplit > 1) hipLaunchKernelGGL(gemv_reduce_k, dim3(cdiv(a.R * a.N, 256)), dim3(256), 0, st, a, ksplit);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}
