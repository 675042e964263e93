// Flow-matching acoustic decoder engine: CausalMaskedDiffWithDiT.inference ->
// CausalConditionalCFM.solve_euler -> DiT.forward for a batch of ragged utterances.
// Reference: CosyVoice/cosyvoice/flow/flow.py:358-403, flow/flow_matching.py:71-124,
// flow/DiT/dit.py:145-176, flow/DiT/modules.py (cited per kernel).
//
// Layout: the DiT works on nseq = 2*B sequences (row 2b = conditional, 2b+1 = CFG branch with
// mu = spks = cond = 0), each padded to Tmax rows: row (s*Tmax + t).  The residual stream h is
// fp32; GEMM operands are bf16 with fp32 MFMA accumulation; the adaLN-zero modulation depends on
// the Euler step only, so all (step, block) vectors are computed once at create.
#include "attn.h"
#include "conv.h"
#include "gemm.h"
#include "runtime.h"
#include <math.h>

struct FlowBlockW {
    bf16_t *wqkv, *wo, *w1, *w2, *wmod;
    float *bqkv, *bo, *b1, *b2, *bmod;
    // lo planes bf16(w - bf16(w)), same layouts, kept only when some weight of flow.pt is not bf16-representable (fy_flow::planes == 2):
    // FY_PRECISE then multiplies by w = hi + lo (gemm_split's w_lo), and the create-time modulation vectors use both planes
    bf16_t *wqkv_lo = nullptr, *wo_lo = nullptr, *w1_lo = nullptr, *w2_lo = nullptr, *wmod_lo = nullptr;
};

struct fy_flow {
    fy_flow_config cfg;
    int max_batch = 0, Tmax = 0, Nmax = 0;       // Tmax mel frames, Nmax = Tmax/2 tokens
    DevPool pool;
    // front
    float *emb_w, *spk_w, *spk_b;
    ConvW pre1, pre2, pos1, pos2;
    // estimator
    bf16_t *w_in, *w_out, *w_t0, *w_t2, *w_fin;
    bf16_t *w_in_lo = nullptr, *w_out_lo = nullptr, *w_t0_lo = nullptr, *w_t2_lo = nullptr, *w_fin_lo = nullptr;
    int planes = 1;            // 2: a general fp32 checkpoint (cli/model.py:65-73) - the lo planes above are kept
    float* ones = nullptr;     // [dim] ones (the gate of a lo-plane pass that adds into an fp32 output)
    float *b_in, *b_out, *b_t0, *b_t2, *b_fin;
    std::vector<FlowBlockW> blk;
    // LayerNorm-modulate folded across the qkv / ff1 products (round 5; dit_forward says when): per (Euler step, block) the weights
    // W' = bf16(W (1 + scale)), u[n] = sum_k W'[n][k] and bias' = bias + W shift - the modulation depends on the step only, so all of
    // it is made once at create (2.2 GB at full size) - and, per call, bf16(h) with its per-(row, 64-column tile) sums from the
    // producing product's epilogue (hb, ln_slots)
    bool fold = false;
    bf16_t *wq_fold = nullptr, *w1_fold = nullptr;       // [n_steps][depth][3 inner][dim], [n_steps][depth][ff][dim]
    float *uq_fold = nullptr, *bq_fold = nullptr, *u1_fold = nullptr, *b1_fold = nullptr;      // [n_steps][depth][3 inner] / [ff]
    bf16_t* hb = nullptr;
    float2* ln_slots = nullptr;
    float* mod = nullptr;      // [(n_steps+1)][depth][6*dim]   (slot n_steps = scratch for fy_dit_estimator)
    float* fin = nullptr;      // [(n_steps+1)][2*dim]
    float2* rope = nullptr;    // [Tmax][head_dim/2] (cos, sin)
    std::vector<float> t_of_step, dt_of_step;
    // activations
    int *tok_all, *seq_len, *blen;     // blen: [5][max_batch] = n_all tokens, T, pmel, prompt tokens, tokens PreLookahead outputs
    float *emb, *pre_a, *mu_tok, *spks, *x, *mu, *cond, *h, *c1, *v, *temb, *tsil, *gpart;
    bf16_t *a_in, *xn, *qkv, *ao, *ff;
    // lo planes of the fp32-class mode (FY_PRECISE): every GEMM operand travels as x = hi + lo, the hi plane in the default mode's
    // own buffer (a_in, xn, qkv, ao, ff: hi = bf16(x) is exactly what the default mode stores), the lo plane here; allocated on first use
    bf16_t *a_in_lo = nullptr, *xn_lo = nullptr, *qkv_lo = nullptr, *ao_lo = nullptr, *ff_lo = nullptr;
    // incremental streaming (FY_INCREMENTAL, batch 1): the q | k | v rows of every (Euler step, block) as they were computed -
    // [n_steps][depth][2 Tmax][3 inner] bf16 - and the ODE state at every step, [(n_steps + 1)][Tmax][mel] fp32; inc_T = rows kept
    bf16_t* inc_qkv = nullptr;
    float* inc_x = nullptr;
    int inc_T = 0;
    int inc_np = -1, inc_npf = -1, inc_ntok = -1;       // the stream's prompt tokens / prompt frames / tokens at the last kept call (a cheap check that a call extends it)
    ~fy_flow() { conv_free(pre1); conv_free(pre2); conv_free(pos1); conv_free(pos2); }
};

extern "C" void fy_flow_default_config(fy_flow_config* c) {
    memset(c, 0, sizeof(*c));
    c->mel = 80; c->spk_in = 192; c->vocab = 6561; c->pre_ch = 1024; c->pre_lookahead = 3;
    c->dim = 1024; c->depth = 22; c->heads = 16; c->head_dim = 64; c->ff_mult = 2;
    c->conv_pos_k = 31; c->conv_pos_groups = 16; c->n_timesteps = 10; c->cfg_rate = 0.7f; c->static_chunk = 50;
    // t_span = 1 - cos(linspace(0,1,n+1) * pi/2)  (flow_matching.py:223-225); callers that need the
    // reference's exact fp32 values overwrite it with torch's
    for (int i = 0; i <= 10; ++i) c->t_span[i] = (float)(1.0 - cos(((double)i / 10.0) * 0.5 * M_PI));
}

// ---- small kernels ------------------------------------------------------------------------------
// F.normalize(embedding, dim=1) then Linear(192 -> 80): flow.py:371-372.  One block per utterance.
__global__ void spk_k(const float* __restrict__ e, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ out, int nin, int nout) {
    __shared__ float sh[256];
    __shared__ float xs[512];
    int bb = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < nin; i += 256) { float v = e[(long)bb * nin + i]; xs[i] = v; s += v * v; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    float nrm = fmaxf(sqrtf(sh[0]), 1e-12f);
    for (int o = threadIdx.x; o < nout; o += 256) {
        float a = 0.f;
        for (int i = 0; i < nin; ++i) a = fmaf(xs[i] / nrm, w[(long)o * nin + i], a);
        out[(long)bb * nout + o] = a + b[o];
    }
}

// concat[prompt_token, token] and the embedding gather, flow.py:375-377 (mask is all ones inside n_all)
__global__ void tok_embed_k(const int* __restrict__ ptok, int p_ld, const int* __restrict__ tok, int t_ld, const int* __restrict__ n_prompt,
                            const int* __restrict__ n_all, int* __restrict__ tok_all, const float* __restrict__ W, float* __restrict__ emb,
                            int Nmax, int C, int vocab) {
    int b = blockIdx.y, i = blockIdx.x;
    if (i >= n_all[b]) return;
    int P = n_prompt[b];
    int id = i < P ? ptok[(long)b * p_ld + i] : tok[(long)b * t_ld + (i - P)];
    id = max(id, 0);
    id = min(id, vocab - 1);
    if (threadIdx.x == 0) tok_all[(long)b * Nmax + i] = id;
    for (int c = threadIdx.x; c < C; c += blockDim.x) emb[((long)b * Nmax + i) * C + c] = W[(long)id * C + c];
}

// mu = repeat_interleave(h, 2), cond = [prompt_feat; 0], x0 = rand_noise[:, :T]   (flow.py:383-390, flow_matching.py:220)
__global__ void flow_setup_k(const float* __restrict__ mu_tok, const float* __restrict__ pfeat, long pf_bs, const float* __restrict__ noise, int noise_ld,
                             const int* __restrict__ T, const int* __restrict__ pmel, float* __restrict__ mu, float* __restrict__ cond,
                             float* __restrict__ x, int Tmax, int Nmax, int C) {
    int b = blockIdx.y, t = blockIdx.x, c = threadIdx.x;
    if (t >= T[b] || c >= C) return;
    long o = ((long)b * Tmax + t) * C + c;
    mu[o] = mu_tok[((long)b * Nmax + (t >> 1)) * C + c];
    cond[o] = t < pmel[b] ? pfeat[(long)b * pf_bs + (long)t * C + c] : 0.f;
    x[o] = noise[(long)c * noise_ld + t];
}

// InputEmbedding concat [x, cond, mu, spks] (dit.py:91-96) for the conditional row and [x, 0, 0, 0]
// for the CFG row (flow_matching.py:95-101), as the bf16 A operand of the input projection.
__global__ void dit_assemble_k(const float* __restrict__ x, const float* __restrict__ cond, const float* __restrict__ mu, const float* __restrict__ spks,
                               const int* __restrict__ T, bf16_t* __restrict__ a, bf16_t* __restrict__ a_lo, int Tmax, int C, int seq_rows, int row_step) {
    // output row of (sequence s, frame t): s * seq_rows + t * row_step (s Tmax + t; or 2 t + s - the incremental streaming layout)
    int s = blockIdx.y, t = blockIdx.x, b = s >> 1, cfg = s & 1;
    int i = threadIdx.x;                     // 0 .. 4C
    if (i >= 4 * C) return;
    float v = 0.f;
    if (t < T[b]) {
        long o = ((long)b * Tmax + t) * C;
        int part = i / C, c = i % C;
        if (part == 0) v = x[o + c];
        else if (!cfg) v = part == 1 ? cond[o + c] : (part == 2 ? mu[o + c] : spks[(long)b * C + c]);
    }
    const bf16_t hb = f32_to_bf16(v);
    const long orow = (long)s * seq_rows + (long)t * row_step;
    a[orow * 4 * C + i] = hb;
    if (a_lo) a_lo[orow * 4 * C + i] = f32_to_bf16(v - bf16_to_f32(hb));      // fp32-class mode (FY_PRECISE): x = hi + lo
}

// LayerNorm(eps 1e-6, no affine) * (1 + scale) + shift -> bf16 (modules.py:238-243, 524, 262-264). One wave per row.
__global__ __launch_bounds__(256) void ln_mod_k(const float* __restrict__ h, const float* __restrict__ scale, const float* __restrict__ shift,
                                                bf16_t* __restrict__ out, bf16_t* __restrict__ out_lo, int M, int D) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* p = h + (long)row * D;
    if ((D & 255) == 0 && D <= 1024) {                       // 16-byte loads, 8-byte stores: four columns per lane and pass
        const int per4 = D / 256;                            // <= 4
        float4 v[4];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[i] = i < per4 ? *reinterpret_cast<const float4*>(p + (lane + 64 * i) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mean = wave_sum(s) / D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < per4) {
                const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
                q += (a * a + b * b) + (c * c + d * d);
            }
        const float rstd = rsqrtf(wave_sum(q) / D + 1e-6f);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < per4) {
                const int c = (lane + 64 * i) * 4;
                const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
                const float y0 = (v[i].x - mean) * rstd * (1.f + sc.x) + sh.x, y1 = (v[i].y - mean) * rstd * (1.f + sc.y) + sh.y;
                const float y2 = (v[i].z - mean) * rstd * (1.f + sc.z) + sh.z, y3 = (v[i].w - mean) * rstd * (1.f + sc.w) + sh.w;
                const bf16_t h0 = f32_to_bf16(y0), h1 = f32_to_bf16(y1), h2 = f32_to_bf16(y2), h3 = f32_to_bf16(y3);
                uint2 pk;
                pk.x = (uint32_t)h0 | ((uint32_t)h1 << 16);
                pk.y = (uint32_t)h2 | ((uint32_t)h3 << 16);
                *reinterpret_cast<uint2*>(out + (long)row * D + c) = pk;
                if (out_lo) {                                    // fp32-class mode: the lo plane of x = hi + lo
                    pk.x = (uint32_t)f32_to_bf16(y0 - bf16_to_f32(h0)) | ((uint32_t)f32_to_bf16(y1 - bf16_to_f32(h1)) << 16);
                    pk.y = (uint32_t)f32_to_bf16(y2 - bf16_to_f32(h2)) | ((uint32_t)f32_to_bf16(y3 - bf16_to_f32(h3)) << 16);
                    *reinterpret_cast<uint2*>(out_lo + (long)row * D + c) = pk;
                }
            }
        return;
    }
    float v[16];
    const int per = D / 64;                   // <= 16
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i] = i < per ? p[lane + 64 * i] : 0.f; s += v[i]; }
    float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { float d = i < per ? v[i] - mean : 0.f; q += d * d; }
    float rstd = rsqrtf(wave_sum(q) / D + 1e-6f);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i < per) {
            int c = lane + 64 * i;
            const float y = (v[i] - mean) * rstd * (1.f + scale[c]) + shift[c];
            const bf16_t hb = f32_to_bf16(y);
            out[(long)row * D + c] = hb;
            if (out_lo) out_lo[(long)row * D + c] = f32_to_bf16(y - bf16_to_f32(hb));
        }
    }
}

// CFG mix and Euler update, flow_matching.py:114-118
__global__ void euler_k(float* __restrict__ x, const float* __restrict__ v, const int* __restrict__ T, int Tmax, int C, float dt, float cfg, int t0,
                        int interleaved) {
    int b = blockIdx.y, t = t0 + blockIdx.x, c = threadIdx.x;
    if (t >= T[b] || c >= C) return;
    // v rows: sequence s at s Tmax + t, or (incremental streaming, batch 1) the two sequences interleaved: 2 t + s
    float vc = interleaved ? v[(2L * t) * C + c] : v[((long)(2 * b) * Tmax + t) * C + c];
    float vu = interleaved ? v[(2L * t + 1) * C + c] : v[((long)(2 * b + 1) * Tmax + t) * C + c];
    float d = (1.0f + cfg) * vc - cfg * vu;
    long o = ((long)b * Tmax + t) * C + c;
    x[o] = x[o] + dt * d;
}

__global__ void silu_k(const float* __restrict__ a, float* __restrict__ o, long n) {
    long i = blockIdx.x * 256L + threadIdx.x;
    if (i < n) o[i] = act_silu(a[i]);
}

// ---- create -----------------------------------------------------------------------------------------
static int to_bf16(fy_flow* f, const float* src, size_t n, bf16_t** dst, hipStream_t st) {
    FY_TRY(f->pool.alloc(dst, n));
    return cast_f32_bf16(src, *dst, n, st);
}
static int to_packed(fy_flow* f, const float* src, int N, int K, bf16_t** dst, hipStream_t st) {
    FY_TRY(f->pool.alloc(dst, gemv_packed_elems(N, K)));
    return gemv_pack(src, *dst, N, K, st);
}
static int copy_f32(fy_flow* f, const float* src, size_t n, float** dst, hipStream_t st) {
    FY_TRY(f->pool.alloc(dst, n));
    HIP_TRY(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    return FY_OK;
}

__global__ void flow_bf16_residual_k(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * 256UL + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i] - bf16_to_f32(f32_to_bf16(src[i]));
}
__global__ void flow_bf16_inexact_k(const float* __restrict__ src, size_t n, unsigned long long* __restrict__ count) {
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * 256UL + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) c += bf16_to_f32(f32_to_bf16(src[i])) != src[i];
    if (c) atomicAdd(count, c);
}
__global__ void flow_fill_k(float* p, float v, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
static unsigned flow_grid(size_t n) { return (unsigned)std::min<size_t>(4096, (n + 255) / 256); }
// the lo plane of src [N][K], row-major (packed = false) or in the GEMV's fragment order; f->planes == 1: nothing
static int lo_plane(fy_flow* f, const float* src, int N, int K, bool packed, bf16_t** dst, hipStream_t st) {
    if (f->planes != 2) return FY_OK;
    float* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, (size_t)N * K * sizeof(float)));
    hipLaunchKernelGGL(flow_bf16_residual_k, dim3(flow_grid((size_t)N * K)), dim3(256), 0, st, src, tmp, (size_t)N * K);
    int rc = packed ? to_packed(f, tmp, N, K, dst, st) : to_bf16(f, tmp, (size_t)N * K, dst, st);
    (void)hipStreamSynchronize(st);
    (void)hipFree(tmp);
    return rc;
}

// One row n of a folded product (gemm.h: GemmEpi::ln_rows): W'[n][k] = bf16(W[n][k] (1 + scale[k])), u[n] = sum_k W'[n][k] (of the ROUNDED
// values: the mean term then cancels exactly), bias'[n] = bias[n] + sum_k W[n][k] shift[k].  One block per row, sums in double.
__global__ __launch_bounds__(256) void fold_row_k(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, bf16_t* __restrict__ Wp, float* __restrict__ u, float* __restrict__ bp, int K) {
    __shared__ double su[256], sv[256];
    const int n = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float w = W[(long)n * K + k];
        const bf16_t q = f32_to_bf16(w * (1.f + scale[k]));
        Wp[(long)n * K + k] = q;
        a += (double)bf16_to_f32(q);
        b += (double)w * (double)shift[k];
    }
    su[threadIdx.x] = a; sv[threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { su[threadIdx.x] += su[threadIdx.x + o]; sv[threadIdx.x] += sv[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { u[n] = (float)su[0]; bp[n] = (float)((double)bias[n] + sv[0]); }
}

// modulation vectors of one set of timesteps: slot0.. (rows = n), see modules.py:606-616, 239-241, 260-261
static int compute_mod(fy_flow* f, const float* t_host, int n, int slot0, hipStream_t st) {
    const fy_flow_config& c = f->cfg;
    const int D = c.dim;
    std::vector<float> emb((size_t)n * 256);
    const int half = 128;
    const float k = logf(10000.f) / (half - 1);
    for (int r = 0; r < n; ++r)
        for (int i = 0; i < half; ++i) {
            float fr = expf((float)i * -k);
            float a = 1000.f * t_host[r] * fr;
            emb[(size_t)r * 256 + i] = sinf(a);
            emb[(size_t)r * 256 + half + i] = cosf(a);
        }
    float* d_emb = f->temb + (size_t)16 * D;       // scratch behind temb/tsil (allocated 16*D*3)
    HIP_TRY(hipMemcpyAsync(d_emb, emb.data(), emb.size() * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    GemvArgs a;
    a.W = f->w_t0; a.x = d_emb; a.ldx = 256; a.R = n; a.N = D; a.K = 256; a.bias = f->b_t0; a.y = f->tsil; a.ldy = D;
    // a product with both weight planes (general fp32 weights): y = W x + b, then y += W_lo x
    auto both = [&](GemvArgs g, const bf16_t* w_lo) -> int {
        FY_TRY(gemv_bf16w(g, st));
        if (!w_lo) return FY_OK;
        g.W = w_lo; g.bias = nullptr; g.mode = GV_ADD;
        return gemv_bf16w(g, st);
    };
    FY_TRY(both(a, f->w_t0_lo));
    hipLaunchKernelGGL(silu_k, dim3(cdiv(n * D, 256)), dim3(256), 0, st, f->tsil, f->tsil, (long)n * D);
    a.W = f->w_t2; a.x = f->tsil; a.ldx = D; a.K = D; a.bias = f->b_t2; a.y = f->temb;
    FY_TRY(both(a, f->w_t2_lo));
    hipLaunchKernelGGL(silu_k, dim3(cdiv(n * D, 256)), dim3(256), 0, st, f->temb, f->tsil, (long)n * D);
    for (int i = 0; i < c.depth; ++i) {
        GemvArgs m;
        m.W = f->blk[i].wmod; m.x = f->tsil; m.ldx = D; m.R = n; m.N = 6 * D; m.K = D; m.bias = f->blk[i].bmod;
        m.y = f->mod + ((size_t)slot0 * c.depth + i) * 6 * D; m.ldy = c.depth * 6 * D;
        FY_TRY(both(m, f->blk[i].wmod_lo));
    }
    GemvArgs m;
    m.W = f->w_fin; m.x = f->tsil; m.ldx = D; m.R = n; m.N = 2 * D; m.K = D; m.bias = f->b_fin;
    m.y = f->fin + (size_t)slot0 * 2 * D; m.ldy = 2 * D;
    FY_TRY(both(m, f->w_fin_lo));
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

extern "C" int fy_flow_create(fy_flow** out, const fy_flow_config* cfg, const fy_tensor* weights, int32_t n_weights,
                              int32_t max_batch, int32_t max_frames, void* stream) {
    FY_CHECK(out && max_batch >= 1 && max_frames >= 2, FY_ERR_ARG, "fy_flow_create: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    fy_flow* f = new fy_flow();
    if (cfg) f->cfg = *cfg; else fy_flow_default_config(&f->cfg);
    const fy_flow_config& c = f->cfg;
    auto fail = [&](int code) { delete f; return code; };
    if (c.head_dim != 64 || c.dim % 256 != 0 || c.dim > 1024 || c.n_timesteps < 1 || c.n_timesteps > 32 || c.mel != 80 ||
        c.heads * c.head_dim != c.dim || (c.dim / c.conv_pos_groups) % 4 != 0) {
        fy_set_error("fy_flow_create: unsupported architecture (head_dim %d, dim %d, heads %d)", c.head_dim, c.dim, c.heads);
        return fail(FY_ERR_ARG);
    }
    f->max_batch = max_batch;
    f->Tmax = (max_frames + 1) & ~1;
    f->Nmax = f->Tmax / 2;
    Weights W;
    int rc = W.init(weights, n_weights);
    if (rc) return fail(rc);
#define TRYC(e) do { int _r = (e); if (_r) return fail(_r); } while (0)
#define GETW(var, name, ...) const float* var = W.get(name, {__VA_ARGS__}); if (!var) return fail(FY_ERR_WEIGHT)
    const int D = c.dim, C = c.mel, inner = c.heads * c.head_dim, FF = D * c.ff_mult;
    {
        GETW(ew, "input_embedding.weight", c.vocab, C);
        GETW(sw, "spk_embed_affine_layer.weight", C, c.spk_in);
        GETW(sb, "spk_embed_affine_layer.bias", C);
        TRYC(copy_f32(f, ew, (size_t)c.vocab * C, &f->emb_w, st));
        TRYC(copy_f32(f, sw, (size_t)C * c.spk_in, &f->spk_w, st));
        TRYC(copy_f32(f, sb, C, &f->spk_b, st));
        GETW(p1w, "pre_lookahead_layer.conv1.weight", c.pre_ch, C, c.pre_lookahead + 1);
        GETW(p1b, "pre_lookahead_layer.conv1.bias", c.pre_ch);
        GETW(p2w, "pre_lookahead_layer.conv2.weight", C, c.pre_ch, 3);
        GETW(p2b, "pre_lookahead_layer.conv2.bias", C);
        TRYC(conv_pack(f->pre1, p1w, nullptr, p1b, c.pre_ch, C, c.pre_lookahead + 1, 1, true, false, st));
        TRYC(conv_pack(f->pre2, p2w, nullptr, p2b, C, c.pre_ch, 3, 1, true, false, st));
    }
    const std::string E = "decoder.estimator.";
    {
        // does one bf16 plane per matrix lose anything?  (every estimator tensor with two or more dimensions: the linears and the
        // position convolutions; synthetic parity weights: no - a real flow.pt: yes, and FY_PRECISE then needs the lo planes)
        unsigned long long* cnt = nullptr;
        TRYC(f->pool.alloc(&cnt, (size_t)2));
        if (hipMemsetAsync(cnt, 0, 16, st) != hipSuccess) { fy_set_error("fy_flow_create: memset failed"); return fail(FY_ERR_HIP); }
        for (int i = 0; i < n_weights; ++i) {
            const fy_tensor& t = weights[i];
            if (!t.name || !t.data || t.ndim < 2 || strncmp(t.name, E.c_str(), E.size()) != 0) continue;
            size_t n = 1;
            for (int d = 0; d < t.ndim; ++d) n *= (size_t)t.shape[d];
            hipLaunchKernelGGL(flow_bf16_inexact_k, dim3(flow_grid(n)), dim3(256), 0, st, (const float*)t.data, n, cnt);
        }
        unsigned long long n_inexact = 0;
        if (hipMemcpyAsync(&n_inexact, cnt, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            fy_set_error("fy_flow_create: weight scan failed");
            return fail(FY_ERR_HIP);
        }
        f->planes = n_inexact ? 2 : 1;
        if (const char* e = getenv("FY_FLOW_WEIGHT_PLANES")) { if (atoi(e) == 1 || atoi(e) == 2) f->planes = atoi(e); }
        TRYC(f->pool.alloc(&f->ones, (size_t)D));
        hipLaunchKernelGGL(flow_fill_k, dim3(cdiv(D, 256)), dim3(256), 0, st, f->ones, 1.0f, D);
    }
    {
        GETW(t0w, E + "time_embed.time_mlp.0.weight", D, 256);
        GETW(t0b, E + "time_embed.time_mlp.0.bias", D);
        GETW(t2w, E + "time_embed.time_mlp.2.weight", D, D);
        GETW(t2b, E + "time_embed.time_mlp.2.bias", D);
        GETW(iw, E + "input_embed.proj.weight", D, 4 * C);
        GETW(ib, E + "input_embed.proj.bias", D);
        GETW(ow, E + "proj_out.weight", C, D);
        GETW(ob, E + "proj_out.bias", C);
        GETW(fw, E + "norm_out.linear.weight", 2 * D, D);
        GETW(fb, E + "norm_out.linear.bias", 2 * D);
        TRYC(to_packed(f, t0w, D, 256, &f->w_t0, st)); TRYC(copy_f32(f, t0b, D, &f->b_t0, st));
        TRYC(to_packed(f, t2w, D, D, &f->w_t2, st)); TRYC(copy_f32(f, t2b, D, &f->b_t2, st));
        TRYC(to_bf16(f, iw, (size_t)D * 4 * C, &f->w_in, st)); TRYC(copy_f32(f, ib, D, &f->b_in, st));
        TRYC(to_bf16(f, ow, (size_t)C * D, &f->w_out, st)); TRYC(copy_f32(f, ob, C, &f->b_out, st));
        TRYC(to_packed(f, fw, 2 * D, D, &f->w_fin, st)); TRYC(copy_f32(f, fb, 2 * D, &f->b_fin, st));
        TRYC(lo_plane(f, t0w, D, 256, true, &f->w_t0_lo, st)); TRYC(lo_plane(f, t2w, D, D, true, &f->w_t2_lo, st));
        TRYC(lo_plane(f, iw, D, 4 * C, false, &f->w_in_lo, st)); TRYC(lo_plane(f, ow, C, D, false, &f->w_out_lo, st));
        TRYC(lo_plane(f, fw, 2 * D, D, true, &f->w_fin_lo, st));
        const int cg = D / c.conv_pos_groups;
        GETW(c1w, E + "input_embed.conv_pos_embed.conv1.0.weight", D, cg, c.conv_pos_k);
        GETW(c1b, E + "input_embed.conv_pos_embed.conv1.0.bias", D);
        GETW(c2w, E + "input_embed.conv_pos_embed.conv2.0.weight", D, cg, c.conv_pos_k);
        GETW(c2b, E + "input_embed.conv_pos_embed.conv2.0.bias", D);
        const bool mf = cg % 32 == 0;
        TRYC(conv_pack(f->pos1, c1w, nullptr, c1b, D, D, c.conv_pos_k, c.conv_pos_groups, true, mf, st));
        TRYC(conv_pack(f->pos2, c2w, nullptr, c2b, D, D, c.conv_pos_k, c.conv_pos_groups, true, mf, st));
    }
    f->blk.resize(c.depth);
    struct FoldSrc { const float *qw, *kw, *vw, *f1w; };
    std::vector<FoldSrc> fsrc(c.depth);
    for (int i = 0; i < c.depth; ++i) {
        const std::string b = E + "transformer_blocks." + std::to_string(i) + ".";
        FlowBlockW& k = f->blk[i];
        GETW(mw, b + "attn_norm.linear.weight", 6 * D, D);
        GETW(mb, b + "attn_norm.linear.bias", 6 * D);
        GETW(qw, b + "attn.to_q.weight", inner, D); GETW(qb, b + "attn.to_q.bias", inner);
        GETW(kw, b + "attn.to_k.weight", inner, D); GETW(kb, b + "attn.to_k.bias", inner);
        GETW(vw, b + "attn.to_v.weight", inner, D); GETW(vb, b + "attn.to_v.bias", inner);
        GETW(ow, b + "attn.to_out.0.weight", D, inner); GETW(ob, b + "attn.to_out.0.bias", D);
        GETW(f1w, b + "ff.ff.0.0.weight", FF, D); GETW(f1b, b + "ff.ff.0.0.bias", FF);
        GETW(f2w, b + "ff.ff.2.weight", D, FF); GETW(f2b, b + "ff.ff.2.bias", D);
        fsrc[i] = FoldSrc{qw, kw, vw, f1w};
        TRYC(to_packed(f, mw, 6 * D, D, &k.wmod, st)); TRYC(copy_f32(f, mb, 6 * D, &k.bmod, st));
        TRYC(f->pool.alloc(&k.wqkv, (size_t)3 * inner * D));
        TRYC(f->pool.alloc(&k.bqkv, (size_t)3 * inner));
        const float* ws[3] = {qw, kw, vw};
        const float* bs[3] = {qb, kb, vb};
        for (int j = 0; j < 3; ++j) {
            TRYC(cast_f32_bf16(ws[j], k.wqkv + (size_t)j * inner * D, (size_t)inner * D, st));
            if (hipMemcpyAsync(k.bqkv + (size_t)j * inner, bs[j], inner * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
                fy_set_error("fy_flow_create: bias copy failed");
                return fail(FY_ERR_HIP);
            }
        }
        TRYC(to_bf16(f, ow, (size_t)D * inner, &k.wo, st)); TRYC(copy_f32(f, ob, D, &k.bo, st));
        TRYC(to_bf16(f, f1w, (size_t)FF * D, &k.w1, st)); TRYC(copy_f32(f, f1b, FF, &k.b1, st));
        TRYC(to_bf16(f, f2w, (size_t)D * FF, &k.w2, st)); TRYC(copy_f32(f, f2b, D, &k.b2, st));
        if (f->planes == 2) {
            TRYC(lo_plane(f, mw, 6 * D, D, true, &k.wmod_lo, st));
            TRYC(f->pool.alloc(&k.wqkv_lo, (size_t)3 * inner * D));
            float* tmp = nullptr;
            if (hipMalloc(&tmp, (size_t)inner * D * sizeof(float)) != hipSuccess) { fy_set_error("fy_flow_create: out of memory"); return fail(FY_ERR_HIP); }
            int rc3 = FY_OK;
            for (int j = 0; j < 3 && rc3 == FY_OK; ++j) {
                hipLaunchKernelGGL(flow_bf16_residual_k, dim3(flow_grid((size_t)inner * D)), dim3(256), 0, st, ws[j], tmp, (size_t)inner * D);
                rc3 = cast_f32_bf16(tmp, k.wqkv_lo + (size_t)j * inner * D, (size_t)inner * D, st);
            }
            (void)hipStreamSynchronize(st);
            (void)hipFree(tmp);
            TRYC(rc3);
            TRYC(lo_plane(f, ow, D, inner, false, &k.wo_lo, st));
            TRYC(lo_plane(f, f1w, FF, D, false, &k.w1_lo, st));
            TRYC(lo_plane(f, f2w, D, FF, false, &k.w2_lo, st));
        }
    }
    // activations
    const size_t B = max_batch, T = f->Tmax, N = f->Nmax, M = 2 * B * T;
    TRYC(f->pool.alloc(&f->tok_all, B * N)); TRYC(f->pool.alloc(&f->seq_len, 2 * B)); TRYC(f->pool.alloc(&f->blen, 5 * B));
    TRYC(f->pool.alloc(&f->emb, B * N * C)); TRYC(f->pool.alloc(&f->pre_a, B * N * c.pre_ch)); TRYC(f->pool.alloc(&f->mu_tok, B * N * C));
    TRYC(f->pool.alloc(&f->spks, B * C)); TRYC(f->pool.alloc(&f->x, B * T * C)); TRYC(f->pool.alloc(&f->mu, B * T * C));
    TRYC(f->pool.alloc(&f->cond, B * T * C)); TRYC(f->pool.alloc(&f->h, M * D)); TRYC(f->pool.alloc(&f->c1, M * D));
    TRYC(f->pool.alloc(&f->v, M * C)); TRYC(f->pool.alloc(&f->temb, (size_t)16 * D * 3)); TRYC(f->pool.alloc(&f->tsil, (size_t)16 * D));
    TRYC(f->pool.alloc(&f->a_in, M * 4 * C)); TRYC(f->pool.alloc(&f->xn, M * D)); TRYC(f->pool.alloc(&f->qkv, M * 3 * inner));
    TRYC(f->pool.alloc(&f->ao, M * inner)); TRYC(f->pool.alloc(&f->ff, M * FF));
    TRYC(f->pool.alloc(&f->hb, M * D)); TRYC(f->pool.alloc(&f->ln_slots, M * (D / 64)));
    TRYC(f->pool.alloc(&f->mod, (size_t)(c.n_timesteps + 1) * c.depth * 6 * D));
    TRYC(f->pool.alloc(&f->fin, (size_t)(c.n_timesteps + 1) * 2 * D));
    TRYC(f->pool.alloc(&f->rope, T * (c.head_dim / 2)));
    {
        std::vector<float2> tab(T * (c.head_dim / 2));
        for (size_t t = 0; t < T; ++t)
            for (int i = 0; i < c.head_dim / 2; ++i) {
                float inv = 1.0f / powf(10000.f, (float)(2 * i) / (float)c.head_dim);
                float ang = (float)t * inv;
                tab[t * (c.head_dim / 2) + i] = make_float2(cosf(ang), sinf(ang));
            }
        if (hipMemcpyAsync(f->rope, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) {
            fy_set_error("fy_flow_create: rope table upload failed");
            return fail(FY_ERR_HIP);
        }
    }
    // Euler schedule with the reference's running accumulation (flow_matching.py:87-88, 118-122)
    {
        const int n = c.n_timesteps;
        float t = c.t_span[0], dt = c.t_span[1] - c.t_span[0];
        for (int step = 1; step <= n; ++step) {
            f->t_of_step.push_back(t);
            f->dt_of_step.push_back(dt);
            t = t + dt;
            if (step < n) dt = c.t_span[step + 1] - t;
        }
        for (int s0 = 0; s0 < n; s0 += 8) TRYC(compute_mod(f, f->t_of_step.data() + s0, std::min(8, n - s0), s0, st));
    }
    // The folded LayerNorm-modulate tables (struct comment): W', u, bias' of the qkv and ff1 products for every (Euler step, block).
    // FY_FLOW_LN_FOLD=0: not built, every LayerNorm-modulate stays its own launch.  Needs the ring kernels' shapes.
    {
        const char* ev = getenv("FY_FLOW_LN_FOLD");
        const bool want = !(ev && atoi(ev) == 0) && D % 128 == 0 && inner % 128 == 0 && FF % 128 == 0 && D % 64 == 0;
        const size_t n_sb = (size_t)c.n_timesteps * c.depth, QN = (size_t)3 * inner;
        size_t free_b = 0, total_b = 0;
        const size_t need = n_sb * (QN + FF) * D * 2 + ((size_t)1 << 30);
        if (want && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > need) {
            TRYC(f->pool.alloc(&f->wq_fold, n_sb * QN * D)); TRYC(f->pool.alloc(&f->w1_fold, n_sb * (size_t)FF * D));
            TRYC(f->pool.alloc(&f->uq_fold, n_sb * QN)); TRYC(f->pool.alloc(&f->bq_fold, n_sb * QN));
            TRYC(f->pool.alloc(&f->u1_fold, n_sb * (size_t)FF)); TRYC(f->pool.alloc(&f->b1_fold, n_sb * (size_t)FF));
            for (int sidx = 0; sidx < c.n_timesteps; ++sidx)
                for (int i = 0; i < c.depth; ++i) {
                    const float* m = f->mod + ((size_t)sidx * c.depth + i) * 6 * D;       // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
                    const size_t sb = (size_t)sidx * c.depth + i;
                    const float* ws[3] = {fsrc[i].qw, fsrc[i].kw, fsrc[i].vw};
                    for (int j = 0; j < 3; ++j)
                        hipLaunchKernelGGL(fold_row_k, dim3(inner), dim3(256), 0, st, ws[j], f->blk[i].bqkv + (size_t)j * inner, m + D, m,
                                           f->wq_fold + (sb * QN + (size_t)j * inner) * D, f->uq_fold + sb * QN + (size_t)j * inner, f->bq_fold + sb * QN + (size_t)j * inner, D);
                    hipLaunchKernelGGL(fold_row_k, dim3(FF), dim3(256), 0, st, fsrc[i].f1w, f->blk[i].b1, m + 4 * D, m + 3 * D,
                                       f->w1_fold + sb * (size_t)FF * D, f->u1_fold + sb * FF, f->b1_fold + sb * FF, D);
                }
            if (hipGetLastError() != hipSuccess) { fy_set_error("fy_flow_create: the fold kernels failed to launch"); return fail(FY_ERR_HIP); }
            f->fold = true;
        }
    }
#undef GETW
#undef TRYC
    if (hipStreamSynchronize(st) != hipSuccess) { fy_set_error("fy_flow_create: stream sync failed"); return fail(FY_ERR_HIP); }
    *out = f;
    return FY_OK;
}

extern "C" void fy_flow_destroy(fy_flow* f) { delete f; }

extern "C" int fy_flow_weight_planes(const fy_flow* f) { return f ? f->planes : 0; }

// lo planes of the fp32-class mode, sized for the handle's capacity, on first use
static int ensure_precise(fy_flow* f) {
    const fy_flow_config& c = f->cfg;
    const size_t M = (size_t)2 * f->max_batch * f->Tmax, D = c.dim, inner = (size_t)c.heads * c.head_dim, FF = D * c.ff_mult;
    // buffer by buffer, so a call after a partial failure (out of memory) allocates only what is still missing
    if (!f->a_in_lo) FY_TRY(f->pool.alloc(&f->a_in_lo, M * 4 * c.mel));
    if (!f->qkv_lo) FY_TRY(f->pool.alloc(&f->qkv_lo, M * 3 * inner));
    if (!f->ao_lo) FY_TRY(f->pool.alloc(&f->ao_lo, M * inner));
    if (!f->ff_lo) FY_TRY(f->pool.alloc(&f->ff_lo, M * FF));
    if (!f->xn_lo) FY_TRY(f->pool.alloc(&f->xn_lo, M * D));
    return FY_OK;
}

// ---- estimator ----------------------------------------------------------------------------------------
// One DiT.forward over nseq sequences whose bf16 input rows are in f->a_in; result rows (fp32, C wide) in f->v.
static int dit_forward(fy_flow* f, int nseq, int Tmax, int slot, bool streaming, uint32_t flags, hipStream_t st) {
    const fy_flow_config& c = f->cfg;
    const int D = c.dim, C = c.mel, inner = c.heads * c.head_dim, FF = D * c.ff_mult;
    const int M = nseq * Tmax;
    // FY_PRECISE: every GEMM operand travels as x = hi + lo, two bf16 planes written by its producer (16 mantissa bits; the weights
    // are bf16-exact in the tests), two MFMAs per fragment into an fp32 accumulator on the same LDS-DMA ring kernel as the default
    // mode (gemm_split); attention keeps the three leading terms of each split product, softmax in fp32 (dit_attention_split); GELU
    // uses the exact tanh: an fp32-class estimator that meets the reference's own bar for swapping the estimator (rtol 1e-2 /
    // atol 1e-4, cosyvoice/bin/export_onnx.py:109) at ~2x the default mode's MFMA work.
    const bool pr = (flags & FY_PRECISE) != 0;
    GemmEpi e;
    e.bias = f->b_in; e.out = f->h; e.out_bf16 = 0; e.ldc = D;
    if (pr) { e.a_lo = f->a_in_lo; e.w_lo = f->w_in_lo; e.gate_ones = f->ones; FY_TRY(gemm_split(f->a_in, 4 * C, f->w_in, M, D, 4 * C, e, st)); }
    else FY_TRY(gemm_bf16(f->a_in, 4 * C, f->w_in, M, D, 4 * C, e, st));
    {   // x = conv_pos_embed(x) + x, modules.py:129-144 (causal, grouped, Mish)
        ConvDesc d;
        memset(&d, 0, sizeof(d));
        d.B = nseq; d.dil = 1; d.stride = 1; d.up = 1; d.groups = c.conv_pos_groups; d.out_scale = 1.f;
        d.x = f->h; d.x_bs = (long)Tmax * D; d.x_ld = D; d.L_in = Tmax; d.in_len = f->seq_len;
        d.y = f->c1; d.y_bs = (long)Tmax * D; d.y_ld = D; d.L_out = Tmax; d.out_len = f->seq_len;
        d.Cin = D; d.Cout = D; d.KW = c.conv_pos_k; d.pad_left = c.conv_pos_k - 1; d.bias = f->pos1.bias; d.post_act = ACT_MISH;
        // (FY_PRECISE on weights that are not bf16-representable: the exact fp32 kernel, which keeps the fp32 weights)
        const bool mf = f->pos1.w_mfma != nullptr && !(flags & FY_DIRECT) && !(pr && f->planes == 2);
        FY_TRY(mf ? conv1d_bf16_mfma(d, f->pos1, (flags & FY_PRECISE) != 0, st) : conv1d_f32_direct(d, f->pos1, st));
        d.x = f->c1; d.y = f->h; d.bias = f->pos2.bias;
        d.add_resid = 1; d.resid = f->h; d.r_bs = (long)Tmax * D; d.r_ld = D;
        FY_TRY(mf ? conv1d_bf16_mfma(d, f->pos2, (flags & FY_PRECISE) != 0, st) : conv1d_f32_direct(d, f->pos2, st));
    }
    const float* modbase = f->mod + (size_t)slot * c.depth * 6 * D;
    bf16_t* const xn_lo = pr ? f->xn_lo : nullptr;
    // LayerNorm-modulate folded across the qkv / ff1 products (round 5).  xn = LN(h) (1 + s) + b feeds ONE product each time, and s, b
    // depend on (Euler step, block) only, so  xn W^T + bias = rstd (h W'^T - mean u) + bias'  with W' = W (1 + s), u = sum_k W',
    // bias' = bias + W b (made at create).  The product that WRITES h (out-projection, ff2: the gated residual) also leaves bf16(h)
    // and the rows' partial (sum, sum of squares) per 64-column tile; every workgroup of the consuming product turns the slots of its
    // own rows into (mean, rstd) while its first operand stage is in flight (as a launch of its own that reduction took 4.4 us - half
    // of what the fold saves; as the tail of the producer's last workgroup per row panel, behind an agent-scope release, 13 us), reads
    // bf16(h) against W' and finishes in its epilogue.  That replaces a launch that reads the fp32 stream and writes
    // the bf16 operand (39.5 MB, 9.3 us at M = 6400, 440 per batch: 8 % of the step's kernel time) by 13 MB more in the producer's
    // epilogue + ~2 us.  Accuracy: the operand is bf16(h) instead of bf16(normalised h) - the same relative rounding as long as a
    // row's mean is small against its spread (tests/micro/ln_fold_error_sim.py: estimator error 1.1e-2 against 0.9e-2).
    // Only on the Euler schedule's own steps (the tables are per step), the default arithmetic and the whole-sequence mask: the
    // estimator entry (any t), FY_PRECISE and the streaming forms (whose incremental chunks must equal the full pass bit for bit)
    // keep the separate launch.  The first LayerNorm of a call reads what the position convolution wrote and the last one feeds the
    // 80-column output projection: those two stay launches as well.
    const bool fold = f->fold && !pr && !streaming && slot < c.n_timesteps && !(flags & FY_DIRECT) && D % 64 == 0;
    const size_t sb0 = (size_t)slot * c.depth;
    auto producer = [&](GemmEpi& e) { e.h_bf16 = f->hb; e.ln_slots = f->ln_slots; };
    for (int i = 0; i < c.depth; ++i) {
        const FlowBlockW& k = f->blk[i];
        const float* m = modbase + (size_t)i * 6 * D;       // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
        const bool fold_q = fold && i > 0;                  // block 0's h comes from the position convolution
        if (!fold_q) hipLaunchKernelGGL(ln_mod_k, dim3(cdiv(M, 4)), dim3(256), 0, st, f->h, m + D, m, f->xn, xn_lo, M, D);
        GemmEpi q;
        q.bias = k.bqkv; q.out = f->qkv; q.out_bf16 = 1; q.ldc = 3 * inner;
        // x-transformers rotary embedding (head 0 of q and k only, modules.py:368-373) in the projection's epilogue, on the fp32
        // sums before they are rounded to bf16 (or split into the two planes)
        q.rope = f->rope; q.rope_T = Tmax; q.rope_half = c.head_dim / 2; q.rope_stride = inner;
        if (pr) {
            q.a_lo = f->xn_lo; q.out_lo = f->qkv_lo; q.w_lo = k.wqkv_lo;
            FY_TRY(gemm_split(f->xn, D, k.wqkv, M, 3 * inner, D, q, st));
            FY_TRY(dit_attention_split(f->qkv, f->qkv_lo, f->ao, f->ao_lo, f->seq_len, nseq, Tmax, c.heads, streaming ? c.static_chunk : 0, st));
        } else {
            if (fold_q) {
                const size_t sb = sb0 + i, QN = (size_t)3 * inner;
                q.bias = f->bq_fold + sb * QN; q.ln_rows_slots = f->ln_slots; q.ln_dim = D; q.ln_u = f->uq_fold + sb * QN;
                FY_TRY(gemm_bf16(f->hb, D, f->wq_fold + sb * QN * D, M, 3 * inner, D, q, st));
            } else FY_TRY(gemm_bf16(f->xn, D, k.wqkv, M, 3 * inner, D, q, st));
            FY_TRY(dit_attention(f->qkv, f->ao, f->seq_len, nseq, Tmax, c.heads, streaming ? c.static_chunk : 0, st));
        }
        GemmEpi o;
        o.mode = EPI_GATE_RESID; o.bias = k.bo; o.resid = f->h; o.gate = m + 2 * D; o.ldc = D;
        if (fold) producer(o);
        if (pr) { o.a_lo = f->ao_lo; o.w_lo = k.wo_lo; o.gate_ones = f->ones; FY_TRY(gemm_split(f->ao, inner, k.wo, M, D, inner, o, st)); }
        else FY_TRY(gemm_bf16(f->ao, inner, k.wo, M, D, inner, o, st));
        GemmEpi g;
        g.bias = k.b1; g.act = ACT_GELU_TANH; g.out = f->ff; g.out_bf16 = 1; g.ldc = FF;
        if (fold) {
            const size_t sb = sb0 + i;
            g.bias = f->b1_fold + sb * FF; g.ln_rows_slots = f->ln_slots; g.ln_dim = D; g.ln_u = f->u1_fold + sb * FF;
            FY_TRY(gemm_bf16(f->hb, D, f->w1_fold + sb * (size_t)FF * D, M, FF, D, g, st));
        } else {
            hipLaunchKernelGGL(ln_mod_k, dim3(cdiv(M, 4)), dim3(256), 0, st, f->h, m + 4 * D, m + 3 * D, f->xn, xn_lo, M, D);
            if (pr) { g.a_lo = f->xn_lo; g.out_lo = f->ff_lo; g.w_lo = k.w1_lo; FY_TRY(gemm_split(f->xn, D, k.w1, M, FF, D, g, st)); }
            else FY_TRY(gemm_bf16(f->xn, D, k.w1, M, FF, D, g, st));
        }
        GemmEpi r;
        r.mode = EPI_GATE_RESID; r.bias = k.b2; r.resid = f->h; r.gate = m + 5 * D; r.ldc = D;
        if (fold && i + 1 < c.depth) producer(r);
        if (pr) { r.a_lo = f->ff_lo; r.w_lo = k.w2_lo; r.gate_ones = f->ones; FY_TRY(gemm_split(f->ff, FF, k.w2, M, D, FF, r, st)); }
        else FY_TRY(gemm_bf16(f->ff, FF, k.w2, M, D, FF, r, st));
    }
    const float* fn = f->fin + (size_t)slot * 2 * D;            // (scale, shift), modules.py:261
    hipLaunchKernelGGL(ln_mod_k, dim3(cdiv(M, 4)), dim3(256), 0, st, f->h, fn, fn + D, f->xn, xn_lo, M, D);
    GemmEpi p;
    p.bias = f->b_out; p.out = f->v; p.out_bf16 = 0; p.ldc = C;
    if (pr) { p.a_lo = f->xn_lo; p.w_lo = f->w_out_lo; p.gate_ones = f->ones; FY_TRY(gemm_split(f->xn, D, f->w_out, M, C, D, p, st)); }
    else FY_TRY(gemm_bf16(f->xn, D, f->w_out, M, C, D, p, st));
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// the caches of the incremental streaming path, on first use: 2 x Tmax x 3 inner bf16 per (step, block) - 2.7 MB per frame row
// of capacity at full size - and the ODE states; an allocation failure just means "not incremental"
static int inc_alloc(fy_flow* f) {
    if (f->inc_qkv && f->inc_x) return FY_OK;
    const fy_flow_config& c = f->cfg;
    const size_t nq = (size_t)c.n_timesteps * c.depth * 2 * f->Tmax * 3 * c.heads * c.head_dim, nx = (size_t)(c.n_timesteps + 1) * f->Tmax * c.mel;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < nq * 2 + nx * 4 + ((size_t)1 << 30)) return FY_ERR_HIP;
    if (!f->inc_qkv && f->pool.alloc(&f->inc_qkv, nq) != FY_OK) return FY_ERR_HIP;
    if (!f->inc_x && f->pool.alloc(&f->inc_x, nx) != FY_OK) return FY_ERR_HIP;
    f->inc_T = 0;
    return FY_OK;
}

// The incremental form of dit_forward for a streaming chunk at batch 1 (two sequences: conditional and CFG branch).  Layout: the
// two sequences' rows INTERLEAVED - frame t of sequence s is row 2 t + s of every buffer - so the new frames [T0, T) of both
// sequences are ONE contiguous run of rows and every product of a block stays one launch.  The input projection and the causal
// position convs run over all T frames (cheap, and they need the frames to the left), then ONLY the rows [2 T0, 2 T) go through
// the blocks, while attention reads the keys / values of the earlier frames from the (step, block) slice of inc_qkv, into which
// the qkv product of the new rows writes directly.  Under the chunk mask with T0 and T on chunk boundaries the frames before T0
// see nothing of the new ones, so their keys, values and outputs are what the previous call computed: same bits as the full
// recompute (tests/test_stream_gpu.py).
static int dit_forward_inc(fy_flow* f, int step, int T0, int T, hipStream_t st) {
    const fy_flow_config& c = f->cfg;
    const int D = c.dim, C = c.mel, inner = c.heads * c.head_dim, FF = D * c.ff_mult, nseq = 2, Mn = nseq * (T - T0);
    const size_t r0 = (size_t)nseq * T0;                                       // first new row
    GemmEpi e;
    e.bias = f->b_in; e.out = f->h; e.out_bf16 = 0; e.ldc = D;
    FY_TRY(gemm_bf16(f->a_in, 4 * C, f->w_in, nseq * T, D, 4 * C, e, st));
    {
        ConvDesc d;
        memset(&d, 0, sizeof(d));
        d.B = nseq; d.dil = 1; d.stride = 1; d.up = 1; d.groups = c.conv_pos_groups; d.out_scale = 1.f;
        d.x = f->h; d.x_bs = D; d.x_ld = nseq * D; d.L_in = T; d.in_len = f->seq_len;          // sequence s = every second row from row s
        d.y = f->c1; d.y_bs = D; d.y_ld = nseq * D; d.L_out = T; d.out_len = f->seq_len;
        d.Cin = D; d.Cout = D; d.KW = c.conv_pos_k; d.pad_left = c.conv_pos_k - 1; d.bias = f->pos1.bias; d.post_act = ACT_MISH;
        const bool mf = f->pos1.w_mfma != nullptr;
        FY_TRY(mf ? conv1d_bf16_mfma(d, f->pos1, false, st) : conv1d_f32_direct(d, f->pos1, st));
        d.x = f->c1; d.y = f->h; d.bias = f->pos2.bias;
        d.add_resid = 1; d.resid = f->h; d.r_bs = D; d.r_ld = nseq * D;
        FY_TRY(mf ? conv1d_bf16_mfma(d, f->pos2, false, st) : conv1d_f32_direct(d, f->pos2, st));
    }
    const float* modbase = f->mod + (size_t)step * c.depth * 6 * D;
    const size_t slice = (size_t)nseq * f->Tmax * 3 * inner;                 // one (step, block) slice of inc_qkv
    float* h = f->h + r0 * D;
    bf16_t* xn = f->xn + r0 * D;
    for (int i = 0; i < c.depth; ++i) {
        const FlowBlockW& k = f->blk[i];
        const float* m = modbase + (size_t)i * 6 * D;
        bf16_t* qkv = f->inc_qkv + ((size_t)step * c.depth + i) * slice;
        hipLaunchKernelGGL(ln_mod_k, dim3(cdiv(Mn, 4)), dim3(256), 0, st, h, m + D, m, xn, (bf16_t*)nullptr, Mn, D);
        GemmEpi q;
        q.bias = k.bqkv; q.out = qkv + r0 * 3 * inner; q.out_bf16 = 1; q.ldc = 3 * inner;
        q.rope = f->rope + (size_t)T0 * (c.head_dim / 2); q.rope_T = T - T0; q.rope_div = nseq; q.rope_half = c.head_dim / 2; q.rope_stride = inner;
        FY_TRY(gemm_bf16(xn, D, k.wqkv, Mn, 3 * inner, D, q, st));
        FY_TRY(dit_attention(qkv, f->ao, f->seq_len, nseq, T, c.heads, c.static_chunk, st, T0, true));
        GemmEpi o;
        o.mode = EPI_GATE_RESID; o.bias = k.bo; o.resid = h; o.gate = m + 2 * D; o.ldc = D;
        FY_TRY(gemm_bf16(f->ao + r0 * inner, inner, k.wo, Mn, D, inner, o, st));
        hipLaunchKernelGGL(ln_mod_k, dim3(cdiv(Mn, 4)), dim3(256), 0, st, h, m + 4 * D, m + 3 * D, xn, (bf16_t*)nullptr, Mn, D);
        GemmEpi g;
        g.bias = k.b1; g.act = ACT_GELU_TANH; g.out = f->ff + r0 * FF; g.out_bf16 = 1; g.ldc = FF;
        FY_TRY(gemm_bf16(xn, D, k.w1, Mn, FF, D, g, st));
        GemmEpi r;
        r.mode = EPI_GATE_RESID; r.bias = k.b2; r.resid = h; r.gate = m + 5 * D; r.ldc = D;
        FY_TRY(gemm_bf16(f->ff + r0 * FF, FF, k.w2, Mn, D, FF, r, st));
    }
    const float* fn = f->fin + (size_t)step * 2 * D;
    hipLaunchKernelGGL(ln_mod_k, dim3(cdiv(Mn, 4)), dim3(256), 0, st, h, fn, fn + D, xn, (bf16_t*)nullptr, Mn, D);
    GemmEpi p;
    p.bias = f->b_out; p.out = f->v + r0 * C; p.out_bf16 = 0; p.ldc = C;
    FY_TRY(gemm_bf16(xn, D, f->w_out, Mn, C, D, p, st));
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

extern "C" int fy_flow_stream_rows(const fy_flow* f) { return f ? f->inc_T : -1; }

extern "C" int fy_flow_stream_reset(fy_flow* f) {
    FY_CHECK(f, FY_ERR_ARG, "fy_flow_stream_reset: null handle");
    f->inc_T = 0;
    return FY_OK;
}

extern "C" int fy_flow_infer(fy_flow* f, const int32_t* token, int32_t tok_ld, const int32_t* n_token, const int32_t* prompt_token,
                             int32_t ptok_ld, const int32_t* n_prompt, const float* prompt_feat, int32_t pfeat_rows, const int32_t* n_pfeat,
                             const float* embedding, const float* rand_noise, int32_t noise_ld, int32_t B, float* mel, int32_t mel_frames,
                             uint32_t flags, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    FY_CHECK(f && token && n_token && n_prompt && n_pfeat && embedding && rand_noise && mel, FY_ERR_ARG, "fy_flow_infer: null argument");
    FY_CHECK(B >= 1 && B <= f->max_batch, FY_ERR_ARG, "fy_flow_infer: batch %d outside [1, %d]", B, f->max_batch);
    const fy_flow_config& c = f->cfg;
    const int C = c.mel, mb = f->max_batch;
    std::vector<int> lens(5 * mb, 0), sl(2 * mb, 0), npr(mb, 0);
    // a streaming chunk (finalize=False, flow.py:382-383): the last pre_lookahead tokens are PreLookahead's context, not output
    const int look = (flags & FY_NO_FINALIZE) ? c.pre_lookahead : 0;
    int Tmax = 0, Nmax = 0;
    for (int b = 0; b < B; ++b) {
        int n_all = n_token[b] + n_prompt[b], T = 2 * (n_all - look);
        FY_CHECK(n_token[b] >= 1 + look && n_prompt[b] >= 0 && n_pfeat[b] >= 0 && n_pfeat[b] <= T && T <= f->Tmax && T <= noise_ld &&
                     T - n_pfeat[b] <= mel_frames && n_token[b] <= tok_ld && n_prompt[b] <= ptok_ld && n_pfeat[b] <= pfeat_rows,
                 FY_ERR_ARG, "fy_flow_infer: utterance %d has inconsistent lengths (tokens %d, prompt %d, prompt frames %d, max frames %d)",
                 b, n_token[b], n_prompt[b], n_pfeat[b], f->Tmax);
        lens[0 * mb + b] = n_all; lens[1 * mb + b] = T; lens[2 * mb + b] = n_pfeat[b]; lens[3 * mb + b] = n_prompt[b];
        lens[4 * mb + b] = n_all - look;
        sl[2 * b] = sl[2 * b + 1] = T;
        Tmax = std::max(Tmax, T); Nmax = std::max(Nmax, n_all);
    }
    // by kernel argument, not by copy + synchronisation: the call can be enqueued behind a previous one that is still running
    FY_TRY(upload_ints(f->blen, lens.data(), (int)lens.size(), st));
    FY_TRY(upload_ints(f->seq_len, sl.data(), (int)sl.size(), st));
    const int *d_nall = f->blen, *d_T = f->blen + mb, *d_pmel = f->blen + 2 * mb, *d_np = f->blen + 3 * mb, *d_nout = f->blen + 4 * mb;
    // front: speaker projection, token embedding, PreLookahead (upsample_encoder.py:82-103)
    hipLaunchKernelGGL(spk_k, dim3(B), dim3(256), 0, st, embedding, f->spk_w, f->spk_b, f->spks, c.spk_in, C);
    hipLaunchKernelGGL(tok_embed_k, dim3(Nmax, B), dim3(128), 0, st, prompt_token, ptok_ld, token, tok_ld, d_np, d_nall, f->tok_all,
                       f->emb_w, f->emb, Nmax, C, c.vocab);
    {
        ConvDesc d;
        memset(&d, 0, sizeof(d));
        d.B = B; d.dil = 1; d.stride = 1; d.up = 1; d.groups = 1; d.out_scale = 1.f;
        d.x = f->emb; d.x_bs = (long)Nmax * C; d.x_ld = C; d.L_in = Nmax; d.in_len = d_nall;
        d.y = f->pre_a; d.y_bs = (long)Nmax * c.pre_ch; d.y_ld = c.pre_ch; d.L_out = Nmax; d.out_len = d_nout;
        d.Cin = C; d.Cout = c.pre_ch; d.KW = c.pre_lookahead + 1; d.pad_left = 0; d.bias = f->pre1.bias;
        d.post_act = ACT_LEAKY; d.post_slope = 0.01f;
        FY_TRY(conv1d_f32_mfma(d, f->pre1, st));
        ConvDesc e;
        memset(&e, 0, sizeof(e));
        e.B = B; e.dil = 1; e.stride = 1; e.up = 1; e.groups = 1; e.out_scale = 1.f;
        e.x = f->pre_a; e.x_bs = (long)Nmax * c.pre_ch; e.x_ld = c.pre_ch; e.L_in = Nmax; e.in_len = d_nout;
        e.y = f->mu_tok; e.y_bs = (long)Nmax * C; e.y_ld = C; e.L_out = Nmax; e.out_len = d_nout;
        e.Cin = c.pre_ch; e.Cout = C; e.KW = 3; e.pad_left = 2; e.bias = f->pre2.bias;
        e.add_resid = 1; e.resid = f->emb; e.r_bs = (long)Nmax * C; e.r_ld = C;
        FY_TRY(conv1d_f32_mfma(e, f->pre2, st));
    }
    hipLaunchKernelGGL(flow_setup_k, dim3(Tmax, B), dim3(128), 0, st, f->mu_tok, prompt_feat, (long)pfeat_rows * C, rand_noise, noise_ld, d_T,
                       d_pmel, f->mu, f->cond, f->x, Tmax, Nmax, C);
    if (flags & FY_PRECISE) FY_TRY(ensure_precise(f));
    // FY_INCREMENTAL (a streaming chunk at batch 1, bf16 mode): T0 = rows the previous call of this stream left (0: none, or not on
    // a chunk boundary - everything is computed, and kept for the next call)
    const bool inc = (flags & FY_INCREMENTAL) && B == 1 && (flags & FY_STREAMING) && (flags & FY_NO_FINALIZE) && !(flags & (FY_PRECISE | FY_DIRECT)) &&
                     c.static_chunk > 0 && Tmax % c.static_chunk == 0 && inc_alloc(f) == FY_OK;
    if (inc) {
        int T0 = f->inc_T;
        if (T0 % c.static_chunk != 0 || T0 >= Tmax) T0 = 0;
        // a call extends the stream only with the same prompt and at least the tokens it had (a caller that forgot fy_flow_stream_reset
        // between two utterances would otherwise attend the previous utterance's keys): anything else starts over
        if (n_prompt[0] != f->inc_np || n_pfeat[0] != f->inc_npf || n_token[0] < f->inc_ntok) T0 = 0;
        f->inc_np = n_prompt[0]; f->inc_npf = n_pfeat[0]; f->inc_ntok = n_token[0];
        float* xs = f->inc_x;
        const size_t xslice = (size_t)f->Tmax * C;
        for (int step = 0; step < c.n_timesteps; ++step) {
            // the state at this step: the earlier rows as the previous calls left them, the new rows as they evolve in f->x
            HIP_TRY(hipMemcpyAsync(xs + step * xslice + (size_t)T0 * C, f->x + (size_t)T0 * C, (size_t)(Tmax - T0) * C * sizeof(float), hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(dit_assemble_k, dim3(Tmax, 2), dim3(4 * C), 0, st, xs + step * xslice, f->cond, f->mu, f->spks, d_T, f->a_in, (bf16_t*)nullptr, Tmax, C, 1, 2);
            FY_TRY(dit_forward_inc(f, step, T0, Tmax, st));
            hipLaunchKernelGGL(euler_k, dim3(Tmax - T0, 1), dim3(128), 0, st, f->x, f->v, d_T, Tmax, C, f->dt_of_step[step], c.cfg_rate, T0, 1);
        }
        HIP_TRY(hipMemcpyAsync(xs + c.n_timesteps * xslice + (size_t)T0 * C, f->x + (size_t)T0 * C, (size_t)(Tmax - T0) * C * sizeof(float), hipMemcpyDeviceToDevice, st));
        f->inc_T = Tmax;
        const int F = 2 * (n_token[0] + n_prompt[0] - look) - n_pfeat[0];
        FY_TRY(transpose_blc_to_bcl(xs + c.n_timesteps * xslice + (size_t)n_pfeat[0] * C, mel, 1, C, F, 0, C, 0, mel_frames, st));
        HIP_TRY(hipGetLastError());
        return FY_OK;
    }
    if (flags & FY_INCREMENTAL) f->inc_T = 0;             // a call that could not be incremental breaks the stream
    for (int step = 0; step < c.n_timesteps; ++step) {
        hipLaunchKernelGGL(dit_assemble_k, dim3(Tmax, 2 * B), dim3(4 * C), 0, st, f->x, f->cond, f->mu, f->spks, d_T, f->a_in,
                           (flags & FY_PRECISE) ? f->a_in_lo : nullptr, Tmax, C, Tmax, 1);
        FY_TRY(dit_forward(f, 2 * B, Tmax, step, (flags & FY_STREAMING) != 0, flags, st));
        hipLaunchKernelGGL(euler_k, dim3(Tmax, B), dim3(128), 0, st, f->x, f->v, d_T, Tmax, C, f->dt_of_step[step], c.cfg_rate, 0, 0);
    }
    // mel[b] = x[b][pmel:, :]^T in the reference's (B, 80, F) layout, flow.py:401
    for (int b = 0; b < B; ++b) {
        int F = 2 * (n_token[b] + n_prompt[b] - look) - n_pfeat[b];       // mel_len2 = h.shape[1] - prompt_feat.shape[1], flow.py:384
        FY_TRY(transpose_blc_to_bcl(f->x + ((long)b * Tmax + n_pfeat[b]) * C, mel + (long)b * C * mel_frames, 1, C, F, 0, C, 0, mel_frames, st));
    }
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// The reference's estimator hand-off (flow_matching.py:126-153): raw device pointers of contiguous
// x (B2,80,T), mask (B2,1,T), mu (B2,80,T), t (B2), spks (B2,80), cond (B2,80,T); result written into x.
// ---- speed change between flow decoder and vocoder (cli/model.py:435-437) ------------------------------------------
__global__ void mel_speed_k(const float* __restrict__ src, float* __restrict__ dst, int F_in, int F_out, float scale) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= F_out) return;
    // aten area_pixel_compute_source_index (align_corners = false), in fp32 like aten's CPU / CUDA kernels
    float x = scale * ((float)i + 0.5f) - 0.5f;
    x = x < 0.f ? 0.f : x;
    const int i0 = min((int)x, F_in - 1), i1 = i0 + (i0 < F_in - 1 ? 1 : 0);
    const float l1 = x - (float)i0, l0 = 1.f - l1;
    const float* r = src + (long)blockIdx.y * F_in;
    dst[(long)blockIdx.y * F_out + i] = l0 * r[i0] + l1 * r[i1];
}
extern "C" int fy_mel_speed(const float* mel, int32_t rows, int32_t F_in, float* out, int32_t F_out, void* stream) {
    FY_CHECK(mel && out && rows >= 1 && rows <= 65535 && F_in >= 1 && F_out >= 1, FY_ERR_ARG, "fy_mel_speed: bad arguments");
    hipLaunchKernelGGL(mel_speed_k, dim3(cdiv(F_out, 256), rows), dim3(256), 0, (hipStream_t)stream, mel, out, F_in, F_out, (float)F_in / (float)F_out);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

extern "C" int fy_dit_estimator(fy_flow* f, float* x, const float* mask, const float* mu, const float* t, const float* spks,
                                const float* cond, int32_t T, int32_t B2, uint32_t flags, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    FY_CHECK(f && x && mu && t && spks && cond, FY_ERR_ARG, "fy_dit_estimator: null argument");
    FY_CHECK(B2 >= 1 && B2 <= 2 * f->max_batch && T >= 1 && T <= f->Tmax, FY_ERR_ARG, "fy_dit_estimator: (B2 %d, T %d) outside the handle's limits", B2, T);
    const fy_flow_config& c = f->cfg;
    const int C = c.mel;
    std::vector<float> th(B2), mh;
    HIP_TRY(hipMemcpyAsync(th.data(), t, B2 * sizeof(float), hipMemcpyDeviceToHost, st));
    std::vector<int> sl(2 * f->max_batch, 0);
    if (mask) {
        mh.resize((size_t)B2 * T);
        HIP_TRY(hipMemcpyAsync(mh.data(), mask, mh.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    for (int s = 0; s < B2; ++s) {
        FY_CHECK(th[s] == th[0], FY_ERR_ARG, "fy_dit_estimator: all rows must share one timestep (as solve_euler passes it)");
        int len = T;
        if (mask) { len = 0; while (len < T && mh[(size_t)s * T + len] != 0.f) ++len; }
        FY_CHECK(len >= 1, FY_ERR_ARG, "fy_dit_estimator: empty mask row %d", s);
        sl[s] = len;
    }
    HIP_TRY(hipMemcpyAsync(f->seq_len, sl.data(), sl.size() * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    FY_TRY(compute_mod(f, th.data(), 1, c.n_timesteps, st));
    // stage the four inputs channels-last in h (fp32 scratch), then pack the bf16 operand rows
    float* sx = f->h;
    float* sc = sx + (size_t)B2 * T * C;
    float* sm = sc + (size_t)B2 * T * C;
    FY_TRY(transpose_bcl_to_blc(x, sx, B2, C, T, (long)C * T, (long)T * C, C, st));
    FY_TRY(transpose_bcl_to_blc(cond, sc, B2, C, T, (long)C * T, (long)T * C, C, st));
    FY_TRY(transpose_bcl_to_blc(mu, sm, B2, C, T, (long)C * T, (long)T * C, C, st));
    // a_in rows: sequence s uses its own (x, cond, mu, spks): reuse dit_assemble_k per sequence with cfg = 0
    // by viewing each sequence as "batch b = s" with a stride of 2 sequences
    if (flags & FY_PRECISE) FY_TRY(ensure_precise(f));
    {
        // pack directly: one launch per sequence keeps the kernel unchanged
        for (int s = 0; s < B2; ++s) {
            hipLaunchKernelGGL(dit_assemble_k, dim3(T, 1), dim3(4 * C), 0, st, sx + (size_t)s * T * C, sc + (size_t)s * T * C,
                               sm + (size_t)s * T * C, spks + (size_t)s * C, f->seq_len + s, f->a_in + (size_t)s * T * 4 * C,
                               (flags & FY_PRECISE) ? f->a_in_lo + (size_t)s * T * 4 * C : nullptr, T, C, T, 1);
        }
    }
    // a_in was written by rows of T (not f->Tmax): run the estimator with Tmax = T; h is overwritten only after a_in is complete
    HIP_TRY(hipStreamSynchronize(st));
    FY_TRY(dit_forward(f, B2, T, c.n_timesteps, (flags & FY_STREAMING) != 0, flags, st));
    FY_TRY(transpose_blc_to_bcl(f->v, x, B2, C, T, (long)T * C, C, (long)C * T, T, st));
    return FY_OK;
}
