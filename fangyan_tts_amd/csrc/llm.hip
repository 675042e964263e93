// Speech-token language model engine: CosyVoice3LM.inference -> Qwen2LM.inference_wrapper ->
// Qwen2Encoder.forward_one_step, CosyVoice/cosyvoice/llm/llm.py:713-748, 511-525, 246-258, for a
// batch of sequences, plus CosyVoiceModel.llm_job's silent-token filter (cli/model.py:101-129).
//
// The Qwen2 body (third-party `transformers`, pinned 4.51.3) is restated as kernels: RMSNorm,
// q/k/v projections with bias, split-half RoPE (theta 1e6), grouped-query causal attention over an
// fp32 KV cache, SwiGLU.  Weights are bf16 in HBM; activations, accumulation, the cache and the
// softmaxes are fp32, so greedy token ids track an fp32 reference on the same (bf16-representable)
// weights.  Prefill rows of all sequences are packed; decode runs one row per sequence with all
// per-sequence state (positions, counts, stop flags) on the device, so a step needs no host sync.
#include "attn.h"
#include "gemm.h"
#include "gemv32.h"
#include "llm_decode.h"
#include "llm_decode32.h"
#include "runtime.h"
#include <algorithm>
#include <math.h>

struct LlmLayerW {
    bf16_t *wqkv, *wo, *wgu, *wd;
    float *bqkv, *ln1, *ln2;
    // row-major bf16 copies [N][K] for the prefill GEMMs (many rows: the 8-row GEMV form would re-stream the weights per group)
    bf16_t *rqkv = nullptr, *ro = nullptr, *rgu = nullptr, *rd = nullptr;
    // exact-weights mode (weight_planes == 2): the lo planes bf16(w - bf16(w)) of all of the above, same layouts
    bf16_t *wqkv_lo = nullptr, *wo_lo = nullptr, *wgu_lo = nullptr, *wd_lo = nullptr;
    bf16_t *rqkv_lo = nullptr, *ro_lo = nullptr, *rgu_lo = nullptr, *rd_lo = nullptr;
};

struct fy_llm {
    fy_llm_config cfg;
    int max_batch = 0, max_ctx = 0, max_rows = 0;
    DevPool pool;
    std::vector<LlmLayerW> L;
    float *norm_w = nullptr, *speech_emb = nullptr, *inv_freq = nullptr;
    bf16_t *w_head = nullptr, *embed_tokens = nullptr;
    // How the matrices are stored.  1: one bf16 plane (exact for bf16-representable weights - rounds 1-4's only form).  2: two planes
    // w = hi + lo (16+ mantissa bits; a general fp32 checkpoint such as llm.pt, cli/model.py:65-73, keeps its values to 2^-17
    // relative - below the fp32 reference's own summation-order noise) and embed_tokens in fp32.  cfg.weight_planes = 0 picks 2
    // exactly when some matrix element is not bf16-representable.
    int planes = 1;
    bf16_t* w_head_lo = nullptr;
    float* embed_tokens_f32 = nullptr;
    float *Kc = nullptr, *Vc = nullptr;            // [layers][max_batch][kv_heads][max_ctx][64]
    float *h, *xn, *qkv, *ao, *act, *hb, *logits, *partial, *logp_keep;
    bf16_t* act_split = nullptr;                     // SwiGLU output as three bf16 planes per 8 rows: [rows/8][24][inter]
    int *row_seq, *row_pos, *row_src, *last_row;
    int *st;                                          // [8][max_batch]: pos, raw_n, n_out, done (2 = sampler gave up), run, min_len, max_len, uniforms used
    // sampler (fy_llm_set_sampler): 0 greedy, 1 repetition-aware sampling from caller-supplied uniforms
    int sampler = 0, top_k = 25, win_size = 10;
    float top_p = 0.8f, rep_thr = 1.0f;
    const float* uniforms = nullptr;                  // device (max_batch, n_uniforms), borrowed
    long n_uniforms = 0;
    int* recent = nullptr;                            // [win_size][max_batch] last raw tokens (ring)
    int *seq_ids;                                     // 0..max_batch-1
    int *counters = nullptr;                          // split-K arrival counters of the down projection
    int B = 0;
    int step_next = 0, steps_cap = 0;      // fy_llm_begin / fy_llm_step: next decode step of the generation in progress, and its bound
    bool all_done = false;                 // every sequence of that generation has ended: further steps are no-ops
    bf16_t* pl3 = nullptr;                                    // prefill GEMM path on the ring kernel: the A operand as three bf16 planes [3][rows][<= inter]
    bool prefill_ring = true;                                 // FY_LLM_PREFILL_RING=0: the register-staged exact-split kernel (A/B; ids identical)
    float *gu = nullptr, *actf = nullptr, *ones = nullptr;   // prefill GEMM path: gate/up products [rows][2 inter], SwiGLU [rows][inter], a vector of ones (the gate of h += W x)
    DecodePlan* dec = nullptr;             // persistent one-launch decode step (llm_decode.hip) when the architecture fits
    int prefill_gemm_rows = 320;           // FY_LLM_PREFILL_GEMM_ROWS: from this many prefill rows on, the tiled GEMMs
    bool prefill_gemm = true;              // FY_LLM_PREFILL_GEMM=0 keeps the 8-row products for the prefill too (A/B measurements)
    int decode_mode = 1;                   // fy_llm_set_decode_mode: 1 = use it, 0 = one launch per operation
    Dec32Plan* dec32 = nullptr;            // persistent decode step for 9 .. 32 sequences on a few CUs (llm_decode32.hip), same switch
    // per-operation products that serve 32 rows per weight pass (gemv32.h; FY_LLM_GEMV32=0 keeps the 8-row products of gemm.h):
    // the operands travel as A images written by the producers' epilogues
    bool gv32 = true;
    bf16_t *img_h = nullptr, *img_ao = nullptr, *img_act = nullptr;
    float *ssq = nullptr, *part32 = nullptr;
    int *cnt32 = nullptr;
    int n_speech() const { return cfg.speech_tokens + 200; }
    int qkv_dim() const { return (cfg.q_heads + 2 * cfg.kv_heads) * cfg.head_dim; }
    size_t cache_layer() const { return (size_t)max_batch * cfg.kv_heads * max_ctx * cfg.head_dim; }
};

extern "C" void fy_llm_default_config(fy_llm_config* c) {
    memset(c, 0, sizeof(*c));
    c->hidden = 896; c->layers = 24; c->q_heads = 14; c->kv_heads = 2; c->head_dim = 64; c->inter = 4864;
    c->vocab = 151936; c->speech_tokens = 6561; c->rms_eps = 1e-6f; c->rope_theta = 1e6f;
    c->weight_planes = 0;
}

__constant__ int c_silent[11] = {1, 2, 28, 29, 55, 248, 494, 2241, 2242, 2322, 2323};   // cli/model.py:414

// ---- kernels -------------------------------------------------------------------------------------------
// lm_input rows (llm.py:728-740): src = id | kind<<30, kind 0 = embed_tokens (bf16), 1 = speech_embedding (fp32)
// (etok32 != null: the text embedding kept in fp32 - the exact-weights mode)
__global__ void embed_rows_k(const int* __restrict__ src, const bf16_t* __restrict__ etok, const float* __restrict__ etok32, const float* __restrict__ semb,
                             float* __restrict__ h, int H) {
    int r = blockIdx.x, s = src[r], id = s & 0x3FFFFFFF, kind = s >> 30;
    for (int c = threadIdx.x; c < H; c += blockDim.x)
        h[(long)r * H + c] = kind ? semb[(long)id * H + c] : (etok32 ? etok32[(long)id * H + c] : bf16_to_f32(etok[(long)id * H + c]));
}

// Qwen2RMSNorm: w * (x * rsqrt(mean(x^2) + eps)).  One wave per row.
// planes != null: the result leaves as the three bf16 planes of its exact split ([3][R][H]: the operand of gemm_exact3) instead of fp32
__global__ __launch_bounds__(256) void rmsnorm_k(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int R, int H, float eps,
                                                 bf16_t* __restrict__ planes = nullptr) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* p = x + (long)row * H;
    float s = 0.f;
    for (int c = lane; c < H; c += 64) s += p[c] * p[c];
    float r = rsqrtf(wave_sum(s) / H + eps);
    if (planes) {
        const long RH = (long)R * H;
        for (int c = lane; c < H; c += 64) {
            unsigned h, m, l;
            gv32_split3(w[c] * (p[c] * r), h, m, l);
            const long o = (long)row * H + c;
            planes[o] = (bf16_t)h; planes[RH + o] = (bf16_t)m; planes[2 * RH + o] = (bf16_t)l;
        }
        return;
    }
    for (int c = lane; c < H; c += 64) y[(long)row * H + c] = w[c] * (p[c] * r);
}

// split-half RoPE on q (in place) and k (into the cache), v copied into the cache. One block per row.
__global__ void rope_kv_k(float* __restrict__ qkv, float* __restrict__ Kc, float* __restrict__ Vc, const int* __restrict__ row_seq,
                          const int* __restrict__ row_pos, const float* __restrict__ inv_freq, int Hq, int Hk, int max_ctx) {
    const int r = blockIdx.x, seq = row_seq[r], pos = row_pos[r];
    const int ld = (Hq + 2 * Hk) * 64;
    float* row = qkv + (long)r * ld;
    for (int i = threadIdx.x; i < (Hq + Hk) * 32; i += blockDim.x) {
        int hd = i >> 5, d = i & 31;
        float ang = (float)pos * inv_freq[d];
        float cs = cosf(ang), sn = sinf(ang);
        float a = row[hd * 64 + d], b = row[hd * 64 + d + 32];
        float o0 = a * cs - b * sn, o1 = b * cs + a * sn;
        if (hd < Hq) {
            row[hd * 64 + d] = o0;
            row[hd * 64 + d + 32] = o1;
        } else {
            float* kc = Kc + (((long)seq * Hk + (hd - Hq)) * max_ctx + pos) * 64;
            kc[d] = o0;
            kc[d + 32] = o1;
        }
    }
    for (int i = threadIdx.x; i < Hk * 64; i += blockDim.x) {
        int hk = i >> 6, d = i & 63;
        Vc[(((long)seq * Hk + hk) * max_ctx + pos) * 64 + d] = row[(Hq + Hk) * 64 + i];
    }
}

__global__ void gather_rows_k(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst, int H) {
    int b = blockIdx.x;
    for (int c = threadIdx.x; c < H; c += blockDim.x) dst[(long)b * H + c] = src[(long)idx[b] * H + c];
}

// What the 32-row decode path (gemv32.h) needs of the sampler besides the next input embedding h[b] = speech_embedding[id]:
// that row's A image under layer 0's input norm weight and its per-tile sums of squares (null img: nothing).
struct NextImg {
    bf16_t* img = nullptr;
    float* ssq = nullptr;
    const float* ln = nullptr;
    int Rpad = 0;
};
__device__ __forceinline__ void next_image(const NextImg& ni, const float* __restrict__ e, int b, int H, int tid) {
    if (!ni.img) return;
    for (int c0 = tid * 4; c0 < H; c0 += 1024) {             // 8 consecutive threads = one 32-column tile
        const float4 v = *reinterpret_cast<const float4*>(e + c0), w = *reinterpret_cast<const float4*>(ni.ln + c0);
        gv32_put4(ni.img, H / 16, b, c0, v.x * w.x, v.y * w.y, v.z * w.z, v.w * w.w);
        float q = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
        if ((tid & 7) == 0) ni.ssq[(long)(c0 >> 5) * ni.Rpad + b] = q;
    }
}

// log_softmax + greedy rule + stop / silent-token bookkeeping + next input embedding.
// st rows: 0 pos, 1 raw_n, 2 n_out, 3 done, 4 silent run, 5 min_len, 6 max_len.
__global__ __launch_bounds__(256) void sample_k(const float* __restrict__ logits, int n_all, int n_real, int* __restrict__ st, int mb,
                                                int* __restrict__ out_ids, int out_ld, const float* __restrict__ semb,
                                                float* __restrict__ h, int H, float* __restrict__ logp_keep, int keep_step, NextImg ni) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (st[3 * mb + b]) return;
    const float* x = logits + (long)b * n_all;
    // the row is read once into registers (all loads in flight together), the three passes run on registers
    constexpr int PER = 32;                                  // 256 x 32 = 8192 >= speech vocabulary + 200
    float xv[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int i = tid + u * 256;
        xv[u] = i < n_all ? x[i] : -3.0e38f;
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int u = 0; u < PER; ++u) mx = fmaxf(mx, xv[u]);
    sv[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) sv[tid] = fmaxf(sv[tid], sv[tid + o]); __syncthreads(); }
    mx = sv[0];
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < PER; ++u) if (tid + u * 256 < n_all) s += expf(xv[u] - mx);
    sv[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) sv[tid] += sv[tid + o]; __syncthreads(); }
    const float lse = logf(sv[0]);
    __syncthreads();
    const int raw_n = st[1 * mb + b];
    const int lim = raw_n < st[5 * mb + b] ? n_real : n_all;       // eos forbidden below min_len
    float best = -3.0e38f;
    int bi = 0x7FFFFFFF;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int i = tid + u * 256;
        if (i < n_all) {
            float lp = (xv[u] - mx) - lse;
            if (logp_keep && keep_step >= 0) logp_keep[((long)keep_step * mb + b) * n_all + i] = lp;
            if (i < lim && (lp > best || (lp == best && i < bi))) { best = lp; bi = i; }
        }
    }
    sv[tid] = best; si[tid] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            float ov = sv[tid + o]; int oi = si[tid + o];
            if (ov > sv[tid] || (ov == sv[tid] && oi < si[tid])) { sv[tid] = ov; si[tid] = oi; }
        }
        __syncthreads();
    }
    const int id = si[0];
    if (id >= n_real) {                      // stop_token_ids, llm.py:520
        if (tid == 0) st[3 * mb + b] = 1;
        return;
    }
    for (int c = tid; c < H; c += 256) h[(long)b * H + c] = semb[(long)id * H + c];      // llm.py:525
    next_image(ni, semb + (long)id * H, b, H, tid);
    if (tid == 0) {
        int run = st[4 * mb + b];
        bool silent = false;
        for (int k = 0; k < 11; ++k) silent |= (c_silent[k] == id);
        bool keep = true;
        if (silent) { run += 1; if (run > 5) keep = false; } else run = 0;      // cli/model.py:121-127
        st[4 * mb + b] = run;
        if (keep) { int n = st[2 * mb + b]; if (n < out_ld) out_ids[(long)b * out_ld + n] = id; st[2 * mb + b] = n + 1; }
        st[1 * mb + b] = raw_n + 1;
        st[0 * mb + b] += 1;                                                   // the new token's position
        if (raw_n + 1 >= st[6 * mb + b]) st[3 * mb + b] = 1;                   // for i in range(max_len), llm.py:514
    }
}

// ---- repetition-aware sampling (ras_sampling / nucleus_sampling / random_sampling, utils/common.py:137-166,
// and the ignore_eos retry loop of TransformerLM.sampling_ids, llm/llm.py:149-164) ---------------------------------------
// torch.multinomial(1) draws from torch's global generator, which no other implementation can follow; the draw is
// therefore DEFINED here (and identically in oracle/llm.py:inv_cdf and in the fixture mint, which patches
// Tensor.multinomial) as the inverse CDF at a caller-supplied uniform u in [0, 1): weights are summed in double, in index
// order, in chunks of ceil(n/256) (chunk sums, then a running sum over chunks, then a running sum inside the chosen
// chunk started from the previous chunks' total); the sample is the first index whose running sum exceeds u x total.
__device__ int inv_cdf_small(const float* p, int n, float u) {
    double tot = 0.0;
    for (int j = 0; j < n; ++j) tot += (double)p[j];
    const double target = (double)u * tot;
    double c = 0.0;
    for (int j = 0; j < n; ++j) {
        c += (double)p[j];
        if (c > target) return j;
    }
    return n - 1;
}

__global__ __launch_bounds__(256) void sample_ras_k(const float* __restrict__ logits, int n_all, int n_real, int* __restrict__ st, int mb,
                                                    int* __restrict__ out_ids, int out_ld, const float* __restrict__ semb,
                                                    float* __restrict__ h, int H, float* __restrict__ logp_keep, int keep_step,
                                                    const float* __restrict__ uni, long n_uni, int* __restrict__ recent,
                                                    int top_k, float top_p, int win, float rep_thr, NextImg ni) {
    extern __shared__ float pl[];                            // softmax(logp) in index order
    __shared__ float sv[256];
    __shared__ int si[256];
    __shared__ double cs[256];
    __shared__ float cand_p[32];
    __shared__ int cand_i[32];
    __shared__ int sh_n, sh_id, sh_flag, sh_cnt;
    __shared__ float sh_u;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (st[3 * mb + b]) return;
    const float* x = logits + (long)b * n_all;
    constexpr int PER = 32;
    float xv[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int i = tid + u * 256;
        xv[u] = i < n_all ? x[i] : -3.0e38f;
    }
    auto block_max = [&](float v) {
        sv[tid] = v;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) sv[tid] = fmaxf(sv[tid], sv[tid + o]); __syncthreads(); }
        float r = sv[0];
        __syncthreads();
        return r;
    };
    auto block_sum = [&](float v) {
        sv[tid] = v;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (tid < o) sv[tid] += sv[tid + o]; __syncthreads(); }
        float r = sv[0];
        __syncthreads();
        return r;
    };
    float mx = -3.0e38f;
#pragma unroll
    for (int u = 0; u < PER; ++u) mx = fmaxf(mx, xv[u]);
    mx = block_max(mx);
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < PER; ++u) if (tid + u * 256 < n_all) s += expf(xv[u] - mx);
    const float lse = logf(block_sum(s));
    // logp, then softmax(logp) as the reference computes it from the stored log-probabilities
    float lpm = -3.0e38f;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int i = tid + u * 256;
        if (i < n_all) {
            xv[u] = (xv[u] - mx) - lse;
            if (logp_keep && keep_step >= 0) logp_keep[((long)keep_step * mb + b) * n_all + i] = xv[u];
            lpm = fmaxf(lpm, xv[u]);
        }
    }
    lpm = block_max(lpm);
    float s2 = 0.f;
#pragma unroll
    for (int u = 0; u < PER; ++u) if (tid + u * 256 < n_all) { xv[u] = expf(xv[u] - lpm); s2 += xv[u]; }
    s2 = block_sum(s2);
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int i = tid + u * 256;
        if (i < n_all) { xv[u] = xv[u] / s2; pl[i] = xv[u]; } else xv[u] = -1.f;
    }
    // nucleus: the stable descending sort's head, taken while cum < top_p and fewer than top_k (common.py:146-156)
    if (tid == 0) { sh_n = 0; sh_flag = 0; }
    __syncthreads();
    float cum = 0.f;
    for (int r = 0; r < top_k && r < 32; ++r) {
        float best = -2.f;
        int bi = 0x7FFFFFFF;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = tid + u * 256;
            if (xv[u] > best || (xv[u] == best && i < bi)) { best = xv[u]; bi = i; }
        }
        sv[tid] = best; si[tid] = bi;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) {
                float ov = sv[tid + o]; int oi = si[tid + o];
                if (ov > sv[tid] || (ov == sv[tid] && oi < si[tid])) { sv[tid] = ov; si[tid] = oi; }
            }
            __syncthreads();
        }
        const float bp = sv[0];
        const int bidx = si[0];
        __syncthreads();
        if (!(cum < top_p)) break;                           // every thread tracks cum: a uniform decision
        if (tid == 0) { cand_p[r] = bp; cand_i[r] = bidx; sh_n = r + 1; }
        cum += bp;
        if ((bidx & 255) == tid) xv[bidx >> 8] = -1.f;       // taken
    }
    __syncthreads();
    const int raw_n = st[1 * mb + b];
    const bool ignore_eos = raw_n < st[5 * mb + b];
    const int n_rec = min(raw_n, win);
    int cnt = st[7 * mb + b];
    bool have_cs = false;
    const int chunk = (n_all + 255) / 256;
    int id = -1;
    for (int trial = 0; trial <= 100; ++trial) {             // sampling_ids: max_trials = 100, then RuntimeError
        if (tid == 0) {
            const float u1 = uni[(long)b * n_uni + min((long)cnt, n_uni - 1)];
            int pick = cand_i[inv_cdf_small(cand_p, sh_n, u1)];
            int rep = 0;
            for (int k = 0; k < n_rec; ++k) rep += recent[k * mb + b] == pick;
            sh_id = pick;
            sh_flag = (float)rep >= rep_thr;
            sh_cnt = cnt + 1;
        }
        __syncthreads();
        cnt = sh_cnt;
        if (sh_flag) {                                       // random_sampling over the whole softmax (common.py:160-162)
            if (!have_cs) {
                double a = 0.0;
                for (int k = 0; k < chunk; ++k) { const int i = tid * chunk + k; if (i < n_all) a += (double)pl[i]; }
                cs[tid] = a;
                have_cs = true;
                __syncthreads();
            }
            if (tid == 0) {
                const float u2 = uni[(long)b * n_uni + min((long)cnt, n_uni - 1)];
                double tot = 0.0;
                for (int t = 0; t < 256; ++t) tot += cs[t];
                const double target = (double)u2 * tot;
                double c = 0.0, base = 0.0;
                int tc = 255;
                for (int t = 0; t < 256; ++t) { base = c; c += cs[t]; if (c > target) { tc = t; break; } }
                int pick = min(tc * chunk + chunk - 1, n_all - 1);
                double w = base;
                for (int k = 0; k < chunk; ++k) {
                    const int i = tc * chunk + k;
                    if (i >= n_all) break;
                    w += (double)pl[i];
                    if (w > target) { pick = i; break; }
                }
                sh_id = pick;
                sh_cnt = cnt + 1;
            }
            __syncthreads();
            cnt = sh_cnt;
        }
        const int cand = sh_id;
        __syncthreads();
        if (!ignore_eos || cand < n_real) { id = cand; break; }
    }
    if (tid == 0) st[7 * mb + b] = cnt;
    if (id < 0) {                            // 'sampling reaches max_trials ... and still get eos': reported by fy_llm_generate
        if (tid == 0) st[3 * mb + b] = 2;
        return;
    }
    if (id >= n_real) {                      // stop_token_ids, llm.py:520
        if (tid == 0) st[3 * mb + b] = 1;
        return;
    }
    for (int c = tid; c < H; c += 256) h[(long)b * H + c] = semb[(long)id * H + c];      // llm.py:525
    next_image(ni, semb + (long)id * H, b, H, tid);
    if (tid == 0) {
        recent[(raw_n % win) * mb + b] = id;                                   // decoded_tokens[-win_size:], order does not matter
        int run = st[4 * mb + b];
        bool silent = false;
        for (int k = 0; k < 11; ++k) silent |= (c_silent[k] == id);
        bool keep = true;
        if (silent) { run += 1; if (run > 5) keep = false; } else run = 0;      // cli/model.py:121-127
        st[4 * mb + b] = run;
        if (keep) { int n = st[2 * mb + b]; if (n < out_ld) out_ids[(long)b * out_ld + n] = id; st[2 * mb + b] = n + 1; }
        st[1 * mb + b] = raw_n + 1;
        st[0 * mb + b] += 1;
        if (raw_n + 1 >= st[6 * mb + b]) st[3 * mb + b] = 1;
    }
}

// act[r][i] = silu(gu[r][2i]) * gu[r][2i+1]: the gate / up rows are interleaved (Qwen2MLP: down(silu(gate(x)) * up(x)))
// planes != null: as three bf16 planes [3][n] instead of fp32 (two columns per thread)
__global__ void swiglu_rows_k(const float* __restrict__ gu, float* __restrict__ act, long n, int inter, bf16_t* __restrict__ planes = nullptr) {
    if (planes) {
        for (long i = (blockIdx.x * 256L + threadIdx.x) * 2; i < n; i += (long)gridDim.x * 512) {      // inter is even: a pair stays in its row
            const long r = i / inter;
            const int c = (int)(i % inter);
            const float4 p = *reinterpret_cast<const float4*>(gu + r * 2 * inter + 2 * c);
            unsigned h0, m0, l0, h1, m1, l1;
            gv32_split3(act_silu(p.x) * p.y, h0, m0, l0);
            gv32_split3(act_silu(p.z) * p.w, h1, m1, l1);
            *reinterpret_cast<unsigned*>(planes + i) = h0 | (h1 << 16);
            *reinterpret_cast<unsigned*>(planes + n + i) = m0 | (m1 << 16);
            *reinterpret_cast<unsigned*>(planes + 2 * n + i) = l0 | (l1 << 16);
        }
        return;
    }
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / inter;
        const int c = (int)(i % inter);
        const float2 p = *reinterpret_cast<const float2*>(gu + r * 2 * inter + 2 * c);
        act[i] = act_silu(p.x) * p.y;
    }
}
__global__ void fill_k(float* p, float v, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- create -------------------------------------------------------------------------------------------------
static int to_bf16(fy_llm* l, const float* src, size_t n, bf16_t** dst, hipStream_t st) {
    FY_TRY(l->pool.alloc(dst, n));
    return cast_f32_bf16(src, *dst, n, st);
}
// fp32 [N][K] -> the GEMV's fragment-ordered bf16
static int to_packed(fy_llm* l, const float* src, int N, int K, bf16_t** dst, hipStream_t st) {
    FY_TRY(l->pool.alloc(dst, gemv_packed_elems(N, K)));
    return gemv_pack(src, *dst, N, K, st);
}
static int copy_f32(fy_llm* l, const float* src, size_t n, float** dst, hipStream_t st) {
    FY_TRY(l->pool.alloc(dst, n));
    HIP_TRY(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    return FY_OK;
}

// dst = src - float(bf16(src)): what one bf16 plane loses (its own bf16 rounding is the lo plane)
__global__ void bf16_residual_k(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * 256UL + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i] - bf16_to_f32(f32_to_bf16(src[i]));
}
// *count += number of elements that are not bf16-representable
__global__ void bf16_inexact_k(const float* __restrict__ src, size_t n, unsigned long long* __restrict__ count) {
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * 256UL + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) c += bf16_to_f32(f32_to_bf16(src[i])) != src[i];
    c += __shfl_xor(c, 32, 64); c += __shfl_xor(c, 16, 64); c += __shfl_xor(c, 8, 64); c += __shfl_xor(c, 4, 64); c += __shfl_xor(c, 2, 64); c += __shfl_xor(c, 1, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}
static unsigned grid_for(size_t n) { return (unsigned)std::min<size_t>(8192, (n + 255) / 256); }
// the lo planes of a matrix src [N][K] (tmp: N*K floats of scratch): fragment order and / or row-major
static int lo_planes(fy_llm* l, const float* src, float* tmp, int N, int K, bf16_t** packed, bf16_t** rowmajor, hipStream_t st) {
    hipLaunchKernelGGL(bf16_residual_k, dim3(grid_for((size_t)N * K)), dim3(256), 0, st, src, tmp, (size_t)N * K);
    if (packed) FY_TRY(to_packed(l, tmp, N, K, packed, st));
    if (rowmajor) FY_TRY(to_bf16(l, tmp, (size_t)N * K, rowmajor, st));
    return FY_OK;
}

// rows (gate_i, up_i) interleaved so one wave owns both halves of a SwiGLU pair
__global__ void interleave_gu_k(const float* __restrict__ g, const float* __restrict__ u, float* __restrict__ out, int inter, int H) {
    long n = (long)2 * inter * H;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        long row = i / H;
        int c = (int)(i % H);
        const float* src = (row & 1) ? u : g;
        out[i] = src[(row >> 1) * H + c];
    }
}

extern "C" int fy_llm_create(fy_llm** out, const fy_llm_config* cfg, const fy_tensor* weights, int32_t n_weights,
                             int32_t max_batch, int32_t max_ctx, void* stream) {
    FY_CHECK(out && max_batch >= 1 && max_ctx >= 8, FY_ERR_ARG, "fy_llm_create: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    fy_llm* l = new fy_llm();
    if (cfg) l->cfg = *cfg; else fy_llm_default_config(&l->cfg);
    const fy_llm_config& c = l->cfg;
    auto fail = [&](int code) { if (l->dec) decode_destroy(l->dec); if (l->dec32) decode32_destroy(l->dec32); delete l; return code; };
    if (c.head_dim != 64 || c.q_heads % c.kv_heads != 0 || c.hidden % 8 != 0 || c.inter % 8 != 0 || c.q_heads * c.head_dim != c.hidden) {
        fy_set_error("fy_llm_create: unsupported architecture (hidden %d, heads %d/%d x %d)", c.hidden, c.q_heads, c.kv_heads, c.head_dim);
        return fail(FY_ERR_ARG);
    }
    l->max_batch = max_batch; l->max_ctx = max_ctx; l->max_rows = max_batch * max_ctx;
    if (const char* e = getenv("FY_LLM_PREFILL_GEMM")) l->prefill_gemm = atoi(e) != 0;
    if (const char* e = getenv("FY_LLM_PREFILL_GEMM_ROWS")) l->prefill_gemm_rows = std::max(1, atoi(e));
    Weights W;
    int rc = W.init(weights, n_weights);
    if (rc) return fail(rc);
#define TRYC(e) do { int _r = (e); if (_r) return fail(_r); } while (0)
#define GETW(var, name, ...) const float* var = W.get(name, {__VA_ARGS__}); if (!var) return fail(FY_ERR_WEIGHT)
    const int H = c.hidden, Q = c.q_heads * 64, KV = c.kv_heads * 64, I = c.inter, NS = l->n_speech();
    const std::string P = "llm.model.model.";
    l->L.resize(c.layers);
    TRYC(l->pool.alloc(&l->Kc, (size_t)c.layers * l->cache_layer()));
    TRYC(l->pool.alloc(&l->Vc, (size_t)c.layers * l->cache_layer()));
    if (c.weight_planes != 0 && c.weight_planes != 1 && c.weight_planes != 2) {
        fy_set_error("fy_llm_create: weight_planes must be 0 (choose), 1 or 2, not %d", c.weight_planes);
        return fail(FY_ERR_ARG);
    }
    l->planes = c.weight_planes == 2 ? 2 : 1;
    if (c.weight_planes == 0) {
        // one pass over the matrices: does bf16 storage lose anything?  (synthetic parity weights: no; a real checkpoint: yes)
        unsigned long long* cnt = nullptr;
        TRYC(l->pool.alloc(&cnt, (size_t)2));
        bool ok = hipMemsetAsync(cnt, 0, 16, st) == hipSuccess;
        auto scan = [&](const std::string& name, long rows, long cols) {
            const float* w = W.get(name, {rows, cols});
            if (!w) return false;
            hipLaunchKernelGGL(bf16_inexact_k, dim3(grid_for((size_t)rows * cols)), dim3(256), 0, st, w, (size_t)rows * cols, cnt);
            return true;
        };
        for (int i = 0; i < c.layers && ok; ++i) {
            const std::string p = P + "layers." + std::to_string(i) + ".";
            ok = scan(p + "self_attn.q_proj.weight", Q, H) && scan(p + "self_attn.k_proj.weight", KV, H) && scan(p + "self_attn.v_proj.weight", KV, H) &&
                 scan(p + "self_attn.o_proj.weight", H, Q) && scan(p + "mlp.gate_proj.weight", I, H) && scan(p + "mlp.up_proj.weight", I, H) &&
                 scan(p + "mlp.down_proj.weight", H, I);
        }
        ok = ok && scan("llm_decoder.weight", NS, H) && scan(P + "embed_tokens.weight", c.vocab, H);
        unsigned long long n_inexact = 0;
        if (!ok) return fail(FY_ERR_WEIGHT);
        if (hipMemcpyAsync(&n_inexact, cnt, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            fy_set_error("fy_llm_create: weight scan failed");
            return fail(FY_ERR_HIP);
        }
        l->planes = n_inexact ? 2 : 1;
    }
    const bool xw = l->planes == 2;
    if (const char* e = getenv("FY_LLM_GEMV32")) l->gv32 = atoi(e) != 0;
    if (H % 32 != 0 || I % 16 != 0 || H / 32 > 32) l->gv32 = false;
    if (xw && !l->gv32) {
        fy_set_error("fy_llm_create: two weight planes need the 32-row products (hidden %% 32 == 0, inter %% 16 == 0, hidden <= 1024; FY_LLM_GEMV32 unset)");
        return fail(FY_ERR_ARG);
    }
    if (!xw) {       // the 8-row persistent step keeps a whole layer of weights in registers: one plane only
        DecodeShape ds;
        ds.H = H; ds.I = I; ds.Hq = c.q_heads; ds.Hk = c.kv_heads; ds.layers = c.layers; ds.NS = NS; ds.max_ctx = max_ctx; ds.mb = max_batch; ds.eps = c.rms_eps;
        if (decode_supported(ds)) TRYC(decode_create(&l->dec, ds, st));
    }
    for (int i = 0; i < c.layers; ++i) {
        const std::string p = P + "layers." + std::to_string(i) + ".";
        LlmLayerW& k = l->L[i];
        GETW(qw, p + "self_attn.q_proj.weight", Q, H); GETW(qb, p + "self_attn.q_proj.bias", Q);
        GETW(kw, p + "self_attn.k_proj.weight", KV, H); GETW(kb, p + "self_attn.k_proj.bias", KV);
        GETW(vw, p + "self_attn.v_proj.weight", KV, H); GETW(vb, p + "self_attn.v_proj.bias", KV);
        GETW(ow, p + "self_attn.o_proj.weight", H, Q);
        GETW(gw, p + "mlp.gate_proj.weight", I, H); GETW(uw, p + "mlp.up_proj.weight", I, H);
        GETW(dw, p + "mlp.down_proj.weight", H, I);
        GETW(n1, p + "input_layernorm.weight", H); GETW(n2, p + "post_attention_layernorm.weight", H);
        // q, k, v rows concatenated; gate / up rows interleaved; all in the GEMV's fragment order
        float *tmp = nullptr, *tmpg = nullptr;
        TRYC(l->pool.alloc(&k.bqkv, (size_t)(Q + 2 * KV)));
        if (hipMalloc(&tmp, (size_t)(Q + 2 * KV) * H * sizeof(float)) != hipSuccess || hipMalloc(&tmpg, (size_t)2 * I * H * sizeof(float)) != hipSuccess) {
            if (tmp) (void)hipFree(tmp);
            fy_set_error("fy_llm_create: out of memory");
            return fail(FY_ERR_HIP);
        }
        bool ok = hipMemcpyAsync(tmp, qw, (size_t)Q * H * 4, hipMemcpyDeviceToDevice, st) == hipSuccess &&
                  hipMemcpyAsync(tmp + (size_t)Q * H, kw, (size_t)KV * H * 4, hipMemcpyDeviceToDevice, st) == hipSuccess &&
                  hipMemcpyAsync(tmp + (size_t)(Q + KV) * H, vw, (size_t)KV * H * 4, hipMemcpyDeviceToDevice, st) == hipSuccess &&
                  hipMemcpyAsync(k.bqkv, qb, Q * sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess &&
                  hipMemcpyAsync(k.bqkv + Q, kb, KV * sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess &&
                  hipMemcpyAsync(k.bqkv + Q + KV, vb, KV * sizeof(float), hipMemcpyDeviceToDevice, st) == hipSuccess;
        int rc2 = ok ? to_packed(l, tmp, Q + 2 * KV, H, &k.wqkv, st) : FY_ERR_HIP;
        if (rc2 == FY_OK) {
            hipLaunchKernelGGL(interleave_gu_k, dim3(4096), dim3(256), 0, st, gw, uw, tmpg, I, H);
            rc2 = to_packed(l, tmpg, 2 * I, H, &k.wgu, st);
        }
        if (rc2 == FY_OK) rc2 = to_bf16(l, tmp, (size_t)(Q + 2 * KV) * H, &k.rqkv, st);
        if (rc2 == FY_OK) rc2 = to_bf16(l, tmpg, (size_t)2 * I * H, &k.rgu, st);
        if (rc2 == FY_OK) rc2 = to_bf16(l, ow, (size_t)H * Q, &k.ro, st);
        if (rc2 == FY_OK) rc2 = to_bf16(l, dw, (size_t)H * I, &k.rd, st);
        if (rc2 == FY_OK && xw) {
            // the lo planes (scratch: 2 I H floats hold the residual of any of the four matrices)
            float* scratch = nullptr;
            if (hipMalloc(&scratch, (size_t)2 * I * H * sizeof(float)) != hipSuccess) { fy_set_error("fy_llm_create: out of memory"); rc2 = FY_ERR_HIP; }
            if (rc2 == FY_OK) rc2 = lo_planes(l, tmp, scratch, Q + 2 * KV, H, &k.wqkv_lo, &k.rqkv_lo, st);
            if (rc2 == FY_OK) rc2 = lo_planes(l, tmpg, scratch, 2 * I, H, &k.wgu_lo, &k.rgu_lo, st);
            if (rc2 == FY_OK) rc2 = lo_planes(l, ow, scratch, H, Q, &k.wo_lo, &k.ro_lo, st);
            if (rc2 == FY_OK) rc2 = lo_planes(l, dw, scratch, H, I, &k.wd_lo, &k.rd_lo, st);
            (void)hipStreamSynchronize(st);
            if (scratch) (void)hipFree(scratch);
        }
        if (rc2 == FY_OK && l->dec) {
            DecodeLayerSrc ds;
            ds.wqkv = tmp; ds.wo = ow; ds.wgu = tmpg; ds.wd = dw; ds.bqkv = k.bqkv; ds.ln1 = n1; ds.ln2 = n2;
            ds.Kc = l->Kc + (size_t)i * l->cache_layer(); ds.Vc = l->Vc + (size_t)i * l->cache_layer();
            rc2 = decode_pack_layer(l->dec, i, ds, st);
        }
        (void)hipStreamSynchronize(st);
        (void)hipFree(tmp);
        (void)hipFree(tmpg);
        if (rc2 != FY_OK) { if (!ok) fy_set_error("fy_llm_create: weight copy failed"); return fail(rc2); }
        TRYC(to_packed(l, ow, H, Q, &k.wo, st));
        TRYC(to_packed(l, dw, H, I, &k.wd, st));
        TRYC(copy_f32(l, n1, H, &k.ln1, st)); TRYC(copy_f32(l, n2, H, &k.ln2, st));
    }
    {
        GETW(nw, P + "norm.weight", H);
        GETW(hw, "llm_decoder.weight", NS, H);
        GETW(sw, "speech_embedding.weight", NS, H);
        GETW(ew, P + "embed_tokens.weight", c.vocab, H);
        TRYC(copy_f32(l, nw, H, &l->norm_w, st));
        TRYC(to_packed(l, hw, NS, H, &l->w_head, st));
        if (l->dec) TRYC(decode_pack_head(l->dec, hw, nw, st));
        TRYC(copy_f32(l, sw, (size_t)NS * H, &l->speech_emb, st));
        if (xw) {
            float* scratch = nullptr;
            if (hipMalloc(&scratch, (size_t)NS * H * sizeof(float)) != hipSuccess) { fy_set_error("fy_llm_create: out of memory"); return fail(FY_ERR_HIP); }
            int rc3 = lo_planes(l, hw, scratch, NS, H, &l->w_head_lo, nullptr, st);
            (void)hipStreamSynchronize(st);
            (void)hipFree(scratch);
            TRYC(rc3);
            TRYC(copy_f32(l, ew, (size_t)c.vocab * H, &l->embed_tokens_f32, st));
        } else TRYC(to_bf16(l, ew, (size_t)c.vocab * H, &l->embed_tokens, st));
    }
    {
        std::vector<float> inv(32);
        for (int i = 0; i < 32; ++i) inv[i] = 1.0f / powf(c.rope_theta, (float)(2 * i) / 64.0f);
        TRYC(l->pool.alloc(&l->inv_freq, 32));
        if (hipMemcpyAsync(l->inv_freq, inv.data(), 32 * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) {
            fy_set_error("fy_llm_create: upload failed");
            return fail(FY_ERR_HIP);
        }
    }
    const size_t R = l->max_rows, B = max_batch;
    TRYC(l->pool.alloc(&l->h, R * H)); TRYC(l->pool.alloc(&l->xn, R * H)); TRYC(l->pool.alloc(&l->qkv, R * l->qkv_dim()));
    TRYC(l->pool.alloc(&l->ao, R * H)); TRYC(l->pool.alloc(&l->act, (size_t)16)); TRYC(l->pool.alloc(&l->hb, B * H));
    // rows the 8- / 32-row products ever see: the tiled GEMMs take a prefill from prefill_gemm_rows on (when they apply at all)
    const size_t Rg = (l->prefill_gemm && H % 64 == 0 && I % 64 == 0) ? std::min<size_t>(R, std::max<size_t>(B, (size_t)l->prefill_gemm_rows)) : R;
    if (l->gv32) {
        const size_t n_h = gv32_image_elems((int)Rg, H), n_a = gv32_image_elems((int)Rg, I), n_s = gv32_ssq_floats((int)Rg, H);
        TRYC(l->pool.alloc(&l->img_h, n_h)); TRYC(l->pool.alloc(&l->img_ao, n_h)); TRYC(l->pool.alloc(&l->img_act, n_a));
        TRYC(l->pool.alloc(&l->ssq, n_s));
        TRYC(l->pool.alloc(&l->part32, gv32_partial_floats((int)Rg, H, I) + 16)); TRYC(l->pool.alloc(&l->cnt32, gv32_counter_ints((int)Rg, H, I) + 16));
        // rows beyond the live ones are never written and must read as zeros (a row of the A operand only reaches its own outputs)
        if (hipMemsetAsync(l->img_h, 0, n_h * 2, st) != hipSuccess || hipMemsetAsync(l->img_ao, 0, n_h * 2, st) != hipSuccess ||
            hipMemsetAsync(l->img_act, 0, n_a * 2, st) != hipSuccess || hipMemsetAsync(l->ssq, 0, n_s * 4, st) != hipSuccess ||
            hipMemsetAsync(l->cnt32, 0, (gv32_counter_ints((int)Rg, H, I) + 16) * sizeof(int), st) != hipSuccess) {
            fy_set_error("fy_llm_create: memset failed");
            return fail(FY_ERR_HIP);
        }
    } else TRYC(l->pool.alloc(&l->act_split, ((R + 7) / 8) * 24 * (size_t)I));
    if (l->gv32) {
        // the few-CU persistent step for 9 .. 32 sequences shares the per-operation path's weights, images and scratch
        Dec32Shape s32;
        s32.H = H; s32.I = I; s32.Hq = c.q_heads; s32.Hk = c.kv_heads; s32.layers = c.layers; s32.NS = NS; s32.max_ctx = max_ctx; s32.mb = max_batch; s32.eps = c.rms_eps;
        if (decode32_supported(s32)) {
            std::vector<Dec32Layer> ly(c.layers);
            for (int i = 0; i < c.layers; ++i) {
                const LlmLayerW& k = l->L[i];
                ly[i].wqkv = k.wqkv; ly[i].wo = k.wo; ly[i].wgu = k.wgu; ly[i].wd = k.wd; ly[i].bqkv = k.bqkv; ly[i].ln1 = k.ln1; ly[i].ln2 = k.ln2;
                ly[i].Kc = l->Kc + (size_t)i * l->cache_layer(); ly[i].Vc = l->Vc + (size_t)i * l->cache_layer();
                ly[i].wqkv_lo = k.wqkv_lo; ly[i].wo_lo = k.wo_lo; ly[i].wgu_lo = k.wgu_lo; ly[i].wd_lo = k.wd_lo;
            }
            TRYC(decode32_create(&l->dec32, s32, ly.data(), l->w_head, l->norm_w, st, l->w_head_lo));
        }
    }
    TRYC(l->pool.alloc(&l->logits, B * NS)); TRYC(l->pool.alloc(&l->partial, gemv_partial_floats((int)R, H, I) + 16));
    TRYC(l->pool.alloc(&l->counters, gemv_counter_ints((int)R, H, I) + 16));
    if (hipMemsetAsync(l->counters, 0, (gemv_counter_ints((int)R, H, I) + 16) * sizeof(int), st) != hipSuccess) { fy_set_error("fy_llm_create: memset failed"); return fail(FY_ERR_HIP); }
    TRYC(l->pool.alloc(&l->logp_keep, (size_t)FY_LLM_KEEP_LOGP * B * NS));
    TRYC(l->pool.alloc(&l->gu, R * 2 * (size_t)I)); TRYC(l->pool.alloc(&l->actf, R * (size_t)I)); TRYC(l->pool.alloc(&l->ones, (size_t)std::max(2 * I, l->qkv_dim())));
    TRYC(l->pool.alloc(&l->pl3, 3 * R * (size_t)std::max(H, I)));
    l->prefill_ring = !(getenv("FY_LLM_PREFILL_RING") && atoi(getenv("FY_LLM_PREFILL_RING")) == 0);
    hipLaunchKernelGGL(fill_k, dim3(cdiv(std::max(2 * I, l->qkv_dim()), 256)), dim3(256), 0, st, l->ones, 1.0f, std::max(2 * I, l->qkv_dim()));
    TRYC(l->pool.alloc(&l->row_seq, R)); TRYC(l->pool.alloc(&l->row_pos, R)); TRYC(l->pool.alloc(&l->row_src, R));
    TRYC(l->pool.alloc(&l->last_row, B)); TRYC(l->pool.alloc(&l->st, 8 * B)); TRYC(l->pool.alloc(&l->seq_ids, B));
    {
        std::vector<int> ids(B);
        for (size_t i = 0; i < B; ++i) ids[i] = (int)i;
        if (hipMemcpyAsync(l->seq_ids, ids.data(), B * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) {
            fy_set_error("fy_llm_create: upload failed");
            return fail(FY_ERR_HIP);
        }
    }
#undef GETW
#undef TRYC
    if (hipStreamSynchronize(st) != hipSuccess) { fy_set_error("fy_llm_create: stream sync failed"); return fail(FY_ERR_HIP); }
    *out = l;
    return FY_OK;
}

extern "C" void fy_llm_destroy(fy_llm* l) {
    if (l && l->recent) (void)hipFree(l->recent);
    if (l && l->dec) decode_destroy(l->dec);
    if (l && l->dec32) decode32_destroy(l->dec32);
    delete l;
}

// ---- forward over R rows (prefill rows or one decode row per sequence) ---------------------------------------------
static bool llm_prefill_by_gemm(const fy_llm* l, int R) {
    return R >= l->prefill_gemm_rows && l->cfg.hidden % 64 == 0 && l->cfg.inter % 64 == 0 && l->prefill_gemm;
}

static int llm_layers(fy_llm* l, int R, const int* row_seq, const int* row_pos, bool decode, hipStream_t st) {
    const fy_llm_config& c = l->cfg;
    const int H = c.hidden, I = c.inter, QKV = l->qkv_dim();
    // Prefill over many rows: tiled MFMA GEMMs with the activations split EXACTLY three ways (gemm_f32a_exact) - the same
    // fidelity as the 8-row products below, which at R rows would stream every weight matrix ceil(R / 8) times.
    // Measured (tests/prefill_probe.py, CosyVoice3-0.5B): 4 x 296 rows 14.9 ms against 42.8; 8 x ~25 rows 10.0 against 7.1 (a
    // handful of 128-row tiles leaves the chip empty) - the crossover is near 300 rows.
    const bool gemm_path = !decode && llm_prefill_by_gemm(l, R);
    for (int i = 0; i < c.layers; ++i) {
        const LlmLayerW& k = l->L[i];
        float* Kc = l->Kc + (size_t)i * l->cache_layer();
        float* Vc = l->Vc + (size_t)i * l->cache_layer();
        if (gemm_path) {
            // Round 4: the exact three-way split products on the LDS-DMA ring kernel (gemm_exact3) - the operand is split into its
            // planes once by a small kernel instead of by every column tile's workgroup, and the K loop is the DiT linears'.
            const bool ring = l->prefill_ring && gemm_exact3_supported(QKV, H) && gemm_exact3_supported(H, H) && gemm_exact3_supported(2 * I, H) && gemm_exact3_supported(H, I);
            // exact(A, K, W, N, e): A == nullptr means the producer already left the planes in pl3
            // W_lo (exact-weights mode): the product is W a + W_lo a - a second pass that ADDS into the first one's output (the gated
            // residual form with a gate of ones; every epilogue here is linear in the product)
            auto exact = [&](const float* A, int K, const bf16_t* W, const bf16_t* W_lo, int N, GemmEpi& e) -> int {
                GemmEpi e2;
                if (W_lo) {
                    e2.mode = EPI_GATE_RESID; e2.gate = l->ones; e2.ldc = e.ldc;
                    e2.resid = e.mode == EPI_GATE_RESID ? e.resid : (float*)e.out;
                }
                if (!ring) {
                    FY_TRY(gemm_f32a_exact(A, K, W, R, N, K, e, st));
                    return W_lo ? gemm_f32a_exact(A, K, W_lo, R, N, K, e2, st) : FY_OK;
                }
                bf16_t *p0 = l->pl3, *p1 = p0 + (size_t)R * K, *p2 = p1 + (size_t)R * K;
                if (A) FY_TRY(split3_planes(A, K, R, K, p0, p1, p2, st));
                e.a_lo = p1; e.a_lo2 = p2;
                FY_TRY(gemm_exact3(p0, K, W, R, N, K, e, st));
                if (!W_lo) return FY_OK;
                e2.a_lo = p1; e2.a_lo2 = p2;
                return gemm_exact3(p0, K, W_lo, R, N, K, e2, st);
            };
            hipLaunchKernelGGL(rmsnorm_k, dim3(cdiv(R, 4)), dim3(256), 0, st, l->h, k.ln1, l->xn, R, H, c.rms_eps, ring ? l->pl3 : nullptr);
            GemmEpi q;
            q.bias = k.bqkv; q.out = l->qkv; q.out_bf16 = 0; q.ldc = QKV;
            FY_TRY(exact(ring ? nullptr : l->xn, H, k.rqkv, k.rqkv_lo, QKV, q));
            hipLaunchKernelGGL(rope_kv_k, dim3(R), dim3(256), 0, st, l->qkv, Kc, Vc, row_seq, row_pos, l->inv_freq, c.q_heads, c.kv_heads, l->max_ctx);
            FY_TRY(llm_attention(l->qkv, QKV, Kc, Vc, row_seq, row_pos, l->ao, H, R, c.q_heads, c.kv_heads, l->max_ctx, st));
            GemmEpi o;
            o.mode = EPI_GATE_RESID; o.resid = l->h; o.gate = l->ones; o.ldc = H;
            FY_TRY(exact(l->ao, H, k.ro, k.ro_lo, H, o));
            hipLaunchKernelGGL(rmsnorm_k, dim3(cdiv(R, 4)), dim3(256), 0, st, l->h, k.ln2, l->xn, R, H, c.rms_eps, ring ? l->pl3 : nullptr);
            GemmEpi g;
            g.out = l->gu; g.out_bf16 = 0; g.ldc = 2 * I;
            FY_TRY(exact(ring ? nullptr : l->xn, H, k.rgu, k.rgu_lo, 2 * I, g));
            hipLaunchKernelGGL(swiglu_rows_k, dim3(std::min(4096, cdiv(R * I, ring ? 512 : 256))), dim3(256), 0, st, l->gu, l->actf, (long)R * I, I, ring ? l->pl3 : nullptr);
            GemmEpi d;
            d.mode = EPI_GATE_RESID; d.resid = l->h; d.gate = l->ones; d.ldc = H;
            FY_TRY(exact(ring ? nullptr : l->actf, I, k.rd, k.rd_lo, H, d));
            continue;
        }
        if (l->gv32) {
            // 32 rows per weight pass; the caller left the A image of (ln1 x h) and its sums of squares (prefill: gv32_split_rows,
            // decode: the sampler), every product's epilogue leaves what the next one reads
            const int NP = H / 32;
            Gv32Args q;
            q.W = k.wqkv; q.W_lo = k.wqkv_lo; q.img = l->img_h; q.R = R; q.N = QKV; q.K = H; q.ssq = l->ssq; q.n_ssq = NP; q.eps = c.rms_eps; q.bias = k.bqkv;
            q.y = l->qkv; q.ldy = QKV;
            FY_TRY(gemv32(q, st));
            if (decode) {
                FY_TRY(llm_attention_step(l->qkv, Kc, Vc, row_seq, row_pos, l->inv_freq, nullptr, H, R, c.q_heads, c.kv_heads, l->max_ctx, st, l->img_ao));
            } else {
                hipLaunchKernelGGL(rope_kv_k, dim3(R), dim3(256), 0, st, l->qkv, Kc, Vc, row_seq, row_pos, l->inv_freq, c.q_heads, c.kv_heads, l->max_ctx);
                FY_TRY(llm_attention(l->qkv, QKV, Kc, Vc, row_seq, row_pos, l->ao, H, R, c.q_heads, c.kv_heads, l->max_ctx, st));
                FY_TRY(gv32_split_rows(l->ao, H, R, H, nullptr, l->img_ao, nullptr, st));
            }
            Gv32Args o;
            o.W = k.wo; o.W_lo = k.wo_lo; o.img = l->img_ao; o.R = R; o.N = H; o.K = H; o.mode = GV32_ADD_IMG; o.y = l->h; o.ldy = H;
            o.ln_next = k.ln2; o.img_out = l->img_h; o.ssq_out = l->ssq;
            FY_TRY(gemv32(o, st));
            Gv32Args g;
            g.W = k.wgu; g.W_lo = k.wgu_lo; g.img = l->img_h; g.R = R; g.N = 2 * I; g.K = H; g.ssq = l->ssq; g.n_ssq = NP; g.eps = c.rms_eps;
            g.mode = GV32_SWIGLU_IMG; g.img_out = l->img_act;
            FY_TRY(gemv32(g, st));
            Gv32Args d;
            d.W = k.wd; d.W_lo = k.wd_lo; d.img = l->img_act; d.R = R; d.N = H; d.K = I; d.mode = GV32_ADD_IMG; d.y = l->h; d.ldy = H;
            d.ln_next = i + 1 < c.layers ? l->L[i + 1].ln1 : l->norm_w; d.img_out = l->img_h; d.ssq_out = l->ssq;
            d.partial = l->part32; d.counters = l->cnt32;
            FY_TRY(gemv32(d, st));
            continue;
        }
        GemvArgs a;          // input RMSNorm fused into the projection
        a.W = k.wqkv; a.x = l->h; a.ldx = H; a.R = R; a.N = QKV; a.K = H; a.bias = k.bqkv; a.y = l->qkv; a.ldy = QKV;
        a.norm_w = k.ln1; a.eps = c.rms_eps;
        FY_TRY(gemv_bf16w(a, st));
        if (decode) {
            FY_TRY(llm_attention_step(l->qkv, Kc, Vc, row_seq, row_pos, l->inv_freq, l->ao, H, R, c.q_heads, c.kv_heads, l->max_ctx, st));
        } else {
            // prefill: every row's key must be in the cache before any row attends
            hipLaunchKernelGGL(rope_kv_k, dim3(R), dim3(256), 0, st, l->qkv, Kc, Vc, row_seq, row_pos, l->inv_freq, c.q_heads, c.kv_heads, l->max_ctx);
            FY_TRY(llm_attention(l->qkv, QKV, Kc, Vc, row_seq, row_pos, l->ao, H, R, c.q_heads, c.kv_heads, l->max_ctx, st));
        }
        GemvArgs o;
        o.W = k.wo; o.x = l->ao; o.ldx = H; o.R = R; o.N = H; o.K = H; o.y = l->h; o.ldy = H; o.mode = GV_ADD;
        FY_TRY(gemv_bf16w(o, st));
        GemvArgs g;          // post-attention RMSNorm fused
        g.W = k.wgu; g.x = l->h; g.ldx = H; g.R = R; g.N = 2 * I; g.K = H; g.y_split = l->act_split; g.ldy = I; g.mode = GV_SWIGLU_SPLIT;
        g.norm_w = k.ln2; g.eps = c.rms_eps;
        FY_TRY(gemv_bf16w(g, st));
        GemvArgs d;          // K = inter is too long to stage: the A operand arrives pre-split from the SwiGLU epilogue
        d.W = k.wd; d.x_split = l->act_split; d.ldx = I; d.R = R; d.N = H; d.K = I; d.y = l->h; d.ldy = H; d.mode = GV_ADD;
        d.partial = l->partial; d.counters = l->counters;
        FY_TRY(gemv_bf16w(d, st));
    }
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

static int llm_sample(fy_llm* l, int B, int32_t* out_ids, int out_ld, int keep_step, hipStream_t st);
// have_img: the A image of (norm.weight x rows) and its sums of squares already stand in img_h / ssq (the last layer's down
// projection left them: a decode step on the 32-row path)
static int llm_head_and_sample(fy_llm* l, int B, const float* rows, int32_t* out_ids, int out_ld, int keep_step, hipStream_t st, bool have_img = false) {
    const fy_llm_config& c = l->cfg;
    const int H = c.hidden, NS = l->n_speech();
    if (l->gv32) {
        if (!have_img) FY_TRY(gv32_split_rows(rows, H, B, H, l->norm_w, l->img_h, l->ssq, st));
        Gv32Args a;
        a.W = l->w_head; a.W_lo = l->w_head_lo; a.img = l->img_h; a.R = B; a.N = NS; a.K = H; a.ssq = l->ssq; a.n_ssq = H / 32; a.eps = c.rms_eps; a.y = l->logits; a.ldy = NS;
        FY_TRY(gemv32(a, st));
        return llm_sample(l, B, out_ids, out_ld, keep_step, st);
    }
    GemvArgs a;              // final RMSNorm fused into the llm_decoder product
    a.W = l->w_head; a.x = rows; a.ldx = H; a.R = B; a.N = NS; a.K = H; a.y = l->logits; a.ldy = NS;
    a.norm_w = l->norm_w; a.eps = c.rms_eps;
    FY_TRY(gemv_bf16w(a, st));
    return llm_sample(l, B, out_ids, out_ld, keep_step, st);
}

// logits (B, n_speech) -> the sampler: token, stop flags, silent-token filter, next input embedding into h
static int llm_sample(fy_llm* l, int B, int32_t* out_ids, int out_ld, int keep_step, hipStream_t st) {
    const fy_llm_config& c = l->cfg;
    const int H = c.hidden, NS = l->n_speech();
    FY_CHECK(NS <= 256 * 32, FY_ERR_ARG, "sample: %d logits exceed the kernel's register tile", NS);
    NextImg ni;              // the next decode step's first operand (32-row path): image of ln1[0] x speech_embedding[id]
    if (l->gv32) { ni.img = l->img_h; ni.ssq = l->ssq; ni.ln = l->L[0].ln1; ni.Rpad = 32 * cdiv(B, 32); }
    if (l->sampler == 1)
        hipLaunchKernelGGL(sample_ras_k, dim3(B), dim3(256), (size_t)NS * sizeof(float), st, l->logits, NS, c.speech_tokens, l->st, l->max_batch,
                           out_ids, out_ld, l->speech_emb, l->h, H, l->logp_keep, keep_step < FY_LLM_KEEP_LOGP ? keep_step : -1,
                           l->uniforms, l->n_uniforms, l->recent, l->top_k, l->top_p, l->win_size, l->rep_thr, ni);
    else
        hipLaunchKernelGGL(sample_k, dim3(B), dim3(256), 0, st, l->logits, NS, c.speech_tokens, l->st, l->max_batch, out_ids, out_ld,
                           l->speech_emb, l->h, H, l->logp_keep, keep_step < FY_LLM_KEEP_LOGP ? keep_step : -1, ni);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// Common tail of fy_llm_begin / fy_llm_prefill: per-sequence state, the prefill over the R packed rows already standing in l->h
// (fp32, sequence after sequence, n_rows[b] each), the first token.
static int llm_begin_rows(fy_llm* l, const int* n_rows, const int32_t* min_len, const int32_t* max_len, int B, int32_t* out_ids, int out_ld, hipStream_t st) {
    const fy_llm_config& c = l->cfg;
    const int mb = l->max_batch, H = c.hidden;
    std::vector<int> rseq, rpos, last(mb, 0), stv(8 * mb, 0);
    int steps = 0;
    for (int b = 0; b < B; ++b) {
        const int Lb = n_rows[b];
        for (int p = 0; p < Lb; ++p) { rseq.push_back(b); rpos.push_back(p); }
        last[b] = (int)rseq.size() - 1;
        stv[0 * mb + b] = Lb - 1;          // position of the last prefill token; sample_k advances it
        stv[5 * mb + b] = min_len[b];
        stv[6 * mb + b] = max_len[b];
        steps = std::max(steps, (int)max_len[b]);
    }
    for (int b = B; b < mb; ++b) stv[3 * mb + b] = 1;
    const int R = (int)rseq.size();
    HIP_TRY(hipMemcpyAsync(l->row_seq, rseq.data(), R * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(l->row_pos, rpos.data(), R * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(l->last_row, last.data(), mb * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(l->st, stv.data(), stv.size() * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(stream_wait(st));
    l->B = B;
    if (l->gv32 && !llm_prefill_by_gemm(l, R)) FY_TRY(gv32_split_rows(l->h, H, R, H, l->L[0].ln1, l->img_h, l->ssq, st));
    FY_TRY(llm_layers(l, R, l->row_seq, l->row_pos, false, st));
    hipLaunchKernelGGL(gather_rows_k, dim3(B), dim3(256), 0, st, l->h, l->last_row, l->hb, H);
    FY_TRY(llm_head_and_sample(l, B, l->hb, out_ids, out_ld, 0, st));
    l->step_next = 1; l->steps_cap = steps;
    return FY_OK;
}

extern "C" int fy_llm_begin(fy_llm* l, const int32_t* text_ids, const int32_t* n_text_all, const int32_t* prompt_speech,
                            const int32_t* n_prompt_speech, const int32_t* min_len, const int32_t* max_len, int32_t B,
                            int32_t* out_ids, int32_t out_ld, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    FY_CHECK(l && text_ids && n_text_all && n_prompt_speech && min_len && max_len && out_ids, FY_ERR_ARG, "fy_llm_generate: null argument");
    l->step_next = l->steps_cap = 0;
    l->all_done = false;
    FY_CHECK(B >= 1 && B <= l->max_batch && out_ld >= 1, FY_ERR_ARG, "fy_llm_generate: batch %d outside [1, %d]", B, l->max_batch);
    const fy_llm_config& c = l->cfg;
    const int H = c.hidden;
    // lm_input = [sos, embed(prompt_text + text), task_id, speech_embedding(prompt_speech)], llm.py:732-740
    std::vector<int> src, rows(B, 0);
    int to = 0, po = 0;
    for (int b = 0; b < B; ++b) {
        const int nt = n_text_all[b], np = n_prompt_speech[b], Lb = 2 + nt + np;
        FY_CHECK(nt >= 1 && np >= 0 && min_len[b] >= 0 && max_len[b] >= 1, FY_ERR_ARG, "fy_llm_generate: sequence %d has bad lengths", b);
        FY_CHECK(Lb + max_len[b] <= l->max_ctx, FY_ERR_ARG, "fy_llm_generate: sequence %d needs %d positions, the handle has %d", b,
                 Lb + max_len[b], l->max_ctx);
        FY_CHECK(np == 0 || prompt_speech, FY_ERR_ARG, "fy_llm_generate: prompt_speech is null");
        for (int p = 0; p < Lb; ++p) {
            int s;
            if (p == 0) s = c.speech_tokens | (1 << 30);
            else if (p <= nt) {
                int id = text_ids[to + p - 1];
                FY_CHECK(id >= 0 && id < c.vocab, FY_ERR_ARG, "fy_llm_generate: text id %d outside the vocabulary", id);
                s = id;
            } else if (p == nt + 1) s = (c.speech_tokens + 2) | (1 << 30);
            else {
                int id = prompt_speech[po + p - nt - 2];
                FY_CHECK(id >= 0 && id < l->n_speech(), FY_ERR_ARG, "fy_llm_generate: prompt speech id %d out of range", id);
                s = id | (1 << 30);
            }
            src.push_back(s);
        }
        rows[b] = Lb;
        to += nt; po += np;
    }
    const int R = (int)src.size();
    FY_CHECK(R <= l->max_rows, FY_ERR_ARG, "fy_llm_generate: %d prefill rows exceed the workspace (%d)", R, l->max_rows);
    HIP_TRY(hipMemcpyAsync(l->row_src, src.data(), R * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(stream_wait(st));                        // src dies with this frame
    hipLaunchKernelGGL(embed_rows_k, dim3(R), dim3(256), 0, st, l->row_src, l->embed_tokens, l->embed_tokens_f32, l->speech_emb, l->h, H);
    return llm_begin_rows(l, rows.data(), min_len, max_len, B, out_ids, out_ld, st);
}

// The embeddings-level entry (the reference's vLLM hand-off, llm.py:482-510: the host assembles lm_input itself and
// passes `prompt_embeds`).  embeds: the B sequences' rows packed back to back, (sum n_rows, hidden), fp32 (dtype 0) or bf16 (1).
__global__ void rows_from_bf16_k(const bf16_t* __restrict__ src, float* __restrict__ dst, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = bf16_to_f32(src[i]);
}
extern "C" int fy_llm_prefill(fy_llm* l, const void* embeds, int32_t dtype, const int32_t* n_rows, const int32_t* min_len, const int32_t* max_len,
                              int32_t B, int32_t* out_ids, int32_t out_ld, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    FY_CHECK(l && embeds && n_rows && min_len && max_len && out_ids && (dtype == 0 || dtype == 1), FY_ERR_ARG, "fy_llm_prefill: bad argument (dtype 0 = fp32, 1 = bf16)");
    l->step_next = l->steps_cap = 0;
    l->all_done = false;
    FY_CHECK(B >= 1 && B <= l->max_batch && out_ld >= 1, FY_ERR_ARG, "fy_llm_prefill: batch %d outside [1, %d]", B, l->max_batch);
    const int H = l->cfg.hidden;
    long R = 0;
    for (int b = 0; b < B; ++b) {
        FY_CHECK(n_rows[b] >= 1 && min_len[b] >= 0 && max_len[b] >= 1, FY_ERR_ARG, "fy_llm_prefill: sequence %d has bad lengths", b);
        FY_CHECK(n_rows[b] + max_len[b] <= l->max_ctx, FY_ERR_ARG, "fy_llm_prefill: sequence %d needs %d positions, the handle has %d", b,
                 n_rows[b] + max_len[b], l->max_ctx);
        R += n_rows[b];
    }
    FY_CHECK(R <= l->max_rows, FY_ERR_ARG, "fy_llm_prefill: %ld prefill rows exceed the workspace (%d)", R, l->max_rows);
    if (dtype == 0) HIP_TRY(hipMemcpyAsync(l->h, embeds, (size_t)R * H * sizeof(float), hipMemcpyDeviceToDevice, st));
    else hipLaunchKernelGGL(rows_from_bf16_k, dim3((unsigned)std::min<long>(2048, (R * H + 255) / 256)), dim3(256), 0, st, (const bf16_t*)embeds, l->h, R * H);
    HIP_TRY(hipGetLastError());
    return llm_begin_rows(l, n_rows, min_len, max_len, B, out_ids, out_ld, st);
}

// decode: row b = sequence b at position st[pos][b]; everything a step needs is on the device
extern "C" int fy_llm_step(fy_llm* l, int32_t n_steps, int32_t* out_ids, int32_t out_ld, int32_t* out_n, int32_t* raw_n,
                           int32_t* finished, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    FY_CHECK(l && out_ids && out_n && out_ld >= 1 && n_steps >= 0, FY_ERR_ARG, "fy_llm_step: bad argument");
    FY_CHECK(l->step_next >= 1, FY_ERR_STATE, "fy_llm_step: no generation in progress (call fy_llm_begin first)");
    const int mb = l->max_batch, B = l->B;
    const int end = l->all_done ? l->step_next : (int)std::min<long>((long)l->step_next + n_steps, l->steps_cap);
    std::vector<int> done(mb, 0);
    unsigned dec_status = 0;
    int step = l->step_next;
    const bool persistent = l->dec && l->decode_mode == 1 && B <= 8;
    const bool persistent32 = !persistent && l->dec32 && (l->decode_mode == 1 || l->decode_mode == 2) && B <= 32;
    auto read_done = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(done.data(), l->st + 3 * mb, mb * sizeof(int), hipMemcpyDeviceToHost, st));
        // only a call that used the persistent step looks at (and clears) its time-out word: a time-out is reported once,
        // and the handle then works on the per-operation path (fy_llm_set_decode_mode(0)) as the message says
        if (persistent) FY_TRY(decode_status(l->dec, &dec_status, st));
        if (persistent32) FY_TRY(decode32_status(l->dec32, &dec_status, st));
        HIP_TRY(stream_wait(st));
        if (dec_status != 0) {
            l->step_next = 0;                // the generation is void: no further steps are launched on it
            FY_CHECK(false, FY_ERR_STATE, "fy_llm_step: the decode kernel's grid hand-off timed out (its workgroups were not all resident: "
                     "another persistent grid on the device, or a CU-masked stream?); call fy_llm_set_decode_mode(handle, 0) or set "
                     "FY_LLM_PERSISTENT=0 to use the multi-launch path, then start the generation again");
        }
        return FY_OK;
    };
    for (; step < end; ++step) {
        if (persistent) {                    // the 24 layers + llm_decoder of the step in one launch, then the sampler
            FY_TRY(decode_step(l->dec, B, l->h, l->st, l->inv_freq, l->logits, st));
            FY_TRY(llm_sample(l, B, out_ids, out_ld, step, st));
        } else if (persistent32) {           // the same in one launch on a few CUs for up to 32 sequences (operands: the 32-row path's images)
            FY_TRY(decode32_step(l->dec32, B, l->h, l->img_h, l->ssq, l->qkv, l->img_ao, l->st, l->inv_freq, l->logits, st));
            FY_TRY(llm_sample(l, B, out_ids, out_ld, step, st));
        } else {
            FY_TRY(llm_layers(l, B, l->seq_ids, l->st, true, st));
            FY_TRY(llm_head_and_sample(l, B, l->h, out_ids, out_ld, step, st, l->gv32));
        }
        if ((step & 7) == 7) {
            FY_TRY(read_done());
            bool all = true;
            for (int b = 0; b < B; ++b) all = all && done[b];
            if (all) { ++step; break; }
        }
    }
    l->step_next = step;
    HIP_TRY(hipMemcpyAsync(out_n, l->st + 2 * mb, B * sizeof(int), hipMemcpyDeviceToDevice, st));
    if (raw_n) HIP_TRY(hipMemcpyAsync(raw_n, l->st + 1 * mb, B * sizeof(int), hipMemcpyDeviceToDevice, st));
    FY_TRY(read_done());
    {
        bool all = true;
        for (int b = 0; b < B; ++b) all = all && done[b];
        l->all_done = all;
    }
    if (finished)
        for (int b = 0; b < B; ++b) finished[b] = (done[b] != 0 || l->step_next >= l->steps_cap) ? 1 : 0;
    if (l->sampler == 1) {                   // the reference raises RuntimeError from sampling_ids (llm.py:161-162)
        std::vector<int> used(mb);
        HIP_TRY(hipMemcpyAsync(used.data(), l->st + 7 * mb, mb * sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(stream_wait(st));
        for (int b = 0; b < B; ++b) {
            FY_CHECK(done[b] != 2, FY_ERR_STATE, "sampling reaches max_trials 100 and still get eos when ignore_eos is True (sequence %d)", b);
            FY_CHECK(used[b] <= l->n_uniforms, FY_ERR_ARG, "fy_llm_generate: sequence %d needed %d uniforms, %ld were supplied", b, used[b], l->n_uniforms);
        }
    }
    return FY_OK;
}

extern "C" int fy_llm_generate(fy_llm* l, const int32_t* text_ids, const int32_t* n_text_all, const int32_t* prompt_speech,
                               const int32_t* n_prompt_speech, const int32_t* min_len, const int32_t* max_len, int32_t B,
                               int32_t* out_ids, int32_t out_ld, int32_t* out_n, int32_t* raw_n, uint32_t flags, void* stream) {
    (void)flags;
    FY_CHECK(out_n, FY_ERR_ARG, "fy_llm_generate: null argument");
    FY_TRY(fy_llm_begin(l, text_ids, n_text_all, prompt_speech, n_prompt_speech, min_len, max_len, B, out_ids, out_ld, stream));
    return fy_llm_step(l, l->steps_cap, out_ids, out_ld, out_n, raw_n, nullptr, stream);
}


extern "C" int fy_llm_set_sampler(fy_llm* l, int32_t kind, const float* uniforms, int64_t n_uniforms, int32_t top_k, float top_p,
                                  int32_t win_size, float tau_r) {
    FY_CHECK(l && (kind == 0 || kind == 1), FY_ERR_ARG, "fy_llm_set_sampler: kind must be 0 (greedy) or 1 (repetition-aware sampling)");
    if (kind == 1) {
        FY_CHECK(uniforms && n_uniforms >= 1 && top_k >= 1 && top_k <= 32 && top_p > 0.f && win_size >= 1 && win_size <= 64 && tau_r >= 0.f,
                 FY_ERR_ARG, "fy_llm_set_sampler: bad sampling parameters");
        if (!l->recent || win_size != l->win_size) {
            if (l->recent) (void)hipFree(l->recent);
            l->recent = nullptr;
            HIP_TRY(hipMalloc(&l->recent, (size_t)win_size * l->max_batch * sizeof(int)));
        }
        l->uniforms = uniforms; l->n_uniforms = n_uniforms; l->top_k = top_k; l->top_p = top_p; l->win_size = win_size;
        l->rep_thr = (float)((double)win_size * (double)tau_r);
    }
    l->sampler = kind;
    return FY_OK;
}

extern "C" int fy_llm_logp(fy_llm* l, int32_t step, float* dst, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    FY_CHECK(l && dst && step >= 0 && step < FY_LLM_KEEP_LOGP && l->B > 0, FY_ERR_ARG, "fy_llm_logp: bad argument or no call made yet");
    const int NS = l->n_speech();
    for (int b = 0; b < l->B; ++b)
        HIP_TRY(hipMemcpyAsync(dst + (size_t)b * NS, l->logp_keep + ((size_t)step * l->max_batch + b) * NS, NS * sizeof(float),
                               hipMemcpyDeviceToDevice, st));
    return FY_OK;
}

// diagnostic, not part of the ABI header: phase time stamps of the persistent decode kernel's last launch
extern "C" int fy_debug_decode_stamps(fy_llm* l, unsigned long long* out, int32_t n, void* stream) {
    FY_CHECK(l && l->dec, FY_ERR_STATE, "fy_debug_decode_stamps: this handle has no persistent decode plan");
    return decode_stamps(l->dec, out, n, (hipStream_t)stream);
}

extern "C" int fy_debug_decode32_stamps(fy_llm* l, unsigned long long* out, int32_t n, void* stream) {
    FY_CHECK(l && l->dec32, FY_ERR_STATE, "fy_debug_decode32_stamps: this handle has no 32-row persistent decode plan");
    return decode32_stamps(l->dec32, out, n, (hipStream_t)stream);
}

extern "C" int fy_llm_set_decode_mode(fy_llm* l, int32_t mode) {
    FY_CHECK(l && (mode == 0 || mode == 1 || mode == 2), FY_ERR_ARG,
             "fy_llm_set_decode_mode: mode must be 0 (one launch per operation), 1 (persistent step) or 2 (the few-CU persistent step only)");
    l->decode_mode = mode;
    return FY_OK;
}

extern "C" int fy_llm_decode_mode(const fy_llm* l) {
    if (!l) return 0;
    if (l->decode_mode == 1) return (l->dec || l->dec32) ? 1 : 0;
    return l->decode_mode == 2 && l->dec32 ? 2 : 0;
}

extern "C" int fy_llm_weight_planes(const fy_llm* l) { return l ? l->planes : 0; }
