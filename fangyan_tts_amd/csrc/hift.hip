// HiFT vocoder engine: CausalHiFTGenerator.inference (finalize=True),
// CosyVoice/cosyvoice/hifigan/generator.py:713-726, as a batch of ragged
// utterances.  Activations live in HBM as fp32 channels-last (B, L, C); every
// convolution reads its input once and writes its output once, with the
// neighbouring element-wise work (Snake, leaky-relu, nearest upsample, reflect
// pad, residual, source fusion, the /3 average) folded into it.
#include "conv.h"
#include <stdlib.h>
#include "runtime.h"
#include <math.h>

#define N_UP 3
#define N_RB 3
#define N_DIL 3

struct HiftConvs {
    ConvW conv_pre, conv_post;
    ConvW ups[N_UP], source_downs[N_UP];
    ConvW f0c[5];
    struct RB { ConvW c1[N_DIL], c2[N_DIL]; float *a1[N_DIL], *a2[N_DIL]; } src_rb[N_UP], rb[N_UP * N_RB];
};

struct fy_hift {
    fy_hift_config cfg;
    int max_batch = 0, max_frames = 0;
    int up_total = 0, stft_per_frame = 0;
    HiftConvs w;
    float *cls_w = nullptr, *cls_b = nullptr;       // f0 classifier (1, f0_ch), (1)
    float *lin_w = nullptr, *lin_b = nullptr;       // m_source.l_linear (1, H+1), (1)
    DevPool pool, wpool;
    // activations (channels-last)
    float *mel_cl, *f0a, *f0b, *f0, *rad_phase, *source, *s_stft, *x_pre, *post, *spec;
    float *xs[N_UP], *x[N_UP], *r[N_UP], *xt[N_UP], *si[N_UP];
    // three bf16 streams per stage: the up-conv's output under the first snake of each of the stage's ResBlocks, written by the up-conv's
    // epilogue (and slot 0, earlier in the stage, the source down-sampler's output under the source ResBlock's first snake)
    bf16_t* sa[N_UP] = {nullptr, nullptr, nullptr};
    // bf16 'super-row' copies of s_stft for the strided source down-samplers (see stft16_k); null when not built
    bf16_t* sb[N_UP] = {nullptr, nullptr, nullptr};
    int sb_ld[N_UP] = {0, 0, 0};
    ConvW sd_mfma[N_UP];
    ConvW ups_poly[N_UP];                            // ups[i] in polyphase form (conv_pack_polyphase)
    int *lens;                                       // device: [8][max_batch] = F, L0, L1, L2conv, L2, S_out, F_f0, F_dec (set_lens)
    int B = 0, Fmax = 0;                             // of the last call (for taps)
    int L(int stage, int F) const {                  // rows of stage tensors for F frames
        int l = F;
        for (int i = 0; i <= stage; ++i) l *= cfg.ups[i];
        return stage == N_UP - 1 ? l + 1 : l;
    }
    int C(int stage) const { return cfg.base >> (stage + 1); }
    ~fy_hift();
};

#define POST_LD 20     // fp32 pitch of the conv_post output rows (18 channels, rounded up for 16-byte row stores)

extern "C" void fy_hift_default_config(fy_hift_config* c) {
    memset(c, 0, sizeof(*c));
    c->mel = 80; c->base = 512; c->harmonics = 8; c->sampling_rate = 24000;
    c->nsf_alpha = 0.1f; c->nsf_sigma = 0.003f; c->voiced_thr = 10.f;
    const int ups[3] = {8, 5, 3}, upk[3] = {16, 11, 7}, rbk[3] = {3, 7, 11}, rbd[3] = {1, 3, 5}, srk[3] = {7, 7, 11};
    for (int i = 0; i < 3; ++i) { c->ups[i] = ups[i]; c->up_k[i] = upk[i]; c->rb_k[i] = rbk[i]; c->rb_d[i] = rbd[i]; c->src_rb_k[i] = srk[i]; }
    c->n_fft = 16; c->hop = 4; c->lrelu = 0.1f; c->audio_limit = 0.99f; c->pre_look_right = 4; c->f0_ch = 512;
}

static void source_down_shape(const fy_hift_config& c, int i, int* k, int* stride) {
    // generator.py:642-652: downsample_cum_rates[::-1][i]
    int rates[N_UP] = {1, c.ups[2], c.ups[1]};
    int cum[N_UP] = {rates[0], rates[0] * rates[1], rates[0] * rates[1] * rates[2]};
    int u = cum[N_UP - 1 - i];
    if (u == 1) { *k = 1; *stride = 1; } else { *k = 2 * u; *stride = u; }
}

fy_hift::~fy_hift() {
    conv_free(w.conv_pre); conv_free(w.conv_post);
    for (int i = 0; i < N_UP; ++i) { conv_free(w.ups[i]); conv_free(w.source_downs[i]); conv_free(sd_mfma[i]); conv_free(ups_poly[i]); }
    for (int i = 0; i < 5; ++i) conv_free(w.f0c[i]);
    auto free_rb = [](HiftConvs::RB& r) { for (int j = 0; j < N_DIL; ++j) { conv_free(r.c1[j]); conv_free(r.c2[j]); } };
    for (int i = 0; i < N_UP; ++i) free_rb(w.src_rb[i]);
    for (int i = 0; i < N_UP * N_RB; ++i) free_rb(w.rb[i]);
}

// ---- weight loading ---------------------------------------------------------------
static int load_wn_conv(fy_hift* h, const Weights& W, const std::string& p, ConvW& cw, int Cout, int Cin, int KW, bool direct, bool mfma, hipStream_t st,
                        bool pad_cout = false) {
    const float* g = W.get(p + ".parametrizations.weight.original0", {Cout, 1, 1});
    const float* v = W.get(p + ".parametrizations.weight.original1", {Cout, Cin, KW});
    const float* b = W.get(p + ".bias", {Cout});
    if (!g || !v || !b) return FY_ERR_WEIGHT;
    if (mfma && ((Cout % 32 != 0 && !pad_cout) || Cin % 4 != 0)) { mfma = false; direct = true; }
    return conv_pack(cw, v, g, b, Cout, Cin, KW, 1, direct, mfma, st);
}

static int dev_copy(DevPool& pool, float** dst, const float* src, size_t n, hipStream_t st) {
    FY_TRY(pool.alloc(dst, n));
    HIP_TRY(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    return FY_OK;
}

static int load_rb(fy_hift* h, const Weights& W, const std::string& p, HiftConvs::RB& rb, int ch, int k, hipStream_t st) {
    for (int j = 0; j < N_DIL; ++j) {
        FY_TRY(load_wn_conv(h, W, p + ".convs1." + std::to_string(j), rb.c1[j], ch, ch, k, true, true, st));
        FY_TRY(load_wn_conv(h, W, p + ".convs2." + std::to_string(j), rb.c2[j], ch, ch, k, true, true, st));
        const float* a1 = W.get(p + ".activations1." + std::to_string(j) + ".alpha", {ch});
        const float* a2 = W.get(p + ".activations2." + std::to_string(j) + ".alpha", {ch});
        if (!a1 || !a2) return FY_ERR_WEIGHT;
        FY_TRY(dev_copy(h->wpool, &rb.a1[j], a1, ch, st));
        FY_TRY(dev_copy(h->wpool, &rb.a2[j], a2, ch, st));
    }
    return FY_OK;
}

// The strided source down-samplers (k = 2s, stride s, s-1 zero rows on the left; k = 1 when s = 1) as stride-1 convs:
// output p reads s_stft rows s*p - s + 1 .. s*p + s, i.e. the two 'super-rows' S'[p], S'[p+1] where S'[j] holds rows
// s*j - s + 1 .. s*j side by side (18 s channels, rounded up to `ld`).  A k = 2 (or 1) conv over Cin' = ld channels then
// runs on the MFMA kernel from the bf16 copies stft16_k writes.  v' [Cout][ld][kk] <- v [Cout][18][k].
__global__ void super_pack_k(const float* __restrict__ v, float* __restrict__ out, int Cout, int C, int k, int s, int ld, int kk) {
    long n = (long)Cout * ld * kk;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        int t2 = i % kk;
        long r = i / kk;
        int cp = r % ld, co = r / ld;
        int pos = cp / C, c = cp % C;
        out[i] = pos < s ? v[((long)co * C + c) * k + t2 * s + pos] : 0.f;
    }
}

static int super_ld(int C, int s) {
    int n = C * s;
    return n <= 128 ? (n + 31) / 32 * 32 : (n + 127) / 128 * 128;
}

extern "C" int fy_hift_create(fy_hift** out, const fy_hift_config* cfg, const fy_tensor* weights, int32_t n_weights,
                              int32_t max_batch, int32_t max_frames, void* stream) {
    FY_CHECK(out && max_batch >= 1 && max_frames >= 1, FY_ERR_ARG, "fy_hift_create: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    fy_hift* h = new fy_hift();
    if (cfg) h->cfg = *cfg; else fy_hift_default_config(&h->cfg);
    const fy_hift_config& c = h->cfg;
    h->max_batch = max_batch; h->max_frames = max_frames;
    h->up_total = c.hop * c.ups[0] * c.ups[1] * c.ups[2];
    h->stft_per_frame = h->up_total / c.hop;
    Weights W;
    int rc = W.init(weights, n_weights);
    auto fail = [&](int code) { delete h; return code; };
    if (rc) return fail(rc);
    if (c.n_fft != 16 || c.hop != 4) { fy_set_error("fy_hift_create: only n_fft 16 / hop 4 is built"); return fail(FY_ERR_ARG); }
#define TRYC(e) do { int _r = (e); if (_r) return fail(_r); } while (0)
    TRYC(load_wn_conv(h, W, "conv_pre", h->w.conv_pre, c.base, c.mel, c.pre_look_right + 1, true, true, st));
    TRYC(load_wn_conv(h, W, "conv_post", h->w.conv_post, c.n_fft + 2, h->C(N_UP - 1), 7, true, true, st, true));
    for (int i = 0; i < N_UP; ++i) {
        TRYC(load_wn_conv(h, W, "ups." + std::to_string(i), h->w.ups[i], h->C(i), c.base >> i, c.up_k[i], true, true, st));
        if ((h->C(i) * c.ups[i]) % 32 == 0 && h->C(i) % 4 == 0 && (c.base >> i) % 4 == 0 && c.ups[i] > 1) {
            const std::string p = "ups." + std::to_string(i);
            TRYC(conv_pack_polyphase(h->ups_poly[i], W.get(p + ".parametrizations.weight.original1", {h->C(i), c.base >> i, c.up_k[i]}),
                                     W.get(p + ".parametrizations.weight.original0", {h->C(i), 1, 1}), W.get(p + ".bias", {h->C(i)}),
                                     h->C(i), c.base >> i, c.up_k[i], c.ups[i], st));
        }
        int k, s;
        source_down_shape(c, i, &k, &s);
        const float* v = W.get("source_downs." + std::to_string(i) + ".weight", {h->C(i), c.n_fft + 2, k});
        const float* b = W.get("source_downs." + std::to_string(i) + ".bias", {h->C(i)});
        if (!v || !b) return fail(FY_ERR_WEIGHT);
        TRYC(conv_pack(h->w.source_downs[i], v, nullptr, b, h->C(i), c.n_fft + 2, k, 1, true, false, st));
        if (h->C(i) % 32 == 0) {
            const int C18 = c.n_fft + 2, ld = super_ld(C18, s), kk = s == 1 ? 1 : 2;
            float* vs = nullptr;
            if (hipMalloc(&vs, (size_t)h->C(i) * ld * kk * sizeof(float)) != hipSuccess) { fy_set_error("fy_hift_create: hipMalloc failed"); return fail(FY_ERR_HIP); }
            hipLaunchKernelGGL(super_pack_k, dim3(256), dim3(256), 0, st, v, vs, h->C(i), C18, k, s, ld, kk);
            int rc2 = conv_pack(h->sd_mfma[i], vs, nullptr, b, h->C(i), ld, kk, 1, false, true, st);     // synchronises st
            (void)hipFree(vs);
            if (rc2) return fail(rc2);
            h->sb_ld[i] = ld;
        }
        TRYC(load_rb(h, W, "source_resblocks." + std::to_string(i), h->w.src_rb[i], h->C(i), c.src_rb_k[i], st));
        for (int j = 0; j < N_RB; ++j)
            TRYC(load_rb(h, W, "resblocks." + std::to_string(i * N_RB + j), h->w.rb[i * N_RB + j], h->C(i), c.rb_k[j], st));
    }
    TRYC(load_wn_conv(h, W, "f0_predictor.condnet.0", h->w.f0c[0], c.f0_ch, c.mel, 4, true, false, st));
    for (int i = 1; i < 5; ++i)
        TRYC(load_wn_conv(h, W, "f0_predictor.condnet." + std::to_string(2 * i), h->w.f0c[i], c.f0_ch, c.f0_ch, 3, true, false, st));
    {
        const float* cw = W.get("f0_predictor.classifier.weight", {1, c.f0_ch});
        const float* cb = W.get("f0_predictor.classifier.bias", {1});
        const float* lw = W.get("m_source.l_linear.weight", {1, c.harmonics + 1});
        const float* lb = W.get("m_source.l_linear.bias", {1});
        if (!cw || !cb || !lw || !lb) return fail(FY_ERR_WEIGHT);
        TRYC(dev_copy(h->wpool, &h->cls_w, cw, c.f0_ch, st));
        TRYC(dev_copy(h->wpool, &h->cls_b, cb, 1, st));
        TRYC(dev_copy(h->wpool, &h->lin_w, lw, c.harmonics + 1, st));
        TRYC(dev_copy(h->wpool, &h->lin_b, lb, 1, st));
    }
    // activations
    const size_t B = max_batch, F = max_frames;
    TRYC(h->pool.alloc(&h->mel_cl, B * F * c.mel));
    TRYC(h->pool.alloc(&h->f0a, B * F * c.f0_ch));
    TRYC(h->pool.alloc(&h->f0b, B * F * c.f0_ch));
    TRYC(h->pool.alloc(&h->f0, B * F));
    TRYC(h->pool.alloc(&h->rad_phase, B * F * (c.harmonics + 1)));
    TRYC(h->pool.alloc(&h->source, B * F * h->up_total));
    const size_t T = F * h->stft_per_frame + 1;
    TRYC(h->pool.alloc(&h->s_stft, B * T * (c.n_fft + 2)));
    TRYC(h->pool.alloc(&h->x_pre, B * F * c.base));
    TRYC(h->pool.alloc(&h->post, B * T * POST_LD));
    for (int i = 0; i < N_UP; ++i) {
        size_t n = B * (size_t)h->L(i, (int)F) * h->C(i);
        TRYC(h->pool.alloc(&h->x[i], n));
        TRYC(h->pool.alloc(&h->xs[i], n));
        TRYC(h->pool.alloc(&h->r[i], n));
        TRYC(h->pool.alloc(&h->xt[i], n));
        TRYC(h->pool.alloc(&h->si[i], n));
        float* sa = nullptr;
        TRYC(h->pool.alloc(&sa, (N_RB * n + 1) / 2));
        h->sa[i] = reinterpret_cast<bf16_t*>(sa);
    }
    for (int i = 0; i < N_UP; ++i)
        if (h->sb_ld[i]) TRYC(h->pool.alloc(&h->sb[i], B * ((size_t)h->L(i, (int)F) + 1) * h->sb_ld[i]));
    TRYC(h->pool.alloc(&h->lens, 8 * B));
#undef TRYC
    if (hipStreamSynchronize(st) != hipSuccess) { fy_set_error("fy_hift_create: stream sync failed"); return fail(FY_ERR_HIP); }
    *out = h;
    return FY_OK;
}

extern "C" void fy_hift_destroy(fy_hift* h) { delete h; }

// ---- small kernels --------------------------------------------------------------
// f0 = |Linear(f0_ch -> 1)(h)|, f0_predictor.py:102-103.  One wave per (b, frame).
__global__ void f0_classifier_k(const float* __restrict__ hin, const float* __restrict__ w, const float* __restrict__ bias,
                                float* __restrict__ f0, const int* __restrict__ frames, int Fmax, int ch) {
    int b = blockIdx.y, t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= frames[b]) return;
    const float* row = hin + ((long)b * Fmax + t) * ch;
    float s = 0.f;
    for (int c = lane; c < ch; c += 64) s = fmaf(row[c], w[c], s);
    s = wave_sum(s);
    if (lane == 0) f0[(long)b * Fmax + t] = fabsf(s + bias[0]);
}

// SineGen2._f02sine per frame (generator.py:233-258, eval + causal): one thread per (b, harmonic) walks the
// frames in order.  rad = (f0*(h+1)/sr) % 1; the x(1/480) linear down-sampling of the x480 nearest-repeated
// signal returns the frame's own value (source index 480 i + 239.5, both neighbours inside frame i, so
// rand_ini - added to sample 0 only - never enters); torch.cumsum on CPU accumulates fp32 data in double
// and rounds each prefix to fp32; phase = ((cum*2)*pi)*480 in fp32; value = sin(phase) * sine_amp.
__global__ void sine_phase_k(const float* __restrict__ f0, float* __restrict__ sines, const int* __restrict__ frames,
                             int Fmax, int H1, float sr, float amp, float up) {
    int b = blockIdx.x, hm = threadIdx.x;
    if (hm >= H1) return;
    int n = frames[b];
    double cum = 0.0;
    const float mult = (float)(hm + 1);
    for (int t = 0; t < n; ++t) {
        float fn = f0[(long)b * Fmax + t] * mult;
        float q = fn / sr;
        float rad = q - floorf(q);                  // torch '%' on floats: result takes the divisor's sign; q >= 0 here
        float ds = 0.5f * rad + 0.5f * rad;
        cum += (double)ds;
        float ph = (float)cum;
        ph = ph * 2.0f;
        ph = ph * 3.14159265358979323846f;
        ph = ph * up;
        sines[((long)b * Fmax + t) * H1 + hm] = sinf(ph) * amp;
    }
}

// SineGen2.forward :303-316 + SourceModuleHnNSF.forward :367-368, per sample.
__global__ void source_k(const float* __restrict__ f0, const float* __restrict__ sines, const float* __restrict__ noise,
                         const float* __restrict__ lw, const float* __restrict__ lb, float* __restrict__ src,
                         const int* __restrict__ frames, int Fmax, int H1, int up, float thr, float sigma, float amp) {
    int b = blockIdx.y;
    long n = blockIdx.x * 256L + threadIdx.x;
    long S = (long)frames[b] * up;
    if (n >= S) return;
    int t = (int)(n / up);
    float f = f0[(long)b * Fmax + t];
    float uv = f > thr ? 1.f : 0.f;
    float namp = uv * sigma + ((1.f - uv) * amp) / 3.f;
    float acc = lb[0];
    for (int hm = 0; hm < H1; ++hm) {
        float sw = sines[((long)b * Fmax + t) * H1 + hm] * uv + namp * noise[n * H1 + hm];
        acc = fmaf(sw, lw[hm], acc);
    }
    src[(long)b * Fmax * up + n] = tanhf(acc);
}

__constant__ float c_cos16[16], c_sin16[16], c_hann16[16];

struct SuperRows {                                   // bf16 super-row copies of s_stft (null = not wanted)
    bf16_t* p[N_UP]; int s[N_UP]; int ld[N_UP]; long bs[N_UP];
};

// torch.stft(n_fft 16, hop 4, periodic hann, center + reflect), generator.py:491-497 -> (B, T, 18) = [re(9), im(9)];
// and the same rows, rounded to bf16, in the super-row layouts of the source down-samplers (super_pack_k): row tt lands in
// S'[ceil(tt/s)] at position tt - s*ceil(tt/s) + s - 1; the thread that writes a super-row's last position also zeroes its
// padding channels, and row 0's thread the s-1 positions before it (the conv's left zero padding).
__global__ void stft16_k(const float* __restrict__ src, float* __restrict__ out, const int* __restrict__ frames_src,
                         const int* __restrict__ frames, int Fmax, int up, int spf, SuperRows sr) {
    int b = blockIdx.y;
    int tt = blockIdx.x * 256 + threadIdx.x;
    int S = frames_src[b] * up;               // samples of source (the reflect bound); a streaming chunk keeps fewer STFT frames
    int T = frames[b] * spf + 1;
    if (tt >= T) return;
    const float* s = src + (long)b * Fmax * up;
    float fr[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        int i = tt * 4 + n - 8;
        if (i < 0) i = -i;
        if (i >= S) i = 2 * (S - 1) - i;
        fr[n] = s[i] * c_hann16[n];
    }
    float* o = out + ((long)b * (Fmax * spf + 1) + tt) * 18;
    float val[18];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        float re = 0.f, im = 0.f;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            int m = (k * n) & 15;
            re = fmaf(fr[n], c_cos16[m], re);
            im = fmaf(fr[n], -c_sin16[m], im);
        }
        o[k] = re;
        o[9 + k] = im;
        val[k] = re; val[9 + k] = im;
    }
    uint32_t pk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) pk[k] = (uint32_t)f32_to_bf16(val[2 * k]) | ((uint32_t)f32_to_bf16(val[2 * k + 1]) << 16);
#pragma unroll
    for (int i = 0; i < N_UP; ++i) {
        if (!sr.p[i]) continue;
        const int sd = sr.s[i], j = (tt + sd - 1) / sd, pos = tt - sd * j + sd - 1;
        bf16_t* row = sr.p[i] + (long)b * sr.bs[i] + (long)j * sr.ld[i];
        uint32_t* dst = reinterpret_cast<uint32_t*>(row + pos * 18);
#pragma unroll
        for (int k = 0; k < 9; ++k) dst[k] = pk[k];
        if (pos == sd - 1) {
            uint32_t* z = reinterpret_cast<uint32_t*>(row + sd * 18);
            for (int k = 0; k < (sr.ld[i] - sd * 18) / 2; ++k) z[k] = 0u;
        }
        if (tt == 0) {
            uint32_t* z = reinterpret_cast<uint32_t*>(row);
            for (int k = 0; k < (sd - 1) * 9; ++k) z[k] = 0u;
        }
    }
}

// generator.py:704-705 + _istft :499-502: conv_post output (B, T, 18) -> complex spectrum (in place layout)
__global__ void spec_k(const float* __restrict__ post, float* __restrict__ spec, const int* __restrict__ frames, int Fmax, int spf) {
    int b = blockIdx.y;
    long i = blockIdx.x * 256L + threadIdx.x;
    long T = (long)frames[b] * spf + 1;
    if (i >= T * 9) return;
    long t = i / 9;
    int k = (int)(i % 9);
    const float* p = post + ((long)b * (Fmax * spf + 1) + t) * POST_LD;
    float mag = fminf(expf(p[k]), 100.f);
    float ph = sinf(p[9 + k]);
    float* o = spec + ((long)b * (Fmax * spf + 1) + t) * 18;
    o[k] = mag * cosf(ph);
    o[9 + k] = mag * sinf(ph);
}

// torch.istft(center=True) + clamp, generator.py:503-505, 710: one thread per output sample.
__global__ void istft16_k(const float* __restrict__ spec, float* __restrict__ wav, const int* __restrict__ frames, const int* __restrict__ s_out,
                          int Fmax, int up, int spf, float limit) {
    int b = blockIdx.y;
    long n = blockIdx.x * 256L + threadIdx.x;
    if (n >= s_out[b]) return;                // a streaming chunk holds its last frame of samples back
    int T = frames[b] * spf + 1;
    int m = (int)n + 8;                       // index in the un-trimmed overlap-add buffer
    int t_hi = min(m / 4, T - 1);
    int t_lo = max((m - 15 + 3) / 4, 0);
    float y = 0.f, env = 0.f;
    for (int t = t_lo; t <= t_hi; ++t) {
        int j = m - 4 * t;                    // 0..15
        const float* sp = spec + ((long)b * (Fmax * spf + 1) + t) * 18;
        float v = sp[0] + ((j & 1) ? -sp[8] : sp[8]);      // DC + Nyquist (imaginary parts drop out)
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            int q = (k * j) & 15;
            v += 2.f * (sp[k] * c_cos16[q] - sp[9 + k] * c_sin16[q]);
        }
        float w = c_hann16[j];
        y += (v * (1.f / 16.f)) * w;
        env += w * w;
    }
    y = y / env;
    wav[(long)b * Fmax * up + n] = fminf(fmaxf(y, -limit), limit);
}

// __constant__ tables are per device: filled once per device (idempotent, so two handles' first calls may both do it)
static PerDeviceOnce g_tables_once;
static int init_tables() {
    const int dslot = current_device_slot();
    if (g_tables_once.done[dslot].load(std::memory_order_acquire)) return FY_OK;
    float c[16], s[16], w[16];
    for (int i = 0; i < 16; ++i) {
        c[i] = (float)cos(2.0 * M_PI * i / 16.0);
        s[i] = (float)sin(2.0 * M_PI * i / 16.0);
        w[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / 16.0));
    }
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_cos16), c, sizeof(c)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_sin16), s, sizeof(s)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_hann16), w, sizeof(w)));
    g_tables_once.done[dslot].store(true, std::memory_order_release);
    return FY_OK;
}

// ---- orchestration -----------------------------------------------------------------
// Per-utterance row counts of every stage.  finalize (the whole utterance): everything follows F.  A streaming chunk
// (finalize = false, generator.py:674-679, 708-709, 722-725; f0_predictor.py:98-99) spends its last frames as look-ahead
// instead of zero padding: the f0 predictor's first conv takes 3 (F_f0 = F - 3 frames of f0 / source), conv_pre another 4
// of those (F_dec = F - 7 decoder frames) and the last frame of samples is held back (S_out = 480 (F - 8)).
static int set_lens(fy_hift* h, const int32_t* frames, int B, int Fmax, hipStream_t st, bool finalize = true) {
    FY_CHECK(h && frames && B >= 1 && B <= h->max_batch && Fmax >= 1 && Fmax <= h->max_frames, FY_ERR_ARG,
             "hift: batch %d / frames %d outside the handle's limits (%d, %d)", B, Fmax, h ? h->max_batch : 0, h ? h->max_frames : 0);
    const int look_f0 = h->w.f0c[0].KW - 1, look_pre = h->cfg.pre_look_right;
    std::vector<int> v(8 * h->max_batch, 0);
    for (int b = 0; b < B; ++b) {
        int F = frames[b];
        FY_CHECK(F >= 1 && F <= Fmax, FY_ERR_ARG, "hift: frames[%d] = %d outside [1, %d]", b, F, Fmax);
        FY_CHECK(finalize || F >= look_f0 + look_pre + 2, FY_ERR_ARG, "hift: a streaming chunk needs at least %d frames (got %d)",
                 look_f0 + look_pre + 2, F);
        const int Ff0 = finalize ? F : F - look_f0, Fd = finalize ? F : Ff0 - look_pre;
        int mb = h->max_batch;
        v[0 * mb + b] = F;
        v[1 * mb + b] = Fd * h->cfg.ups[0];
        v[2 * mb + b] = Fd * h->cfg.ups[0] * h->cfg.ups[1];
        v[3 * mb + b] = Fd * h->stft_per_frame;          // stage-2 conv rows before the reflect pad
        v[4 * mb + b] = Fd * h->stft_per_frame + 1;
        v[5 * mb + b] = (finalize ? Fd : Fd - 1) * h->up_total;
        v[6 * mb + b] = Ff0;
        v[7 * mb + b] = Fd;
    }
    FY_TRY(upload_ints(h->lens, v.data(), (int)v.size(), st));      // by kernel argument: no stream synchronisation (runtime.h)
    h->B = B; h->Fmax = Fmax;
    return init_tables();
}

static ConvDesc base_desc(int B) {
    ConvDesc d;
    memset(&d, 0, sizeof(d));
    d.B = B; d.dil = 1; d.stride = 1; d.up = 1; d.groups = 1; d.out_scale = 1.f;
    return d;
}

static int run_conv(const ConvDesc& d, const ConvW& w, uint32_t flags, hipStream_t st) {
    if (!(flags & FY_DIRECT) && w.w_mfma) return conv1d_bf16_mfma(d, w, (flags & FY_PRECISE) != 0, st);
    return conv1d_f32_direct(d, w, st);
}

static int hift_f0(fy_hift* h, int B, int Fmax, hipStream_t st) {
    const fy_hift_config& c = h->cfg;
    const int* lF = h->lens;                               // mel frames
    const int* lF0 = h->lens + 6 * h->max_batch;           // f0 frames (= mel frames unless a streaming chunk)
    // condnet[0]: k4, 3 frames of right look-ahead; then 4x k3 left-causal; ELU after each (f0_predictor.py:74-90)
    const float* in = h->mel_cl;
    int in_ld = c.mel;
    float* bufs[2] = {h->f0a, h->f0b};
    for (int i = 0; i < 5; ++i) {
        ConvDesc d = base_desc(B);
        d.x = in; d.x_bs = (long)Fmax * in_ld; d.x_ld = in_ld; d.L_in = Fmax; d.in_len = i == 0 ? lF : lF0;
        d.y = bufs[i & 1]; d.y_bs = (long)Fmax * c.f0_ch; d.y_ld = c.f0_ch; d.L_out = Fmax; d.out_len = lF0;
        d.Cin = i == 0 ? c.mel : c.f0_ch; d.Cout = c.f0_ch; d.KW = h->w.f0c[i].KW;
        d.pad_left = i == 0 ? 0 : (d.KW - 1);
        d.bias = h->w.f0c[i].bias; d.post_act = ACT_ELU;
        FY_TRY(conv1d_f32_mfma(d, h->w.f0c[i], st));
        in = bufs[i & 1]; in_ld = c.f0_ch;
    }
    hipLaunchKernelGGL(f0_classifier_k, dim3(cdiv(Fmax, 4), B), dim3(256), 0, st, in, h->cls_w, h->cls_b, h->f0, lF0, Fmax, c.f0_ch);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

static int hift_source(fy_hift* h, const float* f0, int B, int Fmax, const float* rand_ini, const float* sine_noise, hipStream_t st) {
    const fy_hift_config& c = h->cfg;
    (void)rand_ini;   // provably without effect on the output (see sine_phase_k); kept in the ABI for fidelity
    const int H1 = c.harmonics + 1;
    hipLaunchKernelGGL(sine_phase_k, dim3(B), dim3(64), 0, st, f0, h->rad_phase, h->lens + 6 * h->max_batch, Fmax, H1, (float)c.sampling_rate,
                       c.nsf_alpha, (float)h->up_total);
    hipLaunchKernelGGL(source_k, dim3(cdiv(Fmax * h->up_total, 256), B), dim3(256), 0, st, f0, h->rad_phase, sine_noise, h->lin_w,
                       h->lin_b, h->source, h->lens + 6 * h->max_batch, Fmax, H1, h->up_total, c.voiced_thr, c.nsf_sigma, c.nsf_alpha);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

// one ResBlock (generator.py:110-117): x_in -> out.  `work` holds the running x between iterations, `xt` the
// inner activation.  final_dst/final_scale/final_acc: where the last iteration's x goes ((conv2 + x)*scale).
// On the default (bf16 MFMA) path the operand of every conv but the first travels as a bf16 stream that its
// producer's epilogue already activated (snake with the consumer's alpha) and rounded -- the very values the
// consumer's LDS staging would have computed from the fp32 tensor -- so `xt` holds two bf16 streams instead of one
// fp32 tensor: snake2_j(conv1_j(.)) for conv2_j, and snake1_{j+1}(x) for conv1_{j+1}.
// x_act0: x_in under snake(.; rb.a1[0]) as a bf16 stream when x_in's producer wrote one (the default path: iteration 0 then stages
// a copy like every other iteration instead of activating fp32 rows, halo included); null: iteration 0 activates x_in itself.
static int run_resblock(fy_hift* h, const HiftConvs::RB& rb, const float* x_in, float* work, float* xt, float* final_dst,
                        float final_scale, int final_acc, int B, int Lmax, const int* len, int C, uint32_t flags, hipStream_t st,
                        const bf16_t* x_act0 = nullptr) {
    const fy_hift_config& c = h->cfg;
    const long bs = (long)Lmax * C;
    const bool streams = !(flags & (FY_DIRECT | FY_PRECISE)) && rb.c1[0].w_mfma && C % 32 == 0;
    bf16_t* a2 = reinterpret_cast<bf16_t*>(xt);
    bf16_t* a1 = a2 + bs * B;
    static const bool fuse_on = !(getenv("FY_HIFT_FUSE") && atoi(getenv("FY_HIFT_FUSE")) == 0);
    bool fuse = streams && fuse_on && !(final_acc && x_in == work);
    for (int j = 0; j < N_DIL && fuse; ++j) fuse = conv_resblock_iter_supported(C, rb.c1[j].KW, c.rb_d[j]) && rb.c2[j].KW == rb.c1[j].KW;
    if (fuse) {
        // one launch per iteration (conv.h: ResIterDesc).  A workgroup reads its conv1 input WITH a halo and writes only its own
        // rows, so the input of an iteration is never the buffer it writes: the bf16 streams alternate between the two slots of
        // `xt`; the fp32 input of iteration 0 (read with a halo) goes to `work`, or to final_dst when x_in IS work.
        const float* cur = x_in;
        const bf16_t* sin = x_act0;
        for (int j = 0; j < N_DIL; ++j) {
            const bool last = j == N_DIL - 1;
            ResIterDesc r{};
            r.x = sin ? nullptr : x_in; r.x_act = sin; r.resid = cur; r.bs = bs; r.ld = C;
            r.alpha1 = rb.a1[j]; r.alpha2 = rb.a2[j]; r.bias1 = rb.c1[j].bias; r.bias2 = rb.c2[j].bias;
            r.KW = rb.c1[j].KW; r.dil = c.rb_d[j];
            r.B = B; r.L = Lmax; r.C = C; r.len = len;
            bf16_t* sout = (sin == a1) ? a2 : a1;
            if (last) { r.y = final_dst; r.out_scale = final_scale; r.accumulate = final_acc; }
            else {
                r.y = (j == 0 && x_in == work) ? final_dst : work;
                r.y_act = sout; r.alpha_out = rb.a1[j + 1]; r.out_scale = 1.f; r.accumulate = 0;
            }
            FY_TRY(conv_resblock_iter(r, rb.c1[j], rb.c2[j], st));
            cur = r.y; sin = sout;
        }
        return FY_OK;
    }
    for (int j = 0; j < N_DIL; ++j) {
        const float* cur = j == 0 ? x_in : work;
        ConvDesc d = base_desc(B);
        d.x = cur; d.x_bs = bs; d.x_ld = C; d.L_in = Lmax; d.in_len = len;
        d.y = xt; d.y_bs = bs; d.y_ld = C; d.L_out = Lmax; d.out_len = len;
        d.Cin = C; d.Cout = C; d.KW = rb.c1[j].KW; d.dil = c.rb_d[j]; d.pad_left = (d.KW - 1) * d.dil;
        d.pre_act = ACT_SNAKE; d.alpha = rb.a1[j]; d.bias = rb.c1[j].bias;
        if (streams) {
            if (j > 0) d.x_act = a1;
            else if (x_act0) d.x_act = x_act0;
            d.y = nullptr; d.y_act = a2; d.alpha_out = rb.a2[j];
        }
        FY_TRY(run_conv(d, rb.c1[j], flags, st));
        ConvDesc e = base_desc(B);
        e.x = xt; e.x_bs = bs; e.x_ld = C; e.L_in = Lmax; e.in_len = len;
        e.L_out = Lmax; e.out_len = len;
        e.Cin = C; e.Cout = C; e.KW = rb.c2[j].KW; e.dil = 1; e.pad_left = e.KW - 1;
        e.pre_act = ACT_SNAKE; e.alpha = rb.a2[j]; e.bias = rb.c2[j].bias;
        e.add_resid = 1; e.resid = cur; e.r_bs = bs; e.r_ld = C;
        if (streams) e.x_act = a2;
        if (j == N_DIL - 1) {
            e.y = final_dst; e.out_scale = final_scale; e.accumulate = final_acc;
        } else {
            e.y = work;
            if (streams) { e.y_act = a1; e.alpha_out = rb.a1[j + 1]; }
        }
        e.y_bs = bs; e.y_ld = C;
        FY_TRY(run_conv(e, rb.c2[j], flags, st));
    }
    return FY_OK;
}

// ReflectionPad1d((1,0)) after the polyphase up-conv wrote rows 1..: row 0 = conv row 1 + si[0] = y[2] - si[2] + si[0]
// ya[k] (optional): row 0 of the up-conv's bf16 streams, snake(row 0; al[k]) as the conv's epilogue writes the other rows
struct Row0Streams { bf16_t* ya[N_RB]; const float* al[N_RB]; };
__global__ void reflect_row0_k(float* __restrict__ y, const float* __restrict__ si, long bs, int C, Row0Streams rs) {
    int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float* yb = y + (long)b * bs;
    const float* sb = si + (long)b * bs;
    const float v = yb[2 * C + c] - sb[2 * C + c] + sb[c];
    yb[c] = v;
#pragma unroll
    for (int k = 0; k < N_RB; ++k)
        if (rs.ya[k]) rs.ya[k][(long)b * bs + c] = f32_to_bf16(snake_fast(v, rs.al[k][c]));
}

static int hift_decode(fy_hift* h, int B, int Fmax, float* wav, uint32_t flags, hipStream_t st) {
    const fy_hift_config& c = h->cfg;
    const int mb = h->max_batch;
    const int spf = h->stft_per_frame, up = h->up_total;
    const int Tmax = Fmax * spf + 1;
    const bool streams = !(flags & (FY_DIRECT | FY_PRECISE));
    // FY_HIFT_ACT0=0: the first iteration of every ResBlock activates its fp32 input itself (rounds 1-4), for A/B runs
    static const bool act0_on = !(getenv("FY_HIFT_ACT0") && atoi(getenv("FY_HIFT_ACT0")) == 0);
    const bool act0 = streams && act0_on;
    SuperRows sr;
    for (int i = 0; i < N_UP; ++i) {
        int k, s;
        source_down_shape(c, i, &k, &s);
        sr.p[i] = streams ? h->sb[i] : nullptr; sr.s[i] = s; sr.ld[i] = h->sb_ld[i];
        sr.bs[i] = ((long)h->L(i, Fmax) + 1) * h->sb_ld[i];
    }
    hipLaunchKernelGGL(stft16_k, dim3(cdiv(Tmax, 256), B), dim3(256), 0, st, h->source, h->s_stft, h->lens + 6 * mb, h->lens + 7 * mb, Fmax, up, spf, sr);
    {   // conv_pre: k5, 4 frames of right look-ahead (generator.py:621-623, 675)
        ConvDesc d = base_desc(B);
        d.x = h->mel_cl; d.x_bs = (long)Fmax * c.mel; d.x_ld = c.mel; d.L_in = Fmax; d.in_len = h->lens + 6 * mb;
        d.y = h->x_pre; d.y_bs = (long)Fmax * c.base; d.y_ld = c.base; d.L_out = Fmax; d.out_len = h->lens + 7 * mb;
        d.Cin = c.mel; d.Cout = c.base; d.KW = c.pre_look_right + 1; d.pad_left = 0; d.bias = h->w.conv_pre.bias;
        FY_TRY(run_conv(d, h->w.conv_pre, flags, st));
    }
    const float* xin = h->x_pre;
    int Cin = c.base, Lin_max = Fmax;
    const int* len_in = h->lens + 7 * mb;
    for (int i = 0; i < N_UP; ++i) {
        const int C = h->C(i);
        const int Lmax = h->L(i, Fmax);
        const int* len = h->lens + (i == N_UP - 1 ? 4 : i + 1) * mb;          // rows of this stage's tensors
        const int* len_conv = h->lens + (i + 1) * mb;                           // rows the up-conv itself produces
        const long bs = (long)Lmax * C;
        {   // source branch: si = source_resblocks[i](source_downs[i](s_stft)), generator.py:690-691
            int k, s;
            source_down_shape(c, i, &k, &s);
            const bf16_t* src_act = nullptr;
            ConvDesc d = base_desc(B);
            d.x = h->s_stft; d.x_bs = (long)Tmax * 18; d.x_ld = 18; d.L_in = Tmax; d.in_len = h->lens + 4 * mb;
            d.y = h->r[i]; d.y_bs = bs; d.y_ld = C; d.L_out = Lmax; d.out_len = len;
            d.Cin = 18; d.Cout = C; d.KW = k; d.stride = s; d.pad_left = s == 1 ? 0 : s - 1; d.bias = h->w.source_downs[i].bias;
            if (sr.p[i]) {       // stride-1 form over the bf16 super-rows (super_pack_k)
                d.x_act = sr.p[i]; d.x_bs = sr.bs[i]; d.x_ld = sr.ld[i]; d.L_in = Lmax + 1; d.in_len = nullptr;
                d.Cin = sr.ld[i]; d.KW = s == 1 ? 1 : 2; d.stride = 1; d.pad_left = 0;
                if (act0 && C % 32 == 0) { d.y_act = h->sa[i]; d.alpha_out = h->w.src_rb[i].a1[0]; src_act = h->sa[i]; }
                FY_TRY(conv1d_bf16_mfma(d, h->sd_mfma[i], false, st));
            } else {
                FY_TRY(conv1d_f32_direct(d, h->w.source_downs[i], st));
            }
            FY_TRY(run_resblock(h, h->w.src_rb[i], h->r[i], h->r[i], h->xt[i], h->si[i], 1.f, 0, B, Lmax, len, C, flags, st, src_act));
        }
        const bf16_t* x_act[N_RB] = {nullptr, nullptr, nullptr};
        {   // x = ups[i](leaky_relu(x)) [reflect pad on the last stage] + si, generator.py:683-692
            ConvDesc d = base_desc(B);
            d.x = xin; d.x_bs = (long)Lin_max * Cin; d.x_ld = Cin; d.L_in = Lin_max; d.in_len = len_in;
            d.y = h->x[i]; d.y_bs = bs; d.y_ld = C; d.L_out = Lin_max * c.ups[i]; d.out_len = len_conv;
            d.Cin = Cin; d.Cout = C; d.KW = c.up_k[i]; d.up = c.ups[i]; d.pad_left = c.up_k[i] - 1;
            d.pre_act = ACT_LEAKY; d.pre_slope = c.lrelu; d.bias = h->w.ups[i].bias;
            d.add_resid = 1; d.resid = h->si[i]; d.r_bs = bs; d.r_ld = C;
            d.reflect1 = i == N_UP - 1;
            if (streams && h->ups_poly[i].w_mfma) {
                // polyphase form: rows [n][phi C + co] of a stride-1 conv over the un-repeated input are rows u n + phi
                const int u = c.ups[i], off = d.reflect1 ? C : 0;
                d.up = 1; d.KW = h->ups_poly[i].KW; d.pad_left = d.KW - 1; d.Cout = C * u; d.bias = h->ups_poly[i].bias;
                d.L_out = Lin_max; d.out_len = len_in;
                d.y = h->x[i] + off; d.y_ld = C * u; d.resid = h->si[i] + off; d.r_ld = C * u; d.reflect1 = 0;
                Row0Streams rs{};
                if (act0 && C % 32 == 0) {
                    // the stage's three ResBlocks all start from this output: one stream each, under their own first snake
                    const size_t n = (size_t)B * bs;
                    for (int j = 0; j < N_RB; ++j) { x_act[j] = h->sa[i] + j * n; rs.ya[j] = h->sa[i] + j * n; rs.al[j] = h->w.rb[i * N_RB + j].a1[0]; }
                    d.y_act = h->sa[i] + off; d.alpha_out = rs.al[0];
                    d.y_act2 = h->sa[i] + n + off; d.alpha_out2 = rs.al[1];
                    d.y_act3 = h->sa[i] + 2 * n + off; d.alpha_out3 = rs.al[2];
                    d.alpha_mod = C;
                }
                FY_TRY(conv1d_bf16_mfma(d, h->ups_poly[i], false, st));
                if (off) {
                    hipLaunchKernelGGL(reflect_row0_k, dim3(cdiv(C, 256), B), dim3(256), 0, st, h->x[i], h->si[i], bs, C, rs);
                    HIP_TRY(hipGetLastError());
                }
            } else {
                FY_TRY(run_conv(d, h->w.ups[i], flags, st));
            }
        }
        for (int j = 0; j < N_RB; ++j)   // x = mean_j resblocks[3i+j](x), generator.py:694-700
            FY_TRY(run_resblock(h, h->w.rb[i * N_RB + j], h->x[i], h->r[i], h->xt[i], h->xs[i], 1.f / N_RB, j > 0, B, Lmax, len, C, flags, st, x_act[j]));
        xin = h->xs[i]; Cin = C; Lin_max = Lmax; len_in = len;
    }
    {   // conv_post on leaky_relu(x, 0.01), generator.py:702-703
        ConvDesc d = base_desc(B);
        d.x = xin; d.x_bs = (long)Lin_max * Cin; d.x_ld = Cin; d.L_in = Lin_max; d.in_len = len_in;
        d.y = h->post; d.y_bs = (long)Tmax * POST_LD; d.y_ld = POST_LD; d.L_out = Tmax; d.out_len = len_in;
        d.Cin = Cin; d.Cout = 18; d.KW = 7; d.pad_left = 6; d.pre_act = ACT_LEAKY; d.pre_slope = 0.01f; d.bias = h->w.conv_post.bias;
        FY_TRY(run_conv(d, h->w.conv_post, flags, st));
    }
    // the spectrum overwrites s_stft (no longer needed)
    hipLaunchKernelGGL(spec_k, dim3(cdiv(Tmax * 9, 256), B), dim3(256), 0, st, h->post, h->s_stft, h->lens + 7 * mb, Fmax, spf);
    hipLaunchKernelGGL(istft16_k, dim3(cdiv(Fmax * up, 256), B), dim3(256), 0, st, h->s_stft, wav, h->lens + 7 * mb, h->lens + 5 * mb, Fmax, up, spf, c.audio_limit);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}

static int load_mel(fy_hift* h, const float* mel, int B, int Fmax, hipStream_t st) {
    return transpose_bcl_to_blc(mel, h->mel_cl, B, h->cfg.mel, Fmax, (long)h->cfg.mel * Fmax, (long)Fmax * h->cfg.mel, h->cfg.mel, st);
}

extern "C" int fy_hift_infer(fy_hift* h, const float* mel, const int32_t* frames, int32_t B, int32_t Fmax, const float* rand_ini,
                             const float* sine_noise, float* wav, float* source, uint32_t flags, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    FY_CHECK(h && mel && sine_noise && wav, FY_ERR_ARG, "fy_hift_infer: null argument");
    FY_TRY(set_lens(h, frames, B, Fmax, st, !(flags & FY_NO_FINALIZE)));
    FY_TRY(load_mel(h, mel, B, Fmax, st));
    FY_TRY(hift_f0(h, B, Fmax, st));
    FY_TRY(hift_source(h, h->f0, B, Fmax, rand_ini, sine_noise, st));
    if (source) HIP_TRY(hipMemcpyAsync(source, h->source, (size_t)B * Fmax * h->up_total * sizeof(float), hipMemcpyDeviceToDevice, st));
    return hift_decode(h, B, Fmax, wav, flags, st);
}

extern "C" int fy_hift_f0(fy_hift* h, const float* mel, const int32_t* frames, int32_t B, int32_t Fmax, float* f0, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    FY_CHECK(h && mel && f0, FY_ERR_ARG, "fy_hift_f0: null argument");
    FY_TRY(set_lens(h, frames, B, Fmax, st));
    FY_TRY(load_mel(h, mel, B, Fmax, st));
    FY_TRY(hift_f0(h, B, Fmax, st));
    HIP_TRY(hipMemcpyAsync(f0, h->f0, (size_t)B * Fmax * sizeof(float), hipMemcpyDeviceToDevice, st));
    return FY_OK;
}

extern "C" int fy_hift_source(fy_hift* h, const float* f0, const int32_t* frames, int32_t B, int32_t Fmax, const float* rand_ini,
                              const float* sine_noise, float* source, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    FY_CHECK(h && f0 && sine_noise && source, FY_ERR_ARG, "fy_hift_source: null argument");
    FY_TRY(set_lens(h, frames, B, Fmax, st));
    HIP_TRY(hipMemcpyAsync(h->f0, f0, (size_t)B * Fmax * sizeof(float), hipMemcpyDeviceToDevice, st));
    FY_TRY(hift_source(h, h->f0, B, Fmax, rand_ini, sine_noise, st));
    HIP_TRY(hipMemcpyAsync(source, h->source, (size_t)B * Fmax * h->up_total * sizeof(float), hipMemcpyDeviceToDevice, st));
    return FY_OK;
}

extern "C" int fy_hift_decode(fy_hift* h, const float* mel, const float* source, const int32_t* frames, int32_t B, int32_t Fmax,
                              float* wav, uint32_t flags, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    FY_CHECK(h && mel && source && wav, FY_ERR_ARG, "fy_hift_decode: null argument");
    FY_TRY(set_lens(h, frames, B, Fmax, st));
    FY_TRY(load_mel(h, mel, B, Fmax, st));
    HIP_TRY(hipMemcpyAsync(h->source, source, (size_t)B * Fmax * h->up_total * sizeof(float), hipMemcpyDeviceToDevice, st));
    return hift_decode(h, B, Fmax, wav, flags, st);
}

extern "C" int fy_hift_tap(fy_hift* h, const char* name, float* dst, int64_t* rows, int64_t* cols, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    FY_CHECK(h && name && dst && rows && cols && h->B > 0, FY_ERR_ARG, "fy_hift_tap: null argument or no call made yet");
    const int F = h->Fmax, spf = h->stft_per_frame;
    const float* src = nullptr;
    std::string n = name;
    if (n == "f0") { src = h->f0; *rows = F; *cols = 1; }
    else if (n == "source") { src = h->source; *rows = (int64_t)F * h->up_total; *cols = 1; }
    else if (n == "conv_pre") { src = h->x_pre; *rows = F; *cols = h->cfg.base; }
    else if (n == "conv_post") {
        *rows = F * spf + 1; *cols = 18;
        HIP_TRY(hipMemcpy2DAsync(dst, 18 * sizeof(float), h->post, POST_LD * sizeof(float), 18 * sizeof(float), (size_t)h->B * (*rows),
                                 hipMemcpyDeviceToDevice, st));
        return FY_OK;
    }
    else if (n.size() == 5 && n.substr(0, 4) == "fuse" && n[4] >= '0' && n[4] < '0' + N_UP) {
        int i = n[4] - '0'; src = h->x[i]; *rows = h->L(i, F); *cols = h->C(i);
    } else if (n.size() == 6 && n.substr(0, 5) == "stage" && n[5] >= '0' && n[5] < '0' + N_UP) {
        int i = n[5] - '0'; src = h->xs[i]; *rows = h->L(i, F); *cols = h->C(i);
    } else {
        fy_set_error("fy_hift_tap: unknown tap '%s'", name);
        return FY_ERR_ARG;
    }
    HIP_TRY(hipMemcpyAsync(dst, src, (size_t)h->B * (*rows) * (*cols) * sizeof(float), hipMemcpyDeviceToDevice, st));
    return FY_OK;
}

extern "C" int fy_hift_resblock(fy_hift* h, int32_t index, const float* x, float* y, int32_t B, int32_t L, uint32_t flags, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    FY_CHECK(h && x && y && index >= 0 && index < N_UP * N_RB && B >= 1 && L >= 1, FY_ERR_ARG, "fy_hift_resblock: bad argument");
    const int stage = index / N_RB, C = h->C(stage);
    FY_CHECK(B <= h->max_batch && L <= h->L(stage, h->max_frames), FY_ERR_ARG,
             "fy_hift_resblock: (B %d, L %d) exceeds the handle's workspace (%d, %d)", B, L, h->max_batch, h->L(stage, h->max_frames));
    return run_resblock(h, h->w.rb[index], x, h->r[stage], h->xt[stage], y, 1.f, 0, B, L, nullptr, C, flags, st);
}
