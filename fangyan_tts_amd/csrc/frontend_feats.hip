// The two 16 kHz feature front ends of the reference's prompt path (CosyVoice/cosyvoice/cli/frontend.py:94-117), which feed its
// ONNX models:
//   kind 0  whisper.log_mel_spectrogram(speech, n_mels=128) -> speech_tokenizer_v3.onnx   (frontend.py:94-108)
//           hann(400) periodic, torch.stft(n_fft 400, hop 160, center, reflect), power of all frames but the last, Slaney mel
//           filterbank librosa.filters.mel(16000, 400, 128), log10(clamp(., 1e-10)), clamp to (global max - 8), (x + 4) / 4.
//   kind 1  torchaudio.compliance.kaldi.fbank(speech, num_mel_bins=80, dither=0, sample_frequency=16000) -> campplus.onnx
//           (frontend.py:110-117; the mean over frames is subtracted there, here on request): 25 ms frames every 10 ms
//           (snip_edges), DC removal, pre-emphasis 0.97 (first sample against itself), povey window (hann^0.85, symmetric),
//           zero-padded to 512, power spectrum, 80 triangular filters on the kaldi mel scale 1127 ln(1 + f / 700) between 20 Hz
//           and Nyquist, log(max(., FLT_EPSILON)).
// Neither whisper nor torchaudio is in the image and neither is vendored by the reference: both are restated from their
// published algorithms - PARITY UNPINNED (oracle/frontend.py says the same); the STFT halves are held to torch.stft /
// torch.fft by the tests.  A 30 s prompt is 3000 frames x 257 bins x 400 taps: a direct DFT per frame (one workgroup per
// frame, twiddles in LDS) is 0.3 GFLOP - no FFT needed.
#include "runtime.h"
#include <float.h>
#include <math.h>
#include <vector>

#define AF_WIN 400
#define AF_HOP 160
#define AF_MAXBINS 257
#define AF_MAXMELS 128

struct fy_audio_feat {
    int kind = 0, nfft = 0, bins = 0, mels = 0;
    DevPool pool;
    float* window = nullptr;     // [400]
    float2* tw = nullptr;        // [nfft] (cos, sin)(2 pi i / nfft)
    float* fb = nullptr;         // [mels][bins]
    float* scratch = nullptr;    // [2]: global maximum (whisper), spare
};

// one workgroup per frame.  KIND 0: frame f covers samples 160 f - 200 .. + 399 of the reflect-padded signal; output (mels, frames)
// as log10(clamp(mel, 1e-10)).  KIND 1: frame f covers samples 160 f .. + 399; output (frames, mels), finished.
template <int KIND>
__global__ __launch_bounds__(256) void audio_feat_k(const float* __restrict__ wav, long S, const float* __restrict__ window,
                                                    const float2* __restrict__ tw, const float* __restrict__ fb, float* __restrict__ out,
                                                    int frames, int nfft, int bins, int mels) {
    __shared__ float xs[AF_WIN];
    __shared__ float2 ts[512];
    __shared__ float pw[AF_MAXBINS + 3];
    __shared__ float red[4];
    const int f = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < nfft; i += 256) ts[i] = tw[i];
    if (KIND == 0) {
        for (int i = tid; i < AF_WIN; i += 256) {
            long j = (long)f * AF_HOP + i - AF_WIN / 2;          // torch's reflect padding (no edge repeat)
            if (j < 0) j = -j;
            if (j >= S) j = 2 * (S - 1) - j;
            xs[i] = wav[j] * window[i];
        }
        __syncthreads();
    } else {
        // mean removal, pre-emphasis, window: x'[i] = (x[i] - m) - 0.97 (x[max(i - 1, 0)] - m)
        float part = 0.f;
        for (int i = tid; i < AF_WIN; i += 256) {
            const float v = wav[(long)f * AF_HOP + i];
            xs[i] = v;
            part += v;
        }
        part = wave_sum(part);
        if ((tid & 63) == 0) red[tid >> 6] = part;
        __syncthreads();
        const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)AF_WIN;
        float y[2];
        for (int q = 0, i = tid; i < AF_WIN; i += 256, ++q) {
            const float cur = xs[i] - mean, prev = xs[i > 0 ? i - 1 : 0] - mean;
            y[q] = (cur - 0.97f * prev) * window[i];
        }
        __syncthreads();
        for (int q = 0, i = tid; i < AF_WIN; i += 256, ++q) xs[i] = y[q];
        __syncthreads();
    }
    for (int k = tid; k < bins; k += 256) {
        float re = 0.f, im = 0.f;
        int idx = 0;                                             // (k n) mod nfft; samples 400 .. nfft-1 are zero padding
        for (int n = 0; n < AF_WIN; ++n) {
            const float2 c = ts[idx];
            re = fmaf(xs[n], c.x, re);
            im = fmaf(xs[n], -c.y, im);
            idx += k;
            if (idx >= nfft) idx -= nfft;
        }
        pw[k] = re * re + im * im;
    }
    __syncthreads();
    if (tid < mels) {
        const float* w = fb + (long)tid * bins;
        float a = 0.f;
        for (int k = 0; k < bins; ++k) a = fmaf(w[k], pw[k], a);
        if (KIND == 0) out[(long)tid * frames + f] = log10f(fmaxf(a, 1e-10f));
        else out[(long)f * mels + tid] = logf(fmaxf(a, FLT_EPSILON));
    }
}

// maximum of n floats -> res[0] (one workgroup; n is at most 128 x 3000)
__global__ __launch_bounds__(1024) void af_max_k(const float* __restrict__ x, long n, float* __restrict__ res) {
    __shared__ float sm[16];
    float m = -3.0e38f;
    for (long i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, x[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) m = fmaxf(m, sm[i]);
        res[0] = m;
    }
}
// whisper: x = (max(x, global max - 8) + 4) / 4
__global__ void af_whisper_norm_k(float* __restrict__ x, long n, const float* __restrict__ mx) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i < n) x[i] = (fmaxf(x[i], mx[0] - 8.0f) + 4.0f) / 4.0f;
}
// kaldi fbank, subtract the mean over frames of every mel column (frontend.py:115): one thread per column
__global__ void af_colmean_k(float* __restrict__ x, int frames, int mels) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= mels) return;
    double s = 0.0;
    for (int f = 0; f < frames; ++f) s += (double)x[(long)f * mels + c];
    const float m = (float)(s / frames);
    for (int f = 0; f < frames; ++f) x[(long)f * mels + c] -= m;
}

// librosa.filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, norm="slaney") (see frontend.hip: the same restatement)
static void slaney_filterbank(int sr, int nfft, int mels, std::vector<float>& fb) {
    const int bins = nfft / 2 + 1;
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    auto hz_to_mel = [&](double f) { return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp; };
    auto mel_to_hz = [&](double m) { return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m; };
    const double m_lo = hz_to_mel(0.0), m_hi = hz_to_mel(sr / 2.0);
    std::vector<double> mel_f(mels + 2);
    for (int i = 0; i < mels + 2; ++i) mel_f[i] = mel_to_hz(m_lo + (m_hi - m_lo) * i / (mels + 1));
    fb.assign((size_t)mels * bins, 0.f);
    for (int m = 0; m < mels; ++m) {
        const double enorm = 2.0 / (mel_f[m + 2] - mel_f[m]);
        for (int k = 0; k < bins; ++k) {
            const double fk = (sr / 2.0) * k / (bins - 1);
            const double lower = (fk - mel_f[m]) / (mel_f[m + 1] - mel_f[m]), upper = (mel_f[m + 2] - fk) / (mel_f[m + 2] - mel_f[m + 1]);
            fb[(size_t)m * bins + k] = (float)(fmax(0.0, fmin(lower, upper)) * enorm);
        }
    }
}

// torchaudio.compliance.kaldi.get_mel_banks(num_bins, 512, sr, low_freq 20, high_freq 0 -> Nyquist, no VTLN warp): triangles in
// the mel domain over the first nfft/2 bins, the Nyquist bin's column is zero (kaldi.py pads it)
static void kaldi_filterbank(int sr, int nfft, int mels, double low_freq, std::vector<float>& fb) {
    const int bins = nfft / 2 + 1, nb = nfft / 2;
    auto mel = [](double f) { return 1127.0 * log(1.0 + f / 700.0); };
    const double mel_low = mel(low_freq), mel_high = mel(sr / 2.0), delta = (mel_high - mel_low) / (mels + 1), bw = (double)sr / nfft;
    fb.assign((size_t)mels * bins, 0.f);
    for (int m = 0; m < mels; ++m) {
        const double left = mel_low + m * delta, center = left + delta, right = center + delta;
        for (int k = 0; k < nb; ++k) {
            const double mk = mel(bw * k);
            const double up = (mk - left) / (center - left), down = (right - mk) / (right - center);
            fb[(size_t)m * bins + k] = (float)fmax(0.0, fmin(up, down));
        }
    }
}

extern "C" int fy_audio_feat_create(fy_audio_feat** out, int32_t kind, void* stream) {
    FY_CHECK(out && (kind == 0 || kind == 1), FY_ERR_ARG, "fy_audio_feat_create: kind must be 0 (whisper log-mel 128) or 1 (kaldi fbank 80)");
    hipStream_t st = (hipStream_t)stream;
    fy_audio_feat* p = new fy_audio_feat();
    p->kind = kind;
    p->nfft = kind == 0 ? 400 : 512;
    p->bins = p->nfft / 2 + 1;
    p->mels = kind == 0 ? 128 : 80;
    std::vector<float> win(AF_WIN), fb;
    std::vector<float2> tw(p->nfft);
    for (int i = 0; i < AF_WIN; ++i) {
        if (kind == 0) win[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / AF_WIN));                           // torch.hann_window(400): periodic
        else win[i] = (float)pow(0.5 - 0.5 * cos(2.0 * M_PI * i / (AF_WIN - 1)), 0.85);                      // povey: hann_window(400, periodic=False) ** 0.85
    }
    for (int i = 0; i < p->nfft; ++i) tw[i] = make_float2((float)cos(2.0 * M_PI * i / p->nfft), (float)sin(2.0 * M_PI * i / p->nfft));
    if (kind == 0) slaney_filterbank(16000, 400, 128, fb);
    else kaldi_filterbank(16000, 512, 80, 20.0, fb);
    int rc = p->pool.alloc(&p->window, (size_t)AF_WIN);
    if (!rc) rc = p->pool.alloc(&p->tw, (size_t)p->nfft);
    if (!rc) rc = p->pool.alloc(&p->fb, fb.size());
    if (!rc) rc = p->pool.alloc(&p->scratch, (size_t)4);
    if (rc) { delete p; return rc; }
    if (hipMemcpyAsync(p->window, win.data(), win.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(p->tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(p->fb, fb.data(), fb.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        delete p;
        fy_set_error("fy_audio_feat_create: upload failed");
        return FY_ERR_HIP;
    }
    *out = p;
    return FY_OK;
}

extern "C" void fy_audio_feat_destroy(fy_audio_feat* p) { delete p; }

extern "C" int fy_audio_feat_mels(const fy_audio_feat* p) { return p ? p->mels : 0; }

// whisper: stft with center=True gives 1 + n / 160 frames and the last is dropped; kaldi (snip_edges): 1 + (n - 400) / 160
extern "C" int fy_audio_feat_frames(const fy_audio_feat* p, int64_t n_samples) {
    if (!p) return 0;
    if (p->kind == 0) return n_samples <= AF_WIN / 2 ? 0 : (int)(n_samples / AF_HOP);
    return n_samples < AF_WIN ? 0 : (int)(1 + (n_samples - AF_WIN) / AF_HOP);
}

// wav: device fp32 (n_samples) at 16 kHz; out: device fp32, kind 0: (128, frames), kind 1: (frames, 80).
// flags bit 0 (kind 1 only): subtract every column's mean over the frames (what frontend.py:115 does before campplus).
extern "C" int fy_audio_feat_run(fy_audio_feat* p, const float* wav, int64_t n_samples, float* out, int32_t frames, uint32_t flags, void* stream) {
    FY_CHECK(p && wav && out, FY_ERR_ARG, "fy_audio_feat_run: null argument");
    FY_CHECK(frames >= 1 && frames == fy_audio_feat_frames(p, n_samples), FY_ERR_ARG, "fy_audio_feat_run: %ld samples give %d frames, not %d",
             (long)n_samples, fy_audio_feat_frames(p, n_samples), frames);
    hipStream_t st = (hipStream_t)stream;
    const long n = (long)frames * p->mels;
    if (p->kind == 0) {
        hipLaunchKernelGGL(audio_feat_k<0>, dim3(frames), dim3(256), 0, st, wav, (long)n_samples, p->window, p->tw, p->fb, out, frames, p->nfft, p->bins, p->mels);
        hipLaunchKernelGGL(af_max_k, dim3(1), dim3(1024), 0, st, out, n, p->scratch);
        hipLaunchKernelGGL(af_whisper_norm_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, n, p->scratch);
    } else {
        hipLaunchKernelGGL(audio_feat_k<1>, dim3(frames), dim3(256), 0, st, wav, (long)n_samples, p->window, p->tw, p->fb, out, frames, p->nfft, p->bins, p->mels);
        if (flags & 1u) hipLaunchKernelGGL(af_colmean_k, dim3(cdiv(p->mels, 64)), dim3(64), 0, st, out, frames, p->mels);
    }
    HIP_TRY(hipGetLastError());
    return FY_OK;
}
