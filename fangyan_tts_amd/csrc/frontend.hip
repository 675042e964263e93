// Prompt log-mel spectrogram of the frontend: matcha's mel_spectrogram as cosyvoice3.yaml configures it
// (CosyVoice/third_party/Matcha-TTS/matcha/utils/audio.py:45-82 with n_fft = win = 1920, hop 480, 80 mels, fmin 0,
// fmax = sr/2, center=False; examples/dialect/cosyvoice3/conf/cosyvoice3.yaml:140-148; called from
// cli/frontend.py:119-125 on the 24 kHz prompt): reflect-pad (n_fft - hop)/2 = 720 samples each side, periodic Hann
// window, 1920-point DFT, magnitude sqrt(re^2 + im^2 + 1e-9), Slaney-normalised mel filterbank, log(clamp(., 1e-5)).
// A 10 s prompt is 500 frames x 961 bins: 1.8 GFLOP as a direct DFT, so no FFT is needed; one workgroup per frame.
// The mel filterbank is librosa's (librosa.filters.mel, htk=False, norm="slaney"); librosa is absent from the image, so
// the filterbank is restated from its published definition on the host (build_mel_filterbank) - parity unpinned for that
// table; the STFT part is pinned against torch.stft by the tests.
#include "runtime.h"
#include <math.h>
#include <vector>

#define PM_NFFT 1920
#define PM_HOP 480
#define PM_BINS 961
#define PM_MELS 80
#define PM_PAD 720

struct fy_prompt_mel {
    DevPool pool;
    float* window = nullptr;     // [1920] periodic Hann
    float2* tw = nullptr;        // [1920] (cos, sin)(2 pi i / 1920)
    float* fb = nullptr;         // [80][961]
};

__global__ __launch_bounds__(256) void prompt_mel_k(const float* __restrict__ wav, int S, const float* __restrict__ window,
                                                    const float2* __restrict__ tw, const float* __restrict__ fb, float* __restrict__ out) {
    __shared__ float xs[PM_NFFT];
    __shared__ float2 ts[PM_NFFT];
    __shared__ float mag[PM_BINS + 3];
    const int f = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < PM_NFFT; i += 256) {
        int j = f * PM_HOP + i - PM_PAD;                    // torch's reflect padding (no edge repeat)
        if (j < 0) j = -j;
        if (j >= S) j = 2 * (S - 1) - j;
        xs[i] = wav[j] * window[i];
        ts[i] = tw[i];
    }
    __syncthreads();
    for (int k = tid; k < PM_BINS; k += 256) {
        float re = 0.f, im = 0.f;
        int idx = 0;                                         // (k n) mod 1920
        for (int n = 0; n < PM_NFFT; ++n) {
            const float2 c = ts[idx];
            re = fmaf(xs[n], c.x, re);
            im = fmaf(xs[n], -c.y, im);
            idx += k;
            if (idx >= PM_NFFT) idx -= PM_NFFT;
        }
        mag[k] = sqrtf(re * re + im * im + 1e-9f);
    }
    __syncthreads();
    if (tid < PM_MELS) {
        const float* w = fb + (long)tid * PM_BINS;
        float a = 0.f;
        for (int k = 0; k < PM_BINS; ++k) a = fmaf(w[k], mag[k], a);
        out[(long)f * PM_MELS + tid] = logf(fmaxf(a, 1e-5f));
    }
}

// librosa.filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, norm="slaney"), restated from its definition:
// Slaney's auditory-toolbox mel scale (linear below 1 kHz at 200/3 Hz per mel, logarithmic above with 27 steps per factor
// 6.4), triangular filters between consecutive mel-spaced centre frequencies, each scaled by 2 / (its band width in Hz).
static void build_mel_filterbank(int sr, std::vector<float>& fb) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    auto hz_to_mel = [&](double f) { return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp; };
    auto mel_to_hz = [&](double m) { return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m; };
    const double m_lo = hz_to_mel(0.0), m_hi = hz_to_mel(sr / 2.0);
    std::vector<double> mel_f(PM_MELS + 2);
    for (int i = 0; i < PM_MELS + 2; ++i) mel_f[i] = mel_to_hz(m_lo + (m_hi - m_lo) * i / (PM_MELS + 1));
    fb.assign((size_t)PM_MELS * PM_BINS, 0.f);
    for (int m = 0; m < PM_MELS; ++m) {
        const double enorm = 2.0 / (mel_f[m + 2] - mel_f[m]);
        for (int k = 0; k < PM_BINS; ++k) {
            const double fk = (sr / 2.0) * k / (PM_BINS - 1);
            const double lower = (fk - mel_f[m]) / (mel_f[m + 1] - mel_f[m]), upper = (mel_f[m + 2] - fk) / (mel_f[m + 2] - mel_f[m + 1]);
            const double w = fmax(0.0, fmin(lower, upper));
            fb[(size_t)m * PM_BINS + k] = (float)(w * enorm);
        }
    }
}

extern "C" int fy_prompt_mel_create(fy_prompt_mel** out, int32_t sample_rate, void* stream) {
    FY_CHECK(out && sample_rate >= 8000, FY_ERR_ARG, "fy_prompt_mel_create: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    fy_prompt_mel* p = new fy_prompt_mel();
    std::vector<float> win(PM_NFFT), fb;
    std::vector<float2> tw(PM_NFFT);
    for (int i = 0; i < PM_NFFT; ++i) {
        win[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / PM_NFFT));      // torch.hann_window(1920): periodic
        tw[i] = make_float2((float)cos(2.0 * M_PI * i / PM_NFFT), (float)sin(2.0 * M_PI * i / PM_NFFT));
    }
    build_mel_filterbank(sample_rate, fb);
    int rc = p->pool.alloc(&p->window, (size_t)PM_NFFT);
    if (!rc) rc = p->pool.alloc(&p->tw, (size_t)PM_NFFT);
    if (!rc) rc = p->pool.alloc(&p->fb, fb.size());
    if (rc) { delete p; return rc; }
    if (hipMemcpyAsync(p->window, win.data(), win.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(p->tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(p->fb, fb.data(), fb.size() * 4, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        delete p;
        fy_set_error("fy_prompt_mel_create: upload failed");
        return FY_ERR_HIP;
    }
    *out = p;
    return FY_OK;
}

extern "C" void fy_prompt_mel_destroy(fy_prompt_mel* p) { delete p; }

extern "C" int fy_prompt_mel_frames(int32_t n_samples) { return n_samples < PM_PAD + 1 ? 0 : (n_samples + 2 * PM_PAD - PM_NFFT) / PM_HOP + 1; }

extern "C" int fy_prompt_mel_run(fy_prompt_mel* p, const float* wav, int32_t n_samples, float* mel, int32_t frames, void* stream) {
    FY_CHECK(p && wav, FY_ERR_ARG, "fy_prompt_mel_run: null argument");
    FY_CHECK(n_samples > PM_PAD, FY_ERR_ARG, "fy_prompt_mel_run: %d samples are fewer than the reflect padding needs (%d)", n_samples, PM_PAD + 1);
    FY_CHECK(frames == fy_prompt_mel_frames(n_samples), FY_ERR_ARG, "fy_prompt_mel_run: %d samples give %d frames, not %d", n_samples,
             fy_prompt_mel_frames(n_samples), frames);
    FY_CHECK(mel, FY_ERR_ARG, "fy_prompt_mel_run: null argument");
    hipLaunchKernelGGL(prompt_mel_k, dim3(frames), dim3(256), 0, (hipStream_t)stream, wav, n_samples, p->window, p->tw, p->fb, mel);
    HIP_TRY(hipGetLastError());
    return FY_OK;
}
