"""ORACLE (test infrastructure, not product code) - the frontend's prompt mel:
matcha.utils.audio.mel_spectrogram as cosyvoice3.yaml:140-148 configures it (n_fft = win = 1920, hop 480, 80 mels, fmin 0,
fmax None = sr/2, center=False), CosyVoice/third_party/Matcha-TTS/matcha/utils/audio.py:45-82.

Pinning: the STFT is torch.stft itself, called as the reference calls it (audio.py:64-76).  The mel filterbank is
librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with librosa's defaults (htk=False, norm="slaney"); librosa is NOT in
this image and not vendored, so `slaney_mel_filterbank` restates its published definition.  The reference holds no test vector
for it; the table is held to an INDEPENDENT port of the same definition, transformers.audio_utils.mel_filter_bank(norm="slaney",
mel_scale="slaney"), to 2e-9 (tests/test_independent_ports_cpu.py) - pinned to a port, not to librosa itself.
"""
import numpy as np
import torch


def slaney_mel_filterbank(sr: int, n_fft: int, n_mels: int, fmin: float = 0.0, fmax=None) -> np.ndarray:
    fmax = sr / 2.0 if fmax is None else fmax
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        w[i] = np.maximum(0.0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def mel_spectrogram(y: torch.Tensor, n_fft: int = 1920, num_mels: int = 80, sampling_rate: int = 24000, hop_size: int = 480,
                    win_size: int = 1920, fmin: float = 0.0, fmax=None) -> torch.Tensor:
    """y (1, S) in [-1, 1] -> log-mel (1, 80, F), audio.py:45-82."""
    fb = torch.from_numpy(slaney_mel_filterbank(sampling_rate, n_fft, num_mels, fmin, fmax))
    pad = int((n_fft - hop_size) / 2)
    y = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.view_as_real(torch.stft(y, n_fft, hop_length=hop_size, win_length=win_size, window=torch.hann_window(win_size),
                                         center=False, pad_mode="reflect", normalized=False, onesided=True, return_complex=True))
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)
    return torch.log(torch.clamp(torch.matmul(fb, spec), min=1e-5))


# ---- the 16 kHz features of the reference's two ONNX models (cli/frontend.py:94-117) -----------------------------------------
# whisper (openai-whisper, whisper/audio.py:log_mel_spectrogram) and torchaudio (torchaudio/compliance/kaldi.py:fbank) are NOT in
# this image and not vendored by the reference: both are restated from their published algorithms.  Neither library can be imported
# here; the restatements are held to independent ports of the same algorithms in transformers (tests/test_independent_ports_cpu.py):
# WhisperFeatureExtractor's numpy extraction (1e-5) and SeamlessM4TFeatureExtractor's "numpy method to mimic Kaldi" (1.4e-4 in the
# log domain after its 2^15 waveform scale) - pinned to ports, not to the libraries themselves.  The transforms are torch.stft /
# torch.fft.rfft, called as those libraries call them.

def whisper_log_mel(audio: torch.Tensor, n_mels: int = 128) -> torch.Tensor:
    """audio (1, S) at 16 kHz -> (1, n_mels, S // 160): hann(400), stft(400, hop 160, center, reflect), |.|^2 of all frames but the
    last, librosa mel filterbank (Slaney), log10(clamp 1e-10), clamp to max - 8, (x + 4) / 4."""
    window = torch.hann_window(400)
    stft = torch.stft(audio, 400, 160, window=window, return_complex=True)
    magnitudes = stft[..., :-1].abs() ** 2
    filters = torch.from_numpy(slaney_mel_filterbank(16000, 400, n_mels))
    log_spec = torch.clamp(filters @ magnitudes, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    return (log_spec + 4.0) / 4.0


def kaldi_mel_banks(num_bins: int = 80, padded: int = 512, sr: float = 16000.0, low_freq: float = 20.0) -> np.ndarray:
    """kaldi.get_mel_banks without VTLN warp: (num_bins, padded / 2 + 1), the Nyquist column zero."""
    mel = lambda f: 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)
    nb = padded // 2
    mel_low, mel_high = mel(low_freq), mel(sr / 2.0)
    delta = (mel_high - mel_low) / (num_bins + 1)
    left = mel_low + np.arange(num_bins)[:, None] * delta
    center, right = left + delta, left + 2.0 * delta
    m = mel(sr / padded * np.arange(nb))[None, :]
    w = np.maximum(0.0, np.minimum((m - left) / (center - left), (right - m) / (right - center)))
    return np.pad(w, ((0, 0), (0, 1))).astype(np.float32)


def kaldi_fbank(speech: torch.Tensor, num_mel_bins: int = 80, sr: float = 16000.0) -> torch.Tensor:
    """speech (1, S) -> (frames, 80): kaldi.fbank's defaults with dither 0 (25 ms / 10 ms frames, snip_edges, remove_dc_offset,
    preemphasis 0.97, povey window, padded to 512, power spectrum, log mel energies floored at float32 epsilon)."""
    x = speech.reshape(-1).to(torch.float32)
    win, hop, padded = int(sr * 0.025), int(sr * 0.010), 512
    m = 1 + (x.numel() - win) // hop
    frames = x.unfold(0, win, hop)[:m].clone()
    frames = frames - frames.mean(dim=1, keepdim=True)
    prev = torch.nn.functional.pad(frames.unsqueeze(0), (1, 0), mode="replicate").squeeze(0)[:, :-1]
    frames = frames - 0.97 * prev
    frames = frames * torch.hann_window(win, periodic=False).pow(0.85)
    frames = torch.nn.functional.pad(frames, (0, padded - win))
    power = torch.fft.rfft(frames).abs().pow(2.0)
    mel = power @ torch.from_numpy(kaldi_mel_banks(num_mel_bins, padded, sr)).T
    return torch.max(mel, torch.tensor(torch.finfo(torch.float32).eps)).log()


def resample_direct(x: np.ndarray, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> np.ndarray:
    """torchaudio.transforms.Resample's output evaluated sample by sample from its published formula (float64), with no filter
    bank and no convolution: output m = q n + j (phase j of block q) is
        sum over i in [-width, width + o) of x[q o + i] * sinc(pi t) * cos(pi t / (2 lpw))^2 * f / o,
        t = clamp((-j / n + i / o) * f, -lpw, lpw),  f = rolloff * min(o, n),  x = 0 outside [0, S)."""
    from math import ceil, gcd
    g = gcd(orig_freq, new_freq)
    o, n = orig_freq // g, new_freq // g
    f = min(o, n) * rolloff
    width = int(ceil(lowpass_filter_width * o / f))
    S = x.shape[-1]
    T = -(-n * S // o)
    out = np.zeros(T, dtype=np.float64)
    xs = np.asarray(x, dtype=np.float64).reshape(-1)
    i = np.arange(-width, width + o)
    for m in range(T):
        q, j = divmod(m, n)
        # phase offsets are float32 quotients in torchaudio (an int64 arange divided by new_freq)
        t = np.clip((float(np.float32(-j) / np.float32(n)) + i / o) * f, -lowpass_filter_width, lowpass_filter_width)
        w = np.cos(t * np.pi / lowpass_filter_width / 2) ** 2
        tp = t * np.pi
        k = np.where(tp == 0, 1.0, np.sin(tp) / np.where(tp == 0, 1.0, tp)) * w * (f / o)
        pos = q * o + i
        ok = (pos >= 0) & (pos < S)
        out[m] = float((xs[pos[ok]] * k.astype(np.float32).astype(np.float64)[ok]).sum())
    return out
