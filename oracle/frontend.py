"""ORACLE (test infrastructure, not product code) - the frontend's prompt mel:
matcha.utils.audio.mel_spectrogram as cosyvoice3.yaml:140-148 configures it (n_fft = win = 1920, hop 480, 80 mels, fmin 0,
fmax None = sr/2, center=False), CosyVoice/third_party/Matcha-TTS/matcha/utils/audio.py:45-82.

Pinning: the STFT is torch.stft itself, called as the reference calls it (audio.py:64-76).  The mel filterbank is
librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with librosa's defaults (htk=False, norm="slaney"); librosa is NOT in
this image and not vendored, so `slaney_mel_filterbank` restates its published definition - PARITY UNPINNED for that table
(the reference holds no test vector for it either).
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
