"""The oracle's restatements of third-party algorithms the reference calls, held to INDEPENDENT implementations of the same
published algorithms that happen to be in this image (transformers 5.x) - the original libraries (librosa, openai-whisper,
torchaudio, x-transformers) are neither installed nor vendored under /root/reference, so they cannot be imported here:

  * the Slaney mel filterbank (librosa.filters.mel as matcha/utils/audio.py:45-82 and whisper/audio.py use it)
      against transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney");
  * whisper.log_mel_spectrogram(speech, n_mels=128) (cosyvoice/cli/frontend.py:97)
      against transformers.WhisperFeatureExtractor's numpy feature extraction (the port the Hugging Face Whisper models are fed with);
  * torchaudio.compliance.kaldi.fbank(num_mel_bins=80, dither=0) minus its mean over frames (cli/frontend.py:111-115)
      against SeamlessM4TFeatureExtractor's "numpy method to mimic Kaldi" (it scales the waveform by 2^15: a constant 2 ln 2^15 in
      the log domain, which the reference's mean subtraction removes);
  * the interleaved-pair rotary embedding of x-transformers (flow/DiT/modules.py:368-373; oracle/flow.py rope_freqs / apply_rope)
      against GPT-J's rotate_every_two form in transformers (the same published convention: theta 10000, pairs (2i, 2i + 1)).
This pins the ARITHMETIC of the restatements; that the reference's libraries follow these conventions rests on their documentation
(DESIGN.md section 2)."""
import numpy as np
import pytest
import torch

from oracle import flow as oflow
from oracle import frontend as ofe

transformers = pytest.importorskip("transformers")


def _wave(seconds, seed):
    rng = np.random.default_rng(seed)
    n = int(16000 * seconds)
    t = np.arange(n)
    return (0.05 * rng.standard_normal(n) + 0.2 * np.sin(t / 37.0) * np.sin(t / 2900.0)).astype(np.float32)


@pytest.mark.parametrize("sr,n_fft,n_mels", [(24000, 1920, 80), (16000, 400, 128), (16000, 400, 80)])
def test_slaney_filterbank_equals_the_transformers_port(sr, n_fft, n_mels):
    from transformers.audio_utils import mel_filter_bank
    want = mel_filter_bank(num_frequency_bins=1 + n_fft // 2, num_mel_filters=n_mels, min_frequency=0.0, max_frequency=sr / 2.0,
                           sampling_rate=sr, norm="slaney", mel_scale="slaney").T
    got = ofe.slaney_mel_filterbank(sr, n_fft, n_mels)
    assert got.shape == want.shape
    assert float(np.abs(got - want).max()) <= 1e-7 * float(np.abs(want).max()) + 1e-9      # measured 2e-9 on tables that peak at 0.02-0.04


@pytest.mark.parametrize("seconds", [0.37, 3.1, 9.98])
def test_whisper_log_mel_equals_the_transformers_port(seconds):
    fe = transformers.WhisperFeatureExtractor(feature_size=128)
    wav = _wave(seconds, 1)
    want = fe._np_extract_fbank_features(wav[None], "cpu")[0]
    got = ofe.whisper_log_mel(torch.from_numpy(wav)[None])[0].numpy()
    assert got.shape == want.shape == (128, len(wav) // 160)
    assert float(np.abs(got - want).max()) <= 1e-4                         # measured 1e-6 .. 1.3e-5 (float32 STFT against float64)


@pytest.mark.parametrize("seconds", [0.31, 2.0, 7.77])
def test_kaldi_fbank_equals_the_transformers_port(seconds):
    fe = transformers.SeamlessM4TFeatureExtractor(feature_size=80, num_mel_bins=80, sampling_rate=16000)
    wav = _wave(seconds, 2)
    want = fe._extract_fbank_features(wav)
    got = ofe.kaldi_fbank(torch.from_numpy(wav)[None]).numpy()
    assert got.shape == want.shape
    # the port works on the waveform scaled to 16-bit integers: + 2 ln 2^15 on every log energy
    assert float(np.abs((want - 2.0 * np.log(32768.0)) - got).max()) <= 1e-3          # measured 2e-5 .. 1.4e-4
    # what the reference feeds the speaker embedder: the features minus their mean over frames
    assert float(np.abs((want - want.mean(0, keepdims=True)) - (got - got.mean(0, keepdims=True))).max()) <= 1e-3


@pytest.mark.parametrize("T", [1, 37, 650])
def test_rotary_embedding_equals_the_gptj_form(T):
    from transformers.models.gptj import modeling_gptj as gj
    g = torch.Generator().manual_seed(T)
    x = torch.randn(2, T, 1024, generator=g)
    got = oflow.apply_rope(x, oflow.rope_freqs(T, 64))
    sincos = gj.create_sinusoidal_positions(T, 64)                         # (T, 64) = [sin | cos], 32 frequencies each
    sin, cos = sincos[None, :, :32], sincos[None, :, 32:]
    want_rot = gj.apply_rotary_pos_emb(x[:, :, None, :64], sin, cos)[:, :, 0, :]
    assert torch.equal(got[..., 64:], x[..., 64:])                         # only the first 64 of the 1024 un-split channels rotate
    assert float((got[..., :64] - want_rot).abs().max()) <= 2e-6 * float(x.abs().max())
