"""ORACLE (test infrastructure, not product code) - CPU restatement of the
CosyVoice3 vocoder: CausalHiFTGenerator + CausalConvRNNF0Predictor.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product path (fangyan_tts_amd) never does.

Pinned against the reference itself: tests/golden/mint_goldens.py imports
/root/reference/CosyVoice/cosyvoice/hifigan/generator.py in the build container,
fills it with fangyan_tts_amd.synth weights and stores its outputs as fixtures
(tests/golden/hift_*.npz); tests/test_oracle_golden.py holds this file to them.

All arithmetic fp32 (torch CPU), like the reference's default path.  Tensors
are (B, C, L) as in the reference.  P maps the reference state_dict names to
torch tensors (weight-norm pairs folded by `prepare`).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from fangyan_tts_amd.spec import HiftCfg

Params = Dict[str, torch.Tensor]


def prepare(sd: Dict[str, np.ndarray]) -> Params:
    """numpy state_dict -> torch fp32, folding weight_norm (g, v) -> weight.

    torch.nn.utils.parametrizations.weight_norm computes
    torch._weight_norm(v, g, dim=0) = g * v / ||v||_(per out channel); the
    reference keeps the pair as `<conv>.parametrizations.weight.original0/1`
    (generator.py:621, 629, 664; f0_predictor.py:74-90)."""
    P: Params = {}
    for k, v in sd.items():
        P[k] = torch.from_numpy(np.ascontiguousarray(v)).float()
    for k in list(P):
        if k.endswith(".parametrizations.weight.original1"):
            base = k[: -len(".parametrizations.weight.original1")]
            g = P[base + ".parametrizations.weight.original0"]
            P[base + ".weight"] = torch._weight_norm(P[k], g, 0)
    return P


# ---- building blocks --------------------------------------------------------

def causal_padding(w, dilation: int = 1) -> int:
    k = w.shape[-1]
    return int((k * dilation - dilation) / 2) * 2 + (k + 1) % 2    # convolution.py:172


def causal_conv1d(x, w, b, dilation: int = 1, causal: str = "left", cache=None):
    """CausalConv1d.forward, transformer/convolution.py:176-187: (k-1)*d zeros (or
    `cache`, the streaming look-ahead / history) on the left ('left') or on the
    right ('right'), then a plain stride-1 conv."""
    pad = causal_padding(w, dilation)
    if cache is None:
        x = F.pad(x, (pad, 0) if causal == "left" else (0, pad))
    else:
        assert cache.shape[2] == pad
        x = torch.cat([cache, x], dim=2) if causal == "left" else torch.cat([x, cache], dim=2)
    y = F.conv1d(x, w, b, dilation=dilation)
    return y


def causal_conv1d_down(x, w, b, stride: int):
    """CausalConv1dDownSample.forward, convolution.py:213-220."""
    return F.conv1d(F.pad(x, (stride - 1, 0)), w, b, stride=stride)


def causal_conv1d_up(x, w, b, stride: int):
    """CausalConv1dUpsample.forward, convolution.py:247-257: nearest repeat by
    `stride`, k-1 zeros on the left, stride-1 conv."""
    x = x.repeat_interleave(stride, dim=2)
    return F.conv1d(F.pad(x, (w.shape[-1] - 1, 0)), w, b)


def snake(x, alpha):
    """Snake.forward, transformer/activation.py:73-84 (alpha_logscale=False)."""
    a = alpha.view(1, -1, 1)
    return x + (1.0 / (a + 1e-9)) * torch.sin(x * a) ** 2


def resblock(x, P: Params, prefix: str, dilations=(1, 3, 5)):
    """ResBlock.forward, hifigan/generator.py:110-117 (causal=True)."""
    for j, d in enumerate(dilations):
        xt = snake(x, P[f"{prefix}.activations1.{j}.alpha"])
        xt = causal_conv1d(xt, P[f"{prefix}.convs1.{j}.weight"], P[f"{prefix}.convs1.{j}.bias"], dilation=d)
        xt = snake(xt, P[f"{prefix}.activations2.{j}.alpha"])
        xt = causal_conv1d(xt, P[f"{prefix}.convs2.{j}.weight"], P[f"{prefix}.convs2.{j}.bias"], dilation=1)
        x = xt + x
    return x


# ---- f0 predictor -----------------------------------------------------------

def f0_predictor(mel, P: Params, finalize: bool = True):
    """CausalConvRNNF0Predictor.forward, f0_predictor.py:95-103.
    mel (B, 80, F) -> f0 (B, F) (finalize) or (B, F - 3) (streaming chunk: the
    last 3 frames are the first conv's look-ahead instead of zeros).  First conv
    looks 3 frames right, the other four are left-causal; ELU after each;
    |Linear(512->1)|."""
    w0 = P["f0_predictor.condnet.0.weight"]
    if finalize:
        x = causal_conv1d(mel, w0, P["f0_predictor.condnet.0.bias"], causal="right")
    else:
        pad = causal_padding(w0)
        x = causal_conv1d(mel[:, :, :-pad], w0, P["f0_predictor.condnet.0.bias"], causal="right", cache=mel[:, :, -pad:])
    x = F.elu(x)
    for i in (2, 4, 6, 8):
        x = F.elu(causal_conv1d(x, P[f"f0_predictor.condnet.{i}.weight"], P[f"f0_predictor.condnet.{i}.bias"]))
    x = x.transpose(1, 2)
    return torch.abs(F.linear(x, P["f0_predictor.classifier.weight"], P["f0_predictor.classifier.bias"]).squeeze(-1))


# ---- harmonic source --------------------------------------------------------

def sine_source(f0, P: Params, cfg: HiftCfg, rand_ini, sine_noise):
    """f0 (B, F) -> source s (B, 1, S) with S = 480 F.

    generator.py:719-721 (nearest x480 upsample of f0), SineGen2.forward /
    _f02sine :289-317 / :233-258 (eval, causal=True), SourceModuleHnNSF.forward
    :358-375.  rand_ini (1, 9) and sine_noise (1, >=S, 9) are the reference's
    fixed non-checkpointed buffers, passed explicitly."""
    up = cfg.upsample_total
    B, Fr = f0.shape
    f0u = f0[:, :, None].repeat_interleave(up, dim=1)                     # (B, S, 1)
    harm = torch.arange(1, cfg.harmonics + 2, dtype=torch.float32).view(1, 1, -1)
    fn = f0u * harm                                                       # generator.py:297
    rad = (fn / 24000.0) % 1                                              # :239 (sampling_rate)
    rad[:, 0, :] = rad[:, 0, :] + rand_ini                                # :243
    # :251-253 linear interpolation down by `up` (align_corners=False)
    rad_ds = F.interpolate(rad.transpose(1, 2), scale_factor=1.0 / up, mode="linear").transpose(1, 2)
    phase = torch.cumsum(rad_ds, dim=1) * 2 * np.pi                       # :255
    phase = F.interpolate(phase.transpose(1, 2) * up, scale_factor=float(up), mode="nearest").transpose(1, 2)
    sines = torch.sin(phase) * cfg.nsf_alpha                              # :258, :300
    uv = (f0u > cfg.voiced_thr).float()                                   # :230
    noise_amp = uv * cfg.nsf_sigma + (1 - uv) * cfg.nsf_alpha / 3         # :308
    noise = noise_amp * sine_noise[:, : sines.shape[1]]                   # :310
    sine_waves = sines * uv + noise                                       # :316
    merged = torch.tanh(F.linear(sine_waves, P["m_source.l_linear.weight"], P["m_source.l_linear.bias"]))
    return merged.transpose(1, 2)                                         # (B, 1, S)


# ---- STFT / iSTFT written out (generator.py:491-505 call torch.stft/istft) --

def _hann(n: int) -> torch.Tensor:
    """scipy.signal.get_window('hann', n, fftbins=True): periodic hann."""
    return (0.5 - 0.5 * torch.cos(2 * math.pi * torch.arange(n, dtype=torch.float64) / n)).float()


def stft(s, cfg: HiftCfg) -> torch.Tensor:
    """s (B, S) -> (B, n_fft+2, S/hop+1) = cat[real(9), imag(9)].
    torch.stft(center=True, pad_mode='reflect', onesided, unnormalised)."""
    n, hop = cfg.n_fft, cfg.hop
    x = F.pad(s[:, None, :], (n // 2, n // 2), mode="reflect")[:, 0]
    frames = x.unfold(1, n, hop)                                          # (B, T, n)
    frames = frames * _hann(n)
    k = torch.arange(n // 2 + 1, dtype=torch.float64)[:, None]
    t = torch.arange(n, dtype=torch.float64)[None, :]
    ang = 2 * math.pi * k * t / n
    cr, ci = torch.cos(ang).float(), (-torch.sin(ang)).float()
    re = torch.einsum("btn,kn->bkt", frames, cr)
    im = torch.einsum("btn,kn->bkt", frames, ci)
    return torch.cat([re, im], dim=1)


def istft(mag, phase, cfg: HiftCfg) -> torch.Tensor:
    """HiFTGenerator._istft, generator.py:499-505: clip(mag, max=100),
    torch.istft(center=True) -> (B, hop*(T-1))."""
    n, hop = cfg.n_fft, cfg.hop
    mag = torch.clip(mag, max=1e2)
    re, im = mag * torch.cos(phase), mag * torch.sin(phase)
    B, K, T = re.shape
    # inverse real DFT: x[t] = 1/n * sum_k c_k (re_k cos - im_k sin), c = 1 for DC/Nyquist else 2
    kk = torch.arange(K, dtype=torch.float64)[:, None]
    tt = torch.arange(n, dtype=torch.float64)[None, :]
    ang = 2 * math.pi * kk * tt / n
    c = torch.full((K, 1), 2.0, dtype=torch.float64)
    c[0] = 1.0
    c[-1] = 1.0
    br, bi = (c * torch.cos(ang) / n).float(), (-c * torch.sin(ang) / n).float()
    fr = torch.einsum("bkt,kn->btn", re, br) + torch.einsum("bkt,kn->btn", im, bi)
    w = _hann(n)
    fr = fr * w
    L = n + hop * (T - 1)
    y = torch.zeros(B, L)
    env = torch.zeros(L)
    for t in range(T):
        y[:, t * hop: t * hop + n] += fr[:, t]
        env[t * hop: t * hop + n] += w * w
    y = y[:, n // 2: L - n // 2] / env[n // 2: L - n // 2]
    return y


# ---- decoder ----------------------------------------------------------------

def decode_taps(mel, s, P: Params, cfg: HiftCfg, finalize: bool = True) -> Dict[str, torch.Tensor]:
    """CausalHiFTGenerator.decode, generator.py:672-711, keeping the per-stage
    tensors the golden fixtures tap.  finalize=False (a streaming chunk, :674-679,
    :708-709): the last `pre_look_right` mel frames are conv_pre's look-ahead, the
    source STFT is trimmed to match and the last 480 samples are held back."""
    taps: Dict[str, torch.Tensor] = {}
    s_stft = stft(s.squeeze(1), cfg)
    up_total = 1
    for u in cfg.ups:
        up_total *= u
    if finalize:
        x = causal_conv1d(mel, P["conv_pre.weight"], P["conv_pre.bias"], causal="right")
    else:
        r = cfg.pre_look_right
        x = causal_conv1d(mel[:, :, :-r], P["conv_pre.weight"], P["conv_pre.bias"], causal="right", cache=mel[:, :, -r:])
        s_stft = s_stft[:, :, :-up_total * r]
    taps["s_stft"] = s_stft
    taps["conv_pre"] = x
    n_up, n_k = len(cfg.ups), len(cfg.rb_k)
    for i in range(n_up):
        x = F.leaky_relu(x, cfg.lrelu)
        x = causal_conv1d_up(x, P[f"ups.{i}.weight"], P[f"ups.{i}.bias"], cfg.ups[i])
        if i == n_up - 1:
            x = F.pad(x, (1, 0), mode="reflect")
        k, st = cfg.source_down(i)
        if st == 1:
            si = causal_conv1d(s_stft, P[f"source_downs.{i}.weight"], P[f"source_downs.{i}.bias"])
        else:
            si = causal_conv1d_down(s_stft, P[f"source_downs.{i}.weight"], P[f"source_downs.{i}.bias"], st)
        si = resblock(si, P, f"source_resblocks.{i}", cfg.rb_d)
        x = x + si
        taps[f"fuse{i}"] = x
        xs = None
        for j in range(n_k):
            r = resblock(x, P, f"resblocks.{i * n_k + j}", cfg.rb_d)
            xs = r if xs is None else xs + r
        x = xs / n_k
        taps[f"stage{i}"] = x
    x = F.leaky_relu(x)                                                    # slope 0.01, generator.py:702
    x = causal_conv1d(x, P["conv_post.weight"], P["conv_post.bias"])
    taps["conv_post"] = x
    half = cfg.n_fft // 2 + 1
    mag = torch.exp(x[:, :half])
    ph = torch.sin(x[:, half:])
    y = istft(mag, ph, cfg)
    if not finalize:
        y = y[:, :-up_total * cfg.hop]
    taps["wav"] = torch.clamp(y, -cfg.audio_limit, cfg.audio_limit)
    return taps


def decode(mel, s, P: Params, cfg: HiftCfg, finalize: bool = True) -> torch.Tensor:
    return decode_taps(mel, s, P, cfg, finalize)["wav"]


def inference(mel, P: Params, cfg: HiftCfg, rand_ini, sine_noise, finalize: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """CausalHiFTGenerator.inference, generator.py:713-726.
    mel (B, 80, F) -> (wav (B, 480 F), source (B, 1, 480 F)); with finalize=False
    (B, 480 (F - 8)) and (B, 1, 480 (F - 3)): 3 frames of f0 look-ahead, 4 of
    conv_pre look-ahead, one frame of samples held back."""
    with torch.no_grad():
        f0 = f0_predictor(mel, P, finalize)
        s = sine_source(f0, P, cfg, rand_ini, sine_noise)
        if finalize:
            return decode(mel, s, P, cfg, True), s
        pad = causal_padding(P["f0_predictor.condnet.0.weight"])
        return decode(mel[:, :, :-pad], s, P, cfg, False), s
