"""Host side of the HiFT vocoder: mirrors `CausalHiFTGenerator.inference`
(CosyVoice/cosyvoice/hifigan/generator.py:713-726) over the C ABI.

torch is used for device memory and streams only; all arithmetic happens in
libfy_cosy3's HIP kernels.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import torch

from . import _lib
from ._lib import FY_DIRECT, FY_NO_FINALIZE, FY_PRECISE, check  # noqa: F401
from .spec import HiftCfg


def _cfg_struct(cfg: HiftCfg) -> _lib.HiftConfig:
    c = _lib.HiftConfig()
    _lib.lib().fy_hift_default_config(C.byref(c))
    c.mel, c.base, c.harmonics, c.sampling_rate = cfg.mel, cfg.base, cfg.harmonics, 24000
    c.nsf_alpha, c.nsf_sigma, c.voiced_thr = cfg.nsf_alpha, cfg.nsf_sigma, cfg.voiced_thr
    for i in range(3):
        c.ups[i], c.up_k[i], c.rb_k[i], c.rb_d[i], c.src_rb_k[i] = cfg.ups[i], cfg.up_k[i], cfg.rb_k[i], cfg.rb_d[i], cfg.src_rb_k[i]
    c.n_fft, c.hop, c.lrelu, c.audio_limit = cfg.n_fft, cfg.hop, cfg.lrelu, cfg.audio_limit
    c.pre_look_right, c.f0_ch = cfg.pre_look_right, cfg.f0_ch
    return c


class HiftEngine:
    """Batched, ragged HiFT vocoder on one GPU.

    weights: the hift.pt state_dict (reference key names, fp32 CUDA tensors)."""

    def __init__(self, weights: Dict[str, torch.Tensor], cfg: HiftCfg = HiftCfg(), max_batch: int = 8,
                 max_frames: int = 1500, device: Optional[torch.device] = None):
        self.cfg = cfg
        self.device = device or next(iter(weights.values())).device
        self.max_batch, self.max_frames = max_batch, max_frames
        self._h = C.c_void_p()
        arr, keep = _lib.tensor_table(weights)
        cs = _cfg_struct(cfg)
        with torch.cuda.device(self.device):
            check(_lib.lib().fy_hift_create(C.byref(self._h), C.byref(cs), arr, len(weights), max_batch, max_frames,
                                            self._stream()))
        del keep

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if self._h:
            _lib.lib().fy_hift_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _prep(self, mel: torch.Tensor, frames: Optional[Sequence[int]]):
        assert mel.is_cuda and mel.dtype == torch.float32 and mel.dim() == 3 and mel.shape[1] == self.cfg.mel
        mel = mel.contiguous()
        B, _, Fmax = mel.shape
        frames = [Fmax] * B if frames is None else list(frames)
        return mel, B, Fmax, _lib.int_array(frames)

    def inference(self, speech_feat: torch.Tensor, rand_ini: torch.Tensor, sine_noise: torch.Tensor,
                  frames: Optional[Sequence[int]] = None, flags: int = 0, want_source: bool = False, finalize: bool = True):
        """speech_feat (B, 80, Fmax) -> (wav (B, 480*Fmax), source (B, 1, 480*Fmax) or None).
        Samples past 480*frames[b] of a shorter utterance are left untouched (zeros).
        finalize=False (generator.py:713-726, a streaming chunk): 3 + 4 frames are look-ahead and one frame of samples is held
        back: utterance b is valid in wav[b, :480*(frames[b] - 8)], its source in [:480*(frames[b] - 3)]."""
        if not finalize:
            flags |= FY_NO_FINALIZE
        mel, B, Fmax, fr = self._prep(speech_feat, frames)
        S = Fmax * self.cfg.upsample_total
        assert sine_noise.is_cuda and sine_noise.is_contiguous() and sine_noise.shape[-2] >= S
        wav = torch.zeros(B, S, device=mel.device, dtype=torch.float32)
        src = torch.zeros(B, S, device=mel.device, dtype=torch.float32) if want_source else None
        check(_lib.lib().fy_hift_infer(self._h, mel.data_ptr(), fr, B, Fmax, rand_ini.data_ptr(), sine_noise.data_ptr(),
                                       wav.data_ptr(), src.data_ptr() if want_source else None, flags, self._stream()))
        return wav, (src[:, None] if want_source else None)

    def f0(self, speech_feat, frames=None):
        mel, B, Fmax, fr = self._prep(speech_feat, frames)
        out = torch.zeros(B, Fmax, device=mel.device, dtype=torch.float32)
        check(_lib.lib().fy_hift_f0(self._h, mel.data_ptr(), fr, B, Fmax, out.data_ptr(), self._stream()))
        return out

    def source(self, f0, rand_ini, sine_noise, frames=None):
        B, Fmax = f0.shape
        fr = _lib.int_array([Fmax] * B if frames is None else list(frames))
        out = torch.zeros(B, Fmax * self.cfg.upsample_total, device=f0.device, dtype=torch.float32)
        check(_lib.lib().fy_hift_source(self._h, f0.contiguous().data_ptr(), fr, B, Fmax, rand_ini.data_ptr(),
                                        sine_noise.data_ptr(), out.data_ptr(), self._stream()))
        return out[:, None]

    def decode(self, speech_feat, source, frames=None, flags: int = 0):
        mel, B, Fmax, fr = self._prep(speech_feat, frames)
        S = Fmax * self.cfg.upsample_total
        src = source.reshape(B, S).contiguous()
        wav = torch.zeros(B, S, device=mel.device, dtype=torch.float32)
        check(_lib.lib().fy_hift_decode(self._h, mel.data_ptr(), src.data_ptr(), fr, B, Fmax, wav.data_ptr(), flags,
                                        self._stream()))
        return wav

    def tap(self, name: str, B: int, Fmax: int) -> torch.Tensor:
        """Internal tensor of the last call in the reference's (B, C, L) layout."""
        rows, cols = C.c_int64(), C.c_int64()
        cap = B * (Fmax * self.cfg.upsample_total + 1) * 64
        buf = torch.empty(cap, device=self.device)
        check(_lib.lib().fy_hift_tap(self._h, name.encode(), buf.data_ptr(), C.byref(rows), C.byref(cols), self._stream()))
        return buf[: B * rows.value * cols.value].view(B, rows.value, cols.value).transpose(1, 2).contiguous()

    def resblock(self, index: int, x: torch.Tensor, flags: int = 0) -> torch.Tensor:
        """One main-stack ResBlock on x (B, C, L) (reference layout) -> (B, C, L)."""
        xcl = x.transpose(1, 2).contiguous()
        y = torch.empty_like(xcl)
        B, L, _ = xcl.shape
        check(_lib.lib().fy_hift_resblock(self._h, index, xcl.data_ptr(), y.data_ptr(), B, L, flags, self._stream()))
        return y.transpose(1, 2).contiguous()
