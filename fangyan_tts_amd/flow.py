"""Host side of the flow-matching decoder: mirrors `CausalMaskedDiffWithDiT.inference`
(CosyVoice/cosyvoice/flow/flow.py:358-403) and the estimator hand-off of
`ConditionalCFM.forward_estimator` (flow/flow_matching.py:126-153) over the C ABI."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import torch

from . import _lib
from ._lib import FY_DIRECT, FY_NO_FINALIZE, FY_PRECISE, FY_STREAMING, check  # noqa: F401
from .spec import FlowCfg


def reference_t_span(n_timesteps: int) -> torch.Tensor:
    """flow_matching.py:223-225, the same torch ops so the fp32 values are the reference's."""
    t = torch.linspace(0, 1, n_timesteps + 1, dtype=torch.float32)
    return 1 - torch.cos(t * 0.5 * torch.pi)


def _cfg_struct(cfg: FlowCfg) -> _lib.FlowConfig:
    c = _lib.FlowConfig()
    _lib.lib().fy_flow_default_config(C.byref(c))
    c.mel, c.spk_in, c.vocab, c.pre_ch, c.pre_lookahead = cfg.mel, cfg.spk_in, cfg.vocab, cfg.pre_ch, cfg.pre_lookahead
    c.dim, c.depth, c.heads, c.head_dim, c.ff_mult = cfg.dim, cfg.depth, cfg.heads, cfg.head_dim, cfg.ff_mult
    c.conv_pos_k, c.conv_pos_groups, c.n_timesteps = cfg.conv_pos_k, cfg.conv_pos_groups, cfg.n_timesteps
    c.cfg_rate, c.static_chunk = cfg.cfg_rate, cfg.static_chunk
    for i, v in enumerate(reference_t_span(cfg.n_timesteps).tolist()):
        c.t_span[i] = v
    return c


class FlowEngine:
    """weights: the flow.pt state_dict (reference key names, fp32 CUDA tensors)."""

    token_mel_ratio = 2
    pre_lookahead_len = 3

    def __init__(self, weights: Dict[str, torch.Tensor], cfg: FlowCfg = FlowCfg(), max_batch: int = 8,
                 max_frames: int = 1500, device: Optional[torch.device] = None):
        self.cfg = cfg
        self.device = device or next(iter(weights.values())).device
        self.max_batch, self.max_frames = max_batch, max_frames
        self._h = C.c_void_p()
        arr, keep = _lib.tensor_table(weights)
        cs = _cfg_struct(cfg)
        with torch.cuda.device(self.device):
            check(_lib.lib().fy_flow_create(C.byref(self._h), C.byref(cs), arr, len(weights), max_batch, max_frames, self._stream()))
        del keep

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if self._h:
            _lib.lib().fy_flow_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def inference(self, token: torch.Tensor, n_token: Sequence[int], prompt_token: torch.Tensor, n_prompt: Sequence[int],
                  prompt_feat: torch.Tensor, n_pfeat: Sequence[int], embedding: torch.Tensor, rand_noise: torch.Tensor,
                  flags: int = 0, streaming: bool = False, finalize: bool = True, incremental: bool = False) -> torch.Tensor:
        """token (B, Nmax) int32, prompt_token (B, Pmax) int32, prompt_feat (B, PMmax, 80), embedding (B, 192),
        rand_noise (1|-, 80, >=T) -> mel (B, 80, 2*max(n_token)); utterance b is valid in [:, :, :2*n_token[b]].
        streaming / finalize as in the reference (flow.py:358-403): the chunk attention mask, and the last
        pre_lookahead tokens as look-ahead context only (valid frames 2*(n_token[b] - pre_lookahead)).
        incremental (with streaming, finalize=False, one utterance): this call extends the previous incremental call of the handle -
        only the new rows go through the DiT blocks (FY_INCREMENTAL; `stream_reset` starts a new stream).  Same result."""
        if incremental:
            flags |= _lib.FY_INCREMENTAL
        if streaming:
            flags |= FY_STREAMING
        if not finalize:
            flags |= FY_NO_FINALIZE
        B = token.shape[0]
        dev = self.device
        token = token.to(dev, torch.int32).contiguous()
        prompt_token = prompt_token.to(dev, torch.int32).contiguous()
        if prompt_token.shape[1] == 0:
            prompt_token = torch.zeros(B, 1, dtype=torch.int32, device=dev)
        prompt_feat = prompt_feat.to(dev, torch.float32).contiguous()
        if prompt_feat.shape[1] == 0:
            prompt_feat = torch.zeros(B, 1, self.cfg.mel, device=dev)
        embedding = embedding.to(dev, torch.float32).contiguous()
        noise = rand_noise.reshape(self.cfg.mel, -1).to(dev, torch.float32).contiguous()
        frames = 2 * max(int(n) for n in n_token)
        mel = torch.zeros(B, self.cfg.mel, frames, device=dev, dtype=torch.float32)
        check(_lib.lib().fy_flow_infer(self._h, token.data_ptr(), token.shape[1], _lib.int_array(n_token),
                                       prompt_token.data_ptr(), prompt_token.shape[1], _lib.int_array(n_prompt),
                                       prompt_feat.data_ptr(), prompt_feat.shape[1], _lib.int_array(n_pfeat),
                                       embedding.data_ptr(), noise.data_ptr(), noise.shape[1], B, mel.data_ptr(), frames,
                                       flags, self._stream()))
        return mel

    @property
    def weight_planes(self) -> int:
        """1: the weights were bf16-representable; 2: a general fp32 checkpoint (lo planes kept for FY_PRECISE)."""
        return int(_lib.lib().fy_flow_weight_planes(self._h))

    def stream_reset(self):
        """Forget what incremental calls have kept: the next one starts a new stream."""
        check(_lib.lib().fy_flow_stream_reset(self._h))

    def stream_rows(self) -> int:
        """Mel rows the incremental calls of the current stream have kept (0: the next call computes everything)."""
        return int(_lib.lib().fy_flow_stream_rows(self._h))

    def speed(self, mel: torch.Tensor, speed: float) -> torch.Tensor:
        """cli/model.py:435-437: F.interpolate(mel, size=int(F / speed), mode="linear") on (B, 80, F)."""
        mel = mel.to(self.device, torch.float32).contiguous()
        B, C, F = mel.shape
        out = torch.empty(B, C, int(F / speed), device=self.device, dtype=torch.float32)
        check(_lib.lib().fy_mel_speed(mel.data_ptr(), B * C, F, out.data_ptr(), out.shape[2], self._stream()))
        return out

    def estimator(self, x, mask, mu, t, spks, cond, streaming: bool = False, flags: int = 0) -> torch.Tensor:
        """DiT.forward(x, mask, mu, t, spks, cond) on contiguous (B2, 80, T) tensors; returns the result (x is not modified)."""
        out = x.detach().clone().contiguous()
        B2, _, T = out.shape
        if streaming:
            flags |= FY_STREAMING
        check(_lib.lib().fy_dit_estimator(self._h, out.data_ptr(), mask.contiguous().data_ptr() if mask is not None else None,
                                          mu.contiguous().data_ptr(), t.contiguous().data_ptr(), spks.contiguous().data_ptr(),
                                          cond.contiguous().data_ptr(), T, B2, flags, self._stream()))
        return out
