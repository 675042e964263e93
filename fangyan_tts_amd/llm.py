"""Host side of the speech-token LM: mirrors `CosyVoice3LM.inference`
(CosyVoice/cosyvoice/llm/llm.py:713-748) plus the token filter of
`CosyVoiceModel.llm_job` (cli/model.py:101-129) over the C ABI, for a batch."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib
from ._lib import check
from .spec import LlmCfg


def _cfg_struct(cfg: LlmCfg, weight_planes: int = 0) -> _lib.LlmConfig:
    c = _lib.LlmConfig()
    _lib.lib().fy_llm_default_config(C.byref(c))
    c.hidden, c.layers, c.q_heads, c.kv_heads, c.head_dim = cfg.hidden, cfg.layers, cfg.q_heads, cfg.kv_heads, cfg.head_dim
    c.inter, c.vocab, c.speech_tokens, c.rms_eps, c.rope_theta = cfg.inter, cfg.vocab, cfg.speech_tokens, cfg.rms_eps, cfg.rope_theta
    c.weight_planes = weight_planes
    return c


class LlmEngine:
    """weights: the llm.pt state_dict (reference key names, fp32 CUDA tensors; lm_head not needed).
    weight_planes: 0 (default) = the library chooses - two bf16 planes per matrix (w = hi + lo) exactly when some weight is not
    bf16-representable, as in a real checkpoint (the reference runs llm.pt in fp32, cli/cosyvoice.py:193), else one; 1 / 2 force it."""

    def __init__(self, weights: Dict[str, torch.Tensor], cfg: LlmCfg = LlmCfg(), max_batch: int = 8, max_ctx: int = 1024,
                 device: Optional[torch.device] = None, keep_weights: bool = False, weight_planes: int = 0):
        self.cfg = cfg
        self._weight_planes = weight_planes
        self.device = device or next(iter(weights.values())).device
        self.max_batch, self.max_ctx = max_batch, max_ctx
        self._h = C.c_void_p()
        weights = {k: v for k, v in weights.items() if "lm_head" not in k}
        self._weights = weights if keep_weights else None          # needed only for load_state_dict
        self._peers = []          # other handles built from the same weights (a model's extra LM handles and lanes): they follow load_state_dict
        self._create(weights)

    def _create(self, weights):
        arr, keep = _lib.tensor_table(weights)
        cs = _cfg_struct(self.cfg, self._weight_planes)
        with torch.cuda.device(self.device):
            check(_lib.lib().fy_llm_create(C.byref(self._h), C.byref(cs), arr, len(weights), self.max_batch, self.max_ctx, self._stream()))
        del keep

    def load_state_dict(self, state_dict, strict: bool = True):
        """torch.nn.Module.load_state_dict's contract for the keys of llm.pt (compare_inference.py:36-43 swaps a
        fine-tuned LM in with strict=False): the engine is rebuilt from the merged weights."""
        if self._weights is None:
            raise RuntimeError("LlmEngine was built without keep_weights=True; it cannot merge a partial state_dict")
        known = set(self._weights) | {"llm.model.lm_head.weight"}
        unexpected = [k for k in state_dict if k not in known]
        missing = [k for k in self._weights if k not in state_dict]
        if strict and (unexpected or missing):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]}, unexpected {unexpected[:5]}")
        for k, v in state_dict.items():
            if k in self._weights:
                if tuple(v.shape) != tuple(self._weights[k].shape):
                    raise RuntimeError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(self._weights[k].shape)}")
                self._weights[k] = v.detach().to(self.device, torch.float32).contiguous()
        for e in [self] + list(self._peers):                      # every handle that was built from these weights
            e.close()
            e._create(self._weights)
        return missing, unexpected

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if self._h:
            _lib.lib().fy_llm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _pack(self, text, prompt_text, prompt_speech, min_len, max_len, min_ratio, max_ratio):
        B = len(text)
        ids, n_all, ps, n_ps = [], [], [], []
        for b in range(B):
            ids += list(prompt_text[b]) + list(text[b])
            n_all.append(len(prompt_text[b]) + len(text[b]))
            ps += list(prompt_speech[b])
            n_ps.append(len(prompt_speech[b]))
        mn = [int(len(text[b]) * min_ratio) for b in range(B)] if min_len is None else list(min_len)
        mx = [int(len(text[b]) * max_ratio) for b in range(B)] if max_len is None else list(max_len)
        out_ld = max(mx)
        out = torch.zeros(B, out_ld, dtype=torch.int32, device=self.device)
        out_n = torch.zeros(B, dtype=torch.int32, device=self.device)
        raw_n = torch.zeros(B, dtype=torch.int32, device=self.device)
        args = (_lib.int_array(ids), _lib.int_array(n_all), _lib.int_array(ps or [0]), _lib.int_array(n_ps), _lib.int_array(mn),
                _lib.int_array(mx), B, out.data_ptr(), out_ld)
        return args, out, out_n, raw_n

    def generate(self, text: Sequence[Sequence[int]], prompt_text: Sequence[Sequence[int]],
                 prompt_speech: Sequence[Sequence[int]], min_len: Optional[Sequence[int]] = None,
                 max_len: Optional[Sequence[int]] = None, min_ratio: float = 2, max_ratio: float = 20):
        """Per-sequence python lists of ids -> (out_ids (B, max(max_len)) int32 CUDA, out_n (B) CUDA, raw_n (B) CUDA).
        min/max default to llm.py:743-744 (2x / 20x the text-only length)."""
        args, out, out_n, raw_n = self._pack(text, prompt_text, prompt_speech, min_len, max_len, min_ratio, max_ratio)
        check(_lib.lib().fy_llm_generate(self._h, *args, out_n.data_ptr(), raw_n.data_ptr(), 0, self._stream()))
        return out, out_n, raw_n

    def begin(self, text, prompt_text, prompt_speech, min_len=None, max_len=None, min_ratio: float = 2, max_ratio: float = 20):
        """The generator form of `inference` (llm.py:511-525 yields ids one by one): prefill + the first token.  Returns
        (out_ids, out_n, raw_n) as `generate`; they fill up as `step` is called."""
        args, out, out_n, raw_n = self._pack(text, prompt_text, prompt_speech, min_len, max_len, min_ratio, max_ratio)
        check(_lib.lib().fy_llm_begin(self._h, *args, self._stream()))
        self._gen = (out, out_n, raw_n, len(text))
        return out, out_n, raw_n

    def prefill_embeds(self, embeds: Sequence[torch.Tensor], min_len: Sequence[int], max_len: Sequence[int]):
        """The embeddings-level entry - the reference's vLLM hand-off (llm.py:482-510 passes `prompt_embeds` = lm_input (L, 896)
        per request): the caller has assembled lm_input itself.  embeds: one (L_b, hidden) CUDA tensor per sequence, fp32 or
        bf16 (all alike).  Prefill + the first token; `step` continues.  Returns (out_ids, out_n, raw_n) as `begin`."""
        B = len(embeds)
        dt = embeds[0].dtype
        assert dt in (torch.float32, torch.bfloat16) and all(e.dtype == dt and e.dim() == 2 and e.shape[1] == self.cfg.hidden for e in embeds)
        rows = torch.cat([e.to(self.device) for e in embeds], dim=0).contiguous()
        out_ld = max(int(m) for m in max_len)
        out = torch.zeros(B, out_ld, dtype=torch.int32, device=self.device)
        out_n = torch.zeros(B, dtype=torch.int32, device=self.device)
        raw_n = torch.zeros(B, dtype=torch.int32, device=self.device)
        check(_lib.lib().fy_llm_prefill(self._h, rows.data_ptr(), 0 if dt == torch.float32 else 1, _lib.int_array([e.shape[0] for e in embeds]),
                                        _lib.int_array(min_len), _lib.int_array(max_len), B, out.data_ptr(), out_ld, self._stream()))
        self._gen = (out, out_n, raw_n, B)
        return out, out_n, raw_n

    def step(self, n_steps: int):
        """Up to `n_steps` more tokens for the sequences still running -> (tokens kept so far per sequence, finished flags)."""
        out, out_n, raw_n, B = self._gen
        fin = (C.c_int32 * B)()
        check(_lib.lib().fy_llm_step(self._h, int(n_steps), out.data_ptr(), out.shape[1], out_n.data_ptr(), raw_n.data_ptr(), fin,
                                     self._stream()))
        return out_n.cpu().tolist(), [bool(f) for f in fin]

    def set_sampler(self, kind: str = "greedy", uniforms: Optional[torch.Tensor] = None, top_k: int = 25, top_p: float = 0.8,
                    win_size: int = 10, tau_r: float = 0.1):
        """"greedy" (SURVEY 8 a4 rule) or "ras": the reference's default repetition-aware sampling (utils/common.py:137-166,
        parameters of cosyvoice3.yaml) with torch.multinomial's draws replaced by the inverse CDF at `uniforms`
        (float32 CUDA (max_batch, n), one row per sequence, consumed from the start at every generate call)."""
        if kind == "greedy":
            check(_lib.lib().fy_llm_set_sampler(self._h, 0, None, 0, top_k, top_p, win_size, tau_r))
            self._uniforms = None
            return
        assert kind == "ras" and uniforms is not None and uniforms.is_cuda and uniforms.dtype == torch.float32
        u = uniforms.reshape(-1, uniforms.shape[-1]).contiguous()
        assert u.shape[0] >= self.max_batch, "one row of uniforms per sequence slot"
        self._uniforms = u                                   # borrowed by the library until the next call
        check(_lib.lib().fy_llm_set_sampler(self._h, 1, u.data_ptr(), u.shape[1], top_k, top_p, win_size, tau_r))

    def set_decode_mode(self, persistent):
        """True / 1 (default): one persistent launch per token step (lowest latency; for <= 8 sequences it holds most of the chip).
        False / 0: one launch per operation, whose short kernels interleave with other streams.
        2: only the few-CU persistent step (any batch <= 32) - what tts_pipeline uses beside its flow decoders."""
        check(_lib.lib().fy_llm_set_decode_mode(self._h, int(persistent)))

    @property
    def persistent(self) -> bool:
        return bool(_lib.lib().fy_llm_decode_mode(self._h))

    @property
    def decode_mode(self) -> int:
        """0, 1 or 2 as set_decode_mode takes them (0 also when the handle has no persistent step for the mode asked for)."""
        return int(_lib.lib().fy_llm_decode_mode(self._h))

    @property
    def weight_planes(self) -> int:
        return int(_lib.lib().fy_llm_weight_planes(self._h))

    def logp(self, step: int, B: int) -> torch.Tensor:
        buf = torch.empty(B, self.cfg.n_speech, device=self.device)
        check(_lib.lib().fy_llm_logp(self._h, step, buf.data_ptr(), self._stream()))
        return buf
