"""ORACLE (test infrastructure, not product code) - CPU restatement of the
CosyVoice3 speech-token language model: CosyVoice3LM.inference ->
Qwen2LM.inference_wrapper -> Qwen2Encoder.forward_one_step (HF Qwen2 body).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it.

The transformer body is third-party (`transformers` Qwen2ForCausalLM, pinned
4.51.3 in CosyVoice/requirements.txt:38; 5.15.0 is importable in the build
container).  Its published algorithm is restated here - RMSNorm, q/k/v bias,
split-half RoPE (theta 1e6), grouped-query causal attention with a KV cache,
SwiGLU - and pinned by tests/golden/mint_goldens.py, which runs the reference's
own llm/llm.py:713-748 over a real Qwen2ForCausalLM filled with synth weights
and stores token ids and log-probs (tests/golden/llm_*.npz).

fp32 torch CPU.
"""
from __future__ import annotations

from typing import Dict, Iterator, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from fangyan_tts_amd.spec import LlmCfg, SILENT_TOKENS, MAX_SILENT_RUN

Params = Dict[str, torch.Tensor]
L = "llm.model.model."


def prepare(sd: Dict[str, np.ndarray]) -> Params:
    return {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in sd.items()}


def rms_norm(x, w, eps):
    """Qwen2RMSNorm: x * rsqrt(mean(x^2) + eps) * w, in fp32."""
    var = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def rope_cos_sin(pos: torch.Tensor, cfg: LlmCfg):
    """Qwen2RotaryEmbedding: inv_freq = theta^(-2i/d); emb = cat(freqs, freqs)."""
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, cfg.head_dim, 2).float() / cfg.head_dim))
    f = pos.float()[:, None] * inv[None, :]
    emb = torch.cat((f, f), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


class KVCache:
    def __init__(self, layers: int):
        self.k: List[Optional[torch.Tensor]] = [None] * layers
        self.v: List[Optional[torch.Tensor]] = [None] * layers

    @property
    def length(self) -> int:
        return 0 if self.k[0] is None else self.k[0].shape[1]


def forward_one_step(xs, cache: KVCache, P: Params, cfg: LlmCfg):
    """Qwen2Encoder.forward_one_step, llm/llm.py:246-258: run (1, n, H) new
    positions over the cache, return hidden_states[-1] (after the final norm)."""
    n = xs.shape[1]
    past = cache.length
    pos = torch.arange(past, past + n)
    cos, sin = rope_cos_sin(pos, cfg)                                     # (n, 64)
    Hq, Hk, D = cfg.q_heads, cfg.kv_heads, cfg.head_dim
    h = xs[0]
    for i in range(cfg.layers):
        p = L + f"layers.{i}."
        r = h
        x = rms_norm(h, P[p + "input_layernorm.weight"], cfg.rms_eps)
        q = F.linear(x, P[p + "self_attn.q_proj.weight"], P[p + "self_attn.q_proj.bias"]).view(n, Hq, D)
        k = F.linear(x, P[p + "self_attn.k_proj.weight"], P[p + "self_attn.k_proj.bias"]).view(n, Hk, D)
        v = F.linear(x, P[p + "self_attn.v_proj.weight"], P[p + "self_attn.v_proj.bias"]).view(n, Hk, D)
        q = q * cos[:, None, :] + rotate_half(q) * sin[:, None, :]
        k = k * cos[:, None, :] + rotate_half(k) * sin[:, None, :]
        k = k.transpose(0, 1)                                             # (Hk, n, D)
        v = v.transpose(0, 1)
        if cache.k[i] is not None:
            k = torch.cat([cache.k[i], k], dim=1)
            v = torch.cat([cache.v[i], v], dim=1)
        cache.k[i], cache.v[i] = k, v
        rep = Hq // Hk
        kk = k.repeat_interleave(rep, dim=0)                              # (Hq, ctx, D)
        vv = v.repeat_interleave(rep, dim=0)
        att = torch.einsum("nhd,hcd->hnc", q, kk) * (D ** -0.5)
        ctx = kk.shape[1]
        causal = torch.arange(ctx)[None, :] <= (past + torch.arange(n))[:, None]
        att = att.masked_fill(~causal[None], float("-inf"))
        att = torch.softmax(att, dim=-1)
        o = torch.einsum("hnc,hcd->nhd", att, vv).reshape(n, Hq * D)
        h = r + F.linear(o, P[p + "self_attn.o_proj.weight"])
        r = h
        x = rms_norm(h, P[p + "post_attention_layernorm.weight"], cfg.rms_eps)
        g = F.linear(x, P[p + "mlp.gate_proj.weight"])
        u = F.linear(x, P[p + "mlp.up_proj.weight"])
        h = r + F.linear(F.silu(g) * u, P[p + "mlp.down_proj.weight"])
    return rms_norm(h, P[L + "norm.weight"], cfg.rms_eps)[None]


def greedy_id(logp: torch.Tensor, ignore_eos: bool, cfg: LlmCfg) -> int:
    """The greedy rule the build adopts: argmax; while eos is
    forbidden (i < min_len) the argmax is taken over the real speech tokens
    only.  Ties -> lowest index.  The reference's sampling_ids
    (llm/llm.py:149-164) has no deterministic rule of its own: it re-draws up to
    100 times while the draw is >= speech_token_size."""
    if ignore_eos:
        return int(torch.argmax(logp[: cfg.speech_tokens]))
    return int(torch.argmax(logp))


def inv_cdf(p, u: float) -> int:
    """The draw that stands in for `torch.multinomial(p, 1)`: torch draws from its global generator, which
    no other implementation can follow, so the draw is DEFINED (here, in the fixture mint, which patches
    Tensor.multinomial with this function, and in csrc/llm.hip:sample_ras_k) as the inverse CDF at a supplied
    uniform u in [0, 1).  Weights are summed in float64, in index order, in chunks of ceil(n/256): chunk sums,
    a running sum over the chunks, then a running sum inside the chosen chunk started from the total of the
    chunks before it; the sample is the first index whose running sum exceeds u * total."""
    import numpy as np
    w = np.asarray(p, dtype=np.float32).astype(np.float64).reshape(-1)
    n = w.size
    chunk = (n + 255) // 256
    cs = np.zeros(256, dtype=np.float64)
    for t in range(256):
        seg = w[t * chunk: min((t + 1) * chunk, n)]
        a = np.float64(0.0)
        for v in seg:                                     # sequential, as one GPU thread does it
            a = a + v
        cs[t] = a
    tot = np.float64(0.0)
    for t in range(256):
        tot = tot + cs[t]
    target = np.float64(np.float32(u)) * tot
    c, base, tc = np.float64(0.0), np.float64(0.0), 255
    for t in range(256):
        base = c
        c = c + cs[t]
        if c > target:
            tc = t
            break
    pick = min(tc * chunk + chunk - 1, n - 1)
    wsum = base
    for k in range(chunk):
        i = tc * chunk + k
        if i >= n:
            break
        wsum = wsum + w[i]
        if wsum > target:
            pick = i
            break
    return int(pick)


class UniformStream:
    """The supplied uniforms, consumed one per multinomial draw."""

    def __init__(self, values):
        self.values, self.pos = values, 0

    def next(self) -> float:
        v = float(self.values[self.pos])
        self.pos += 1
        return v


def nucleus_sampling(logp: torch.Tensor, us: UniformStream, top_p: float = 0.8, top_k: int = 25) -> int:
    """utils/common.py:146-159."""
    prob, indices = [], []
    cum_prob = 0.0
    sorted_value, sorted_idx = logp.softmax(dim=0).sort(descending=True, stable=True)
    for i in range(len(sorted_idx)):
        if cum_prob < top_p and len(prob) < top_k:
            cum_prob += sorted_value[i]
            prob.append(sorted_value[i])
            indices.append(sorted_idx[i])
        else:
            break
    prob = torch.tensor(prob).to(logp)
    return int(indices[inv_cdf(prob.numpy(), us.next())])


def ras_id(logp: torch.Tensor, decoded: List[int], us: UniformStream, top_p=0.8, top_k=25, win_size=10, tau_r=0.1) -> int:
    """ras_sampling, utils/common.py:137-143 (repetition aware sampling): a nucleus draw; when it already occurs in the
    last win_size tokens at least win_size*tau_r times, a draw from the whole softmax replaces it (:160-162)."""
    top = nucleus_sampling(logp, us, top_p, top_k)
    rep_num = int((torch.tensor(decoded[-win_size:]) == top).sum()) if decoded else 0
    if rep_num >= win_size * tau_r:
        top = inv_cdf(logp.softmax(dim=0).numpy(), us.next())
    return int(top)


def sampling_ids(logp: torch.Tensor, decoded: List[int], us: UniformStream, ignore_eos: bool, cfg: LlmCfg) -> int:
    """TransformerLM.sampling_ids, llm/llm.py:149-164."""
    num_trials, max_trials = 0, 100
    while True:
        top = ras_id(logp, decoded, us)
        if (not ignore_eos) or top < cfg.speech_tokens:
            break
        num_trials += 1
        if num_trials > max_trials:
            raise RuntimeError("sampling reaches max_trials {} and still get eos when ignore_eos is True, check your input!".format(max_trials))
    return top


def lm_input(text, prompt_text, prompt_speech_token, P: Params, cfg: LlmCfg):
    """CosyVoice3LM.inference, llm/llm.py:728-744: [sos, embed(prompt_text+text),
    task_id, speech_embedding(prompt_speech_token)] and (min_len, max_len)."""
    ids = torch.cat([prompt_text, text], dim=1).long()
    emb = F.embedding(ids, P[L + "embed_tokens.weight"])
    se = P["speech_embedding.weight"]
    parts = [se[cfg.sos].reshape(1, 1, -1), emb, se[cfg.task_id].reshape(1, 1, -1)]
    if prompt_speech_token.shape[1] != 0:
        parts.append(F.embedding(prompt_speech_token.long(), se))
    n_text = text.shape[1]
    return torch.cat(parts, dim=1), int(n_text * 2), int(n_text * 20)


def inference(text, prompt_text, prompt_speech_token, P: Params, cfg: LlmCfg,
              min_len: Optional[int] = None, max_len: Optional[int] = None,
              logp_out: Optional[list] = None, uniforms=None) -> Iterator[int]:
    """CosyVoice3LM.inference + Qwen2LM.inference_wrapper (HF branch),
    llm/llm.py:713-748 and :511-525, with the greedy rule above, or - when `uniforms` is given - the reference's
    default repetition-aware sampling with its multinomial draws taken from that stream (inv_cdf)."""
    with torch.no_grad():
        x, mn, mx = lm_input(text, prompt_text, prompt_speech_token, P, cfg)
        min_len = mn if min_len is None else min_len
        max_len = mx if max_len is None else max_len
        cache = KVCache(cfg.layers)
        us = UniformStream(uniforms) if uniforms is not None else None
        out_tokens: List[int] = []
        for i in range(max_len):
            y = forward_one_step(x, cache, P, cfg)
            logp = F.linear(y[:, -1], P["llm_decoder.weight"]).log_softmax(dim=-1).squeeze(0)
            if logp_out is not None:
                logp_out.append(logp.clone())
            tid = greedy_id(logp, i < min_len, cfg) if us is None else sampling_ids(logp, out_tokens, us, i < min_len, cfg)
            if tid >= cfg.speech_tokens:                   # stop_token_ids, llm.py:520,667
                break
            yield tid
            out_tokens.append(tid)
            x = P["speech_embedding.weight"][tid].reshape(1, 1, -1)


def silent_filter(tokens) -> List[int]:
    """CosyVoiceModel.llm_job, cli/model.py:101-129: drop a silent/breath token
    once a run of them exceeds 5."""
    out, run = [], 0
    for t in tokens:
        if t in SILENT_TOKENS:
            run += 1
            if run > MAX_SILENT_RUN:
                continue
        else:
            run = 0
        out.append(t)
    return out
