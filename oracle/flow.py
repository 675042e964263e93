"""ORACLE (test infrastructure, not product code) - CPU restatement of the
CosyVoice3 acoustic decoder: CausalMaskedDiffWithDiT.inference ->
CausalConditionalCFM (Euler + classifier-free guidance) -> DiT estimator.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it.

Pinned against the reference itself by tests/golden/mint_goldens.py (fixtures
tests/golden/flow_*.npz).  One piece of arithmetic lives in an un-vendored
dependency, x-transformers==2.11.24 (RotaryEmbedding / apply_rotary_pos_emb,
imported at flow/DiT/dit.py:17 and modules.py:20): its published algorithm is
restated in `rope_freqs` / `apply_rope` below and the SAME restatement is what
the mint script installs for the reference import, so every DiT golden is
"conditional on a15"  - that sub-step's parity is unpinned.

fp32 torch CPU; tensors laid out as in the reference.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from fangyan_tts_amd.spec import FlowCfg

Params = Dict[str, torch.Tensor]
E = "decoder.estimator."


def prepare(sd: Dict[str, np.ndarray]) -> Params:
    return {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in sd.items()}


# ---- pieces outside the estimator ------------------------------------------

def prelookahead(x, P: Params, cfg: FlowCfg, context=None):
    """PreLookaheadLayer.forward, transformer/upsample_encoder.py:82-103.
    x (B, N, 80) -> (B, N, 80); `context` (B, pre_lookahead, 80) replaces the zero
    look-ahead of a streaming chunk."""
    h = x.transpose(1, 2)
    if context is None:
        h = F.pad(h, (0, cfg.pre_lookahead))
    else:
        assert context.shape[1] == cfg.pre_lookahead
        h = torch.cat([h, context.transpose(1, 2)], dim=2)
    h = F.leaky_relu(F.conv1d(h, P["pre_lookahead_layer.conv1.weight"], P["pre_lookahead_layer.conv1.bias"]))
    h = F.pad(h, (2, 0))
    h = F.conv1d(h, P["pre_lookahead_layer.conv2.weight"], P["pre_lookahead_layer.conv2.bias"])
    return h.transpose(1, 2) + x


def t_span(cfg: FlowCfg) -> torch.Tensor:
    """flow_matching.py:223-225: cosine schedule over n_timesteps+1 points."""
    t = torch.linspace(0, 1, cfg.n_timesteps + 1, dtype=torch.float32)
    return 1 - torch.cos(t * 0.5 * torch.pi)


def euler_times(cfg: FlowCfg):
    """The (t, dt) pair each Euler step uses, with the reference's running
    accumulation t += dt; dt = t_span[k+1] - t (flow_matching.py:87-88,118-122)."""
    ts = t_span(cfg)
    t, dt = ts[0], ts[1] - ts[0]
    out = []
    for step in range(1, len(ts)):
        out.append((t.clone(), dt.clone()))
        t = t + dt
        if step < len(ts) - 1:
            dt = ts[step + 1] - t
    return out


# ---- DiT estimator ----------------------------------------------------------

def timestep_embedding(t, P: Params):
    """TimestepEmbedding + SinusPositionEmbedding(256), DiT/modules.py:71-83,606-616."""
    half = 128
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half).float() * -emb)
    emb = 1000 * t.unsqueeze(1) * emb.unsqueeze(0)
    emb = torch.cat((emb.sin(), emb.cos()), dim=-1)
    h = F.linear(emb, P[E + "time_embed.time_mlp.0.weight"], P[E + "time_embed.time_mlp.0.bias"])
    return F.linear(F.silu(h), P[E + "time_embed.time_mlp.2.weight"], P[E + "time_embed.time_mlp.2.bias"])


def conv_pos_embed(x, P: Params, cfg: FlowCfg):
    """CausalConvPositionEmbedding.forward (mask=None), DiT/modules.py:129-144."""
    k, g = cfg.conv_pos_k, cfg.conv_pos_groups
    h = x.permute(0, 2, 1)
    for c in ("conv1", "conv2"):
        h = F.pad(h, (k - 1, 0))
        h = F.mish(F.conv1d(h, P[E + f"input_embed.conv_pos_embed.{c}.0.weight"],
                            P[E + f"input_embed.conv_pos_embed.{c}.0.bias"], groups=g))
    return h.permute(0, 2, 1)


def input_embed(x, cond, mu, spks, P: Params, cfg: FlowCfg):
    """InputEmbedding.forward, DiT/dit.py:84-98: cat order [x, cond, mu, spks]."""
    T = x.shape[1]
    cat = torch.cat([x, cond, mu, spks[:, None, :].expand(-1, T, -1)], dim=-1)
    h = F.linear(cat, P[E + "input_embed.proj.weight"], P[E + "input_embed.proj.bias"])
    return conv_pos_embed(h, P, cfg) + h


def rope_freqs(T: int, dim: int = 64) -> torch.Tensor:
    """x-transformers 2.x RotaryEmbedding(dim).forward_from_seq_len(T): (1, T, dim),
    freqs[p] = interleave(p * inv_freq, p * inv_freq), inv_freq = 10000^(-2j/dim)."""
    inv = 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))
    f = torch.einsum("i,j->ij", torch.arange(T).float(), inv)
    return torch.stack((f, f), dim=-1).reshape(1, T, dim)


def apply_rope(t, freqs):
    """x-transformers 2.x apply_rotary_pos_emb(t, freqs, scale=1): only the first
    freqs.shape[-1] channels are rotated, pairs are interleaved (2i, 2i+1)."""
    rot = freqs.shape[-1]
    a, rest = t[..., :rot], t[..., rot:]
    x = a.reshape(*a.shape[:-1], rot // 2, 2)
    x1, x2 = x.unbind(dim=-1)
    rh = torch.stack((-x2, x1), dim=-1).reshape(a.shape)
    a = a * freqs.cos() + rh * freqs.sin()
    return torch.cat((a, rest), dim=-1)


def layer_norm(x):
    return F.layer_norm(x, (x.shape[-1],), eps=1e-6)


def dit_block(x, t_emb, attn_mask, freqs, P: Params, cfg: FlowCfg, i: int):
    """DiTBlock.forward, DiT/modules.py:516-530, AttnProcessor :349-407."""
    b = E + f"transformer_blocks.{i}."
    emb = F.linear(F.silu(t_emb), P[b + "attn_norm.linear.weight"], P[b + "attn_norm.linear.bias"])
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = torch.chunk(emb, 6, dim=1)
    h = layer_norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
    q = F.linear(h, P[b + "attn.to_q.weight"], P[b + "attn.to_q.bias"])
    k = F.linear(h, P[b + "attn.to_k.weight"], P[b + "attn.to_k.bias"])
    v = F.linear(h, P[b + "attn.to_v.weight"], P[b + "attn.to_v.bias"])
    q, k = apply_rope(q, freqs), apply_rope(k, freqs)          # before the head split: head 0 only
    B, T, _ = q.shape
    H, D = cfg.heads, cfg.head_dim
    q, k, v = (z.view(B, T, H, D).transpose(1, 2) for z in (q, k, v))
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask)
    o = o.transpose(1, 2).reshape(B, T, H * D)
    o = F.linear(o, P[b + "attn.to_out.0.weight"], P[b + "attn.to_out.0.bias"])
    row_mask = attn_mask[:, 0, -1].unsqueeze(-1)               # modules.py:401-405
    o = o.masked_fill(~row_mask, 0.0)
    x = x + gate_msa.unsqueeze(1) * o
    h = layer_norm(x) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
    h = F.linear(h, P[b + "ff.ff.0.0.weight"], P[b + "ff.ff.0.0.bias"])
    h = F.gelu(h, approximate="tanh")
    h = F.linear(h, P[b + "ff.ff.2.weight"], P[b + "ff.ff.2.bias"])
    return x + gate_mlp.unsqueeze(1) * h


def chunk_attn_mask(mask, T: int, chunk: int):
    """add_optional_chunk_mask as DiT.forward calls it, utils/mask.py:161-236 with
    dit.py:163-166: non-streaming -> key-padding mask repeated over rows;
    streaming -> additionally col < (row//chunk + 1)*chunk (subsequent_chunk_mask,
    mask.py:127-158); rows left with no true entry become all-true."""
    m = mask.bool()                                            # (B, 1, T)
    if chunk <= 0:
        return m.repeat(1, T, 1).unsqueeze(1)
    r = torch.arange(T)
    cm = r[None, :] < ((r[:, None] // chunk) + 1) * chunk      # (T, T)
    am = m & cm[None]
    empty = am.sum(dim=-1) == 0
    am = am.masked_fill(empty[..., None], True)
    return am.unsqueeze(1)


def dit_forward(x, mask, mu, t, spks, cond, P: Params, cfg: FlowCfg, streaming: bool = False):
    """DiT.forward, DiT/dit.py:145-176.  x, mu, cond (B, 80, T); mask (B, 1, T);
    t (B,); spks (B, 80) -> (B, 80, T)."""
    x, mu, cond = x.transpose(1, 2), mu.transpose(1, 2), cond.transpose(1, 2)
    T = x.shape[1]
    t_emb = timestep_embedding(t, P)
    h = input_embed(x, cond, mu, spks, P, cfg)
    freqs = rope_freqs(T, cfg.head_dim)
    am = chunk_attn_mask(mask, T, cfg.static_chunk if streaming else 0)
    for i in range(cfg.depth):
        h = dit_block(h, t_emb, am, freqs, P, cfg, i)
    emb = F.linear(F.silu(t_emb), P[E + "norm_out.linear.weight"], P[E + "norm_out.linear.bias"])
    scale, shift = torch.chunk(emb, 2, dim=1)                  # note order, modules.py:261
    h = layer_norm(h) * (1 + scale)[:, None, :] + shift[:, None, :]
    return F.linear(h, P[E + "proj_out.weight"], P[E + "proj_out.bias"]).transpose(1, 2)


# ---- CFM solver and the flow front --------------------------------------------

def solve_euler(z, mu, mask, spks, cond, P: Params, cfg: FlowCfg, streaming: bool = False):
    """ConditionalCFM.solve_euler, flow/flow_matching.py:71-124 (batch-2 CFG)."""
    x = z
    T = x.shape[2]
    for t, dt in euler_times(cfg):
        x_in = torch.cat([x, x], dim=0)
        mask_in = torch.cat([mask, mask], dim=0)
        mu_in = torch.cat([mu, torch.zeros_like(mu)], dim=0)
        t_in = t.reshape(1).repeat(2)
        spks_in = torch.cat([spks, torch.zeros_like(spks)], dim=0)
        cond_in = torch.cat([cond, torch.zeros_like(cond)], dim=0)
        d = dit_forward(x_in, mask_in, mu_in, t_in, spks_in, cond_in, P, cfg, streaming)
        v = (1.0 + cfg.cfg_rate) * d[:1] - cfg.cfg_rate * d[1:]
        x = x + dt * v
    return x.float()


def flow_front(token, prompt_token, prompt_feat, embedding, P: Params, cfg: FlowCfg, finalize: bool = True):
    """flow.py:370-390: everything before the CFM solver.
    Returns mu (1, 80, T), spks (1, 80), cond (1, 80, T), P_mel."""
    emb = F.normalize(embedding, dim=1)
    spks = F.linear(emb, P["spk_embed_affine_layer.weight"], P["spk_embed_affine_layer.bias"])
    tok = torch.cat([prompt_token, token], dim=1).long()
    h = F.embedding(torch.clamp(tok, min=0), P["input_embedding.weight"])      # mask is all ones for B=1
    if finalize:
        h = prelookahead(h, P, cfg)
    else:                                                                      # flow.py:382-383
        h = prelookahead(h[:, :-cfg.pre_lookahead], P, cfg, context=h[:, -cfg.pre_lookahead:])
    h = h.repeat_interleave(2, dim=1)
    p_mel = prompt_feat.shape[1]
    T = h.shape[1]
    cond = torch.zeros(1, T, cfg.mel)
    cond[:, :p_mel] = prompt_feat
    return h.transpose(1, 2).contiguous(), spks, cond.transpose(1, 2).contiguous(), p_mel


def inference(token, prompt_token, prompt_feat, embedding, P: Params, cfg: FlowCfg, rand_noise,
              streaming: bool = False, finalize: bool = True):
    """CausalMaskedDiffWithDiT.inference, flow/flow.py:358-403.
    token (1, n) int, prompt_token (1, P_tok) int, prompt_feat (1, P_mel, 80),
    embedding (1, 192), rand_noise (1, 80, >=T) -> mel (1, 80, 2n), or
    (1, 80, 2(n - pre_lookahead)) with finalize=False."""
    with torch.no_grad():
        mu, spks, cond, p_mel = flow_front(token, prompt_token, prompt_feat, embedding, P, cfg, finalize)
        T = mu.shape[2]
        mask = torch.ones(1, 1, T)
        z = rand_noise[:, :, :T]
        mel = solve_euler(z, mu, mask, spks, cond, P, cfg, streaming)
        return mel[:, :, p_mel:].float()
