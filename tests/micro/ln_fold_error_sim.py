"""CPU simulation: what does folding LayerNorm-modulate into the consuming product cost in accuracy?

VERDICT r4 item 5: "LN-modulate commutes past the product ... the consuming product uses W' = W (.) (1 + scale) ... y = rstd (acc - mu u) + v
... Measure the estimator error first (un-centred bf16 operand)".  The engine's default mode rounds every GEMM operand to bf16
(fp32 accumulate, fp32 residual stream).  This script runs the oracle's DiT (oracle/flow.py, fp32) three ways on the same
inputs and prints max / mean |err| of the estimator output against the fp32 run:
  today : xn = bf16(LN(h) (1 + s) + b), products with bf16 operands (q | k | v, attention output, GELU output rounded to bf16 too)
  fold  : the consuming products take bf16(h) itself against W' = bf16(W (1 + s)), then r (acc - mu u) + v with u = sum_k W', v = W b
Run on the CPU, nothing here is used by the library:  python tests/micro/ln_fold_error_sim.py [tiny|full] [T]
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fangyan_tts_amd import synth  # noqa: E402
from fangyan_tts_amd.spec import FlowCfg  # noqa: E402
from oracle import flow as o  # noqa: E402

E = o.E


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def block(x, t_emb, am, freqs, P, cfg, i, mode):
    b = E + f"transformer_blocks.{i}."
    emb = F.linear(F.silu(t_emb), P[b + "attn_norm.linear.weight"], P[b + "attn_norm.linear.bias"])
    sh1, sc1, g1, sh2, sc2, g2 = torch.chunk(emb, 6, dim=1)

    def ln_linear(x, scale, shift, W, bias):
        """LN(x) (1 + scale) + shift -> Linear(W, bias), in the three arithmetic models"""
        if mode == "fp32":
            return F.linear(o.layer_norm(x) * (1 + scale[:, None]) + shift[:, None], W, bias)
        if mode == "today":
            return F.linear(bf(o.layer_norm(x) * (1 + scale[:, None]) + shift[:, None]), W, bias)
        # fold: per sequence (scale / shift differ per batch row only through t, equal here) - do it per batch element
        outs = []
        for bi in range(x.shape[0]):
            xb = x[bi]
            mu = xb.mean(dim=-1, keepdim=True)
            r = torch.rsqrt(xb.var(dim=-1, unbiased=False, keepdim=True) + 1e-6)
            Wp = bf(W * (1 + scale[bi])[None, :])                      # W'[n][k] = W[n][k] (1 + s[k]), stored bf16
            u = Wp.sum(dim=1)                                        # of the ROUNDED W': the mean term cancels exactly
            v = W @ shift[bi]
            acc = bf(xb) @ Wp.t()
            outs.append(r * (acc - mu * u[None, :]) + v[None, :] + bias[None, :])
        return torch.stack(outs)
    Wqkv = torch.cat([P[b + "attn.to_q.weight"], P[b + "attn.to_k.weight"], P[b + "attn.to_v.weight"]])
    bqkv = torch.cat([P[b + "attn.to_q.bias"], P[b + "attn.to_k.bias"], P[b + "attn.to_v.bias"]])
    qkv = ln_linear(x, sc1, sh1, Wqkv, bqkv)
    D = cfg.dim
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    q, k = o.apply_rope(q, freqs), o.apply_rope(k, freqs)
    if mode != "fp32":
        q, k, v = bf(q), bf(k), bf(v)
    B, T, _ = q.shape
    H, Dh = cfg.heads, cfg.head_dim
    q, k, v = (z.reshape(B, T, H, Dh).transpose(1, 2) for z in (q, k, v))
    a = F.scaled_dot_product_attention(q, k, v, attn_mask=am)
    a = a.transpose(1, 2).reshape(B, T, H * Dh)
    if mode != "fp32":
        a = bf(a)
    a = F.linear(a, P[b + "attn.to_out.0.weight"], P[b + "attn.to_out.0.bias"])
    x = x + g1.unsqueeze(1) * a
    h = ln_linear(x, sc2, sh2, P[b + "ff.ff.0.0.weight"], P[b + "ff.ff.0.0.bias"])
    h = F.gelu(h, approximate="tanh")
    if mode != "fp32":
        h = bf(h)
    h = F.linear(h, P[b + "ff.ff.2.weight"], P[b + "ff.ff.2.bias"])
    return x + g2.unsqueeze(1) * h


def forward(x, mask, mu, t, spks, cond, P, cfg, mode):
    x, mu, cond = x.transpose(1, 2), mu.transpose(1, 2), cond.transpose(1, 2)
    T = x.shape[1]
    t_emb = o.timestep_embedding(t, P)
    h = o.input_embed(x, cond, mu, spks, P, cfg)
    freqs = o.rope_freqs(T, cfg.head_dim)
    am = o.chunk_attn_mask(mask, T, 0)
    stats = []
    for i in range(cfg.depth):
        if mode == "fp32":
            stats.append(float((h.mean(-1).abs() / h.std(-1)).max()))
        h = block(h, t_emb, am, freqs, P, cfg, i, mode)
    emb = F.linear(F.silu(t_emb), P[E + "norm_out.linear.weight"], P[E + "norm_out.linear.bias"])
    scale, shift = torch.chunk(emb, 2, dim=1)
    hn = o.layer_norm(h) * (1 + scale)[:, None, :] + shift[:, None, :]
    if mode != "fp32":
        hn = bf(hn)
    return F.linear(hn, P[E + "proj_out.weight"], P[E + "proj_out.bias"]).transpose(1, 2), stats


def main():
    size = sys.argv[1] if len(sys.argv) > 1 else "full"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    cfg = FlowCfg.tiny() if size == "tiny" else FlowCfg()
    for general in (False, True):
        prev = synth.set_weight_rounding(not general)
        P = {k: torch.from_numpy(v) for k, v in synth.state_dict(cfg.manifest()).items()}
        synth.set_weight_rounding(prev)
        x = torch.from_numpy(synth.normal(f"in.dit.x.{T}", (2, 80, T)))
        mu = torch.from_numpy(synth.normal(f"in.dit.mu.{T}", (2, 80, T)))
        cond = torch.from_numpy(synth.normal(f"in.dit.cond.{T}", (2, 80, T)))
        spks = torch.from_numpy(synth.normal(f"in.dit.spks.{T}", (2, 80)))
        t = torch.tensor([0.3, 0.3])
        mask = torch.ones(2, 1, T)
        with torch.no_grad():
            ref, stats = forward(x, mask, mu, t, spks, cond, P, cfg, "fp32")
            print(f"{size} T={T} {'general fp32' if general else 'bf16-exact'} weights: max over blocks and rows of |row mean| / row std of the residual stream: {max(stats):.3f}")
            for mode in ("today", "fold"):
                if general and mode == "today":
                    P2 = {k: (bf(v) if v.dim() >= 2 else v) for k, v in P.items()}
                    y, _ = forward(x, mask, mu, t, spks, cond, P2, cfg, mode)
                else:
                    y, _ = forward(x, mask, mu, t, spks, cond, P, cfg, mode)
                e = (y - ref).abs()
                print(f"   {mode:6s}: max |err| {float(e.max()):.3e}   mean |err| {float(e.mean()):.3e}   (output std {float(ref.std()):.2f})")


if __name__ == "__main__":
    main()
