"""Data-parallel sharding of independent utterances over the GPUs of one node.

Utterances share nothing (own KV cache, ODE state, vocoder), so the path shards with no
data-path collective; the single exchange is the all-gather of the finished audio
(SURVEY 8e).  `torch.distributed` with backend "nccl" is RCCL over xGMI on ROCm; the same
code runs on gloo for the CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous block of utterance indices owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def gather_audio(wav: torch.Tensor, n_samples: Sequence[int], group=None, b_cap: int = 64,
                 b_max: int = 0, s_max: int = 0) -> Tuple[torch.Tensor, List[List[int]]]:
    """wav (b, S) of this rank's utterances, n_samples their valid lengths ->
    (all wavs (world * b_max, S_max) in rank order, per-rank length lists) on every rank.

    With the bounds known to every rank up front (b_max utterances per rank, s_max samples: what the engines were created
    for) the exchange is ONE collective and nothing returns to the host in between: the lengths travel bit-cast behind
    the audio in the same fixed-size record.  Without them: one tiny all-gather of (b, S, lengths) in a fixed-size record
    (b <= b_cap), sizes read on the host, then one fused all-gather of the padded audio."""
    world = dist.get_world_size(group)
    dev = wav.device
    b, S = wav.shape
    assert b <= b_cap and len(n_samples) == b, "more utterances per rank than the header record holds"
    if b_max and s_max:
        assert b <= b_max and S <= s_max and wav.dtype == torch.float32
        rec = b_max * s_max + b_max + 1
        mine = torch.zeros(rec, dtype=torch.float32, device=dev)
        mine[: b_max * s_max].view(b_max, s_max)[:b, :S] = wav
        head = torch.zeros(b_max + 1, dtype=torch.int32)
        head[0] = b
        head[1: 1 + b] = torch.as_tensor(list(n_samples), dtype=torch.int32)
        mine[b_max * s_max:] = head.view(torch.float32).to(dev, non_blocking=True)
        out = torch.empty(world * rec, dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(out, mine, group=group)
        out = out.view(world, rec)
        heads = out[:, b_max * s_max:].contiguous().view(torch.int32).cpu()      # the caller's first look at the result
        per_rank = [heads[r, 1: 1 + int(heads[r, 0])].tolist() for r in range(world)]
        return out[:, : b_max * s_max].reshape(world * b_max, s_max), per_rank
    rec = 2 + b_cap
    head = torch.zeros(rec, dtype=torch.int64)
    head[0], head[1] = b, S
    head[2: 2 + b] = torch.as_tensor(list(n_samples), dtype=torch.int64)
    heads = torch.zeros(world * rec, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(heads, head.to(dev), group=group)
    heads = heads.cpu().reshape(world, rec)
    b_max, s_max = int(heads[:, 0].max()), int(heads[:, 1].max())
    if (b, S) != (b_max, s_max):
        pad = torch.zeros(b_max, s_max, dtype=wav.dtype, device=dev)
        pad[:b, :S] = wav
        wav = pad
    out = torch.empty(world * b_max, s_max, dtype=wav.dtype, device=dev)
    dist.all_gather_into_tensor(out, wav.contiguous(), group=group)
    per_rank = [heads[r, 2: 2 + int(heads[r, 0])].tolist() for r in range(world)]
    return out, per_rank
