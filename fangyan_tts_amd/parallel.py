"""Data-parallel sharding of independent utterances over the GPUs of one node.

Utterances share nothing (own KV cache, ODE state, vocoder), so the path shards with no
data-path collective; the single exchange is the all-gather of the finished audio
(SURVEY 8e).  `torch.distributed` with backend "nccl" is RCCL over xGMI on ROCm; the same
code runs on gloo for the CPU tests.

Which utterances a rank gets: `plan_shards` sorts them by expected length and cuts the sorted
order into batches, so a batch pads little (the LM decodes a batch for as many steps as its
longest member needs; SURVEY 8e: "sorted by expected length to balance; LLM step count dominates
imbalance"), then deals the batches to the ranks longest-first onto the least loaded rank.  The
reference's own multi-GPU inference hands each rank a strided share of the dataset and lets every
rank write its own files (CosyVoice/runtime/triton_trtllm/offline_inference.py:312-322, its
DistributedSampler); north_star asks for the audio on every rank instead, hence `AudioGather`.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous block of utterance indices owned by `rank` (sizes differ by at most one): the plan when nothing is known
    about the utterances' lengths."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def plan_shards(costs: Sequence[float], world: int, batch: Optional[int] = None) -> List[List[List[int]]]:
    """Length-sorted sharding.  costs[i] = expected length of utterance i (text ids, expected speech tokens - anything
    monotone in the LM's step count).  Returns plan[rank] = that rank's batches, each a list of utterance indices.

    * utterances are sorted by cost, longest first (ties keep the input order), and the sorted order is cut into batches of
      `batch`: the members of a batch have neighbouring lengths, so the steps a batch decodes for its longest member are
      wasted on few rows.  Default: ceil(n / world) rounded DOWN to batches of at most 8 (the engines' batch) with at least two
      batches per rank when there are that many utterances - one contiguous block of the sorted order per rank would hand
      rank 0 all the longest utterances (flow and vocoder work follow the SUM of lengths) and could balance nothing;
    * a batch costs max(cost) * size (what the padded decode runs); batches go, most expensive first, to the rank with the
      least cost so far (ties: the lowest rank) - the longest-processing-time rule, deterministic, computed identically on
      every rank from the same costs with no communication."""
    n = len(costs)
    assert world >= 1
    if batch is None:
        batch = max(1, min(8, -(-n // (2 * world))))
    order = sorted(range(n), key=lambda i: (-float(costs[i]), i))
    batches = [order[o: o + batch] for o in range(0, n, batch)]
    weight = [max(float(costs[i]) for i in b) * len(b) for b in batches]
    plan: List[List[List[int]]] = [[] for _ in range(world)]
    load = [0.0] * world
    count = [0] * world
    for k in sorted(range(len(batches)), key=lambda k: (-weight[k], k)):
        # least accumulated cost; among equals the rank with the fewest batches, then the lowest rank (equal-cost batches
        # spread one per rank instead of piling up on rank 0)
        r = min(range(world), key=lambda r: (load[r], count[r], r))
        plan[r].append(batches[k])
        load[r] += weight[k]
        count[r] += 1
    return plan


def plan_order(plan: Sequence[Sequence[Sequence[int]]]) -> List[int]:
    """The utterance index of every row of the gathered result, rank by rank, batch by batch (the order `AudioGather`
    returns rows in when each rank passes its batches in plan order)."""
    return [i for rank in plan for b in rank for i in b]


def unshard(rows_by_rank: Sequence[Sequence], plan: Sequence[Sequence[Sequence[int]]]) -> list:
    """rows_by_rank[r] = the per-utterance results of rank r in the order of its plan -> the results in input order (the
    inverse of the permutation `plan_shards` applied)."""
    order = [[i for b in rank for i in b] for rank in plan]
    n = sum(len(o) for o in order)
    out = [None] * n
    for r, idx in enumerate(order):
        assert len(rows_by_rank[r]) == len(idx), (r, len(rows_by_rank[r]), len(idx))
        for j, i in enumerate(idx):
            out[i] = rows_by_rank[r][j]
    return out


class AudioGather:
    """The one exchange of the data-parallel path with everything allocated ONCE: every rank's b <= b_max finished utterances
    (wav (b, S <= s_max) fp32 + their valid lengths) to every rank in ONE `all_gather_into_tensor` of a fixed-size record -
    b_max rows of s_max samples, then the count and the lengths bit-cast to float behind them.  On the GPU the record is
    packed by the library's kernel with the lengths in its arguments (`fy_audio_record_pack`): no allocation, no host-to-device
    copy and no stream synchronisation per step; the caller's first look at the lengths is the one device-to-host read of
    the gathered headers.

    With one rank the returned audio is a VIEW of this object's preallocated `out` buffer (with more ranks the rows are copied out
    from between the headers): the next call overwrites it.  A caller that keeps a result across calls clones it (gather_audio
    does, unless told `reuse=True`)."""

    def __init__(self, b_max: int, s_max: int, device, group=None):
        self.b_max, self.s_max, self.group = int(b_max), int(s_max), group
        self.world = dist.get_world_size(group)
        self.device = torch.device(device)
        self.rec = self.b_max * self.s_max + self.b_max + 1
        self.mine = torch.zeros(self.rec, dtype=torch.float32, device=self.device)
        self.out = torch.empty(self.world * self.rec, dtype=torch.float32, device=self.device)
        self._head = torch.zeros(self.b_max + 1, dtype=torch.int32)            # CPU tensors only (gloo tests)

    def __call__(self, wav: torch.Tensor, n_samples: Sequence[int]) -> Tuple[torch.Tensor, List[List[int]]]:
        b, S = wav.shape
        per = self.b_max * self.s_max
        assert b <= self.b_max and S <= self.s_max and len(n_samples) == b and wav.dtype == torch.float32 and wav.device == self.device
        if wav.is_cuda:
            from . import _lib
            assert wav.stride(1) == 1
            _lib.check(_lib.lib().fy_audio_record_pack(wav.data_ptr(), wav.stride(0), _lib.int_array(n_samples), b, self.b_max, self.s_max,
                                                       self.mine.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream))
        else:
            rows = self.mine[:per].view(self.b_max, self.s_max)
            rows.zero_()
            self._head.zero_()
            self._head[0] = b
            for j, n in enumerate(n_samples):
                k = min(int(n), self.s_max)
                rows[j, :k] = wav[j, :k]
                self._head[1 + j] = k
            self.mine[per:] = self._head.view(torch.float32)
        dist.all_gather_into_tensor(self.out, self.mine, group=self.group)
        out = self.out.view(self.world, self.rec)
        heads = out[:, per:].contiguous().view(torch.int32).cpu()      # the caller's first look at the result
        per_rank = [heads[r, 1: 1 + int(heads[r, 0])].tolist() for r in range(self.world)]
        return out[:, :per].reshape(self.world * self.b_max, self.s_max), per_rank


# one AudioGather per (bounds, device, group): keyed by the group OBJECT through a WeakKeyDictionary (an id() can be recycled by a
# new group after the old one was destroyed; the entry of a destroyed group goes with it), the default group under a plain key
import weakref

_gathers_by_group: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
_gathers_default = {}


def _gather_cache(group):
    if group is None:
        # the default group can be destroyed and re-created: entries made for an earlier one are dropped
        cur = dist.distributed_c10d._get_default_group() if dist.is_initialized() else None
        if _gathers_default.get("group") is not cur:
            _gathers_default.clear()
            _gathers_default["group"] = cur
        return _gathers_default
    try:
        return _gathers_by_group.setdefault(group, {})
    except TypeError:                      # a group object that cannot be weakly referenced: no caching
        return {}


def gather_audio(wav: torch.Tensor, n_samples: Sequence[int], group=None, b_cap: int = 64,
                 b_max: int = 0, s_max: int = 0, reuse: bool = False) -> Tuple[torch.Tensor, List[List[int]]]:
    """wav (b, S) of this rank's utterances, n_samples their valid lengths ->
    (all wavs (world * b_max, S_max) in rank order, per-rank length lists) on every rank.

    With the bounds known to every rank up front (b_max utterances per rank, s_max samples: what the engines were created
    for) the exchange is ONE collective on buffers made once per (bounds, device, group) - `AudioGather`; the result is a fresh
    tensor (a clone of the gather's buffer) unless `reuse=True`, which hands out the buffer itself - valid until the next call with
    the same bounds (a step loop that consumes each result before the next step).  Without them: one tiny all-gather of
    (b, S, lengths) in a fixed-size record (b <= b_cap), sizes read on the host, then one fused all-gather of the padded audio."""
    world = dist.get_world_size(group)
    dev = wav.device
    b, S = wav.shape
    assert b <= b_cap and len(n_samples) == b, "more utterances per rank than the header record holds"
    if b_max and s_max:
        cache = _gather_cache(group)
        key = (b_max, s_max, str(dev), world)
        g = cache.get(key)
        if g is None:
            g = cache[key] = AudioGather(b_max, s_max, dev, group)
        out, per_rank = g(wav, n_samples)
        # with more than one rank the rows come out of the record buffer by a copy already (the headers sit between the ranks' audio);
        # only a result that still shares the buffer's storage (world 1) needs the clone
        shares = out.untyped_storage().data_ptr() == g.out.untyped_storage().data_ptr()
        return (out.clone() if shares and not reuse else out), per_rank
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
