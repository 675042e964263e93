"""CPU tests of the N > 1 path: utterance sharding and the audio all-gather on a
world_size-2 gloo group (the GPU run uses the same code on RCCL)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fangyan_tts_amd.parallel import gather_audio, plan_order, plan_shards, shard_range, unshard


def test_shard_range_partitions():
    for n in (1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            got = [i for r in range(w) for i in shard_range(n, r, w)]
            assert got == list(range(n))
            sizes = [len(shard_range(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_utts = 5
    mine = shard_range(n_utts, rank, world)
    lens = [1000 + 37 * i for i in mine]
    S = max(lens) + (3 if rank == 0 else 11)          # ranks pad differently
    wav = torch.zeros(len(mine), S)
    for j, i in enumerate(mine):
        wav[j, : lens[j]] = torch.arange(lens[j], dtype=torch.float32) + 10000.0 * i
    ok = True
    b_max = max(len(shard_range(n_utts, r, world)) for r in range(world))
    for fixed in (False, True):                       # header + audio in two collectives, or one fixed-size record
        out, per_rank = gather_audio(wav, lens, b_max=b_max, s_max=1400) if fixed else gather_audio(wav, lens)
        ok &= _check(out, per_rank, n_utts, world, b_max)
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def _check(out, per_rank, n_utts, world, b_max):
    ok = True
    for r in range(world):
        for j, i in enumerate(shard_range(n_utts, r, world)):
            n = per_rank[r][j]
            row = out[r * b_max + j]
            ok &= n == 1000 + 37 * i
            ok &= bool(torch.equal(row[:n], torch.arange(n, dtype=torch.float32) + 10000.0 * i))
            ok &= float(row[n:].abs().max()) == 0.0
    return ok


def test_gather_audio_gloo_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))


def test_plan_shards_is_a_balanced_partition():
    """Length-sorted sharding: every utterance exactly once; a batch holds neighbours of the sorted order; the
    ranks' padded costs differ by at most one batch; the plan is a pure function of the costs (every rank computes the same)."""
    import random
    rnd = random.Random(7)
    for n, world, batch in ((64, 8, 8), (64, 8, None), (61, 8, 8), (5, 2, None), (3, 4, None), (128, 8, 8), (1, 1, None)):
        costs = [rnd.randint(10, 40) for _ in range(n)]
        plan = plan_shards(costs, world, batch)
        assert plan == plan_shards(list(costs), world, batch)
        flat = plan_order(plan)
        assert sorted(flat) == list(range(n))
        bsz = batch or max(1, min(8, -(-n // (2 * world))))
        for rank in plan:
            for b in rank:
                assert 1 <= len(b) <= bsz
        batches = [b for rank in plan for b in rank]
        # neighbours in the sorted order: the ranges of two batches do not interleave
        spans = sorted((min(costs[i] for i in b), max(costs[i] for i in b)) for b in batches)
        for (lo0, hi0), (lo1, hi1) in zip(spans, spans[1:]):
            assert hi0 <= lo1 or (lo0 == lo1)
        load = [sum(max(costs[i] for i in b) * len(b) for b in rank) for rank in plan]
        biggest = max(max(costs[i] for i in b) * len(b) for b in batches)
        assert max(load) - min(load) <= biggest
        counts = [len(rank) for rank in plan]
        assert max(counts) - min(counts) <= 1 or batch is None
        # against contiguous blocks in input order: the padded work (sum over batches of max x size) never grows
        naive = 0
        for r in range(world):
            idx = list(shard_range(n, r, world))
            for o in range(0, len(idx), bsz):
                blk = idx[o: o + bsz]
                naive += max(costs[i] for i in blk) * len(blk)
        assert sum(load) <= naive
        # against the strided share the reference's DistributedSampler gives a rank (offline_inference.py:312-322), cut into batches of
        # the same size in that order: the most loaded rank is never worse off (flow / vocoder work follows the sum of lengths, the LM's
        # the padded sum)
        strided_pad, strided_sum = [], []
        for r in range(world):
            idx = list(range(r, n, world))
            strided_pad.append(sum(max(costs[i] for i in idx[o: o + bsz]) * len(idx[o: o + bsz]) for o in range(0, len(idx), bsz)))
            strided_sum.append(sum(costs[i] for i in idx))
        if n >= 2 * world:
            assert max(load) <= max(strided_pad) + biggest // 2, (n, world, batch, load, strided_pad)
            sums = [sum(costs[i] for b in rank for i in b) for rank in plan]
            assert max(sums) - min(sums) <= biggest, (sums, strided_sum)
    # unshard is the inverse permutation
    costs = [5, 9, 1, 7, 3, 8, 2]
    plan = plan_shards(costs, 3, 2)
    rows = [[("utt", i) for b in rank for i in b] for rank in plan]
    assert unshard(rows, plan) == [("utt", i) for i in range(len(costs))]


def _worker_sorted(rank, world, port, ret):
    """Ragged utterances, length-sorted plan, two steps through ONE preallocated gather; rows go back to input order."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_utts, batch = 11, 3
    lens_all = [700 + 53 * ((7 * i) % 11) for i in range(n_utts)]            # ragged, not monotone in i
    plan = plan_shards(lens_all, world, batch)
    steps = max(len(p) for p in plan)
    got = {}
    ok = True
    for k in range(steps):
        mine = plan[rank][k] if k < len(plan[rank]) else []
        lens = [lens_all[i] for i in mine]
        wav = torch.zeros(len(mine), 1300)
        for j, i in enumerate(mine):
            wav[j, : lens[j]] = torch.arange(lens[j], dtype=torch.float32) + 10000.0 * i
        out, per_rank = gather_audio(wav, lens, b_max=batch, s_max=1300)
        for r in range(world):
            idx = plan[r][k] if k < len(plan[r]) else []
            ok &= len(per_rank[r]) == len(idx)
            for j, i in enumerate(idx):
                got[i] = (per_rank[r][j], out[r * batch + j].clone())
    from fangyan_tts_amd import parallel
    cache = parallel._gather_cache(None)
    ok &= len([k for k in cache if k != "group"]) == 1 and steps >= 2          # one set of buffers served every step
    # a result handed out is the caller's: the next call does not overwrite it (reuse=True hands out the shared buffer instead)
    w0 = torch.full((1, 1300), float(rank + 1))
    a, _ = gather_audio(w0, [1300], b_max=batch, s_max=1300)
    b, _ = gather_audio(w0 * 2, [1300], b_max=batch, s_max=1300, reuse=True)
    c, _ = gather_audio(w0 * 3, [1300], b_max=batch, s_max=1300, reuse=True)
    ok &= float(a[0, 0]) == 1.0 and float(c[0, 0]) == 3.0 and a.data_ptr() != c.data_ptr() and a.data_ptr() != b.data_ptr()
    if world == 1:
        ok &= b.data_ptr() == c.data_ptr() and float(b[0, 0]) == 3.0
    ok &= sorted(got) == list(range(n_utts))
    for i in range(n_utts):
        n, row = got[i]
        ok &= n == lens_all[i]
        ok &= bool(torch.equal(row[:n], torch.arange(n, dtype=torch.float32) + 10000.0 * i)) and float(row[n:].abs().max()) == 0.0
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_sorted_shards_gather_gloo_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_sorted, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))
