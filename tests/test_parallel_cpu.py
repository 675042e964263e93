"""CPU tests of the N > 1 path: utterance sharding and the audio all-gather on a
world_size-2 gloo group (the GPU run uses the same code on RCCL)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fangyan_tts_amd.parallel import gather_audio, shard_range


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
