"""GPU parity cases for the other BASELINE.json configs (they are parity cases, not bench lines):

  configs[2]  zero-shot with a 10 s prompt: 250 prompt speech tokens in the LM prefill (~300 positions),
              500 prompt mel frames in the flow (T = 650), batch 4;
  configs[3]  batch 64 sharded data-parallel: per-rank micro-batches are independent - a batch of 8 equals
              two batches of 4 (the all-gather itself is covered by tests/test_parallel_cpu.py);
  configs[4]  HiFT-only on a long random mel (size-independent property: a long utterance equals the same
              frames vocoded as a prefix - the net is causal - and ragged batches equal solo runs).
"""
import numpy as np
import pytest
import torch

from fangyan_tts_amd import synth
from fangyan_tts_amd.spec import FlowCfg, HiftCfg, LlmCfg, ModelCfg
from gpu_util import maxerr, note, synth_mel

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_zero_shot_long_prefill_tokens_full_size():
    """CosyVoice3-0.5B shapes; prefill of 2 + 42 + 250 positions; tokens bit-exact against the oracle."""
    from fangyan_tts_amd.llm import LlmEngine
    from oracle import llm as ollm
    cfg = LlmCfg()
    sd = synth.state_dict_torch(cfg.manifest(), DEV, skip=("lm_head",))
    eng = LlmEngine(sd, cfg, max_batch=4, max_ctx=400)
    hi = 151643
    texts = [synth.randint(f"zs.text.{b}", (1, 12), 0, hi)[0].tolist() for b in range(2)]
    ptexts = [synth.randint(f"zs.ptext.{b}", (1, 30), 0, hi)[0].tolist() for b in range(2)]
    ptoks = [synth.randint(f"zs.ptok.{b}", (1, 250 - 17 * b), 0, cfg.speech_tokens)[0].tolist() for b in range(2)]
    out, out_n, raw_n = eng.generate(texts, ptexts, ptoks, max_len=[30, 30])
    P = {k: v.cpu() for k, v in sd.items()}
    for b in range(2):
        ref = list(ollm.inference(torch.tensor([texts[b]]), torch.tensor([ptexts[b]]), torch.tensor([ptoks[b]]), P, cfg, max_len=30))
        ref = ollm.silent_filter(ref)
        got = out[b, : int(out_n[b])].cpu().tolist()
        note("parity_configs.json", f"zero_shot.llm.{b}", [len(got), len(ref)])
        assert got == ref, (b, got[:8], ref[:8])


def test_zero_shot_flow_T650_tiny_against_oracle_and_batch4():
    """T = 650 (500 prompt frames + 150 generated): oracle parity at the reduced size, batch 4 = solo."""
    from fangyan_tts_amd.flow import FlowEngine
    from oracle import flow as oflow
    cfg = FlowCfg.tiny()
    sd = synth.state_dict_torch(cfg.manifest(), DEV)
    eng = FlowEngine(sd, cfg, max_batch=4, max_frames=660)
    P = {k: v.cpu() for k, v in sd.items()}
    n, p = 75, 250
    z = torch.from_numpy(synth.flow_rand_noise(2 * (n + p)))
    toks = torch.from_numpy(synth.randint("zs.flow.tok", (4, n), 0, cfg.vocab))
    ptok = torch.from_numpy(synth.randint("zs.flow.ptok", (4, p), 0, cfg.vocab))
    pfeat = torch.from_numpy(np.concatenate([synth_mel(f"zs.flow.pfeat.{b}", 2 * p) for b in range(4)]))
    emb = torch.from_numpy(synth.normal("zs.flow.spk", (4, cfg.spk_in)))
    mel = eng.inference(toks, [n] * 4, ptok, [p] * 4, pfeat, [2 * p] * 4, emb, z)
    with torch.no_grad():
        ref = oflow.inference(toks[:1], ptok[:1], pfeat[:1], emb[:1], P, cfg, z)
    e = maxerr(mel[:1], ref)
    note("parity_configs.json", "zero_shot.flow_T650.max", e)
    assert e < 6e-2
    solo = eng.inference(toks[2:3], [n], ptok[2:3], [p], pfeat[2:3], [2 * p], emb[2:3], z)
    assert maxerr(mel[2:3], solo) < 1e-5


def test_batch8_equals_two_batches_of_4():
    """Data-parallel sharding is exact: utterances do not interact (tiny model, full path)."""
    from fangyan_tts_amd.cli.model import CosyVoice3Model
    cfg = ModelCfg.tiny()
    sd = [synth.state_dict_torch(m.manifest(), DEV) for m in (cfg.llm, cfg.flow, cfg.hift)]
    noise = torch.from_numpy(synth.flow_rand_noise(2 * (16 + 40)))
    m = CosyVoice3Model(sd[0], sd[1], sd[2], cfg, device=DEV, max_batch=8, max_text=32, max_prompt_tokens=16, max_tokens=40,
                        rand_noise=noise, rand_ini=torch.from_numpy(synth.hift_rand_ini()),
                        sine_noise=torch.from_numpy(synth.hift_sine_noise(2 * 40 * 480)))
    ins = []
    for b in range(8):
        nt = 6 + b % 3
        ins.append({
            "text": torch.from_numpy(synth.randint(f"dp.text.{b}", (1, nt), 0, cfg.llm.vocab)),
            "prompt_text": torch.from_numpy(synth.randint(f"dp.ptext.{b}", (1, 4), 0, cfg.llm.vocab)),
            "llm_prompt_speech_token": torch.zeros(1, 0, dtype=torch.int32),
            "flow_prompt_speech_token": torch.from_numpy(synth.randint(f"dp.ptok.{b}", (1, 10 + b % 4), 0, 6561)),
            "prompt_speech_feat": torch.from_numpy(synth_mel(f"dp.pfeat.{b}", 2 * (10 + b % 4))),
            "flow_embedding": torch.from_numpy(synth.normal(f"dp.spk.{b}", (1, 192))),
        })
    lens = [20 + 2 * b for b in range(8)]
    wav, samples, toks = m.tts_batch(ins, min_len=lens, max_len=lens)
    from fangyan_tts_amd.parallel import shard_range
    for r in range(2):
        idx = list(shard_range(8, r, 2))
        w2, s2, t2 = m.tts_batch([ins[i] for i in idx], min_len=[lens[i] for i in idx], max_len=[lens[i] for i in idx])
        for j, i in enumerate(idx):
            assert s2[j] == samples[i] and torch.equal(t2[j], toks[i])
            assert maxerr(w2[j, : s2[j]], wav[i, : samples[i]]) < 1e-5


def test_hift_long_mel_causal_prefix_property():
    """configs[4] shape in miniature (the full 32 x 10 000 run is bench_hift.py): 2 000-frame random mel; the first
    1 000 frames' audio does not depend on the later frames except through the 4-frame look-ahead of conv_pre /
    the 3-frame look-ahead of the f0 predictor."""
    from fangyan_tts_amd.hift import HiftEngine
    cfg = HiftCfg()
    sd = synth.state_dict_torch(cfg.manifest(), DEV)
    eng = HiftEngine(sd, cfg, max_batch=2, max_frames=2000)
    g = torch.Generator().manual_seed(3)
    mel = torch.rand(1, 80, 2000, generator=g).to(DEV)
    ri = torch.from_numpy(synth.hift_rand_ini()).to(DEV)
    sn = torch.rand(1, 2000 * 480, 9, generator=g).to(DEV)
    full, _ = eng.inference(mel, ri, sn)
    pre, _ = eng.inference(mel[:, :, :1004].contiguous(), ri, sn)
    # frames < 1000 of the 1004-frame run see the same look-ahead as in the full run
    n = 1000 * 480
    e = maxerr(full[:, :n], pre[:, :n])
    note("parity_configs.json", "hift.long_prefix.max", e)
    assert e < 1e-5
    assert float(full.abs().max()) <= 0.99 + 1e-6


def test_hift_config5_full_size_properties():
    """BASELINE.json configs[4] at its FULL size - 32 utterances x 10 000 frames, the shape bench_hift.py times - checked
    through properties that do not need a reference of that size (the stage tensors pass 2^31 elements here, which is
    where an index width shows):
      * ragged batch = solo: utterances 0 and 31 (the last one, 9 000 frames in a ragged batch) equal solo runs of their own mel;
      * causal prefix: utterance 31's first 5 000 frames equal a solo run of its first 5 004 frames (conv_pre looks 4 ahead).
        The prefix is longer than 4 096 frames on purpose: the exact-fp32 f0 predictor picks its launch form (K split over
        the waves or not) from the sequence length, so a SHORT prefix sums in another order, f0 moves in the last bit and the
        harmonic source integrates that over a thousand frames (measured 7.6e-4 on the waveform for a 1 004-frame prefix -
        an fp32 ordering effect, checked at the vocoder's stated tolerance below, not a causality leak);
      * |wav| <= 0.99 (generator.py:746 clamp), no NaN, and the ragged tail beyond an utterance's length is silent."""
    from fangyan_tts_amd.hift import HiftEngine
    cfg = HiftCfg()
    sd = synth.state_dict_torch(cfg.manifest(), DEV)
    B, F = 32, 10000
    eng = HiftEngine(sd, cfg, max_batch=B, max_frames=F)
    g = torch.Generator(device=DEV).manual_seed(5)
    mel = torch.rand(B, 80, F, device=DEV, generator=g)
    ri = torch.from_numpy(synth.hift_rand_ini()).to(DEV)
    sn = torch.rand(1, F * 480, 9, device=DEV, generator=g)
    frames = [F] * B
    frames[31], frames[7] = 9000, 6001
    wav, _ = eng.inference(mel, ri, sn, frames=frames)
    assert wav.shape == (B, F * 480)
    assert bool(torch.isfinite(wav).all()) and float(wav.abs().max()) <= 0.99 + 1e-6
    assert float(wav[31, 9000 * 480:].abs().max()) == 0.0 and float(wav[7, 6001 * 480:].abs().max()) == 0.0
    for b in (0, 31):
        solo, _ = eng.inference(mel[b: b + 1, :, : frames[b]].contiguous(), ri, sn)
        e = maxerr(wav[b: b + 1, : frames[b] * 480], solo[:, : frames[b] * 480])
        note("parity_configs.json", f"hift.cfg5.ragged_equals_solo.{b}", e)
        assert e < 1e-5, (b, e)
    pre, _ = eng.inference(mel[31:32, :, :5004].contiguous(), ri, sn)
    e = maxerr(wav[31:32, : 5000 * 480], pre[:, : 5000 * 480])
    note("parity_configs.json", "hift.cfg5.prefix5000.31", e)
    assert e < 1e-5
    pre, _ = eng.inference(mel[31:32, :, :1004].contiguous(), ri, sn)
    e = maxerr(wav[31:32, : 1000 * 480], pre[:, : 1000 * 480])
    note("parity_configs.json", "hift.cfg5.prefix1000.31", e)
    assert e < 2.5e-3                                  # the other launch form of the f0 convs: fp32 summation order (docstring)
    eng.close()


def test_instruct_batch8_full_size_against_oracle():
    """BASELINE.json configs[1] at full size (CosyVoice3-0.5B shapes, 8 instruct utterances of mixed text length behind a 5 s
    prompt, 75 forced tokens = the benchmark's batch): utterances 0 and 7 of the batch are held to the CPU oracle's whole
    per-utterance path - ids exact, mel <= 4e-2 (bf16 flow decoder, measured ~1.3e-2), waveform <= 2.5e-3 from the oracle
    vocoder run on the engine's own mel."""
    import bench
    from fangyan_tts_amd.cli.model import CosyVoice3Model
    from oracle import hift as ohift, pipeline as opipe
    cfg = ModelCfg()
    sd_llm = synth.state_dict_torch(cfg.llm.manifest(), DEV, skip=("lm_head",))
    sd_flow = synth.state_dict_torch(cfg.flow.manifest(), DEV)
    sd_hift = synth.state_dict_torch(cfg.hift.manifest(), DEV)
    N, P = bench.N_TOK, bench.P_TOK
    noise = torch.from_numpy(synth.flow_rand_noise(2 * (P + N))).to(DEV)
    ri = torch.from_numpy(synth.hift_rand_ini()).to(DEV)
    sn = torch.from_numpy(synth.hift_sine_noise(2 * N * 480)).to(DEV)
    m = CosyVoice3Model(sd_llm, sd_flow, sd_hift, cfg, device=DEV, max_batch=8, max_text=64, max_prompt_tokens=P, max_tokens=N,
                        rand_noise=noise, rand_ini=ri, sine_noise=sn)
    inputs = bench.make_inputs(cfg, 0)
    forced = [N] * 8
    wav, samples, toks = m.tts_batch(inputs, min_len=forced, max_len=forced)
    mel = m.last_mel.cpu()
    torch.set_num_threads(16)
    PL = {k: v.cpu() for k, v in sd_llm.items()}
    PF = {k: v.cpu() for k, v in sd_flow.items()}
    PH = ohift.prepare({k: v.cpu().numpy() for k, v in sd_hift.items()})
    for b in (0, 7):
        ref = opipe.tts(inputs[b], PL, PF, PH, cfg, noise.cpu(), ri.cpu(), sn.cpu(), min_len=N, max_len=N)
        assert toks[b].cpu().reshape(-1).tolist() == ref["tokens"].reshape(-1).tolist(), b
        e_mel = maxerr(mel[b: b + 1, :, : 2 * N], ref["mel"])
        S = samples[b]
        ref_wav, _ = ohift.inference(mel[b: b + 1, :, : 2 * N], PH, cfg.hift, ri.cpu(), sn.cpu()[:, :S])
        e_wav = maxerr(wav[b: b + 1, :S], ref_wav)
        note("parity_configs.json", f"instruct_b8_full.{b}", [e_mel, e_wav])
        assert e_mel <= 4e-2 and e_wav <= 2.5e-3, (b, e_mel, e_wav)
    m.close()


def test_zero_shot_batch4_full_size_against_oracle():
    """BASELINE.json configs[2] at full size (CosyVoice3-0.5B shapes, zero-shot with a 10 s prompt: 30 prompt-text ids + 250 prompt
    speech tokens in the LM = a 296-row prefill per sequence, 500 prompt mel frames = DiT sequence 650, batch 4, 75 forced tokens -
    the batch `bench.py`'s `zero_shot_b4` times): utterances 0 and 3 of the batch are held to the CPU oracle's whole per-utterance
    path - ids exact, mel <= 4e-2 (bf16 flow decoder), waveform <= 2.5e-3 from the oracle vocoder run on the engine's own mel."""
    import bench
    from fangyan_tts_amd.cli.model import CosyVoice3Model
    from oracle import hift as ohift, pipeline as opipe
    cfg = ModelCfg()
    sd_llm = synth.state_dict_torch(cfg.llm.manifest(), DEV, skip=("lm_head",))
    sd_flow = synth.state_dict_torch(cfg.flow.manifest(), DEV)
    sd_hift = synth.state_dict_torch(cfg.hift.manifest(), DEV)
    N, P, B = bench.N_TOK, 250, 4
    noise = torch.from_numpy(synth.flow_rand_noise(2 * (P + N))).to(DEV)
    ri = torch.from_numpy(synth.hift_rand_ini()).to(DEV)
    sn = torch.from_numpy(synth.hift_sine_noise(2 * N * 480)).to(DEV)
    m = CosyVoice3Model(sd_llm, sd_flow, sd_hift, cfg, device=DEV, max_batch=B, max_text=64, max_prompt_tokens=P, max_tokens=N,
                        rand_noise=noise, rand_ini=ri, sine_noise=sn)
    inputs = bench.zero_shot_inputs(cfg, B, P)
    forced = [N] * B
    wav, samples, toks = m.tts_batch(inputs, min_len=forced, max_len=forced)
    mel = m.last_mel.cpu()
    torch.set_num_threads(16)
    PL = {k: v.cpu() for k, v in sd_llm.items()}
    PF = {k: v.cpu() for k, v in sd_flow.items()}
    PH = ohift.prepare({k: v.cpu().numpy() for k, v in sd_hift.items()})
    for b in (0, 3):
        ref = opipe.tts(inputs[b], PL, PF, PH, cfg, noise.cpu(), ri.cpu(), sn.cpu(), min_len=N, max_len=N)
        assert toks[b].cpu().reshape(-1).tolist() == ref["tokens"].reshape(-1).tolist(), b
        e_mel = maxerr(mel[b: b + 1, :, : 2 * N], ref["mel"])
        S = samples[b]
        ref_wav, _ = ohift.inference(mel[b: b + 1, :, : 2 * N], PH, cfg.hift, ri.cpu(), sn.cpu()[:, :S])
        e_wav = maxerr(wav[b: b + 1, :S], ref_wav)
        note("parity_configs.json", f"zero_shot_b4_full.{b}", [e_mel, e_wav])
        assert e_mel <= 4e-2 and e_wav <= 2.5e-3, (b, e_mel, e_wav)
    m.close()


def test_bench_two_ranks_on_one_gpu_rehearsal():
    """BASELINE.json configs[3]'s launch path (`bench.py --gpus N`: self-started ranks, per-rank pipelines, one fused all-gather of
    the finished audio per step, max-over-ranks timing) rehearsed with two ranks that share this box's one GPU over gloo
    (FY_BENCH_REHEARSAL=1: the path, not a measurement): rank 0 prints one JSON line that says n_gpus 2 and carries the
    aggregate of both ranks."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FY_BENCH_REHEARSAL="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "4", "--no-extras", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak"
    # 2 ranks x 8 utterances x 3 s per step
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 2 * 8 * 3.0) < 1e-3


def test_allgather_audio_c_entry_single_rank(tmp_path):
    """fy_allgather_audio (the C entry a host with its own RCCL communicator calls; Python hosts use parallel.gather_audio) on a
    one-rank communicator made with the RCCL copy torch ships: the fixed-size record is packed, gathered and unpacked - audio
    zero-padded / cut to (b_max, s_max), count and lengths (min(n, s_max)) behind it.  In a child process with a time limit: a
    bootstrap that never completes (no usable network interface) skips the test instead of hanging the suite; an error code
    from RCCL fails it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import ctypes as C, os, sys, torch
torch.cuda.set_device(0)
from fangyan_tts_amd import _lib
L = _lib.lib()
assert L.fy_allgather_audio(None, 1, None, 1, None, 0, 1, 1, None, None, None, None) != 0          # argument check, no RCCL needed
rccl = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=C.RTLD_GLOBAL)
class UID(C.Structure):
    _fields_ = [("b", C.c_char * 128)]
uid = UID()
rc = rccl.ncclGetUniqueId(C.byref(uid))
if rc != 0:
    print("FAIL ncclGetUniqueId returned", rc); sys.exit(1)
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
rc = rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0)
if rc != 0:
    print("FAIL ncclCommInitRank returned", rc); sys.exit(1)
# world must equal the communicator's size (the gather writes nranks records into a scratch sized from `world`)
assert L.fy_allgather_audio(comm, 2, 1, 1, 1, 0, 1, 1, 1, 1, 1, None) != 0 and b"ranks" in L.fy_last_error()
dev = torch.device("cuda:0")
b, b_max, s_max, ld = 3, 4, 1000, 1200
g = torch.Generator().manual_seed(1)
wav = torch.rand(b, ld, generator=g).to(dev)
lens = torch.tensor([1100, 17, 640], dtype=torch.int32, device=dev)      # row 0 is longer than s_max: cut, and published as s_max
scratch = torch.zeros(L.fy_allgather_audio_scratch_floats(1, b_max, s_max), device=dev)
wav_all = torch.full((b_max, s_max), -7.0, device=dev)
n_all = torch.full((1, b_max + 1), -1, dtype=torch.int32, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(L.fy_allgather_audio(comm, 1, wav.data_ptr(), ld, lens.data_ptr(), b, b_max, s_max, scratch.data_ptr(), wav_all.data_ptr(), n_all.data_ptr(), st))
torch.cuda.synchronize()
assert n_all.cpu().tolist() == [[3, 1000, 17, 640, 0]], n_all
for r, n in enumerate([1000, 17, 640]):
    assert torch.equal(wav_all[r, :n], wav[r, :n]) and float(wav_all[r, n:].abs().max() if n < s_max else 0.0) == 0.0
assert float(wav_all[3].abs().max()) == 0.0
rccl.ncclCommDestroy.argtypes = [C.c_void_p]
rccl.ncclCommDestroy(comm)
print("OK")
"""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([root] + sys.path), NCCL_SOCKET_IFNAME=os.environ.get("NCCL_SOCKET_IFNAME", "lo"),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    try:
        r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=180)
    except subprocess.TimeoutExpired:
        pytest.skip("making a one-rank RCCL communicator did not finish on this box")
    assert r.returncode == 0, r.stderr[-2000:]
    # an ERROR from ncclGetUniqueId / ncclCommInitRank is a failure (the child says which); only the time-out above - a box whose
    # bootstrap never completes - skips
    assert "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_audio_record_pack_and_cuda_gather_single_rank(tmp_path):
    """The DEFAULT multi-GPU delivery path of bench.py on a GPU: AudioGather's CUDA branch - fy_audio_record_pack (lengths in the
    kernel arguments, a row longer than s_max, fewer rows than b_max, a row pitch that is not the row length) + one
    all_gather_into_tensor over RCCL - against the CPU branch over gloo bit for bit, on a one-rank group of each kind.  In a child
    process with a time limit (a bootstrap that never completes skips; an error fails)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cpu_group = dist.new_group(backend="gloo")
from fangyan_tts_amd import _lib
from fangyan_tts_amd.parallel import AudioGather, gather_audio
b, b_max, s_max, S, pitch = 3, 5, 900, 880, 1216
g = torch.Generator().manual_seed(3)
big = torch.rand(b, pitch, generator=g)
wav_cpu = big[:, :S]                                  # not contiguous: the row pitch is 1216
wav_gpu = big.to(dev)[:, :S]
assert wav_gpu.stride(0) == pitch
lens = [2000, 17, 640]                                # row 0 claims more samples than the record holds: cut to s_max, published as min(n, S?) below
ga, gc = AudioGather(b_max, s_max, dev), AudioGather(b_max, s_max, "cpu", cpu_group)
# the record itself: the CPU branch's packing is the definition
out_c, per_c = gc(wav_cpu, [min(n, S) for n in lens])
out_g, per_g = ga(wav_gpu, [min(n, S) for n in lens])
torch.cuda.synchronize()
assert per_g == per_c == [[880, 17, 640]], (per_g, per_c)
assert torch.equal(ga.mine.cpu().view(torch.int32), gc.mine.view(torch.int32)), "records differ"
assert torch.equal(out_g.cpu(), out_c)
# a length beyond the capacity is clamped in the header and the row is cut (the pack kernel never reads past s_max)
wide = torch.rand(2, 1000, generator=g)
o2, p2 = ga(wide.to(dev)[:, :900], [950, 3])      # 950 > s_max (and <= the row pitch, which the entry checks)
torch.cuda.synchronize()
assert p2 == [[900, 3]] and torch.equal(o2[0].cpu(), wide[0, :900]) and float(o2[1, 3:].abs().max()) == 0.0 and float(o2[2:].abs().max()) == 0.0
# gather_audio on the GPU: a fresh tensor unless reuse=True
a, _ = gather_audio(wav_gpu, [min(n, S) for n in lens], b_max=b_max, s_max=s_max)
b2, _ = gather_audio(wav_gpu * 2, [min(n, S) for n in lens], b_max=b_max, s_max=s_max, reuse=True)
torch.cuda.synchronize()
assert torch.equal(a.cpu(), out_c) and a.data_ptr() != b2.data_ptr()
dist.destroy_process_group()
print("OK")
"""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([root] + sys.path), NCCL_SOCKET_IFNAME=os.environ.get("NCCL_SOCKET_IFNAME", "lo"),
               GLOO_SOCKET_IFNAME=os.environ.get("GLOO_SOCKET_IFNAME", "lo"), MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    try:
        r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired:
        pytest.skip("making a one-rank RCCL process group did not finish on this box")
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    assert "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
