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
