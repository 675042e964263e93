"""CPU tests: the oracle (oracle/) against the golden fixtures minted from the
reference's own modules (tests/golden/mint_goldens.py, run in the build
container).  This is what pins the oracle before any HIP kernel trusts it.

Tolerances: the oracle and the reference both run fp32 torch CPU ops on the
same values, so agreement is at rounding level; 1e-4 relative leaves room for a
different BLAS thread count on another host.
"""
import os

import numpy as np
import pytest
import torch

from _digest import check
from fangyan_tts_amd import synth
from fangyan_tts_amd.spec import FlowCfg, HiftCfg, LlmCfg, ModelCfg
from oracle import flow as oflow
from oracle import hift as ohift
from oracle import llm as ollm
from oracle import pipeline as opipe

G = os.path.join(os.path.dirname(__file__), "golden")
RT, AT = 1e-4, 1e-5


def fx(name):
    p = os.path.join(G, name)
    if not os.path.exists(p):
        pytest.skip(f"{name} not minted")
    return np.load(p)


def synth_mel(name, frames):
    return np.clip(synth.normal(name, (1, frames, 80), -5.0, 2.0), -11.5, 2.0)


def llm_case(cfg, n_text, n_ptext, p_tok, tag):
    hi = min(cfg.vocab, 151643)
    return (torch.from_numpy(synth.randint(f"in.llm.text.{tag}", (1, n_text), 0, hi)),
            torch.from_numpy(synth.randint(f"in.llm.ptext.{tag}", (1, n_ptext), 0, hi)),
            torch.from_numpy(synth.randint(f"in.llm.ptok.{tag}", (1, p_tok), 0, cfg.speech_tokens)))


def dit_inputs(T):
    return [torch.from_numpy(a) for a in (
        synth.normal(f"in.dit.x.{T}", (2, 80, T)), synth.normal(f"in.dit.mu.{T}", (2, 80, T)),
        synth.normal(f"in.dit.cond.{T}", (2, 80, T)), synth.normal(f"in.dit.spks.{T}", (2, 80)),
        np.array([0.3, 0.3], dtype=np.float32))]


# ------------------------------------------------------------------ manifests

def test_manifest_sizes():
    """Parameter counts the survey probed on the reference."""
    n = lambda m: sum(int(np.prod(s)) for s in m.values())
    assert n(HiftCfg().manifest()) == 20779887
    ml = LlmCfg().manifest()
    assert n(ml) - int(np.prod(ml["llm.model.lm_head.weight"])) == 506_178_560 or n(ml) > 5e8
    assert abs(n(FlowCfg().manifest()) - 332.3e6) < 0.5e6


def test_synth_is_index_addressable():
    full = synth.tensor("llm.model.model.embed_tokens.weight", (1000, 256))
    rows = synth.tensor("llm.model.model.embed_tokens.weight", (1000, 256), rows=[3, 999, 0])
    assert np.array_equal(rows, full[[3, 999, 0]])
    assert np.array_equal(synth.bf16_round(full), full)
    a = synth.hift_sine_noise(1000)
    b = synth.hift_sine_noise(400)
    assert np.array_equal(a[:, :400], b)


# ------------------------------------------------------------------ HiFT

@pytest.mark.parametrize("tag,cfg,frames", [("tiny", HiftCfg.tiny(), (12, 30)), ("full", HiftCfg(), (30,))])
def test_hift_against_reference(tag, cfg, frames):
    f = fx(f"hift_{tag}.npz")
    P = ohift.prepare(synth.state_dict(cfg.manifest()))
    ri = torch.from_numpy(synth.hift_rand_ini())
    with torch.no_grad():
        for Fr in frames:
            mel = torch.from_numpy(synth.uniform(f"in.hift.mel.{Fr}", (1, 80, Fr), 0.0, 1.0))
            sn = torch.from_numpy(synth.hift_sine_noise(Fr * cfg.upsample_total))
            f0 = ohift.f0_predictor(mel, P)
            check(f0, f, f"F{Fr}.f0", RT, 1e-3)
            s = ohift.sine_source(f0, P, cfg, ri, sn)
            check(s, f, f"F{Fr}.source", RT, AT)
            taps = ohift.decode_taps(mel, s, P, cfg)
            for k in ("conv_pre", "fuse0", "fuse1", "fuse2", "conv_post"):
                check(taps[k], f, f"F{Fr}.{k}", RT, 1e-4)
            check(taps["wav"], f, f"F{Fr}.wav", RT, AT)
            if f"F{Fr}.wav_full" in f:
                np.testing.assert_allclose(taps["wav"].numpy(), f[f"F{Fr}.wav_full"], rtol=RT, atol=AT)
        for i in range(3):
            for j in range(3):
                x = torch.from_numpy(synth.normal(f"in.hift.rb.{i}.{j}", (1, cfg.stage_ch(i), 200)))
                check(ohift.resblock(x, P, f"resblocks.{3 * i + j}", cfg.rb_d), f, f"rb{3 * i + j}", RT, AT)


def test_stft_istft_roundtrip_and_torch():
    """The written-out DFTs equal torch.stft/istft (what generator.py:491-505 calls)."""
    cfg = HiftCfg()
    s = torch.from_numpy(synth.normal("in.stft", (2, 4800)))
    spec = ohift.stft(s, cfg)
    w = torch.from_numpy(np.hanning(17)[:16].astype(np.float32))
    ref = torch.view_as_real(torch.stft(s, 16, 4, 16, window=w, return_complex=True))
    np.testing.assert_allclose(spec[:, :9].numpy(), ref[..., 0].numpy(), atol=2e-5)
    np.testing.assert_allclose(spec[:, 9:].numpy(), ref[..., 1].numpy(), atol=2e-5)
    mag = torch.sqrt(spec[:, :9] ** 2 + spec[:, 9:] ** 2)
    ph = torch.atan2(spec[:, 9:], spec[:, :9])
    back = ohift.istft(mag, ph, cfg)
    np.testing.assert_allclose(back.numpy(), s.numpy(), atol=2e-4)


# ------------------------------------------------------------------ flow

@pytest.mark.parametrize("tag,cfg,Ts,cases", [
    ("tiny", FlowCfg.tiny(), (16, 150), ((20, 10), (24, 0))),
    ("full", FlowCfg(), (16,), ()),
    ("sized", FlowCfg(), (650,), ()),          # BASELINE config 3's DiT sequence (10 s prompt + 3 s); T = 512 and the 10-step mels are GPU-side only (CPU minutes)
])
def test_flow_against_reference(tag, cfg, Ts, cases):
    f = fx(f"flow_{tag}.npz")
    P = oflow.prepare(synth.state_dict(cfg.manifest()))
    with torch.no_grad():
        for T in Ts:
            x, mu, cond, spks, t = dit_inputs(T)
            mask = torch.ones(2, 1, T)
            y = oflow.dit_forward(x, mask, mu, t, spks, cond, P, cfg)
            check(y, f, f"est{T}", RT, 1e-4)
            if f"est{T}.full" in f:
                np.testing.assert_allclose(y.numpy(), f[f"est{T}.full"], rtol=RT, atol=1e-4)
            ys = oflow.dit_forward(x, mask, mu, t, spks, cond, P, cfg, streaming=True)
            check(ys, f, f"est{T}.stream", RT, 1e-4)
        for n, p in cases:
            token = torch.from_numpy(synth.randint(f"in.flow.token.{n}", (1, n), 0, cfg.vocab))
            ptoken = torch.from_numpy(synth.randint(f"in.flow.ptoken.{p}", (1, p), 0, cfg.vocab))
            pfeat = torch.from_numpy(synth_mel(f"in.flow.pfeat.{p}", 2 * p))
            emb = torch.from_numpy(synth.normal("in.flow.spk", (1, cfg.spk_in)))
            z = torch.from_numpy(synth.flow_rand_noise(2 * (n + p)))
            mel = oflow.inference(token, ptoken, pfeat, emb, P, cfg, z)
            check(mel, f, f"cfm{n}_{p}", RT, 1e-4)


def test_euler_schedule():
    """t accumulates (t += dt) and dt is re-derived from it, flow_matching.py:118-122."""
    cfg = FlowCfg()
    ts = oflow.t_span(cfg)
    steps = oflow.euler_times(cfg)
    assert len(steps) == 10 and float(steps[0][0]) == 0.0
    assert abs(float(steps[-1][0] + steps[-1][1]) - 1.0) < 1e-6
    assert abs(float(steps[3][0]) - float(ts[3])) < 1e-6


# ------------------------------------------------------------------ LLM

@pytest.mark.parametrize("tag,cfg,cases,cap", [
    ("tiny", LlmCfg.tiny(), ((12, 8, 0), (10, 6, 30)), None),
    ("full", LlmCfg(), ((12, 8, 0),), 12),
    ("sized", LlmCfg(), ((14, 30, 250),), 6),    # behind a 250-token prompt (296-row prefill): BASELINE config 3's LM shape
])
def test_llm_tokens_bit_exact(tag, cfg, cases, cap):
    f = fx(f"llm_{tag}.npz")
    big = tag in ("full", "sized")
    if big:
        # only the rows the case gathers: the 151 936-row table is 136 M values
        man = {k: v for k, v in cfg.manifest().items() if "embed_tokens" not in k and "lm_head" not in k}
        P = ollm.prepare(synth.state_dict(man))
    else:
        P = ollm.prepare(synth.state_dict(cfg.manifest()))
    for c in cases:
        ctag = "%d_%d_%d" % c
        text, ptext, ptok = llm_case(cfg, *c, ctag)
        if big:
            ids = torch.cat([ptext, text], dim=1)[0].tolist()
            uniq = sorted(set(ids))
            rows = synth.tensor("llm.model.model.embed_tokens.weight", (cfg.vocab, cfg.hidden), rows=uniq)
            table = torch.zeros(max(uniq) + 1, cfg.hidden)
            table[uniq] = torch.from_numpy(rows)
            P["llm.model.model.embed_tokens.weight"] = table
        ref = f[f"c{ctag}.tokens"].tolist()
        lp = []
        toks = []
        for tid in ollm.inference(text, ptext, ptok, P, cfg, logp_out=lp):
            toks.append(tid)
            if cap and len(toks) >= cap:
                break
        n = len(toks) if cap else len(ref)
        assert toks[:n] == ref[:n]
        if not cap and not bool(f[f"c{ctag}.capped"]):
            assert len(toks) == len(ref)
        for s in range(min(3, len(lp))):
            check(lp[s], f, f"c{ctag}.logp{s}", 1e-4, 2e-4)


def test_greedy_rule_and_silent_filter():
    cfg = LlmCfg.tiny()
    lp = torch.full((cfg.n_speech,), -10.0)
    lp[cfg.eos] = 0.0
    lp[17] = -1.0
    lp[18] = -1.0
    assert ollm.greedy_id(lp, True, cfg) == 17            # eos forbidden, tie -> lowest index
    assert ollm.greedy_id(lp, False, cfg) == cfg.eos
    toks = [5, 1, 2, 28, 29, 55, 248, 494, 7, 1, 1]
    assert ollm.silent_filter(toks) == [5, 1, 2, 28, 29, 55, 7, 1, 1]


# ------------------------------------------------------------------ end to end

def test_tts_against_reference():
    f = fx("e2e_tiny.npz")
    cfg = ModelCfg.tiny()
    PL = ollm.prepare(synth.state_dict(cfg.llm.manifest()))
    PF = oflow.prepare(synth.state_dict(cfg.flow.manifest()))
    PH = ohift.prepare(synth.state_dict(cfg.hift.manifest()))
    for (n_text, n_ptext, p_llm, p_flow) in ((8, 6, 0, 12), (6, 5, 20, 20)):
        ctag = f"{n_text}_{n_ptext}_{p_llm}_{p_flow}"
        text, ptext, ptok = llm_case(cfg.llm, n_text, n_ptext, p_llm, ctag)
        inp = {
            "text": text, "prompt_text": ptext, "llm_prompt_speech_token": ptok,
            "flow_prompt_speech_token": torch.from_numpy(synth.randint(f"in.flow.ptoken.{p_flow}", (1, p_flow), 0, 6561)),
            "prompt_speech_feat": torch.from_numpy(synth_mel(f"in.flow.pfeat.{p_flow}", 2 * p_flow)),
            "flow_embedding": torch.from_numpy(synth.normal("in.flow.spk", (1, 192))),
        }
        max_T = 2 * (p_flow + 20 * n_text)
        out = opipe.tts(inp, PL, PF, PH, cfg, torch.from_numpy(synth.flow_rand_noise(max_T)),
                        torch.from_numpy(synth.hift_rand_ini()),
                        torch.from_numpy(synth.hift_sine_noise(2 * 20 * n_text * 480)))
        assert out["tokens"][0].tolist() == f[f"c{ctag}.tokens"].tolist()
        check(out["mel"], f, f"c{ctag}.mel", RT, 1e-4)
        check(out["tts_speech"], f, f"c{ctag}.wav", 1e-3, 1e-4)


# ------------------------------------------------------------------ streaming forms

def test_stream_against_reference():
    """finalize=False vocoder / flow chunk and CosyVoice3Model.tts(stream=True), chunk by chunk."""
    f = fx("stream_tiny.npz")
    cfg = ModelCfg.tiny()
    PL = ollm.prepare(synth.state_dict(cfg.llm.manifest()))
    PF = oflow.prepare(synth.state_dict(cfg.flow.manifest()))
    PH = ohift.prepare(synth.state_dict(cfg.hift.manifest()))
    ri = torch.from_numpy(synth.hift_rand_ini())
    Fr = 30
    mel = torch.from_numpy(synth.uniform(f"in.hift.mel.{Fr}", (1, 80, Fr), 0.0, 1.0))
    wav, s = ohift.inference(mel, PH, cfg.hift, ri, torch.from_numpy(synth.hift_sine_noise(Fr * 480)), finalize=False)
    assert wav.shape == (1, (Fr - 8) * 480) and s.shape == (1, 1, (Fr - 3) * 480)
    check(s, f, f"hift.F{Fr}.source", 1e-3, 1e-4)
    np.testing.assert_allclose(wav.numpy(), f[f"hift.F{Fr}.wav_full"], rtol=1e-3, atol=1e-4)
    n, p_tok = 31, 10
    token = torch.from_numpy(synth.randint(f"in.flow.token.{n}", (1, n), 0, cfg.flow.vocab))
    ptoken = torch.from_numpy(synth.randint(f"in.flow.ptoken.{p_tok}", (1, p_tok), 0, cfg.flow.vocab))
    pfeat = torch.from_numpy(synth_mel(f"in.flow.pfeat.{p_tok}", 2 * p_tok))
    emb = torch.from_numpy(synth.normal("in.flow.spk", (1, cfg.flow.spk_in)))
    m = oflow.inference(token, ptoken, pfeat, emb, PF, cfg.flow, torch.from_numpy(synth.flow_rand_noise(2 * (n + p_tok))),
                        streaming=True, finalize=False)
    assert m.shape == (1, 80, 2 * (n - cfg.flow.pre_lookahead))
    check(m, f, f"flow.{n}_{p_tok}", RT, 1e-4)
    n_text, n_ptext, p_llm, p_flow = 40, 6, 0, 12
    ctag = f"{n_text}_{n_ptext}_{p_llm}_{p_flow}"
    text, ptext, ptok = llm_case(cfg.llm, n_text, n_ptext, p_llm, ctag)
    inp = {
        "text": text, "prompt_text": ptext, "llm_prompt_speech_token": ptok,
        "flow_prompt_speech_token": torch.from_numpy(synth.randint(f"in.flow.ptoken.{p_flow}", (1, p_flow), 0, 6561)),
        "prompt_speech_feat": torch.from_numpy(synth_mel(f"in.flow.pfeat.{p_flow}", 2 * p_flow)),
        "flow_embedding": torch.from_numpy(synth.normal("in.flow.spk", (1, 192))),
    }
    out = opipe.tts_stream(inp, PL, PF, PH, cfg, torch.from_numpy(synth.flow_rand_noise(2 * (p_flow + 20 * n_text))), ri,
                           torch.from_numpy(synth.hift_sine_noise(2 * 20 * n_text * 480)))
    assert out["tokens"][0].tolist() == f[f"e2e.c{ctag}.tokens"].tolist()
    assert [c.shape[1] for c in out["chunks"]] == f[f"e2e.c{ctag}.chunk_samples"].tolist()
    for i, c in enumerate(out["chunks"]):
        check(c, f, f"e2e.c{ctag}.chunk{i}", 1e-3, 1e-4)
    check(out["tts_speech"], f, f"e2e.c{ctag}.wav", 1e-3, 1e-4)


# ------------------------------------------------------------------ the reference's default sampler

def test_inv_cdf_is_a_sampler():
    """The stand-in for torch.multinomial: monotone in u, exact on dyadic weights, never an index of zero weight."""
    p = np.array([0.25, 0.0, 0.5, 0.25], dtype=np.float32)
    assert [ollm.inv_cdf(p, u) for u in (0.0, 0.2499, 0.25, 0.7499, 0.75, 0.9999)] == [0, 0, 2, 2, 3, 3]
    rng = np.random.default_rng(0)
    w = rng.random(6761).astype(np.float32)
    us = np.sort(rng.random(200).astype(np.float32))
    picks = [ollm.inv_cdf(w, u) for u in us]
    assert picks == sorted(picks)
    cdf = np.cumsum(w.astype(np.float64)) / w.astype(np.float64).sum()
    assert all(abs(cdf[i] - u) < 1e-3 for i, u in zip(picks, us))


def test_llm_ras_against_reference():
    """ras_sampling inside sampling_ids, draws from the supplied uniforms: token ids identical to the reference's, the
    give-up RuntimeError included."""
    f = fx("llm_ras_tiny.npz")
    cfg = ModelCfg.tiny().llm
    P = ollm.prepare(synth.state_dict(cfg.manifest()))
    for (n_text, n_ptext, p_tok) in ((12, 8, 0), (10, 6, 30), (16, 4, 10), (30, 5, 0)):
        ctag = f"{n_text}_{n_ptext}_{p_tok}"
        text, ptext, ptok = llm_case(cfg, n_text, n_ptext, p_tok, ctag)
        u = synth.uniform(f"in.llm.ras_u.{ctag}", (4096,), 0.0, 1.0)
        toks, raised = [], ""
        try:
            for t in ollm.inference(text, ptext, ptok, P, cfg, uniforms=u):
                toks.append(t)
        except RuntimeError as e:
            raised = str(e)
        assert raised == str(f[f"c{ctag}.raised"])
        assert toks == f[f"c{ctag}.tokens"].tolist()
