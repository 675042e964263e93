"""GPU parity of the whole per-utterance path (LM -> flow -> HiFT) through the host-side
CosyVoice3Model mirror, against the e2e fixtures minted from the reference's own
CosyVoice3Model.tts and against the oracle; plus batching invariants.

Stated tolerance (bf16 MFMA in flow and HiFT): tokens bit-exact; mel max |err| <= 8e-2
(values of scale ~1.5); wav max |err| <= 2e-2 with |wav| ~ 0.1-0.2.
"""
import numpy as np
import pytest
import torch

from _digest import check
from fangyan_tts_amd import synth
from fangyan_tts_amd.spec import ModelCfg
from gpu_util import golden, llm_case, maxerr, note, synth_mel

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def e2e_input(cfg, n_text, n_ptext, p_llm, p_flow):
    ctag = f"{n_text}_{n_ptext}_{p_llm}_{p_flow}"
    t, pt, pk = llm_case(cfg.llm, n_text, n_ptext, p_llm, ctag)
    return {
        "text": torch.tensor([t], dtype=torch.int32), "prompt_text": torch.tensor([pt], dtype=torch.int32),
        "llm_prompt_speech_token": torch.tensor([pk], dtype=torch.int32).reshape(1, -1),
        "flow_prompt_speech_token": torch.from_numpy(synth.randint(f"in.flow.ptoken.{p_flow}", (1, p_flow), 0, 6561)),
        "prompt_speech_feat": torch.from_numpy(synth_mel(f"in.flow.pfeat.{p_flow}", 2 * p_flow)),
        "flow_embedding": torch.from_numpy(synth.normal("in.flow.spk", (1, 192))),
    }, ctag


@pytest.fixture(scope="module")
def tiny():
    from fangyan_tts_amd.cli.model import CosyVoice3Model
    cfg = ModelCfg.tiny()
    sd = [synth.state_dict_torch(m.manifest(), DEV) for m in (cfg.llm, cfg.flow, cfg.hift)]
    noise = torch.from_numpy(synth.flow_rand_noise(2 * (20 + 20 * 8)))
    ri = torch.from_numpy(synth.hift_rand_ini())
    sn = torch.from_numpy(synth.hift_sine_noise(2 * 20 * 8 * 480))
    m = CosyVoice3Model(sd[0], sd[1], sd[2], cfg, device=DEV, max_batch=4, max_text=32, max_prompt_tokens=32, max_tokens=160,
                        rand_noise=noise, rand_ini=ri, sine_noise=sn)
    return m, cfg


CASES = [(8, 6, 0, 12), (6, 5, 20, 20)]


def test_tts_against_reference_fixture(tiny):
    m, cfg = tiny
    f = golden("e2e_tiny.npz")
    assert f is not None
    for c in CASES:
        inp, ctag = e2e_input(cfg, *c)
        wav, samples, toks = m.tts_batch([inp])
        assert toks[0].cpu().tolist() == f[f"c{ctag}.tokens"].tolist()
        out = next(m.tts(**inp))["tts_speech"]
        assert out.shape == (1, samples[0]) and out.device.type == "cpu"
        assert torch.equal(out, wav[:, : samples[0]])
        check(out, f, f"c{ctag}.wav", 5e-2, 2e-2)


def test_batch_equals_solo(tiny):
    m, cfg = tiny
    ins = [e2e_input(cfg, *c)[0] for c in CASES]
    wav, samples, toks = m.tts_batch(ins)
    for b, inp in enumerate(ins):
        w1, s1, t1 = m.tts_batch([inp])
        assert s1[0] == samples[b] and torch.equal(t1[0], toks[b])
        e = maxerr(wav[b, : samples[b]], w1[0, : s1[0]])
        note("parity_e2e.json", f"batch_vs_solo.{b}", e)
        assert e < 1e-5


def test_full_size_against_reference_fixture():
    """CosyVoice3-0.5B shapes, the reference's CosyVoice3Model.tts fixture (8 text ids, 1 s prompt).

    tokens: bit-exact.  mel: max |err| <= 8e-2 against the fixture.  wav: the vocoder's harmonic source
    integrates f0 over time (phase = 2 pi 480 cumsum(h f0 / 24000), generator.py:255-258), so a 1e-2 mel
    difference grows into an O(1) phase difference of the upper harmonics within ~50 frames - the reason
    the reference keeps its f0 predictor on the CPU (generator.py:715).  Sample-wise wav parity against
    the fp32 reference is therefore asserted (a) on the first 10 frames, (b) in full against the oracle
    vocoder run on the engine's own mel (<= 1.5e-2), which isolates the vocoder from the mel rounding."""
    f = golden("e2e_full.npz")
    if f is None:
        pytest.skip("e2e_full.npz not minted")
    from fangyan_tts_amd.cli.model import CosyVoice3Model
    from oracle import hift as ohift
    cfg = ModelCfg()
    sd = [synth.state_dict_torch(mm.manifest(), DEV, skip=("lm_head",)) for mm in (cfg.llm, cfg.flow, cfg.hift)]
    noise = torch.from_numpy(synth.flow_rand_noise(2 * (25 + 20 * 8)))
    ri = torch.from_numpy(synth.hift_rand_ini())
    sn = torch.from_numpy(synth.hift_sine_noise(2 * 20 * 8 * 480))
    m = CosyVoice3Model(sd[0], sd[1], sd[2], cfg, device=DEV, max_batch=1, max_text=32, max_prompt_tokens=32, max_tokens=160,
                        rand_noise=noise, rand_ini=ri, sine_noise=sn)
    inp, ctag = e2e_input(cfg, 8, 8, 0, 25)
    wav, samples, toks = m.tts_batch([inp])
    assert toks[0].cpu().tolist() == f[f"c{ctag}.tokens"].tolist()
    mel = m.last_mel.cpu()
    check(mel, f, f"c{ctag}.mel", 5e-2, 8e-2)
    P = ohift.prepare({k: v.cpu().numpy() for k, v in sd[2].items()})
    ref, _ = ohift.inference(mel, P, cfg.hift, ri, sn[:, : samples[0]])
    e = maxerr(wav[:, : samples[0]], ref)
    note("parity_e2e.json", "full.wav_vs_oracle_vocoder_on_engine_mel", e)
    assert e < 1.5e-2
    from _digest import sample_idx
    idx = sample_idx(samples[0])
    early = idx < 4800
    got = wav[0, : samples[0]].numpy()[idx][early]
    want = f[f"c{ctag}.wav.samples"][early]
    note("parity_e2e.json", "full.wav_first10frames_maxerr", float(np.abs(got - want).max()))
    assert np.abs(got - want).max() < 2e-2
