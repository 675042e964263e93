"""GPU parity: the HIP HiFT vocoder (through the C ABI) against the CPU oracle
and the golden fixture minted from the reference.

Modes and their stated tolerances (absolute, on tensors whose scale is O(1);
wav is in [-0.99, 0.99] with |wav| ~ 0.1 for the synthetic weights):
  FY_DIRECT   exact fp32 VALU convolutions                 -> 2e-4
  FY_PRECISE  bf16 MFMA with split (hi+lo) activations     -> 2e-3
  default     bf16 MFMA, activations rounded to bf16       -> 3e-2 on taps, 1e-2 on wav
"""
import json
import os

import numpy as np
import pytest
import torch

from _digest import check
from fangyan_tts_amd import synth
from fangyan_tts_amd.spec import HiftCfg

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
OUT = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out")


def note(key, value):
    os.makedirs(OUT, exist_ok=True)
    p = os.path.join(OUT, "parity_hift.json")
    d = json.load(open(p)) if os.path.exists(p) else {}
    d[key] = value
    json.dump(d, open(p, "w"), indent=1, sort_keys=True)


def maxerr(a, b):
    return float((a.detach().cpu().float() - b.detach().cpu().float()).abs().max())


@pytest.fixture(scope="module")
def env():
    from fangyan_tts_amd.hift import HiftEngine
    from oracle import hift as ohift
    cfg = HiftCfg()
    dev = torch.device("cuda:0")
    sd = synth.state_dict_torch(cfg.manifest(), dev)
    eng = HiftEngine(sd, cfg, max_batch=4, max_frames=64)
    P = ohift.prepare({k: v.cpu().numpy() for k, v in sd.items()})
    ri = torch.from_numpy(synth.hift_rand_ini())
    sn = torch.from_numpy(synth.hift_sine_noise(64 * 480))
    return dict(cfg=cfg, eng=eng, P=P, o=ohift, dev=dev, ri=ri, sn=sn, ri_d=ri.to(dev), sn_d=sn.to(dev).contiguous())


def mel_of(Fr):
    return torch.from_numpy(synth.uniform(f"in.hift.mel.{Fr}", (1, 80, Fr), 0.0, 1.0))


def test_f0_and_source(env):
    o, eng, cfg = env["o"], env["eng"], env["cfg"]
    mel = mel_of(30)
    with torch.no_grad():
        f0_ref = o.f0_predictor(mel, env["P"])
        s_ref = o.sine_source(f0_ref, env["P"], cfg, env["ri"], env["sn"][:, : 30 * 480])
    f0 = eng.f0(mel.to(env["dev"]))
    e = maxerr(f0, f0_ref)
    note("f0_maxerr", e)
    assert e < 2e-3 * float(f0_ref.abs().max())
    # source from the oracle's own f0: isolates the sine generator
    s = eng.source(f0_ref.to(env["dev"]), env["ri_d"], env["sn_d"])
    e = maxerr(s, s_ref)
    note("source_maxerr_same_f0", e)
    assert e < 2e-4


@pytest.mark.parametrize("mode,flags,tol_tap,tol_wav", [
    ("direct", 2, 2e-5, 2e-6), ("precise", 1, 4e-5, 2e-6), ("bf16", 0, 4.5e-2, 1.5e-3)])
def test_decode_against_oracle(env, mode, flags, tol_tap, tol_wav):
    o, eng, cfg = env["o"], env["eng"], env["cfg"]
    Fr = 30
    mel = mel_of(Fr)
    with torch.no_grad():
        f0 = o.f0_predictor(mel, env["P"])
        s = o.sine_source(f0, env["P"], cfg, env["ri"], env["sn"][:, : Fr * 480])
        taps = o.decode_taps(mel, s, env["P"], cfg)
    wav = eng.decode(mel.to(env["dev"]), s.to(env["dev"]), flags=flags)
    for k in ("conv_pre", "fuse0", "stage0", "fuse1", "stage1", "fuse2", "stage2", "conv_post"):
        got = eng.tap(k, 1, Fr)
        e = maxerr(got, taps[k])
        note(f"decode.{mode}.{k}", e)
        assert e < tol_tap * max(1.0, float(taps[k].abs().max())), (k, e)
    e = maxerr(wav, taps["wav"])
    note(f"decode.{mode}.wav", e)
    assert e < tol_wav


def test_golden_reference_wav(env):
    """The fixture minted from the reference's CausalHiFTGenerator (F=30, full size)."""
    p = os.path.join(G, "hift_full.npz")
    if not os.path.exists(p):
        pytest.skip("hift_full.npz not minted")
    f = np.load(p)
    eng = env["eng"]
    wav, src = eng.inference(mel_of(30).to(env["dev"]), env["ri_d"], env["sn_d"], flags=2, want_source=True)
    check(src.cpu(), f, "F30.source", 1e-3, 5e-3)
    check(wav.cpu(), f, "F30.wav", 1e-3, 2e-3)
    np.testing.assert_allclose(wav.cpu().numpy(), f["F30.wav_full"], atol=2e-3)
    wav2, _ = eng.inference(mel_of(30).to(env["dev"]), env["ri_d"], env["sn_d"], flags=0)
    np.testing.assert_allclose(wav2.cpu().numpy(), f["F30.wav_full"], atol=1.5e-3)       # measured 4.6e-4
    note("golden.wav_bf16_maxerr", float(np.abs(wav2.cpu().numpy() - f["F30.wav_full"]).max()))


@pytest.mark.parametrize("index", range(9))
def test_resblock(env, index):
    o, eng, cfg = env["o"], env["eng"], env["cfg"]
    i, j = divmod(index, 3)
    x = torch.from_numpy(synth.normal(f"in.hift.rb.{i}.{j}", (1, cfg.stage_ch(i), 200)))
    with torch.no_grad():
        ref = o.resblock(x, env["P"], f"resblocks.{index}", cfg.rb_d)
    f = np.load(os.path.join(G, "hift_full.npz")) if os.path.exists(os.path.join(G, "hift_full.npz")) else None
    for mode, flags, tol in (("direct", 2, 1.5e-5), ("precise", 1, 3e-5), ("bf16", 0, 1.6e-2)):      # measured 4e-6 / 9e-6 / 5.3e-3
        y = eng.resblock(index, x.to(env["dev"]), flags)
        e = maxerr(y, ref)
        note(f"resblock{index}.{mode}", e)
        assert e < tol * max(1.0, float(ref.abs().max())), (mode, e)
        if f is not None and mode == "direct":
            check(y.cpu(), f, f"rb{index}", 1e-3, 1e-3)


def test_ragged_batch_equals_solo(env):
    """Utterances of different length in one batch match their solo runs (causal net, per-utterance lengths)."""
    eng, dev = env["eng"], env["dev"]
    frames = [30, 17, 24]
    mels = [mel_of(30)[:, :, :f] for f in frames]
    batch = torch.zeros(3, 80, 30)
    for b, m in enumerate(mels):
        batch[b, :, : frames[b]] = m[0]
    # garbage beyond each utterance's length must not leak in
    batch[1, :, 17:] = 7.0
    wav, _ = eng.inference(batch.to(dev), env["ri_d"], env["sn_d"], frames=frames, flags=0)
    for b, m in enumerate(mels):
        solo, _ = eng.inference(m.contiguous().to(dev), env["ri_d"], env["sn_d"], flags=0)
        n = frames[b] * 480
        assert torch.equal(wav[b, :n], solo[0, :n]), b
        assert float(wav[b, n:].abs().max()) == 0.0 if n < wav.shape[1] else True


def test_full_inference_against_oracle(env):
    o, eng, cfg = env["o"], env["eng"], env["cfg"]
    Fr = 24
    mel = mel_of(30)[:, :, :Fr].contiguous()
    with torch.no_grad():
        ref, _ = o.inference(mel, env["P"], cfg, env["ri"], env["sn"][:, : Fr * 480])
    for mode, flags, tol in (("direct", 2, 2.5e-5), ("bf16", 0, 2.5e-3)):                # measured 7e-6 / 7.4e-4
        wav, _ = eng.inference(mel.to(env["dev"]), env["ri_d"], env["sn_d"], flags=flags)
        e = maxerr(wav, ref)
        note(f"inference.{mode}.wav", e)
        assert e < tol


def test_fused_resblock_iterations_equal_the_two_launch_form(env, tmp_path):
    """A ResBlock iteration runs as ONE launch (conv1 -> snake -> conv2 with the inner activation
    in LDS, conv.h: ResIterDesc).  Every accumulator sees the same MFMA sequence as in the two-launch form, so the waveform must
    be identical bit for bit - checked on a ragged batch, the other form forced through FY_HIFT_FUSE=0 in a child process."""
    import subprocess
    import sys
    eng, dev = env["eng"], env["dev"]
    frames = [30, 17, 24, 1]
    batch = torch.zeros(4, 80, 30)
    for b, f in enumerate(frames):
        batch[b, :, :f] = mel_of(30)[0, :, :f] * (1.0 + 0.1 * b)
    wav, _ = eng.inference(batch.to(dev), env["ri_d"], env["sn_d"], frames=frames, flags=0)
    rb = [eng.resblock(i, torch.from_numpy(synth.normal(f"in.hift.rb.{i // 3}.{i % 3}", (1, env["cfg"].stage_ch(i // 3), 333))).to(dev), 0).cpu().numpy() for i in range(9)]
    np.save(tmp_path / "mel.npy", batch.numpy())
    script = f"""
import numpy as np, torch, sys
sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})
from fangyan_tts_amd import synth
from fangyan_tts_amd.spec import HiftCfg
from fangyan_tts_amd.hift import HiftEngine
cfg = HiftCfg(); dev = torch.device('cuda:0')
eng = HiftEngine(synth.state_dict_torch(cfg.manifest(), dev), cfg, max_batch=4, max_frames=64)
ri = torch.from_numpy(synth.hift_rand_ini()).to(dev); sn = torch.from_numpy(synth.hift_sine_noise(64 * 480)).to(dev).contiguous()
wav, _ = eng.inference(torch.from_numpy(np.load({str(tmp_path / 'mel.npy')!r})).to(dev), ri, sn, frames={frames!r}, flags=0)
rb = [eng.resblock(i, torch.from_numpy(synth.normal(f"in.hift.rb.{{i // 3}}.{{i % 3}}", (1, cfg.stage_ch(i // 3), 333))).to(dev), 0).cpu().numpy() for i in range(9)]
np.savez({str(tmp_path / 'out.npz')!r}, wav=wav.cpu().numpy(), **{{f"rb{{k}}": v for k, v in enumerate(rb)}})
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the two-launch form, the fused form extended to the 256-channel stage (off by default: it measured slower), and rounds 1-4's
    # first iterations (FY_HIFT_ACT0=0: each ResBlock activates the fp32 up-conv / source output itself instead of copying the bf16
    # stream the producer's epilogue wrote under that ResBlock's first snake - the same values by construction)
    for extra in ({"FY_HIFT_FUSE": "0"}, {"FY_HIFT_FUSE256": "1"}, {"FY_HIFT_ACT0": "0"}, {"FY_HIFT_ACT0": "0", "FY_HIFT_FUSE": "0"}):
        envv = dict(os.environ, PYTHONPATH=os.pathsep.join([root] + sys.path), **extra)
        r = subprocess.run([sys.executable, "-c", script], env=envv, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        other = np.load(tmp_path / "out.npz")
        assert np.array_equal(wav.cpu().numpy(), other["wav"]), extra
        for k, v in enumerate(rb):
            assert np.array_equal(v, other[f"rb{k}"]), (extra, k)
