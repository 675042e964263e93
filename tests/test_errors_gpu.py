"""The C ABI refuses what it cannot run - with a status and a message, before any kernel is launched - instead of
faulting: sizes beyond what a handle was created for, ids outside the tables, empty utterances, unfinished state.
(The reference raises Python exceptions at the same places: index errors of nn.Embedding, shape asserts, RuntimeError.)"""
import pytest
import torch

from fangyan_tts_amd import synth
from fangyan_tts_amd._lib import FyError
from fangyan_tts_amd.spec import ModelCfg
from gpu_util import llm_case, synth_mel

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def engines():
    from fangyan_tts_amd.flow import FlowEngine
    from fangyan_tts_amd.hift import HiftEngine
    from fangyan_tts_amd.llm import LlmEngine
    cfg = ModelCfg.tiny()
    llm = LlmEngine(synth.state_dict_torch(cfg.llm.manifest(), DEV, skip=("lm_head",)), cfg.llm, max_batch=2, max_ctx=64)
    flow = FlowEngine(synth.state_dict_torch(cfg.flow.manifest(), DEV), cfg.flow, max_batch=2, max_frames=64)
    hift = HiftEngine(synth.state_dict_torch(cfg.hift.manifest(), DEV), cfg.hift, max_batch=2, max_frames=40)
    return cfg, llm, flow, hift


def test_llm_limits(engines):
    cfg, llm, _, _ = engines
    t, pt, pk = llm_case(cfg.llm, 10, 6, 0, "10_6_0")
    with pytest.raises(FyError, match="batch"):
        llm.generate([t] * 3, [pt] * 3, [pk] * 3, max_len=[8] * 3)
    with pytest.raises(FyError, match="positions"):                       # 2 + 16 + 60 > max_ctx 64
        llm.generate([t], [pt], [pk], max_len=[60])
    with pytest.raises(FyError, match="vocabulary"):
        llm.generate([[cfg.llm.vocab + 5] + t[1:]], [pt], [pk], max_len=[8])
    with pytest.raises(FyError, match="prompt speech id"):
        llm.generate([t], [pt], [[cfg.llm.speech_tokens + 500]], max_len=[8])
    with pytest.raises(FyError, match="bad lengths"):
        llm.generate([[]], [[]], [[]], max_len=[8])
    out, out_n, _ = llm.generate([t], [pt], [pk], max_len=[8])          # and the handle still works afterwards
    assert 1 <= int(out_n.cpu()[0]) <= 8


def test_llm_step_without_begin():
    from fangyan_tts_amd.llm import LlmEngine
    cfg = ModelCfg.tiny()
    llm = LlmEngine(synth.state_dict_torch(cfg.llm.manifest(), DEV, skip=("lm_head",)), cfg.llm, max_batch=1, max_ctx=64)
    llm._gen = (torch.zeros(1, 8, dtype=torch.int32, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV),
                torch.zeros(1, dtype=torch.int32, device=DEV), 1)
    with pytest.raises(FyError, match="no generation in progress"):
        llm.step(4)


def test_flow_limits(engines):
    cfg, _, flow, _ = engines
    noise = torch.from_numpy(synth.flow_rand_noise(128))
    emb = torch.from_numpy(synth.normal("in.flow.spk", (1, 192)))
    ptok = torch.from_numpy(synth.randint("in.flow.ptoken.4", (1, 4), 0, 6561))
    pfeat = torch.from_numpy(synth_mel("in.flow.pfeat.4", 8))
    tok = torch.from_numpy(synth.randint("in.err.tok", (1, 40), 0, 6561)).to(torch.int32)
    with pytest.raises(FyError):                                           # 2 * (4 + 40) = 88 frames > max_frames 64
        flow.inference(tok, [40], ptok, [4], pfeat, [8], emb, noise)
    with pytest.raises(FyError):                                           # an utterance without tokens
        flow.inference(tok[:, :8], [0], ptok, [4], pfeat, [8], emb, noise)
    with pytest.raises(FyError):                                           # finalize=False needs more than the look-ahead
        flow.inference(tok[:, :8], [3], ptok, [4], pfeat, [8], emb, noise, streaming=True, finalize=False)
    with pytest.raises(FyError, match="batch"):
        flow.inference(tok[:, :8].repeat(3, 1), [8] * 3, ptok.repeat(3, 1), [4] * 3, pfeat.repeat(3, 1, 1), [8] * 3, emb.repeat(3, 1), noise)
    mel = flow.inference(tok[:, :8], [8], ptok, [4], pfeat, [8], emb, noise)     # still fine afterwards
    assert mel.shape == (1, 80, 16) and bool(torch.isfinite(mel).all())


def test_hift_limits(engines):
    cfg, _, _, hift = engines
    ri = torch.from_numpy(synth.hift_rand_ini()).to(DEV)
    sn = torch.from_numpy(synth.hift_sine_noise(64 * 480)).to(DEV)
    with pytest.raises(FyError):                                           # 48 frames > max_frames 40
        hift.inference(torch.zeros(1, 80, 48, device=DEV), ri, sn)
    with pytest.raises(FyError, match="streaming chunk"):
        hift.inference(torch.zeros(1, 80, 6, device=DEV), ri, sn, finalize=False)
    with pytest.raises(FyError):
        hift.inference(torch.zeros(3, 80, 8, device=DEV), ri, sn)
    wav, _ = hift.inference(torch.zeros(1, 80, 8, device=DEV), ri, sn)
    assert wav.shape[1] == 8 * 480 and bool(torch.isfinite(wav).all())
