"""GPU test of the user-facing facade: AutoModel(model_dir) -> inference_instruct2 / zero_shot generators,
model.model.llm.load_state_dict(..., strict=False), model.sample_rate - the calls compare_inference.py makes
(compare_inference.py:29-61) - on a synthetic model directory with a stand-in frontend."""
import os

import numpy as np
import pytest
import torch

from fangyan_tts_amd import synth
from fangyan_tts_amd.spec import ModelCfg
from gpu_util import maxerr, synth_mel

pytestmark = pytest.mark.gpu

YAML = """
# scalars only matter; tags are HyperPyYAML's
sample_rate: 24000
llm_input_size: 256
chunk_size: 25
token_mel_ratio: 2
flow: !new:cosyvoice.flow.flow.CausalMaskedDiffWithDiT
    vocab_size: 6561
    decoder: !new:cosyvoice.flow.flow_matching.CausalConditionalCFM
        cfm_params: !new:omegaconf.DictConfig
            content:
                inference_cfg_rate: 0.7
hift: !new:cosyvoice.hifigan.generator.CausalHiFTGenerator
    upsample_rates: [8, 5, 3]
    nsf_voiced_threshold: 10
"""


class FakeFrontEnd:
    """Stands in for cli/frontend.py: deterministic ids / features from the strings."""

    def __init__(self, cfg):
        self.cfg, self.spk2info = cfg, {}

    def text_normalize(self, text, split=True, text_frontend=True):
        return [text] if split else text

    def _ids(self, text, n):
        return torch.from_numpy(synth.randint("fe." + text, (1, n), 0, self.cfg.llm.vocab))

    def frontend_zero_shot(self, tts_text, prompt_text, prompt_wav, resample_rate, zero_shot_spk_id):
        p = 12
        tok = torch.from_numpy(synth.randint("fe.tok." + prompt_wav, (1, p), 0, 6561))
        feat = torch.from_numpy(synth_mel("fe.feat." + prompt_wav, 2 * p))
        emb = torch.from_numpy(synth.normal("fe.spk." + prompt_wav, (1, 192)))
        return {"text": self._ids(tts_text, 7), "text_len": torch.tensor([7]), "prompt_text": self._ids(prompt_text, 5),
                "prompt_text_len": torch.tensor([5]), "llm_prompt_speech_token": tok, "llm_prompt_speech_token_len": torch.tensor([p]),
                "flow_prompt_speech_token": tok, "flow_prompt_speech_token_len": torch.tensor([p]), "prompt_speech_feat": feat,
                "prompt_speech_feat_len": torch.tensor([2 * p]), "llm_embedding": emb, "flow_embedding": emb}

    def frontend_sft(self, tts_text, spk_id):                               # cli/frontend.py:162-166
        emb = torch.from_numpy(synth.normal("fe.spk." + spk_id, (1, 192)))
        return {"text": self._ids(tts_text, 7), "text_len": torch.tensor([7]), "llm_embedding": emb, "flow_embedding": emb}

    def frontend_vc(self, source_speech_16k, prompt_wav, resample_rate):    # cli/frontend.py:215-224
        d = self.frontend_zero_shot("x", "y", prompt_wav, resample_rate, "")
        src = torch.from_numpy(synth.randint("fe.src." + source_speech_16k, (1, 20), 0, 6561))
        return {"source_speech_token": src, "source_speech_token_len": torch.tensor([20]),
                "flow_prompt_speech_token": d["flow_prompt_speech_token"], "flow_prompt_speech_token_len": d["flow_prompt_speech_token_len"],
                "prompt_speech_feat": d["prompt_speech_feat"], "prompt_speech_feat_len": d["prompt_speech_feat_len"],
                "flow_embedding": d["flow_embedding"]}

    def frontend_instruct2(self, tts_text, instruct_text, prompt_wav, resample_rate, zero_shot_spk_id):
        d = self.frontend_zero_shot(tts_text, instruct_text, prompt_wav, resample_rate, zero_shot_spk_id)
        del d["llm_prompt_speech_token"], d["llm_prompt_speech_token_len"]
        return d


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("cv3")
    cfg = ModelCfg.tiny()
    (d / "cosyvoice3.yaml").write_text(YAML)
    for name, m in (("llm", cfg.llm), ("flow", cfg.flow), ("hift", cfg.hift)):
        sd = {k: torch.from_numpy(v) for k, v in synth.state_dict(m.manifest()).items()}
        if name == "hift":
            sd = {"generator." + k: v for k, v in sd.items()}             # as saved by the GAN trainer, cli/model.py:71
        if name == "llm":
            sd["epoch"] = torch.tensor(3)                                   # train bookkeeping must be ignored
        torch.save(sd, d / f"{name}.pt")
    return str(d), cfg


def test_automodel_drop_in_calls(model_dir):
    from cosyvoice.cli.cosyvoice import AutoModel                            # the reference's import path
    path, cfg = model_dir
    model = AutoModel(model_dir=path, frontend=FakeFrontEnd(cfg), max_tokens=160, max_prompt_tokens=32, sampler="greedy")
    assert model.model.sampler == "greedy"        # the facade's default is the reference's ("ras"); greedy makes the calls below comparable
    assert model.sample_rate == 24000 and model.cfg == cfg
    outs = list(model.inference_instruct2("你好世界", "用四川话说<|endofprompt|>", "prompt.wav", stream=False))
    assert len(outs) == 1
    wav = outs[0]["tts_speech"]
    assert wav.dtype == torch.float32 and wav.device.type == "cpu" and wav.dim() == 2 and wav.shape[0] == 1
    assert wav.shape[1] % 480 == 0 and float(wav.abs().max()) <= 0.99 + 1e-6
    # same call through the batched engine entry
    fe = FakeFrontEnd(cfg)
    w2, s2, _ = model.model.tts_batch([fe.frontend_instruct2("你好世界", "用四川话说<|endofprompt|>", "prompt.wav", 24000, "")])
    assert maxerr(w2[:, : s2[0]], wav) == 0.0
    z = list(model.inference_zero_shot("你好世界", "提示<|endofprompt|>", "prompt.wav"))[0]["tts_speech"]
    assert z.shape[1] % 480 == 0
    # no prompt at all (inference_sft) and no LM at all (inference_vc: 20 source tokens -> 40 frames)
    s = list(model.inference_sft("你好世界", "speaker0"))[0]["tts_speech"]
    assert s.shape[1] % 480 == 0 and s.shape[1] > 0
    v = list(model.inference_vc("source.wav", "prompt.wav"))[0]["tts_speech"]
    assert v.shape == (1, 20 * 2 * 480)
    vs = torch.cat([o["tts_speech"] for o in model.inference_vc("source.wav", "prompt.wav", stream=True)], dim=1)
    assert vs.shape == v.shape
    # compare_inference.py:36-43: swap LLM weights, strict=False, extra keys ignored
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict(cfg.llm.manifest()).items() if "layers.1." in k}
    sd = {k: v * 0.5 for k, v in sd.items()}
    sd["step"] = torch.tensor(1)
    sd = {k: v for k, v in sd.items() if not k.startswith("epoch") and not k.startswith("step")}
    model.model.llm.load_state_dict(sd, strict=False)
    wav3 = list(model.inference_instruct2("你好世界", "用四川话说<|endofprompt|>", "prompt.wav"))[0]["tts_speech"]
    assert wav3.shape != wav.shape or maxerr(wav3, wav) > 1e-4                # the swapped weights are really in use
    with pytest.raises(RuntimeError):
        model.model.llm.load_state_dict(sd, strict=True)
    # a text generator (inference_bistream) fails for CosyVoice3 in the reference with this AttributeError (llm.py:545)
    with pytest.raises(AttributeError, match="llm_embedding"):
        list(model.model.tts(text=(torch.zeros(1, 1, dtype=torch.int32) for _ in range(3)), flow_embedding=torch.zeros(1, 192)))


def test_facade_against_the_oracle(model_dir):
    """inference_instruct2 / inference_zero_shot through the facade (checkpoint files on disk -> engines) against the CPU oracle
    fed the same model_input and the same state dicts: ids exact, mel and waveform within the engines' stated tolerances
    (tests/test_e2e_gpu.py) - the checkpoint ingestion (generator. prefix, weight-norm folding, shape-inferred sizes) is part
    of what is compared."""
    from cosyvoice.cli.cosyvoice import AutoModel
    from oracle import hift as ohift, pipeline as opipe
    path, cfg = model_dir
    fe = FakeFrontEnd(cfg)
    model = AutoModel(model_dir=path, frontend=fe, max_tokens=160, max_prompt_tokens=32, sampler="greedy")
    m = model.model
    sds = [{k: torch.from_numpy(v) for k, v in synth.state_dict(x.manifest()).items()} for x in (cfg.llm, cfg.flow, cfg.hift)]
    PH = ohift.prepare({k: v.numpy() for k, v in sds[2].items()})
    noise, ri, sn = m.rand_noise.cpu(), m.rand_ini.cpu(), m.sine_noise.cpu()
    for call, inp in ((lambda: model.inference_instruct2("你好世界", "用四川话说<|endofprompt|>", "prompt.wav"),
                       fe.frontend_instruct2("你好世界", "用四川话说<|endofprompt|>", "prompt.wav", 24000, "")),
                      (lambda: model.inference_zero_shot("你好世界", "提示<|endofprompt|>", "prompt.wav"),
                       fe.frontend_zero_shot("你好世界", "提示<|endofprompt|>", "prompt.wav", 24000, ""))):
        wav = list(call())[0]["tts_speech"]
        ref = opipe.tts(inp, sds[0], sds[1], PH, cfg, noise, ri, sn)
        assert wav.shape == ref["tts_speech"].shape                      # same token count (ids drive the length)
        mel = m.last_mel.cpu()
        assert maxerr(mel, ref["mel"]) < 4e-2
        ref_wav, _ = ohift.inference(mel, PH, cfg.hift, ri, sn[:, : wav.shape[1]])
        assert maxerr(wav, ref_wav) < 2.5e-3
        # fp32-class mode: the whole waveform against the oracle's
        from fangyan_tts_amd._lib import FY_DIRECT, FY_PRECISE
        m.flow_flags, m.hift_flags = FY_PRECISE, FY_DIRECT
        try:
            wav_p = list(call())[0]["tts_speech"]
        finally:
            m.flow_flags, m.hift_flags = 0, 0
        assert maxerr(wav_p, ref["tts_speech"]) < 1e-4


def test_concurrent_tts_calls(model_dir):
    """Two threads inside tts() on one model object (the reference's contract: runtime/python/grpc/server.py:68-69,
    cli/model.py:330-333) both get the audio a lone call gets - with one engine set (calls take turns) and with
    concurrency=2 (two engine sets, both calls run at once)."""
    import threading
    from cosyvoice.cli.cosyvoice import AutoModel
    path, cfg = model_dir
    fe = FakeFrontEnd(cfg)
    jobs = [("你好世界", "用四川话说<|endofprompt|>", "prompt.wav"), ("今天天气不错", "用粤语说<|endofprompt|>", "other.wav")]
    for conc in (1, 2):
        model = AutoModel(model_dir=path, frontend=fe, max_tokens=160, max_prompt_tokens=32, sampler="greedy", concurrency=conc)
        alone = [list(model.inference_instruct2(*j))[0]["tts_speech"] for j in jobs]
        got, errs = [None, None], []

        def work(i):
            try:
                for _ in range(3):
                    got[i] = list(model.inference_instruct2(*jobs[i]))[0]["tts_speech"]
            except BaseException as e:            # noqa: BLE001
                errs.append(e)
        ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs
        for i in range(2):
            assert got[i].shape == alone[i].shape and maxerr(got[i], alone[i]) == 0.0, (conc, i)


def test_capacity_errors_come_up_front(model_dir):
    from cosyvoice.cli.cosyvoice import AutoModel
    path, cfg = model_dir
    fe = FakeFrontEnd(cfg)
    model = AutoModel(model_dir=path, frontend=fe, max_tokens=100, max_prompt_tokens=32, sampler="greedy")
    with pytest.raises(ValueError, match="max_tokens"):         # 7 text ids -> up to 140 tokens > 100
        list(model.inference_instruct2("你好世界", "用四川话说<|endofprompt|>", "prompt.wav"))
    model = AutoModel(model_dir=path, frontend=fe, max_tokens=160, max_prompt_tokens=8, sampler="greedy")
    with pytest.raises(ValueError, match="max_prompt_tokens"):
        list(model.inference_instruct2("你好世界", "用四川话说<|endofprompt|>", "prompt.wav"))


def test_automodel_errors(tmp_path):
    from fangyan_tts_amd.cli.cosyvoice import AutoModel
    with pytest.raises(ValueError):
        AutoModel(model_dir=str(tmp_path / "nope"))
    with pytest.raises(TypeError):
        AutoModel(model_dir=str(tmp_path))


def test_load_state_dict_reaches_every_lane(model_dir):
    """compare_inference.py:42 swaps a fine-tuned LM in through `model.model.llm.load_state_dict`.  With concurrency=2 (two engine
    sets) BOTH lanes must decode with the new weights: two threads inside tts() at once get the audio a one-lane model gets
    after the same swap."""
    import threading
    from cosyvoice.cli.cosyvoice import AutoModel
    path, cfg = model_dir
    fe = FakeFrontEnd(cfg)
    job = ("你好世界", "用四川话说<|endofprompt|>", "prompt.wav")
    sd = {k: torch.from_numpy(v) * 0.5 for k, v in synth.state_dict(cfg.llm.manifest()).items() if "layers.1." in k}
    ref_model = AutoModel(model_dir=path, frontend=fe, max_tokens=160, max_prompt_tokens=32, sampler="greedy")
    before = list(ref_model.inference_instruct2(*job))[0]["tts_speech"]
    ref_model.model.llm.load_state_dict(sd, strict=False)
    after = list(ref_model.inference_instruct2(*job))[0]["tts_speech"]
    assert after.shape != before.shape or maxerr(after, before) > 1e-4
    model = AutoModel(model_dir=path, frontend=fe, max_tokens=160, max_prompt_tokens=32, sampler="greedy", concurrency=2)
    model.model.llm.load_state_dict(sd, strict=False)
    got, errs = [None, None], []
    gate = threading.Barrier(2)

    def work(i):
        try:
            gate.wait()                            # both inside tts() together: one call per lane
            got[i] = list(model.inference_instruct2(*job))[0]["tts_speech"]
        except BaseException as e:                # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i in range(2):
        assert got[i].shape == after.shape and maxerr(got[i], after) == 0.0, i
    # and each lane on its own, whichever the queue hands out: every LM handle of the model was rebuilt
    for ln in model.model.lanes:
        out, out_n, _ = ln.llm.generate([[5, 6, 7, 8]], [[1, 2]], [[]], max_len=[12])
        ref, ref_n, _ = ref_model.model.llm.generate([[5, 6, 7, 8]], [[1, 2]], [[]], max_len=[12])
        assert torch.equal(out.cpu(), ref.cpu()) and torch.equal(out_n.cpu(), ref_n.cpu())


def test_abandoned_stream_generator_says_so(model_dir):
    """A stream=True generator that is neither exhausted nor closed keeps the model's only lane; the next call on the same thread
    must say so instead of waiting forever, and closing the generator frees the lane."""
    from cosyvoice.cli.cosyvoice import AutoModel
    path, cfg = model_dir
    fe = FakeFrontEnd(cfg)
    model = AutoModel(model_dir=path, frontend=fe, max_tokens=160, max_prompt_tokens=32, sampler="greedy")
    job = ("你好世界今天天气不错", "用四川话说<|endofprompt|>", "prompt.wav")
    whole = list(model.inference_instruct2(*job))[0]["tts_speech"]
    gen = model.inference_instruct2(*job, stream=True)
    first = next(gen)["tts_speech"]
    assert first.shape[1] > 0
    with pytest.raises(RuntimeError, match="every lane of this model is held by this thread"):
        list(model.inference_instruct2(*job))
    gen.close()
    again = list(model.inference_instruct2(*job))[0]["tts_speech"]
    assert again.shape == whole.shape and maxerr(again, whole) == 0.0
