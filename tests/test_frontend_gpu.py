"""GPU tests of the frontend: the prompt-mel HIP kernel (csrc/frontend.hip, through the C ABI) against the oracle - whose STFT
is torch.stft as the reference calls it (matcha/utils/audio.py:64-76) - and the facade running inference_instruct2 /
inference_zero_shot from a prompt WAVEFORM through `CosyVoiceFrontEnd` (GPU mel, injected tokenizer / speech tokenizer /
x-vector callables in place of the files and ONNX models that are not in this image)."""
import numpy as np
import pytest
import torch

from fangyan_tts_amd import synth
from fangyan_tts_amd.spec import ModelCfg
from gpu_util import maxerr, note

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def prompt(seconds, seed=0):
    g = torch.Generator().manual_seed(seed)
    n = int(24000 * seconds)
    t = torch.arange(n) / 24000.0
    x = 0.3 * torch.sin(2 * torch.pi * 220 * t) + 0.1 * torch.sin(2 * torch.pi * 3100 * t) + 0.05 * (torch.rand(n, generator=g) * 2 - 1)
    return (x * torch.linspace(0.2, 1.0, n)).unsqueeze(0)


@pytest.mark.parametrize("seconds", [0.5, 3.0, 10.0, 1.2345])
def test_prompt_mel_against_oracle(seconds):
    from fangyan_tts_amd.cli.frontend import PromptMel
    from oracle import frontend as ofe
    y = prompt(seconds)
    ref = ofe.mel_spectrogram(y)
    got = PromptMel(24000, DEV)(y).cpu()
    assert got.shape == ref.shape == (1, 80, (y.shape[1] + 1440 - 1920) // 480 + 1)
    e = maxerr(got, ref)
    note("parity_frontend.json", f"prompt_mel.{seconds}s.max_abs_err_log_domain", e)
    assert e < 2e-3, e                      # log-mel values span [-11.5, 3]; fp32 1920-term DFT sums


def test_prompt_mel_errors():
    from fangyan_tts_amd.cli.frontend import PromptMel
    pm = PromptMel(24000, DEV)
    with pytest.raises(RuntimeError, match="reflect padding"):
        pm(torch.zeros(1, 500))


class Tok:
    def encode(self, text, allowed_special="all"):
        return [(ord(c) * 7919) % 1000 for c in text]


def test_facade_with_the_real_frontend_class(tmp_path):
    """AutoModel(model_dir, tokenizer=..., speech_tokenizer=..., spk_embedder=...): inference_instruct2 / inference_zero_shot from a
    prompt waveform - text split, prompt mel on the GPU, two frames per token, dicts - equals the model driven with the same
    dict built by hand."""
    from test_facade_gpu import YAML
    from cosyvoice.cli.cosyvoice import AutoModel
    from oracle import frontend as ofe
    cfg = ModelCfg.tiny()
    (tmp_path / "cosyvoice3.yaml").write_text(YAML)
    for name, m in (("llm", cfg.llm), ("flow", cfg.flow), ("hift", cfg.hift)):
        torch.save({k: torch.from_numpy(v) for k, v in synth.state_dict(m.manifest()).items()}, tmp_path / f"{name}.pt")
    wav = prompt(1.0, seed=5)
    spk = torch.from_numpy(synth.normal("fe.spk.real", (1, 192)))
    stok = lambda s16: [int(v) for v in synth.randint("fe.stok.real", (1, 30), 0, 6561)[0]]       # 30 tokens for 1 s: cut to 25 = 50 frames / 2
    model = AutoModel(model_dir=str(tmp_path), tokenizer=Tok(), speech_tokenizer=stok, spk_embedder=lambda s16: spk, max_tokens=260,
                      max_prompt_tokens=64, sampler="greedy")
    outs = list(model.inference_instruct2("你好世界", "用四川话说<|endofprompt|>", (wav[0].numpy(), 24000)))
    assert len(outs) == 1 and outs[0]["tts_speech"].shape[1] % 480 == 0
    # the same model_input by hand, the mel from the oracle
    mel = ofe.mel_spectrogram(wav)[0].t().unsqueeze(0)[:, :50]
    tok = Tok()
    # text_normalize closes the sentence (split_paragraph appends the stop mark): the LM sees "你好世界。"
    inp = {"text": torch.tensor([tok.encode("你好世界。")], dtype=torch.int32), "prompt_text": torch.tensor([tok.encode("用四川话说<|endofprompt|>")], dtype=torch.int32),
           "flow_prompt_speech_token": torch.tensor([stok(None)[:25]], dtype=torch.int32), "prompt_speech_feat": mel, "flow_embedding": spk}
    w2, s2, _ = model.model.tts_batch([inp])
    assert w2[:, : s2[0]].shape == outs[0]["tts_speech"].shape
    e = maxerr(w2[:, : s2[0]], outs[0]["tts_speech"])
    note("parity_frontend.json", "facade.instruct2.wav_vs_hand_built_input", e)
    assert e < 2.5e-3          # the two prompt mels differ by the kernel's 1e-3 (above); same tolerance as the vocoder's
    z = list(model.inference_zero_shot("你好世界。今天天气不错！", "提示文本<|endofprompt|>", (wav[0].numpy(), 24000)))
    assert len(z) == 1 and z[0]["tts_speech"].shape[1] > 0
    # no tokenizer files in the model dir and none passed: the reason surfaces at the first call
    bare = AutoModel(model_dir=str(tmp_path), max_tokens=160, max_prompt_tokens=64)
    with pytest.raises(RuntimeError, match="CosyVoice-BlankEN"):
        list(bare.inference_instruct2("你好", "说<|endofprompt|>", (wav[0].numpy(), 24000)))
