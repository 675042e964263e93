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


def speech16k(seconds, seed=0):
    g = torch.Generator().manual_seed(seed)
    n = int(16000 * seconds)
    t = torch.arange(n) / 16000.0
    x = 0.3 * torch.sin(2 * torch.pi * 180 * t) + 0.1 * torch.sin(2 * torch.pi * 2600 * t) + 0.05 * (torch.rand(n, generator=g) * 2 - 1)
    return (x * torch.linspace(0.3, 1.0, n)).unsqueeze(0)


@pytest.mark.parametrize("seconds", [0.3, 2.0, 10.0, 29.9, 1.2345])
def test_whisper_log_mel_against_oracle(seconds):
    """The speech tokenizer's input (cli/frontend.py:97) on the GPU against the restated whisper.log_mel_spectrogram
    (oracle/frontend.py: torch.stft as whisper calls it; parity unpinned for the library itself): (1, 128, S // 160), the normalised
    log10 values within 1e-3 (they span a range of 2)."""
    from fangyan_tts_amd.cli.frontend import AudioFeat
    from oracle import frontend as ofe
    y = speech16k(seconds)
    ref = ofe.whisper_log_mel(y)
    got = AudioFeat("whisper", DEV)(y).cpu()
    assert got.shape == ref.shape == (1, 128, y.shape[1] // 160)
    e = maxerr(got, ref)
    note("parity_frontend.json", f"whisper_log_mel.{seconds}s.max_abs_err", e)
    assert e < 1e-3, e


@pytest.mark.parametrize("seconds", [0.03, 2.0, 10.0, 1.2345])
def test_kaldi_fbank_against_oracle(seconds):
    """The speaker embedder's input (cli/frontend.py:111-115) on the GPU against the restated kaldi.fbank (oracle/frontend.py:
    torch.fft.rfft as torchaudio calls it): (frames, 80) log mel energies within 2e-3, with and without the mean over frames."""
    from fangyan_tts_amd.cli.frontend import AudioFeat
    from oracle import frontend as ofe
    y = speech16k(seconds, seed=3)
    ref = ofe.kaldi_fbank(y)
    fb = AudioFeat("fbank", DEV)
    got = fb(y).cpu()
    assert got.shape == ref.shape == (1 + (y.shape[1] - 400) // 160, 80)
    e = maxerr(got, ref)
    note("parity_frontend.json", f"kaldi_fbank.{seconds}s.max_abs_err_log_domain", e)
    assert e < 2e-3, e
    got_m = fb(y, subtract_mean=True).cpu()
    assert maxerr(got_m, ref - ref.mean(dim=0, keepdim=True)) < 2e-3
    with pytest.raises(ValueError, match="too few"):
        fb(torch.zeros(1, 399))


def test_onnx_sessions_are_built_only_when_they_can_be(tmp_path):
    """AutoModel builds the reference's two ONNX sessions itself when onnxruntime imports and the files are in the model directory;
    otherwise the callables stay injectable and the first prompt-wav call says what is missing."""
    from fangyan_tts_amd.cli.frontend import onnx_prompt_models
    try:
        import onnxruntime  # noqa: F401
        have = True
    except Exception:
        have = False
    tok, emb = onnx_prompt_models(str(tmp_path), DEV)          # no .onnx files there
    assert tok is None and emb is None
    if not have:
        from fangyan_tts_amd.cli.frontend import CosyVoiceFrontEnd
        fe = CosyVoiceFrontEnd(Tok(), device=DEV)
        with pytest.raises(NotImplementedError, match="onnxruntime"):
            fe.speech_tokenizer(torch.zeros(1, 16000))
