"""CPU tests of the frontend: the restated text utilities against fixtures minted from the reference's own
functions (tests/golden/mint_frontend.py imports cosyvoice/utils/frontend_utils.py), the model_input dicts of
frontend_zero_shot / instruct2 / cross_lingual / sft / vc over injected callables, the token table, the oracle's prompt mel.

What stays unpinned, and why: the dict assembly cannot be minted by import (cli/frontend.py needs onnxruntime, whisper,
inflect, torchaudio - absent), so it is checked against the logic cited at cli/frontend.py:162-224; the Slaney mel
filterbank is restated from librosa's definition (librosa absent); load_wav's resampler is not torchaudio's."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from fangyan_tts_amd.cli import frontend as fe

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(G, "frontend_text.json"), encoding="utf-8") as f:
        return json.load(f)


def test_text_utilities_match_the_reference(fx):
    for i, t in enumerate(fx["texts"]):
        assert fe.contains_chinese(t) == fx["contains_chinese"][i], t
        assert fe.replace_corner_mark(t) == fx["replace_corner_mark"][i], t
        assert fe.remove_bracket(t) == fx["remove_bracket"][i], t
        assert fe.is_only_punctuation(t) == fx["is_only_punctuation"][i], t
        assert fe.spell_out_number(t, lambda s: "<" + s + ">") == fx["spell_out_number"][i], t
        want = fx["replace_blank"][i]
        if want is None:
            with pytest.raises(IndexError):
                fe.replace_blank(t)
        else:
            assert fe.replace_blank(t) == want, t


def test_split_paragraph_matches_the_reference(fx):
    tok = lambda s: s.split()
    assert len(fx["split"]) >= 100
    for c in fx["split"]:
        mx, mn, mg, comma = c["args"]
        if c["out"] is None:
            with pytest.raises(IndexError):
                fe.split_paragraph(c["text"], tok, c["lang"], token_max_n=mx, token_min_n=mn, merge_len=mg, comma_split=comma)
        else:
            got = fe.split_paragraph(c["text"], tok, c["lang"], token_max_n=mx, token_min_n=mn, merge_len=mg, comma_split=comma)
            assert got == c["out"], (c["text"][:30], c["lang"], c["args"])


def test_special_token_table(fx):
    toks = fe.cv3_special_tokens()
    assert len(toks) == fx["special_tokens"]["count"] == len(set(toks))
    assert hashlib.sha256("\n".join(toks).encode("utf-8")).hexdigest() == fx["special_tokens"]["sha256"]
    assert toks[:3] == fx["special_tokens"]["first"] and "<|endofprompt|>" in toks


class Tok:
    def encode(self, text, allowed_special="all"):
        return [ord(c) % 1000 for c in text]


def make(**kw):
    calls = {"wav": []}

    def loader(wav, sr):
        calls["wav"].append((wav, sr))
        n = {16000: 16000 * 3, 24000: 24000 * 3}[sr]
        return torch.zeros(1, n)
    f = fe.CosyVoiceFrontEnd(Tok(), feat_extractor=lambda s: torch.arange(80 * kw.get("frames", 151), dtype=torch.float32).reshape(1, 80, -1),
                             speech_tokenizer=lambda s: list(range(kw.get("tokens", 70))), spk_embedder=lambda s: torch.ones(1, 192) * 0.5,
                             device=torch.device("cpu"), wav_loader=loader)
    return f, calls


@pytest.mark.parametrize("frames,tokens,want", [(151, 70, 70), (120, 70, 60), (141, 70, 70), (10, 3, 3)])
def test_zero_shot_forces_two_frames_per_token(frames, tokens, want):
    """cli/frontend.py:174-178: token_len = min(int(feat_len / 2), n_tokens); feat cut to 2 token_len, tokens to token_len."""
    f, calls = make(frames=frames, tokens=tokens)
    d = f.frontend_zero_shot("你好", "提示", "p.wav", 24000, "")
    assert set(d) == {"prompt_text", "prompt_text_len", "llm_prompt_speech_token", "llm_prompt_speech_token_len", "flow_prompt_speech_token",
                      "flow_prompt_speech_token_len", "prompt_speech_feat", "prompt_speech_feat_len", "llm_embedding", "flow_embedding", "text", "text_len"}
    assert d["prompt_speech_feat"].shape == (1, 2 * want, 80) and int(d["prompt_speech_feat_len"]) == 2 * want
    assert d["flow_prompt_speech_token"].shape == (1, want) and int(d["flow_prompt_speech_token_len"]) == want
    assert d["llm_prompt_speech_token"] is d["flow_prompt_speech_token"]
    assert d["text"].tolist() == [[ord(c) % 1000 for c in "你好"]] and int(d["text_len"]) == 2 and d["text"].dtype == torch.int32
    # the feature is (1, F, 80): the extractor's (1, 80, F) transposed (frontend.py:121)
    assert float(d["prompt_speech_feat"][0, 1, 0]) == 1.0 and float(d["prompt_speech_feat"][0, 0, 1]) == float(frames)
    assert [sr for _, sr in calls["wav"]] == [24000, 16000, 16000]
    # at another resample rate nothing is cut (CosyVoice 1)
    d = f.frontend_zero_shot("你好", "提示", "p.wav", 22050, "")
    assert d["prompt_speech_feat"].shape[1] == frames and d["flow_prompt_speech_token"].shape[1] == tokens


def test_instruct2_cross_lingual_sft_vc_dicts():
    f, _ = make()
    z = f.frontend_zero_shot("你好", "提示", "p.wav", 24000, "")
    i2 = f.frontend_instruct2("你好", "用四川话说<|endofprompt|>", "p.wav", 24000, "")
    assert set(z) - set(i2) == {"llm_prompt_speech_token", "llm_prompt_speech_token_len"}            # frontend.py:209-213
    assert i2["prompt_text"].shape[1] == len("用四川话说<|endofprompt|>")
    cl = f.frontend_cross_lingual("你好", "p.wav", 24000, "")
    assert set(z) - set(cl) == {"prompt_text", "prompt_text_len", "llm_prompt_speech_token", "llm_prompt_speech_token_len"}
    f.spk2info["spk"] = {"embedding": torch.ones(1, 192)}
    s = f.frontend_sft("你好", "spk")
    assert set(s) == {"text", "text_len", "llm_embedding", "flow_embedding"}
    ins = f.frontend_instruct("你好", "spk", "慢一点")
    assert "llm_embedding" not in ins and "prompt_text" in ins
    vc = f.frontend_vc("src.wav", "p.wav", 24000)
    assert set(vc) == {"source_speech_token", "source_speech_token_len", "flow_prompt_speech_token", "flow_prompt_speech_token_len",
                       "prompt_speech_feat", "prompt_speech_feat_len", "flow_embedding"}
    # a stored speaker (add_zero_shot_spk) replaces the prompt extraction
    f.spk2info["me"] = {k: v for k, v in z.items() if k not in ("text", "text_len")}
    z2 = f.frontend_zero_shot("再见", "", "", 24000, "me")
    assert z2["prompt_speech_feat"] is z["prompt_speech_feat"] and z2["text"].shape[1] == 2


def test_text_normalize_pipeline():
    f, _ = make()
    assert f.text_normalize("<|en|>keep as is", split=True) == ["<|en|>keep as is"]            # control symbols: untouched
    assert f.text_normalize("", split=False) == ""
    out = f.text_normalize(" 今天 天气不错.  我们走吧,、", split=True)
    assert out == ["今天天气不错。我们走吧。"]
    long = "这是一个比较长的句子用来测试分段逻辑是否正确。" * 9
    segs = f.text_normalize(long, split=True)
    assert "".join(segs) == long and len(segs) > 1 and all(len(s) <= 80 + 23 for s in segs)
    f.number_speller = lambda s: "N"
    assert f.text_normalize("I have 25 apples", split=True) == ["I have N apples."]
    assert f.text_normalize("。。。", split=True) == []                                         # only punctuation: dropped
    g = (x for x in ["a", "b"])
    assert f.text_normalize(g) == [g]


def test_missing_pieces_fail_loudly():
    f = fe.CosyVoiceFrontEnd(Tok(), feat_extractor=lambda s: torch.zeros(1, 80, 10), device=torch.device("cpu"),
                             wav_loader=lambda w, sr: torch.zeros(1, sr))
    with pytest.raises(NotImplementedError, match="speech tokenizer"):
        f.frontend_zero_shot("你好", "提示", "p.wav", 24000, "")
    with pytest.raises(FileNotFoundError):
        fe.load_qwen_tokenizer("/nonexistent")


def test_load_wav(tmp_path):
    from scipy.io import wavfile
    sr = 48000
    t = np.arange(sr) / sr
    stereo = np.stack([np.sin(2 * np.pi * 440 * t), np.zeros_like(t)], axis=1)
    wavfile.write(tmp_path / "a.wav", sr, (stereo * 32767).astype(np.int16))
    x = fe.load_wav(str(tmp_path / "a.wav"), 24000)
    assert x.shape == (1, 24000) and x.dtype == torch.float32
    ref = 0.5 * np.sin(2 * np.pi * 440 * np.arange(24000) / 24000)                              # channel mean, half rate
    assert float(np.abs(x[0, 100:-100].numpy() - ref[100:-100]).max()) < 2e-3
    y = fe.load_wav((np.ones(16000, dtype=np.float32), 16000), 16000)
    assert y.shape == (1, 16000)
    with pytest.raises(AssertionError):
        fe.load_wav((np.ones(8000, dtype=np.float32), 8000), 16000)


def test_oracle_prompt_mel():
    """The oracle's mel: frames = S / 480, the STFT is torch.stft; the restated Slaney filterbank has the properties of librosa's
    (triangles on a mel-spaced grid, unit-area normalisation 2 / bandwidth, linear below 1 kHz)."""
    from oracle import frontend as ofe
    fb = ofe.slaney_mel_filterbank(24000, 1920, 80)
    assert fb.shape == (80, 961) and fb.dtype == np.float32 and (fb >= 0).all()
    peaks = fb.argmax(axis=1)
    assert (np.diff(peaks) > 0).all() and fb[:, 0].sum() == 0.0
    hz = peaks * 12.5
    lin = hz[hz < 900]
    assert np.allclose(np.diff(lin), np.diff(lin)[0], atol=12.5)                                 # equally spaced below 1 kHz
    area = fb.sum(axis=1) * 12.5                                                                  # integral over Hz of each triangle = 1 (slaney norm)
    assert np.allclose(area, 1.0, atol=0.08)
    g = torch.Generator().manual_seed(3)
    y = (torch.rand(1, 24000 * 2, generator=g) * 2 - 1) * 0.3
    m = ofe.mel_spectrogram(y)
    assert m.shape == (1, 80, 100) and torch.isfinite(m).all()
    tone = 0.5 * torch.sin(2 * torch.pi * 1000.0 * torch.arange(48000) / 24000.0).unsqueeze(0)
    mt = ofe.mel_spectrogram(tone)
    assert int(mt[0, :, 50].argmax()) == int(np.abs(hz - 1000).argmin())                         # a 1 kHz tone lands in the 1 kHz filter


# ---- load_wav's resampler: torchaudio.transforms.Resample's default filter bank, restated (cli/frontend.py:resample) --------
@pytest.mark.parametrize("orig,new", [(16000, 24000), (44100, 16000), (48000, 24000), (22050, 24000)])
def test_resample_equals_the_direct_evaluation_of_the_formula(orig, new):
    """The filter-bank / strided-convolution form against a sample-by-sample evaluation of torchaudio's published formula
    (oracle/frontend.py:resample_direct), on a chirp: same length (ceil(new S / orig)) and the same samples to fp32 rounding."""
    import numpy as np
    import torch
    from fangyan_tts_amd.cli.frontend import resample
    from oracle.frontend import resample_direct
    S = 1501
    t = np.arange(S) / orig
    x = (0.6 * np.sin(2 * np.pi * (200.0 + 1500.0 * t) * t) + 0.2 * np.sin(2 * np.pi * 3100.0 * t)).astype(np.float32)
    y = resample(torch.from_numpy(x)[None], orig, new)
    ref = resample_direct(x, orig, new)
    assert y.shape == (1, -(-new * S // orig))
    assert float(np.abs(y[0].numpy().astype(np.float64) - ref).max()) < 2e-6


def test_resample_keeps_a_tone_and_load_wav_uses_it(tmp_path):
    """A 440 Hz tone at 44.1 kHz comes out as the same tone at 16 kHz (interior samples, < 2e-3: the filter's pass-band ripple),
    through load_wav on a PCM16 file: mono mean, int16 scale, then the windowed-sinc resampler."""
    import numpy as np
    from scipy.io import wavfile
    from fangyan_tts_amd.cli.frontend import load_wav
    sr, S = 44100, 22050
    t = np.arange(S) / sr
    tone = 0.5 * np.sin(2 * np.pi * 440.0 * t)
    pcm = np.stack([tone, tone], axis=1)
    path = str(tmp_path / "tone.wav")
    wavfile.write(path, sr, (pcm * 32767.0).astype(np.int16))
    y = load_wav(path, 16000)
    n = -(-16000 * S // sr)
    assert y.shape == (1, n)
    want = 0.5 * np.sin(2 * np.pi * 440.0 * np.arange(n) / 16000.0)
    assert float(np.abs(y[0].numpy()[200:-200] - want[200:-200]).max()) < 2e-3
    same = load_wav(path, 44100)
    assert same.shape == (1, S)


def test_whisper_and_fbank_oracles_are_sane():
    """The two restated 16 kHz feature front ends: shapes and the fixed points their definitions imply.  whisper: S // 160 frames of
    128 bins, values within [(max - 8 + 4) / 4, (max + 4) / 4]; a 1 kHz tone peaks in the mel band around 1 kHz.  kaldi fbank:
    1 + (S - 400) // 160 frames of 80 bins, silence gives log(float32 eps) everywhere, the filterbank's Nyquist column is zero and
    every triangle peaks at 1 or below."""
    import numpy as np
    import torch
    from oracle.frontend import kaldi_fbank, kaldi_mel_banks, slaney_mel_filterbank, whisper_log_mel
    S = 16000
    tone = (0.5 * torch.sin(2 * np.pi * 1000.0 * torch.arange(S) / 16000.0))[None]
    w = whisper_log_mel(tone)
    assert w.shape == (1, 128, S // 160)
    assert float(w.max() - w.min()) <= 2.0 + 1e-6
    fb = slaney_mel_filterbank(16000, 400, 128)
    centre = int(np.argmax(fb[:, 25]))                          # bin 25 = 1000 Hz
    assert abs(int(w[0, :, 50].argmax()) - centre) <= 1
    f = kaldi_fbank(torch.zeros(1, 4000))
    assert f.shape == (1 + (4000 - 400) // 160, 80)
    assert torch.allclose(f, torch.full_like(f, float(np.log(np.finfo(np.float32).eps))))
    kb = kaldi_mel_banks()
    assert kb.shape == (80, 257) and float(kb[:, 256].max()) == 0.0 and float(kb.max()) <= 1.0
    ft = kaldi_fbank(tone)
    assert ft.shape == (98, 80) and int(ft.mean(dim=0).argmax()) == int(np.argmax(kb[:, 32]))      # bin 32 = 1000 Hz
