"""Text / audio frontend with the reference's surface (`CosyVoiceFrontEnd`, CosyVoice/cosyvoice/cli/frontend.py:29-224):
text_normalize + split, and the `model_input` dicts of frontend_sft / zero_shot / cross_lingual / instruct2 / vc.

What the reference's frontend is made of, and what stands here:

  piece (reference)                                    here
  ---------------------------------------------------  ----------------------------------------------------------------
  text clean-up + split_paragraph (utils/               restated below; pinned by fixtures minted from the reference's
    frontend_utils.py:20-135, cli/frontend.py:127-160)  own functions (tests/golden/frontend_text.json)
  Qwen tokenizer + CosyVoice3's added special tokens    `load_qwen_tokenizer` (transformers.AutoTokenizer over
    (tokenizer/tokenizer.py:274-313)                    <model_dir>/CosyVoice-BlankEN + the token table) - or inject one
  prompt mel: matcha mel_spectrogram                    `PromptMel`: HIP kernel (csrc/frontend.hip) through the C ABI
    (matcha/utils/audio.py:45-82)
  speech tokens: whisper log-mel -> ORT                 `AudioFeat("whisper")`: whisper's 128-bin log-mel as a HIP kernel
    speech_tokenizer_v3.onnx (frontend.py:94-108)       (csrc/frontend_feats.hip, fy_audio_feat_*), fed to the ONNX session
                                                        `onnx_prompt_models(model_dir)` builds when onnxruntime imports and
                                                        the .onnx files are there; else an injected callable
                                                        `speech_tokenizer(wav16k) -> list[int]`.  onnxruntime is NOT in this
                                                        image: the session branch has never executed here (the algorithm of
                                                        whisper's log-mel is restated: parity unpinned, whisper is absent)
  x-vector: kaldi fbank -> ORT campplus.onnx            `AudioFeat("fbank")`: kaldi's 80-bin fbank minus its mean, same
    (frontend.py:110-117)                               kernel file, same session rule; else an injected callable
                                                        `spk_embedder(wav16k) -> (1, 192)` (torchaudio absent: unpinned)
  load_wav: torchaudio load + Resample                  `load_wav`: scipy wav reader + `resample`, torchaudio's default
    (utils/file_utils.py:44-50)                         windowed-sinc filter bank (Hann, width 6, rolloff 0.99) restated and
                                                        held to a sample-by-sample evaluation of the published formula
                                                        (parity with torchaudio itself unpinned: it is absent)
  number spelling (inflect), wetext / ttsfrd            injected `number_speller` / `text_normalizer`; absent = the
                                                        reference's own no-frontend fallback (frontend.py:73-75)

Everything from `model_input` on runs on the GPU engines (cli/model.py).
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Callable, Dict, Generator, List, Optional, Sequence

import numpy as np
import torch

_CJK = re.compile(r"[一-鿿]+")


# ---- text utilities (utils/frontend_utils.py) ------------------------------------------------------------------------
def contains_chinese(text: str) -> bool:                                  # :20-23
    return _CJK.search(text) is not None


def replace_corner_mark(text: str) -> str:                                # :26-30
    return text.replace("²", "平方").replace("³", "立方")


def remove_bracket(text: str) -> str:                                     # :33-39
    for ch in "（）【】`":
        text = text.replace(ch, "")
    return text.replace("——", " ")


def spell_out_number(text: str, number_to_words: Callable[[str], str]) -> str:     # :42-60
    """Every maximal run of str.isdigit() characters goes through `number_to_words` (inflect's in the reference)."""
    out, run = [], ""
    for ch in text:
        if ch.isdigit():
            run += ch
            continue
        if run:
            out.append(number_to_words(run))
            run = ""
        out.append(ch)
    if run:
        out.append(number_to_words(run))
    return "".join(out)


def replace_blank(text: str) -> str:                                      # :124-135
    """A blank survives only between two non-blank ASCII characters."""
    out = []
    for i, ch in enumerate(text):
        if ch != " ":
            out.append(ch)
            continue
        nxt, prv = text[i + 1], text[i - 1]        # as the reference: a trailing blank raises IndexError, a leading one looks at text[-1]
        if nxt.isascii() and nxt != " " and prv.isascii() and prv != " ":
            out.append(ch)
    return "".join(out)


def is_only_punctuation(text: str) -> bool:                               # :138-141
    import regex
    return regex.fullmatch(r"^[\p{P}\p{S}]*$", text) is not None


def split_paragraph(text: str, tokenize: Callable[[str], Sequence[int]], lang: str = "zh", token_max_n: int = 80, token_min_n: int = 60,
                    merge_len: int = 20, comma_split: bool = False) -> List[str]:               # :63-121
    """Sentences are cut after a stop mark (a closing quote right behind it stays with the sentence); consecutive sentences are
    packed into segments: a segment is closed when adding the next sentence would pass token_max_n and it already holds more
    than token_min_n; a last segment shorter than merge_len joins the one before.  Length = characters for zh, tokens else."""
    zh = lang == "zh"
    length = (lambda t: len(t)) if zh else (lambda t: len(tokenize(t)))
    stops = list("。？！；：、.?!;") if zh else list(".?!;:")
    if comma_split:
        stops += ["，", ","]
    if text[-1] not in stops:
        text += "。" if zh else "."
    sentences, start, i = [], 0, 0
    for i, ch in enumerate(text):
        if ch not in stops:
            continue
        if i > start:
            sentences.append(text[start: i + 1])
        if i + 1 < len(text) and text[i + 1] in ('"', "”"):
            # the reference pops the last sentence whatever it was (an IndexError when there is none)
            sentences[-1] = sentences[-1] + text[i + 1]
            start = i + 2
        else:
            start = i + 1
    segments, cur = [], ""
    for s in sentences:
        if length(cur + s) > token_max_n and length(cur) > token_min_n:
            segments.append(cur)
            cur = ""
        cur += s
    if cur:
        if segments and length(cur) < merge_len:
            segments[-1] += cur
        else:
            segments.append(cur)
    return segments


# ---- tokenizer ---------------------------------------------------------------------------------------------------------
_HERE = os.path.dirname(os.path.abspath(__file__))


def cv3_special_tokens() -> List[str]:
    """The additional special tokens CosyVoice3Tokenizer registers, in its order (tokenizer/tokenizer.py:274-313): a vocabulary
    table, kept as data (cv3_special_tokens.txt, one per line, written by tests/golden/mint_frontend.py)."""
    with open(os.path.join(_HERE, "cv3_special_tokens.txt"), encoding="utf-8") as f:
        return [line.rstrip("\n") for line in f if line.rstrip("\n")]


class QwenTokenizer:
    """CosyVoice3Tokenizer's behaviour (tokenizer/tokenizer.py:242-313): HF tokenizer + the added special tokens;
    encode(text, **kw) -> ids of `tokenizer([text])`."""

    def __init__(self, token_path: str, skip_special_tokens: bool = True):
        from transformers import AutoTokenizer
        self.tokenizer = AutoTokenizer.from_pretrained(token_path)
        self.special_tokens = {"eos_token": "<|endoftext|>", "pad_token": "<|endoftext|>", "additional_special_tokens": cv3_special_tokens()}
        self.tokenizer.add_special_tokens(self.special_tokens)
        self.skip_special_tokens = skip_special_tokens

    def encode(self, text, **kwargs):
        return self.tokenizer([text], return_tensors="pt")["input_ids"][0].cpu().tolist()

    def decode(self, tokens):
        return self.tokenizer.batch_decode([torch.tensor(tokens, dtype=torch.int64)], skip_special_tokens=self.skip_special_tokens)[0]


def load_qwen_tokenizer(model_dir: str) -> QwenTokenizer:
    path = os.path.join(model_dir, "CosyVoice-BlankEN")
    if not os.path.isdir(path):
        raise FileNotFoundError(f"{path} not found: the Qwen tokenizer files come with the model directory (cli/cosyvoice.py:201); "
                                "pass frontend=CosyVoiceFrontEnd(tokenizer=...) to use another tokenizer")
    return QwenTokenizer(path)


# ---- audio -------------------------------------------------------------------------------------------------------------
def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """The filter bank of torchaudio.transforms.Resample with its defaults (resampling_method "sinc_interp_hann",
    lowpass_filter_width 6, rolloff 0.99; torchaudio/functional/functional.py:_get_sinc_resample_kernel, torchaudio 2.x), restated
    from the published algorithm - torchaudio is not in the image, so this table is unpinned:
    with o = orig / gcd, n = new / gcd, f = rolloff * min(o, n), width = ceil(lpw * o / f), phase j in [0, n) and tap
    i in [-width, width + o):   t = clamp((-j / n + i / o) * f, -lpw, lpw);
    kernel[j][i] = sinc(pi t) * cos(pi t / (2 lpw))^2 * f / o     (float64, stored as float32).  Returns (kernels (n, taps), width, o, n)."""
    from math import ceil, gcd, pi
    g = gcd(int(orig_freq), int(new_freq))
    o, n = int(orig_freq) // g, int(new_freq) // g
    base = min(o, n) * rolloff
    width = int(ceil(lowpass_filter_width * o / base))
    idx = torch.arange(-width, width + o, dtype=torch.float64)[None, :] / o
    # torchaudio divides an int64 arange by new_freq: a float32 quotient, promoted to float64 when idx is added
    t = (torch.arange(0, -n, -1, dtype=torch.int64)[:, None] / n).to(torch.float64) + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * pi / lowpass_filter_width / 2) ** 2
    t = t * pi
    k = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / o)
    return k.to(torch.float32), width, o, n


def resample(x: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """torchaudio.transforms.Resample(orig_freq, new_freq)(x) for x (channels, S) (functional.py:_apply_sinc_resample_kernel):
    pad (width, width + o), correlate with the n phase filters at stride o, interleave the phases, keep ceil(n S / o) samples."""
    if int(orig_freq) == int(new_freq):
        return x
    k, width, o, n = sinc_resample_kernel(orig_freq, new_freq)
    S = x.shape[-1]
    xp = torch.nn.functional.pad(x.to(torch.float32).reshape(-1, S), (width, width + o))
    y = torch.nn.functional.conv1d(xp[:, None], k[:, None], stride=o)            # (channels, n, frames)
    y = y.transpose(1, 2).reshape(xp.shape[0], -1)
    target = -(-n * S // o)                                                         # ceil(n S / o)
    return y[:, :target].reshape(x.shape[:-1] + (target,))


def load_wav(wav, target_sr: int, min_sr: int = 16000) -> torch.Tensor:
    """utils/file_utils.py:44-50: mono (channel mean), resampled to target_sr -> float32 (1, S) in [-1, 1].
    `wav`: a path to a PCM / float .wav, or (samples, sample_rate) with samples (S,) or (channels, S).  Resampling restates
    torchaudio.transforms.Resample's default windowed-sinc filter bank (`resample` above)."""
    if isinstance(wav, (tuple, list)):
        data, sr = wav
        x = torch.as_tensor(np.asarray(data), dtype=torch.float32)
        x = x.reshape(1, -1) if x.dim() == 1 else x
    else:
        from scipy.io import wavfile
        sr, data = wavfile.read(wav)
        a = np.asarray(data)
        if a.dtype.kind == "i":
            a = a.astype(np.float32) / float(2 ** (8 * a.dtype.itemsize - 1))
        elif a.dtype.kind == "u":
            a = (a.astype(np.float32) - 128.0) / 128.0
        a = a.astype(np.float32)
        x = torch.from_numpy(a.reshape(-1, 1).T.copy() if a.ndim == 1 else a.T.copy())
    x = x.mean(dim=0, keepdim=True)
    if int(sr) != int(target_sr):
        assert sr >= min_sr, "wav sample rate {} must be greater than {}".format(sr, target_sr)
        x = resample(x, int(sr), int(target_sr))
    return x


class PromptMel:
    """matcha mel_spectrogram(n_fft 1920, hop 480, 80 mels, center=False) on the GPU (csrc/frontend.hip):
    speech (1, S) float32 at 24 kHz -> (1, 80, F) like the reference's feat_extractor (the frontend transposes it to (1, F, 80))."""

    def __init__(self, sample_rate: int = 24000, device: Optional[torch.device] = None):
        from .. import _lib
        self._lib = _lib
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().fy_prompt_mel_create(C.byref(self._h), int(sample_rate), self._stream()))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def __call__(self, speech: torch.Tensor) -> torch.Tensor:
        x = speech.reshape(-1).to(self.device, torch.float32).contiguous()
        L = self._lib.lib()
        frames = L.fy_prompt_mel_frames(int(x.numel()))
        out = torch.empty(frames, 80, device=self.device)
        self._lib.check(L.fy_prompt_mel_run(self._h, x.data_ptr(), int(x.numel()), out.data_ptr(), frames, self._stream()))
        return out.t().unsqueeze(0)

    def __del__(self):
        try:
            if self._h:
                self._lib.lib().fy_prompt_mel_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


class AudioFeat:
    """The 16 kHz features of the reference's two ONNX models on the GPU (csrc/frontend_feats.hip):
    kind "whisper": whisper.log_mel_spectrogram(speech, n_mels=128) -> (1, 128, frames)   (cli/frontend.py:97)
    kind "fbank":   kaldi.fbank(speech, num_mel_bins=80, dither=0, sample_frequency=16000), minus its mean over the frames when
                    subtract_mean -> (frames, 80)                                           (cli/frontend.py:111-115)"""

    def __init__(self, kind: str, device: Optional[torch.device] = None):
        from .. import _lib
        assert kind in ("whisper", "fbank")
        self._lib, self.kind = _lib, kind
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().fy_audio_feat_create(C.byref(self._h), 0 if kind == "whisper" else 1, self._stream()))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def __call__(self, speech: torch.Tensor, subtract_mean: bool = False) -> torch.Tensor:
        x = speech.reshape(-1).to(self.device, torch.float32).contiguous()
        L = self._lib.lib()
        frames, mels = L.fy_audio_feat_frames(self._h, int(x.numel())), L.fy_audio_feat_mels(self._h)
        if frames < 1:
            raise ValueError(f"{x.numel()} samples are too few for one {self.kind} frame")
        out = torch.empty((mels, frames) if self.kind == "whisper" else (frames, mels), device=self.device)
        self._lib.check(L.fy_audio_feat_run(self._h, x.data_ptr(), int(x.numel()), out.data_ptr(), frames, 1 if subtract_mean else 0, self._stream()))
        return out.unsqueeze(0) if self.kind == "whisper" else out

    def __del__(self):
        try:
            if self._h:
                self._lib.lib().fy_audio_feat_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


def onnx_prompt_models(model_dir: str, device: Optional[torch.device] = None):
    """The reference's two ONNX sessions (cli/frontend.py:41-46: campplus.onnx on the CPU provider, speech_tokenizer_v3.onnx on the
    GPU provider when there is one) over this package's feature kernels, as the callables CosyVoiceFrontEnd takes:
    (speech_tokenizer(speech16k) -> list[int], spk_embedder(speech16k) -> (1, 192)).  (None, None) when onnxruntime does not
    import or the files are missing - the caller then injects its own callables."""
    try:
        import onnxruntime
    except Exception:
        return None, None
    camp, tok = os.path.join(model_dir, "campplus.onnx"), os.path.join(model_dir, "speech_tokenizer_v3.onnx")
    if not (os.path.exists(camp) and os.path.exists(tok)):
        return None, None
    option = onnxruntime.SessionOptions()
    option.graph_optimization_level = onnxruntime.GraphOptimizationLevel.ORT_ENABLE_ALL
    option.intra_op_num_threads = 1
    have = set(onnxruntime.get_available_providers())
    gpu = [p for p in ("ROCMExecutionProvider", "MIGraphXExecutionProvider", "CUDAExecutionProvider") if p in have]
    camp_s = onnxruntime.InferenceSession(camp, sess_options=option, providers=["CPUExecutionProvider"])
    tok_s = onnxruntime.InferenceSession(tok, sess_options=option, providers=(gpu[:1] or []) + ["CPUExecutionProvider"])
    whisper_feat, fbank = AudioFeat("whisper", device), AudioFeat("fbank", device)

    def speech_tokenizer(speech16k):
        assert speech16k.shape[1] / 16000 <= 30, "do not support extract speech token for audio longer than 30s"      # frontend.py:96
        feat = whisper_feat(speech16k)
        ins = tok_s.get_inputs()
        return tok_s.run(None, {ins[0].name: feat.detach().cpu().numpy(),
                                ins[1].name: np.array([feat.shape[2]], dtype=np.int32)})[0].flatten().tolist()

    def spk_embedder(speech16k):
        feat = fbank(speech16k, subtract_mean=True)
        emb = camp_s.run(None, {camp_s.get_inputs()[0].name: feat.unsqueeze(dim=0).cpu().numpy()})[0].flatten().tolist()
        return torch.tensor([emb])
    return speech_tokenizer, spk_embedder


def _missing(what: str, ref: str):
    def f(*a, **kw):
        raise NotImplementedError(f"{what} is not available here ({ref}); pass it to CosyVoiceFrontEnd(...) as a callable")
    return f


class CosyVoiceFrontEnd:
    """tokenizer: object with encode(text, allowed_special=...) -> list[int].
    feat_extractor(speech24k (1, S)) -> (1, 80, F); speech_tokenizer(speech16k (1, S)) -> list[int];
    spk_embedder(speech16k (1, S)) -> (1, 192) tensor.  Defaults: PromptMel on the GPU; the two ONNX models must be injected."""

    def __init__(self, tokenizer, feat_extractor: Optional[Callable] = None, speech_tokenizer: Optional[Callable] = None,
                 spk_embedder: Optional[Callable] = None, spk2info: str = "", allowed_special: str = "all",
                 number_speller: Optional[Callable[[str], str]] = None, text_normalizer: Optional[Dict[str, Callable[[str], str]]] = None,
                 device: Optional[torch.device] = None, wav_loader: Callable = load_wav):
        self.tokenizer = tokenizer
        self.device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.feat_extractor = feat_extractor if feat_extractor is not None else PromptMel(24000, self.device if self.device.type == "cuda" else None)
        self.speech_tokenizer = speech_tokenizer or _missing("the speech tokenizer (speech_tokenizer_v3.onnx over whisper's 128-bin log-mel)",
                                                             "cli/frontend.py:94-108; onnxruntime is not installed or the .onnx file is missing: "
                                                             "AudioFeat('whisper') computes the model's input")
        self.spk_embedder = spk_embedder or _missing("the speaker embedder (campplus.onnx over an 80-bin kaldi fbank)",
                                                     "cli/frontend.py:110-117; onnxruntime is not installed or the .onnx file is missing: "
                                                     "AudioFeat('fbank') computes the model's input")
        self.spk2info = torch.load(spk2info, map_location=self.device, weights_only=True) if spk2info and os.path.exists(spk2info) else {}
        self.allowed_special = allowed_special
        self.number_speller = number_speller                  # inflect.engine().number_to_words in the reference
        self.text_normalizer = text_normalizer or {}          # {"zh": fn, "en": fn}: wetext's normalizers in the reference
        self.text_frontend = "injected" if self.text_normalizer else ""
        self.load_wav = wav_loader

    # ---- extraction (cli/frontend.py:78-125) ----
    def _extract_text_token(self, text):
        if isinstance(text, Generator):
            return self._extract_text_token_generator(text), torch.tensor([0], dtype=torch.int32).to(self.device)
        ids = self.tokenizer.encode(text, allowed_special=self.allowed_special)
        tok = torch.tensor([ids], dtype=torch.int32).to(self.device)
        return tok, torch.tensor([tok.shape[1]], dtype=torch.int32).to(self.device)

    def _extract_text_token_generator(self, text_generator):
        for text in text_generator:
            tok, _ = self._extract_text_token(text)
            for i in range(tok.shape[1]):
                yield tok[:, i: i + 1]

    def _extract_speech_token(self, prompt_wav):
        speech = self.load_wav(prompt_wav, 16000)
        assert speech.shape[1] / 16000 <= 30, "do not support extract speech token for audio longer than 30s"
        ids = list(self.speech_tokenizer(speech))
        tok = torch.tensor([ids], dtype=torch.int32).to(self.device)
        return tok, torch.tensor([tok.shape[1]], dtype=torch.int32).to(self.device)

    def _extract_spk_embedding(self, prompt_wav):
        emb = self.spk_embedder(self.load_wav(prompt_wav, 16000))
        return torch.as_tensor(emb, dtype=torch.float32).reshape(1, -1).to(self.device)

    def _extract_speech_feat(self, prompt_wav):
        speech = self.load_wav(prompt_wav, 24000)
        feat = self.feat_extractor(speech).squeeze(dim=0).transpose(0, 1).to(self.device).unsqueeze(dim=0)
        return feat, torch.tensor([feat.shape[1]], dtype=torch.int32).to(self.device)

    # ---- text (cli/frontend.py:127-160) ----
    def text_normalize(self, text, split=True, text_frontend=True):
        if isinstance(text, Generator):
            return [text]
        if "<|" in text and "|>" in text:                      # ssml-like control symbols: leave the text alone
            text_frontend = False
        if text_frontend is False or text == "":
            return [text] if split is True else text
        text = text.strip()
        enc = lambda t: self.tokenizer.encode(t, allowed_special=self.allowed_special)
        if contains_chinese(text):
            if "zh" in self.text_normalizer:
                text = self.text_normalizer["zh"](text)
            text = text.replace("\n", "")
            text = replace_corner_mark(replace_blank(text))
            text = text.replace(".", "。").replace(" - ", "，")
            text = remove_bracket(text)
            text = re.sub(r"[，,、]+$", "。", text)
            texts = split_paragraph(text, enc, "zh", token_max_n=80, token_min_n=60, merge_len=20, comma_split=False)
        else:
            if "en" in self.text_normalizer:
                text = self.text_normalizer["en"](text)
            if self.number_speller is not None:
                text = spell_out_number(text, self.number_speller)
            texts = split_paragraph(text, enc, "en", token_max_n=80, token_min_n=60, merge_len=20, comma_split=False)
        texts = [t for t in texts if not is_only_punctuation(t)]
        return texts if split is True else text

    # ---- model_input dicts (cli/frontend.py:162-224) ----
    def frontend_sft(self, tts_text, spk_id):
        tok, n = self._extract_text_token(tts_text)
        emb = self.spk2info[spk_id]["embedding"]
        return {"text": tok, "text_len": n, "llm_embedding": emb, "flow_embedding": emb}

    def frontend_zero_shot(self, tts_text, prompt_text, prompt_wav, resample_rate, zero_shot_spk_id):
        tok, n = self._extract_text_token(tts_text)
        if zero_shot_spk_id == "":
            ptok, pn = self._extract_text_token(prompt_text)
            feat, feat_len = self._extract_speech_feat(prompt_wav)
            stok, stok_len = self._extract_speech_token(prompt_wav)
            if resample_rate == 24000:
                # two mel frames per speech token, exactly (frontend.py:174-178): both are cut to the shorter
                token_len = min(int(feat.shape[1] / 2), stok.shape[1])
                feat, feat_len[:] = feat[:, : 2 * token_len], 2 * token_len
                stok, stok_len[:] = stok[:, :token_len], token_len
            emb = self._extract_spk_embedding(prompt_wav)
            model_input = {"prompt_text": ptok, "prompt_text_len": pn,
                           "llm_prompt_speech_token": stok, "llm_prompt_speech_token_len": stok_len,
                           "flow_prompt_speech_token": stok, "flow_prompt_speech_token_len": stok_len,
                           "prompt_speech_feat": feat, "prompt_speech_feat_len": feat_len,
                           "llm_embedding": emb, "flow_embedding": emb}
        else:
            model_input = {**self.spk2info[zero_shot_spk_id]}
        model_input["text"], model_input["text_len"] = tok, n
        return model_input

    def frontend_cross_lingual(self, tts_text, prompt_wav, resample_rate, zero_shot_spk_id):
        d = self.frontend_zero_shot(tts_text, "", prompt_wav, resample_rate, zero_shot_spk_id)
        for k in ("prompt_text", "prompt_text_len", "llm_prompt_speech_token", "llm_prompt_speech_token_len"):
            del d[k]                                           # no prompt in the LM
        return d

    def frontend_instruct(self, tts_text, spk_id, instruct_text):
        d = self.frontend_sft(tts_text, spk_id)
        del d["llm_embedding"]
        d["prompt_text"], d["prompt_text_len"] = self._extract_text_token(instruct_text)
        return d

    def frontend_instruct2(self, tts_text, instruct_text, prompt_wav, resample_rate, zero_shot_spk_id):
        d = self.frontend_zero_shot(tts_text, instruct_text, prompt_wav, resample_rate, zero_shot_spk_id)
        del d["llm_prompt_speech_token"], d["llm_prompt_speech_token_len"]
        return d

    def frontend_vc(self, source_speech_16k, prompt_wav, resample_rate):
        ptok, pn = self._extract_speech_token(prompt_wav)
        feat, feat_len = self._extract_speech_feat(prompt_wav)
        emb = self._extract_spk_embedding(prompt_wav)
        stok, sn = self._extract_speech_token(source_speech_16k)
        return {"source_speech_token": stok, "source_speech_token_len": sn,
                "flow_prompt_speech_token": ptok, "flow_prompt_speech_token_len": pn,
                "prompt_speech_feat": feat, "prompt_speech_feat_len": feat_len, "flow_embedding": emb}
