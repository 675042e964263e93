"""ORACLE (test infrastructure, not product code) - the whole per-utterance path
CosyVoice3Model.tts: llm_job -> token2wav (flow -> hift), stream=False and
stream=True, CosyVoice/cosyvoice/cli/model.py:101-129, 324-389, 416-441.

`model_input` uses the reference's own keys (cli/frontend.py:168-213):
text, prompt_text, llm_prompt_speech_token, flow_prompt_speech_token,
prompt_speech_feat, llm_embedding, flow_embedding.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from fangyan_tts_amd.spec import ModelCfg
from . import flow as oflow
from . import hift as ohift
from . import llm as ollm


def tts(model_input: Dict[str, torch.Tensor], PL, PF, PH, cfg: ModelCfg,
        rand_noise, rand_ini, sine_noise, speed: float = 1.0,
        min_len: Optional[int] = None, max_len: Optional[int] = None) -> Dict[str, torch.Tensor]:
    z = torch.zeros(1, 0, dtype=torch.int32)
    toks = list(ollm.inference(model_input["text"], model_input.get("prompt_text", z),
                               model_input.get("llm_prompt_speech_token", z), PL, cfg.llm,
                               min_len=min_len, max_len=max_len))
    toks = ollm.silent_filter(toks)
    token = torch.tensor(toks, dtype=torch.int32).unsqueeze(0)
    mel = oflow.inference(token, model_input["flow_prompt_speech_token"], model_input["prompt_speech_feat"],
                          model_input["flow_embedding"], PF, cfg.flow, rand_noise)
    if speed != 1.0:                                       # cli/model.py:435-437
        mel = F.interpolate(mel, size=int(mel.shape[2] / speed), mode="linear")
    wav, _ = ohift.inference(mel, PH, cfg.hift, rand_ini, sine_noise)
    return {"tokens": token, "mel": mel, "tts_speech": wav}


def tts_stream(model_input: Dict[str, torch.Tensor], PL, PF, PH, cfg: ModelCfg,
               rand_noise, rand_ini, sine_noise, token_hop_len: int = 25,
               min_len: Optional[int] = None, max_len: Optional[int] = None) -> Dict[str, object]:
    """CosyVoice3Model.tts(stream=True), cli/model.py:339-369 with token2wav :416-441.
    The reference polls a token list that the LM thread fills; its chunk boundaries depend only on the
    token count (a chunk is cut as soon as hop + look-ahead tokens beyond the offset exist), so the
    sequential restatement below yields the same chunks: every chunk re-runs the flow decoder over all
    tokens so far (streaming mask, finalize=False) and the vocoder over the whole mel so far, and emits
    the samples beyond those already emitted."""
    z = torch.zeros(1, 0, dtype=torch.int32)
    toks = list(ollm.inference(model_input["text"], model_input.get("prompt_text", z),
                               model_input.get("llm_prompt_speech_token", z), PL, cfg.llm,
                               min_len=min_len, max_len=max_len))
    toks = ollm.silent_filter(toks)
    ptok, pfeat, emb = model_input["flow_prompt_speech_token"], model_input["prompt_speech_feat"], model_input["flow_embedding"]
    look = cfg.flow.pre_lookahead
    pad = int(-(-ptok.shape[1] // token_hop_len) * token_hop_len - ptok.shape[1])        # :341
    offset, mel_all, speech_offset, chunks = 0, None, 0, []

    def token2wav(token, offset, mel_all, speech_offset, stream, finalize):
        mel = oflow.inference(token, ptok, pfeat, emb, PF, cfg.flow, rand_noise, streaming=stream, finalize=finalize)
        mel = mel[:, :, offset * 2:]
        mel_all = mel if mel_all is None else torch.cat([mel_all, mel], dim=2)
        wav, _ = ohift.inference(mel_all, PH, cfg.hift, rand_ini, sine_noise, finalize=finalize)
        wav = wav[:, speech_offset:]
        return wav, mel_all, speech_offset + wav.shape[1]

    while True:
        hop = token_hop_len + pad if offset == 0 else token_hop_len
        if len(toks) - offset >= hop + look:
            token = torch.tensor(toks[: offset + hop + look], dtype=torch.int32).unsqueeze(0)
            wav, mel_all, speech_offset = token2wav(token, offset, mel_all, speech_offset, True, False)
            offset += hop
            chunks.append(wav)
        else:
            break
    token = torch.tensor(toks, dtype=torch.int32).unsqueeze(0)
    wav, mel_all, speech_offset = token2wav(token, offset, mel_all, speech_offset, False, True)
    chunks.append(wav)
    return {"tokens": token, "mel": mel_all, "chunks": chunks, "tts_speech": torch.cat(chunks, dim=1)}
