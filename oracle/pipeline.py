"""ORACLE (test infrastructure, not product code) - the whole per-utterance path
CosyVoice3Model.tts(stream=False): llm_job -> token2wav (flow -> hift),
CosyVoice/cosyvoice/cli/model.py:101-129, 324-389, 416-441.

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
