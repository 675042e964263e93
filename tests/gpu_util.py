"""Shared helpers of the -m gpu tests."""
import json
import os

import numpy as np
import torch

from fangyan_tts_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
OUT = os.path.join(ROOT, "gpurun_out")


def note(fname, key, value):
    os.makedirs(OUT, exist_ok=True)
    p = os.path.join(OUT, fname)
    d = json.load(open(p)) if os.path.exists(p) else {}
    d[key] = value
    json.dump(d, open(p, "w"), indent=1, sort_keys=True)


def maxerr(a, b):
    return float((a.detach().cpu().float() - b.detach().cpu().float()).abs().max())


def to_dev(sd, dev):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in sd.items()}


def golden(name):
    p = os.path.join(G, name)
    return np.load(p) if os.path.exists(p) else None


def synth_mel(name, frames):
    return np.clip(synth.normal(name, (1, frames, 80), -5.0, 2.0), -11.5, 2.0)


def llm_case(cfg, n_text, n_ptext, p_tok, tag):
    hi = min(cfg.vocab, 151643)
    return (synth.randint(f"in.llm.text.{tag}", (1, n_text), 0, hi)[0].tolist(),
            synth.randint(f"in.llm.ptext.{tag}", (1, n_ptext), 0, hi)[0].tolist(),
            synth.randint(f"in.llm.ptok.{tag}", (1, p_tok), 0, cfg.speech_tokens)[0].tolist())


def dit_inputs(T):
    return [torch.from_numpy(a) for a in (
        synth.normal(f"in.dit.x.{T}", (2, 80, T)), synth.normal(f"in.dit.mu.{T}", (2, 80, T)),
        synth.normal(f"in.dit.cond.{T}", (2, 80, T)), synth.normal(f"in.dit.spks.{T}", (2, 80)),
        np.array([0.3, 0.3], dtype=np.float32))]
