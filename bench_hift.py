#!/usr/bin/env python3
"""BASELINE.json configs[4]: HiFT-only vocoder path, random mel (B, 80, F) -> 24 kHz wav on one MI355X.

    python bench_hift.py [--batch 32] [--frames 10000] [--steps 3] [--warmup 1] [--flags 0]

Reports audio seconds per second and the fraction of the HBM roofline using the algorithmic figures of
SURVEY 8(d): 4.156 MB (fp32 activation I/O, every conv reads its input once and writes its output once)
and 674.0 MFLOP per mel frame.  One JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
BYTES_PER_FRAME = 4.156e6
FLOP_PER_FRAME = 674.0e6
HBM_PEAK = 8.0e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=10000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--flags", type=int, default=0)
    a = ap.parse_args()
    print(json.dumps(run(a.batch, a.frames, a.steps, a.warmup, a.flags)))


def run(batch=32, frames=10000, steps=3, warmup=1, flags=0):
    """BASELINE.json configs[4] as a function (bench.py reports it as "hift_cfg5")."""
    import types
    a = types.SimpleNamespace(batch=batch, frames=frames, steps=steps, warmup=warmup, flags=flags)
    from fangyan_tts_amd import _lib, synth
    from fangyan_tts_amd.hift import HiftEngine
    from fangyan_tts_amd.spec import HiftCfg
    dev = torch.device("cuda:0")
    cfg = HiftCfg()
    sd = synth.state_dict_torch(cfg.manifest(), dev)
    eng = HiftEngine(sd, cfg, max_batch=a.batch, max_frames=a.frames, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    mel = torch.rand(a.batch, 80, a.frames, device=dev, generator=g)               # generator.py:740
    ri = torch.from_numpy(synth.hift_rand_ini()).to(dev)
    sn = torch.rand(1, a.frames * 480, 9, device=dev, generator=g)
    for _ in range(a.warmup):
        eng.inference(mel, ri, sn, flags=a.flags)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        wav, _ = eng.inference(mel, ri, sn, flags=a.flags)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    L = _lib.lib()
    L.fy_prof_reset(); L.fy_prof_enable(1)
    eng.inference(mel, ri, sn, flags=a.flags)
    torch.cuda.synchronize()
    L.fy_prof_enable(0)
    ms, flops, n = _lib.prof_get("conv_mfma")
    L.fy_prof_reset()
    frames = a.batch * a.frames
    audio = frames * 480 / 24000.0
    # HBM-side traffic: rocprofv3 --pmc cannot sit under a torch process on this pool, so the per-frame figure is the one
    # measured on the stand-alone driver of the same engine (profiles/r02_hift_pmc.json says how), scaled to this run
    traffic, pmc = None, os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r02_hift_pmc.json")
    if os.path.exists(pmc) and a.flags == 0:
        traffic = int(json.load(open(pmc))["traffic_bytes_per_frame"] * frames)
    bw = frames * BYTES_PER_FRAME / dt
    del eng, mel, sn, wav
    torch.cuda.empty_cache()
    return ({
        "metric": "HiFT vocoder audio sec/sec", "value": round(audio / dt, 1), "unit": "audio_s/s", "ms_per_step": round(1e3 * dt, 2),
        "config": {"workload": f"HiFT-only, random mel, batch {a.batch} x {a.frames} frames", "flags": a.flags},
        "roofline": {"bound": "hbm", "achieved": round(bw / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(bw / HBM_PEAK, 4),
                     "traffic": traffic, "traffic_unit": "bytes per step (L2-miss bytes per frame from profiles/r02_hift_pmc.json x frames)",
                     "frac_on_counter_bytes": round(traffic / dt / HBM_PEAK, 4) if traffic else None,
                     "algorithmic_bytes": int(frames * BYTES_PER_FRAME), "algorithmic_tflops": round(frames * FLOP_PER_FRAME / dt / 1e12, 1),
                     "conv_mfma_ms": round(ms, 2), "conv_mfma_tflops": round(flops / (ms * 1e-3) / 1e12, 1) if ms else None,
                     "conv_mfma_launches": n}})


if __name__ == "__main__":
    main()
