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
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    print(json.dumps(run(a.batch, a.frames, a.steps, a.warmup, a.flags, check=not a.no_check)))


def check_output(eng, cfg, sd, mel, ri, sn, wav, flags, prefix_frames=1000):
    """What the timed call produced, checked without a reference of that size (size-independent properties) plus the
    CPU oracle where it is affordable:
      * the LAST utterance of the batch (its stage tensors sit beyond 2^31 elements at 32 x 10 000 frames) equals a solo run
        of its own mel (batch = solo, <= 1e-5: the kernels are batch-invariant up to the last bit or two);
      * utterance 0's first `prefix_frames` frames against the oracle vocoder (oracle/hift.py, CPU fp32) on the mel prefix plus
        the 4 frames conv_pre looks ahead (the net is causal: CosyVoice/cosyvoice/hifigan/generator.py:739-746), tolerance 2.5e-3
        as in tests/test_e2e_gpu.py (bf16 MFMA operands; measured 5-9e-4);
      * |wav| <= 0.99 (the clamp of generator.py:746) and no NaN."""
    from oracle import hift as ohift
    B, _, F = mel.shape
    up = cfg.upsample_total
    solo, _ = eng.inference(mel[B - 1: B].contiguous(), ri, sn, flags=flags)
    e_solo = float((wav[B - 1: B] - solo).abs().max())
    del solo
    n = min(prefix_frames, F - 4)
    P = ohift.prepare({k: v.cpu().numpy() for k, v in sd.items()})
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    ref, _ = ohift.inference(mel[0:1, :, : n + 4].cpu(), P, cfg, ri.cpu(), sn[:, : (n + 4) * up].cpu())
    e_ref = float((wav[0:1, : n * up].cpu() - ref[:, : n * up]).abs().max())
    peak, finite = float(wav.abs().max()), bool(torch.isfinite(wav).all())
    return {"what": f"last utterance of the batch vs a solo run of its mel; utterance 0's first {n} frames vs the CPU oracle vocoder; clamp; finite",
            "last_utterance_vs_solo_max_abs": round(e_solo, 8), "solo_tol": 1e-5, "utt0_prefix_vs_oracle_max_abs": round(e_ref, 6), "oracle_tol": 2.5e-3,
            "peak_abs": round(peak, 4), "finite": finite, "ok": bool(e_solo <= 1e-5 and e_ref <= 2.5e-3 and peak <= 0.99 + 1e-6 and finite)}


def run(batch=32, frames=10000, steps=3, warmup=1, flags=0, check=True):
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
    checked = check_output(eng, cfg, sd, mel, ri, sn, wav, a.flags) if check else None
    frames = a.batch * a.frames
    audio = frames * 480 / 24000.0
    # HBM-side traffic: rocprofv3 --pmc cannot sit under a torch process on this pool, so the per-frame figure is the one
    # measured on the stand-alone driver of the same engine (profiles/r05_hift_pmc.json says how; tests/micro/hift_pmc.py collects it), scaled to this run
    traffic, pmc_name = None, None
    for pmc_name in ("r05_hift_pmc.json", "r04_hift_pmc.json", "r03_hift_pmc.json", "r02_hift_pmc.json"):      # the newest collected
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", pmc_name)
        if os.path.exists(pmc) and a.flags == 0:
            traffic = int(json.load(open(pmc))["traffic_bytes_per_frame"] * frames)
            break
    bw = frames * BYTES_PER_FRAME / dt
    del eng, mel, sn, wav
    torch.cuda.empty_cache()
    return ({
        "metric": "HiFT vocoder audio sec/sec", "value": round(audio / dt, 1), "unit": "audio_s/s", "ms_per_step": round(1e3 * dt, 2),
        "config": {"workload": f"HiFT-only, random mel, batch {a.batch} x {a.frames} frames", "flags": a.flags},
        "roofline": {"bound": "hbm", "achieved": round(bw / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(bw / HBM_PEAK, 4),
                     "traffic": traffic, "traffic_unit": f"bytes per step (L2-miss bytes per frame from profiles/{pmc_name} x frames)",
                     "frac_on_counter_bytes": round(traffic / dt / HBM_PEAK, 4) if traffic else None,
                     "algorithmic_bytes": int(frames * BYTES_PER_FRAME), "algorithmic_tflops": round(frames * FLOP_PER_FRAME / dt / 1e12, 1),
                     "conv_mfma_ms": round(ms, 2), "conv_mfma_tflops": round(flops / (ms * 1e-3) / 1e12, 1) if ms else None,
                     "conv_mfma_launches": n},
        "checked": checked})


if __name__ == "__main__":
    main()
