#!/usr/bin/env python3
"""Benchmark of the north-star path: CosyVoice3-0.5B instruct inference, batch 8 per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path (speech-token LM greedy decode -> 10-step CFG flow matching over
the DiT -> HiFT vocoder) over one batch of 8 synthetic instruct utterances per GPU, inputs resident
in HBM, outputs (wavs) copied to host memory inside the timed region; with N > 1 every
rank runs its own 8 utterances (data-parallel, no data-path collective), one RCCL all-gather collects
the finished audio and every rank copies it to the host.  `--gpus N` without an outer launcher starts
its own N ranks (one child `torch.distributed.run`).  After the timed region utterance 0 of the last
timed step is checked against the CPU oracle ("checked" in the JSON line).

Workload (cfg 2 = BASELINE.json configs[1]): instruct text 8 ids + tts text U{10..20} ids,
5 s prompt (125 speech tokens / 250 mel frames, x-vector 192), forced length n = 75 tokens per
utterance (3 s of audio each), random-init weights of the CosyVoice3-0.5B architecture
(fangyan_tts_amd.synth), bf16 MFMA arithmetic for DiT / HiFT, fp32 activations in the LM.

Consecutive steps are software-pipelined (defaults: 40 timed steps, 4 warm-up): ONE speech-token LM call decodes the batches of
4 consecutive steps together (32 sequences per weight pass, one persistent launch per token step: csrc/llm_decode32.hip) beside
the flow decoders + vocoders of the steps before, two of which run side by side (--flow-workers 2); every step still runs the
whole path on its own batch of 8 inside the timed region, and the record says what is in flight (`config`) and gives the strict
one-batch-at-a-time figure beside the headline (`value_batch8_unpipelined`) and the figure with fresh input tensors every step
(`value_fresh_inputs`).  `roofline` is the kernel family with the largest GPU-time share of the timed region (the DiT linears),
measured in the timed configuration with HIP events on the launch stream - per launch as the contract defines it, plus
`chip_level` (two launches share the chip) and `one_step_alone`; `roofline_lm` the LM's decode; every `traffic` comes from `rocprofv3 --pmc` over this very
program (profiles/r05_bench_pmc.json; tests/micro/prof_r05.sh).  Secondary objects: `precise_mode`, `latency_b1`, `first_chunk`,
`zero_shot_b4` (configs[2]), `hift_cfg5` (configs[4], output checked), `cpu_baseline`, `checked`.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_TOK = 75            # forced speech tokens per utterance (3 s)
P_TOK = 125           # prompt speech tokens (5 s)
BATCH = 8
PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBPS = 8000.0        # HBM3E, same guide (measured float4 copy: 6290 GB/s)


def make_inputs(cfg, rank):
    import numpy as np
    from fangyan_tts_amd import synth
    inputs = []
    for b in range(BATCH):
        tag = f"bench.r{rank}.u{b}"
        n_text = 10 + int(synth.randint(tag + ".len", (1,), 0, 11)[0])
        hi = min(cfg.llm.vocab, 151643)
        mel = np.clip(synth.normal(tag + ".pfeat", (1, 2 * P_TOK, 80), -5.0, 2.0), -11.5, 2.0)
        inputs.append({
            "text": torch.from_numpy(synth.randint(tag + ".text", (1, n_text), 0, hi)),
            "prompt_text": torch.from_numpy(synth.randint(tag + ".instruct", (1, 8), 0, hi)),
            "llm_prompt_speech_token": torch.zeros(1, 0, dtype=torch.int32),        # instruct2 drops it, frontend.py:209-213
            "flow_prompt_speech_token": torch.from_numpy(synth.randint(tag + ".ptok", (1, P_TOK), 0, cfg.flow.vocab)),
            "prompt_speech_feat": torch.from_numpy(mel),
            "flow_embedding": torch.from_numpy(synth.normal(tag + ".spk", (1, 192))),
        })
    return inputs


def cpu_baseline(cfg, sd_llm, sd_flow, sd_hift, inp, noise, ri, sn, timed=None):
    """The oracle (CPU restatement, kind 'port') on ONE utterance of the same workload; with `timed` (the outputs of the
    last timed step) utterance 0 of the timed run is checked against it."""
    from oracle import flow as oflow, hift as ohift, llm as ollm, pipeline as opipe
    # the GPU box gives one job a 16-core share whatever os.cpu_count() says
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    torch.set_num_threads(cores)
    PL = {k: v.cpu() for k, v in sd_llm.items()}
    PF = {k: v.cpu() for k, v in sd_flow.items()}
    PH = ohift.prepare({k: v.cpu().numpy() for k, v in sd_hift.items()})
    t0 = time.time()
    out = opipe.tts(inp, PL, PF, PH, cfg, noise.cpu(), ri.cpu(), sn.cpu(), min_len=N_TOK, max_len=N_TOK)
    dt = time.time() - t0
    audio = out["tts_speech"].shape[1] / 24000.0
    base = {"value": round(audio / dt, 4), "unit": "audio_s/s", "cores": cores, "kind": "port",
            "sample": f"1 utterance of the same workload (5 s prompt, {N_TOK} forced tokens -> {audio:.1f} s audio) in {dt:.1f} s, "
                      f"torch fp32 on {cores} threads"}
    checked = None
    if timed is not None:
        # utterance 0 of the LAST TIMED step against the oracle: ids exact; mel within 3x the error measured for the bf16
        # flow decoder at this size (1.3e-2, gpurun_out/parity_flow.json); wav against the oracle vocoder run on the engine's
        # own mel (sample-wise comparison with an fp32-mel waveform is only meaningful for the first frames: the harmonic
        # source integrates f0, tests/test_e2e_gpu.py) within 3x the measured 8e-4, and the log-mel distance between the two
        # complete waveforms (phase-insensitive) for the whole utterance: 3 dB = 3x the measured 1.0 (a random-weight
        # vocoder turns uniform +-1e-2 noise on the mel into 2.7 dB; the fp32-class mode matches the reference's waveform
        # sample by sample over its whole length, tests/test_e2e_gpu.py)
        ids = timed["toks"][0].reshape(-1).tolist()
        ref_ids = out["tokens"].reshape(-1).tolist()
        n = len(ref_ids)
        mel = timed["mel"][0:1, :, : 2 * n].float().cpu()
        e_mel = float((mel - out["mel"]).abs().max()) if mel.shape == out["mel"].shape else float("inf")
        S = out["tts_speech"].shape[1]
        wav = timed["wav"][0:1, :S]
        ref_wav, _ = ohift.inference(mel, PH, cfg.hift, ri.cpu(), sn.cpu()[:, :S])
        e_wav = float((wav - ref_wav).abs().max())
        e_lm = logmel_distance(wav, out["tts_speech"])
        checked = {"what": "utterance 0 of the last timed step vs the CPU oracle", "ids_equal": ids == ref_ids, "n_ids": n,
                   "mel_max_abs_err": round(e_mel, 5), "mel_tol": 4e-2,
                   "wav_vs_oracle_vocoder_on_engine_mel": round(e_wav, 6), "wav_tol": 2.5e-3,
                   "logmel_db_vs_oracle_wav": round(e_lm, 3), "logmel_db_tol": 3.0,
                   "ok": bool(ids == ref_ids and e_mel <= 4e-2 and e_wav <= 2.5e-3 and e_lm <= 3.0)}
    return base, checked


def logmel_distance(a, b, n_fft=1024, hop=256, n_mels=80, sr=24000):
    """Mean absolute difference, in dB, of the log-mel spectra of two waveforms: a phase-insensitive comparison (the
    harmonic source's phase is an integral of f0, so two correct waveforms from slightly different mels drift apart
    sample-wise while sounding the same)."""
    import math
    n = min(a.shape[-1], b.shape[-1])
    win = torch.hann_window(n_fft)
    fb = torch.zeros(n_mels, n_fft // 2 + 1)
    mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
    pts = torch.linspace(mel(0.0), mel(sr / 2), n_mels + 2)
    hz = 700.0 * (10.0 ** (pts / 2595.0) - 1.0)
    bins = torch.arange(n_fft // 2 + 1) * sr / n_fft
    for m in range(n_mels):
        lo, c, hi = hz[m], hz[m + 1], hz[m + 2]
        fb[m] = torch.clamp(torch.minimum((bins - lo) / (c - lo), (hi - bins) / (hi - c)), min=0.0)
    def lm(x):
        sp = torch.stft(x.reshape(-1)[:n].float(), n_fft, hop, window=win, return_complex=True).abs() ** 2
        return 10.0 * torch.log10(fb @ sp + 1e-7)
    return float((lm(a) - lm(b)).abs().mean())


def spawn_ranks(n):
    import socket
    import subprocess
    from fangyan_tts_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):                  # one builder, before the ranks exist (they require it prebuilt)
        build.build(verbose=False)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def _stage_profile(L, _lib, fn, names=("gemm_bf16", "conv_mfma", "gemv", "llm_decode", "gemm_exact")):
    """One call of fn() with HIP events around every profiled launch: {name: (ms, work, launches)}."""
    L.fy_prof_reset()
    L.fy_prof_enable(1)
    fn()
    torch.cuda.synchronize()
    L.fy_prof_enable(0)
    out = {k: _lib.prof_get(k) for k in names}
    L.fy_prof_reset()
    return out


def _pmc(name, *keys):
    """A figure from this round's PMC summary under profiles/ (rocprofv3 --pmc, collected as the guide's HBM section prescribes);
    None when the file or the key is absent."""
    p = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(p):
        return None
    d = json.load(open(p))
    for k in keys:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def dit_roofline(ms, flops, n, where, traffic):
    ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    return {"bound": "mfma", "kernel": "DiT linears: gemm64_k (qkv, ff1: 64-deep register-staged stages, 320x256 / 256x256 tiles, LayerNorm-modulate folded in) and gemm256_k (out-projection, ff2: LDS-DMA ring, 128x128 tiles, gated fp32 residual epilogue), 16x16x32 MFMAs",
            "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
            "traffic_unit": "HBM-side bytes per launch, mean over every gemm64_k / gemm256_k launch of the run (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE over bench.py itself, separate passes: profiles/r05_bench_pmc.json, families.dit_linears)",
            "launches": n, "avg_launch_us": round(1e3 * ms / max(n, 1), 2), "gflop_per_launch": round(flops / max(n, 1) / 1e9, 3), "measured_over": where}


def zero_shot_inputs(cfg, B=4, P=250):
    """configs[2]'s synthetic batch: 14 text ids behind 30 prompt-text ids, P prompt speech tokens (LM and flow), 2 P prompt mel frames."""
    import numpy as np
    from fangyan_tts_amd import synth
    hi = min(cfg.llm.vocab, 151643)
    inputs = []
    for b in range(B):
        tag = f"bench.zs.u{b}"
        ptok = torch.from_numpy(synth.randint(tag + ".ptok", (1, P), 0, cfg.flow.vocab))
        inputs.append({"text": torch.from_numpy(synth.randint(tag + ".text", (1, 14), 0, hi)),
                       "prompt_text": torch.from_numpy(synth.randint(tag + ".ptext", (1, 30), 0, hi)),
                       "llm_prompt_speech_token": ptok, "flow_prompt_speech_token": ptok,
                       "prompt_speech_feat": torch.from_numpy(np.clip(synth.normal(tag + ".pfeat", (1, 2 * P, 80), -5.0, 2.0), -11.5, 2.0)),
                       "flow_embedding": torch.from_numpy(synth.normal(tag + ".spk", (1, 192)))})
    return inputs


def bench_zero_shot(cfg, sd_llm, sd_flow, sd_hift, dev, L, _lib, steps=3, flow_workers=2):
    """BASELINE.json configs[2] as a secondary object: zero-shot with a 10 s prompt (30 prompt-text ids, 250 prompt speech
    tokens in the LM = a ~296-row prefill per sequence, 500 prompt mel frames: DiT sequence 650), batch 4, 75 forced tokens,
    steps one after the other on one stream."""
    from fangyan_tts_amd import synth
    from fangyan_tts_amd.cli.model import CosyVoice3Model
    B, P = 4, 250
    noise = torch.from_numpy(synth.flow_rand_noise(2 * (P + N_TOK))).to(dev)
    ri = torch.from_numpy(synth.hift_rand_ini()).to(dev)
    sn = torch.from_numpy(synth.hift_sine_noise(2 * N_TOK * 480)).to(dev)
    m = CosyVoice3Model(sd_llm, sd_flow, sd_hift, cfg, device=dev, max_batch=B, max_text=64, max_prompt_tokens=P, max_tokens=N_TOK,
                        rand_noise=noise, rand_ini=ri, sine_noise=sn)
    inputs = zero_shot_inputs(cfg, B, P)
    forced = [N_TOK] * B
    m.tts_batch(inputs, min_len=forced, max_len=forced)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wav, samples, toks = m.tts_batch(inputs, min_len=forced, max_len=forced)          # wavs end on the host
    dt = (time.perf_counter() - t0) / steps
    audio = sum(samples) / 24000.0
    # utterances 0 and 3 of the last timed step, for the check against the CPU oracle (check_zero_shot, in the cpu_baseline leg)
    keep = {"inputs": inputs, "noise": noise, "ri": ri, "sn": sn, "samples": samples, "wav": wav.clone(),
            "toks": [t.cpu() for t in toks], "mel": m.last_mel.float().cpu()}
    # where the time goes: the LM alone (prefill of 4 x 296 rows + 75 token steps), then every profiled stage of one step
    text = [d["text"].reshape(-1).tolist() for d in inputs]
    ptext = [d["prompt_text"].reshape(-1).tolist() for d in inputs]
    ptoks = [d["llm_prompt_speech_token"].reshape(-1).tolist() for d in inputs]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    m.llm.generate(text, ptext, ptoks, min_len=forced, max_len=forced)
    torch.cuda.synchronize()
    lm_ms = 1e3 * (time.perf_counter() - t1)
    prof = _stage_profile(L, _lib, lambda: m.tts_batch(inputs, min_len=forced, max_len=forced))
    ms, flops, n = prof["gemm_bf16"]
    ms_d, bytes_d, n_d = prof["llm_decode"]
    out = {"workload": "CosyVoice3-0.5B zero-shot, batch 4, 10 s prompt (250 LM prompt tokens, DiT sequence 650), 75 forced tokens each, unpipelined",
           "ms": round(1e3 * dt, 2), "audio_s_per_s": round(audio / dt, 2), "steps": steps, "lm_decode": "persistent" if m.llm.persistent else "per-op",
           "stage_split_ms": {"lm_generate_alone": round(lm_ms, 2), "flow_and_vocoder": round(1e3 * dt - lm_ms, 2),
                              "by_events_in_one_step": {"dit_linears": round(ms, 2), "vocoder_convs": round(prof["conv_mfma"][0], 2),
                                                        "lm_decode_launches": round(ms_d, 2), "lm_prefill_gemms": round(prof["gemm_exact"][0], 2),
                                                        "lm_8_or_32_row_products": round(prof["gemv"][0], 2)}},
           "roofline": dit_roofline(ms, flops, n, "HIP events on the launch stream around every DiT linear of one step (8 sequences x 650 rows: M = 5200)", None),
           "roofline_lm": {"bound": "hbm", "kernel": "llm_decode_k (persistent token step, 4 sequences)", "achieved": round(bytes_d / (ms_d * 1e-3) / 1e9, 1) if ms_d else None,
                           "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(bytes_d / (ms_d * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4) if ms_d else None,
                           "launches": n_d, "avg_launch_us": round(1e3 * ms_d / max(n_d, 1), 1)}}
    del m
    torch.cuda.empty_cache()
    # the same workload through the software pipeline of the headline (one LM call per 4 steps = 16 sequences per weight pass)
    mp = CosyVoice3Model(sd_llm, sd_flow, sd_hift, cfg, device=dev, max_batch=B, max_text=64, max_prompt_tokens=P, max_tokens=N_TOK,
                         rand_noise=noise, rand_ini=ri, sine_noise=sn, lm_group=4, flow_workers=flow_workers)
    mp.prepare_pipeline(0)
    k = 12

    def run_p():
        for wav_p, samples_p, _ in mp.tts_pipeline([inputs] * k, min_len=[forced] * k, max_len=[forced] * k, keep_on_device=True):
            wav_p.cpu()
    run_p()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    run_p()
    torch.cuda.synchronize()
    dtp = (time.perf_counter() - t2) / k
    out["pipelined"] = {"ms_per_step": round(1e3 * dtp, 2), "audio_s_per_s": round(audio / dtp, 2), "steps": k, "lm_group": 4, "flow_workers": flow_workers,
                        "note": "tts_pipeline as in the headline run: the LM of 4 steps' batches in one call beside the flow decoders of flow_workers steps"}
    mp.close()
    del mp
    torch.cuda.empty_cache()
    return out, keep


def check_zero_shot(cfg, sd_llm, sd_flow, sd_hift, keep, which=(0, 3)):
    """BASELINE.json configs[2] at full size: utterances 0 and 3 of the timed batch-4 step (10 s prompt: 296-row LM prefill, DiT
    sequence 650 = M 5200 for the batch) against the CPU oracle's whole per-utterance path - ids exact, mel <= 4e-2, waveform
    <= 2.5e-3 from the oracle vocoder on the engine's own mel (the same bars as the headline's `checked`)."""
    from oracle import hift as ohift, pipeline as opipe
    PL = {k: v.cpu() for k, v in sd_llm.items()}
    PF = {k: v.cpu() for k, v in sd_flow.items()}
    PH = ohift.prepare({k: v.cpu().numpy() for k, v in sd_hift.items()})
    noise, ri, sn = keep["noise"].cpu(), keep["ri"].cpu(), keep["sn"].cpu()
    res, ok = [], True
    t0 = time.time()
    for b in which:
        ref = opipe.tts(keep["inputs"][b], PL, PF, PH, cfg, noise, ri, sn, min_len=N_TOK, max_len=N_TOK)
        ids, ref_ids = keep["toks"][b].reshape(-1).tolist(), ref["tokens"].reshape(-1).tolist()
        n = len(ref_ids)
        mel = keep["mel"][b: b + 1, :, : 2 * n]
        e_mel = float((mel - ref["mel"]).abs().max()) if mel.shape == ref["mel"].shape else float("inf")
        S = keep["samples"][b]
        ref_wav, _ = ohift.inference(mel, PH, cfg.hift, ri, sn[:, :S])
        e_wav = float((keep["wav"][b: b + 1, :S] - ref_wav).abs().max())
        good = bool(ids == ref_ids and e_mel <= 4e-2 and e_wav <= 2.5e-3)
        ok &= good
        res.append({"utterance": b, "ids_equal": ids == ref_ids, "n_ids": n, "mel_max_abs_err": round(e_mel, 5),
                    "wav_vs_oracle_vocoder_on_engine_mel": round(e_wav, 6), "ok": good})
    return {"what": "utterances 0 and 3 of the last timed full-size batch-4 step vs the CPU oracle (oracle.pipeline.tts)", "mel_tol": 4e-2, "wav_tol": 2.5e-3,
            "utterances": res, "oracle_s": round(time.time() - t0, 1), "ok": bool(ok)}


def bench_latency_b1(cfg, sd_llm, sd_flow, sd_hift, dev, inputs, reps=4):
    """BASELINE.json configs[0]'s shape as a secondary object: ONE utterance alone (5 s prompt, 75 forced tokens = 3 s of audio)
    through tts_batch, wav on the host - what a single request waits (the LM decodes with the persistent one-launch-per-token step)."""
    from fangyan_tts_amd import synth
    from fangyan_tts_amd.cli.model import CosyVoice3Model
    m = CosyVoice3Model(sd_llm, sd_flow, sd_hift, cfg, device=dev, max_batch=1, max_text=64, max_prompt_tokens=P_TOK, max_tokens=N_TOK,
                        rand_noise=torch.from_numpy(synth.flow_rand_noise(2 * (P_TOK + N_TOK))).to(dev),
                        rand_ini=torch.from_numpy(synth.hift_rand_ini()).to(dev),
                        sine_noise=torch.from_numpy(synth.hift_sine_noise(2 * N_TOK * 480)).to(dev))
    one = inputs[:1]
    m.tts_batch(one, min_len=[N_TOK], max_len=[N_TOK])
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        wav, samples, _ = m.tts_batch(one, min_len=[N_TOK], max_len=[N_TOK])
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]
    out = {"workload": "one utterance alone (batch 1, 5 s prompt, 75 forced tokens), wav on the host", "ms": round(1e3 * dt, 2),
           "audio_s": round(samples[0] / 24000.0, 2), "times_real_time": round(samples[0] / 24000.0 / dt, 1),
           "lm_decode": "persistent" if m.llm.persistent else "per-op"}
    del m
    torch.cuda.empty_cache()
    return out


def bench_first_chunk(cfg, sd_llm, sd_flow, sd_hift, dev, inputs, reps=3):
    """stream=True on one utterance (5 s prompt): wall time from the call to the first audio chunk on the host - the figure the
    reference quotes as "latency as low as 150 ms" (CosyVoice/README.md:19) - and to the last one.  The LM is stepped on demand
    (persistent token step), the first chunk leaves after hop + look-ahead = 28 tokens (the 125 prompt tokens are a multiple of the
    hop of 25), flow decoder with the chunk mask over prompt + 28 tokens, vocoder over the 50 valid frames minus its look-ahead."""
    from fangyan_tts_amd import synth
    from fangyan_tts_amd.cli.model import CosyVoice3Model
    n_max = 20 * 20
    m = CosyVoice3Model(sd_llm, sd_flow, sd_hift, cfg, device=dev, max_batch=1, max_text=64, max_prompt_tokens=P_TOK, max_tokens=n_max,
                        rand_noise=torch.from_numpy(synth.flow_rand_noise(2 * (P_TOK + n_max))).to(dev),
                        rand_ini=torch.from_numpy(synth.hift_rand_ini()).to(dev),
                        sine_noise=torch.from_numpy(synth.hift_sine_noise(2 * n_max * 480)).to(dev))
    one = inputs[0]
    list(m.tts(**one, stream=True))
    torch.cuda.synchronize()
    first, total, chunks, audio = [], [], 0, 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        g = m.tts(**one, stream=True)
        c0 = next(g)["tts_speech"]
        t1 = time.perf_counter()
        rest = [c["tts_speech"] for c in g]
        t2 = time.perf_counter()
        first.append(t1 - t0); total.append(t2 - t0)
        chunks, audio = 1 + len(rest), (c0.shape[1] + sum(r.shape[1] for r in rest)) / 24000.0
    out = {"workload": "one utterance, stream=True, 5 s prompt, the LM's own stopping rule (random-init weights)", "first_chunk_ms": round(1e3 * sorted(first)[len(first) // 2], 2),
           "first_chunk_audio_s": round(c0.shape[1] / 24000.0, 3), "all_chunks_ms": round(1e3 * sorted(total)[len(total) // 2], 2), "chunks": chunks,
           "audio_s": round(audio, 2), "reference_claim": "latency as low as 150 ms (CosyVoice/README.md:19, other hardware)"}
    m.close()
    del m
    torch.cuda.empty_cache()
    return out


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary objects (zero-shot batch 4, HiFT-only config 5, ...)")
    ap.add_argument("--no-pipeline", action="store_true", help="run the steps strictly one after another")
    ap.add_argument("--llm-streams", type=int, default=1, help="LM handles decoding different steps' batches concurrently")
    ap.add_argument("--lm-isolate", action="store_true", help="LM streams run ONLY on the CUs the flow stream is kept off")
    ap.add_argument("--lm-group", type=int, default=4, help="consecutive steps whose LM decode runs as one call (32 rows per weight pass)")
    ap.add_argument("--flow-cu-exclude", type=int, default=0, help="CUs kept clear of the flow / vocoder stream")
    ap.add_argument("--flow-workers", type=int, default=2, help="consecutive steps whose flow decoder + vocoder run side by side (own handles and streams); "
                    "default 2: 51.9 against 60.6 ms per step since round 4 (the tails, epilogues and launch gaps of one step's kernels run under the "
                    "other's K loops).  Two DiT products then share the chip, so a launch's own duration is stretched: `roofline` keeps the "
                    "contract's per-launch figure and adds `chip_level` (flops / time with at least one of them running) and `one_step_alone`")
    ap.add_argument("--flow-group", type=int, default=1, help="consecutive steps whose batches go through the flow decoder + vocoder as ONE ragged batch "
                    "(not the default: the metric's step is a batch of 8; see DESIGN.md section 10)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no outer launcher: start one rank per GPU ourselves (the reference's own multi-GPU inference pattern,
        # CosyVoice/runtime/triton_trtllm/offline_inference.py:312-322).  A child process, started before this one has
        # touched the GPU; its stdout (rank 0's JSON line) and exit code are relayed.
        sys.exit(spawn_ranks(a.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # FY_BENCH_REHEARSAL=1: several ranks share GPU 0 over gloo (a 1-GPU box cannot hold an RCCL group); the
    # audio is gathered through host memory.  Only for exercising the N > 1 code path, never for numbers.
    rehearsal = os.environ.get("FY_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    from fangyan_tts_amd import _lib, build, synth
    from fangyan_tts_amd.cli.model import CosyVoice3Model
    from fangyan_tts_amd.spec import ModelCfg
    if not os.path.exists(_lib.LIB_PATH):
        if world > 1:
            raise RuntimeError("libfy_cosy3.so is missing: build it once before starting the ranks (python -m fangyan_tts_amd.build)")
        build.build(verbose=False)
    cfg = ModelCfg()
    log("generating synthetic weights on the GPU")
    sd_llm = synth.state_dict_torch(cfg.llm.manifest(), dev, skip=("lm_head",))
    sd_flow = synth.state_dict_torch(cfg.flow.manifest(), dev)
    sd_hift = synth.state_dict_torch(cfg.hift.manifest(), dev)
    T = 2 * (P_TOK + N_TOK)
    noise = torch.from_numpy(synth.flow_rand_noise(T)).to(dev)
    ri = torch.from_numpy(synth.hift_rand_ini()).to(dev)
    sn = torch.from_numpy(synth.hift_sine_noise(2 * N_TOK * 480)).to(dev)
    pipelined = not a.no_pipeline
    n_llm, group = (a.llm_streams, a.lm_group) if pipelined else (1, 1)
    model = CosyVoice3Model(sd_llm, sd_flow, sd_hift, cfg, device=dev, max_batch=BATCH, max_text=64, max_prompt_tokens=P_TOK,
                            max_tokens=N_TOK, rand_noise=noise, rand_ini=ri, sine_noise=sn, n_llm=n_llm, lm_group=group,
                            flow_workers=a.flow_workers if pipelined else 1, flow_group=a.flow_group if pipelined else 1)
    if pipelined:
        model.prepare_pipeline(a.flow_cu_exclude)        # stream placement on the hardware pipes: set-up, not part of a step
    log("engines ready")
    inputs = make_inputs(cfg, rank)
    forced = [N_TOK] * BATCH
    from fangyan_tts_amd.parallel import gather_audio

    S_MAX = N_TOK * 2 * cfg.hift.upsample_total
    last = {}

    def deliver(wav, samples, toks):
        """The metric's end point: all wavs resident on the host - after the all-gather for N > 1."""
        if world > 1:
            wav, _ = gather_audio(wav.cpu() if rehearsal else wav, samples, b_max=BATCH, s_max=S_MAX, reuse=not rehearsal)   # the GPU path copies the result to the host at once
        last["wav"], last["toks"], last["mel"] = wav.cpu(), toks, model.last_mel       # D2H inside the timed region
        return samples

    def step():
        return deliver(*model.tts_batch(inputs, min_len=forced, max_len=forced, keep_on_device=True))

    def fresh(ins):
        """The same utterances as NEW tensor objects: what a host that builds its model_input dicts per request presents (the
        prompt cache of cli/model.py is keyed by tensor identity, so every step pads and copies its prompts to the device)."""
        return [{k: v.clone() for k, v in d.items()} for d in ins]

    def run(k, fresh_inputs=False):
        """k steps.  Consecutive steps are software-pipelined over HIP streams: the speech-token LM decodes the batches of
        `lm_group` consecutive steps in ONE call (32 sequences per weight pass) beside the flow decoder + vocoder of the
        steps before; every step still runs the whole path on its own batch of 8, inside the timed region."""
        if not pipelined:
            return [step() for _ in range(k)][-1]
        samples = None
        steps_in = [fresh(inputs) for _ in range(k)] if fresh_inputs else [inputs] * k
        for out in model.tts_pipeline(steps_in, min_len=[forced] * k, max_len=[forced] * k, keep_on_device=True,
                                      flow_cu_exclude=a.flow_cu_exclude, lm_isolate=a.lm_isolate):
            samples = deliver(*out)
        return samples

    if a.warmup:
        samples = run(a.warmup)
        torch.cuda.synchronize()
        log(f"{a.warmup} warmup steps done")
    L = _lib.lib()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    cpu0 = time.process_time()                      # CPU seconds of this process, every thread (LM producer, feeder, flow workers, consumer)
    t0 = time.perf_counter()
    samples = run(a.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    host_cpu_ms = 1e3 * (time.process_time() - cpu0) / a.steps
    timed = {k: (v.clone() if torch.is_tensor(v) else [t.cpu() for t in v]) for k, v in last.items()}      # the last timed step's outputs
    if world > 1:
        t = torch.tensor([dt], device="cpu" if rehearsal else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    audio_per_step = world * sum(samples) / 24000.0
    value = audio_per_step * a.steps / dt
    log(f"timed {a.steps} steps: {1e3 * dt / a.steps:.1f} ms/step, {value:.1f} audio_s/s")

    # ---- the same K steps with FRESH input tensors per step (SURVEY 8d starts the clock at "model_input dicts resident on host"): the
    # prompt cache never hits, every step pads its prompts and copies them to the device (3 H2D copies + a stream wait per batch)
    value_fresh = None
    if pipelined:
        torch.cuda.synchronize()
        tf = time.perf_counter()
        run(a.steps, fresh_inputs=True)
        torch.cuda.synchronize()
        dtf = time.perf_counter() - tf
        if world > 1:
            t = torch.tensor([dtf], device="cpu" if rehearsal else dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtf = float(t.item())
        value_fresh = {"value": round(audio_per_step * a.steps / dtf, 3), "ms_per_step": round(1e3 * dtf / a.steps, 3),
                       "note": "the timed run repeated with new tensor objects for every step's model_input dicts: the prompt cache (keyed by tensor identity) "
                               "never hits, so prompt padding + 3 host-to-device copies + a stream wait per batch are inside the step"}
        log(f"fresh inputs: {1e3 * dtf / a.steps:.1f} ms/step")

    # ---- roofline of the timed configuration: the SAME K steps twice more, with HIP events (on the stream each kernel is
    # launched on) around every launch of one kernel family per pass - the DiT linears on the flow stream, then the LM's
    # decode products on the LM stream.  Not inside the timed pass itself: two event records per launch cost the recorded
    # stream a few per cent.  Event-to-event intervals of a kernel that shares the chip include the time its workgroups wait
    # for CUs the other stream holds, exactly as rocprofv3's begin/end stamps do (profiles/r05_bench_kernel_stats.csv).
    def events_pass(name):
        L.fy_prof_reset()
        L.fy_prof_only(name.encode())
        L.fy_prof_enable(1)
        run(a.steps)
        torch.cuda.synchronize()
        L.fy_prof_enable(0)
        L.fy_prof_only(None)
        r = _lib.prof_get(name) + (_lib.prof_union(name),)
        L.fy_prof_reset()
        return r
    ms_t, flops_t, n_t, union_t = events_pass("gemm_bf16")
    # the LM's launches in the timed configuration: the few-CU persistent 32-row step (one launch per token step) when the pipeline decodes
    # 9 .. 32 sequences per call, else the per-operation products
    lm32 = pipelined and BATCH * group > 8 and os.environ.get("FY_PIPE_LM_PERSISTENT32", "1") != "0" and os.environ.get("FY_LLM_PERSISTENT32", "1") != "0"
    ms_vt, bytes_vt, n_vt, _ = events_pass("llm_decode32" if lm32 else "gemv")

    # ---- one batch alone (no pipelining): what tts_batch takes, LM on the persistent one-launch token step
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    latency_ms = 1e3 * (time.perf_counter() - t1)
    prof = _stage_profile(L, _lib, step)
    ms, flops, n = prof["gemm_bf16"]

    where_t = (f"HIP events on the launch stream around every launch, over a repeat of the {a.steps} timed steps in the timed configuration "
               f"(pipelined: {pipelined})")
    roofline = dit_roofline(ms_t, flops_t, n_t, where_t, _pmc("r05_bench_pmc.json", "families", "dit_linears", "traffic_bytes"))
    roofline["why_this_kernel"] = ("the kernel family with the largest share of GPU time in the timed region (profiles/r05_bench_kernel_stats.csv); "
                                   "bound MFMA: 2 M N K flop per launch, SURVEY 8(d)")
    n_side = a.flow_workers if pipelined else 1
    if n_side > 1 and union_t > 0:
        # flow_workers steps' flow decoders run side by side, each on its own stream: two of these launches share the chip most of the
        # time, so a launch's own duration (above, as the contract defines `achieved`) is stretched by its neighbour.  The rate the
        # chip reaches on this kernel family = the flops of all launches / the time during which at least one of them was running.
        roofline["launches_side_by_side"] = n_side
        roofline["chip_level"] = {"achieved": round(flops_t / (union_t * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                                  "frac": round(flops_t / (union_t * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                  "busy_ms_per_step": round(union_t / a.steps, 3), "sum_of_launch_ms_per_step": round(ms_t / a.steps, 3),
                                  "what": "sum of 2MNK over every launch / the union of the launches' event intervals (both flow streams)"}
    roofline["one_step_alone"] = dit_roofline(ms, flops, n, "the same events over one un-pipelined step (nothing else on the GPU)", None)
    roofline["stage_ms_per_step_alone"] = {k: round(v[0], 3) for k, v in prof.items()}

    # ---- the LM's decode (second by GPU time): HBM-bound by SURVEY 8(d) - every bf16 weight is streamed once per token step, for 32 rows
    # (4 steps' batches) per pass.  algorithmic bytes = the weights (727.8 MB per token step).  Measured in the timed configuration
    # (events over the repeat of the timed steps, above) and alone, for both forms: the few-CU persistent step (what the pipeline
    # runs since round 4) and the per-operation products of round 3.
    text = [d["text"].reshape(-1).tolist() for d in inputs]
    ptext = [d["prompt_text"].reshape(-1).tolist() for d in inputs]
    G = group
    eng = model.llms[0]
    was = eng.persistent
    step_bytes = 727810048

    def lm_alone(persistent, name):
        eng.set_decode_mode(persistent)
        gen = lambda: eng.generate(text * G, ptext * G, [[] for _ in range(BATCH * G)], min_len=forced * G, max_len=forced * G)
        gen()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        gen()
        torch.cuda.synchronize()
        ms_call = 1e3 * (time.perf_counter() - t2)
        ms_k, bytes_k, n_k = _stage_profile(L, _lib, gen, names=(name,))[name]
        gbps = bytes_k / (ms_k * 1e-3) / 1e9 if ms_k > 0 else 0.0
        return {"achieved": round(gbps, 1), "frac": round(gbps / PEAK_HBM_GBPS, 4), "avg_launch_us": round(1e3 * ms_k / max(n_k, 1), 2), "launches": n_k,
                "generate_ms": round(ms_call, 2), "rows": BATCH * G, "ms_per_token_step": round(ms_call / N_TOK, 4),
                "weight_GBps_at_token_step_level": round(step_bytes / (ms_call / N_TOK * 1e-3) / 1e9, 1)}
    alone32 = lm_alone(True, "llm_decode32") if BATCH * G > 8 else None
    alone_ops = lm_alone(False, "gemv")
    eng.set_decode_mode(was)
    gb_t = bytes_vt / (ms_vt * 1e-3) / 1e9 if ms_vt > 0 else 0.0
    if lm32:
        kname = (f"llm_decode32_k: ONE persistent launch per token step for {BATCH * G} sequences on 76 workgroups / CUs (24 layers + llm_decoder, weights "
                 "HBM -> registers a tile ahead, operands as A images in registers, 120 grid-wide hand-offs)")
    else:
        kname = f"LM decode products at {BATCH * G} rows per weight pass: gemv32_k (qkv, o-proj, gate/up, down of 24 layers + llm_decoder; 97 launches per token step)"
    roofline_lm = {"bound": "hbm", "kernel": kname,
                   "achieved": round(gb_t, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(gb_t / PEAK_HBM_GBPS, 4),
                   "traffic": _pmc("r05_bench_pmc.json", "kernels", "llm_decode32_k" if lm32 else "gemv32_k", "traffic_bytes"),
                   "traffic_unit": "HBM-side bytes per launch, mean over every launch of the run (rocprofv3 --pmc over bench.py itself: profiles/r05_bench_pmc.json)",
                   "launches": n_vt, "avg_launch_us": round(1e3 * ms_vt / max(n_vt, 1), 2), "algorithmic_bytes_per_launch": int(bytes_vt / max(n_vt, 1)),
                   "measured_over": where_t,
                   "one_generation_alone": alone32 if lm32 else alone_ops,
                   "per_operation_products_alone": alone_ops,
                   "note": "alone: one LM call of 32 sequences with nothing else on the GPU (prefill included in generate_ms)"}

    # ---- the same decode as ONE persistent launch per token step (llm_decode.hip; what tts / tts_batch / stream=True use for up
    # to 8 sequences): HIP events around every launch of a generation run alone.
    roofline_lm_p = None
    eng.set_decode_mode(True)
    if eng.persistent:
        none8 = [[] for _ in inputs]
        eng.generate(text, ptext, none8, min_len=forced, max_len=forced)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        eng.generate(text, ptext, none8, min_len=forced, max_len=forced)
        torch.cuda.synchronize()
        lm_p_ms = 1e3 * (time.perf_counter() - t3)
        pp = _stage_profile(L, _lib, lambda: eng.generate(text, ptext, none8, min_len=forced, max_len=forced), names=("llm_decode",))
        ms_p, bytes_p, n_p = pp["llm_decode"]
        gb = bytes_p / (ms_p * 1e-3) / 1e9 if ms_p > 0 else 0.0
        roofline_lm_p = {"bound": "hbm", "kernel": "llm_decode_k: one persistent launch per token step (24 layers + llm_decoder, 152 workgroups, "
                         "weights register-resident a layer ahead, 121 grid-wide hand-offs)", "achieved": round(gb, 1), "peak": PEAK_HBM_GBPS,
                         "unit": "GB/s", "frac": round(gb / PEAK_HBM_GBPS, 4),
                         "traffic": _pmc("r05_bench_pmc.json", "kernels", "llm_decode_k", "traffic_bytes"),
                         "traffic_unit": "HBM-side bytes per launch (rocprofv3 --pmc over bench.py itself: profiles/r05_bench_pmc.json, kernels.llm_decode_k.traffic_bytes)",
                         "launches": n_p, "avg_launch_us": round(1e3 * ms_p / max(n_p, 1), 1), "algorithmic_bytes_per_launch": int(bytes_p / max(n_p, 1)),
                         "generate_ms_batch8_75_tokens": round(lm_p_ms, 2),
                         "measured_over": "HIP events on the launch stream around every launch of one 75-token generation at batch 8, run alone"}
    eng.set_decode_mode(was)

    # ---- the price of the reference's own estimator-swap bar (rtol 1e-2 / atol 1e-4, export_onnx.py:109): the flow decoder in its
    # fp32-class mode (FY_PRECISE: split operands, fp32 attention), un-pipelined steps
    precise = None
    zs_keep = None
    if rank == 0 and world == 1 and not a.no_extras:
        model.flow_flags = _lib.FY_PRECISE
        try:
            step()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            p_ms = 1e3 * (time.perf_counter() - t4) / 3
            pp_ms = None
            if pipelined:                                   # and the headline's own pipelined run in that mode
                run(4)
                torch.cuda.synchronize()
                t5 = time.perf_counter()
                run(a.steps)
                torch.cuda.synchronize()
                pp_ms = 1e3 * (time.perf_counter() - t5) / a.steps
        finally:
            model.flow_flags = 0
        precise = {"what": "one batch alone with the flow decoder in FY_PRECISE mode (meets rtol 1e-2 / atol 1e-4 against the fp32 estimator, tests/test_flow_gpu.py)",
                   "ms": round(p_ms, 2), "audio_s_per_s": round(audio_per_step / (p_ms * 1e-3), 2), "default_mode_ms": round(latency_ms, 2)}
        if pp_ms:
            precise["pipelined"] = {"ms_per_step": round(pp_ms, 2), "audio_s_per_s": round(audio_per_step / (pp_ms * 1e-3), 2), "steps": a.steps,
                                    "note": "the timed run of the headline repeated with the flow decoder in FY_PRECISE"}

    # ---- the price of a GENERAL fp32 checkpoint in the LM (llm.pt is fp32, cli/model.py:65-73; the reference runs it as saved): every
    # matrix as two bf16 planes w = hi + lo - the mode fy_llm_create picks by itself when a weight is not bf16-representable, and in
    # which the ids equal the reference's on unrounded weights (tests/test_fp32w_gpu.py).  Same workload and pipeline as the headline,
    # a second model whose LM is FORCED to two planes on the same weights (their lo planes are zero: same ids, the cost of the mode).
    exact_w = None
    if rank == 0 and world == 1 and not a.no_extras:
        log("the same workload with the LM in its exact-weights mode (two bf16 planes per matrix)")
        model2 = CosyVoice3Model(sd_llm, sd_flow, sd_hift, cfg, device=dev, max_batch=BATCH, max_text=64, max_prompt_tokens=P_TOK,
                                 max_tokens=N_TOK, rand_noise=noise, rand_ini=ri, sine_noise=sn, n_llm=n_llm, lm_group=group,
                                 flow_workers=a.flow_workers if pipelined else 1, flow_group=a.flow_group if pipelined else 1, llm_weight_planes=2)
        try:
            assert all(e.weight_planes == 2 for e in model2.llms)
            step2 = lambda: model2.tts_batch(inputs, min_len=forced, max_len=forced, keep_on_device=True)

            def run2(k):
                r = None
                for r in model2.tts_pipeline([inputs] * k, min_len=[forced] * k, max_len=[forced] * k, keep_on_device=True):
                    r[0].cpu()
                return r
            wav2, _, toks2 = step2()
            torch.cuda.synchronize()
            ids_equal = all(torch.equal(x.cpu(), y.cpu()) for x, y in zip(toks2, model.tts_batch(inputs, min_len=forced, max_len=forced, keep_on_device=True)[2]))
            t6 = time.perf_counter()
            for _ in range(3):
                step2()[0].cpu()
            torch.cuda.synchronize()
            xw_ms = 1e3 * (time.perf_counter() - t6) / 3
            xw_pipe = None
            if pipelined:
                model2.prepare_pipeline(a.flow_cu_exclude)
                run2(a.warmup or 4)
                torch.cuda.synchronize()
                t7 = time.perf_counter()
                run2(a.steps)
                torch.cuda.synchronize()
                xw_pipe = 1e3 * (time.perf_counter() - t7) / a.steps
            e2 = model2.llms[0]
            e2.set_decode_mode(2)
            gen2 = lambda: e2.generate(text * G, ptext * G, [[] for _ in range(BATCH * G)], min_len=forced * G, max_len=forced * G)
            gen2()
            torch.cuda.synchronize()
            t8 = time.perf_counter()
            gen2()
            torch.cuda.synchronize()
            xw_gen = 1e3 * (time.perf_counter() - t8)
            exact_w = {"what": "the headline's workload with the LM storing every matrix as two bf16 planes (w = hi + lo: a general fp32 llm.pt keeps its "
                               "values to 2^-17 relative; ids equal the reference's on unrounded weights, tests/test_fp32w_gpu.py); 1.456 GB of weights per token step",
                       "ms_per_step_pipelined": round(xw_pipe, 3) if xw_pipe else None,
                       "audio_s_per_s_pipelined": round(audio_per_step / (xw_pipe * 1e-3), 2) if xw_pipe else None,
                       "ms_per_step_unpipelined": round(xw_ms, 2), "audio_s_per_s_unpipelined": round(audio_per_step / (xw_ms * 1e-3), 2),
                       "one_plane_ms_per_step": {"pipelined": round(1e3 * dt / a.steps, 3), "unpipelined": round(latency_ms, 2)},
                       "lm_call_alone_ms": {"rows": BATCH * G, "tokens": N_TOK, "ms": round(xw_gen, 2), "ms_per_token_step": round(xw_gen / N_TOK, 4),
                                            "weight_GBps_at_token_step_level": round(2 * step_bytes / (xw_gen / N_TOK * 1e-3) / 1e9, 1)},
                       "ids_equal_one_plane_run": bool(ids_equal), "steps": a.steps}
        finally:
            model2.close()
            del model2
            torch.cuda.empty_cache()

    out = {
        "metric": "synthesised audio sec/sec (RTF^-1) CosyVoice3-0.5B instruct, batch 8",
        "value": round(value, 3), "unit": "audio_s/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "CosyVoice3-0.5B instruct (inference_instruct2), batch 8 mixed-length utterances per GPU, "
                               "5 s prompt, 75 forced speech tokens (3 s) each, LM greedy -> 10-step CFG flow (DiT-22) -> HiFT",
                   "batch_per_gpu": BATCH, "tokens_per_utt": N_TOK, "prompt_tokens": P_TOK, "parallelism": f"dp{world}",
                   "steps_pipelined": pipelined, "llm_streams": n_llm, "lm_group": group, "flow_workers": a.flow_workers if pipelined else 1, "flow_group": a.flow_group if pipelined else 1,
                   "lm_rows_per_weight_pass": BATCH * group,
                   "utterances_in_flight_max": BATCH * (3 * group * n_llm + a.flow_workers) if pipelined else BATCH,
                   "in_flight_note": "one LM call decodes lm_group steps' batches together; the ids of up to 2 x lm_group finished batches wait in a queue; flow_workers batches are in the flow decoder / vocoder side by side",
                   "flow_cu_exclude": a.flow_cu_exclude, "batch_latency_ms_unpipelined": round(latency_ms, 1),
                   "weights": "random-init, CosyVoice3-0.5B shapes (859 M params)"},
        # the strict batch-8 figure: one batch at a time, nothing of another batch in flight (tts_batch; LM on the persistent step)
        "value_batch8_unpipelined": round(audio_per_step / (latency_ms * 1e-3), 2),
        "value_fresh_inputs": value_fresh,
        # CPU time this rank's process spent per timed step, all threads (time.process_time over the timed region): what 8 ranks on
        # one host need of it (x 8) against the host's cores
        "host_cpu_ms_per_step": round(host_cpu_ms, 2),
        "roofline": roofline,
        "roofline_lm": roofline_lm,
        "roofline_lm_persistent": roofline_lm_p,
        "precise_mode": precise,
        "exact_weights": exact_w,
    }
    if rank == 0 and world == 1 and not a.no_extras:
        # behind the timed region: BASELINE.json configs[0], configs[2] and configs[4] as secondary objects of the same record
        log("one utterance alone (config 1's shape)")
        out["latency_b1"] = bench_latency_b1(cfg, sd_llm, sd_flow, sd_hift, dev, inputs)
        log("first streaming chunk")
        out["first_chunk"] = bench_first_chunk(cfg, sd_llm, sd_flow, sd_hift, dev, inputs)
        log("zero-shot batch 4 (config 3)")
        out["zero_shot_b4"], zs_keep = bench_zero_shot(cfg, sd_llm, sd_flow, sd_hift, dev, L, _lib, flow_workers=a.flow_workers)
        log("HiFT-only 32 x 10 000 frames (config 5)")
        model.close()
        del model, eng
        torch.cuda.empty_cache()
        import bench_hift
        h5 = bench_hift.run(steps=2, warmup=1)
        out["hift_cfg5"] = {"workload": h5["config"]["workload"], "ms": h5["ms_per_step"], "audio_s_per_s": h5["value"], "roofline": h5["roofline"],
                            "checked": h5["checked"]}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        log("timing the CPU oracle on one utterance")
        out["cpu_baseline"], out["checked"] = cpu_baseline(cfg, sd_llm, sd_flow, sd_hift, inputs[0], noise, ri, sn, timed)
        if zs_keep is not None:
            log("checking the zero-shot batch against the CPU oracle")
            out["zero_shot_b4"]["checked"] = check_zero_shot(cfg, sd_llm, sd_flow, sd_hift, zs_keep)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
