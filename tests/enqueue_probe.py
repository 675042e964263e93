"""How long does the HOST need to enqueue one batch of flow decoder + vocoder launches (~1700)?  The stream is first blocked by a
long spin kernel, so the call's wall time is pure enqueue time - unless the runtime bounds the number of launches in flight, in
which case the call returns only when the GPU has caught up.  (gpurun: python tests/enqueue_probe.py)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from fangyan_tts_amd import synth
from fangyan_tts_amd.cli.model import CosyVoice3Model
from fangyan_tts_amd.spec import ModelCfg

dev = torch.device("cuda:0")
cfg = ModelCfg()
sd = [synth.state_dict_torch(m.manifest(), dev, skip=("lm_head",)) for m in (cfg.llm, cfg.flow, cfg.hift)]
N, P = bench.N_TOK, bench.P_TOK
m = CosyVoice3Model(sd[0], sd[1], sd[2], cfg, device=dev, max_batch=8, max_text=64, max_prompt_tokens=P, max_tokens=N,
                    rand_noise=torch.from_numpy(synth.flow_rand_noise(2 * (P + N))).to(dev), rand_ini=torch.from_numpy(synth.hift_rand_ini()).to(dev),
                    sine_noise=torch.from_numpy(synth.hift_sine_noise(2 * N * 480)).to(dev))
inputs = bench.make_inputs(cfg, 0)
forced = [N] * 8
wav, samples, toks = m.tts_batch(inputs, min_len=forced, max_len=forced)
out = torch.stack([t for t in toks]).to(dev)
n_tok = [N] * 8
for blocked_ms in (0, 300):
    for rep in range(3):
        torch.cuda.synchronize()
        if blocked_ms:
            torch.cuda._sleep(int(blocked_ms * 1e-3 * 2.1e9))          # ~blocked_ms of spinning in front of the launches
        t0 = time.perf_counter()
        w, s = m._token2wav(inputs, out, n_tok, 1.0)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"stream blocked for ~{blocked_ms} ms: enqueue returned after {1e3 * (t1 - t0):.1f} ms, GPU done after {1e3 * (t2 - t0):.1f} ms", flush=True)
