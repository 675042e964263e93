"""Not a test: how much of the B = 8 LM time is the prefill?  python tests/prefill_probe.py (through gpurun)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from fangyan_tts_amd import synth
from fangyan_tts_amd.llm import LlmEngine
from fangyan_tts_amd.spec import ModelCfg

dev = torch.device("cuda:0")
cfg = ModelCfg()
sd = synth.state_dict_torch(cfg.llm.manifest(), dev, skip=("lm_head",))
llm = LlmEngine(sd, cfg.llm, max_batch=8, max_ctx=2 + 64 + bench.P_TOK + bench.N_TOK)
inputs = bench.make_inputs(cfg, 0)
text = [d["text"].reshape(-1).tolist() for d in inputs]
ptext = [d["prompt_text"].reshape(-1).tolist() for d in inputs]
print("prefill rows per sequence:", [2 + len(a) + len(b) for a, b in zip(text, ptext)])
for n in (1, 2, 9, 75):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        llm.generate(text, ptext, [[] for _ in inputs], min_len=[n] * 8, max_len=[n] * 8)
        torch.cuda.synchronize()
        dt = 1e3 * (time.perf_counter() - t0)
    print(f"generate with {n:2d} tokens: {dt:.2f} ms", flush=True)
