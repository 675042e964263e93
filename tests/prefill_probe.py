"""Not a test: LM prefill time, tiled exact-split GEMMs against the 8-row products (FY_LLM_PREFILL_GEMM=0), at the benchmark's
instruct shape (8 sequences x ~25 rows) and the zero-shot shape (4 x 296 rows).  python tests/prefill_probe.py (through gpurun)"""
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
inputs = bench.make_inputs(cfg, 0)
text = [d["text"].reshape(-1).tolist() for d in inputs]
ptext = [d["prompt_text"].reshape(-1).tolist() for d in inputs]
hi = 151643
zs_text = [synth.randint(f"pp.t{b}", (1, 14), 0, hi)[0].tolist() for b in range(4)]
zs_ptext = [synth.randint(f"pp.p{b}", (1, 30), 0, hi)[0].tolist() for b in range(4)]
zs_ptok = [synth.randint(f"pp.k{b}", (1, 250), 0, 6561)[0].tolist() for b in range(4)]
outs = {}
for mode in ("1", "0"):
    os.environ["FY_LLM_PREFILL_GEMM"] = mode
    llm = LlmEngine(sd, cfg.llm, max_batch=8, max_ctx=2 + 64 + 250 + bench.N_TOK)
    for name, (t, p, k) in (("instruct 8 x ~25 rows", (text, ptext, [[] for _ in text])), ("zero-shot 4 x 296 rows", (zs_text, zs_ptext, zs_ptok))):
        B = len(t)
        for n in (1, 12):
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out, out_n, _ = llm.generate(t, p, k, min_len=[n] * B, max_len=[n] * B)
                torch.cuda.synchronize()
                dt = 1e3 * (time.perf_counter() - t0)
            if n == 12:
                outs[(mode, name)] = out.cpu()
            print(f"prefill_gemm={mode} {name}: prefill + {n} token(s) {dt:.2f} ms", flush=True)
    llm.close()
for name in ("instruct 8 x ~25 rows", "zero-shot 4 x 296 rows"):
    print(name, "ids equal between the two prefill paths:", bool(torch.equal(outs[("1", name)], outs[("0", name)])))
