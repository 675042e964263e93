"""Not a test: B = 8 LM decode, persistent one-launch step (llm_decode.hip) against the multi-launch path.
python tests/lm_decode_probe.py (through gpurun); FY_LLM_PERSISTENT selects the path at handle creation."""
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
res = {}
for mode in ("1", "0"):
    os.environ["FY_LLM_PERSISTENT"] = mode
    llm = LlmEngine(sd, cfg.llm, max_batch=8, max_ctx=2 + 64 + bench.P_TOK + bench.N_TOK)
    for B in (8, 1):
        for n in (1, 75):
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out, out_n, _ = llm.generate(text[:B], ptext[:B], [[] for _ in range(B)], min_len=[n] * B, max_len=[n] * B)
                torch.cuda.synchronize()
                dt = 1e3 * (time.perf_counter() - t0)
            res[(mode, B, n)] = (dt, out.cpu())
            print(f"persistent={mode} B={B} generate {n:2d} tokens: {dt:.2f} ms", flush=True)
    print(f"persistent={mode}: B=8 per decode step {(res[(mode, 8, 75)][0] - res[(mode, 8, 1)][0]) / 74:.4f} ms, "
          f"B=1 {(res[(mode, 1, 75)][0] - res[(mode, 1, 1)][0]) / 74:.4f} ms", flush=True)
    llm.close()
for B in (8, 1):
    print(f"B={B}: ids equal between the two paths:", bool(torch.equal(res[("1", B, 75)][1], res[("0", B, 75)][1])))

# phase time stamps of one decode step (persistent kernel, workgroup 0)
import ctypes as C
from fangyan_tts_amd import _lib
os.environ["FY_LLM_PERSISTENT"] = "1"
llm = LlmEngine(sd, cfg.llm, max_batch=8, max_ctx=2 + 64 + bench.P_TOK + bench.N_TOK)
L = _lib.lib()
N = 2 + 18 * cfg.llm.layers + 8
buf = (C.c_uint64 * N)()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
L.fy_debug_decode_stamps(llm._h, buf, N, st)                 # arm
llm.generate(text, ptext, [[] for _ in range(8)], min_len=[20] * 8, max_len=[20] * 8)
torch.cuda.synchronize()
L.fy_debug_decode_stamps(llm._h, buf, N, st)
t = [int(x) for x in buf]
names = ["P1 staged", "P1 products", "P1 epilogue", "P1 -> P2 (granules, no hand-off)", "P2 attention", "P2 hand-off", "P3 staged", "P3 products", "P3 epilogue",
         "P3 hand-off", "P4 staged", "P4 gate/up", "P4 swiglu", "P4 down", "P4 stores", "P4 hand-off", "P5 reduce", "P5 hand-off"]
print(f"P0 + hand-off: {(t[1] - t[0]) / 100:.2f} us")
for layer in (0, 1, 12, 23):
    base = 2 + 18 * layer
    prev = t[base - 1]
    row = []
    for i, nm in enumerate(names):
        row.append(f"{nm} {(t[base + i] - prev) / 100:.2f}")
        prev = t[base + i]
    print(f"layer {layer}: total {(t[base + 17] - t[base - 1]) / 100:.2f} us | " + " | ".join(row))
print(f"whole step: {(t[2 + 18 * 24 + 0] - t[0]) / 100:.1f} us to the first head chunk")
