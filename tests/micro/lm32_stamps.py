# per-phase time stamps of the few-CU persistent 32-row decode step (workgroup 0, 100 MHz clock)
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from fangyan_tts_amd import _lib, synth
from fangyan_tts_amd.llm import LlmEngine
from fangyan_tts_amd.spec import ModelCfg
cfg = ModelCfg(); dev = torch.device("cuda:0")
sd = synth.state_dict_torch(cfg.llm.manifest(), dev, skip=("lm_head",))
eng = LlmEngine(sd, cfg.llm, max_batch=32, max_ctx=2 + 64 + 125 + 75, device=dev)
eng.set_decode_mode(True)
inputs = bench.make_inputs(cfg, 0)
text = [d["text"].reshape(-1).tolist() for d in inputs] * 4
ptext = [d["prompt_text"].reshape(-1).tolist() for d in inputs] * 4
forced = [75] * 32
eng.begin(text, ptext, [[] for _ in range(32)], min_len=forced, max_len=forced)
eng.step(40)
L = _lib.lib()
n = 2 + 11 * cfg.llm.layers + 1
buf = (C.c_uint64 * n)()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(L.fy_debug_decode32_stamps(eng._h, buf, n, st))      # arms
eng.step(3)
torch.cuda.synchronize()
_lib.check(L.fy_debug_decode32_stamps(eng._h, buf, n, st))
t = [buf[i] for i in range(n)]
names = ["P1 qkv", "handoff", "P2 attention", "handoff", "P3 o-proj", "handoff", "P4a gate/up", "P4b down", "handoff", "P5 reduce", "handoff"]
tot = [0.0] * 11
for l in range(cfg.llm.layers):
    for i in range(11):
        tot[i] += (t[1 + 11 * l + i] - t[11 * l + i]) * 0.01
print("us per layer (mean over layers), workgroup 0:")
for nm, v in zip(names, tot):
    print("  %-14s %6.2f" % (nm, v / cfg.llm.layers))
print("  layer total    %6.2f ; head %.1f us ; whole launch %.1f us" % (sum(tot) / cfg.llm.layers, (t[1 + 11 * cfg.llm.layers] - t[11 * cfg.llm.layers]) * 0.01, (t[1 + 11 * cfg.llm.layers] - t[0]) * 0.01))
