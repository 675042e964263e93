# LM alone at 32 rows per-op: time per generate (bench-like)
import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from fangyan_tts_amd import synth
from fangyan_tts_amd.llm import LlmEngine
from fangyan_tts_amd.spec import ModelCfg
cfg = ModelCfg(); dev = torch.device("cuda:0")
sd = synth.state_dict_torch(cfg.llm.manifest(), dev, skip=("lm_head",))
eng = LlmEngine(sd, cfg.llm, max_batch=32, max_ctx=2 + 64 + 125 + 75, device=dev)
import os
eng.set_decode_mode(os.environ.get("LM32_PERSISTENT", "0") == "1")
print("persistent:", eng.persistent)
inputs = bench.make_inputs(cfg, 0)
text = [d["text"].reshape(-1).tolist() for d in inputs] * 4
ptext = [d["prompt_text"].reshape(-1).tolist() for d in inputs] * 4
forced = [75] * 32
for _ in range(2):
    out, n, _ = eng.generate(text, ptext, [[] for _ in range(32)], min_len=forced, max_len=forced)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    out, n, _ = eng.generate(text, ptext, [[] for _ in range(32)], min_len=forced, max_len=forced)
torch.cuda.synchronize()
print("LM 32 rows generate: %.1f ms" % ((time.perf_counter() - t0) / 3 * 1e3), "ids checksum", int(out.sum()))
