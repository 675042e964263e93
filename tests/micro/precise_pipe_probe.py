"""Not a test: the headline's pipelined run with the flow decoder in FY_PRECISE, for A/B of the split-operand GEMM tilings
(FY_GEMM_SPLIT_TILE=256).  python tests/micro/precise_pipe_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from fangyan_tts_amd import _lib, synth
from fangyan_tts_amd.cli.model import CosyVoice3Model
from fangyan_tts_amd.spec import ModelCfg
dev = torch.device("cuda:0")
cfg = ModelCfg()
sd = [synth.state_dict_torch(m.manifest(), dev, skip=("lm_head",) if i == 0 else ()) for i, m in enumerate((cfg.llm, cfg.flow, cfg.hift))]
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
inputs = bench.make_inputs(cfg, 0)
N = bench.N_TOK
m = CosyVoice3Model(sd[0], sd[1], sd[2], cfg, device=dev, max_batch=8, max_text=64, max_prompt_tokens=bench.P_TOK, max_tokens=N, lm_group=4, flow_workers=2)
m.flow_flags = _lib.FY_PRECISE
forced = [N] * 8
def run(n):
    for w, s, _ in m.tts_pipeline([inputs] * n, min_len=[forced] * n, max_len=[forced] * n, keep_on_device=True):
        w.cpu()
run(4)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); run(k); torch.cuda.synchronize()
    print(f"FY_PRECISE pipelined, split tile {os.environ.get('FY_GEMM_SPLIT_TILE', 'default')}: {1e3 * (time.perf_counter() - t0) / k:.2f} ms per step", flush=True)
m.close()
