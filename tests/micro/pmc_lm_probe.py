"""Driver for `rocprofv3 --pmc` over the persistent LM decode kernel (and the per-operation products) inside a torch process.
Under --pmc the profiler dies in torch's big elementwise launches that fill the synthetic 0.5 B-parameter model on the GPU
(SIGSEGV below at::native::mul_kernel_cuda in the launch path, gpurun_out/pmc_bench.err) while small torch kernels and every
kernel of this library run fine (tests/micro/pmc_torch_probe.py), so here the weights are generated on the CPU and copied.
Run as:  LD_LIBRARY_PATH=/opt/rocm/lib LD_PRELOAD="libamdhip64.so libhsa-runtime64.so" rocprofv3 --pmc FETCH_SIZE ... -- python3 this.py"""
import dataclasses
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from fangyan_tts_amd import synth
from fangyan_tts_amd.llm import LlmEngine
from fangyan_tts_amd.spec import LlmCfg

cfg = dataclasses.replace(LlmCfg(), vocab=2048)        # the text-embedding table plays no part in a decode step
man = {k: v for k, v in cfg.manifest().items() if "lm_head" not in k}
sd = {k: torch.from_numpy(v).cuda() for k, v in synth.state_dict(man).items()}
print("weights on the GPU", flush=True)
for persistent, B in ((True, 8), (True, 32), (False, 32)):       # 8-row persistent step; few-CU 32-row persistent step; per-operation products at 32 rows
    text = [[(7 * i + b) % 2000 for i in range(12 + b % 9)] for b in range(B)]
    ptext = [[(11 * i + b) % 2000 for i in range(8)] for b in range(B)]
    eng = LlmEngine(sd, cfg, max_batch=B, max_ctx=256)
    eng.set_decode_mode(persistent)
    out, n, _ = eng.generate(text, ptext, [[] for _ in range(B)], min_len=[6] * B, max_len=[6] * B)
    torch.cuda.synchronize()
    print("persistent" if persistent else "per-op", B, out[0, :6].tolist(), flush=True)
    eng.close()
