"""Not a test: the LM prefill on the ring kernel's three-plane products (gemm_exact3) against the register-staged exact-split kernel
(FY_LLM_PREFILL_RING=0): zero-shot shape (4 x 296 rows) and the pipeline's (32 sequences x ~25 rows = 800 rows); ids must agree."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from fangyan_tts_amd import synth
from fangyan_tts_amd.llm import LlmEngine
from fangyan_tts_amd.spec import ModelCfg
dev = torch.device("cuda:0")
cfg = ModelCfg()
sd = synth.state_dict_torch(cfg.llm.manifest(), dev, skip=("lm_head",))
inputs = bench.make_inputs(cfg, 0)
text = [d["text"].reshape(-1).tolist() for d in inputs] * 4
ptext = [d["prompt_text"].reshape(-1).tolist() for d in inputs] * 4
hi = 151643
zs_text = [synth.randint(f"pp.t{b}", (1, 14), 0, hi)[0].tolist() for b in range(4)]
zs_ptext = [synth.randint(f"pp.p{b}", (1, 30), 0, hi)[0].tolist() for b in range(4)]
zs_ptok = [synth.randint(f"pp.k{b}", (1, 250), 0, 6561)[0].tolist() for b in range(4)]
outs = {}
for mode in ("1", "0"):
    os.environ["FY_LLM_PREFILL_RING"] = mode
    llm = LlmEngine(sd, cfg.llm, max_batch=32, max_ctx=2 + 64 + 250 + bench.N_TOK)
    for name, (t, p, k) in (("pipeline 32 x ~25 rows", (text, ptext, [[] for _ in text])), ("zero-shot 4 x 296 rows", (zs_text, zs_ptext, zs_ptok))):
        B = len(t)
        for n in (1, 24):
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out, out_n, _ = llm.generate(t, p, k, min_len=[n] * B, max_len=[n] * B)
                torch.cuda.synchronize()
                dt = 1e3 * (time.perf_counter() - t0)
            if n == 24:
                outs[(mode, name)] = out.cpu()
            print(f"ring={mode} {name}: prefill + {n} token(s) {dt:.2f} ms", flush=True)
    del llm
for name in ("pipeline 32 x ~25 rows", "zero-shot 4 x 296 rows"):
    print(name, "ids equal:", bool(torch.equal(outs[("1", name)], outs[("0", name)])))
