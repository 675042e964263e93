# stream=True on one utterance (bench.py's first_chunk workload): first chunk / all chunks under the environment's switches
# (FY_STREAM_LM_AHEAD, FY_STREAM_LM_MODE, FY_GEMM_DEEP); prints one line
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from fangyan_tts_amd import synth
from fangyan_tts_amd.cli.model import CosyVoice3Model
from fangyan_tts_amd.spec import ModelCfg
cfg = ModelCfg(); dev = torch.device("cuda:0")
sd = [synth.state_dict_torch(m.manifest(), dev, skip=("lm_head",)) for m in (cfg.llm, cfg.flow, cfg.hift)]
inputs = bench.make_inputs(cfg, 0)
n_max = 400
m = CosyVoice3Model(sd[0], sd[1], sd[2], cfg, device=dev, max_batch=1, max_text=64, max_prompt_tokens=bench.P_TOK, max_tokens=n_max,
                    rand_noise=torch.from_numpy(synth.flow_rand_noise(2 * (bench.P_TOK + n_max))).to(dev),
                    rand_ini=torch.from_numpy(synth.hift_rand_ini()).to(dev),
                    sine_noise=torch.from_numpy(synth.hift_sine_noise(2 * n_max * 480)).to(dev))
one = inputs[0]
ref = [c["tts_speech"] for c in m.tts(**one, stream=True)]
torch.cuda.synchronize()
first, total = [], []
for _ in range(5):
    t0 = time.perf_counter()
    g = m.tts(**one, stream=True)
    c0 = next(g)["tts_speech"]
    t1 = time.perf_counter()
    rest = [c["tts_speech"] for c in g]
    t2 = time.perf_counter()
    first.append(t1 - t0); total.append(t2 - t0)
same = all(torch.equal(a, b) for a, b in zip(ref, [c0] + rest))
import hashlib
h = hashlib.sha1(b"".join(c.numpy().tobytes() for c in ref)).hexdigest()[:12]
sw = " ".join(f"{k}={os.environ[k]}" for k in ("FY_STREAM_LM_AHEAD", "FY_STREAM_LM_MODE", "FY_GEMM_DEEP") if k in os.environ)
print("%-50s first chunk %.1f ms, all %d chunks %.1f ms (min %.1f)  repeatable %s  sha1 %s" % (sw or "(defaults)", 1e3 * sorted(first)[2], 1 + len(rest), 1e3 * sorted(total)[2], 1e3 * min(total), same, h), flush=True)
m.close()
