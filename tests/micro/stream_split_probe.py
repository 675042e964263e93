# stream=True on one utterance (bench.py's first_chunk workload): where a chunk's time goes - the LM steps, the flow call, the vocoder call
# (each followed by a device synchronisation, so the parts add up to a little more than the un-instrumented chunk)
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
list(m.tts(**one, stream=True))
torch.cuda.synchronize()
ln = m.lanes[0]
acc = {}
def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc.setdefault(tag, []).append(1e3 * (time.perf_counter() - t))
        return r
    setattr(obj, name, g)
wrap(ln.llm, "step", "lm_step"); wrap(ln.llm, "begin", "lm_begin"); wrap(ln.flow, "inference", "flow"); wrap(ln.hift, "inference", "hift")
t0 = time.perf_counter()
chunks = list(m.tts(**one, stream=True))
tot = 1e3 * (time.perf_counter() - t0)
print("chunks", len(chunks), "total %.1f ms (instrumented)" % tot)
for k, v in acc.items():
    print("%-9s n=%3d sum %.1f ms : %s" % (k, len(v), sum(v), " ".join("%.1f" % x for x in v)))
