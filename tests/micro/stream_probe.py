# stream=True on one utterance (5 s prompt, the LM's own stopping rule): first chunk / all chunks, incremental flow chunks on and off
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from fangyan_tts_amd import synth
from fangyan_tts_amd.cli.model import CosyVoice3Model
from fangyan_tts_amd.spec import ModelCfg
cfg = ModelCfg(); dev = torch.device("cuda:0")
sd = [synth.state_dict_torch(m.manifest(), dev, skip=("lm_head",)) for m in (cfg.llm, cfg.flow, cfg.hift)]
inputs = bench.make_inputs(cfg, 0)
import sys as _s
n_text = int(_s.argv[1]) if len(_s.argv) > 1 else 0
if n_text:
    inputs[0] = dict(inputs[0], text=torch.from_numpy(synth.randint('probe.long', (1, n_text), 0, 151643)))
n_max = max(400, 20 * n_text)
for inc in (True, False):
    m = CosyVoice3Model(sd[0], sd[1], sd[2], cfg, device=dev, max_batch=1, max_text=64, max_prompt_tokens=bench.P_TOK, max_tokens=n_max,
                        rand_noise=torch.from_numpy(synth.flow_rand_noise(2 * (bench.P_TOK + n_max))).to(dev),
                        rand_ini=torch.from_numpy(synth.hift_rand_ini()).to(dev),
                        sine_noise=torch.from_numpy(synth.hift_sine_noise(2 * n_max * 480)).to(dev), incremental_stream=inc)
    one = inputs[0]
    ref = [c["tts_speech"] for c in m.tts(**one, stream=True)]
    torch.cuda.synchronize()
    first, total = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        g = m.tts(**one, stream=True)
        c0 = next(g)["tts_speech"]
        t1 = time.perf_counter()
        rest = [c["tts_speech"] for c in g]
        t2 = time.perf_counter()
        first.append(t1 - t0); total.append(t2 - t0)
    audio = (c0.shape[1] + sum(r.shape[1] for r in rest)) / 24000.0
    print("incremental %-5s: first chunk %.1f ms, all %d chunks (%.2f s of audio) %.1f ms" % (inc, 1e3 * sorted(first)[1], 1 + len(rest), audio, 1e3 * sorted(total)[1]), flush=True)
    if inc: keep = [c0] + rest
    else: print("chunks identical to the incremental run:", all(torch.equal(a, b) for a, b in zip(keep, [c0] + rest)))
    m.close(); del m; torch.cuda.empty_cache()
