import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from fangyan_tts_amd import synth
from fangyan_tts_amd.cli.model import CosyVoice3Model
from fangyan_tts_amd.spec import ModelCfg
dev = torch.device("cuda:0")
cfg = ModelCfg()
sd = [synth.state_dict_torch(m.manifest(), dev, skip=("lm_head",)) for m in (cfg.llm, cfg.flow, cfg.hift)]
T = 2 * (bench.P_TOK + bench.N_TOK)
m = CosyVoice3Model(sd[0], sd[1], sd[2], cfg, device=dev, max_batch=8, max_text=64, max_prompt_tokens=bench.P_TOK, max_tokens=bench.N_TOK,
                    rand_noise=torch.from_numpy(synth.flow_rand_noise(T)).to(dev), rand_ini=torch.from_numpy(synth.hift_rand_ini()).to(dev),
                    sine_noise=torch.from_numpy(synth.hift_sine_noise(2 * bench.N_TOK * 480)).to(dev))
inp = bench.make_inputs(cfg, 0)
B = len(inp)
z = torch.zeros(1, 0, dtype=torch.int32)
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    text = [d["text"].reshape(-1).tolist() for d in inp]; ptext = [d.get("prompt_text", z).reshape(-1).tolist() for d in inp]
    t0b = time.perf_counter()
    out, out_n, _ = m.llm.generate(text, ptext, [[] for _ in inp], min_len=[75] * B, max_len=[75] * B)
    t1a = time.perf_counter()
    n_tok = out_n.cpu().tolist()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    wav, samples = m._token2wav(inp, out, n_tok, 1.0)
    t2a = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    w = wav.cpu(); t3 = time.perf_counter()
    wav2, s2, _ = m.tts_batch(inp, min_len=[75] * B, max_len=[75] * B); t4 = time.perf_counter()
    print(f"rep {rep}: lists {1e3*(t0b-t0):.2f} | LM call returns {1e3*(t1a-t0b):.1f}, ids on host {1e3*(t1-t0b):.1f} | token2wav returns {1e3*(t2a-t1):.1f}, done {1e3*(t2-t1):.1f} | wav D2H {1e3*(t3-t2):.2f} | tts_batch {1e3*(t4-t3):.1f} ms", flush=True)
