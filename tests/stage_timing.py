"""Not a test: prints per-stage wall times of one batch on the GPU (run by hand through gpurun)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fangyan_tts_amd import synth, _lib
from fangyan_tts_amd.cli.model import CosyVoice3Model
from fangyan_tts_amd.spec import ModelCfg

def t(msg, t0):
    torch.cuda.synchronize(); print(f"{msg}: {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)

dev = torch.device("cuda:0"); cfg = ModelCfg()
t0 = time.perf_counter()
sd = [synth.state_dict_torch(m.manifest(), dev, skip=("lm_head",)) for m in (cfg.llm, cfg.flow, cfg.hift)]
t("weights", t0)
T = 2 * (bench.P_TOK + bench.N_TOK)
noise = torch.from_numpy(synth.flow_rand_noise(T)).to(dev); ri = torch.from_numpy(synth.hift_rand_ini()).to(dev)
sn = torch.from_numpy(synth.hift_sine_noise(2 * bench.N_TOK * 480)).to(dev)
t0 = time.perf_counter()
m = CosyVoice3Model(sd[0], sd[1], sd[2], cfg, device=dev, max_batch=8, max_text=64, max_prompt_tokens=bench.P_TOK, max_tokens=bench.N_TOK, rand_noise=noise, rand_ini=ri, sine_noise=sn)
t("engines", t0)
inputs = bench.make_inputs(cfg, 0)
text = [d["text"].reshape(-1).tolist() for d in inputs]; ptext = [d["prompt_text"].reshape(-1).tolist() for d in inputs]
for rep in range(3):
    t0 = time.perf_counter()
    out, out_n, _ = m.llm.generate(text, ptext, [[] for _ in inputs], min_len=[75]*8, max_len=[75]*8)
    t(f"llm rep{rep}", t0)
    n_tok = out_n.cpu().tolist()
    ptok = torch.cat([d["flow_prompt_speech_token"] for d in inputs]).to(torch.int32); pfeat = torch.cat([d["prompt_speech_feat"] for d in inputs])
    emb = torch.cat([d["flow_embedding"] for d in inputs])
    t0 = time.perf_counter()
    mel = m.flow.inference(out, n_tok, ptok, [125]*8, pfeat, [250]*8, emb, noise)
    t(f"flow rep{rep}", t0)
    t0 = time.perf_counter()
    wav, _ = m.hift.inference(mel, ri, sn, frames=[150]*8)
    t(f"hift rep{rep}", t0)
L = _lib.lib(); L.fy_prof_reset(); L.fy_prof_enable(1)
m.tts_batch(inputs, min_len=[75]*8, max_len=[75]*8, keep_on_device=True); torch.cuda.synchronize(); L.fy_prof_enable(0)
for k in ("gemm_bf16", "conv_mfma", "gemv"):
    ms, w, n = _lib.prof_get(k); print(k, f"{ms:.2f} ms over {n} launches, work {w:.3e}, rate {w/ms/1e9 if ms else 0:.1f} G/s")
