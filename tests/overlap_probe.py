"""Not a test: probes how much the LM decode (latency-bound, few CUs) overlaps with flow + HiFT of another batch
on a second stream."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from fangyan_tts_amd import synth
from fangyan_tts_amd.cli.model import CosyVoice3Model
from fangyan_tts_amd.spec import ModelCfg

dev = torch.device("cuda:0"); cfg = ModelCfg()
sd = [synth.state_dict_torch(m.manifest(), dev, skip=("lm_head",)) for m in (cfg.llm, cfg.flow, cfg.hift)]
T = 2 * (bench.P_TOK + bench.N_TOK)
noise = torch.from_numpy(synth.flow_rand_noise(T)).to(dev); ri = torch.from_numpy(synth.hift_rand_ini()).to(dev)
sn = torch.from_numpy(synth.hift_sine_noise(2 * bench.N_TOK * 480)).to(dev)
m = CosyVoice3Model(sd[0], sd[1], sd[2], cfg, device=dev, max_batch=8, max_text=64, max_prompt_tokens=bench.P_TOK, max_tokens=bench.N_TOK, rand_noise=noise, rand_ini=ri, sine_noise=sn)
inputs = bench.make_inputs(cfg, 0)
text = [d["text"].reshape(-1).tolist() for d in inputs]; ptext = [d["prompt_text"].reshape(-1).tolist() for d in inputs]
ptok = torch.cat([d["flow_prompt_speech_token"] for d in inputs]).to(torch.int32).to(dev); pfeat = torch.cat([d["prompt_speech_feat"] for d in inputs]).to(dev)
emb = torch.cat([d["flow_embedding"] for d in inputs]).to(dev)
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(exclude):
    """a HIP stream whose kernels may not use the first `exclude` CUs (bit i of the mask = CU i enabled)"""
    words = (ctypes.c_uint32 * 8)(*([0xFFFFFFFF] * 8))
    for i in range(exclude):
        words[i // 32] &= ~(1 << (i % 32))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)
EXCL = int(os.environ.get("EXCL", "0"))
s_llm = torch.cuda.Stream(device=dev, priority=-1); s_fh = masked_stream(EXCL) if EXCL else torch.cuda.Stream(device=dev)
print("flow stream excluded CUs:", EXCL, flush=True)

def run_llm():
    with torch.cuda.stream(s_llm):
        out, out_n, _ = m.llm.generate(text, ptext, [[] for _ in inputs], min_len=[75]*8, max_len=[75]*8)
        s_llm.synchronize()
    return out

def run_fh(out):
    with torch.cuda.stream(s_fh):
        mel = m.flow.inference(out, [75]*8, ptok, [125]*8, pfeat, [250]*8, emb, noise)
        wav, _ = m.hift.inference(mel, ri, sn, frames=[150]*8)
        s_fh.synchronize()
    return wav

out = run_llm(); run_fh(out)
for rep in range(2):
    t0 = time.perf_counter(); out = run_llm(); t1 = time.perf_counter(); run_fh(out); t2 = time.perf_counter()
    print(f"sequential: llm {1e3*(t1-t0):.1f} ms, flow+hift {1e3*(t2-t1):.1f} ms, total {1e3*(t2-t0):.1f}", flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    res = {}
    ta = threading.Thread(target=lambda: res.__setitem__("o", run_llm())); tb = threading.Thread(target=lambda: res.__setitem__("w", run_fh(out)))
    tdone = {}
    ta = threading.Thread(target=lambda: (res.__setitem__("o", run_llm()), tdone.__setitem__("l", time.perf_counter())))
    tb = threading.Thread(target=lambda: (res.__setitem__("w", run_fh(out)), tdone.__setitem__("f", time.perf_counter())))
    ta.start(); tb.start(); ta.join(); tb.join()
    print(f"overlapped: llm done at {1e3*(tdone['l']-t0):.1f} ms, flow+hift done at {1e3*(tdone['f']-t0):.1f} ms", flush=True)
