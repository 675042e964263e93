"""Does `rocprofv3 --pmc` survive a torch process on this image?  (round 1: SIGSEGV in the first torch kernel launch.)
The torch wheel bundles its own libamdhip64.so / libhsa-runtime64.so (ROCm 7.0) and asks for them by FILE name; the rocprofv3
tool library has /opt/rocm's libhsa-runtime64.so.1 (7.2) loaded already, whose SONAME does not match that file name, so the
process ends up with TWO HSA runtimes: the tool programs counters through one, torch's queues belong to the other.
Run as:  LD_LIBRARY_PATH=/opt/rocm/lib LD_PRELOAD="libamdhip64.so libhsa-runtime64.so" rocprofv3 --pmc FETCH_SIZE ... -- python3 this.py
(the bare-name preload makes the loader satisfy torch's request with the copy that is already there: one runtime)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
maps = open("/proc/self/maps").read()
libs = sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l or "libhsa-runtime64" in l})
print("runtimes mapped:", libs, flush=True)
x = torch.arange(1 << 20, device="cuda") ^ 5
y = (x.float() * 2).sum().item()
print("torch kernels ran:", y, flush=True)
from fangyan_tts_amd import synth
from fangyan_tts_amd.hift import HiftEngine
from fangyan_tts_amd.spec import HiftCfg
cfg = HiftCfg.tiny()
eng = HiftEngine(synth.state_dict_torch(cfg.manifest(), torch.device("cuda:0")), cfg, max_batch=1, max_frames=40)
mel = torch.rand(1, 80, 40, device="cuda")
wav, _ = eng.inference(mel, torch.from_numpy(synth.hift_rand_ini()).cuda(), torch.from_numpy(synth.hift_sine_noise(40 * 480)).cuda())
torch.cuda.synchronize()
print("library kernels ran:", float(wav.abs().mean()), flush=True)
