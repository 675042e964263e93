"""Write the synthetic HiFT weights as a flat binary for tests/micro/hift_bench (no torch on the GPU needed)."""
import os, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from fangyan_tts_amd import synth
from fangyan_tts_amd.spec import HiftCfg

out = sys.argv[1]
sd = synth.state_dict(HiftCfg().manifest())
with open(out, "wb") as f:
    f.write(struct.pack("<i", len(sd)))
    for k, v in sd.items():
        v = np.ascontiguousarray(v, dtype=np.float32)
        nb = k.encode()
        f.write(struct.pack("<i", len(nb))); f.write(nb)
        f.write(struct.pack("<i", v.ndim)); f.write(struct.pack("<" + "q" * v.ndim, *v.shape))
        f.write(v.tobytes())
print("wrote", out)
