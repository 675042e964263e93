"""Does hipDeviceScheduleBlockingSync work under torch on this stack, and what does a waiting thread cost with and without it?
Round 5, MI355X box: spin 1.39 cores, blocking event 1.00, blocking-sync flag 1.00 AND the process hangs at exit - which is why
the library waits by polling hipStreamQuery with sleeps instead (fy_set_host_wait).  Run each leg under its own `timeout`: a leg that hangs must not take the call with it.
    python tests/micro/blocking_sync_probe.py spin|block|event
"""
import sys
import time

import torch

sys.path.insert(0, ".")
mode = sys.argv[1] if len(sys.argv) > 1 else "spin"
dev = torch.device("cuda:0")
x = torch.randn(8192, 8192, device=dev)
torch.cuda.synchronize()
if mode == "block":
    import ctypes
    import os
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))       # the copy torch has loaded
    print("hipSetDeviceFlags(hipDeviceScheduleBlockingSync) ->", hip.hipSetDeviceFlags(4), flush=True)


def busy(n=40):
    y = x
    for _ in range(n):
        y = y @ x
        y = y / y.abs().max()
    return y


busy(2)
torch.cuda.synchronize()
print("warm", flush=True)
c0, t0 = time.process_time(), time.perf_counter()
for _ in range(5):
    busy()
    if mode == "event":
        ev = torch.cuda.Event(blocking=True)
        ev.record()
        ev.synchronize()
    else:
        torch.cuda.current_stream().synchronize()
c1, t1 = time.process_time(), time.perf_counter()
print(f"{mode}: wall {t1 - t0:.3f} s, process CPU {c1 - c0:.3f} s = {(c1 - c0) / (t1 - t0):.2f} cores", flush=True)
z = busy(2).sum().cpu()          # a blocking copy
print("copy ok", float(z) == float(z), flush=True)
