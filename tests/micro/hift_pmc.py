"""Collect the HiFT vocoder's L2-miss traffic and MFMA-busy counters on the GPU box and write gpurun_out/hift_pmc.json
(copied to profiles/ by hand).  Run from the repository root:   python3 tests/micro/hift_pmc.py

rocprofv3 --pmc cannot sit under a torch process on this pool, so the engine is driven stand-alone
through the C ABI (tests/micro/hift_bench.cpp, built here with g++) on the synthetic weights written by
tests/micro/dump_hift_weights.py; three separate passes; 8 x 5000 mel frames, default (bf16) mode; the LAST inference of the
run is reported.  Units and the gfx950 read correction follow /opt/skills/guides/MI355X_MICROARCH.md."""
import csv
import glob
import json
import os
import re
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out")
BENCH = "/tmp/hift_bench"
WEIGHTS = "/tmp/hift_weights.bin"
B, F = 8, 5000
PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"]]
ALG_BYTES_PER_FRAME = 4156000.0


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.strip()


def run_pass(i, counters):
    d = f"/tmp/pmc_hift_{i}"
    subprocess.run(["rm", "-rf", d])
    env = dict(os.environ, TMPDIR="/tmp", LD_LIBRARY_PATH=os.path.join(ROOT, "fangyan_tts_amd", "lib") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    subprocess.run(["rocprofv3", "--pmc", *counters, "--kernel-trace", "-d", d, "--output-format", "csv", "--", BENCH, WEIGHTS, str(B), str(F), "1"],
                   check=True, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = int(r["Dispatch_Id"])
            rows.setdefault(k, {"kernel": short(r["Kernel_Name"])})
            rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [rows[k] for k in sorted(rows)]


def main():
    os.makedirs(OUT, exist_ok=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "micro", "dump_hift_weights.py"), WEIGHTS], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["g++", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "micro", "hift_bench.cpp"), "-o", BENCH,
                    "-L" + os.path.join(ROOT, "fangyan_tts_amd", "lib"), "-lfy_cosy3", "-L/opt/rocm/lib", "-lamdhip64"], check=True)
    per = [run_pass(i, c) for i, c in enumerate(PASSES)]
    n = len(per[0])
    assert all(len(p) == n for p in per), [len(p) for p in per]
    # two inferences per run (one warm-up): the second half of the dispatches after the set-up kernels is the reported one.
    # The inference starts at its tr_bcl_blc_k launch: take the dispatches from the LAST one on.
    start = max(i for i, r in enumerate(per[0]) if r["kernel"].startswith("tr_bcl_blc_k"))
    agg = defaultdict(lambda: {"launches": 0, "FETCH_SIZE_KiB": 0.0, "WRITE_SIZE_KiB": 0.0, "mfma": 0.0, "cu": 0.0})
    for i in range(start, n):
        a = agg[per[0][i]["kernel"]]
        a["launches"] += 1
        a["FETCH_SIZE_KiB"] += per[0][i]["FETCH_SIZE"]
        a["WRITE_SIZE_KiB"] += per[1][i]["WRITE_SIZE"]
        a["mfma"] += per[2][i]["SQ_VALU_MFMA_BUSY_CYCLES"]
        a["cu"] += per[2][i]["SQ_BUSY_CU_CYCLES"]
    kernels, total = {}, 0
    for k, a in agg.items():
        t = int((2 * a["FETCH_SIZE_KiB"] + a["WRITE_SIZE_KiB"]) * 1024)
        total += t
        kernels[k] = {"launches": a["launches"], "FETCH_SIZE_KiB": round(a["FETCH_SIZE_KiB"], 1), "WRITE_SIZE_KiB": round(a["WRITE_SIZE_KiB"], 1),
                      "traffic_bytes": t, "mfma_busy_fraction": round(a["mfma"] / max(a["cu"], 1.0) / 4, 3)}
    frames = B * F
    doc = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES (separate passes, "
                     f"tests/micro/hift_pmc.py) on `hift_bench weights.bin {B} {F} 1` (the C ABI driven stand-alone: HiFT, {B} x {F} mel frames, "
                     "default bf16 mode), MI355X; the last inference of the run",
           "units": "FETCH_SIZE/WRITE_SIZE in KiB as reported; traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction of "
                    "MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of wide coalesced reads); L2-miss traffic incl. Infinity-Cache hits, an "
                    "upper bound of HBM traffic; SQ_VALU_MFMA_BUSY_CYCLES sums the 4 SIMDs of a CU: mfma/cu/4 = fraction of MFMA issue capacity "
                    "while the CU is busy",
           "frames": frames, "algorithmic_bytes_per_frame_fp32_io": ALG_BYTES_PER_FRAME, "total_traffic_bytes": total,
           "traffic_bytes_per_frame": total / frames, "traffic_over_algorithmic": round(total / frames / ALG_BYTES_PER_FRAME, 3), "kernels": kernels}
    json.dump(doc, open(os.path.join(OUT, "hift_pmc.json"), "w"), indent=1)
    print(f"traffic {total / frames / 1e6:.3f} MB per frame = {doc['traffic_over_algorithmic']} x the fp32-I/O algorithmic bytes")
    for k, v in kernels.items():
        print(f"{k[:60]:60s} {v['launches']:3d} launches  {v['traffic_bytes'] / 1e9:8.2f} GB  mfma busy {v['mfma_busy_fraction']}")


if __name__ == "__main__":
    sys.exit(main())
