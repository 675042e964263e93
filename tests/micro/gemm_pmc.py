"""Collect the L2-miss traffic and MFMA-busy counters of the DiT products on the GPU box and write
gpurun_out/gemm_pmc.json (copied to profiles/ by hand).  Run from the repository root:

    python3 tests/micro/gemm_pmc.py

Three separate rocprofv3 --pmc passes over `tests/micro/gemm_bench pmc` (3 launches per shape, the library's automatic
kernel choice); the last launch of every shape is reported.  Units and the gfx950 read correction follow
/opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE / WRITE_SIZE in KiB; wide coalesced reads are reported at half size).
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BENCH = os.path.join(ROOT, "tests", "micro", "gemm_bench")
OUT = os.path.join(ROOT, "gpurun_out")
PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"]]
# the shapes gemm_bench runs, in its order: (name, M, N, K, epilogue, algorithmic bytes)
def _alg(M, N, K, mode):
    a = 2 * M * K + 2 * N * K
    return a + {0: 2 * M * N, 1: 8 * M * N, 2: 4 * M * N}[mode]
SHAPES = [("qkv", 6400, 3072, 1024, 0), ("out", 6400, 1024, 1024, 1), ("ff1", 6400, 2048, 1024, 0), ("ff2", 6400, 1024, 2048, 1),
          ("out-shape bf16", 6400, 1024, 1024, 0), ("M=1280", 1280, 1024, 1024, 0), ("M=12800", 12800, 1024, 1024, 0),
          ("M=12800 qkv", 12800, 3072, 1024, 0), ("M=12800 ff1", 12800, 2048, 1024, 0), ("M=3200 qkv", 3200, 3072, 1024, 0),
          ("M=3200 out", 3200, 1024, 1024, 1), ("M=3200 ff2", 3200, 1024, 2048, 1), ("in-proj", 6400, 1024, 320, 2)]
EPI = {0: "bf16 out", 1: "gated fp32 residual", 2: "fp32 out"}


def run_pass(i, counters):
    d = f"/tmp/pmc_gemm_{i}"
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--pmc", *counters, "--kernel-trace", "-d", d, "--output-format", "csv", "--", BENCH, "pmc"],
                   check=True, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm" not in r["Kernel_Name"]:
                continue
            k = int(r["Dispatch_Id"])
            rows.setdefault(k, {"kernel": r["Kernel_Name"][:48]})
            rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [rows[k] for k in sorted(rows)]


def main():
    os.makedirs(OUT, exist_ok=True)
    per = [run_pass(i, c) for i, c in enumerate(PASSES)]
    n = len(SHAPES)
    for p in per:
        assert len(p) == 3 * n, f"expected {3 * n} gemm dispatches, saw {len(p)}"
    res = {}
    for i, (name, M, N, K, mode) in enumerate(SHAPES):
        f, w, b = per[0][3 * i + 2], per[1][3 * i + 2], per[2][3 * i + 2]
        res[name] = {"kernel": f["kernel"], "M": M, "N": N, "K": K, "epilogue": EPI[mode],
                     "FETCH_SIZE_KiB": f["FETCH_SIZE"], "WRITE_SIZE_KiB": w["WRITE_SIZE"],
                     "traffic_bytes": int((2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024), "algorithmic_bytes": _alg(M, N, K, mode),
                     "mfma_busy_fraction": round(b["SQ_VALU_MFMA_BUSY_CYCLES"] / max(b["SQ_BUSY_CU_CYCLES"], 1.0) / 4, 3)}
    doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES (separate passes, "
                     "tests/micro/gemm_pmc.py) on `tests/micro/gemm_bench pmc` (the DiT shapes of the benchmark: M = 2*8*400 rows, the "
                     "library's automatic kernel choice), MI355X; last of 3 launches per shape",
           "units": "FETCH_SIZE/WRITE_SIZE in KiB as reported; traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 - the gfx950 correction of "
                    "MI355X_MICROARCH.md (FETCH_SIZE reports half the bytes of wide coalesced reads); Infinity-Cache hits are counted, so this "
                    "is L2-miss traffic, an upper bound of HBM traffic; SQ_VALU_MFMA_BUSY_CYCLES is summed over the 4 SIMDs of a CU while "
                    "SQ_BUSY_CU_CYCLES counts CU cycles: mfma_busy_fraction = their ratio / 4",
           # a DiT block launches qkv, out, ff1 and ff2 once each: the per-launch mean over those four is what bench.py reports
           "dit_mix_traffic_bytes_per_launch": int(sum(res[k]["traffic_bytes"] for k in ("qkv", "out", "ff1", "ff2")) / 4),
           "dit_mix_algorithmic_bytes_per_launch": int(sum(res[k]["algorithmic_bytes"] for k in ("qkv", "out", "ff1", "ff2")) / 4),
           "shapes": res}
    json.dump(doc, open(os.path.join(OUT, "gemm_pmc.json"), "w"), indent=1)
    for k, v in res.items():
        print(f"{k:16s} {v['kernel'][:30]:30s} traffic {v['traffic_bytes'] / 1e6:8.1f} MB  alg {v['algorithmic_bytes'] / 1e6:7.1f} MB  mfma busy {v['mfma_busy_fraction']}")


if __name__ == "__main__":
    sys.exit(main())
