"""Reduce a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES pass to the MFMA-busy fraction per kernel instance.

    python tests/micro/pmc_mfma_aggregate.py <pmc_dir> <out.json> [note]

SQ_VALU_MFMA_BUSY_CYCLES is summed over the 4 SIMDs of a CU, SQ_BUSY_CU_CYCLES counts CU cycles: busy fraction = ratio / 4
(MI355X_MICROARCH.md, SQ counters; the same reduction as tests/micro/gemm_pmc.py)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:80]


def main():
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES":
                cnt[k] += 1
    out = {"note": sys.argv[3] if len(sys.argv) > 3 else "", "kernels": {}}
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CU_CYCLES", 0.0)):
        b = acc[k]
        if b.get("SQ_BUSY_CU_CYCLES", 0.0) <= 0:
            continue
        out["kernels"][k] = {"launches": cnt[k], "mfma_busy_fraction": round(b.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / b["SQ_BUSY_CU_CYCLES"] / 4, 4)}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    for k, v in list(out["kernels"].items())[:14]:
        print(f"{k[:70]:70s} {v['launches']:7d} launches  MFMA busy {100 * v['mfma_busy_fraction']:5.1f} %")


if __name__ == "__main__":
    main()
