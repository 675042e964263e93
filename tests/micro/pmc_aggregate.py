"""Reduce rocprofv3 counter-collection CSVs (one --pmc pass each) to per-kernel HBM-side traffic per launch.

    python tests/micro/pmc_aggregate.py <fetch_dir> <write_dir> <out.json> [note]

FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes
of wide coalesced reads - the read side is doubled; WRITE_SIZE is exact for 16-byte streaming stores."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


FAMILIES = {"dit_linears": ("gemm256_k", "gemm64_k")}      # every launch bench.py's "gemm_bf16" events bracket


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("<")[0].strip()


def collect(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            acc[k][0] += float(r["Counter_Value"]) * 1024.0
            acc[k][1] += 1
    return acc


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    out = {"note": sys.argv[4] if len(sys.argv) > 4 else "", "unit": "bytes per launch (mean)", "read_correction": "FETCH_SIZE x 2 (gfx950)",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 1])[0] * 2 + write.get(k, [0, 1])[0])):
        fb, fn = fetch.get(k, [0.0, 0])
        wb, wn = write.get(k, [0.0, 0])
        n = max(fn, wn, 1)
        out["kernels"][k] = {"launches": n, "read_bytes": round(2 * fb / max(fn, 1)), "write_bytes": round(wb / max(wn, 1)),
                             "traffic_bytes": round(2 * fb / max(fn, 1) + wb / max(wn, 1))}
    # kernel families a roofline entry is quoted on: the launch-weighted mean over their members (bench.py reads families.dit_linears)
    out["families"] = {}
    for fam, members in FAMILIES.items():
        ks = [out["kernels"][m] for m in members if m in out["kernels"]]
        n = sum(k["launches"] for k in ks)
        if n:
            out["families"][fam] = {"members": [m for m in members if m in out["kernels"]], "launches": n,
                                    "read_bytes": round(sum(k["read_bytes"] * k["launches"] for k in ks) / n),
                                    "write_bytes": round(sum(k["write_bytes"] * k["launches"] for k in ks) / n),
                                    "traffic_bytes": round(sum(k["traffic_bytes"] * k["launches"] for k in ks) / n)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in list(out["kernels"].items())[:12]:
        print(f"{k[:50]:50s} {v['launches']:6d} launches  read {v['read_bytes'] / 1e6:10.2f} MB  write {v['write_bytes'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
