"""Where the DiT products' waves spend their cycles: rocprofv3 --pmc over `tests/micro/gemm_bench pmc` (3 launches per shape, automatic
kernel choice), the last launch of the block's four shapes.  Passes (counters that fit one pass each):
  SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY   wave-parked (s_waitcnt / barrier) vs issue-stalled vs issuing
  SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS           LDS-array cycles, conflict cycles, LDS issue stalls
  TCC_HIT_sum TCC_MISS_sum                                          L2 hit rate
Run from the repository root on the GPU box:  python3 tests/micro/gemm_wait_pmc.py  -> gpurun_out/gemm_wait_pmc.json"""
import csv, glob, json, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BENCH = os.path.join(ROOT, "tests", "micro", "gemm_bench")
PASSES = [["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"], ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS"],
          ["TCC_HIT_sum", "TCC_MISS_sum"], ["SQ_BUSY_CU_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"]]
NAMES = ["qkv", "out", "ff1", "ff2"]
res = {n: {} for n in NAMES}
for i, c in enumerate(PASSES):
    d = f"/tmp/pmc_gw_{i}"
    r = subprocess.run(["rocprofv3", "--pmc", *c, "--kernel-trace", "-d", d, "--output-format", "csv", "--", BENCH, "pmc"], cwd="/tmp",
                       env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    if r.returncode:
        print("pass", c, "failed:", r.stderr.decode()[-300:]); continue
    rows = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for x in csv.DictReader(open(f)):
            if "gemm" not in x["Kernel_Name"]: continue
            k = int(x["Dispatch_Id"])
            rows.setdefault(k, {"kernel": x["Kernel_Name"][:60]})
            rows[k][x["Counter_Name"]] = rows[k].get(x["Counter_Name"], 0.0) + float(x["Counter_Value"])
    seq = [rows[k] for k in sorted(rows)]
    for j, n in enumerate(NAMES):
        res[n].update(seq[3 * j + 2])
for n, v in res.items():
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    v["frac_parked"] = round(v.get("SQ_WAIT_ANY", 0) / wc, 3); v["frac_issue_stalled"] = round(v.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
    v["frac_issuing"] = round(v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
    if v.get("SQ_LDS_IDX_ACTIVE"): v["lds_conflict_share"] = round(v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"], 3)
    if v.get("SQ_BUSY_CU_CYCLES"): v["lds_array_busy"] = round(v.get("SQ_LDS_IDX_ACTIVE", 0) / v["SQ_BUSY_CU_CYCLES"], 3); v["mfma_busy"] = round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / v["SQ_BUSY_CU_CYCLES"] / 4, 3)
    if v.get("TCC_HIT_sum") is not None: v["l2_hit"] = round(v.get("TCC_HIT_sum", 0) / max(v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0), 1), 3)
    print(n, {k: v[k] for k in v if k.startswith(("frac", "lds", "mfma", "l2", "kernel"))})
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "gemm_wait_pmc.json"), "w"), indent=1)
