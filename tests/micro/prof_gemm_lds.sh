# LDS-side counters of the DiT products (gpurun: bash tests/micro/prof_gemm_lds.sh): is the sixteen-wave 256x256 kernel's K loop at the
# LDS's pace?  gemm_bench with an argument launches every shape a few times with the automatic tiling; counter passes carry --kernel-trace only.
set -e
R=${GRAFT_REPO_ROOT:-.}
cd $R
F="-O3 -std=c++17 --offload-arch=gfx950 -I fangyan_tts_amd/csrc -I include"
for f in fangyan_tts_amd/csrc/gemm.hip fangyan_tts_amd/csrc/runtime.hip tests/micro/gemm_bench.hip; do
  hipcc $F -c $f -o /tmp/$(basename $f .hip).o 2>/dev/null
done
hipcc --offload-arch=gfx950 /tmp/gemm_bench.o /tmp/gemm.o /tmp/runtime.o -o /tmp/gemm_bench
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_INSTS_VALU_MFMA[A-Z_0-9]*\|SQ_ACTIVE_INST_[A-Z]*" | sort -u | tr '\n' ' ' | head -c 1500; echo
i=0
for set in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1)); rm -rf /tmp/pmc_g_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_g_$i -- /tmp/gemm_bench pmc > /tmp/pmc_g_$i.out 2> /tmp/pmc_g_$i.err || { echo "pass $i ($set) failed: $(grep -i "error\|invalid\|not" /tmp/pmc_g_$i.err | head -2)"; continue; }
done
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("/tmp/pmc_g_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if "gemm" not in k: continue
        key = (k, r["Grid_Size"])
        a = acc[key][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for key in sorted(acc):
    m = {c: v[0] / max(v[1], 1) for c, v in acc[key].items()}
    print(key[0][:44], "grid", key[1])
    for c in sorted(m): print("    %-28s %.4g" % (c, m[c]))
    if "SQ_BUSY_CU_CYCLES" in m and m["SQ_BUSY_CU_CYCLES"]:
        b = m["SQ_BUSY_CU_CYCLES"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m: print("    MFMA busy %.1f %%" % (100 * m["SQ_VALU_MFMA_BUSY_CYCLES"] / 4 / b))
        if "SQ_LDS_IDX_ACTIVE" in m: print("    LDS index active / CU busy %.1f %%" % (100 * m["SQ_LDS_IDX_ACTIVE"] / b))
        if "SQ_LDS_BANK_CONFLICT" in m: print("    LDS bank-conflict cycles / CU busy %.1f %%" % (100 * m["SQ_LDS_BANK_CONFLICT"] / b))
PY
