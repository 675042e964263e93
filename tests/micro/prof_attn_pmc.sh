# Counters of the DiT attention kernel alone (gpurun: bash tests/micro/prof_attn_pmc.sh): the clock the chip holds in it
# (GRBM_GUI_ACTIVE / kernel duration), vector-ALU and MFMA busy fractions.  Counter passes carry --kernel-trace only.
set -e
R=${GRAFT_REPO_ROOT:-.}
cd $R
F="-O3 -std=c++17 --offload-arch=gfx950 -I fangyan_tts_amd/csrc -I include"
hipcc $F -fno-slp-vectorize -c fangyan_tts_amd/csrc/attn_dit.hip -o /tmp/attn_dit.o
for f in fangyan_tts_amd/csrc/attn.hip fangyan_tts_amd/csrc/gemv32.hip fangyan_tts_amd/csrc/runtime.hip tests/micro/attn_bench.hip; do
  hipcc $F -c $f -o /tmp/$(basename $f .hip).o 2>/dev/null
done
hipcc --offload-arch=gfx950 /tmp/attn_bench.o /tmp/attn.o /tmp/gemv32.o /tmp/runtime.o /tmp/attn_dit.o -o /tmp/attn_bench
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY"; do
  i=$((i+1)); rm -rf /tmp/pmc_attn_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_attn_$i -- /tmp/attn_bench 400 16 > /tmp/pmc_attn_$i.out 2> /tmp/pmc_attn_$i.err || { echo "pass $i ($set) failed"; tail -3 /tmp/pmc_attn_$i.err; continue; }
done
python3 - <<'PY'
import csv, glob, collections
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_attn_1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dit_attention2_k" in r["Kernel_Name"]: dur[1].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
acc = collections.defaultdict(lambda: [0.0, 0])
for i in (1, 2, 3, 4):
    for f in glob.glob(f"/tmp/pmc_attn_{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "dit_attention2_k" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
m = {k: v[0] / max(v[1], 1) for k, v in acc.items()}
d = sum(dur[1]) / max(len(dur[1]), 1)
print("dit_attention2_k, 16 x 400 x 16 heads, mean over", len(dur[1]), "launches under the counters: %.1f us per launch" % d)
for k in sorted(m): print("  %-28s %.4g per launch" % (k, m[k]))
if "GRBM_GUI_ACTIVE" in m and d: print("  clock while the kernel runs: GRBM_GUI_ACTIVE / duration = %.2f GHz" % (m["GRBM_GUI_ACTIVE"] / d / 1e3))
PY
