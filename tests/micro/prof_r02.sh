# Round 2 profiles.  Run on the GPU box from the repository root (through gpurun):  bash tests/micro/prof_r02.sh
# 1. rocprofv3 --kernel-trace --stats over the default bench command and over --no-pipeline (where the single LM stream uses
#    the persistent decode kernel) -> per-kernel summaries (copied to profiles/ afterwards);
# 2. HBM-side traffic of the LM decode kernels by PMC (FETCH_SIZE and WRITE_SIZE in separate passes) over
#    tests/micro/pmc_lm_probe.py.  bench.py itself cannot run under --pmc: with ONE HSA runtime in the process (the pre-load
#    below; DESIGN.md section 6 has the analysis) counters work for every kernel of this library and for small torch kernels,
#    but the profiler still dies in the launch path of torch's large elementwise kernels that fill the synthetic model on the GPU;
# 3. the DiT products' traffic and MFMA-busy counters (tests/micro/gemm_pmc.py over the stand-alone gemm_bench).
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
summ() {
python3 - "$1" "$2" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["name", "total_calls", "total_duration_us", "average_us", "percentage"])
for r in rows:
    w.writerow([r["Name"], r["Calls"], round(float(r["TotalDurationNs"]) / 1e3, 3), round(float(r["AverageNs"]) / 1e3, 3), r["Percentage"]])
PY
}
rocprofv3 --kernel-trace --stats -d /tmp/p_pipe -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/r02_bench_under_rocprof.json 2> /tmp/prof.err || (tail -5 /tmp/prof.err; exit 1)
summ "$(find /tmp/p_pipe -name '*kernel_stats.csv' | head -1)" $O/r02_bench_kernel_stats.csv
head -8 $O/r02_bench_kernel_stats.csv | cut -c1-140
echo "[prof] pipelined done"
rocprofv3 --kernel-trace --stats -d /tmp/p_nopipe -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-pipeline --steps 5 > $O/r02_bench_nopipeline_under_rocprof.json 2> /tmp/prof.err || (tail -5 /tmp/prof.err; exit 1)
summ "$(find /tmp/p_nopipe -name '*kernel_stats.csv' | head -1)" $O/r02_bench_nopipeline_kernel_stats.csv
head -8 $O/r02_bench_nopipeline_kernel_stats.csv | cut -c1-140
echo "[prof] unpipelined done"
export LD_LIBRARY_PATH=/opt/rocm/lib LD_PRELOAD="libamdhip64.so libhsa-runtime64.so"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_lm_$c -- python3 $R/tests/micro/pmc_lm_probe.py > /tmp/pmc_lm_$c.out 2> /tmp/pmc_lm_$c.err || (tail -5 /tmp/pmc_lm_$c.err; exit 1)
done
unset LD_PRELOAD LD_LIBRARY_PATH
python3 $R/tests/micro/pmc_aggregate.py /tmp/pmc_lm_FETCH_SIZE /tmp/pmc_lm_WRITE_SIZE $O/r02_llm_decode_pmc.json "tests/micro/pmc_lm_probe.py: batch 8, 6 tokens, persistent then per-operation decode"
echo "[prof] LM decode PMC done"
python3 $R/tests/micro/gemm_pmc.py
cp $O/gemm_pmc.json $O/r02_gemm_pmc.json
echo "[prof] GEMM PMC done"
# 4. the vocoder alone at BASELINE.json configs[4] (32 x 10 000 frames): per-kernel times, and its traffic on the stand-alone driver
bash $R/tests/micro/prof_hift.sh
python3 $R/tests/micro/hift_pmc.py
cp $O/hift_pmc.json $O/r02_hift_pmc.json
