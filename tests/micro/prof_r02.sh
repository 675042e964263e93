# Round 2 profiles.  Run on the GPU box from the repository root (through gpurun):  bash tests/micro/prof_r02.sh
# 1. rocprofv3 --kernel-trace --stats over the default bench command and over --no-pipeline (where the single LM stream uses
#    the persistent decode kernel) -> per-kernel summaries (copied to profiles/ afterwards);
# 2. HBM-side traffic by PMC, FETCH_SIZE and WRITE_SIZE in separate passes, over the SAME bench.py process.  Round 1 could not
#    do that (SIGSEGV in the first torch launch): the torch wheel brings its own libamdhip64.so / libhsa-runtime64.so and asks
#    for them by FILE name, the rocprofv3 tool has /opt/rocm's libhsa-runtime64.so.1 loaded already, whose SONAME does not match
#    that file name, and the process ended up with TWO HSA runtimes (gpurun_out/pmc_probe_plain.log lists both mapped).
#    Pre-loading /opt/rocm's copies under the bare file names makes the loader satisfy torch's request with the copy that is
#    already there: one runtime, counters work (tests/micro/pmc_torch_probe.py).
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
rocprofv3 --kernel-trace --stats -d /tmp/p_nopipe -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-pipeline --steps 5 > $O/r02_bench_nopipeline_under_rocprof.json 2> /tmp/prof.err || (tail -5 /tmp/prof.err; exit 1)
summ "$(find /tmp/p_nopipe -name '*kernel_stats.csv' | head -1)" $O/r02_bench_nopipeline_kernel_stats.csv
head -8 $O/r02_bench_nopipeline_kernel_stats.csv | cut -c1-140
export LD_LIBRARY_PATH=/opt/rocm/lib LD_PRELOAD="libamdhip64.so libhsa-runtime64.so"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_nopipe_$c -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-pipeline --steps 1 --warmup 1 > /tmp/pmc_$c.json 2> /tmp/pmc_$c.err || (tail -5 /tmp/pmc_$c.err; exit 1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_pipe_$c -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /tmp/pmcp_$c.json 2> /tmp/pmcp_$c.err || (tail -5 /tmp/pmcp_$c.err; exit 1)
done
unset LD_PRELOAD LD_LIBRARY_PATH
python3 $R/tests/micro/pmc_aggregate.py /tmp/pmc_nopipe_FETCH_SIZE /tmp/pmc_nopipe_WRITE_SIZE $O/r02_bench_nopipeline_pmc.json "bench.py --no-pipeline --steps 1 --warmup 1 (persistent LM decode)"
python3 $R/tests/micro/pmc_aggregate.py /tmp/pmc_pipe_FETCH_SIZE /tmp/pmc_pipe_WRITE_SIZE $O/r02_bench_pmc.json "bench.py --steps 2 --warmup 1 (default, pipelined)"
