# Run on the GPU box from the repository root: the default bench line, then the same command under
# rocprofv3 --kernel-trace --stats, reduced to the per-kernel summary that is committed under profiles/.
set -e
R=$GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -c 600 gpurun_out/bench_final.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_under_rocprof.json 2> /tmp/prof.err || (tail -5 /tmp/prof.err; exit 1)
F=$(find /tmp/prof_final -name "*kernel_stats.csv" | head -1)
python3 - "$F" "$R/gpurun_out/bench_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["name", "total_calls", "total_duration_us", "average_us", "percentage"])
for r in rows:
    w.writerow([r["Name"], r["Calls"], round(float(r["TotalDurationNs"]) / 1e3, 3), round(float(r["AverageNs"]) / 1e3, 3), r["Percentage"]])
PY
head -12 $R/gpurun_out/bench_kernel_stats.csv | cut -c1-150
# the same with the steps one after the other (no LM streams beside the flow decoder): the kernels' stand-alone durations
rocprofv3 --kernel-trace --stats -d /tmp/prof_nopipe -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-pipeline --steps 5 > $R/gpurun_out/bench_nopipeline_under_rocprof.json 2> /tmp/prof.err || (tail -5 /tmp/prof.err; exit 1)
F=$(find /tmp/prof_nopipe -name "*kernel_stats.csv" | head -1)
python3 - "$F" "$R/gpurun_out/bench_nopipeline_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["name", "total_calls", "total_duration_us", "average_us", "percentage"])
for r in rows:
    w.writerow([r["Name"], r["Calls"], round(float(r["TotalDurationNs"]) / 1e3, 3), round(float(r["AverageNs"]) / 1e3, 3), r["Percentage"]])
PY
head -12 $R/gpurun_out/bench_nopipeline_kernel_stats.csv | cut -c1-150
