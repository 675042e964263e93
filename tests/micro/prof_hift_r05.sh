set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_hift
rocprofv3 --kernel-trace --stats -d /tmp/p_hift -o hift --output-format csv -- python3 $R/bench_hift.py --steps 3 > $R/gpurun_out/r05_hift_under_rocprof.json 2> /tmp/prof.err || (tail -5 /tmp/prof.err; exit 1)
python3 - "$(find /tmp/p_hift -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/r05_hift_config5_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["name", "total_calls", "total_duration_us", "average_us", "percentage"])
for r in rows:
    w.writerow([r["Name"], r["Calls"], round(float(r["TotalDurationNs"]) / 1e3, 3), round(float(r["AverageNs"]) / 1e3, 3), r["Percentage"]])
for r in rows[:16]:
    print(r["Name"][:80].ljust(80), r["Calls"].rjust(5), ("%.1f" % (float(r["TotalDurationNs"]) / 1e3)).rjust(11), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), r["Percentage"][:5])
PY
