# quick per-kernel summary of an unpipelined bench under rocprofv3 (gpurun: bash tests/micro/prof_quick.sh [pattern])
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_nopipe
rocprofv3 --kernel-trace --stats -d /tmp/p_nopipe -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-pipeline --steps 3 --warmup 1 > $O/r2_quick_nopipe.json 2> /tmp/prof.err || (tail -5 /tmp/prof.err; exit 1)
python3 - "$(find /tmp/p_nopipe -name '*kernel_stats.csv' | head -1)" "${1:-}" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
for r in (rows[:14] if not pat else [r for r in rows if pat in r["Name"]]):
    print(r["Name"][:110], r["Calls"], round(float(r["TotalDurationNs"]) / 1e3, 1), round(float(r["AverageNs"]) / 1e3, 2), r["Percentage"])
PY
