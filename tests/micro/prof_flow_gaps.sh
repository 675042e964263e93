# Gaps between consecutive kernels of the flow decoder + vocoder in un-pipelined steps, by (kernel before, kernel after)
# (gpurun: bash tests/micro/prof_flow_gaps.sh)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_fgaps
rocprofv3 --kernel-trace -d /tmp/p_fgaps -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-pipeline --steps 4 --warmup 1 > $R/gpurun_out/r3_fgaps_bench.json 2> /tmp/prof.err || (tail -5 /tmp/prof.err; exit 1)
python3 - "$(find /tmp/p_fgaps -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
v = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]) for r in rows)
# the flow decoder's launches: from a dit_assemble_k to the last istft16_k that follows
idx = [i for i, x in enumerate(v) if x[2].startswith("flow_setup_k")]
print(len(v), "dispatches,", len(idx), "flow passes")
cnt = collections.defaultdict(lambda: [0, 0.0])
tot_gap = tot_busy = 0.0
n = 0
for s in idx[2:]:
    e = s
    while e + 1 < len(v) and not v[e][2].startswith("istft16_k"):
        e += 1
    w = v[s:e + 1]
    for i in range(len(w) - 1):
        g = max(0, w[i + 1][0] - w[i][1])
        cnt[(w[i][2], w[i + 1][2])][0] += 1
        cnt[(w[i][2], w[i + 1][2])][1] += g
        tot_gap += g
    tot_busy += sum(b - a for a, b, _ in w)
    n += 1
    print(f"pass: {len(w)} launches, span {(w[-1][1] - w[0][0]) / 1e6:.2f} ms, in kernels {sum(b - a for a, b, _ in w) / 1e6:.2f} ms")
print(f"per pass: gaps {tot_gap / n / 1e6:.2f} ms, kernels {tot_busy / n / 1e6:.2f} ms")
for k, (c, t) in sorted(cnt.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"  {c // n:5d} per pass x {t / c / 1e3:6.2f} us = {t / n / 1e6:5.2f} ms   after {k[0]:44s} before {k[1]}")
PY
