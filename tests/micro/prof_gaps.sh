# Per hardware queue of the pipelined bench: time inside kernels against the span they cover (gpurun: bash tests/micro/prof_gaps.sh).
# CAUTION: under the profiler the streams land on other hardware queues than in a plain run (two LM streams were seen sharing one),
# so long holes in this view are not evidence about the plain run - HIP events on the flow stream (FY_PIPE_TRACE) are.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_gaps
rocprofv3 --kernel-trace -d /tmp/p_gaps -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 12 --warmup 2 > $R/gpurun_out/r2_gaps_bench.json 2> /tmp/prof.err || (tail -5 /tmp/prof.err; exit 1)
python3 - "$(find /tmp/p_gaps -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "dispatches; columns:", list(rows[0].keys())[:14])
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get("Queue_Id", "?")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    v.sort()
    if len(v) < 2000:
        continue
    # the steady part: drop the first 40 % (set-up, warm-up) and the last 25 % (the LM chains end before the flow stream)
    a, b = int(0.45 * len(v)), int(0.75 * len(v))
    w = v[a:b]
    busy = sum(e - s for s, e, _ in w)
    span = w[-1][1] - w[0][0]
    gaps = sorted(max(0, w[i + 1][0] - w[i][1]) for i in range(len(w) - 1))
    names = collections.Counter(n.split("(")[0][:28] for _, _, n in w).most_common(3)
    print(f"queue {q}: {len(v)} launches; steady window {len(w)} launches over {span / 1e6:.1f} ms: in kernels {busy / 1e6:.1f} ms ({100 * busy / span:.0f} %), "
          f"gap median {gaps[len(gaps) // 2] / 1e3:.1f} us, mean {sum(gaps) / len(gaps) / 1e3:.1f} us, p90 {gaps[int(0.9 * len(gaps))] / 1e3:.1f} us; mean kernel {busy / len(w) / 1e3:.1f} us; {names}")
    big = [(w[i + 1][0] - w[i][1], w[i][2].split("(")[0][-40:], w[i + 1][2].split("(")[0][-40:]) for i in range(len(w) - 1) if w[i + 1][0] - w[i][1] > 50000]
    print(f"   gaps > 50 us: {len(big)}, {sum(g for g, _, _ in big) / 1e6:.1f} ms in all; by the kernels around them:")
    cnt = collections.defaultdict(lambda: [0, 0])
    for g, a_, b_ in big:
        cnt[(a_, b_)][0] += 1; cnt[(a_, b_)][1] += g
    for k, (n, t) in sorted(cnt.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"      {n:5d} x  {t / n / 1e3:8.1f} us   after {k[0]}  before {k[1]}")
PY
