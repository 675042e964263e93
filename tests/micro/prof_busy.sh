# How busy is the GPU in the pipelined bench?  Union of kernel intervals over the steady window, per queue and overall,
# and the mean number of kernels in flight (gpurun: bash tests/micro/prof_busy.sh [bench flags])
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_busy
rocprofv3 --kernel-trace -d /tmp/p_busy -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 24 --warmup 4 "$@" > $R/gpurun_out/r3_busy_bench.json 2> /tmp/prof.err || (tail -5 /tmp/prof.err; exit 1)
python3 - "$(find /tmp/p_busy -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("void ", "")[:30]) for r in rows)
# the timed pass: the longest stretch of gemm256_k launches; take the window between 35 % and 60 % of all dispatches (inside the first timed run)
a, b = ev[int(0.30 * len(ev))][0], ev[int(0.45 * len(ev))][0]
w = [e for e in ev if e[0] >= a and e[1] <= b]
span = b - a
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
print(f"window {span / 1e6:.1f} ms, {len(w)} dispatches; any kernel running {100 * union([(s, e) for s, e, _, _ in w]) / span:.1f} %; sum of kernel time / span = {sum(e - s for s, e, _, _ in w) / span:.2f}")
byq = collections.defaultdict(list)
for s, e, q, n in w: byq[q].append((s, e, n))
for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    names = collections.Counter(n for _, _, n in v).most_common(2)
    print(f"  queue {q}: {len(v)} dispatches, busy {100 * union([(s, e) for s, e, _ in v]) / span:.1f} %, kernel time {sum(e - s for s, e, _ in v) / 1e6:.1f} ms; {names}")
    if len(v) < 1000:
        continue
    v.sort()
    gaps = sorted(max(0, v[i + 1][0] - v[i][1]) for i in range(len(v) - 1))
    big = [g for g in gaps if g > 100000]
    print(f"     gaps: median {gaps[len(gaps) // 2] / 1e3:.1f} us, mean {sum(gaps) / len(gaps) / 1e3:.1f}, p90 {gaps[int(0.9 * len(gaps))] / 1e3:.1f}, p99 {gaps[int(0.99 * len(gaps))] / 1e3:.1f} us; "
          f"{len(big)} gaps > 100 us totalling {sum(big) / 1e6:.1f} ms; all gaps {sum(gaps) / 1e6:.1f} ms")
    cnt = collections.defaultdict(lambda: [0, 0])
    for i in range(len(v) - 1):
        g = max(0, v[i + 1][0] - v[i][1])
        cnt[(v[i][2], v[i + 1][2])][0] += 1; cnt[(v[i][2], v[i + 1][2])][1] += g
    for k, (n, t) in sorted(cnt.items(), key=lambda kv: -kv[1][1])[:6]:
        print(f"       {n:6d} x {t / n / 1e3:7.1f} us = {t / 1e6:6.1f} ms   after {k[0]:30s} before {k[1]}")
PY
