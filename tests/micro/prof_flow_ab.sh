# flow decoder kernels of an unpipelined step with and without a switch (gpurun: bash tests/micro/prof_flow_ab.sh FY_FLOW_LN_FOLD)
# prints, per setting, the kernels of the flow decoder with calls, total and average time, and the sum over them
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
V=${1:-FY_FLOW_LN_FOLD}
cd /tmp && export TMPDIR=/tmp
for val in 1 0; do
  rm -rf /tmp/p_ab
  export $V=$val
  rocprofv3 --kernel-trace --stats -d /tmp/p_ab -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-pipeline --steps 3 --warmup 1 > /tmp/ab.json 2> /tmp/prof.err || (tail -5 /tmp/prof.err; exit 1)
  python3 - "$(find /tmp/p_ab -name '*kernel_stats.csv' | head -1)" "$V=$val" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keys = ("gemm256_k", "gemm64_k", "gemm_bf16_k", "dit_attention_k", "ln_mod_k", "ln_rowstats_k", "euler_k", "dit_assemble_k", "conv1d_bf16_mfma_k")
tot = 0.0
print("---", sys.argv[2])
for r in rows:
    if any(k in r["Name"] for k in keys):
        t = float(r["TotalDurationNs"]) / 1e3
        tot += t
        print("  %-86s %6s calls %10.1f us total %8.2f us avg" % (r["Name"][:86], r["Calls"], t, float(r["AverageNs"]) / 1e3))
print("  sum over these kernels: %.1f us (4 estimator-steps x 10 x 22 blocks in it)" % tot)
PY
done
