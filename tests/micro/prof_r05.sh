# Round 5 profiles.  Run on the GPU box from the repository root (through gpurun):  bash tests/micro/prof_r05.sh [stage ...]
#   stats   rocprofv3 --kernel-trace --stats over the DEFAULT bench command and over --no-pipeline -> per-kernel summaries
#   pmc     HBM-side traffic by PMC over bench.py ITSELF (FETCH_SIZE and WRITE_SIZE in separate passes, one HSA runtime in the
#           process: DESIGN.md section 6); the synthetic weights are filled by the library's own kernel (fy_synth_uniform), so the
#           process launches none of the torch int64 elementwise kernels the profiler died under in rounds 1-2
#   mfma    MFMA-busy fraction per kernel (SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES / 4) over bench.py itself
#   lmpmc   the same counters over tests/micro/pmc_lm_probe.py (LM decode alone: persistent at 8 rows, per-operation at 32 rows)
# Summaries land in gpurun_out/ (copy the ones to keep into profiles/).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
STAGES="${@:-stats pmc lmpmc}"
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
for s in $STAGES; do
case $s in
stats)
  rm -rf /tmp/p_pipe /tmp/p_nopipe
  rocprofv3 --kernel-trace --stats -d /tmp/p_pipe -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/r05_bench_under_rocprof.json 2> /tmp/prof.err || { tail -5 /tmp/prof.err; exit 1; }
  summ "$(find /tmp/p_pipe -name '*kernel_stats.csv' | head -1)" $O/r05_bench_kernel_stats.csv
  head -8 $O/r05_bench_kernel_stats.csv | cut -c1-150
  echo "[prof] default (pipelined) command done"
  rocprofv3 --kernel-trace --stats -d /tmp/p_nopipe -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-pipeline --steps 8 > $O/r05_bench_nopipeline_under_rocprof.json 2> /tmp/prof.err || { tail -5 /tmp/prof.err; exit 1; }
  summ "$(find /tmp/p_nopipe -name '*kernel_stats.csv' | head -1)" $O/r05_bench_nopipeline_kernel_stats.csv
  head -8 $O/r05_bench_nopipeline_kernel_stats.csv | cut -c1-150
  echo "[prof] unpipelined done"
  ;;
pmc)
  export LD_LIBRARY_PATH=/opt/rocm/lib LD_PRELOAD="libamdhip64.so libhsa-runtime64.so"
  ok=1
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_bench_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_bench_$c -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 4 > $O/r05_bench_under_pmc_$c.json 2> $O/r05_pmc_bench_$c.err || { echo "[prof] bench.py under --pmc $c FAILED"; tail -12 $O/r05_pmc_bench_$c.err; ok=0; break; }
  done
  unset LD_PRELOAD LD_LIBRARY_PATH
  if [ $ok = 1 ]; then
    python3 $R/tests/micro/pmc_aggregate.py /tmp/pmc_bench_FETCH_SIZE /tmp/pmc_bench_WRITE_SIZE $O/r05_bench_pmc.json "rocprofv3 --pmc over bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 4 itself (the default pipelined configuration), FETCH_SIZE and WRITE_SIZE in separate passes"
    echo "[prof] bench.py PMC done"
  fi
  ;;
mfma)
  export LD_LIBRARY_PATH=/opt/rocm/lib LD_PRELOAD="libamdhip64.so libhsa-runtime64.so"
  rm -rf /tmp/pmc_bench_mfma
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_bench_mfma -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 4 > $O/r05_bench_under_pmc_mfma.json 2> $O/r05_pmc_bench_mfma.err || { echo "[prof] bench.py under --pmc (MFMA busy) FAILED"; tail -12 $O/r05_pmc_bench_mfma.err; }
  unset LD_PRELOAD LD_LIBRARY_PATH
  python3 $R/tests/micro/pmc_mfma_aggregate.py /tmp/pmc_bench_mfma $O/r05_bench_mfma_busy.json "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES over bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 4 (the default pipelined configuration)"
  echo "[prof] MFMA busy done"
  ;;
lmpmc)
  export LD_LIBRARY_PATH=/opt/rocm/lib LD_PRELOAD="libamdhip64.so libhsa-runtime64.so"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_lm_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_lm_$c -- python3 $R/tests/micro/pmc_lm_probe.py > /tmp/pmc_lm_$c.out 2> /tmp/pmc_lm_$c.err || { tail -5 /tmp/pmc_lm_$c.err; exit 1; }
  done
  unset LD_PRELOAD LD_LIBRARY_PATH
  python3 $R/tests/micro/pmc_aggregate.py /tmp/pmc_lm_FETCH_SIZE /tmp/pmc_lm_WRITE_SIZE $O/r05_llm_decode_pmc.json "tests/micro/pmc_lm_probe.py: 6 tokens; persistent decode at batch 8, per-operation decode (gemv32_k) at batch 32"
  echo "[prof] LM decode PMC done"
  ;;
esac
done
