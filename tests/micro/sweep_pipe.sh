# pipelined-step sweep over LM handle count / steps per LM call / CUs kept clear of the flow stream
# (gpurun: bash tests/micro/sweep_pipe.sh "<streams>:<group>:<exclude> ..." [steps])
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
STEPS=${2:-20}
for cfg in $1; do
  IFS=: read s g x w <<< "$cfg"; w=${w:-1}
  f=$O/r3_sweep_s${s}_g${g}_x${x}.json
  timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --no-extras --llm-streams $s --lm-group $g --flow-cu-exclude $x --flow-workers $w $EXTRA --steps $STEPS --warmup 4 > $f 2> $O/r3_sweep.err || { tail -5 $O/r3_sweep.err; exit 1; }
  python3 - $f $cfg <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "ms_per_step", d["ms_per_step"], "value", d["value"], "unpipelined", d["config"]["batch_latency_ms_unpipelined"], flush=True)
PY
done
