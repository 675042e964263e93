# Clock and power of the GPU while bench.py runs (rocm-smi polled from a second shell process): what the chip holds under this load.
# Run on the GPU box from the repository root:  bash tests/micro/clock_watch.sh  -> gpurun_out/clock_watch.txt
O=gpurun_out/clock_watch.txt
: > $O
( for i in $(seq 1 400); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power" | tr '\n' ' ' >> $O; echo >> $O; sleep 0.1; done ) &
W=$!
python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 4 2>/dev/null | tail -1 | cut -c1-160
kill $W 2>/dev/null
wait $W 2>/dev/null
awk 'NF' $O | sed 's/  */ /g' | sort | uniq -c | sort -rn | head -12
