# Does the measured stream placement (fy_stream_overlap) find a clash-free set whatever the queue count / process history?
for q in 4 8 6; do
  echo "GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q FY_PIPE_TRACE=1 python bench.py --no-cpu-baseline 2>&1 | grep "timed\|streams (" | cut -c1-220
done
