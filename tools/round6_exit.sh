#!/bin/bash
# does bench.py's process end?  (SIGABRT after the limit: faulthandler prints every thread's stack)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for w in 32 16; do
  echo "## --c5-workers $w"
  ( time timeout -s ABRT ${LIMIT:-150} python -X faulthandler bench.py --no-cpu-baseline --no-configs --no-unfused --steps 5 --warmup 2 --min-timed-ms 200 --c5-scale ${SCALE:-0.1} --c5-workers $w > gpurun_out/exit_$w.json 2> gpurun_out/exit_$w.err ; echo "rc $?" ) 2>&1 | grep "rc \|real"
  tail -c 3000 gpurun_out/exit_$w.err | grep -v "^$" | tail -40
done
