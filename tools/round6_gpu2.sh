#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
brief='import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); print("%.1f M frames/s, digest_ok %s, kernels %s" % (d["value"]/1e6, d.get("pcm_digest_ok"), {k: round(v*1e3,2) for k,v in d["kernels_ms"].items()}))
except Exception as e:
    print("FAILED", e, t[-3:])'
run() { # label, env...
  for st in 3 1; do
    echo -n "$1 streams=$st: "
    env "${@:2}" timeout 300 python bench.py --no-configs --no-cpu-baseline --no-unfused --c5-scale 0 --steps 100 --min-timed-ms 1000 --streams $st 2>gpurun_out/r06b_err.log | python -c "$brief"
  done
}
NVH_DEBUG_OCC=1 NVH_FPW=2 python bench.py --no-configs --no-cpu-baseline --no-unfused --c5-scale 0 --steps 5 --min-timed-ms 50 --streams 1 2>&1 >/dev/null | grep "frame groups" | sort | uniq -c
NVH_DEBUG_OCC=1 NVH_FPW=2 NVH_GROUP_WIDE=1 python bench.py --no-configs --no-cpu-baseline --no-unfused --c5-scale 0 --steps 5 --min-timed-ms 50 --streams 1 2>&1 >/dev/null | grep "frame groups" | sort | uniq -c
for r in 1 2; do
  run fpw1 NVH_FPW=1
  run fpw2 NVH_FPW=2
  run fpw2wide NVH_FPW=2 NVH_GROUP_WIDE=1
  run fpw2nopf NVH_FPW=2 NVH_NO_PREFETCH=1
done
