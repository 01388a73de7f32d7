#!/bin/bash
# round6_abl.sh -- where the joint walk's time is: the headline loop (three streams and one, --no-check: wrong PCM by construction) with
# the floor multiply / the cascade stages left out of residue_walk_two
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
brief='import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); print("%.1f M frames/s, %.2f us per pass, kernels %s" % (d["value"]/1e6, d["config"]["ms_per_pass"]*1e3, {k: round(v*1e3,2) for k,v in d["kernels_ms"].items()}))
except Exception as e:
    print("FAILED", e, t[-3:])'
for r in 1 2; do
  for n in cur nofloor2 nochain2; do
    if [ $n = cur ]; then L=""; else L="NVH_ALLOW_STALE=1 NVH_LIB=$GRAFT_REPO_ROOT/build_ab/lib_$n.so"; fi
    for st in 3 1; do
      echo -n "$n streams=$st: "
      env $L timeout 300 python bench.py --no-check --no-configs --no-cpu-baseline --no-unfused --c5-scale 0 --steps 100 --min-timed-ms 1200 --streams $st 2>gpurun_out/abl_err.log | python -c "$brief"
    done
  done
done
