#!/bin/bash
# ab_bench.sh NAME... -- same-box A/B of library builds build_ab/lib_NAME.so ("cur" = the tree's own library) in the headline
# loop of bench.py (3 streams, HBM-resident, ~1 s sustained per run), interleaved, 3 rounds
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for n in "$@"; do
    if [ "$n" = cur ]; then L=""; else L="NVH_ALLOW_STALE=1 NVH_LIB=$GRAFT_REPO_ROOT/build_ab/lib_$n.so"; fi
    echo -n "$n: "
    env $L python bench.py --no-configs --no-cpu-baseline --no-unfused --c5-scale 0 --steps 100 --min-timed-ms 1000 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f M frames/s HBM-resident, %.1f M L3-resident, kernels %s' % (d['value']/1e6, d['roofline']['l3_resident']['frames_per_s']/1e6, {k: round(v*1e3,2) for k,v in d['kernels_ms'].items()}))"
  done
done
