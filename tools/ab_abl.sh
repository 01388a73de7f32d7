#!/bin/bash
# ab_abl.sh NAME... -- ablation builds (build_ab/lib_NAME.so; "cur" = the tree's own) in the headline loop (three streams) and on
# one stream; --no-check: an ablation build's PCM is wrong by construction
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for n in "$@"; do
    if [ "$n" = cur ]; then L=""; else L="NVH_ALLOW_STALE=1 NVH_LIB=$GRAFT_REPO_ROOT/build_ab/lib_$n.so"; fi
    echo -n "$n: "
    env $L python bench.py --no-check --no-configs --no-cpu-baseline --no-unfused --c5-scale 0 --steps 100 --min-timed-ms 700 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('3 streams %.2f us per pass (%.1f M frames/s), L3-resident %.1f M, one stream kernels %s' % (4096e6/d['value'], d['value']/1e6, d['roofline']['l3_resident']['frames_per_s']/1e6, {k: round(v*1e3,2) for k,v in d['kernels_ms'].items()}))"
  done
done
