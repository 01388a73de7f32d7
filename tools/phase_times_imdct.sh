cd $GRAFT_REPO_ROOT
for m in 15 31 47 63; do
  NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so NVH_DEBUG_SPECTRUM_MASK=$m python bench.py --no-cpu-baseline --no-check --steps 60 --warmup 10 --streams 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mask %2d' % $m, {k: round(v*1000,2) for k,v in d['kernels_ms'].items()})"
done
