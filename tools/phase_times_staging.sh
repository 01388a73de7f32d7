# phase_times_staging.sh -- what the staging phase of k_spectrum_imdct is made of (profiling build; NVH_DEBUG_SPECTRUM_MASK bits
# 64: return at once, 128: return after the frame + mapping records, 256: no pair records / chain-head list; 0: staging only)
cd $GRAFT_REPO_ROOT
for m in 64 128 256 0 4; do
  NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so NVH_DEBUG_SPECTRUM_MASK=$m python bench.py --no-cpu-baseline --no-check --steps 60 --warmup 10 --streams 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mask %3d' % $m, {k: round(v*1000,2) for k,v in d['kernels_ms'].items()})"
done
