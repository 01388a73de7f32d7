# phase_times.sh -- marginal cost of the phases of k_spectrum_imdct: kernel time (HIP events, bench.py) with phases masked out
# (profiling build: python -m nvorbis_amd.build --debug; NVH_DEBUG_SPECTRUM_MASK bit0 residue walk, bit1 floor multiply (+ everything
# after it), bit2 floor unwrap, bit3 inverse MDCT).  The masked kernels write garbage; only their duration is of interest.
cd $GRAFT_REPO_ROOT
for m in 15 7 5 6 3 14 11 13 4 1 0; do
  NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so NVH_DEBUG_SPECTRUM_MASK=$m python bench.py --no-cpu-baseline --no-check --steps 60 --warmup 10 --streams 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mask %2d' % $m, {k: round(v*1000,2) for k,v in d['kernels_ms'].items()})"
done
