#!/bin/bash
# k_parse_slab(_g) duration with pieces of the slab output left out (profiling build): which piece costs what
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-32768}
for M in ${2:-0 1 2 4 8 15}; do
  OUT=gpurun_out/prof_mask_$M; rm -rf $OUT; mkdir -p $OUT
  NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so NVH_DEBUG_SPECTRUM_MASK=$((15 + 256 * M)) FRAMES=$N timeout 120 rocprofv3 --kernel-trace --stats -d $OUT -- python tools/e2e_gpu_parse.py > $OUT/log.txt 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
for r in cur.execute("select name, count(*), avg(duration)/1e3 from kernels k where name like 'k_parse_slab%' and grid_x > 1000 group by name"):
    print("frames $N leave-out mask $M", r)
PY

  rm -rf $OUT
done
