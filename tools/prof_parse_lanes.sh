#!/bin/bash
# k_parse duration against packets per wavefront (NVH_PARSE_LANES) for one batch size: rocprofv3 kernel trace of tools/e2e_gpu_parse.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
N=${1:-32768}
for L in ${2:-4 8 16 32}; do
  OUT=gpurun_out/prof_lanes_$L; rm -rf $OUT; mkdir -p $OUT
  NVH_PARSE_LANES=$L FRAMES=$N rocprofv3 --kernel-trace --stats -d $OUT -- python tools/e2e_gpu_parse.py > $OUT/log.txt 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
for r in cur.execute("select name, count(*), avg(duration)/1e3, max(grid_x), max(lds_size) from kernels k where name like 'k_parse%' and grid_x > 1000 group by name"):
    print("frames $N lanes $L", r)
PY
  rm -rf $OUT
done
