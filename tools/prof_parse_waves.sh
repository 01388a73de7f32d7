#!/bin/bash
# k_parse duration against wavefronts per workgroup (NVH_PARSE_WAVES) for several batch sizes: rocprofv3 kernel trace of tools/e2e_gpu_parse.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for N in ${1:-1024 4096 32768}; do
for W in ${2:-4 8}; do
  OUT=gpurun_out/prof_waves_$W; rm -rf $OUT; mkdir -p $OUT
  NVH_PARSE_WAVES=$W FRAMES=$N timeout 200 rocprofv3 --kernel-trace --stats -d $OUT -- python tools/e2e_gpu_parse.py > $OUT/log.txt 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
for r in cur.execute("select name, count(*), avg(duration)/1e3, max(grid_x), max(workgroup_x), max(lds_size) from kernels k where name like 'k_parse%' and name not like '%links%' and name not like '%fetch%' and name not like '%result%' group by name"):
    print("frames $N waves $W", r)
PY
  rm -rf $OUT
done; done
