#!/bin/bash
# round6_c5_sweep.sh TAG -- C5 at its stated size, later calls of a process: the lockstep nest at 8 packets per wavefront and 16 workers (round 5's
# form) against the lean multi-packet parser at the pool's default, 16 / 32 / 48 workers (tools/c5_sweep.py; profiles/r06_cursor.txt section 2)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06c5}
export NVH_CORPUS_KEEP_CTX=1
( NVH_PARSE_CUR=0 timeout 600 python tools/c5_sweep.py --scale 1.0 --reps 3 --cases "16,8,0,0,0"
  timeout 900 python tools/c5_sweep.py --scale 1.0 --reps 3 --cases "16,0,0,0,0;32,0,0,0,0;48,0,0,0,0" ) 2>&1 | grep "^workers" | cut -c1-330 | tee gpurun_out/${TAG}_c5_sweep.txt
