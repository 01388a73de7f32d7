#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06c9}
export NVH_CORPUS_KEEP_CTX=1
( NVH_PARSE_CUR=0 python tools/c5_sweep.py --scale 1.0 --reps 3 --cases "16,8,0,0,0"
  python tools/c5_sweep.py --scale 1.0 --reps 3 --cases "16,8,0,0,0;16,16,0,0,0;16,32,0,0,0;16,64,0,0,0;32,32,0,0,0;32,64,0,0,0;48,64,0,0,0" ) 2>&1 | grep -v "^host cpus\|corpus ready" | cut -c1-330 | tee gpurun_out/${TAG}_c5_sweep.txt
