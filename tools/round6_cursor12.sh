#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06c12}
bash tools/parse_kernels.sh "FRAMES=32768" "FRAMES=3000 CORPUS=1 NVH_PARSE_LANES=32" 2>&1 | grep -v "k_parse_links\|result_out" | tee gpurun_out/${TAG}_kernels.txt
export NVH_CORPUS_KEEP_CTX=1
( python tools/c5_sweep.py --scale 1.0 --reps 3 --cases "16,0,0,0,0;32,0,0,0,0;48,0,0,0,0" ) 2>&1 | grep "^workers" | cut -c1-330 | tee gpurun_out/${TAG}_c5_sweep.txt
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 | tee gpurun_out/${TAG}_gpu_tests.txt
