#!/bin/bash
# the cursor form of the multi-lane GPU parser (NVH_PARSE_CUR): parity, then timing against the lockstep nest
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06c}
( NVH_PARSE_LANES=8 NVH_GPU_PARSE=1 timeout 900 python -m pytest tests/test_gpu_parse.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 ) | tee gpurun_out/${TAG}_tests_split.txt
( NVH_PARSE_LANES=8 NVH_PARSE_CUR=1 NVH_GPU_PARSE=1 timeout 900 python -m pytest tests/test_gpu_parse.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 ) | tee gpurun_out/${TAG}_tests_one.txt
for f in 3000 32768; do
  for cur in 0 2; do
    for l in 8 16 32 64; do NVH_PARSE_CUR=$cur NVH_PARSE_LANES=$l FRAMES=$f CORPUS=1 timeout 300 python tools/time_parse.py | sed "s/^/CUR $cur /"; done
  done
  for l in 8 32; do NVH_PARSE_CUR=2 NVH_PARSE_LANES=$l FRAMES=$f timeout 300 python tools/time_parse.py | sed "s/^/CUR 2 /"; done
  NVH_PARSE_CUR=0 NVH_PARSE_LANES=8 FRAMES=$f timeout 300 python tools/time_parse.py | sed "s/^/CUR 0 /"
done 2>&1 | tee gpurun_out/${TAG}_time_parse.txt
