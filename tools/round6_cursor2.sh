#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06c2}
( NVH_PARSE_LANES=8 timeout 900 python -m pytest tests/test_gpu_parse.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8 ) | tee gpurun_out/${TAG}_tests_gpu_parse.txt
( NVH_PARSE_LANES=8 NVH_GPU_PARSE=1 NVH_TEST_CHILD=1 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_full_depth.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -8 ) | tee gpurun_out/${TAG}_tests_parity.txt
bash tools/parse_kernels.sh "FRAMES=3000 CORPUS=1 NVH_PARSE_LANES=8 NVH_PARSE_CUR=0" "FRAMES=3000 CORPUS=1 NVH_PARSE_LANES=8 NVH_PARSE_CUR=2" "FRAMES=3000 CORPUS=1 NVH_PARSE_LANES=32 NVH_PARSE_CUR=2" "FRAMES=3000 CORPUS=1 NVH_PARSE_LANES=32 NVH_PARSE_CUR=1" "FRAMES=3000 NVH_PARSE_LANES=8 NVH_PARSE_CUR=0" "FRAMES=3000 NVH_PARSE_LANES=8 NVH_PARSE_CUR=2" "FRAMES=3000 NVH_PARSE_LANES=32 NVH_PARSE_CUR=2" "FRAMES=32768 NVH_PARSE_LANES=8 NVH_PARSE_CUR=0" "FRAMES=32768 NVH_PARSE_LANES=32 NVH_PARSE_CUR=2" "FRAMES=32768 NVH_PARSE_LANES=32 NVH_PARSE_CUR=2 NVH_PARSE_WAVES=16" 2>&1 | tee gpurun_out/${TAG}_kernels.txt
