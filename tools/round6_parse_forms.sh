#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06c6}; shift
( NVH_PARSE_LANES=${LANES:-8} timeout 900 python -m pytest tests/test_gpu_parse.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 ) | tee gpurun_out/${TAG}_tests_gpu_parse.txt
( NVH_PARSE_LANES=${LANES:-8} NVH_GPU_PARSE=1 NVH_TEST_CHILD=1 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_full_depth.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -8 ) | tee gpurun_out/${TAG}_tests_parity.txt
bash tools/parse_kernels.sh "$@" 2>&1 | grep -v "k_parse_links\|result_out" | tee gpurun_out/${TAG}_kernels.txt
