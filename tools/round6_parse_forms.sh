#!/bin/bash
# round6_parse_forms.sh TAG "ENV=.. ENV=.." ... -- parity of the multi-packet parser under a forced launch shape (LANES, default 8: the parser
# tests, then the parity suite + the full-depth configs GPU-parsed), then the kernel durations of tools/time_parse.py's child under each environment
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06c6}; shift
( NVH_PARSE_LANES=${LANES:-8} timeout 900 python -m pytest tests/test_gpu_parse.py -m gpu -q -x -p no:cacheprovider -k "not multi_packet" 2>&1 | tail -8 ) | tee gpurun_out/${TAG}_tests_gpu_parse.txt
( NVH_PARSE_LANES=${LANES:-8} NVH_GPU_PARSE=1 NVH_TEST_CHILD=1 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_full_depth.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -8 ) | tee gpurun_out/${TAG}_tests_parity.txt
timeout 900 bash tools/parse_kernels.sh "$@" 2>&1 | grep -v "k_parse_links\|result_out" | tee gpurun_out/${TAG}_kernels.txt
