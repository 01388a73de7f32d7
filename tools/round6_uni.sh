#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06u}
( timeout 600 python -m pytest tests/test_gpu_parse.py -m gpu -q -x -p no:cacheprovider -k "not multi_packet" 2>&1 | tail -4 ) | tee gpurun_out/${TAG}_tests_gpu_parse.txt
( NVH_GPU_PARSE=1 NVH_TEST_CHILD=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_depth.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4 ) | tee gpurun_out/${TAG}_tests_parity.txt
for f in 1024 4096; do FRAMES=$f timeout 200 python tools/time_parse.py; done 2>&1 | tee gpurun_out/${TAG}_time_parse.txt
FRAMES=4096 CORPUS=1 timeout 300 python tools/time_parse.py 2>&1 | tee -a gpurun_out/${TAG}_time_parse.txt
timeout 300 bash tools/parse_kernels.sh "FRAMES=4096" 2>&1 | grep -v "k_parse_links\|result_out" | tee gpurun_out/${TAG}_kernels.txt
