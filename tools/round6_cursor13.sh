#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06c13}
( NVH_PARSE_LANES=32 timeout 600 python -m pytest tests/test_gpu_parse.py -m gpu -q -x -p no:cacheprovider -k "not multi_packet" 2>&1 | tail -4 ) | tee gpurun_out/${TAG}_tests_gpu_parse.txt
( NVH_PARSE_LANES=16 NVH_GPU_PARSE=1 NVH_TEST_CHILD=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_depth.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4 ) | tee gpurun_out/${TAG}_tests_parity.txt
timeout 600 bash tools/parse_kernels.sh "FRAMES=32768" "FRAMES=3000 CORPUS=1 NVH_PARSE_LANES=32" "FRAMES=3000 NVH_PARSE_LANES=32" "FRAMES=3000 NVH_PARSE_LANES=16" 2>&1 | grep -v "k_parse_links\|result_out" | tee gpurun_out/${TAG}_kernels.txt
export NVH_CORPUS_KEEP_CTX=1
( timeout 900 python tools/c5_sweep.py --scale 1.0 --reps 3 --cases "16,0,0,0,0;32,0,0,0,0;32,16,0,0,0" ) 2>&1 | grep "^workers" | cut -c1-330 | tee gpurun_out/${TAG}_c5_sweep.txt
