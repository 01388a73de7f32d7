#!/bin/bash
# round6_parse.sh TAG -- one gpurun call: the GPU parser's window decode (k_parse_slab_u): parity tests under NVH_GPU_PARSE first, then
# the duration of a parse (tools/time_parse.py: 3test's packets and the C5 writer's), then the corpus pass.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06d}
NVH_GPU_PARSE=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parse.py -m gpu -x -q -p no:cacheprovider -k "not fallback" 2>&1 | tail -6
for f in 1024 4096 32768; do FRAMES=$f python tools/time_parse.py; FRAMES=$f CORPUS=1 python tools/time_parse.py; done 2>&1 | tee gpurun_out/${TAG}_time_parse.txt
timeout 900 python -m pytest tests/test_full_depth.py tests/test_multi_rank_gpu.py -m gpu -x -q -p no:cacheprovider -k "c5 or corpus or damaged or ranks" 2>&1 | tail -6
( NVH_CORPUS_KEEP_CTX=1 python tools/c5_sweep.py --scale 1.0 --reps 4 --cases "16,0,0,0,0;16,1,0,0,0"; NVH_CORPUS_FULL_INDEX=1 NVH_CORPUS_KEEP_CTX=1 python tools/c5_sweep.py --scale 1.0 --reps 3 --cases "16,0,0,0,0" ) > gpurun_out/${TAG}_c5_pass.txt 2>&1
grep -E "^workers" gpurun_out/${TAG}_c5_pass.txt | cut -c1-260 | head -30
