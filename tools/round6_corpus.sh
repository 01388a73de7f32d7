#!/bin/bash
# round6_corpus.sh TAG -- one gpurun call: the corpus pass with the lacing-only index (tests, then timings against the round-5 index).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06c}
timeout 900 python -m pytest tests/test_full_depth.py tests/test_multi_rank_gpu.py -m gpu -x -q -p no:cacheprovider -k "c5 or corpus or damaged or ranks" 2>&1 | tail -6
( NVH_CORPUS_KEEP_CTX=1 python tools/c5_sweep.py --scale 1.0 --reps 4 --cases "16,0,0,0,0"; NVH_CORPUS_FULL_INDEX=1 NVH_CORPUS_KEEP_CTX=1 python tools/c5_sweep.py --scale 1.0 --reps 3 --cases "16,0,0,0,0" ) > gpurun_out/${TAG}_c5_pass.txt 2>&1
grep -E "^workers|index|decode" gpurun_out/${TAG}_c5_pass.txt | cut -c1-260 | head -30
