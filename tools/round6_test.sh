#!/bin/bash
# round6_test.sh TAG -- one gpurun call: the C3 / file-level parity tests first (fast failure), the whole -m gpu suite, the bench line.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06b}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_full_depth.py -m gpu -x -q -p no:cacheprovider -k "ogg_files or c3 or markov or partial_reads or resident_batches" 2>&1 | tail -8
( time bash tools/round4_gpu.sh $TAG ) 2>&1 | tail -40
