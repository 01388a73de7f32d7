#!/bin/bash
# round5_gpu.sh TAG [ab names...] -- one gpurun call: the -m gpu suite (timed), then same-box A/B of library builds
# (build_ab/lib_NAME.so; "cur" = the tree's own) in the headline loop and on one stream.
cd $GRAFT_REPO_ROOT
TAG=${1:-r05a}; shift
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/${TAG}_pytest.log 2>&1
tail -8 gpurun_out/${TAG}_pytest.log
if [ $# -gt 0 ]; then
  bash tools/ab_bench.sh "$@" 2>&1 | tee gpurun_out/${TAG}_ab_bench.txt
  bash tools/ab_libs.sh "$@" 2>&1 | tee gpurun_out/${TAG}_ab_libs.txt
fi
