#!/bin/bash
# round3_bench.sh TAG -- bench line + rocprofv3 passes (no test suite)
cd $GRAFT_REPO_ROOT
TAG=${1:-r03a}
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
head -c 1500 gpurun_out/${TAG}_bench.json
bash tools/profile_round.sh $TAG "round 3 ($TAG): headline loop, one stream, working set past the Infinity Cache"
