#!/bin/bash
# round6_base.sh TAG -- one gpurun call: rocprofv3 summaries of the headline loop (three streams, one stream), the -m gpu suite, the bench line.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06a}
bash tools/profile_round.sh ${TAG}_3stream "round 6 ($TAG), headline loop (three streams)" "--streams 3" > gpurun_out/${TAG}_prof3.log 2>&1
bash tools/profile_round.sh ${TAG}_1stream "round 6 ($TAG), one stream" "--streams 1" > gpurun_out/${TAG}_prof1.log 2>&1
head -12 gpurun_out/prof_${TAG}_3stream/summary.txt; head -12 gpurun_out/prof_${TAG}_1stream/summary.txt
( time bash tools/round4_gpu.sh $TAG ) 2>&1 | tail -14
