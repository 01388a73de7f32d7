#!/bin/bash
# round6_profiles.sh TAG -- the rocprofv3 summaries of the headline loop for the build that is committed (three streams: the timed loop,
# with the PMC passes behind traffic.json; one stream), copied to gpurun_out/ for profiles/
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06}
bash tools/profile_round.sh ${TAG}_3stream "round 6 closing build, headline loop (three streams)" "--streams 3" > gpurun_out/${TAG}_prof3.log 2>&1
head -5 gpurun_out/prof_${TAG}_3stream/summary.txt
bash tools/profile_round.sh ${TAG}_1stream "round 6 closing build, one stream" "--streams 1" > gpurun_out/${TAG}_prof1.log 2>&1
head -5 gpurun_out/prof_${TAG}_1stream/summary.txt
