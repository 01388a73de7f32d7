#!/bin/bash
# round6_profiles.sh TAG STREAMS -- the rocprofv3 summary of the headline loop for the build that is committed (STREAMS = 3: the timed
# loop, with the PMC passes behind traffic.json; 1: one decoder instance)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06}; ST=${2:-3}
( time bash tools/profile_round.sh ${TAG}_${ST}stream "round 6 closing build, headline loop (${ST} stream(s))" "--streams $ST" ) > gpurun_out/${TAG}_prof${ST}.log 2>&1
head -12 gpurun_out/prof_${TAG}_${ST}stream/summary.txt; tail -4 gpurun_out/${TAG}_prof${ST}.log
