#!/bin/bash
# round3_gpu.sh TAG -- one gpurun call: the GPU test suite (timed), the bench line, the rocprofv3 passes of the headline loop.
cd $GRAFT_REPO_ROOT
TAG=${1:-r03a}
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > gpurun_out/${TAG}_pytest.log 2>&1
tail -5 gpurun_out/${TAG}_pytest.log
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 3000 gpurun_out/${TAG}_bench.json
bash tools/profile_round.sh $TAG "round 3 ($TAG): headline loop, one stream, working set past the Infinity Cache"
# the same loop with the overlap-add left to k_ola_compact (no paired emission): k_synth and k_ola_compact on their own
NVH_NO_EMIT=1 bash tools/profile_round.sh ${TAG}_unfused "round 3 ($TAG, NVH_NO_EMIT=1): headline loop without paired emission, one stream"
