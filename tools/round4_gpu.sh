#!/bin/bash
# round4_gpu.sh TAG [pytest args] -- one gpurun call: the GPU test suite (timed) and the bench line.
cd $GRAFT_REPO_ROOT
TAG=${1:-r04a}
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -x -q -p no:cacheprovider $2 ) > gpurun_out/${TAG}_pytest.log 2>&1
tail -15 gpurun_out/${TAG}_pytest.log
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 1500 gpurun_out/${TAG}_bench.err
python tools/bench_brief.py $TAG < gpurun_out/${TAG}_bench.json; tail -c 2500 gpurun_out/${TAG}_bench.json
