#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06c7}; shift
for c in "" 1; do
echo "## NVH_PARSE_LANES=32, 3000 packets CORPUS=$c"
CORPUS=$c NVH_PARSE_LANES=32 NVH_ALLOW_STALE=1 NVH_LIB=$GRAFT_REPO_ROOT/nvorbis_amd/libnvorbis_hip_dbg.so python tools/dbg_phase_parse.py 3000 2>&1 | grep "left to"
done | tee gpurun_out/${TAG}_bail.txt
bash tools/parse_kernels.sh "$@" 2>&1 | grep -v "k_parse_links\|result_out" | tee gpurun_out/${TAG}_kernels.txt
