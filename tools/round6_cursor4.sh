#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06c4}
for l in 8 32; do
echo "## NVH_PARSE_LANES=$l NVH_PARSE_CUR=2, 3000 packets"
NVH_PARSE_LANES=$l NVH_PARSE_CUR=2 NVH_ALLOW_STALE=1 NVH_LIB=$GRAFT_REPO_ROOT/nvorbis_amd/libnvorbis_hip_dbg.so python tools/dbg_phase_parse.py 3000 2>&1 | tail -12
done | tee gpurun_out/${TAG}_phase_parse.txt
