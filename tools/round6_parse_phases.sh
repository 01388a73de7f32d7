#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06c8}; shift
for l in 32; do
echo "## NVH_PARSE_LANES=$l, 3000 packets"
NVH_PARSE_LANES=$l NVH_ALLOW_STALE=1 NVH_LIB=$GRAFT_REPO_ROOT/nvorbis_amd/libnvorbis_hip_dbg.so python tools/dbg_phase_parse.py 3000 2>&1 | tail -14 | grep -v "slab \|amdgpu.ids"
done | tee gpurun_out/${TAG}_phase.txt
bash tools/parse_kernels.sh "$@" 2>&1 | grep -v "k_parse_links\|result_out" | tee gpurun_out/${TAG}_kernels.txt
