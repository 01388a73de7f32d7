#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06c5}; shift
( NVH_PARSE_LANES=${LANES:-8} timeout 900 python -m pytest tests/test_gpu_parse.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 ) | tee gpurun_out/${TAG}_tests_gpu_parse.txt
( NVH_PARSE_LANES=${LANES:-8} NVH_GPU_PARSE=1 NVH_TEST_CHILD=1 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_full_depth.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -8 ) | tee gpurun_out/${TAG}_tests_parity.txt
for l in 8 32; do
echo "## NVH_PARSE_LANES=$l, 3000 packets"
NVH_PARSE_LANES=$l NVH_ALLOW_STALE=1 NVH_LIB=$GRAFT_REPO_ROOT/nvorbis_amd/libnvorbis_hip_dbg.so python tools/dbg_phase_parse.py 3000 2>&1 | tail -11 | grep -v "slab "
done | tee gpurun_out/${TAG}_phase_parse.txt
bash tools/parse_kernels.sh "$@" 2>&1 | grep -v "k_parse_links\|result_out" | tee gpurun_out/${TAG}_kernels.txt
