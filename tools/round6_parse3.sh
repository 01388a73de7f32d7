#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06f}
NVH_ALLOW_STALE=1 NVH_LIB=$GRAFT_REPO_ROOT/nvorbis_amd/libnvorbis_hip_dbg.so python tools/dbg_phase_parse.py 4096 2>&1 | tail -9 | tee gpurun_out/${TAG}_phase_parse.txt
for f in 1024 4096 32768; do FRAMES=$f python tools/time_parse.py; FRAMES=$f CORPUS=1 python tools/time_parse.py; done 2>&1 | tee gpurun_out/${TAG}_time_parse.txt
timeout 900 python -m pytest tests/test_gpu_parse.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "gpu_parse or parse or general" 2>&1 | tail -4
( NVH_CORPUS_KEEP_CTX=1 python tools/c5_sweep.py --scale 1.0 --reps 3 --cases "16,0,0,0,0;16,1,0,0,0;16,2,0,0,0" ) > gpurun_out/${TAG}_c5_pass.txt 2>&1
grep -E "^workers" gpurun_out/${TAG}_c5_pass.txt | cut -c1-200
