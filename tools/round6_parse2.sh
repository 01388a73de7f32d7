#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-r06e}
NVH_ALLOW_STALE=1 NVH_LIB=$GRAFT_REPO_ROOT/nvorbis_amd/libnvorbis_hip_dbg.so python tools/dbg_phase_parse.py 4096 2>&1 | tail -12 | tee gpurun_out/${TAG}_phase_parse.txt
timeout 900 python -m pytest tests/test_gpu_parse.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "gpu_parse or parse or fallback or general" 2>&1 | tail -6
brief='import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); print("%.1f M frames/s, digest_ok %s, kernels %s" % (d["value"]/1e6, d.get("pcm_digest_ok"), {k: round(v*1e3,2) for k,v in d["kernels_ms"].items()}))
except Exception as e:
    print("FAILED", e, t[-3:])'
for st in 3 4 5 6 8; do
  echo -n "streams=$st: "
  timeout 300 python bench.py --no-configs --no-cpu-baseline --no-unfused --c5-scale 0 --steps 100 --min-timed-ms 1500 --streams $st 2>gpurun_out/${TAG}_err.log | python -c "$brief"
done
