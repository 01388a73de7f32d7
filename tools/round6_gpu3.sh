#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
brief='import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); print("%.1f M frames/s, digest_ok %s, kernels %s" % (d["value"]/1e6, d.get("pcm_digest_ok"), {k: round(v*1e3,2) for k,v in d["kernels_ms"].items()}))
except Exception as e:
    print("FAILED", e, t[-3:])'
run() { # label, env...
  for st in 3 1; do
    echo -n "$1 streams=$st: "
    env "${@:2}" timeout 300 python bench.py --no-configs --no-cpu-baseline --no-unfused --c5-scale 0 --steps 100 --min-timed-ms 1000 --streams $st 2>gpurun_out/r06c_err.log | python -c "$brief"
  done
}
NVH_DEBUG_OCC=1 NVH_FPW=2 python bench.py --no-configs --no-cpu-baseline --no-unfused --c5-scale 0 --steps 5 --min-timed-ms 50 --streams 1 2>&1 >/dev/null | grep "frame groups" | sort | uniq -c
for r in 1 2; do
  run fpw1 NVH_FPW=1
  run fpw2 NVH_FPW=2
  run fpw2pf NVH_FPW=2 NVH_GROUP_PREFETCH=1
  run fpw4 NVH_FPW=4
done
for f in "NVH_FPW=2" "NVH_FPW=4" "NVH_FPW=2 NVH_GPU_PARSE=1" "NVH_FPW=2 NVH_EMIT_ALWAYS=1" "NVH_FPW=2 NVH_POISON_PLANES=1"; do
  echo "== pytest subset $f"
  env $f timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ogg_files or clip_samples or partial_reads or fuzzed or resident_batches or bench_workload or synthetic_configs or stream_chunks or pipelined" 2>&1 | tail -3
done
