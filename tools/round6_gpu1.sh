#!/bin/bash
# round6_gpu1.sh -- first GPU session of round 6: frame groups (NVH_FPW=2 / 4) against the one-frame form (NVH_FPW=1): the headline
# loop with its digest check, one stream and three; then the file-level parity tests under each form.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
brief='import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); print("%.1f M frames/s, pair %s, digest_ok %s, kernels %s" % (d["value"]/1e6, d["roofline"].get("kernel"), d.get("pcm_digest_ok"), {k: round(v*1e3,2) for k,v in d["kernels_ms"].items()}))
except Exception as e:
    print("FAILED", e, t[-3:])'
for r in 1 2; do
  for f in 1 2 4; do
    for st in 3 1; do
      echo -n "FPW=$f streams=$st: "
      NVH_FPW=$f timeout 300 python bench.py --no-configs --no-cpu-baseline --no-unfused --c5-scale 0 --steps 100 --min-timed-ms 1000 --streams $st 2>gpurun_out/r06a_err_$f.log | python -c "$brief"
    done
  done
done
for f in 2 4; do
  echo "== pytest subset NVH_FPW=$f"
  NVH_FPW=$f timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ogg_files or clip_samples or partial_reads or fuzzed or resident_batches or bench_workload or synthetic_configs or stream_chunks or pipelined" 2>&1 | tail -5
done
