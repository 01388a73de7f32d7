#!/bin/bash
# round6_walk2.sh -- the two-frame residue walk (residue_walk_two) against the two walks one after the other (NVH_NO_WALK_TWO=1): the
# headline loop with its digest check on three streams and one, the file-level parity tests, the workgroup's phases.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
brief='import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); print("%.1f M frames/s, digest_ok %s, kernels %s" % (d["value"]/1e6, d.get("pcm_digest_ok"), {k: round(v*1e3,2) for k,v in d["kernels_ms"].items()}))
except Exception as e:
    print("FAILED", e, t[-3:])'
for r in 1 2; do
  for v in "NVH_NO_WALK_TWO=1" "NVH_X=0"; do
    for st in 3 1; do
      echo -n "$v streams=$st: "
      env $v timeout 300 python bench.py --no-configs --no-cpu-baseline --no-unfused --c5-scale 0 --steps 100 --min-timed-ms 1500 --streams $st 2>gpurun_out/r06m_err.log | python -c "$brief"
    done
  done
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_depth.py -m gpu -x -q -p no:cacheprovider -k "ogg_files or clip_samples or partial_reads or fuzzed or resident_batches or bench_workload or synthetic_configs or c3 or markov or grand or c2" 2>&1 | tail -4
NVH_ALLOW_STALE=1 NVH_LIB=$GRAFT_REPO_ROOT/nvorbis_amd/libnvorbis_hip_dbg.so python tools/dbg_phase_group.py 2>&1 | grep -A9 "^== alone" | tee gpurun_out/r06m_group_phases.txt
