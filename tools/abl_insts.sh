#!/bin/bash
# abl_insts.sh NAME... -- instruction counts (VALU / SALU / LDS wave-instructions per launch) of ablation builds
# (build_ab/lib_NAME.so; "cur" = the tree's own) on one stream: where the pass's instructions are, phase by phase.
# Few launches per build: a --pmc pass serialises them (never trace near-empty kernels through bench.py's 2 s loops).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in "$@"; do
  if [ "$n" = cur ]; then L=""; else L="NVH_ALLOW_STALE=1 NVH_LIB=$GRAFT_REPO_ROOT/build_ab/lib_$n.so"; fi
  rm -rf gpurun_out/ai_$n
  env $L timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d gpurun_out/ai_$n -- \
    python bench.py --no-check --no-configs --no-cpu-baseline --no-unfused --c5-scale 0 --steps 2 --warmup 1 --min-timed-ms 20 --streams 1 > gpurun_out/ai_$n.log 2>&1
  echo "== $n"
  python tools/pmc_dump.py $(find gpurun_out/ai_$n -name '*.db') 2>&1 | grep -A5 "^k_synth"
  rm -rf gpurun_out/ai_$n
done
