#!/bin/bash
# round6_evidence2.sh -- one gpurun call on the committed build: (1) the corpus pass's kernels (tools/c5_kernels.sh with the pool's
# 32 workers), (2) wait / issue / LDS counters of the headline loop's kernel on one stream (three --pmc passes, nothing else traced),
# (3) one frame per workgroup (NVH_FPW=1, the round-5 form) against frame groups, one decoder instance alone and three.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
W=32 timeout 240 bash tools/c5_kernels.sh > gpurun_out/r06e_c5_kernels.txt 2>&1; tail -14 gpurun_out/r06e_c5_kernels.txt
B="python bench.py --no-cpu-baseline --no-configs --no-unfused --c5-scale 0 --steps 5 --warmup 2 --min-timed-ms 100 --streams 1"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  i=$((i+1)); rm -rf gpurun_out/ps$i
  timeout 120 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/ps$i -- $B > gpurun_out/ps$i.log 2>&1 || tail -2 gpurun_out/ps$i.log
done
python tools/pmc_dump.py $(find gpurun_out/ps* -name '*.db') > gpurun_out/r06e_stalls.txt 2>&1
rm -rf gpurun_out/ps[0-9] gpurun_out/ps[0-9].log
grep -A16 "^k_synth_group2" gpurun_out/r06e_stalls.txt | head -24
BB="python bench.py --no-cpu-baseline --no-configs --no-unfused --c5-scale 0 --steps 100 --min-timed-ms 700"
for s in 1 3; do for f in 2 1; do
  echo -n "NVH_FPW=$f streams $s: "; NVH_FPW=$f timeout 120 $BB --streams $s 2>/dev/null | python tools/bench_brief.py
done; done | tee gpurun_out/r06e_fpw.txt
