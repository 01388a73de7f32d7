# pmc_stalls_run.sh -- wave-cycle breakdown of the run kernel next to the two-kernel path (default), one stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --steps 3 --warmup 1 --streams 1 --min-timed-ms 5"
for mode in run norun; do
  if [ $mode = run ]; then export NVH_RUN=1; else unset NVH_RUN; fi
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM"; do
    i=$((i+1))
    rm -rf gpurun_out/pr_${mode}$i
    rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pr_${mode}$i -- $B > gpurun_out/pr_${mode}$i.log 2>&1 || tail -2 gpurun_out/pr_${mode}$i.log
  done
  echo "== $mode"
  python tools/pmc_dump.py $(find gpurun_out/pr_${mode}* -name '*.db') 2>&1 | grep -v "^k_copy\|^k_parse" 
done
