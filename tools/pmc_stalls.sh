# pmc_stalls.sh -- where the fused spectrum kernel's wave cycles go: wait / issue / LDS / instruction-fetch counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-configs --no-unfused --steps 5 --warmup 2 --min-timed-ms 200 --streams 1"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf gpurun_out/ps$i
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/ps$i -- $B > gpurun_out/ps$i.log 2>&1 || tail -2 gpurun_out/ps$i.log
done
python tools/pmc_dump.py $(find gpurun_out/ps* -name '*.db') 2>&1 | grep -A30 "^k_synth"
python tools/pmc_dump.py $(find gpurun_out/ps* -name '*.db') > gpurun_out/pmc_stalls.txt 2>&1
rm -rf gpurun_out/ps[0-9]
