cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pp1 gpurun_out/pp2
FRAMES=4096 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d gpurun_out/pp1 -- python tools/e2e_gpu_parse.py > gpurun_out/pp1.log 2>&1
FRAMES=4096 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/pp2 -- python tools/e2e_gpu_parse.py > gpurun_out/pp2.log 2>&1
python tools/pmc_dump.py $(find gpurun_out/pp1 gpurun_out/pp2 -name '*.db') | grep -A9 "^k_parse_slab$"
