# pmc_phases.sh -- VALU/SALU instruction counts of k_spectrum with phases masked out (NVH_DEBUG_SPECTRUM_MASK:
# needs the profiling build first: python -m nvorbis_amd.build --debug
# bit0 residue, bit1 fused tail, bit2 floor prepare, bit3 inverse MDCT); differences give the per-phase cost.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 15 7 6 5 3; do
  rm -rf gpurun_out/pm$m
  NVH_MULTI=0 NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so NVH_DEBUG_SPECTRUM_MASK=$m rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES -d gpurun_out/pm$m -- python bench.py --no-cpu-baseline --steps 5 --warmup 2 --streams 1 > gpurun_out/pm$m.log 2>&1
  echo "mask $m"; python tools/pmc_dump.py $(find gpurun_out/pm$m -name '*.db') | grep -A6 "^k_spectrum"
done
