# pmc_c4.sh -- LDS / instruction counters of the general spectrum kernel on the full-depth six-channel stream (tools/dbg_phase_c4.py as the driver)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc4
NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY -d gpurun_out/pmc4 -- python tools/dbg_phase_c4.py > gpurun_out/pmc4.log 2>&1
python tools/pmc_dump.py $(find gpurun_out/pmc4 -name '*.db') | grep -A8 "^k_spectrum_gen8"
rm -rf gpurun_out/pmc4
