cd $GRAFT_REPO_ROOT
NVH_LIB=$GRAFT_REPO_ROOT/nvorbis_amd/libnvorbis_hip_dbg.so NVH_ALLOW_STALE=1 python tools/dbg_phase_synth.py > gpurun_out/r05a_phase_greal.txt 2>&1
NVH_LIB=$GRAFT_REPO_ROOT/nvorbis_amd/libnvorbis_hip_dbg.so NVH_ALLOW_STALE=1 python tools/dbg_phase_synth.py grand > gpurun_out/r05a_phase_grand.txt 2>&1
cat gpurun_out/r05a_phase_greal.txt gpurun_out/r05a_phase_grand.txt
bash tools/pmc_stalls.sh 2>&1 | tail -70
