#!/bin/bash
# round3_phase.sh -- phase stamps of k_synth (profiling build) on G-real and G-rand
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so
python tools/dbg_phase_synth.py > gpurun_out/phase_synth.txt 2>&1
python tools/dbg_phase_synth.py grand >> gpurun_out/phase_synth.txt 2>&1
cat gpurun_out/phase_synth.txt
