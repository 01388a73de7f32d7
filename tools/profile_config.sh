cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/cfgprof -- python tools/dbg_phase_gen.py floor0_stereo > gpurun_out/cfgprof.log 2>&1
python tools/rocprof_summary.py --trace $(find gpurun_out/cfgprof -name '*.db') --out gpurun_out/cfgprof_summary 2>&1 | tail -3
head -12 gpurun_out/cfgprof_summary.txt
