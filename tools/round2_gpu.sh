#!/bin/bash
# round2_gpu.sh -- the round-2 GPU session: parity suite, bench lines, rocprofv3 summaries (default pipeline and the run kernel),
# the stall / bank-conflict counters DESIGN.md quotes, run-kernel phase timings.  Raw rocprofv3 databases are removed again:
# only the summaries travel back.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$1" != "profiles-only" ]; then
  timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r02_gputest.log
  python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
  python bench.py --no-cpu-baseline --streams 1 > gpurun_out/r02_bench_s1.json 2>/dev/null
fi
bash tools/profile_round.sh r02a_default "round 2 default pipeline: k_spectrum_imdct (coupling in the chain walk, per-channel 8-bin tail) + k_ola_compact" > gpurun_out/r02a.log 2>&1
NVH_RUN=1 bash tools/profile_round.sh r02b_run "round 2 run kernel (opt-in NVH_RUN=1): k_run6, everything in one launch" > gpurun_out/r02b.log 2>&1
bash tools/pmc_stalls_run.sh > gpurun_out/r02_stalls.txt 2>&1
bash tools/dbg_phase_run.sh > gpurun_out/r02_run_phases.txt 2>&1
bash tools/run_variants.sh > gpurun_out/r02_run_variants.txt 2>&1
find gpurun_out -name '*.db' -delete
rm -rf gpurun_out/pr_run* gpurun_out/pr_norun* gpurun_out/valu_rate
tail -4 gpurun_out/r02_gputest.log 2>/dev/null; du -sh gpurun_out
