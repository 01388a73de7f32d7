#!/bin/bash
# round2b_gpu.sh -- the closing GPU session of round 2: parity suite, the bench line, rocprofv3 summary + PMC passes of the default
# pipeline, the other configurations' kernel timings.  Raw rocprofv3 databases are removed again: only the summaries travel back.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r02c_gputest.log
python bench.py > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.err
python bench.py --no-cpu-baseline --streams 1 > gpurun_out/r02c_bench_s1.json 2>/dev/null
bash tools/profile_round.sh r02c_final "round 2 final: k_spectrum_imdct (chain-head list, rotated IMDCT layout, table loads ahead of their phase) + k_ola_compact" > gpurun_out/r02c.log 2>&1
python tools/bench_configs.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02c_bench_configs.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02c_smoke.log 2>&1
find gpurun_out -name '*.db' -delete
tail -3 gpurun_out/r02c_gputest.log; tail -2 gpurun_out/r02c_smoke.log; du -sh gpurun_out
