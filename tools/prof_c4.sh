#!/bin/bash
# per-kernel durations of tools/bench_configs.py lines (NVH_BENCH_ONLY selects them): rocprofv3 kernel trace -> summary
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_cfg
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python tools/bench_configs.py > $OUT/log.txt 2>&1
python tools/rocprof_summary.py --trace $(find $OUT/trace -name '*.db') --out $OUT/summary --note "bench_configs ${NVH_BENCH_ONLY}" > /dev/null 2>&1
rm -rf $OUT/trace
grep -v "^W2026" $OUT/log.txt | tail -4
head -12 $OUT/summary.txt
