#!/bin/bash
# round5_closing.sh -- the closing evidence session behind profiles/r05_{bench_configs,c5_full,stress}.txt (one gpurun call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_multi_rank_gpu.py -x -q -p no:cacheprovider 2>&1 | tail -5
python tools/bench_configs.py > gpurun_out/r05z_bench_configs.txt 2>&1; tail -25 gpurun_out/r05z_bench_configs.txt
( python tools/corpus_c5.py --run --scale 1.0 --workers 16; python tools/corpus_c5.py --run --scale 1.0 --workers 16 --gpu-parse ) > gpurun_out/r05z_c5_full.txt 2>&1
grep -v "^\[" gpurun_out/r05z_c5_full.txt | tail -8
for t in stress_slab stress_oracle stress_fuzz stress_gpu_parse stress_chunks_seek; do
  ( time timeout 600 python tools/$t.py ) > gpurun_out/r05z_$t.txt 2>&1; echo "== $t"; tail -6 gpurun_out/r05z_$t.txt
done
