#!/bin/bash
# round5_closing.sh -- the closing evidence session of a round (one gpurun call): the two rocprofv3 summaries + traffic.json, the -m gpu
# suite and the bench line, the other configurations, C5 at its stated size with both parsers, the stress runs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/profile_round.sh r05_3stream "round 5 closing build, headline loop (three streams)" "--streams 3" > gpurun_out/r05y_prof3.log 2>&1
bash tools/profile_round.sh r05_1stream "round 5 closing build, one stream" "--streams 1" > gpurun_out/r05y_prof1.log 2>&1
head -4 gpurun_out/prof_r05_3stream/summary.txt
( time bash tools/round4_gpu.sh r05final2 ) 2>&1 | tail -12
python tools/bench_configs.py > gpurun_out/r05y_bench_configs.txt 2>&1; grep -c "us/batch" gpurun_out/r05y_bench_configs.txt
( python tools/corpus_c5.py --run --scale 1.0 --workers 16; python tools/corpus_c5.py --run --scale 1.0 --workers 16 --gpu-parse ) > gpurun_out/r05y_c5_full.txt 2>&1
grep -o '"gpu_parse": [a-z]*\|"decode_s": [0-9.]*\|"verdict": "[^"]*"' gpurun_out/r05y_c5_full.txt | tr '\n' ' '; echo
for t in stress_slab stress_oracle stress_fuzz stress_gpu_parse stress_chunks_seek; do
  ( time timeout 600 python tools/$t.py ) > gpurun_out/r05y_$t.txt 2>&1; echo "== $t"; grep -v amdgpu gpurun_out/r05y_$t.txt | head -1
done
