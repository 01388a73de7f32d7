#!/bin/bash
# round5_final.sh -- the last evidence session of round 5 (one gpurun call) for the build that is committed: the two rocprofv3 summaries
# + traffic.json, the -m gpu suite and the bench line, the corpus pass (first call of a process, later calls, fresh arena per call),
# C5 at its stated size with both parsers checked against the oracle's digests.  (tools/round5_closing.sh: + the other configurations
# and the stress runs.)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/profile_round.sh r05_3stream "round 5 closing build, headline loop (three streams)" "--streams 3" > gpurun_out/r05f_prof3.log 2>&1
bash tools/profile_round.sh r05_1stream "round 5 closing build, one stream" "--streams 1" > gpurun_out/r05f_prof1.log 2>&1
head -4 gpurun_out/prof_r05_3stream/summary.txt
( time bash tools/round4_gpu.sh r05final3 ) 2>&1 | tail -12
( python tools/c5_sweep.py --scale 1.0 --reps 1 --cases "16,0,0,0,0;16,0,0,0,0"; NVH_CORPUS_KEEP_CTX=1 python tools/c5_sweep.py --scale 1.0 --reps 4 --cases "16,0,0,0,0";
  NVH_SWEEP_EMPTY_CACHE=1 NVH_CORPUS_KEEP_CTX=1 python tools/c5_sweep.py --scale 1.0 --reps 3 --cases "16,0,0,0,0"; python tools/c5_sweep.py --scale 0.1 --reps 3 --cases "16,0,0,0,0" ) > gpurun_out/r05f_c5_pass.txt 2>&1
grep "^workers" gpurun_out/r05f_c5_pass.txt | cut -c1-200
( python tools/corpus_c5.py --run --scale 1.0 --workers 16; python tools/corpus_c5.py --run --scale 1.0 --workers 16 --gpu-parse ) > gpurun_out/r05f_c5_full.txt 2>&1
grep -o '"gpu_parse": [a-z]*\|"decode_s": [0-9.]*\|"verdict": "[^"]*"' gpurun_out/r05f_c5_full.txt | tr '\n' ' '; echo
