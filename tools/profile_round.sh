#!/bin/bash
# profile_round.sh NAME -- the rocprofv3 passes behind profiles/NAME.{txt,json} (run on the GPU box via gpurun):
#   kernel trace + stats, then FETCH_SIZE, WRITE_SIZE and the instruction mix in separate --pmc passes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NAME=${1:-r01}
OUT=gpurun_out/prof_$NAME
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --steps 20 --warmup 5 --streams 1"
rocprofv3 --kernel-trace --stats -d $OUT/trace -- $B > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- $B > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d $OUT/insts -- $B > $OUT/insts.log 2>&1
python tools/rocprof_summary.py --trace $(find $OUT/trace -name '*.db') --fetch $(find $OUT/fetch -name '*.db') \
  --write $(find $OUT/write -name '*.db') --insts $(find $OUT/insts -name '*.db') --out $OUT/summary \
  --note "$2"
tail -1 $OUT/trace.log
