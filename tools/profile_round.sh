#!/bin/bash
# profile_round.sh NAME [NOTE] [BENCH FLAGS] -- the rocprofv3 passes behind profiles/NAME.{txt,json} (run on the GPU box via gpurun):
#   kernel trace + stats, then FETCH_SIZE, WRITE_SIZE and the instruction mix in separate --pmc passes (never together with
#   a trace domain other than --kernel-trace).  The profiled command is bench.py's headline loop: the passes rotate over
#   enough resident batches to exceed the Infinity Cache (bench.py --working-set-mib), so FETCH_SIZE / WRITE_SIZE are HBM bytes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NAME=${1:-r03}
OUT=gpurun_out/prof_$NAME
rm -rf $OUT; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-configs --no-unfused --c5-scale 0 --steps 20 --warmup 5 --min-timed-ms 300 --streams 1 $3"
# (the counter passes serialise the launches: a shorter timed region there -- the counters are per launch, a few thousand launches are plenty)
BP="python bench.py --no-cpu-baseline --no-configs --no-unfused --c5-scale 0 --steps 10 --warmup 2 --min-timed-ms 40 --streams 1 $3"
BUILD=$(python -c "from nvorbis_amd import native; print(native.build_id())" 2>/dev/null | tail -1)
rocprofv3 --kernel-trace --stats -d $OUT/trace -- $B > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- $BP > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- $BP > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d $OUT/insts -- $BP > $OUT/insts.log 2>&1
python tools/rocprof_summary.py --trace $(find $OUT/trace -name '*.db') --fetch $(find $OUT/fetch -name '*.db') \
  --write $(find $OUT/write -name '*.db') --insts $(find $OUT/insts -name '*.db') --out $OUT/summary \
  --note "$2" --traffic-out $OUT/traffic.json --build "$BUILD" --calibration-from profiles/traffic.json
# the raw rocpd databases are tens of MiB per pass: only the summaries travel back (gpurun merges <= 64 MiB)
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/insts
tail -1 $OUT/trace.log
