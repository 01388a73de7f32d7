#!/bin/bash
# abl_trace.sh NAME... -- rocprofv3 kernel durations (no counters) of library builds build_ab/lib_NAME.so in the headline loop
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in "$@"; do
  if [ "$n" = cur ]; then L=""; else L="NVH_ALLOW_STALE=1 NVH_LIB=$GRAFT_REPO_ROOT/build_ab/lib_$n.so"; fi
  rm -rf gpurun_out/tr_$n
  env $L rocprofv3 --kernel-trace --stats -d gpurun_out/tr_$n -- python bench.py --no-check --no-configs --no-cpu-baseline --no-unfused --steps 20 --warmup 5 --min-timed-ms 300 --streams ${STREAMS:-3} > gpurun_out/tr_$n.log 2>&1
  echo "== $n"
  python - <<PY
import sqlite3, glob
db = glob.glob("gpurun_out/tr_$n/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
for r in cur.execute("select name, count(*), avg(duration), min(duration), max(duration) from kernels k where grid_x = (select max(grid_x) from kernels k2 where k2.name = k.name) and name like 'k_synth%' group by name"):
    print("  %-16s calls %6d avg %8.2f us min %8.2f max %8.2f" % (r[0], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3))
# gaps: time between consecutive kernel starts over the whole timed region
rows = cur.execute("select start, end from kernels where name like 'k_synth%' order by start").fetchall()
if len(rows) > 100:
    rows = rows[len(rows) // 4: 3 * len(rows) // 4]
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e in rows)
    print("  mid-half: %d launches, span %.1f us per launch, summed durations / span = %.2f kernels in flight" % (len(rows), span / len(rows) / 1e3, busy / span))
PY
  rm -rf gpurun_out/tr_$n
done
