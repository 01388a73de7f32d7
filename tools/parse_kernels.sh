#!/bin/bash
# parse_kernels.sh "ENV=.. ENV=.." ... -- per-kernel durations (rocprofv3 --kernel-trace) of tools/time_parse.py's child under each environment
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  OUT=/tmp/prof_parse; rm -rf $OUT; mkdir -p $OUT
  env $cfg NVH_TIME_PARSE_CHILD=1 rocprofv3 --kernel-trace -d $OUT -- python tools/time_parse.py > $OUT/log.txt 2>&1
  echo "## $cfg"
  python - <<'PY'
import glob, sqlite3
db = glob.glob("/tmp/prof_parse/**/*.db", recursive=True)
if not db:
    print(open("/tmp/prof_parse/log.txt").read()[-1500:])
else:
    cur = sqlite3.connect(db[0]).cursor()
    for r in cur.execute("select name, count(*), avg(duration), min(duration), max(duration), max(grid_x), max(workgroup_x), max(lds_size), max(vgpr_count) from kernels where name like 'k_parse%' group by name order by sum(duration) desc").fetchall():
        print("%-20s n %3d avg %9.1f us min %9.1f max %9.1f grid %8d wg %5d lds %7d vgpr %d" % (r[0], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5], r[6], r[7], r[8]))
PY
done
