#!/bin/bash
# parse_insts.sh -- instruction mix of the GPU packet parser on 4096 real packets (tools/time_parse.py's upload loop): k_parse_slab_u (one
# packet per wavefront, wave-uniform) against k_parse_slab (NVH_NO_PARSE_UNI=1).  Counters in a pass of their own (--kernel-trace only).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_parse; rm -rf $OUT; mkdir -p $OUT
for V in uni div; do
  if [ $V = div ]; then export NVH_NO_PARSE_UNI=1; else unset NVH_NO_PARSE_UNI; fi
  NVH_TIME_PARSE_CHILD=1 FRAMES=4096 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d $OUT/$V -- python tools/time_parse.py > $OUT/$V.log 2>&1
  NVH_TIME_PARSE_CHILD=1 FRAMES=4096 rocprofv3 --kernel-trace --stats -d $OUT/${V}_t -- python tools/time_parse.py > $OUT/${V}_t.log 2>&1
done
python - <<'PY' > $OUT/summary.txt
import glob, sqlite3
print("# k_parse_slab_u (wave-uniform, one packet per wavefront) against k_parse_slab on the same 4096 packets of 3test.ogg: per launch")
for v in ("uni", "div"):
    db = glob.glob("gpurun_out/prof_parse/%s/**/*.db" % v, recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection c where kernel_name like 'k_parse_slab%' and "
                       "grid_size_x = (select max(grid_size_x) from counters_collection c2 where c2.kernel_name = c.kernel_name) "
                       "group by kernel_name, counter_name").fetchall()
    for r in rows:
        print("  %-4s %-16s %-18s avg %14.0f over %d launches" % (v, r[0], r[1], r[2], r[3]))
    db = glob.glob("gpurun_out/prof_parse/%s_t/**/*.db" % v, recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    for r in cur.execute("select name, count(*), avg(duration), min(duration), max(vgpr_count), max(sgpr_count), max(lds_size) from kernels where name like 'k_parse%' "
                         "group by name order by sum(duration) desc"):
        print("  %-4s %-20s %4d launches avg %8.1f us min %8.1f us  vgpr %d sgpr %d lds %d" % (v, r[0], r[1], r[2] / 1e3, r[3] / 1e3, r[4], r[5], r[6]))
PY
rm -rf $OUT/uni $OUT/div $OUT/uni_t $OUT/div_t
cat $OUT/summary.txt
