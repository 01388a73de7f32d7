#!/bin/bash
# profile_parse.sh -- rocprofv3 kernel trace of the end-to-end path with the GPU packet parser (profiles/r01d_gpu_parse.txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_parse
rm -rf $OUT; mkdir -p $OUT
for n in 4096 32768; do
  FRAMES=$n rocprofv3 --kernel-trace --stats -d $OUT/t$n -- python tools/e2e_gpu_parse.py > $OUT/t$n.log 2>&1
  grep "parse:" $OUT/t$n.log
done
python - <<'PY' > gpurun_out/prof_parse/summary.txt
import sqlite3, glob
print("# rocprofv3 --kernel-trace --stats -- python tools/e2e_gpu_parse.py (FRAMES=4096 / 32768): GPU packet parser + synthesis, one stream")
print("# end-to-end lines of the same runs (host timer, includes PCIe):")
for n in (4096, 32768):
    for l in open("gpurun_out/prof_parse/t%d.log" % n):
        if "parse:" in l: print("#   FRAMES=%d  %s" % (n, l.strip()))
print("%-8s %-20s %6s %12s %12s %10s %6s %6s %8s" % ("frames", "kernel", "calls", "avg_us", "max_us", "grid", "vgpr", "sgpr", "lds_B"))
for n in (4096, 32768):
    db = glob.glob("gpurun_out/prof_parse/t%d/**/*.db" % n, recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    for r in cur.execute("select name, count(*), avg(duration)/1e3, max(duration)/1e3, max(grid_x), max(vgpr_count), max(sgpr_count), max(lds_size) "
                         "from kernels k where name like 'k_%' and grid_x = (select max(grid_x) from kernels k2 where k2.name = k.name) group by name order by 3 desc"):
        print("%-8d %-20s %6d %12.1f %12.1f %10d %6d %6d %8d" % ((n,) + tuple(r)))
PY
cat gpurun_out/prof_parse/summary.txt
