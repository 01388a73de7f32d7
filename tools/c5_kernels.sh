#!/bin/bash
# c5_kernels.sh -- which kernels the corpus pass (BASELINE configs[4] at its stated size, one GPU, W workers (default 16), GPU parser) spends
# the GPU on: rocprofv3 --kernel-trace --stats around tools/c5_sweep.py's child (three jobs over kept worker contexts).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_c5
rm -rf $OUT; mkdir -p $OUT
python tools/c5_sweep.py --scale 1.0 --reps 1 --cases "${W:-16},0,0,0,0" > /dev/null 2>&1   # (generates and caches the corpus)
NVH_CORPUS_KEEP_CTX=1 NVH_CORPUS_TIMING=1 rocprofv3 --kernel-trace --stats -d $OUT/trace -- python tools/c5_sweep.py --child --scale 1.0 --workers ${W:-16} --reps 3 > $OUT/trace.log 2>&1
W=${W:-16} python - <<'PY' > $OUT/summary.txt
import glob, os, sqlite3
db = glob.glob("gpurun_out/prof_c5/trace/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
span = cur.execute("select min(start), max(end) from kernels").fetchone()
print("# rocprofv3 --kernel-trace --stats: three corpus jobs (1004 files, 21.6 GB of PCM each) over kept worker contexts, %s workers, GPU parser" % os.environ.get("W", "16"))
print("# kernel, launches, total ms, avg us, min us, max us   (first to last kernel: %.1f ms of wall clock)" % ((span[1] - span[0]) / 1e6))
for r in rows:
    print("%-28s %7d %10.1f %9.1f %9.1f %9.1f" % (r[0][:28], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3))
PY
grep "decode pass\|REP_S" $OUT/trace.log >> $OUT/summary.txt
rm -rf $OUT/trace
cat $OUT/summary.txt
