# pmc_quick.sh -- instruction counts + resource usage of the current kernels (one rocprofv3 pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pq
B="python bench.py --no-cpu-baseline --steps 5 --warmup 2"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d gpurun_out/pq -- $B > gpurun_out/pq.log 2>&1
python tools/pmc_dump.py $(find gpurun_out/pq -name '*.db')
python - <<'PY'
import sqlite3,glob
db=glob.glob('gpurun_out/pq/**/*.db',recursive=True)[0]
cur=sqlite3.connect(db).cursor()
for r in cur.execute("select name, count(*), avg(duration), min(duration), max(grid_x), workgroup_x, max(lds_size), vgpr_count, sgpr_count from kernels where name like 'k_%' group by name"): print(r)
PY
