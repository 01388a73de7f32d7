# run_variants.sh -- the run kernel (NVH_RUN=1; 4 / 6 wavefronts per workgroup, run lengths) against the default two-kernel path, one stream
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --streams 1 --steps 20 --warmup 5 --min-timed-ms 50"
for v in "NVH_RUN=1 NVH_RUN_WAVES=6" "NVH_RUN=1 NVH_RUN_WAVES=4" "NVH_RUN=1 NVH_RUN_WAVES=6 NVH_RUN_LEN=1" "NVH_RUN=1 NVH_RUN_WAVES=6 NVH_RUN_LEN=6" "NVH_RUN=1 NVH_RUN_WAVES=4 NVH_RUN_LEN=2" "NVH_X=0"; do
  echo "== $v"
  env $v $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], round(d['config']['ms_per_pass']*1000,1),'us/pass')"
done
