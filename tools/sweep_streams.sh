cd $GRAFT_REPO_ROOT
for e in "" "NVH_NO_EMIT=1"; do for n in 2 3 4 6 8; do
  echo -n "$e streams=$n: "
  env $e timeout 300 python bench.py --streams $n --no-configs --no-cpu-baseline --no-unfused --steps 100 --min-timed-ms 1500 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f M frames/s HBM-resident, %.1f M L3-resident, pass %.2f us' % (d['value']/1e6, d['roofline']['l3_resident']['frames_per_s']/1e6, d['config']['ms_per_pass']*1e3))"
done; done
