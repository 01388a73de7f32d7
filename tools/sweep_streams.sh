cd $GRAFT_REPO_ROOT
for s in 1 2 3 4 6 8; do
  echo -n "streams $s: "
  python bench.py --no-configs --no-cpu-baseline --no-unfused --c5-scale 0 --steps 100 --min-timed-ms 700 --streams $s 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f us per pass (%.1f M frames/s) digest %s' % (4096e6/d['value'], d['value']/1e6, d['pcm_digest_ok']))"
done
