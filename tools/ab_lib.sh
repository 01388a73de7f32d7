cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  echo "noslp:"; NVH_LIB=$GRAFT_REPO_ROOT/build_ab/lib_noslp.so python bench.py --no-cpu-baseline --streams 1 2>&1 | python tools/bench_brief.py
  echo "default:"; NVH_LIB=$GRAFT_REPO_ROOT/build_ab/lib_old.so python bench.py --no-cpu-baseline --streams 1 2>&1 | python tools/bench_brief.py
done
