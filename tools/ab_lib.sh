# ab_lib.sh NAME... -- same-box A/B of prebuilt libraries build_ab/lib_NAME.so (NVH_LIB override), single stream
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for n in "$@"; do
    echo "$n:"; NVH_LIB=$GRAFT_REPO_ROOT/build_ab/lib_$n.so python bench.py --no-cpu-baseline --streams 1 2>&1 | python tools/bench_brief.py
  done
done
