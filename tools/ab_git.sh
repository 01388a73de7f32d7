#!/bin/bash
# ab_git.sh -- same-box A/B of the working tree against HEAD: builds HEAD's library into /tmp, then alternates bench runs.
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  echo "new:"; python bench.py --no-cpu-baseline --streams ${STREAMS:-1} 2>&1 | python tools/bench_brief.py
  echo "old:"; NVH_LIB=$GRAFT_REPO_ROOT/build_ab/lib_old.so python bench.py --no-cpu-baseline --streams ${STREAMS:-1} 2>&1 | python tools/bench_brief.py
done
