#!/bin/bash
# ab_libs.sh NAME... -- same-box A/B of library builds build_ab/lib_NAME.so ("cur" = the tree's own library), interleaved, 3 rounds
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for n in "$@"; do
    if [ "$n" = cur ]; then python tools/time_batch.py 2>&1 | grep -v amdgpu.ids | tail -1
    else NVH_ALLOW_STALE=1 NVH_LIB=$GRAFT_REPO_ROOT/build_ab/lib_$n.so python tools/time_batch.py 2>&1 | grep -v amdgpu.ids | tail -1; fi
  done
done
