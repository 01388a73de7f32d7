#!/bin/bash
# repro_uncached.sh -- the round-4 experiment: work planes in hipDeviceMallocUncached memory (NVH_UNCACHED_PLANES=1), host parser and
# GPU parser, over the parity tests that decode through the streaming reader and resident batches
cd $GRAFT_REPO_ROOT
for gp in 0 1; do
  echo "== NVH_UNCACHED_PLANES=1 NVH_GPU_PARSE=$gp"
  if [ $gp = 1 ]; then export NVH_GPU_PARSE=1; else unset NVH_GPU_PARSE; fi
  NVH_TEST_CHILD=1 NVH_UNCACHED_PLANES=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parse.py tests/test_full_depth.py -m gpu -q -p no:cacheprovider -x -k "not fallback and not c5_corpus_1004" 2>&1 | tail -15
done
