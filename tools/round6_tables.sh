#!/bin/bash
# round6_tables.sh -- books with an explicit table and vector overrun in the slab kernels (round 6): the parity tests of the new shapes
# and of everything around them, with both parsers and with the descriptor kernels forced
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parse.py -m gpu -x -q -p no:cacheprovider -k "table_books or synthetic or general_bin or fallback or overrun" 2>&1 | tail -6
  NVH_NO_SLAB=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "table_books or overrun" 2>&1 | tail -3 ) | tee gpurun_out/r06_tables_tests.txt
