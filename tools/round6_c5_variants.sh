cd $GRAFT_REPO_ROOT; export NVH_CORPUS_KEEP_CTX=1
( timeout 300 python tools/c5_sweep.py --scale 1.0 --reps 3 --cases "24,0,0,0,0;32,0,0,0,0"
  NVH_NO_SLEEP_WAIT=1 timeout 300 python tools/c5_sweep.py --scale 1.0 --reps 3 --cases "32,0,0,0,0"
  GPU_MAX_HW_QUEUES=32 timeout 300 python tools/c5_sweep.py --scale 1.0 --reps 3 --cases "32,0,0,0,0"
  NVH_CORPUS_BATCH=8192 timeout 300 python tools/c5_sweep.py --scale 1.0 --reps 3 --cases "32,0,0,0,0" ) 2>&1 | grep "^workers\|summed over the workers" | cut -c1-300
