cd $GRAFT_REPO_ROOT
export NVH_CORPUS_TIMING=1
for r in 1 2; do
for b in 4096 8192 16384 32768; do
  echo -n "batch $b: "; NVH_CORPUS_BATCH=$b python tools/corpus_c5.py --run --scale 1.0 --workers 16 --gpu-parse 2>&1 | grep -o 'decode pass [0-9.]* s\|"decode_s": [0-9.]*\|"verdict": "[^"]*"' | tr '\n' ' '; echo
done; done
for b in 4096 16384; do for l in 2 4; do
  echo -n "batch $b lanes $l: "; NVH_PARSE_LANES=$l NVH_CORPUS_BATCH=$b python tools/corpus_c5.py --run --scale 1.0 --workers 16 --gpu-parse 2>&1 | grep -o 'decode pass [0-9.]* s\|"decode_s": [0-9.]*' | tr '\n' ' '; echo
done; done
echo -n "host parser batch 16384: "; NVH_CORPUS_BATCH=16384 python tools/corpus_c5.py --run --scale 1.0 --workers 16 2>&1 | grep -o 'decode pass [0-9.]* s\|"decode_s": [0-9.]*' | tr '\n' ' '; echo
