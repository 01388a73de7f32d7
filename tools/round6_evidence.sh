#!/bin/bash
# round6_evidence.sh -- one gpurun call, no timing claims beyond what each tool prints: every BASELINE configuration and synthetic shape
# on the committed build (tools/bench_configs.py), C5 at its stated size with both parsers, the randomized parity runs (tools/stress_*.py).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -c "from nvorbis_amd import native; print(native.build_id())" 2>/dev/null | tail -1 > gpurun_out/r06e_build.txt; cat gpurun_out/r06e_build.txt
( time timeout 300 python tools/bench_configs.py ) > gpurun_out/r06e_bench_configs.txt 2>&1; grep -c "us/batch" gpurun_out/r06e_bench_configs.txt
( timeout 150 python tools/corpus_c5.py --run --scale 1.0 --workers 32 --gpu-parse; timeout 150 python tools/corpus_c5.py --run --scale 1.0 --workers 16 ) > gpurun_out/r06e_c5_full.txt 2>&1
grep -o '"gpu_parse": [a-z]*\|"decode_s": [0-9.]*\|"verdict": "[^"]*"' gpurun_out/r06e_c5_full.txt | tr '\n' ' '; echo
( time timeout 100 python tools/stress_slab.py 60 ) > gpurun_out/r06e_stress_slab.txt 2>&1; echo "== stress_slab"; grep -v amdgpu gpurun_out/r06e_stress_slab.txt | head -1
for t in stress_oracle stress_fuzz stress_gpu_parse stress_chunks_seek; do
  ( time timeout 120 python tools/$t.py ) > gpurun_out/r06e_$t.txt 2>&1; echo "== $t"; grep -v amdgpu gpurun_out/r06e_$t.txt | head -1
done
