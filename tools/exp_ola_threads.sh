for r in 1 2; do
for t in 64 128 256; do echo "OLA_THREADS=$t"; NVH_OLA_THREADS=$t python bench.py --no-cpu-baseline 2>&1 | python tools/bench_brief.py; done
done
for t in 64 128 256; do echo "1-stream OLA_THREADS=$t"; NVH_OLA_THREADS=$t python bench.py --no-cpu-baseline --streams 1 2>&1 | python tools/bench_brief.py; done
