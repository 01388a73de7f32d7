# ab.sh VAR -- same-box A/B of an environment toggle: bench timings + VALU instruction counts
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$1
for r in 1 2; do
  echo "default:"; python bench.py --no-cpu-baseline 2>&1 | python tools/bench_brief.py
  echo "$V=1:"; env $V=1 python bench.py --no-cpu-baseline 2>&1 | python tools/bench_brief.py
done
B="python bench.py --no-cpu-baseline --steps 5 --warmup 2"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d gpurun_out/ab_a -- $B > gpurun_out/ab_a.log 2>&1
env $V=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d gpurun_out/ab_b -- $B > gpurun_out/ab_b.log 2>&1
echo "default:"; python tools/pmc_dump.py $(find gpurun_out/ab_a -name '*.db') | grep -A4 k_spectrum
echo "$V=1:"; python tools/pmc_dump.py $(find gpurun_out/ab_b -name '*.db') | grep -A4 k_spectrum
