"""Stress: GPU packet parser vs host parser on many random synthetic streams (every shape inside the GPU parser's limits),
random batch sizes and parser launch shapes (nvh_ctx_set_parse_lanes); PCM must be identical bit for bit.  Not part of the test suite (minutes of runtime)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nvorbis_amd as nv
from tests import synth_stream as ss, oracle_py
from tests.test_gpu_parse import _decode
orc = oracle_py.load()
ctx = nv.Context(0)
rng = np.random.default_rng(2026)
names = ["mono_res0_small_blocks", "stereo_res1_coupled", "three_ch_res2_misaligned", "six_ch_res2_4096", "two_submaps",
         "equal_blocks_overrun", "mono_8192", "stereo_8192", "res0_slab", "odd_dims_slab", "res2_alias_stereo", "two_pass_slab", "res0_3ch"]
t0 = time.time(); n = 0; frames = 0
seeds = int(os.environ.get("SEEDS", "12"))
for name in names:
    for seed in range(100, 100 + seeds):
        consistent = bool(seed & 1)
        pk, gr, fl = ss.filtered_stream(orc, name, int(rng.integers(20, 120)), seed, consistent)
        bf = int(rng.choice([1, 2, 3, 7, 16, 50, 400]))
        a = _decode(nv, ctx, pk, gr, fl, False, bf)
        ctx.set_parse_lanes(int(rng.choice([0, 0, 1, 2, 4, 8, 64])))  # the launch shape a worker pool asks for: same PCM
        b = _decode(nv, ctx, pk, gr, fl, True, bf)
        ctx.set_parse_lanes(0)
        assert a.size == b.size and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (name, seed, bf)
        n += 1; frames += len(pk) - 3
print("gpu-parse == host-parse on %d random streams (%d packets), %.0f s" % (n, frames, time.time() - t0))
