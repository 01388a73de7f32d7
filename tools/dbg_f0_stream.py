"""Floor0 stream decode, GPU vs oracle: how many samples differ, and by how much (several seeds, clip on / off)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import nvorbis_amd as nv
from tests import synth_stream as ss, oracle_py
from tests.test_gpu_parity import _decode_gpu
orc = oracle_py.load()
ctx = nv.Context(0)
tot = bad = 0
for seed in range(1, 13):
    pk, gr, fl = ss.filtered_stream(orc, "floor0_stereo", 120, seed, seed % 2 == 0)
    for clip in (True, False):
        ref, _ = orc.decode_packets(pk, gr, fl, clip=clip)
        got = _decode_gpu(nv, ctx, pk, gr, fl, clip, 37)
        assert got.size == ref.size
        neq = got.view(np.uint32) != ref.view(np.uint32)
        tot += got.size; bad += int(neq.sum())
        if neq.any():
            print("seed", seed, "clip", clip, "differ", int(neq.sum()), "of", got.size, "max abs", float(np.abs(got[neq] - ref[neq]).max()))
print("samples %d, differing %d" % (tot, bad))
