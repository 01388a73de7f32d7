import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nvorbis_amd as nv
from tests import synth_stream as ss
from tests import oracle_py
from tests.test_gpu_parity import _decode_gpu
import tests.conftest as cf
orc = oracle_py.load()
ctx = nv.Context(0)
pk, gr, fl = ss.filtered_stream(orc, "floor0_stereo", 200, 5)
ref, info = orc.decode_packets(pk, gr, fl, clip=True)
got = _decode_gpu(nv, ctx, pk, gr, fl, True, 64)
d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
bad = np.nonzero(d > 1e-6)[0]
print("n", got.size, "bad", bad.size, "first", bad[:10], "last", bad[-5:])
if bad.size:
    ch = 2
    s = bad // ch
    print("bad sample ranges:", s.min(), s.max(), "distinct 64-blocks:", np.unique(s // 64)[:40])
    print(got[bad[:8]], ref[bad[:8]])
