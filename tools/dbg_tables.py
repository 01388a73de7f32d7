import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import nvorbis_amd as nv
from tests import synth_stream as ss, oracle_py
from tests.test_gpu_parity import _decode_gpu
orc = oracle_py.load()
ctx = nv.Context(0)
name = sys.argv[1] if len(sys.argv) > 1 else "table_books_pair"
pk, gr, fl = ss.filtered_stream(orc, name, 150, 41, consistent_windows=False)
for clip in (True, False):
    ref, info = orc.decode_packets(pk, gr, fl, clip=clip)
    got = _decode_gpu(nv, ctx, pk, gr, fl, clip, 1024)
    bad = np.nonzero(got.view(np.uint32) != ref.view(np.uint32))[0]
    print(name, os.environ.get("NVH_NO_SLAB"), "clip", clip, "sizes", got.size, ref.size, "mismatches", bad.size, "first", bad[:8], "got", got[bad[:4]], "ref", ref[bad[:4]])
    if bad.size:
        print("  span", bad.min(), bad.max(), "channels", set((bad % info["channels"]).tolist()))
