import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import nvorbis_amd as nv
import bench
from tests import oracle_py
headers, ll, ch = bench.ll_packets(nv, os.path.join(ROOT, "tests/golden/3test.ogg"))
orc = oracle_py.load()
seq = [ll[i % len(ll)] for i in range(1 + 4096 * 2)]
pk = list(headers) + seq
ref, _ = orc.decode_packets(pk, [-1] * len(pk), [0] * len(pk), clip=True, chunk=1 << 18)
per = 4096 * 1024 * 2
ctx = nv.Context(0)
st, bs = bench.make_batches(nv, torch, ctx, headers, ll, ch, bench.FRAMES, 2, seed_off=0)
for rep in range(3):
    for j, (b, p) in enumerate(bs):
        b.synth(p.data_ptr(), p.numel())
        ctx.synchronize()
        a = p.cpu().numpy()
        r = ref[j * per:(j + 1) * per]
        bad = np.flatnonzero(a.view(np.uint32) != r.view(np.uint32))
        print("rep", rep, "batch", j, "mismatching floats", bad.size, "first", bad[:4], "last", bad[-4:] if bad.size else None, flush=True)
