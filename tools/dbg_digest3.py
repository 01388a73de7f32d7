import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import nvorbis_amd as nv
import bench
from tests import oracle_py
headers, ll, ch = bench.ll_packets(nv, os.path.join(ROOT, "tests/golden/3test.ogg"))
orc = oracle_py.load()
per = 4096 * 1024 * 2
refs = {}
for k in range(3):
    s = 13 * k
    seq = [ll[(s + i) % len(ll)] for i in range(1 + 4096 * 3)]
    pk = list(headers) + seq
    refs[s], _ = orc.decode_packets(pk, [-1] * len(pk), [0] * len(pk), clip=True, chunk=1 << 18)
insts = []
for k in range(3):
    ctx = nv.Context(0)
    st, bs = bench.make_batches(nv, torch, ctx, headers, ll, ch, bench.FRAMES, 3, seed_off=13 * k)
    insts.append((ctx, st, bs, 13 * k))
def report(tag):
    torch.cuda.synchronize()
    for ctx, st, bs, seed in insts:
        for j, (b, p) in enumerate(bs):
            a = p.cpu().numpy()
            r = refs[seed][j * per:(j + 1) * per]
            bad = np.flatnonzero(a.view(np.uint32) != r.view(np.uint32))
            print(tag, "seed", seed, "batch", j, "bad", bad.size, bad[:3], bad[-3:] if bad.size else "", flush=True)
order = [insts[k][2][j] for j in range(3) for k in range(3)]
for rep in range(4):
    for i in range(900):
        b, p = order[i % len(order)]
        b.synth(p.data_ptr(), p.numel())
    report("loop%d" % rep)
