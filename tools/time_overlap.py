"""Do two HIP streams overlap the VALU-bound synthesis kernel with the HBM-bound overlap-add kernel?  Two contexts (one HIP
stream each), one Vorbis stream of resident batches on each; passes issued alternately without synchronisation, against the
same passes on one context.    python tools/time_overlap.py [batches per stream]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import nvorbis_amd as nv
import bench
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
headers, audio, ch = bench.ll_packets(nv, os.path.join(root, "tests", "golden", "3test.ogg"))
if len(sys.argv) > 3 and sys.argv[3] == "c3":  # block kinds from the 256/2048 Markov chain, full depth (bench.py: C3_markov)
    import numpy as np
    from tests import vorbis_encode as ve
    hdr3 = ve.shipped_headers(open(os.path.join(root, "tests", "golden", "3test.ogg"), "rb").read())
    S3 = ve.setup_of(hdr3)
    pool3 = ve.packet_pool(S3, 20260928, per_kind=256)
    p, _ = ve.stream_from_pool(S3, hdr3, pool3, ve.markov_kinds(np.random.default_rng(7), 4700), np.random.default_rng(7))
    headers, audio = p[:3], p[3:]
nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctxs = [nv.Context(0) for _ in range(nctx)]
sets = []
for i, c in enumerate(ctxs):
    st, bl = bench.make_batches(nv, torch, c, headers, audio, ch, 4096, nb, seed_off=17 * i)
    sets.append((st, bl))
one = nv.Context(0)
st1 = [bench.make_batches(nv, torch, one, headers, audio, ch, 4096, nb, seed_off=17 * i) for i in range(nctx)]

def sync():
    for c in ctxs + [one]:
        c.synchronize()
    torch.cuda.synchronize()

def run(order, n):
    for k in range(n):
        b, p = order[k % len(order)]
        b.synth(p.data_ptr(), p.numel())

def order_of(s):
    o = []
    for j in range(nb):
        for st, bl in s:
            o.append(bl[j])
    return o

for name, s in (("%d HIP streams" % nctx, sets), ("one HIP stream", st1), ("%d HIP streams" % nctx, sets), ("one HIP stream", st1)):
    o = order_of(s)
    run(o, 200); sync()
    t0 = time.perf_counter()
    run(o, 4000); sync()
    el = time.perf_counter() - t0
    print("%-16s %.2f us per pass of 4096 frames  %.1f M frames/s" % (name, el / 4000 * 1e6, 4096 * 4000 / el / 1e6), flush=True)
