"""C4 (2048 six-channel n = 4096 frames, full depth; psize 48 or 32) on one stream (kernel durations by events) and as three
decoder instances on their own streams (sustained passes): NVH_EMIT8 etc. A/B.   python tools/time_c4_streams.py [psize]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nvorbis_amd as nv
import bench
from tests import vorbis_encode as ve
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
psize = int(sys.argv[1]) if len(sys.argv) > 1 else 48
hdr3 = ve.shipped_headers(open(os.path.join(root, "tests", "golden", "3test.ogg"), "rb").read())
h4 = ve.c4_headers(hdr3, psize=psize)
S4 = ve.setup_of(h4)
pool4 = ve.packet_pool(S4, 148, per_kind=128, class_weights=[0] + [1] * 9)
p, _ = ve.stream_from_pool(S4, h4, pool4, np.ones(2100, dtype=bool), np.random.default_rng(7))
insts = []
for k in range(3):
    ctx = nv.Context(0)
    st, bl = bench.make_batches(nv, torch, ctx, p[:3], p[3:], 6, 2048, 2, seed_off=k)
    insts.append((ctx, st, bl))
b, pcm = insts[0][2][0]
b.time(pcm.data_ptr(), pcm.numel(), 10)
tot, km = b.time(pcm.data_ptr(), pcm.numel(), 100)
one = " ".join("%s %.2f" % (n, v * 1e3) for n, v in zip(b.kernels(), km) if n != "-")
order = [insts[k][2][j] for j in range(2) for k in range(3)]
def run(n):
    for i in range(n):
        bb, pp = order[i % len(order)]
        bb.synth(pp.data_ptr(), pp.numel())
run(60); torch.cuda.synchronize()
for c, _, _ in insts: c.synchronize()
t0 = time.perf_counter(); run(600)
for c, _, _ in insts: c.synchronize()
dt = (time.perf_counter() - t0) / 600
print("psize %d" % psize, {k: os.environ[k] for k in os.environ if k.startswith("NVH_") and k not in ("NVH_LIB", "NVH_ALLOW_STALE")}, "| one stream:", one, "pass %.2f us | three streams: %.2f us per pass" % (tot / 100 * 1e3, dt * 1e6))
