"""C3 (4096 stereo frames, block kinds from the 256/2048 Markov chain, full depth) on one stream and as three decoder instances:
NVH_EMIT_ALWAYS / NVH_NO_EMIT A/B of the paired-emission threshold.   python tools/time_c3.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nvorbis_amd as nv
import bench
from tests import vorbis_encode as ve
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
hdr3 = ve.shipped_headers(open(os.path.join(root, "tests", "golden", "3test.ogg"), "rb").read())
S3 = ve.setup_of(hdr3)
pool3 = ve.packet_pool(S3, 20260928, per_kind=256)
p, _ = ve.stream_from_pool(S3, hdr3, pool3, ve.markov_kinds(np.random.default_rng(7), 4700), np.random.default_rng(7))
insts = []
for k in range(3):
    ctx = nv.Context(0)
    st, bl = bench.make_batches(nv, torch, ctx, p[:3], p[3:], 2, 4096, 2, seed_off=k)
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
run(60)
for c, _, _ in insts: c.synchronize()
t0 = time.perf_counter(); run(900)
for c, _, _ in insts: c.synchronize()
dt = (time.perf_counter() - t0) / 900
print("C3", {k: os.environ[k] for k in os.environ if k.startswith("NVH_") and k not in ("NVH_LIB", "NVH_ALLOW_STALE")}, "| one stream:", one, "pass %.2f us | three streams: %.2f us per pass" % (tot / 100 * 1e3, dt * 1e6))
