"""Kernel durations of the C4 workload (2048 six-channel n=4096 frames, psize 48, full depth): NVH_OLA_SEGS etc. A/B."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nvorbis_amd as nv
import bench
from tests import vorbis_encode as ve
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
ctx = nv.Context(0)
hdr3 = ve.shipped_headers(open(os.path.join(root, "tests", "golden", "3test.ogg"), "rb").read())
h4 = ve.c4_headers(hdr3, psize=48)
S4 = ve.setup_of(h4)
pool4 = ve.packet_pool(S4, 148, per_kind=128, class_weights=[0] + [1] * 9)
p, _ = ve.stream_from_pool(S4, h4, pool4, np.ones(2100, dtype=bool), np.random.default_rng(7))
st, bl = bench.make_batches(nv, torch, ctx, p[:3], p[3:], 6, 2048, 1)
b, pcm = bl[0]
b.time(pcm.data_ptr(), pcm.numel(), 10)
tot, km = b.time(pcm.data_ptr(), pcm.numel(), 100)
print({k: os.environ[k] for k in os.environ if k.startswith("NVH_")}, " ".join("%s %.2f" % (n, v * 1e3) for n, v in zip(b.kernels(), km) if n != "-"), "pass %.2f us" % (tot / 100 * 1e3))
