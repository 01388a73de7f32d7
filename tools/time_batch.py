"""Kernel durations (hipEvents, nvh_batch_time) of one resident 4096-frame batch of the bench workload (G-real) and of
full-depth packets (G-rand): the quick same-box A/B of library builds.
  NVH_LIB=build_ab/lib_x.so NVH_ALLOW_STALE=1 python tools/time_batch.py [iters]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nvorbis_amd as nv
import bench
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = nv.Context(0)
out = []
for which in ("greal", "grand"):
    if which == "grand":
        from tests import vorbis_encode as ve
        hdr = ve.shipped_headers(open(os.path.join(root, "tests", "golden", "3test.ogg"), "rb").read())
        S3 = ve.setup_of(hdr)
        pool = ve.packet_pool(S3, 20260928, per_kind=256)
        p, _ = ve.stream_from_pool(S3, hdr, pool, np.ones(4200, dtype=bool), np.random.default_rng(7))
        headers, audio = p[:3], p[3:]
    else:
        headers, audio, ch = bench.ll_packets(nv, os.path.join(root, "tests", "golden", "3test.ogg"))
    st, bl = bench.make_batches(nv, torch, ctx, headers, audio, 2, 4096, 1)
    b, pcm = bl[0]
    b.time(pcm.data_ptr(), pcm.numel(), 20)
    tot, km = b.time(pcm.data_ptr(), pcm.numel(), iters)
    names = b.kernels()
    out.append("%s: %s pass %.2f us" % (which, " ".join("%s %.2f" % (n, v * 1e3) for n, v in zip(names, km) if n != "-"), tot / iters * 1e3))
    b.free(); st.close()
print(os.environ.get("NVH_LIB", "default"), "|", " | ".join(out))
