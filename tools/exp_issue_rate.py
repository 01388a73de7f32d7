import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import nvorbis_amd as nv
import bench
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
headers, audio, ch = bench.ll_packets(nv, os.path.join(root, "tests", "golden", "3test.ogg"))
ctxs = [nv.Context(0) for _ in range(4)]
sets = [bench.make_batches(nv, torch, c, headers, audio, ch, 4096, 2, seed_off=17 * i) for i, c in enumerate(ctxs)]
order = [bl[j] for j in range(2) for st, bl in sets]
def sync():
    for c in ctxs: c.synchronize()
for n in (64, 128, 256, 512, 2000):
    sync()
    t0 = time.perf_counter()
    for k in range(n):
        b, p = order[k % len(order)]
        b.synth(p.data_ptr(), p.numel())
    t1 = time.perf_counter()
    sync()
    t2 = time.perf_counter()
    print("n=%d issue %.2f us/pass, total %.2f us/pass" % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)
