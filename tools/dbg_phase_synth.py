"""Per-workgroup phase timestamps of k_synth (kernels_synth.hip) on the bench workload (G-real) or on full-depth packets
(G-rand):  NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so python tools/dbg_phase_synth.py [grand]
Needs the profiling build (python -m nvorbis_amd.build --debug)."""
import os, sys, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import nvorbis_amd as nv
import bench
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
ctx = nv.Context(0)
nframes, nchan = 4096, 2
if len(sys.argv) > 1 and sys.argv[1] == "c4":
    from tests import vorbis_encode as ve
    hdr3 = ve.shipped_headers(open(os.path.join(root, "tests", "golden", "3test.ogg"), "rb").read())
    h4 = ve.c4_headers(hdr3, psize=48)
    S4 = ve.setup_of(h4)
    pool4 = ve.packet_pool(S4, 148, per_kind=128, class_weights=[0] + [1] * 9)
    p, _ = ve.stream_from_pool(S4, h4, pool4, np.ones(2100, dtype=bool), np.random.default_rng(7))
    headers, audio = p[:3], p[3:]
    nframes, nchan = 2048, 6
elif len(sys.argv) > 1 and sys.argv[1] == "grand":
    from tests import vorbis_encode as ve
    hdr = ve.shipped_headers(open(os.path.join(root, "tests", "golden", "3test.ogg"), "rb").read())
    S3 = ve.setup_of(hdr)
    pool = ve.packet_pool(S3, 20260928, per_kind=256)
    p, _ = ve.stream_from_pool(S3, hdr, pool, np.ones(4200, dtype=bool), np.random.default_rng(7))
    headers, audio = p[:3], p[3:]
else:
    headers, audio, ch = bench.ll_packets(nv, os.path.join(root, "tests", "golden", "3test.ogg"))
st, bl = bench.make_batches(nv, torch, ctx, headers, audio, nchan, nframes, 1)
b, pcm = bl[0]
print(b.stats())
dbg = torch.zeros(nframes * 24, dtype=torch.int64, device="cuda")
L = nv.lib(); L.nvh_debug_set_buffer.argtypes = [ctypes.c_void_p]
for _ in range(3): b.synth(pcm.data_ptr(), pcm.numel())
ctx.synchronize()
L.nvh_debug_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
b.synth(pcm.data_ptr(), pcm.numel())
ctx.synchronize(); torch.cuda.synchronize()
L.nvh_debug_set_buffer(None)
print(b.kernels())
d = dbg.cpu().numpy().reshape(nframes, 24)
names = ["DMA round trip + clear + barrier", "header (+ rest of a big slab)", "residue walk", "(coupling +) floor multiply", "inverse MDCT + store"]
for k in range(5):
    dt = d[:, k + 1] - d[:, k]
    print("%-34s mean %8.0f  p50 %8.0f  p90 %8.0f cycles" % (names[k], dt.mean(), np.median(dt), np.percentile(dt, 90)))
if d[:, 6].any():  # k_synth8 / unfused floor: the wait for the slowest walker, the coupling passes, the floor multiply
    for nm, a, b_ in (("  wait for the slowest walker", 3, 6), ("  coupling passes", 6, 7), ("  floor multiply", 7, 4)):
        dt = d[:, b_] - d[:, a]
        print("%-34s mean %8.0f  p50 %8.0f  p90 %8.0f cycles" % (nm, dt.mean(), np.median(dt), np.percentile(dt, 90)))
# odd frames (first launch: k_synth_tail) and even frames (second launch: k_synth_emit) apart, with the transform's own stamps
# (imdct_wave.h: 10 entry, 11 step 0 done, 12 radix passes done, 13 D = 4, 2, 1 done, 14 end) and the emission's (15 staging
# issued, 16 transform done, 17 behind the barrier, 18 emitted)
def show(tag, sel, pairs):
    for nm, a, b_ in pairs:
        ok = sel & (d[:, a] != 0) & (d[:, b_] != 0)
        if ok.sum() == 0:
            continue
        dt = (d[:, b_] - d[:, a])[ok]
        print("  %-5s %-40s mean %8.0f  p50 %8.0f  p90 %8.0f cycles (n=%d)" % (tag, nm, dt.mean(), np.median(dt), np.percentile(dt, 90), int(ok.sum())))
fidx = np.arange(nframes)
odd, even = (fidx & 1) == 1, (fidx & 1) == 0
common = [("DMA + clear + barrier", 0, 1), ("header", 1, 2), ("walk (+ fused floor)", 2, 3), ("walk end -> transform entry", 3, 10),
          ("step 0 (incl. barrier, table loads)", 10, 11), ("radix passes", 11, 12), ("D = 4, 2, 1", 12, 13), ("bit reversal + steps 7, 8 (+ stores)", 13, 14)]
show("odd", odd, common + [("transform end -> WG end", 14, 5), ("lifetime", 0, 5)])
show("even", even, common + [("walk end -> staging issued", 3, 15), ("transform end -> barrier passed", 16, 17), ("emission", 17, 18), ("lifetime", 0, 5)])
life = d[:, 5] - d[:, 0]
print("WG lifetime mean %.0f p50 %.0f p90 %.0f cycles" % (life.mean(), np.median(life), np.percentile(life, 90)))
w0 = d[:, 22].min()
start = (d[:, 22] - w0) / 100.0  # 100 MHz wall clock -> us
end = (d[:, 23] - w0) / 100.0
print("wall clock: first start 0, last start %.1f us, last end %.1f us; starts p50 %.1f p90 %.1f; lifetime mean %.2f us" % (
    start.max(), end.max(), np.median(start), np.percentile(start, 90), (end - start).mean()))
hist, edges = np.histogram(start, bins=12)
print("start histogram (us):", [(round(float(edges[i]), 1), int(hist[i])) for i in range(len(hist))])
tot, km = b.time(pcm.data_ptr(), pcm.numel(), 20)
print({n: round(v * 1e3, 1) for n, v in zip(b.kernels(), km)})
