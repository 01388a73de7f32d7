import os, sys, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import nvorbis_amd as nv, bench
headers, ll, ch = bench.ll_packets(nv, os.path.join(bench.ROOT, "tests", "golden", "3test.ogg"))
ctx = nv.Context(0); st = nv.Stream(ctx, *headers)
st.push_packet(ll[0], -1, 0); st.synth_host()
for i in range(4096): st.push_packet(ll[(i+1) % len(ll)], -1, 0)
b = st.upload_batch(); print(b.stats())
pcm = torch.empty(b.samples*ch, dtype=torch.float32, device="cuda")
dbg = torch.zeros(4096*24, dtype=torch.int64, device="cuda")
# needs the profiling build: python -m nvorbis_amd.build --debug; NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so
L = nv.lib(); L.nvh_debug_set_buffer.argtypes=[ctypes.c_void_p]; L.nvh_debug_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
for _ in range(3): b.synth(pcm.data_ptr(), pcm.numel())
ctx.synchronize(); torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(4096, 24)
t0 = d[:,0].min()
names = ["frame-load", "stage+zero+barrier", "residue", "coupling", "floor", "writeout"]
for k in range(6):
    dt = d[:,k+1]-d[:,k]
    print("%-22s mean %8.0f  p50 %8.0f  p90 %8.0f cycles" % (names[k], dt.mean(), np.median(dt), np.percentile(dt,90)))
print("residue detail: pass/R load %.0f cycles; per-stage mean cycles (0 = stage empty): %s; frames with stage: %s" % (
    (d[:,7]-d[:,2]).mean(), [round(float(d[:,8+k][d[:,8+k]>0].mean())) if (d[:,8+k]>0).any() else 0 for k in range(8)],
    [int((d[:,8+k]>0).sum()) for k in range(8)]))
print('ops per stage (mean over frames):', [round(float(d[:,16+k].mean()),1) for k in range(8)])
life = d[:,6]-d[:,0]
print("WG lifetime mean %.0f p50 %.0f ; kernel span %.0f cycles; start spread p50 %.0f p99 %.0f" % (life.mean(), np.median(life), d[:,6].max()-t0, np.median(d[:,0]-t0), np.percentile(d[:,0]-t0, 99)))
pre = d[:,20]-d[:,1]; prep = d[:,21]-d[:,20]; post = d[:,2]-d[:,21]
print('staging split: loads+copies before prep %.0f, floor_prepare %.0f, barrier wait after %.0f cycles' % (pre.mean(), prep.mean(), post.mean()))
w0, w1 = d[:,22], d[:,23]
ok = w1 > w0
if ok.any():
    span_us = (w1[ok].max() - w0[ok].min()) / 100.0
    mhz = (life[ok] / ((w1[ok] - w0[ok]) / 100.0)).mean()
    order = np.argsort(w0)
    print("wall clock: kernel span %.1f us; shader clock ~%.0f MHz; WG lifetime %.2f us mean; WGs in flight at mid-kernel: %d" % (
        span_us, mhz, ((w1[ok]-w0[ok])/100.0).mean(), int(((w0 <= (w0.min()+w1.max())//2) & (w1 >= (w0.min()+w1.max())//2)).sum())))
    print("start times (us) percentiles 10/50/90/99: %s" % [round(float(np.percentile((w0-w0.min())/100.0, q)),1) for q in (10,50,90,99)])
hw = d[:,19]; hwid = hw & 0xffffffff; xcc = (hw >> 32) & 0xf
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 0x7
key = xcc * 1000 + se * 100 + sh * 20 + cu
t = (w0 - w0.min()) / 100.0
print("distinct (xcc,se,sh,cu):", len(np.unique(key)), "xcc values", np.unique(xcc), "se", np.unique(se), "sh", np.unique(sh), "cu", np.unique(cu))
hist, edges = np.histogram(t, bins=np.arange(0, t.max() + 1.0, 1.0))
print("WG starts per us:", hist.tolist())
blk = np.arange(4096)
print("start time by blockIdx (us) for blocks 0,8,64,512,1024,2047,2048,3000,4095:", [round(float(t[i]),1) for i in (0,8,64,512,1024,2047,2048,3000,4095)])
first = {}
for k in np.unique(key):
    sel = np.sort(t[key == k]); first[k] = sel
cnt = np.array([len(v) for v in first.values()])
print("WGs per CU: min %d max %d mean %.1f" % (cnt.min(), cnt.max(), cnt.mean()))
n_at = lambda T: np.array([(v <= T).sum() for v in first.values()])
for T in (0.3, 1, 2, 4, 6, 8):
    a = n_at(T); print("  by %.1f us: WGs started per CU min %d max %d mean %.2f" % (T, a.min(), a.max(), a.mean()))
L.nvh_debug_set_buffer(None)
# which SIMD does hardware wave 0 of each workgroup land on, and how do blockIdx residues spread over one CU?
simd = (hwid >> 4) & 3
print("SIMD of hw wave 0: counts", np.bincount(simd, minlength=4).tolist())
for k in list(np.unique(key))[:3]:
    sel = np.nonzero(key == k)[0]
    o = sel[np.argsort(t[sel])]
    print(" CU", int(k), "blockIdx in start order:", o.tolist())
im = d[:, 8:13]
if (im[:, 0] > 0).all():
    names = ["step 0 (spectrum -> LDS, _a loads)", "radix passes (twiddle loads)", "D = 4, 2, 1", "bit reversal + steps 7, 8 (_c, _b loads) + stores"]
    print("inverse MDCT of channel 0 (wavefront 0), cycles: " + "; ".join("%s %.0f" % (names[k], (im[:, k + 1] - im[:, k]).mean()) for k in range(4)) +
          "; total %.0f; barrier -> transform start %.0f" % ((im[:, 4] - im[:, 0]).mean(), (im[:, 0] - d[:, 4]).mean()))
