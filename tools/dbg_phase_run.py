"""Per-phase wall-clock timestamps of the frame-loop kernels (profiling build only):
   python -m nvorbis_amd.build --debug && NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so python tools/dbg_phase_run.py [waves | multi]
   waves = 4 | 6: the run kernel (kernels_run.hip, NVH_RUN=1); multi: k_spectrum_imdct2 (kernels_spectrum2.hip)"""
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench, nvorbis_amd as nv
multi = len(sys.argv) > 1 and sys.argv[1] == "multi"
waves = 4 if multi else (int(sys.argv[1]) if len(sys.argv) > 1 else 6)
if multi:
    os.environ["NVH_MULTI"] = "1"
else:
    os.environ["NVH_RUN_WAVES"] = str(waves)
    os.environ["NVH_RUN"] = "1"
L = nv.lib(); L.nvh_debug_set_buffer.argtypes = [ctypes.c_void_p]
headers, ll, ch = bench.ll_packets(nv, os.path.join(bench.ROOT, "tests", "golden", "3test.ogg"))
ctx = nv.Context(0)
st = nv.Stream(ctx, headers[0], headers[1], headers[2])
st.push_packet(ll[0], -1, 0); st.synth_host()
for i in range(4096): st.push_packet(ll[(i + 1) % len(ll)], -1, 0)
b = st.upload_batch()
pcm = torch.empty(b.samples * ch, dtype=torch.float32, device="cuda")
nwg = 4096
dbg = torch.zeros(nwg * waves * 8 * 8, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
for _ in range(3): b.synth(pcm.data_ptr(), pcm.numel())
ctx.synchronize()
L.nvh_debug_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
b.synth(pcm.data_ptr(), pcm.numel()); ctx.synchronize()
L.nvh_debug_set_buffer(None)
print(b.kernels())
d = dbg.cpu().numpy().reshape(nwg, waves, 8, 8).astype(np.float64)
used = d[:, 0, 0, 0] > 0
d = d[used]
print("workgroups", d.shape[0])
t0 = d[:, :, 0, 7].min()  # first stamp of all
tick = 1e-2  # wall_clock64: 100 MHz -> 10 ns
def us(x): return (x) * tick
print("kernel span: first stamp -> last stamp %.1f us" % us(d[d > 0].max() - t0))
print("per workgroup: start spread %.1f us, lifetime mean %.1f us" % (us(d[:, 0, 0, 7].max() - d[:, 0, 0, 7].min()), us((d[:, :, 7, 5].max(axis=1) - d[:, 0, 0, 7]).mean())))
print("prologue A (wave mean, per role): " + " ".join("%.1f" % us((d[:, w, 0, 6] - d[:, w, 0, 7]).mean()) for w in range(waves)))
for slot in range(8):
    if not (d[:, 0, slot, 0] > 0).all(): break
    top, sw, tl = d[:, :, slot, 0], d[:, :, slot, 1], d[:, :, slot, 2]
    c_end = d[:, :, slot, 3]
    print("frame slot %d: sweep %.1f  tail %.1f  C/A per wave: %s" % (slot, us((sw - top).mean()), us((tl - sw).mean()),
          " ".join("%.1f" % us((c_end[:, w] - tl[:, w]).mean()) for w in range(waves))))
    if slot + 1 < 8 and (d[:, 0, slot + 1, 0] > 0).all():
        print("    -> next top barrier after %.1f us (from tail done)" % us((d[:, :, slot + 1, 0] - tl).mean()))
starts = np.sort(d[:, 0, 0, 7] - t0) * tick
print("workgroups started within 2 us: %d, 5 us: %d, 20 us: %d, 60 us: %d; median start %.1f us" % ((starts < 2).sum(), (starts < 5).sum(), (starts < 20).sum(), (starts < 60).sum(), np.median(starts)))
