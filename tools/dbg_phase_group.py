"""Per-workgroup phase timestamps of k_synth_group2 (kernels_synth.hip: synth_group_body) on the bench workload, alone on the GPU and
with two other decoder instances running beside it (the timed loop's regime):
  NVH_ALLOW_STALE=1 NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so python tools/dbg_phase_group.py     (python -m nvorbis_amd.build --debug)"""
import os, sys, ctypes, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import nvorbis_amd as nv
import bench
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
headers, audio, ch = bench.ll_packets(nv, os.path.join(root, "tests", "golden", "3test.ogg"))
nframes = 4096
insts = []
for k in range(3):
    ctx = nv.Context(0)
    st, bl = bench.make_batches(nv, torch, ctx, headers, audio, 2, nframes, 1, seed_off=k * 13)
    insts.append((ctx, st, bl[0]))
L = nv.lib(); L.nvh_debug_set_buffer.argtypes = [ctypes.c_void_p]
names = ["slabs + constants arrive (DMA round trip, clear, barrier)", "walk of frame 0", "walk of frame 1", "overlap table + barrier",
         "transform (wavefront 0) + plane / slice stores", "barrier (staging drained)", "carry / emission + PCM stores"]

def run(load):
    dbg = torch.zeros(nframes * 24, dtype=torch.int64, device="cuda")
    ctx, st, (b, pcm) = insts[0]
    for c, s_, (bb, pp) in insts:
        for _ in range(3): bb.synth(pp.data_ptr(), pp.numel())
        c.synchronize()
    stop = [False]
    def bg(i):
        c, s_, (bb, pp) = insts[i]
        while not stop[0]:
            for _ in range(8): bb.synth(pp.data_ptr(), pp.numel())
            c.synchronize()
    ths = [threading.Thread(target=bg, args=(i,)) for i in (1, 2)] if load else []
    for t in ths: t.start()
    for _ in range(20): b.synth(pcm.data_ptr(), pcm.numel())
    L.nvh_debug_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
    b.synth(pcm.data_ptr(), pcm.numel())
    ctx.synchronize()
    L.nvh_debug_set_buffer(None)
    stop[0] = True
    for t in ths: t.join()
    d = dbg.cpu().numpy().reshape(nframes, 24)[0::2]
    ok = d[:, 7] != 0
    grp = np.arange(d.shape[0])
    for tag, sel in (("export groups (first launch)", ok & (grp % 2 == 1)), ("import groups (second launch)", ok & (grp % 2 == 0))):
        print("== %s, %s: %d workgroups" % ("two other instances running" if load else "alone", tag, int(sel.sum())))
        for k in range(7):
            dt = (d[:, k + 1] - d[:, k])[sel]
            print("  %-58s mean %7.0f  p50 %7.0f  p90 %7.0f cycles" % (names[k], dt.mean(), np.median(dt), np.percentile(dt, 90)))
        life = (d[:, 7] - d[:, 0])[sel]
        print("  %-58s mean %7.0f  p50 %7.0f  p90 %7.0f cycles" % ("workgroup lifetime", life.mean(), np.median(life), np.percentile(life, 90)))
run(False)
run(True)
