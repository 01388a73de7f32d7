"""Experiment: does the host parse scale over Python threads (GIL released inside the library)? And the full path?"""
import os, sys, time, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nvorbis_amd as nv
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
data = open(os.path.join(root, "tests", "golden", "3test.ogg"), "rb").read()
pa = nv.demux_ogg_array(data)
def parse_only(reps):
    for _ in range(reps):
        st = nv.Stream(None, pa[0], pa[1], pa[2]); nxt = 3
        while nxt < len(pa) and not st.position()[2]:
            nxt += st.push_packets(pa, nxt, 100000)
        st.close()
def full(reps, ctx):
    for _ in range(reps):
        st = nv.Stream(ctx, pa[0], pa[1], pa[2]); nxt = 3
        while nxt < len(pa) and not st.position()[2]:
            nxt += st.push_packets(pa, nxt, 100000)
        st.push_end(); pcm = st.synth_host(); st.close()
def gpu_only(reps, ctx):
    st = nv.Stream(ctx, pa[0], pa[1], pa[2]); nxt = 3
    while nxt < len(pa) and not st.position()[2]:
        nxt += st.push_packets(pa, nxt, 100000)
    b = st.upload_batch()
    pcm = torch.empty(b.samples * 2, dtype=torch.float32, device="cuda")
    host = torch.empty(b.samples * 2, dtype=torch.float32).pin_memory()
    for _ in range(reps):
        b.synth(pcm.data_ptr(), pcm.numel()); ctx.synchronize()
    b.free(); st.close()
for mode in ("parse", "full", "gpu"):
    for T in (1, 4, 16, 32):
        reps = 40
        ctxs = [nv.Context(0) for _ in range(T)] if mode != "parse" else [None] * T
        fn = {"parse": lambda c: parse_only(reps), "full": lambda c: full(reps, c), "gpu": lambda c: gpu_only(reps, c)}[mode]
        th = [threading.Thread(target=fn, args=(ctxs[i],)) for i in range(T)]
        t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; dt = time.perf_counter() - t0
        print("%-5s threads %2d: %.0f files/s (%.0f k frames/s)" % (mode, T, T * reps / dt, T * reps * 366 / dt / 1e3), flush=True)
        for c in ctxs:
            if c is not None: c.close()
