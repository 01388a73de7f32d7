"""Experiment: per-stage latency of the file decode path as the number of host threads grows."""
import os, sys, time, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nvorbis_amd as nv
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
datas = [open(os.path.join(root, "tests", "golden", n + ".ogg"), "rb").read() for n in ("1test", "2test", "3test", "issue6test")]
def worker(ctx, reps, acc):
    T = dict(demux=0.0, open=0.0, push=0.0, synth=0.0, close=0.0)
    for r in range(reps):
        data = datas[r % 4]
        t0 = time.perf_counter(); pa = nv.demux_ogg_array(data)
        t1 = time.perf_counter(); st = nv.Stream(ctx, pa[0], pa[1], pa[2])
        t2 = time.perf_counter(); nxt = 3
        while nxt < len(pa) and not st.position()[2]:
            nxt += st.push_packets(pa, nxt, 100000)
        st.push_end()
        t3 = time.perf_counter(); pcm = st.synth_host()
        t4 = time.perf_counter(); st.close(); t5 = time.perf_counter()
        T["demux"] += t1 - t0; T["open"] += t2 - t1; T["push"] += t3 - t2; T["synth"] += t4 - t3; T["close"] += t5 - t4
    acc.append(T)
for T in (1, 4, 16, 32):
    reps = 48
    ctxs = [nv.Context(0) for _ in range(T)]
    acc = []
    th = [threading.Thread(target=worker, args=(ctxs[i], reps, acc)) for i in range(T)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; dt = time.perf_counter() - t0
    tot = {k: sum(a[k] for a in acc) / (T * reps) * 1e3 for k in acc[0]}
    print("threads %2d: %.0f files/s; per-file ms: %s" % (T, T * reps / dt, {k: round(v, 2) for k, v in tot.items()}), flush=True)
    for c in ctxs: c.close()
