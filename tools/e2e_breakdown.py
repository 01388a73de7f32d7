"""Where the end-to-end time of a small file goes (one host thread): demux / open (setup parse + upload) / parse / synth."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nvorbis_amd as nv
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
ctx = nv.Context(0)
for name in ("1test", "3test", "issue6test"):
    data = open(os.path.join(root, "tests", "golden", name + ".ogg"), "rb").read()
    T = dict(demux=0.0, open=0.0, push=0.0, synth=0.0, close=0.0)
    R = 30
    frames = 0
    for r in range(R):
        t0 = time.perf_counter(); pa = nv.demux_ogg_array(data)
        t1 = time.perf_counter(); st = nv.Stream(ctx, pa[0], pa[1], pa[2])
        t2 = time.perf_counter()
        nxt = 3; tp = 0.0; ts = 0.0
        while True:
            a = time.perf_counter()
            if nxt < len(pa) and not st.position()[2]:
                nxt += st.push_packets(pa, nxt, 4096)
            else:
                st.push_end()
            b = time.perf_counter()
            fr, smp = st.pending()
            frames += fr
            pcm = st.synth_host() if fr else None
            c = time.perf_counter()
            tp += b - a; ts += c - b
            if nxt >= len(pa) or st.position()[2]:
                if fr == 0: break
        t3 = time.perf_counter(); st.close(); t4 = time.perf_counter()
        T["demux"] += t1 - t0; T["open"] += t2 - t1; T["push"] += tp; T["synth"] += ts; T["close"] += t4 - t3
    print(name, "frames/file %d" % (frames // R), {k: "%.2f ms" % (v / R * 1e3) for k, v in T.items()})
