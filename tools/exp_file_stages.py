"""Where a corpus worker's time goes, one thread, a handful of C5 files: demux, index, stream open, push (= host parse or
packet staging for k_parse), synthesis calls, close.    python tools/exp_file_stages.py [scale] [gpu_parse 0/1] [files]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import nvorbis_amd as nv
from nvorbis_amd.reader import Stream, demux_ogg_array, Context
from tests import c5_corpus
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
gp = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
nfiles = int(sys.argv[3]) if len(sys.argv) > 3 else 12
ws = c5_corpus.writer_setup()
files = [c5_corpus.corpus_file(ws, i, scale) for i in range(nfiles)]
ctx = Context(0)
arena = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
T = {k: 0.0 for k in ("demux", "index", "open", "push", "synth", "close")}
frames = 0
for rep in range(2):
    for k in T: T[k] = 0.0
    frames = 0
    t_all = time.perf_counter()
    for f in files:
        t0 = time.perf_counter(); pa = demux_ogg_array(f); t1 = time.perf_counter(); T["demux"] += t1 - t0
        st = Stream(None, pa[0], pa[1], pa[2]); st.index_packets(pa, 3); st.close(); t2 = time.perf_counter(); T["index"] += t2 - t1
        st = Stream(ctx, pa[0], pa[1], pa[2])
        if gp: st.set_gpu_parse(True)
        t3 = time.perf_counter(); T["open"] += t3 - t2
        nxt = 3
        while True:
            t0 = time.perf_counter()
            if nxt < len(pa) and not st.position()[2]:
                nxt += st.push_packets(pa, nxt, 4096)
                last = nxt >= len(pa) or st.position()[2]
            else:
                last = True
            if last and not st.position()[2]:
                st.push_end()
            t1 = time.perf_counter(); T["push"] += t1 - t0
            if st.pending()[0]:
                frames += st.pending()[0]
                st.synth_device(arena.data_ptr(), arena.numel())
            t2 = time.perf_counter(); T["synth"] += t2 - t1
            if last:
                break
        t0 = time.perf_counter(); st.close(); T["close"] += time.perf_counter() - t0
    ctx.synchronize()
    t_all = time.perf_counter() - t_all
mb = sum(len(f) for f in files) / 1e6
print("scale %g gpu_parse %d: %d files, %.1f MB, %d frames, %.1f ms total = %.2f ms/file, %.2f us/frame" % (scale, gp, nfiles, mb, frames, t_all * 1e3, t_all * 1e3 / nfiles, t_all * 1e6 / max(frames, 1)))
print("  " + "  ".join("%s %.2f ms/file" % (k, v * 1e3 / nfiles) for k, v in T.items()))
