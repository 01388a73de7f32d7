"""Experiment: how the corpus decode pass (nvorbis_amd/corpus.py: decode_files_to_device) scales over worker threads, and
where a worker's wall time goes under contention.  The C5 corpus at full size, its demultiplexed packet arrays prepared up
front; each run decodes all files with T workers into a device arena.
  [HOST_PARSE=1] [BF=4096] python tools/exp_corpus_scaling.py [threads ...]"""
import os, sys, time, threading, queue
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nvorbis_amd as nv
from nvorbis_amd import corpus
from nvorbis_amd.reader import Stream, demux_ogg_array, Context
from tests import c5_corpus
gp = os.environ.get("HOST_PARSE") is None
BF = int(os.environ.get("BF", "4096"))
files = c5_corpus.build_files(1.0)
arrays = [demux_ogg_array(f) for f in files]
frames = sum(len(pa) - 3 for pa in arrays)
print("files %d, packets %d, gpu_parse %s, batch_frames %d" % (len(files), frames, gp, BF), flush=True)
buf = torch.empty(1 << 28, dtype=torch.float32, device="cuda")  # 1 GiB scratch: every file's PCM lands at its start
order = sorted(range(len(files)), key=lambda i: -len(files[i]))
for T in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16]:
    q = queue.Queue()
    for i in order:
        q.put(i)
    stage = {"open": 0.0, "push": 0.0, "synth": 0.0, "close": 0.0}
    lock = threading.Lock()
    def work():
        ctx = Context(0)
        loc = dict.fromkeys(stage, 0.0)
        while True:
            try:
                i = q.get_nowait()
            except queue.Empty:
                break
            pa = arrays[i]
            t0 = time.perf_counter()
            st = Stream(ctx, pa[0], pa[1], pa[2])
            if gp:
                st.set_gpu_parse(True)
            t1 = time.perf_counter(); loc["open"] += t1 - t0
            nxt = 3
            while True:
                t0 = time.perf_counter()
                if nxt < len(pa) and not st.position()[2]:
                    nxt += st.push_packets(pa, nxt, BF)
                    last = nxt >= len(pa) or st.position()[2]
                else:
                    last = True
                if last and not st.position()[2]:
                    st.push_end()
                t1 = time.perf_counter(); loc["push"] += t1 - t0
                if st.pending()[0]:
                    st.synth_device(buf.data_ptr(), buf.numel())
                t2 = time.perf_counter(); loc["synth"] += t2 - t1
                if last:
                    break
            t0 = time.perf_counter()
            st.close()
            loc["close"] += time.perf_counter() - t0
        ctx.close()
        with lock:
            for k in stage:
                stage[k] += loc[k]
    th = [threading.Thread(target=work) for _ in range(T)]
    torch.cuda.synchronize()
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("threads %2d: %.3f s, %.2f M frames/s; thread-seconds: %s" % (T, dt, frames / dt / 1e6, {k: round(v, 2) for k, v in stage.items()}), flush=True)
