"""Where a cycle of the pipelined GPU-parser path goes on the host thread: push / synth_begin / synth_end per batch (ms)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import nvorbis_amd as nv, bench
headers, ll, ch = bench.ll_packets(nv, os.path.join(bench.ROOT, "tests", "golden", "3test.ogg"))
ctx = nv.Context(0)
N = int(os.environ.get("FRAMES", "32768"))
pk = [ll[(i + 1) % len(ll)] for i in range(N)]
offs = np.zeros(N + 1, np.int64); offs[1:] = np.cumsum([len(p) for p in pk])
pa = nv.PacketArray(np.frombuffer(b"".join(pk), np.uint8), offs, np.full(N, -1, np.int64), np.zeros(N, np.uint8))
st = nv.Stream(ctx, *headers)
st.set_gpu_parse(True)
st.push_packet(ll[0], -1, 0); st.synth_host()
rows = []
outstanding = 0
T0 = time.perf_counter()
R = 14
ts = []
for r in range(R):
    t0 = time.perf_counter(); st.push_packets(pa, 0, N)
    t1 = time.perf_counter(); st.synth_begin(); outstanding += 1
    t2 = time.perf_counter()
    if outstanding == 2:
        st.synth_end(); outstanding -= 1
    t3 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, t3 - t2)); ts.append(t3)
while outstanding:
    st.synth_end(); outstanding -= 1
T1 = time.perf_counter()
for r in rows:
    print("push %.2f  begin %.2f  end(prev) %.2f ms" % tuple(x * 1e3 for x in r))
c = (ts[-1] - ts[3]) / (R - 4)  # steady state: the first batches pay the one-time allocations
print("steady state per batch %.2f ms -> %.2f M frames/s, %.1f GB/s of PCM" % (c * 1e3, N / c / 1e6, N * 8192 / c / 1e9))
