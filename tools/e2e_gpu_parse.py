"""End-to-end rate with the packets parsed on the GPU (nvh_stream_set_gpu_parse) vs on the host, one host thread."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nvorbis_amd as nv, bench
headers, ll, ch = bench.ll_packets(nv, os.path.join(bench.ROOT, "tests", "golden", "3test.ogg"))
ctx = nv.Context(0)
N = int(os.environ.get("FRAMES", "4096"))
pk = [ll[(i + 1) % len(ll)] for i in range(N)]
offs = np.zeros(N + 1, np.int64); offs[1:] = np.cumsum([len(p) for p in pk])
pa = nv.PacketArray(np.frombuffer(b"".join(pk), np.uint8), offs, np.full(N, -1, np.int64), np.zeros(N, np.uint8))
res = {}
for mode in ("host", "gpu", "gpu+pinned"):
    st = nv.Stream(ctx, *headers)
    st.set_gpu_parse(mode != "host")
    st.push_packet(ll[0], -1, 0); st.synth_host()
    best = None
    for rep in range(6):
        t0 = time.perf_counter()
        took = st.push_packets(pa, 0, N); assert took == N
        t1 = time.perf_counter()
        pcm = st.synth_host(pinned=mode.endswith("pinned"))
        t2 = time.perf_counter()
        if best is None or (t2 - t0) < sum(best): best = (t1 - t0, t2 - t1)
    res[mode] = pcm.copy()
    print("%-10s parse: host side %.2f ms, synth (upload + [k_parse] + kernels + D2H) %.2f ms -> %.0f k frames/s end to end" % (
        mode, best[0] * 1e3, best[1] * 1e3, N / sum(best) / 1e3), flush=True)
    st.close()
print("PCM identical:", bool((res["host"] == res["gpu"]).all() and (res["host"] == res["gpu+pinned"]).all()), res["host"].size)
# pipelined read-back (nvh_stream_synth_begin / _end): the transfer of batch i overlaps push + parse + kernels of batch i + 1
st = nv.Stream(ctx, *headers)
st.set_gpu_parse(True)
st.push_packet(ll[0], -1, 0); st.synth_host()
ROUNDS = 12
for rep in range(3):
    t0 = time.perf_counter()
    outstanding = 0
    got = 0
    for r in range(ROUNDS):
        took = st.push_packets(pa, 0, N); assert took == N
        st.synth_begin(); outstanding += 1
        if outstanding == 2:
            pcm = st.synth_end(); outstanding -= 1; got += 1
    while outstanding:
        pcm = st.synth_end(); outstanding -= 1; got += 1
    t1 = time.perf_counter()
# (bit-exactness of this path: tests/test_gpu_parity.py::test_pipelined_read_back_is_the_same_pcm)
print("gpu+pinned, pipelined (two batches outstanding): %.2f ms per batch -> %.0f k frames/s end to end" % (
    (t1 - t0) / ROUNDS * 1e3, N * ROUNDS / (t1 - t0) / 1e3), flush=True)
st.close()
