"""Duration of the GPU packet parse of one batch (k_parse_fetch + k_parse_slab + k_parse_links, as the host waits for it:
NVH_TIME_UPLOAD's "wait for k_parse") against the launch shape.  FRAMES=4096 NVH_PARSE_LANES=.. NVH_PARSE_WAVES=.. python tools/time_parse.py"""
import os, sys, time, subprocess, re
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
if os.environ.get("NVH_TIME_PARSE_CHILD"):
    import numpy as np, torch
    import nvorbis_amd as nv, bench
    headers, ll, ch = bench.ll_packets(nv, os.path.join(bench.ROOT, "tests", "golden", "3test.ogg"))
    ctx = nv.Context(0)
    N = int(os.environ.get("FRAMES", "4096"))
    pk = [ll[(i + 1) % len(ll)] for i in range(N)]
    if os.environ.get("CORPUS"):  # packets of a C5 corpus file (tests/vorbis_encode.corpus_file: full-depth writer packets) instead of 3test's own
        from tests import vorbis_encode as ve
        S = ve.setup_of(headers)
        pool = ve.packet_pool(S, 5, per_kind=64)
        cpk, _, _ = nv.demux_ogg(ve.corpus_file(S, list(headers), pool, 700, scale=1.0))
        cpk = cpk[3:]
        pk = [cpk[i % len(cpk)] for i in range(N)]
    offs = np.zeros(N + 1, np.int64); offs[1:] = np.cumsum([len(p) for p in pk])
    pa = nv.PacketArray(np.frombuffer(b"".join(pk), np.uint8), offs, np.full(N, -1, np.int64), np.zeros(N, np.uint8))
    st = nv.Stream(ctx, *headers)
    st.set_gpu_parse(True)
    st.push_packet(ll[0], -1, 0); st.synth_host()
    for rep in range(8):
        took = st.push_packets(pa, 0, N); assert took == N
        b = st.upload_batch()
        b.free()
    st.close()
    sys.exit(0)
env = dict(os.environ); env["NVH_TIME_PARSE_CHILD"] = "1"; env["NVH_TIME_UPLOAD"] = "1"
r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
w = [float(x) for x in re.findall(r"wait for k_parse ([0-9.]+) ms", r.stdout)]
if not w:
    print(r.stdout[-2000:])
print("FRAMES %s%s LANES %s WAVES %s: wait for k_parse min %.3f ms median %.3f ms (%d uploads)" % (
    os.environ.get("FRAMES", "4096"), " (corpus packets)" if os.environ.get("CORPUS") else "", os.environ.get("NVH_PARSE_LANES", "-"), os.environ.get("NVH_PARSE_WAVES", "-"),
    min(w[2:]) if len(w) > 2 else -1, sorted(w[2:])[len(w[2:]) // 2] if len(w) > 2 else -1, len(w)))
