"""Per-packet phase timestamps of k_parse_slab (kernels_parse.hip) on the bench workload: where a lane's cycles go.
  NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so python tools/dbg_phase_parse.py [frames]   (needs python -m nvorbis_amd.build --debug)"""
import os, sys, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import nvorbis_amd as nv
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
headers, ll, ch = bench.ll_packets(nv, os.path.join(bench.ROOT, "tests", "golden", "3test.ogg"))
ctx = nv.Context(0)
pk = [ll[(i + 1) % len(ll)] for i in range(N)]
if os.environ.get("CORPUS"):  # packets of a C5 corpus file instead of 3test's own
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
for _ in range(2):
    st.push_packets(pa, 0, N); st.synth_host()
dbg = torch.zeros(N * 24, dtype=torch.int64, device="cuda")
L = nv.lib(); L.nvh_debug_set_buffer.argtypes = [ctypes.c_void_p]
L.nvh_debug_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
st.push_packets(pa, 0, N); b = st.upload_batch()
ctx.synchronize(); torch.cuda.synchronize()
L.nvh_debug_set_buffer(None)
d = dbg.cpu().numpy().reshape(N, 24)
names = ["frame record + packet into LDS + bit reader", "floors (Floor1.Unpack x channels)", "residue (classes + entries + records)",
         "slab tail: heads, partition table, entries copy", "slab floors (wavefront) + header"]
for k in range(5):
    dt = d[:, k + 1] - d[:, k]
    print("%-48s mean %8.0f  p50 %8.0f  p90 %8.0f cycles" % (names[k], dt.mean(), np.median(dt), np.percentile(dt, 90)))
for nm, k in (("  inside residue: entry loops of the vectors", 8), ("  inside residue: class words", 9),
              ("  inside residue: cursor steps between vectors (incl. class words)", 10), ("  vectors (rounds of the cursor walk)", 11),
              ("  inside residue (k_parse_slab_f): entries to memory + record", 21)):
    print("%-48s mean %8.0f  p50 %8.0f  p90 %8.0f cycles" % (nm, d[:, k].mean(), np.median(d[:, k]), np.percentile(d[:, k], 90)))
if os.environ.get("NVH_PARSE_CUR", "2") == "2" and int(os.environ.get("NVH_PARSE_LANES", "1")) > 1:
    d[:, 5] = d[:, 3]  # the split form: the parse kernel's stamps end with the residue
if d[:, 20].any() or os.environ.get("NVH_PARSE_CUR", "2") == "2":
    print("k_parse_slab_f left to the general body: %d of %d packets; by test:" % ((d[:, 20] != 0).sum(), N), dict(zip(*np.unique(d[:, 20], return_counts=True))))
life = d[:, 5] - d[:, 0]
print("lane lifetime mean %.0f p50 %.0f p90 %.0f cycles" % (life.mean(), np.median(life), np.percentile(life, 90)))
b.free(); st.close()
