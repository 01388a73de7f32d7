"""Phase stamps of the general spectrum kernel on the C4 stream (six channels, n = 4096, full-depth packets); profiling build:
   python -m nvorbis_amd.build --debug && NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so python tools/dbg_phase_c4.py"""
import os, sys, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nvorbis_amd as nv
from tests import vorbis_encode as ve
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
data = open(os.path.join(root, "tests", "golden", "3test.ogg"), "rb").read()
hdr3 = ve.shipped_headers(data)
h4 = ve.c4_headers(hdr3, psize=48)
S4 = ve.setup_of(h4)
rng = np.random.default_rng(7)
pool4 = ve.packet_pool(S4, 148, per_kind=128, class_weights=[0] + [1] * 9)
p, g = ve.stream_from_pool(S4, h4, pool4, np.ones(2100, dtype=bool), rng)
ctx = nv.Context(0)
st = nv.Stream(ctx, p[0], p[1], p[2])
audio = p[3:]
st.push_packet(audio[0], -1, 0); st.synth_host()
k = 0
while st.pending()[0] < 2048:
    st.push_packet(audio[1 + k % (len(audio) - 1)], -1, 0); k += 1
b = st.upload_batch(); print(b.stats())
pcm = torch.empty(b.samples * st.channels, dtype=torch.float32, device="cuda")
nf = b.frames
dbg = torch.zeros(nf * 24, dtype=torch.int64, device="cuda")
L = nv.lib(); L.nvh_debug_set_buffer.argtypes = [ctypes.c_void_p]
for _ in range(3): b.synth(pcm.data_ptr(), pcm.numel())
ctx.synchronize()
L.nvh_debug_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
b.synth(pcm.data_ptr(), pcm.numel()); ctx.synchronize()
L.nvh_debug_set_buffer(None)
print(b.kernels())
d = dbg.cpu().numpy().reshape(nf, 24).astype(np.float64)
names = ["frame record", "staging + pair records", "residue", "coupling", "floors (unwrap + multiply)", "inverse MDCT / write-out"]
for k in range(6):
    dt = d[:, k + 1] - d[:, k]
    print("%-28s mean %8.0f  p50 %8.0f  p90 %8.0f cycles" % (names[k], dt.mean(), np.median(dt), np.percentile(dt, 90)))
life = d[:, 6] - d[:, 0]
w0, w1 = d[:, 22], d[:, 23]
print("workgroup lifetime mean %.0f cycles; per-stage residue cycles %s ops %s" % (life.mean(), [round(float(d[:, 8 + s].mean())) for s in range(4)], [round(float(d[:, 16 + s].mean())) for s in range(4)]))
t = (w0 - w0.min()) / 100.0
hist, _ = np.histogram(t, bins=np.arange(0, t.max() + 10.0, 10.0))
print("workgroup starts per 10 us:", hist.tolist())
