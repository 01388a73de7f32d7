"""Per-workgroup phase timestamps of the spectrum kernel for any synthetic stream shape (tools/dbg_phase.py is the bench
workload's version):  python tools/dbg_phase_gen.py six_ch_res2_4096"""
import os, sys, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import nvorbis_amd as nv
from tests import synth_stream as ss, oracle_py
name = sys.argv[1] if len(sys.argv) > 1 else "six_ch_res2_4096"
orc = oracle_py.load()
pk, gr, fl = ss.filtered_stream(orc, name, 300, 3, True)
ctx = nv.Context(0); st = nv.Stream(ctx, pk[0], pk[1], pk[2])
audio = pk[3:]
if os.environ.get("NOCLIP"):
    st.set_clip(False)
st.push_packet(audio[0], -1, 0); st.synth_host()
k = 0
while st.pending()[0] < 4096:
    st.push_packet(audio[1 + k % (len(audio) - 1)], -1, 0); k += 1
b = st.upload_batch(); print(b.stats())
ch = st.channels
pcm = torch.empty(b.samples * ch, dtype=torch.float32, device="cuda")
dbg = torch.zeros(4096 * 24, dtype=torch.int64, device="cuda")
# needs the profiling build: python -m nvorbis_amd.build --debug; NVH_LIB=nvorbis_amd/libnvorbis_hip_dbg.so
L = nv.lib(); L.nvh_debug_set_buffer.argtypes = [ctypes.c_void_p]; L.nvh_debug_set_buffer(ctypes.c_void_p(dbg.data_ptr()))
for _ in range(3): b.synth(pcm.data_ptr(), pcm.numel())
ctx.synchronize(); torch.cuda.synchronize()
print(b.kernels())
d = dbg.cpu().numpy().reshape(4096, 24)
geo_n = d[:, 6] > 0
names = ["frame-load", "stage+zero+barrier", "residue", "coupling", "floor", "writeout"]
for k in range(6):
    dt = (d[:, k + 1] - d[:, k])[geo_n]
    print("%-22s mean %8.0f  p50 %8.0f  p90 %8.0f cycles" % (names[k], dt.mean(), np.median(dt), np.percentile(dt, 90)))
life = (d[:, 6] - d[:, 0])[geo_n]
print("WG lifetime mean %.0f p50 %.0f cycles (%.1f us at 2.1 GHz)" % (life.mean(), np.median(life), life.mean() / 2100.0))
print("residue per-stage mean cycles:", [round(float(d[:, 8 + k][d[:, 8 + k] > 0].mean())) if (d[:, 8 + k] > 0).any() else 0 for k in range(8)],
      "ops per stage:", [round(float(d[:, 16 + k].mean()), 1) for k in range(3)])
L.nvh_debug_set_buffer(None)
tot, km = b.time(pcm.data_ptr(), pcm.numel(), 20)
print({n: round(v * 1e3, 1) for n, v in zip(b.kernels(), km)})
