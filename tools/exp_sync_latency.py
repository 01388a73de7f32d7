import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nvorbis_amd as nv
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
data = open(os.path.join(root, "tests", "golden", "3test.ogg"), "rb").read()
pa = nv.demux_ogg_array(data)
ctx = nv.Context(0)
st = nv.Stream(ctx, pa[0], pa[1], pa[2]); nxt = 3
while nxt < len(pa) and not st.position()[2]:
    nxt += st.push_packets(pa, nxt, 100000)
b = st.upload_batch()
pcm = torch.empty(b.samples * 2, dtype=torch.float32, device="cuda")
for _ in range(20): b.synth(pcm.data_ptr(), pcm.numel()); ctx.synchronize()
N = 300
t0 = time.perf_counter()
for _ in range(N): b.synth(pcm.data_ptr(), pcm.numel())
t1 = time.perf_counter(); ctx.synchronize(); t2 = time.perf_counter()
print("enqueue %.1f us per synth; drain %.1f us total" % ((t1 - t0) / N * 1e6, (t2 - t1) * 1e6))
t0 = time.perf_counter()
for _ in range(N): b.synth(pcm.data_ptr(), pcm.numel()); ctx.synchronize()
t1 = time.perf_counter()
print("synth + sync: %.1f us" % ((t1 - t0) / N * 1e6))
host = np.empty(b.samples * 2, np.float32)
t0 = time.perf_counter()
for _ in range(50): pcm_h = pcm.cpu()
t1 = time.perf_counter()
print("D2H %.1f MB pageable via torch: %.1f us" % (pcm.numel() * 4 / 1e6, (t1 - t0) / 50 * 1e6))
print("kernels:", b.kernels(), "frames", b.frames)
tot, km = b.time(pcm.data_ptr(), pcm.numel(), 20)
print("per-slot ms:", [round(x, 4) for x in km])
