"""Experiment: S independent decoder instances (own nvh_ctx / HIP stream / batch), steps issued round-robin."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import nvorbis_amd as nv, bench
headers, ll, ch = bench.ll_packets(nv, os.path.join(bench.ROOT, "tests", "golden", "3test.ogg"))
FR = int(os.environ.get("FRAMES", "4096"))
def make(k):
    ts = torch.cuda.Stream()
    ctx = nv.Context(0); ctx.set_hip_stream(ts.cuda_stream)
    st = nv.Stream(ctx, *headers)
    st.push_packet(ll[k % len(ll)], -1, 0); st.synth_host()
    for i in range(FR): st.push_packet(ll[(i + k + 1) % len(ll)], -1, 0)
    b = st.upload_batch()
    pcm = torch.empty(b.samples * ch, dtype=torch.float32, device="cuda")
    return ts, ctx, st, b, pcm
for S in (1, 2, 3, 4, 6):
    inst = [make(k) for k in range(S)]
    steps = 1200
    for w in range(60):
        _, _, _, b, pcm = inst[w % S]; b.synth(pcm.data_ptr(), pcm.numel())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        _, _, _, b, pcm = inst[i % S]; b.synth(pcm.data_ptr(), pcm.numel())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("streams %d: %.1f us/step, %.2f M frames/s" % (S, dt / steps * 1e6, FR * steps / dt / 1e6), flush=True)
    for ts, ctx, st, b, pcm in inst:
        b.free(); st.close(); ctx.close()
