"""Experiment: do the pipeline's kernels overlap when two independent batches run on two HIP streams?"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import nvorbis_amd as nv, bench
headers, ll, ch = bench.ll_packets(nv, os.path.join(bench.ROOT, "tests", "golden", "3test.ogg"))
def make(nframes):
    ctx = nv.Context(0); st = nv.Stream(ctx, *headers)
    st.push_packet(ll[0], -1, 0); st.synth_host()
    for i in range(nframes): st.push_packet(ll[(i+1) % len(ll)], -1, 0)
    b = st.upload_batch()
    pcm = torch.empty(b.samples*ch, dtype=torch.float32, device="cuda")
    return ctx, st, b, pcm
def run(objs, steps):
    for o in objs: o[0].synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for ctx, st, b, pcm in objs: b.synth(pcm.data_ptr(), pcm.numel())
    for o in objs: o[0].synchronize()
    return (time.perf_counter() - t0) / steps
one = [make(4096)]
run(one, 10); t1 = run(one, 100)
two = [make(2048), make(2048)]
run(two, 10); t2 = run(two, 100)
four = [make(1024) for _ in range(4)]
run(four, 10); t4 = run(four, 100)
print("1 stream x4096: %.1f us/step   2 streams x2048: %.1f us/step   4 streams x1024: %.1f us/step" % (t1*1e6, t2*1e6, t4*1e6))
