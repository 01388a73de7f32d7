"""k_prepare_slabs time per upload (nvh_batch_stats[7]) the way bench.py creates its instances (torch stream handed to the
context) and on a context's own stream."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import nvorbis_amd as nv, bench
headers, audio, ch = bench.ll_packets(nv, "tests/golden/3test.ogg")
for mode in ("torch stream", "own stream", "torch stream"):
    ctx = nv.Context(0)
    if mode == "torch stream":
        ts = torch.cuda.Stream()
        ctx.set_hip_stream(ts.cuda_stream)
    st, bl = bench.make_batches(nv, torch, ctx, headers, audio, 2, 4096, 4)
    print(mode, [round(b.stats()["prepare_ns"] / 1e3, 1) for b, p in bl], flush=True)
