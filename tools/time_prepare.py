import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import nvorbis_amd as nv, bench
ctx = nv.Context(0)
headers, audio, ch = bench.ll_packets(nv, "tests/golden/3test.ogg")
out = []
for r in range(4):
    st, bl = bench.make_batches(nv, torch, ctx, headers, audio, 2, 4096, 1)
    b, pcm = bl[0]
    out.append(b.stats()["prepare_ns"] / 1e3)
    b.free(); st.close()
print(os.environ.get("NVH_LIB", "cur"), "prepare us", out)
