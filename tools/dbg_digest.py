"""Which of bench.py's resident batches reproduce the oracle's digest: alone (one synth, synchronised) and after the rotating loop."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import nvorbis_amd as nv
import bench
want = json.load(open(os.path.join(ROOT, "tests/golden/bench_pcm_digests.json")))["digests"]
headers, ll, ch = bench.ll_packets(nv, os.path.join(ROOT, "tests/golden/3test.ogg"))
insts = []
for k in range(3):
    ctx = nv.Context(0)
    st, bs = bench.make_batches(nv, torch, ctx, headers, ll, ch, bench.FRAMES, 4, seed_off=13 * k)
    insts.append((ctx, st, bs, 13 * k))
def hashes(tag):
    torch.cuda.synchronize()
    for ctx, st, bs, seed in insts:
        for j, (b, p) in enumerate(bs):
            a = p.cpu().numpy()
            ok = hashlib.sha256(a.tobytes()).hexdigest() == want["seed%d" % seed][j]
            print(tag, "seed", seed, "batch", j, "ok" if ok else "MISMATCH", flush=True)
            if not ok and os.environ.get("DUMP"):
                np.save("gpurun_out/dbg_%s_%d_%d.npy" % (tag, seed, j), a)
for ctx, st, bs, seed in insts:
    for b, p in bs:
        b.synth(p.data_ptr(), p.numel())
        ctx.synchronize()
hashes("alone")
order = [insts[k][2][j] for j in range(4) for k in range(3)]
for rep in range(3):
    for i in range(600):
        b, p = order[i % len(order)]
        b.synth(p.data_ptr(), p.numel())
    for ctx, _, _, _ in insts:
        ctx.synchronize()
    hashes("loop%d" % rep)
