"""Stress: truncated / bit-flipped / emptied packets of the shipped files, many seeds, both parsers, vs the oracle.
Where the oracle throws, the GPU path must raise too; otherwise PCM must be identical bit for bit."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nvorbis_amd as nv
from tests import oracle_py
orc = oracle_py.load()
ctx = nv.Context(0)
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
trials = int(os.environ.get("TRIALS", "40"))
t0 = time.time(); ok = 0; thrown = 0
for name in ("1test", "2test", "3test", "issue6test"):
    data = open(os.path.join(root, "tests", "golden", name + ".ogg"), "rb").read()
    pk, gr, fl = nv.demux_ogg(data)
    gr, fl = gr.tolist(), fl.tolist()
    for trial in range(trials):
        rng = np.random.default_rng(1000 * trial + len(name))
        pk2, g2, f2 = list(pk[:3]), gr[:3], fl[:3]
        for i in range(3, len(pk)):
            p = bytearray(pk[i]); r = rng.random()
            if r < 0.10 and len(p) > 2: p = p[: int(rng.integers(0, len(p)))]
            elif r < 0.20 and len(p) > 0:
                j = int(rng.integers(0, len(p))); p[j] ^= 1 << int(rng.integers(0, 8))
            elif r < 0.23: p = bytearray()
            pk2.append(bytes(p)); g2.append(gr[i]); f2.append(fl[i])
        gp = bool(trial & 1)
        bf = int(rng.choice([3, 50, 1000]))
        try:
            ref, info = orc.decode_packets(pk2, g2, f2)
        except RuntimeError:
            try:
                nv.StreamDecoder(ctx, pk2, g2, f2, bf, gpu_parse=gp).Read(np.zeros(1 << 22, np.float32), 0, 1 << 22)
                raise AssertionError("oracle threw, GPU path did not: %s trial %d" % (name, trial))
            except nv.NvhError:
                thrown += 1
            continue
        dec = nv.StreamDecoder(ctx, pk2, g2, f2, batch_frames=bf, gpu_parse=gp)
        buf = np.zeros(ref.size + 4096, np.float32)
        n = dec.Read(buf, 0, buf.size - buf.size % dec.Channels)
        assert n == ref.size, (name, trial, n, ref.size)
        assert np.array_equal(buf[:n].view(np.uint32), ref.view(np.uint32)), (name, trial, gp, bf)
        dec.close(); ok += 1
print("fuzz: %d streams identical to the oracle, %d where both refuse; %.0f s" % (ok, thrown, time.time() - t0))
