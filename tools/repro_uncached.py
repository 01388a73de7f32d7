"""The round-4 puzzle, reproduced: with the work planes in hipDeviceMallocUncached memory (NVH_UNCACHED_PLANES=1) the shipped files
decode bit-exactly one at a time, and wrong after other decodes went through the same context (the order the parity suite
has).  Prints, per decode, how many samples differ from the oracle's and what they look like.
  NVH_UNCACHED_PLANES=1 python tools/repro_uncached.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import nvorbis_amd as nv
from tests import oracle_py
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
orc = oracle_py.load()
ctx = nv.Context(0)
files = {n: open(os.path.join(root, "tests", "golden", n + ".ogg"), "rb").read() for n in ("1test", "2test", "3test", "issue6test")}
refs = {n: orc.decode_ogg(d)[0] for n, d in files.items()}
seq = [(n, 7) for n in files] + [(n, 1024) for n in files] + [(n, 1024) for n in files]
if len(sys.argv) > 1:
    seq = [(a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[1:]]
for name, bf in seq:
    ref = refs[name]
    rd = nv.VorbisReader(files[name], ctx=ctx, batch_frames=bf)
    got = rd.read_all()
    ch = rd.Channels
    rd.close()
    bad = np.flatnonzero(got.view(np.uint32) != ref.view(np.uint32))
    print("%-10s batch %4d: %d of %d samples differ; NaN %d inf %d" % (name, bf, bad.size, ref.size, int(np.isnan(got).sum()), int(np.isinf(got).sum())))
    if bad.size:
        t = bad // ch
        runs = np.split(t, np.flatnonzero(np.diff(t) > 1) + 1)
        print("  runs of differing sample times (start, length), first 10 of %d:" % len(runs), [(int(r[0]), int(r[-1]) - int(r[0]) + 1) for r in runs[:10]])
        k = bad[:6]
        print("  first bad: idx", k.tolist(), "gpu", got[k].tolist(), "ref", ref[k].tolist())
