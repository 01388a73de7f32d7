"""Debug aid: where a synthetic config's GPU PCM first differs from the oracle's."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import nvorbis_amd as nv
from tests import oracle_py, synth_stream as ss
orc = oracle_py.load()
ctx = nv.Context(0)
for name in sys.argv[1:]:
    pk, gr, fl = ss.filtered_stream(orc, name, 150, 12, True)
    ref, info = orc.decode_packets(pk, gr, fl, clip=True)
    dec = nv.StreamDecoder(ctx, pk, gr, fl, batch_frames=1024)
    buf = np.zeros(ref.size + 4096, np.float32)
    n = dec.Read(buf, 0, buf.size - buf.size % dec.Channels)
    got = buf[:n]
    ch = dec.Channels
    dec.close()
    bad = np.flatnonzero(got.view(np.uint32)[:min(n, ref.size)] != ref.view(np.uint32)[:min(n, ref.size)])
    print(name, "sizes", n, ref.size, "mismatches", bad.size, "first", bad[:6], "channels of first", (bad[:6] % ch), "sample", bad[:6] // ch)
    if bad.size:
        i = bad[0]
        print("  got", got[i:i + 4], "ref", ref[i:i + 4])
