"""Kernel timings of the other BASELINE.json configurations (parity-test cases, not bench.py lines):
C3 mixed short/long blocks (the shipped 3test.ogg as is), C4 six channels n=4096 (synthetic, Residue2 + coupling),
plus the remaining synthetic shapes.  The C2 / C3 / C4 lines marked "full depth" use packets written by the structured
encoder (tests/vorbis_encode.py: a classification for every partition, a VQ entry for every vector of every cascade stage);
the other synthetic shapes are random-bit packets, so those numbers describe the code paths, not an encoder's output."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nvorbis_amd as nv
from tests import synth_stream as ss
from tests import oracle_py
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
orc = oracle_py.load()
ctx = nv.Context(0)
ONLY = os.environ.get("NVH_BENCH_ONLY")  # run only the lines whose name contains this
def run(name, packets, target_frames=4096):
    if ONLY and ONLY not in name:
        return
    hdr, audio = packets[:3], packets[3:]
    st = nv.Stream(ctx, hdr[0], hdr[1], hdr[2])
    st.push_packet(audio[0], -1, 0); st.synth_host()
    k = 0
    while st.pending()[0] < target_frames:
        st.push_packet(audio[1 + k % (len(audio) - 1)], -1, 0); k += 1
    b = st.upload_batch()
    pcm = torch.empty(max(b.samples * st.channels, 1), dtype=torch.float32, device="cuda")
    tot, km = b.time(pcm.data_ptr(), pcm.numel(), 30)
    names = b.kernels()
    chf = b.frames * st.channels
    print("%-28s ch %d frames %5d samples/frame %6.0f : %7.1f us/batch  %6.2f M ch-frames/s  %s" % (
        name, st.channels, b.frames, b.samples / max(b.frames, 1), tot / 30 * 1e3, chf / (tot / 30 * 1e-3) / 1e6,
        {names[i]: round(km[i] * 1e3, 1) for i in range(4) if names[i] != "-"}), flush=True)
    b.free(); st.close()
data = open(os.path.join(root, "tests", "golden", "3test.ogg"), "rb").read()
pk, gr, fl = nv.demux_ogg(data)
# C3: the file's own packet order, tiled over a loop whose seam is window-consistent: it starts at a long block
# that declares long neighbours on both sides and ends right before another one
hs = nv.Stream(None, pk[0], pk[1], pk[2])
ll = []
for i in range(3, len(pk) - 1):
    before = hs.pending()[0]
    hs.push_packet(pk[i], -1, 0)
    if hs.pending()[0] == before + 1:
        g = hs.pending_geometry()[-1]
        if g[0] == 2048 and g[1] == 0 and g[2] == 1024 and g[3] == 2048:  # n, start, valid, total of a long/long/long block
            ll.append(i)
hs.close()
lo, hi = ll[0], ll[-1]
seg = pk[lo:hi]
nshort = 0
run("C3 3test.ogg mixed 256/2048 (%d-packet loop)" % len(seg), pk[:3] + [pk[lo]] + seg * 40)
from tests import vorbis_encode as ve
hdr3 = ve.shipped_headers(data)
S3 = ve.setup_of(hdr3)
rng = np.random.default_rng(7)
pool3 = ve.packet_pool(S3, 20260928, per_kind=256)
p, g = ve.stream_from_pool(S3, hdr3, pool3, np.ones(4200, dtype=bool), rng)
run("C2 G-rand full depth (stereo n=2048)", p)
p, g = ve.stream_from_pool(S3, hdr3, pool3, ve.markov_kinds(np.random.default_rng(7), 4700), rng)
run("C3 Markov full depth (256/2048)", p)
for ps in (48, 32):
    h4 = ve.c4_headers(hdr3, psize=ps)
    S4 = ve.setup_of(h4)
    pool4 = ve.packet_pool(S4, 100 + ps, per_kind=128, class_weights=[0] + [1] * 9)
    p, g = ve.stream_from_pool(S4, h4, pool4, np.ones(2100, dtype=bool), rng)
    run("C4 6ch n=4096 psize %d full depth" % ps, p, target_frames=2048)
for name in ("six_ch_res2_4096", "stereo_res1_coupled", "three_ch_res2_misaligned", "two_submaps", "mono_8192", "stereo_8192", "floor0_stereo", "floor0_slab", "mono_res0_small_blocks",
             "res0_slab", "odd_dims_slab", "res2_alias_stereo", "two_pass_slab", "res0_3ch"):
    p, g, f = ss.filtered_stream(orc, name, 300, 3, True)
    run(name, p)
