"""Stress of the slab synthesis kernels (parser-written slabs -> k_synth / k_synth8) against the CPU oracle: full-depth encoded streams
(tests/vorbis_encode.py) on the stereo 3test setup and the six-channel C4 setup (psize 48, and psize 32: the bin walk of quirk B-1),
plus the synthetic setups of the general bin walk (Residue0, odd dimensions, aliasing stereo Residue2, two passes) on random-bit packets;
random lengths, block kinds from a
Markov chain with random transition rates, random look-ahead batch sizes (1 ... 700 frames), clipping on / off, host and GPU
packet parser, plus the four shipped files with random batch sizes; every PCM must equal the oracle's bit for bit.
  python tools/stress_slab.py [seconds]"""
import os, sys, time
os.environ["NVH_EMIT_ALWAYS"] = "1"  # ... with paired emission whenever a batch has a steady-state frame (default: 7/8 of its frames)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import nvorbis_amd as nv
from tests import oracle_py, synth_stream as ss, vorbis_encode as ve
GENERAL = ["res0_slab", "odd_dims_slab", "res2_alias_stereo", "two_pass_slab", "res0_3ch", "two_submaps", "floor0_stereo"]
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
orc = oracle_py.load()
ctx = nv.Context(0)
rng = np.random.default_rng(20260928)
files = {n: open(os.path.join(root, "tests", "golden", n + ".ogg"), "rb").read() for n in ("1test", "2test", "3test", "issue6test")}
hdr3 = ve.shipped_headers(files["3test"])
S3 = ve.setup_of(hdr3)
pool3 = ve.packet_pool(S3, 11, per_kind=96)
h4 = ve.c4_headers(hdr3, psize=48)
S4 = ve.setup_of(h4)
pool4 = ve.packet_pool(S4, 12, per_kind=48, class_weights=[0] + [1] * 9)
h5 = ve.c4_headers(hdr3, psize=32)
S5 = ve.setup_of(h5)
pool5 = ve.packet_pool(S5, 13, per_kind=48, class_weights=[0] + [1] * 9)


def decode_gpu(pk, gr, fl, clip, bf, gpu_parse):
    dec = nv.StreamDecoder(ctx, pk, gr, fl, batch_frames=bf, gpu_parse=gpu_parse)
    dec.ClipSamples = clip
    chunks = []
    buf = np.zeros(1 << 21, np.float32)
    buf = buf[: buf.size - buf.size % dec.Channels]
    while True:
        n = dec.Read(buf, 0, buf.size)
        if n == 0:
            break
        chunks.append(buf[:n].copy())
    dec.close()
    return np.concatenate(chunks) if chunks else np.zeros(0, np.float32)


t0 = time.time()
runs = frames = bad = 0
while time.time() - t0 < budget:
    which = int(rng.integers(0, 5))
    if which == 4:  # the general bin walk: random-bit packets on the synthetic setups (tests/synth_stream.py), random seeds
        what = GENERAL[int(rng.integers(0, len(GENERAL)))]
        pk, gr, fl = ss.filtered_stream(orc, what, int(rng.integers(20, 200)), int(rng.integers(0, 1 << 30)), bool(rng.integers(0, 2)))
        gr, fl = list(gr), list(fl)
    elif which == 2:
        name = list(files)[int(rng.integers(0, 4))]
        pk, gr, fl = nv.demux_ogg(files[name])
        gr, fl = gr.tolist(), fl.tolist()
        what = name
    else:
        S, hdr, pool = (S3, hdr3, pool3) if which == 0 else ((S4, h4, pool4) if which == 1 else (S5, h5, pool5))
        nfr = int(rng.integers(8, 500 if which == 0 else 160))
        kinds = ve.markov_kinds(rng, nfr, float(rng.uniform(0.0, 0.3)), float(rng.uniform(0.05, 0.6)))
        pk, gr = ve.stream_from_pool(S, hdr, pool, kinds, rng)
        fl = [0] * len(pk)
        what = "stereo" if which == 0 else ("six_ch" if which == 1 else "six_ch_psize32")
    clip = bool(rng.integers(0, 2))
    bf = int(rng.integers(1, 700))
    gp = bool(rng.integers(0, 2)) and what != "floor0_stereo"  # (the GPU packet parser refuses Floor0 streams)
    ref, _ = orc.decode_packets(pk, gr, fl, clip=clip)
    got = decode_gpu(pk, gr, fl, clip, bf, gp)
    ok = got.size == ref.size and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    runs += 1
    frames += len(pk) - 3
    if not ok:
        bad += 1
        print("MISMATCH", what, "packets", len(pk), "clip", clip, "batch", bf, "gpu_parse", gp, got.size, ref.size, flush=True)
print("stress_slab: %d streams, %d packets, %d mismatches, %.0f s" % (runs, frames, bad, time.time() - t0))
sys.exit(1 if bad else 0)
