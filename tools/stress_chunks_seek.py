"""Stress: the chunked decode (nvorbis_amd.corpus.plan_stream_chunks / decode_stream_chunk) and SeekTo on damaged
streams -- truncated / bit-flipped / emptied packets of the shipped files, random window-flag sequences of the synthetic
shapes -- against the serial decode of the same packets (itself checked against the oracle).  Whatever the damage, the
chunks must concatenate to the serial PCM and a seek must return the serial samples from its target on."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nvorbis_amd as nv
from nvorbis_amd import corpus
from tests import oracle_py, synth_stream as ss
orc = oracle_py.load()
ctx = nv.Context(0)
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
trials = int(os.environ.get("TRIALS", "25"))
t0 = time.time(); ok = skipped = seeks = 0


def check(tag, pk, gr, fl, rng, gp):
    global ok, skipped, seeks
    try:
        ref, info = orc.decode_packets(pk, gr, fl)
    except RuntimeError:
        skipped += 1
        return
    ch = info["channels"]
    world = int(rng.integers(2, 7))
    chunks = corpus.plan_stream_chunks(pk, gr, fl, world)
    parts = [corpus.decode_stream_chunk(ctx, pk, gr, fl, c, i == len(chunks) - 1, batch_frames=int(rng.choice([7, 64, 4096])), gpu_parse=gp)[0]
             for i, c in enumerate(chunks)]
    got = np.concatenate(parts) if parts else np.zeros(0, np.float32)
    assert got.size == ref.size, (tag, got.size, ref.size, [(c["first"], c["last"]) for c in chunks])
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (tag, world)
    ok += 1
    if ref.size < 4 * ch:
        return
    dec = nv.StreamDecoder(ctx, pk, gr, fl, batch_frames=int(rng.choice([5, 200])), gpu_parse=gp)
    try:
        gpos, state, end_pos = dec._granule_index()
        first = end_pos - ref.size // ch
        buf = np.empty(3000 * ch, np.float32)
        for t in rng.integers(first, end_pos, 4):
            t = int(t)
            try:
                dec.SeekTo(t)
            except RuntimeError:
                continue  # pre-roll packet does not decode: the reference throws there as well
            n = dec.Read(buf, 0, buf.size)
            want = ref[(t - first) * ch:(t - first) * ch + buf.size]
            # a seek restarts with ONE lead-in packet (as the reference does); where that packet's own overlap reached
            # into its tail the samples of the first block after the seek legitimately differ from the serial decode
            lead = int(np.searchsorted(gpos, t, side="left")) - 1
            if lead >= 0 and not (state[lead] & 2):
                continue
            # the two packets SeekTo reads are read against the position BEFORE the seek (ResetDecoder leaves _currentPosition
            # alone, StreamDecoder.cs:294-305): an end-of-stream trim among them is not the serial decode's
            if any(fl[i] & 1 for i in range(3 + max(lead, 0), min(3 + lead + 2, len(fl)))):
                continue
            assert n == want.size and np.array_equal(buf[:n].view(np.uint32), want.view(np.uint32)), (tag, t)
            seeks += 1
    finally:
        dec.close()


for name in ("1test", "2test", "3test", "issue6test"):
    data = open(os.path.join(root, "tests", "golden", name + ".ogg"), "rb").read()
    pk, gr, fl = nv.demux_ogg(data)
    gr, fl = gr.tolist(), fl.tolist()
    for trial in range(trials):
        rng = np.random.default_rng(7000 * trial + len(name))
        pk2, g2, f2 = list(pk[:3]), gr[:3], fl[:3]
        for i in range(3, len(pk)):
            p = bytearray(pk[i]); r = rng.random()
            if r < 0.05 and len(p) > 2: p = p[: int(rng.integers(0, len(p)))]
            elif r < 0.10 and len(p) > 0:
                j = int(rng.integers(0, len(p))); p[j] ^= 1 << int(rng.integers(0, 8))
            elif r < 0.12: p = bytearray()
            pk2.append(bytes(p)); g2.append(gr[i]); f2.append(fl[i])
        check((name, trial), pk2, g2, f2, rng, bool(trial & 1))
for name in ("stereo_res1_coupled", "three_ch_res2_misaligned", "two_submaps", "equal_blocks_overrun", "mono_res0_small_blocks"):
    for trial in range(max(2, trials // 3)):
        for consistent in (True, False):
            pk, gr, fl = ss.filtered_stream(orc, name, 120, 500 + trial, consistent_windows=consistent)
            check((name, trial, consistent), pk, list(gr), list(fl), np.random.default_rng(trial), False)
print("chunks/seek stress: %d streams whose chunked decode equals the serial one, %d seeks verified, %d streams the oracle refuses; %.0f s" % (
    ok, seeks, skipped, time.time() - t0))
