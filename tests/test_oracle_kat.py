"""Pins for the CPU oracle that need neither a GPU nor the (unrunnable, managed C#) reference.

The reference ships no golden vectors, so the oracle is pinned by: sample-count known answers derived
from the shipped TestFiles (granule positions), the closed-form IMDCT identity, window power
complementarity, TDAC perfect reconstruction with the reference's window / overlap geometry, and
integer cross-checks of the floor line renderer.
"""
import collections
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# final granule position of each file == number of samples per channel the decoder must emit
# (SURVEY App. C).  issue6test.ogg ends with an EOS page that carries a single zero-length packet:
# NVorbis' PageReader rejects a page without packets (Ogg/PageReader.cs:131) before the EOS flag is
# seen, so the last real packet is never EOS-trimmed (StreamDecoder.cs:429-437 needs isEndOfStream)
# and the provider running dry drains its tail (:352-356): 548223 + 961 = 549184 by the reference's
# own rules, not the file's granule.
EXPECTED_SAMPLES = {"1test": 17318, "2test": 315790, "3test": 288094, "issue6test": 549184}
EXPECTED_KINDS = {  # (start, valid, total) -> count   (issue6test without the empty packet)
    "1test": {(0, 128, 256): 8, (448, 1024, 2048): 1, (0, 1024, 2048): 16},
    "2test": {(0, 128, 256): 1, (448, 1024, 2048): 1, (0, 1024, 2048): 308},
    "3test": {(0, 128, 256): 95, (448, 1024, 2048): 9, (0, 1024, 2048): 254, (0, 1472, 1600): 8},
    "issue6test": {(0, 128, 256): 79, (448, 1024, 2048): 16, (0, 1024, 2048): 495, (0, 1472, 1600): 15},
}


@pytest.mark.parametrize("name", list(EXPECTED_SAMPLES))
def test_testfile_sample_counts(oracle, ogg_bytes, name):
    pcm, info = oracle.decode_ogg(ogg_bytes[name], trace=True)
    ch = info["channels"]
    assert pcm.size == EXPECTED_SAMPLES[name] * ch
    assert np.isfinite(pcm).all()
    assert np.abs(pcm).max() <= np.float32(0.99999994)
    tr = info["trace"]
    kinds = collections.Counter((int(a), int(b), int(c)) for a, b, c, ok, _, _ in tr if ok)
    assert dict(kinds) == EXPECTED_KINDS[name]
    assert info["position"] >= EXPECTED_SAMPLES[name]


def test_read_chunking_is_irrelevant(oracle, ogg_bytes):
    """StreamDecoder.Read partial reads (StreamDecoder.cs:338-379): any chunking gives the same PCM."""
    a, _ = oracle.decode_ogg(ogg_bytes["3test"], chunk=4096)
    b, _ = oracle.decode_ogg(ogg_bytes["3test"], chunk=1)
    c, _ = oracle.decode_ogg(ogg_bytes["3test"], chunk=1000003)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(a.view(np.uint32), c.view(np.uint32))


def _closed_form_imdct(X, n):
    i = np.arange(n)[:, None]
    k = np.arange(n // 2)[None, :]
    return (np.cos(np.pi / (n / 2) * (i + 0.5 + n / 4) * (k + 0.5)) * X[None, :]).sum(1)


@pytest.mark.parametrize("n", [256, 512, 1024, 2048, 4096, 8192])
def test_imdct_closed_form_identity(oracle, n):
    """Mdct.CalcReverse == unnormalised IMDCT, scale 1, to fp32 accuracy."""
    rng = np.random.default_rng(n)
    X = rng.uniform(-1, 1, n // 2).astype(np.float32)
    got = oracle.mdct_reverse(X, n).astype(np.float64)
    want = _closed_form_imdct(X.astype(np.float64), n)
    assert np.abs(got - want).max() / np.abs(want).max() < 6e-7
    # structural symmetry of the output (used by the GPU kernels)
    assert np.array_equal(got[: n // 2], -got[: n // 2][::-1])
    assert np.array_equal(got[n // 2:], got[n // 2:][::-1])


@pytest.mark.parametrize("n", [64, 128])
def test_imdct_small_blocks_are_not_a_transform(oracle, n):
    """Quirk A.3 / B-10: for n < 256 the reference's stage loops over-count; mirrored, not fixed."""
    rng = np.random.default_rng(n)
    X = rng.uniform(-1, 1, n // 2).astype(np.float32)
    got = oracle.mdct_reverse(X, n).astype(np.float64)
    want = _closed_form_imdct(X.astype(np.float64), n)
    assert np.abs(got - want).max() / np.abs(want).max() > 0.1


@pytest.mark.parametrize("b0,b1", [(256, 2048), (512, 4096), (64, 8192), (1024, 1024)])
def test_window_identities(oracle, b0, b1):
    """Mode.CalcWindow (Mode.cs:69-100): power complementary slopes, exact 0/1 plateaus, mirror symmetry."""
    for prev in (b0, b1):
        for nxt in (b0, b1):
            w = oracle.window(prev, b1, nxt).astype(np.float64)
            left, right = prev // 2, nxt // 2
            lb = b1 // 4 - left // 2
            rb = b1 - b1 // 4 - right // 2
            assert (w[:lb] == 0).all() and (w[rb + right:] == 0).all()
            assert (w[lb + left:rb] == 1).all()
            sl = w[lb:lb + left]
            sr = w[rb:rb + right]
            assert np.abs(sl ** 2 + sl[::-1] ** 2 - 1).max() < 5e-7
            assert np.abs(sr ** 2 + sr[::-1] ** 2 - 1).max() < 5e-7
            if left == right:
                assert np.array_equal(sl, sr[::-1])
    s, v, t = oracle.overlap(b1, b1, b1)
    assert (s, v, t) == (0, b1 // 2, b1)
    s, v, t = oracle.overlap(b0, b1, b0)
    assert (s, v, t) == (b1 // 4 - b0 // 4, b1 // 4 * 3 - b0 // 4, b1 // 4 * 3 + b0 // 4)


def test_tdac_perfect_reconstruction(oracle):
    """Forward MDCT (numpy, double) -> oracle IMDCT + window + overlap-add with the CalcWindow / CalcOverlap
    geometry reconstructs the signal.  Block sequence exercises all four long windows."""
    b0, b1 = 256, 2048
    seq = "LLLSSSLLSLLL"
    rng = np.random.default_rng(5)
    sizes = [b1 if c == "L" else b0 for c in seq]
    # block centres: consecutive blocks overlap so that centre distance = n_prev/4 + n_cur/4
    centres = [b1 // 2]
    for i in range(1, len(sizes)):
        centres.append(centres[-1] + sizes[i - 1] // 4 + sizes[i] // 4)
    total_len = centres[-1] + b1
    x = rng.uniform(-0.5, 0.5, total_len)
    out = np.zeros(total_len)
    prev_tail = None
    emitted = []
    pos_check = None
    for i, n in enumerate(sizes):
        prev = sizes[i - 1] if i > 0 else n
        nxt = sizes[i + 1] if i + 1 < len(sizes) else n
        if n == b0:
            prev = nxt = b0
        w = oracle.window(min(prev, n) if n == b1 else b0, n, min(nxt, n) if n == b1 else b0).astype(np.float64)
        s, v, t = oracle.overlap(min(prev, n), n, min(nxt, n)) if n == b1 else (0, n // 2, n)
        seg = x[centres[i] - n // 2: centres[i] + n // 2]
        ii = np.arange(n)[:, None]
        kk = np.arange(n // 2)[None, :]
        X = (4.0 / n) * ((w * seg)[:, None] * np.cos(np.pi / (n / 2) * (ii + 0.5 + n / 4) * (kk + 0.5))).sum(0)
        y = oracle.mdct_reverse(X.astype(np.float32), n).astype(np.float64) * w
        if prev_tail is not None:
            y[s:s + prev_tail.size] += prev_tail
            emitted.append((centres[i] - n // 2 + s, y[s:v].copy()))
        prev_tail = y[v:t].copy()
    # emitted segments are contiguous and reproduce x
    pos = emitted[0][0]
    for p, seg in emitted:
        assert p == pos
        assert np.abs(seg - x[p:p + seg.size]).max() < 5e-6
        pos += seg.size


def test_floor1_line_closed_form_matches_incremental(oracle):
    """The closed form used on the GPU for Floor1.RenderLineMulti (Floor1.cs:316-341) equals the reference's
    incremental error-term loop for every x, including lines whose end point was clipped to n/2."""
    rng = np.random.default_rng(11)
    import ctypes as C
    for _ in range(3000):
        x0 = int(rng.integers(0, 1000))
        x1 = x0 + int(rng.integers(1, 600))
        y0, y1 = int(rng.integers(0, 256)), int(rng.integers(0, 256))
        v = np.ones(x1 + 1, np.float32)
        assert oracle.L.orc_render_line_multi(x0, y0, x1, y1, v.ctypes.data, v.size) == 0
        dy, adx = y1 - y0, x1 - x0
        ady, sy = abs(dy), (-1 if dy < 0 else 1)
        b = int(dy / adx)  # truncation toward zero
        ady -= abs(b) * adx
        t = np.arange(0, adx)
        y = y0 + b * t + sy * ((ady * t) // adx)
        want = np.array([oracle.L.orc_inverse_db(int(q)) for q in y], np.float32)
        assert np.array_equal(v[x0:x1], want)


def test_db_table_copies_agree(oracle):
    a = open(os.path.join(ROOT, "oracle", "floor1_db_table.inc")).read()
    b = open(os.path.join(ROOT, "nvorbis_amd", "csrc", "floor1_db_table.inc")).read()
    assert a == b
    vals = np.array([oracle.L.orc_inverse_db(i) for i in range(256)], np.float64)
    closed = np.exp((np.arange(256) - 255) * 0.546875 * 0.11512925)
    assert np.abs(vals / closed - 1).max() < 1e-7
    assert vals[255] == 1.0


@pytest.mark.parametrize("name", ["2test", "3test"])
def test_floor1_apply_flat_curve_and_empty(oracle, ogg_bytes, name):
    """Floor1.Apply in closed form (Floor1.cs:186-222): equal end posts and no used inner posts give one flat line and
    the trailing flat run, i.e. every bin times inverse_dB_table[y * multiplier]; a floor without posts clears."""
    import ctypes as C
    data = ogg_bytes[name]
    err = C.c_int(0)
    d = oracle.L.orc_open_ogg(data, len(data), C.byref(err))
    assert d
    try:
        b0, b1 = oracle.L.orc_block0(d), oracle.L.orc_block1(d)
        nfloors = oracle.L.orc_floor_info(d, 0, None, None, None)
        assert nfloors >= 1
        rng = np.random.default_rng(2)
        for fi in range(nfloors):
            t, pc, rg = C.c_int(), C.c_int(), C.c_int()
            oracle.L.orc_floor_info(d, fi, C.byref(t), C.byref(pc), C.byref(rg))
            assert t.value == 1 and 2 <= pc.value <= 64
            mult = {256: 1, 128: 2, 86: 3, 64: 4}[rg.value]  # Floor1.cs:74-76
            for n in (b0, b1):
                for y in (0, 1, rg.value // 2, rg.value - 1):
                    if y * mult > 255:
                        continue
                    posts = np.zeros(64, np.int32)
                    posts[:2] = y
                    x = rng.standard_normal(b1).astype(np.float32)
                    v = x.copy()
                    assert oracle.L.orc_floor1_apply_posts(d, fi, n, posts.ctypes.data, pc.value, v.ctypes.data, b1) == 0
                    want = x[:n // 2] * np.float32(oracle.L.orc_inverse_db(y * mult))
                    assert np.array_equal(v[:n // 2], want)
                    assert np.array_equal(v[n // 2:], x[n // 2:])
                v = rng.standard_normal(b1).astype(np.float32)
                assert oracle.L.orc_floor1_apply_posts(d, fi, n, None, 0, v.ctypes.data, b1) == 0
                assert not v[:n // 2].any()
    finally:
        oracle.L.orc_close(d)


def test_residue_call_trace_and_replay(oracle, ogg_bytes):
    """The oracle's record of Mapping.DecodePacket's IResidue.Decode calls: one call per submap, cursor inside the packet;
    a replay is deterministic, and a call with every channel marked do-not-decode reads nothing (Residue0.cs:125)."""
    import ctypes as C
    data = ogg_bytes["3test"]
    err = C.c_int(0)
    d = oracle.L.orc_open_ogg(data, len(data), C.byref(err))
    assert d
    try:
        ch, b1 = oracle.L.orc_channels(d), oracle.L.orc_block1(d)
        import nvorbis_amd as nv
        pk, _, _ = nv.demux_ogg(data)
        scratch = np.zeros(ch * b1, np.float32)
        seen = 0
        for i in range(3, 40):
            a, b, c, e = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            if oracle.L.orc_decode_packet_block(d, pk[i], len(pk[i]), scratch.ctypes.data, C.byref(a), C.byref(b), C.byref(c), C.byref(e)) != 1:
                continue
            pos, idx, anyx = np.zeros(16, np.int32), np.zeros(16, np.int32), C.c_int()
            n = oracle.L.orc_last_residue_calls(d, pos.ctypes.data, idx.ctypes.data, 16, C.byref(anyx))
            assert n == 1 and 0 < pos[0] < len(pk[i]) * 8
            zero = np.zeros(ch * b1, np.float32)
            bits = C.c_int()
            assert oracle.L.orc_residue_decode_at(d, int(idx[0]), pk[i], len(pk[i]), int(pos[0]), 1, e.value, zero.ctypes.data, C.byref(bits)) == 0
            if anyx.value:
                assert bits.value > 0 and zero.any()
                seen += 1
            again = np.zeros(ch * b1, np.float32)
            assert oracle.L.orc_residue_decode_at(d, int(idx[0]), pk[i], len(pk[i]), int(pos[0]), 1, e.value, again.ctypes.data, C.byref(bits)) == 0
            assert np.array_equal(zero, again)
            none = np.ones(ch * b1, np.float32)
            assert oracle.L.orc_residue_decode_at(d, int(idx[0]), pk[i], len(pk[i]), int(pos[0]), 0, e.value, none.ctypes.data, C.byref(bits)) == 0
            assert bits.value == 0 and (none == 1).all()
        assert seen > 20
    finally:
        oracle.L.orc_close(d)


def test_reference_pcm_digests_when_present(oracle, ogg_bytes):
    """Digests of the REFERENCE's own PCM (csharp/GoldenGenerator, run on a machine with .NET against the unmodified
    NVorbis project) pin the oracle when they have been generated and committed as tests/golden/<name>[.noclip].pcm.sha256.
    None can be generated in this image (no dotnet / mono / csc), which is why the oracle header says "parity unpinned"."""
    import glob
    import hashlib
    import os
    from tests.conftest import GOLDEN
    files = sorted(glob.glob(os.path.join(GOLDEN, "*.pcm.sha256")))
    if not files:
        pytest.skip("no reference digests committed (needs a .NET box: csharp/GoldenGenerator)")
    for path in files:
        name = os.path.basename(path).split(".")[0]
        digest, count, channels, rate, clip = open(path).read().split()
        pcm, info = oracle.decode_ogg(ogg_bytes[name], clip=clip.endswith("1"))
        assert (pcm.size, info["channels"], info["sample_rate"]) == (int(count), int(channels), int(rate)), path
        assert hashlib.sha256(pcm.astype("<f4").tobytes()).hexdigest() == digest, path


def test_oracle_pcm_is_frozen():
    """The oracle's PCM for the four shipped files (clipping on and off) equals the digests committed by
    tools/freeze_oracle.py: a change to oracle/ that moves one sample fails here, independently of the product."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("freeze_oracle", os.path.join(root, "tools", "freeze_oracle.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(mod.OUT))["digests"]
    assert mod.digests() == want


def test_c5_corpus_fixture_sample():
    """tests/golden/c5_digests_scale*.json (the oracle's PCM digest of every file of the C5 corpus, tools/corpus_c5.py):
    the corpus writer is deterministic and a sample of files decodes to the committed digests on this machine too."""
    from tests import c5_corpus, oracle_py
    ws = c5_corpus.writer_setup()
    for scale in c5_corpus.DIGEST_SCALES:
        d = c5_corpus.load_digests(scale)
        assert d is not None and len(d["digests"]) == 1004 and d["total_floats"] == sum(r[1] for r in d["digests"])
        for i in ((0, 17, 999, 1000, 1003) if scale < 1 else (3,)):
            f = c5_corpus.corpus_file(ws, i, scale)
            pcm, info = oracle_py.load().decode_ogg(f)
            assert [c5_corpus.file_digest(f), int(pcm.size), c5_corpus.pcm_digest(pcm), int(info["channels"])] == d["digests"][i], (scale, i)
