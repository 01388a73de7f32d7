"""Damaged containers: a flipped CRC byte, a dropped page, a page out of sequence, junk between pages,
regressing granule positions.  What the reference's page reader does with them (Ogg/PageReaderBase.cs:227-292 byte-wise resync;
Ogg/StreamPageReader.cs:44-91 granule sanity + "sequence jump counts as a resync"; Ogg/PacketProvider.cs:324-438 packet
assembly) is restated twice, independently -- oracle/orc_ogg.c and the product's host_ogg.cpp: both must deliver the same
packets, granule positions and end-of-stream / resync flags, and the decoders behind them the same PCM."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import ogg_py, vorbis_encode as ve


def _oracle_demux(oracle, data, forward_only=False):
    L = oracle.L
    fn = L.orc_ogg_demux_forward if forward_only else L.orc_ogg_demux
    fn.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                   C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    b, o, g, f, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()
    rc = fn(data, len(data), C.byref(b), C.byref(o), C.byref(g), C.byref(f), C.byref(n))
    if rc != 0:
        return rc, None
    cnt = n.value
    offs = np.ctypeslib.as_array(C.cast(o, C.POINTER(C.c_int64)), shape=(cnt + 1,)).copy()
    gran = np.ctypeslib.as_array(C.cast(g, C.POINTER(C.c_int64)), shape=(max(cnt, 1),)).copy()[:cnt]
    flags = np.ctypeslib.as_array(C.cast(f, C.POINTER(C.c_uint8)), shape=(max(cnt, 1),)).copy()[:cnt]
    raw = C.string_at(b, int(offs[-1])) if cnt else b""
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for ptr in (b, o, g, f):
        libc.free(ptr)
    return 0, ([raw[int(offs[i]):int(offs[i + 1])] for i in range(cnt)], gran.tolist(), flags.tolist())


def _product_demux(data, forward_only=False):
    import nvorbis_amd as nv
    from nvorbis_amd import native
    try:
        pk, gr, fl = nv.demux_ogg(data, forward_only)
    except native.NvhError as e:
        return e.code, None
    return 0, (pk, gr.tolist(), fl.tolist())


def _stream(ogg_bytes, seed=1, frames=90, page_packets=4):
    hdr = ve.shipped_headers(ogg_bytes["3test"])
    S = ve.setup_of(hdr)
    rng = np.random.default_rng(seed)
    kinds = ve.markov_kinds(rng, frames, 0.1, 0.3)
    kinds[:4] = True
    pool = ve.packet_pool(S, seed, per_kind=8)
    pk, gr = ve.stream_from_pool(S, hdr, pool, kinds, rng)
    data = ogg_py.write_ogg(pk, gr, page_packets=page_packets)
    return data, ogg_py.read_pages(data)


def _damage(data, pages, kind, rng):
    """Returns the damaged byte string."""
    k = int(rng.integers(3, len(pages) - 2))  # an audio page in the middle
    pg = pages[k]
    a, b = pg["offset"], pg["offset"] + pg["length"]
    if kind == "crc":  # one bit of the body: the page fails its CRC, the reader resyncs byte by byte to the next page
        buf = bytearray(data)
        buf[a + 27 + len(pg["segs"]) + int(rng.integers(0, max(1, pg["length"] - 27 - len(pg["segs"]))))] ^= 0x10
        return bytes(buf)
    if kind == "drop":  # the page is gone: the next one arrives with a sequence gap
        return data[:a] + data[b:]
    if kind == "junk":  # garbage between two pages: sync lost and found again
        return data[:a] + bytes(rng.integers(0, 256, int(rng.integers(1, 300))).astype(np.uint8)) + data[a:]
    if kind == "swap":  # two neighbouring pages exchanged: sequence numbers out of order, granule position regresses
        nb = pages[k + 1]
        return data[:a] + data[nb["offset"]:nb["offset"] + nb["length"]] + data[a:b] + data[nb["offset"] + nb["length"]:]
    if kind == "seq":  # a page whose sequence number jumps (rewritten with a valid CRC): counted as a resync
        seg, body = pg["segs"], pg["body"]
        new = ogg_py.make_page(pg["serial"], pg["seq"] + 5, pg["granule"], pg["flags"], seg, body)
        return data[:a] + new + data[b:]
    if kind == "truncate":
        return data[: a + pg["length"] // 2]
    raise KeyError(kind)


@pytest.mark.parametrize("kind", ["crc", "drop", "junk", "swap", "seq", "truncate"])
def test_damaged_pages_same_packets_and_flags(oracle, ogg_bytes, kind):
    rng = np.random.default_rng(hash(kind) & 0xFFFF)
    seen_resync = False
    for trial in range(6):
        data, pages = _stream(ogg_bytes, seed=trial + 1, page_packets=int(rng.integers(1, 6)))
        bad = _damage(data, pages, kind, rng)
        rc_o, got_o = _oracle_demux(oracle, bad)
        rc_p, got_p = _product_demux(bad)
        assert (rc_o == 0) == (rc_p == 0), (kind, trial, rc_o, rc_p)
        if rc_o != 0:
            assert kind == "swap"  # "Granule Position regressed?!" (Ogg/StreamPageReader.cs:59-63): both refuse the file
            assert rc_o == -1 and rc_p == -1
            continue
        assert got_o[0] == got_p[0], (kind, trial)
        assert got_o[1] == got_p[1] and got_o[2] == got_p[2], (kind, trial)
        seen_resync = seen_resync or any(f & 2 for f in got_p[2])
        # the oracle behind its own demux == the oracle fed the product's packets
        a, _ = oracle.decode_ogg(bad)
        b, _ = oracle.decode_packets(got_p[0], got_p[1], got_p[2])
        assert a.size == b.size and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    if kind in ("crc", "drop", "junk", "seq"):
        assert seen_resync, "the damage never produced a resync packet"


def test_undamaged_stream_has_no_resync(oracle, ogg_bytes):
    data, pages = _stream(ogg_bytes)
    rc, got = _product_demux(data)
    assert rc == 0 and not any(f & 2 for f in got[2]) and (got[2][-1] & 1)
    rc_o, got_o = _oracle_demux(oracle, data)
    assert rc_o == 0 and got_o == got


# ---- the reader for sources that cannot seek (Ogg/ForwardOnlyPageReader.cs, Ogg/ForwardOnlyPacketProvider.cs) ----

def test_forward_only_reader_on_intact_files(oracle, ogg_bytes):
    """ForwardOnlyPacketProvider on well-formed files: the same packet bytes as the seekable reader; the first packet is a resync
    packet (the beginning-of-stream page, Ogg/ForwardOnlyPacketProvider.cs:38-46); granule positions and the end-of-stream mark
    sit on the last packet that is complete on its page; product == restatement."""
    for name in ("1test", "2test", "3test", "issue6test"):
        data = ogg_bytes[name]
        rc, fwd = _product_demux(data, True)
        rc_o, fwd_o = _oracle_demux(oracle, data, True)
        assert rc == 0 and rc_o == 0 and fwd == fwd_o, name
        _, seek = _product_demux(data, False)
        assert fwd[0] == seek[0], name                 # same packets
        assert fwd[2][0] & 2 and not any(f & 2 for f in fwd[2][1:])
        # granule positions only differ around packets that span pages (3test.ogg: the setup header and three audio packets):
        # the forward-only provider gives such a packet none and stamps the packet before it with its page's value instead
        diff = [i for i in range(len(seek[1])) if seek[1][i] != fwd[1][i]]
        assert all(seek[1][i] == -1 or fwd[1][i] == -1 for i in diff), (name, diff[:5])
        assert len(diff) <= 40
        assert (fwd[2][-1] & 1) == (seek[2][-1] & 1)


@pytest.mark.parametrize("page_packets,max_segments", [(4, 255), (3, 9), (None, 5)])
def test_forward_only_reader_with_continued_packets(oracle, ogg_bytes, page_packets, max_segments):
    """Packets that continue over two and more pages: assembled, but without granule position or end-of-stream mark
    (Ogg/ForwardOnlyPacketProvider.cs:176-231); product == restatement, and the bytes are the seekable reader's."""
    hdr = ve.shipped_headers(ogg_bytes["3test"])
    S = ve.setup_of(hdr)
    rng = np.random.default_rng(4)
    kinds = ve.markov_kinds(rng, 200, 0.1, 0.3)
    kinds[:3] = True
    pool = ve.packet_pool(S, 2, per_kind=6)
    pk, gr = ve.stream_from_pool(S, hdr, pool, kinds, rng)
    data = ogg_py.write_ogg(pk, gr, page_packets=page_packets, max_segments=max_segments)
    rc, fwd = _product_demux(data, True)
    rc_o, fwd_o = _oracle_demux(oracle, data, True)
    assert rc == 0 and rc_o == 0 and fwd == fwd_o
    assert fwd[0] == [bytes(p) for p in pk]
    _, seek = _product_demux(data, False)
    if max_segments < 255:
        lost = [i for i in range(len(pk)) if seek[1][i] >= 0 and fwd[1][i] < 0]
        assert lost, "no continued packet carried a granule position in the seekable list"


@pytest.mark.parametrize("kind", ["crc", "drop", "junk", "seq", "truncate", "zero"])
def test_forward_only_reader_on_damaged_files(oracle, ogg_bytes, kind):
    """Damage, forward-only: sequence gaps and lost sync mark packets as resync; a page that begins with the tail of a lost packet
    has its packets cut from the start of the page data (the skipped tail's length is never added to the data offset,
    Ogg/ForwardOnlyPacketProvider.cs:147-165) -- the product cuts the same bytes; a zero-length packet is delivered (:270-284)."""
    rng = np.random.default_rng(hash(kind) & 0xFFF)
    seen_resync = seen_shift = seen_empty = False
    for trial in range(6):
        if kind == "zero":
            hdr = ve.shipped_headers(ogg_bytes["3test"])
            S = ve.setup_of(hdr)
            pool = ve.packet_pool(S, 3, per_kind=4)
            kinds = np.ones(40, dtype=bool)
            pk, gr = ve.stream_from_pool(S, hdr, pool, kinds, np.random.default_rng(trial))
            pk = list(pk)
            pk.insert(10 + trial, b"")  # an empty packet between two audio packets: lacing value 0
            gr = list(gr)
            gr.insert(10 + trial, gr[9 + trial])
            bad = ogg_py.write_ogg(pk, gr, page_packets=5)
        else:
            hdr = ve.shipped_headers(ogg_bytes["3test"])
            S = ve.setup_of(hdr)
            r2 = np.random.default_rng(trial + 1)
            kinds = ve.markov_kinds(r2, 90, 0.1, 0.3)
            kinds[:4] = True
            pool = ve.packet_pool(S, trial + 1, per_kind=8)
            pk, gr = ve.stream_from_pool(S, hdr, pool, kinds, r2)
            # small pages: packets continue across them, so that a lost page leaves a continuation page at a packet start
            data = ogg_py.write_ogg(pk, gr, page_packets=int(rng.integers(2, 5)), max_segments=int(rng.integers(3, 12)))
            bad = _damage(data, ogg_py.read_pages(data), kind, rng)
        rc_o, got_o = _oracle_demux(oracle, bad, True)
        rc_p, got_p = _product_demux(bad, True)
        assert rc_o == 0 and rc_p == 0
        assert got_o == got_p, (kind, trial)
        seen_resync = seen_resync or any(f & 2 for f in got_p[2][1:])
        seen_empty = seen_empty or any(len(p) == 0 for p in got_p[0])
        _, seek = _product_demux(bad, False)
        if seek is not None:
            whole = set(bytes(p) for p in pk)
            seen_shift = seen_shift or any(bytes(p) not in whole and len(p) > 0 for p in got_p[0][3:])
    if kind in ("crc", "drop", "junk", "seq"):
        assert seen_resync
    if kind == "zero":
        assert seen_empty
    if kind == "drop":
        assert seen_shift, "no trial produced a page whose packets are cut from the wrong offset"


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["crc", "drop", "junk", "seq"])
def test_damaged_files_decode_like_the_oracle(oracle, gpu_ctx, ogg_bytes, kind):
    """VorbisReader over damaged files: a resync packet makes the decoder pick up a new position (StreamDecoder.cs:481-484);
    the PCM equals the oracle's, bit for bit, with both packet parsers."""
    import nvorbis_amd as nv
    rng = np.random.default_rng(7)
    for trial in range(4):
        data, pages = _stream(ogg_bytes, seed=20 + trial, frames=300, page_packets=int(rng.integers(2, 9)))
        bad = _damage(data, pages, kind, rng)
        ref, info = oracle.decode_ogg(bad)
        for gpu_parse in (False, True):
            rd = nv.VorbisReader(bad, ctx=gpu_ctx, batch_frames=64, gpu_parse=gpu_parse)
            got = rd.read_all()
            rd.close()
            assert got.size == ref.size and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (kind, trial, gpu_parse)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["3test", "issue6test"])
def test_forward_only_reader_decodes_like_the_oracle(oracle, gpu_ctx, ogg_bytes, name):
    """VorbisReader(forward_only=True): the packets ForwardOnlyPacketProvider delivers (first packet a resync packet, continued
    packets without granule position) through the GPU path == the oracle's decoder fed the oracle's forward-only list; seeking is
    refused the way StreamDecoder.SeekTo refuses a provider that cannot seek (StreamDecoder.cs:565)."""
    import nvorbis_amd as nv
    data = ogg_bytes[name]
    rc, (pk, gr, fl) = _oracle_demux(oracle, data, True)
    assert rc == 0
    ref, _ = oracle.decode_packets(pk, gr, fl)
    rd = nv.VorbisReader(data, ctx=gpu_ctx, batch_frames=200, forward_only=True)
    try:
        got = rd.read_all()
        assert got.size == ref.size and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        with pytest.raises(RuntimeError):
            rd.SeekTo(1000)
    finally:
        rd.close()


def test_container_fuzz_valid_crc_mutations(oracle, ogg_bytes):
    """Pages rewritten with a VALID checksum but mutated headers (flags flipped, granule positions moved / set to -1, sequence numbers
    changed, lacing tables re-cut) and raw byte damage on top: whatever the three readers make of it -- seekable list, forward-only
    list, seek search -- the product and the oracle's restatements make the same of it (packets, granule positions, flags, error
    codes), and nothing crashes."""
    import nvorbis_amd as nv
    L = nv.lib()
    hdr = ve.shipped_headers(ogg_bytes["3test"])
    S = ve.setup_of(hdr)
    pool = ve.packet_pool(S, 9, per_kind=4)
    cases = agree = refused = seeks = 0
    for trial in range(int(os.environ.get("NVH_FUZZ_TRIALS", "60"))):
        rng = np.random.default_rng(1000 + trial)
        kinds = ve.markov_kinds(rng, int(rng.integers(20, 70)), 0.15, 0.3)
        kinds[:2] = True
        pk, gr = ve.stream_from_pool(S, hdr, pool, kinds, rng)
        data = ogg_py.write_ogg(pk, gr, serial=int(rng.integers(1, 1 << 30)), page_packets=int(rng.integers(1, 6)),
                                max_segments=int(rng.choice([255, 12, 5])))
        pages = ogg_py.read_pages(data)
        out = []
        for i, pg in enumerate(pages):
            flags, granule, seq, segs, body = pg["flags"], pg["granule"], pg["seq"], list(pg["segs"]), pg["body"]
            r = rng.random()
            if i >= 2 and r < 0.10:
                flags ^= int(rng.choice([1, 2, 4]))
            elif i >= 2 and r < 0.18:
                granule = int(rng.choice([-1, 0, granule + int(rng.integers(-2000, 2000)), granule + 448, granule - 448, granule + (1 << 33), granule + (1 << 32) - (1 << 7), (1 << 40) + 5]))
            elif i >= 2 and r < 0.24:
                seq += int(rng.integers(-2, 5))
            elif i >= 2 and r < 0.30 and len(segs) > 1:
                # re-cut the lacing table: merge two packets / terminate one early (the body stays)
                k = int(rng.integers(0, len(segs) - 1))
                if segs[k] < 255 and segs[k] + segs[k + 1] <= 255:
                    segs[k:k + 2] = [segs[k] + segs[k + 1]]
            elif i >= 2 and r < 0.33:
                continue  # page dropped
            out.append(ogg_py.make_page(pg["serial"], seq, granule, flags, segs, body))
        bad = bytearray(b"".join(out))
        if rng.random() < 0.4 and len(bad) > 200:
            for _ in range(int(rng.integers(1, 4))):
                bad[int(rng.integers(100, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        bad = bytes(bad)
        for fwd in (False, True):
            rc_o, got_o = _oracle_demux(oracle, bad, fwd)
            rc_p, got_p = _product_demux(bad, fwd)
            cases += 1
            assert rc_o == rc_p, (trial, fwd, rc_o, rc_p)
            if rc_o != 0:
                refused += 1
                continue
            assert got_o == got_p, (trial, fwd)
            agree += 1
        # the seek search on whatever page table the seekable reader builds
        rc_p, got_p = _product_demux(bad, False)
        if rc_p != 0 or len(got_p[0]) < 4:
            continue
        try:
            st = nv.Stream(None, got_p[0][0], got_p[0][1], got_p[0][2])
            d = oracle.open_ogg(bad)
        except Exception:
            continue
        buf = (C.c_uint8 * len(bad)).from_buffer_copy(bad)
        h = C.c_void_p()
        assert L.nvh_ogg_index_open(buf, len(bad), 0, C.byref(h)) == 0
        try:
            top = max([g for g in got_p[1] if g >= 0] + [1])
            for g in [0, 1, 129, top // 3, top // 2, top - 1, top, top + 1] + [int(x) for x in rng.integers(0, top + 2, 6)]:
                for pr in (0, 1):
                    a, b = C.c_int64(), C.c_int64()
                    rc = L.nvh_ogg_seek(h, st._h, int(g), pr, C.byref(a), C.byref(b))
                    orc = oracle.ogg_seek(bad, d, g, pr)
                    mine = (rc, a.value, b.value) if rc == 0 else (rc, 0, 0)
                    ref = orc if orc[0] == 0 else (orc[0], 0, 0)
                    assert mine == ref, (trial, g, pr, mine, ref)
                    seeks += 1
        finally:
            L.nvh_ogg_index_close(h)
            oracle.L.orc_close(d)
            st.close()
    assert agree > 60 and seeks > 400, (cases, agree, refused, seeks)
