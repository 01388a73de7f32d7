"""CPU tests of the host slab writer (nvorbis_amd/csrc/host_slab.cpp): the parser's output in the form the synthesis kernels fetch.

The floor half is checked against the oracle's Floor1.Apply (oracle/orc_floor.c) bin by bin: the slab's segments are walked here
with the kernel's own arithmetic (a 64-bit (value : fraction) register stepped by the segment's signed 32.32 increment,
kernels_synth.hip: floor_walk_fx) and must reproduce inverse_dB_table[y(x)] of every bin.  The residue half is checked for
structure (every vector write of the packet exactly once, a chain's records in stage order, entry offsets inside the frame) --
its arithmetic lives on the GPU and is covered by the -m gpu parity tests.
"""
import ctypes as C

import numpy as np
import pytest

FLAG_SWEEP_COUPLES, FLAG_MG1, FLAG_COUPLE_PASS, FLAG_FAULT, FLAG_SLOT, FLAG_FUSE = 1, 2, 4, 8, 16, 32


def parse_slab(words):
    """NvhSlabHdr (nvh_format.h) + sections of one slab given as uint32 words."""
    h = {}
    w0, w1, w2, w3, w4, w5, frame, cpl = (int(x) for x in words[:8])
    h["n"] = w0 & 0xFFFF
    h["exec_mask"] = (w0 >> 16) & 0xFF
    h["flags"] = w0 >> 24
    h["nheads"], h["nrec"] = w1 & 0xFFFF, w1 >> 16
    h["off_heads"], h["off_rec"] = w2 & 0xFFFF, w2 >> 16
    h["off_ent"], h["vecs"] = w3 & 0xFFFF, w3 >> 16
    h["lpc"], h["rgeom"], h["group"] = w4 & 0xFFFF, (w4 >> 16) & 0xFF, w4 >> 24
    h["lpc_magic"], h["frame"], h["coupling"] = w5, frame, cpl
    h["chan"] = [int(x) for x in words[8:16]]
    return h


def walk_floor(words, cw, half):
    """Curve value y of every bin of [0, half) from the slab's segment list + per-four-bins table: floor_walk_fx's arithmetic."""
    ns, oseg = (cw >> 8) & 0xFF, cw >> 16
    segs = words[oseg * 4:(oseg + ns) * 4].reshape(ns, 4)
    tab = words[(oseg + ns) * 4:].view(np.uint8)
    M64 = (1 << 64) - 1
    ys = np.zeros(half, np.int64)
    for x0 in range(0, half, 4):
        sg = int(tab[x0 >> 2])
        s = [int(v) for v in segs[sg]]
        step = (s[3] << 32) | s[2]
        t = x0 - (s[0] & 0xFFFF)
        neg = 0xFFFFFFFF if (s[3] >> 31) else 0
        st = ((((s[1] << 32) | neg) + step * t) & M64)
        xend = s[0] >> 16
        for q in range(4):
            if x0 + q >= xend:
                sg += 1
                s = [int(v) for v in segs[sg]]
                step = (s[3] << 32) | s[2]
                neg = 0xFFFFFFFF if (s[3] >> 31) else 0
                st = (s[1] << 32) | neg
                xend = s[0] >> 16
            y = st >> 32
            if y >= 1 << 31:
                y -= 1 << 32
            ys[x0 + q] = y
            st = (st + step) & M64
    return ys


def oracle_curve(oracle, d, floor_index, n, posts, post_count, b1):
    v = np.ones(b1, np.float32)
    p = np.zeros(64, np.int32)
    p[:len(posts)] = posts
    rc = oracle.L.orc_floor1_apply_posts(d, floor_index, n, p.ctypes.data, post_count, v.ctypes.data, b1)
    return rc, v[:n // 2]


def raw_floor_posts(S, pkt):
    """Raw Floor1 Y values of every channel of an audio packet (None: unused floor), read with the specification decoder's bit
    reader and Floor1 packet decode (tests/vorbis_spec.py, 7.2.3) -- a parser that is neither the product's nor the oracle's."""
    from tests import vorbis_spec as vs
    r = vs.BitReader(pkt)
    assert r.read(1) == 0
    mode = r.read(S.mode_bits)
    long_block, mapping_idx = S.modes[mode]
    if long_block:
        r.read(2)
    m = S.mappings[mapping_idx]
    out = []
    for c in range(S.channels):
        fl = S.floors[m.submap_floor[m.mux[c]]]
        out.append((m.submap_floor[m.mux[c]], fl.decode(r, S.books)))
    return out


def check_frame_floors(oracle, d, S, stream, pkt, slab, db, b1, tag):
    """The floor sections of one frame's slab against the oracle's Floor1.Apply of the same raw posts; returns curves checked."""
    h = parse_slab(slab)
    n = h["n"]
    checked = 0
    for cc, (fi, ys) in enumerate(raw_floor_posts(S, pkt)):
        cw = h["chan"][cc]
        mode = cw & 0xFF
        executes = (h["exec_mask"] >> cc) & 1
        if ys is None:
            assert mode == (2 if executes else 0), tag  # Floor1.cs:218-221: an executing channel without posts is cleared
            continue
        if not executes:
            assert mode == 0, tag
            continue
        assert mode == 1, tag
        rc, curve = oracle_curve(oracle, d, fi, n, ys, len(ys), b1)
        if rc != 0:  # the reference indexes inverse_dB_table out of range (quirk B-7)
            assert h["flags"] & FLAG_FAULT, tag
            continue
        got = walk_floor(slab, cw, n // 2)
        assert got.min() >= 0 and got.max() <= 255, tag
        assert np.array_equal(db[got].view(np.uint32), curve.view(np.uint32)), tag
        # the per-four-bins table names the last segment that starts at or before the group's first bin
        ns, oseg = (cw >> 8) & 0xFF, cw >> 16
        xs = slab[oseg * 4:(oseg + ns) * 4].reshape(ns, 4)[:, 0] & 0xFFFF
        tab = slab[(oseg + ns) * 4:].view(np.uint8)[:n // 8]
        want_tab = np.searchsorted(xs, np.arange(0, n // 2, 4), side="right") - 1
        assert np.array_equal(tab, want_tab), tag
        checked += 1
    return checked


@pytest.mark.parametrize("name", ["1test", "2test", "3test", "issue6test"])
def test_slab_floors_match_oracle_on_shipped_files(oracle, ogg_bytes, name):
    """Audio packets of a shipped file: the slab's floor sections (Floor1.UnwrapPosts + the sorted walk done by the host parser's
    thread, stored as fixed-point line segments) reproduce the oracle's Floor1 curve of every executing channel bin for bin, and
    the channel modes (curve / clear / none) follow the execute flags."""
    import nvorbis_amd as nv
    from tests import vorbis_spec as vs
    data = ogg_bytes[name]
    pk, gr, fl = nv.demux_ogg(data)
    S = vs.Setup(pk[0], pk[2])
    s = nv.Stream(None, pk[0], pk[1], pk[2])
    err = C.c_int(0)
    d = oracle.L.orc_open_ogg(data, len(data), C.byref(err))
    assert d
    try:
        b1 = oracle.L.orc_block1(d)
        db = np.array([oracle.L.orc_inverse_db(i) for i in range(256)], np.float32)
        checked = 0
        for i in range(3, min(len(pk), 200)):
            s.drop_pending()
            s.push_packet(pk[i], -1, 0)
            if s.pending()[0] != 1 or int(s.pending_geometry()[-1][0]) == 0:
                continue
            words, first = s.pending_slabs()
            assert int(first[0]) == 0 and int(first[1]) * 4 == words.size
            h = parse_slab(words)
            assert h["vecs"] * 4 == words.size and h["n"] == int(s.pending_geometry()[-1][0])
            checked += check_frame_floors(oracle, d, S, s, pk[i], words, db, b1, (name, i))
        assert checked > (3 if name == "1test" else 20)
    finally:
        oracle.L.orc_close(d)
        s.close()


def test_slab_floors_random_posts_incl_faults(oracle, ogg_bytes):
    """Encoded packets with posts drawn over the floor's whole range (steep lines, values that leave inverse_dB_table): curve ==
    oracle where the reference draws one, NVH_SLAB_FLOOR_FAULT where it throws (quirk B-7)."""
    import nvorbis_amd as nv
    from tests import vorbis_encode as ve
    data = ogg_bytes["3test"]
    hdr = ve.shipped_headers(data)
    S = ve.setup_of(hdr)
    enc = ve.PacketEncoder(S)
    rng = np.random.default_rng(11)
    long_mode = next(i for i, (f, _) in enumerate(S.modes) if f)
    short_mode = next(i for i, (f, _) in enumerate(S.modes) if not f)
    s = nv.Stream(None, hdr[0], hdr[1], hdr[2])
    err = C.c_int(0)
    d = oracle.L.orc_open_ogg(data, len(data), C.byref(err))
    assert d
    try:
        b1 = oracle.L.orc_block1(d)
        db = np.array([oracle.L.orc_inverse_db(i) for i in range(256)], np.float32)
        checked = faults = 0
        for k in range(120):
            kw = [dict(y01=(0, 128), p_zero=0.2, geo=0.05), dict(y01=(100, 256), p_zero=0.3, geo=0.1), dict(y01=(16, 80), p_zero=0.7, geo=0.4)][k % 3]
            pkt = enc.packet(rng, long_mode if k % 4 else short_mode, floor_kw=kw, silent=(1,) if k % 17 == 5 else ())
            s.drop_pending()
            s.push_packet(pkt, -1, 0)
            assert s.pending()[0] == 1
            words, first = s.pending_slabs()
            faults += 1 if parse_slab(words)["flags"] & FLAG_FAULT else 0
            checked += check_frame_floors(oracle, d, S, s, pkt, words, db, b1, k)
        assert checked > 100  # (this setup cannot code a value outside the table: the fault branch is covered by the -m gpu B-7 tests)
    finally:
        oracle.L.orc_close(d)
        s.close()


def test_slab_residue_records_structure(ogg_bytes):
    """Chain-major records: every vector write the host parser recorded appears exactly once, the records of a chain (one
    partition / channel through the cascade stages) are consecutive with `more` on all but the last, entry offsets stay inside
    the frame's entry section, and the sections tile the slab without gaps."""
    import nvorbis_amd as nv
    data = ogg_bytes["3test"]
    pk, gr, fl = nv.demux_ogg(data)
    s = nv.Stream(None, pk[0], pk[1], pk[2])
    try:
        seen = 0
        for i in range(3, 120):
            s.drop_pending()
            s.push_packet(pk[i], -1, 0)
            if s.pending()[0] != 1 or int(s.pending_geometry()[-1][0]) == 0:
                continue
            words, _ = s.pending_slabs()
            h = parse_slab(words)
            assert h["off_heads"] <= h["off_rec"] <= h["off_ent"] <= h["vecs"]
            assert h["off_rec"] == h["off_heads"] + (h["nheads"] + 3) // 4 and h["off_ent"] == h["off_rec"] + (h["nrec"] + 1) // 2
            hw = words[h["off_heads"] * 4:h["off_rec"] * 4][:h["nheads"]]
            heads, xb = hw & 0xFFFF, hw >> 16
            recs = words[h["off_rec"] * 4:h["off_ent"] * 4][:2 * h["nrec"]].reshape(h["nrec"], 2)
            nent = (h["vecs"] - h["off_ent"]) * 8
            more = recs[:, 1] >> 31
            # a head is a record whose predecessor does not continue into it
            starts = np.flatnonzero(np.concatenate(([1], 1 - more[:-1].astype(np.int64)))) if h["nrec"] else np.zeros(0, np.int64)
            assert np.array_equal(starts, heads.astype(np.int64))
            if h["nrec"]:
                assert more[-1] == 0
                dims = (recs[:, 1] >> 20) & 31
                assert (dims >= 2).all() and ((dims & 1) == 0).all()
                assert (recs[:, 0] >> 16 == (65536 + dims - 1) // dims).all()
                if h["rgeom"] & 8:  # digit form: a record's run of partition_size bytes, offset in units of two bytes
                    psz = h["lpc"] * h["group"]
                    assert (((recs[:, 0] & 0xFFFF).astype(np.int64) * 2 + psz) <= nent * 2).all()
                    assert np.unique(recs[:, 0] & 0xFFFF).size == h["nrec"]  # runs do not overlap
                else:
                    assert ((recs[:, 0] & 0xFFFF) < nent).all()
                stage = (recs[:, 1] >> 28) & 7
                for a, b in zip(starts, np.append(starts[1:], h["nrec"])):
                    assert (np.diff(stage[a:b].astype(np.int64)) > 0).all()  # one chain: strictly rising cascade stages
                assert np.unique(xb).size == xb.size  # one chain per partition (Residue2: one channel); heads follow the op order
                seen += h["nrec"]
        assert seen > 500
    finally:
        s.close()


@pytest.mark.parametrize("name", ["res0_slab", "odd_dims_slab", "res2_alias_stereo", "two_pass_slab", "res0_3ch"])
def test_slab_general_group_list(oracle, name):
    """Frames outside the pair walk (Residue0, books of odd dimension, Residue2 over two channels with aliasing partitions, two
    residue passes) carry a group list (NvhSlabHdr::group == 1): per (pass, channel) -- per pass for a Residue2 -- the residue's
    geometry and the chain of every partition.  Checked here: the list tiles the slab's tail, every chain of the frame is some
    group's partition chain exactly once, a chain's head offset is the partition's first bin, partitions of one group share a
    bin only where the geometry says so (quirk B-1), passes are numbered in order."""
    import nvorbis_amd as nv
    from tests import synth_stream as ss
    pk, gr, fl = ss.filtered_stream(oracle, name, 60, 5)
    s = nv.Stream(None, pk[0], pk[1], pk[2])
    try:
        general = chains = 0
        for i in range(3, len(pk)):
            s.drop_pending()
            s.push_packet(pk[i], -1, 0)
            if s.pending()[0] != 1 or int(s.pending_geometry()[-1][0]) == 0:
                continue
            words, _ = s.pending_slabs()
            h = parse_slab(words)
            if h["nrec"] == 0:
                continue
            assert h["group"] == 1, (name, h["group"])
            general += 1
            hw = words[h["off_heads"] * 4:h["off_rec"] * 4][:h["nheads"]]
            xb = hw >> 16
            g = words[h["lpc"] * 4:h["vecs"] * 4]
            ng = int(g[0])
            assert 1 <= ng <= 16 and h["lpc"] > h["off_ent"]
            pchain = g[4 + 8 * ng:].view(np.uint16)
            used = []
            last_pass = 0
            for k in range(ng):
                rbegin, psz, nparts, cover, geom, magic, pco = (int(x) for x in g[4 + 8 * k:4 + 8 * k + 7])
                rtype, rch, pss, ch0 = geom & 15, (geom >> 4) & 15, (geom >> 8) & 15, (geom >> 12) & 15
                assert pss in (last_pass, last_pass + 1)
                last_pass = pss
                assert magic == (2 ** 32 + psz - 1) // psz and cover == (psz + rch - 1) // rch and ch0 + rch <= (3 if name == "res0_3ch" else 2)
                assert (rtype == 2) == (name in ("res2_alias_stereo",) or (name == "two_pass_slab" and rch == 2))
                pc = pchain[pco:pco + nparts]
                for p, cix in enumerate(pc):
                    if cix != 0xFFFF:
                        assert int(xb[cix]) == (rbegin + p * psz) // rch
                        used.append(int(cix))
            assert sorted(used) == list(range(h["nheads"]))  # every chain belongs to exactly one (group, partition)
            chains += h["nheads"]
            if name == "two_pass_slab":
                assert last_pass == 1
        assert general > 20 and chains > 200
    finally:
        s.close()


def _slab_residue_sums(words, h, lat, nch, half, vq=None):
    """The residue sums [nch][half] a slab's chains encode, added in the reference's order (cascade stage by stage, inside a
    stage the partitions in order; Residue0.cs:132-175), for every walk of the synthesis kernels: the pair walk (group 2 / 8 /
    2 * channels), the quirk-B-1 bin walk (group 0) and the general one (group 1, with its group list)."""
    spec = np.zeros((nch, half), np.float32)
    if h["nrec"] == 0:
        return spec
    hw = words[h["off_heads"] * 4:h["off_rec"] * 4][:h["nheads"]]
    first, xb = (hw & 0xFFFF).astype(np.int64), (hw >> 16).astype(np.int64)
    recs = words[h["off_rec"] * 4:][:2 * h["nrec"]].reshape(h["nrec"], 2)
    ent = words[h["off_ent"] * 4:].view(np.uint16)
    dig = bool(h["rgeom"] & 8)  # digit form (nvh_format.h: NVH_SLAB_RGEOM_DIGITS): one byte per component, 4 * digit
    dbytes = words[h["off_ent"] * 4:].view(np.uint8)
    # (pass, type, channels of a Residue2 interleave, partition size) of every chain
    geo = [None] * h["nheads"]
    if h["group"] == 1:
        g = words[h["lpc"] * 4:h["vecs"] * 4]
        ng = int(g[0])
        pchain = g[4 + 8 * ng:].view(np.uint16)
        for k in range(ng):
            rbegin, psz, nparts, cover, geom, magic, pco = (int(x) for x in g[4 + 8 * k:4 + 8 * k + 7])
            for cix in pchain[pco:pco + nparts]:
                if cix != 0xFFFF:
                    geo[int(cix)] = ((geom >> 8) & 15, geom & 15, (geom >> 4) & 15, psz)
    else:
        rtype, rch = h["rgeom"] & 7, h["rgeom"] >> 4
        psz = int(words[h["lpc"] * 4 + 1]) if h["group"] == 0 else h["lpc"] * h["group"]
        geo = [(0, rtype, rch if rtype == 2 else 1, psz)] * h["nheads"]
    writes = []  # (pass, stage, first bin, chain, record)
    for cix in range(h["nheads"]):
        o = int(first[cix])
        while True:
            x, y = int(recs[o, 0]), int(recs[o, 1])
            writes.append((geo[cix][0], (y >> 28) & 7, int(xb[cix]), cix, x, y))
            if not (y >> 31):
                break
            o += 1
    writes.sort(key=lambda w: (w[0], w[1], w[2]))
    for pss, stage, x0, cix, x, y in writes:
        _, rtype, rch, psz = geo[cix]
        dims, lv, lat_off, chan = (y >> 20) & 31, (y >> 12) & 0xFF, y & 0xFFF, (y >> 25) & 7
        eb = ent[(x & 0xFFFF):]
        db = dbytes[2 * (x & 0xFFFF):]
        steps = psz // dims
        span = psz if rtype == 0 else -(-psz // dims) * dims  # (Residue1 / 2: whole entries, the last may run over the partition's end)
        for q in range(span):
            j, comp = (q % steps, q // steps) if rtype == 0 else (q // dims, q % dims)
            if dig:
                b4 = int(db[j * dims + comp])
                assert b4 % 4 == 0 and b4 // 4 <= lv
                if b4 // 4 == lv:  # the book's +0.0f slot: no vector was added here
                    assert lat[lat_off + lv] == 0
                    continue
                v = lat[lat_off + b4 // 4: lat_off + b4 // 4 + 1].view(np.float32)[0]
            else:
                e = int(eb[j])
                if e == 0xFFFF:
                    continue
                if lv == 0:  # a book with an explicit table: the lattice-pool word is its offset in the VQ pool
                    v = vq[int(lat[lat_off]) + e * dims + comp]
                else:
                    v = lat[lat_off + (e // lv ** comp) % lv: lat_off + (e // lv ** comp) % lv + 1].view(np.float32)[0]
            c, b = (q % rch, x0 + q // rch) if (rtype == 2 and rch > 1) else (chan, x0 + q)
            if b < half:
                spec[c, b] = np.float32(spec[c, b] + v)
    return spec


@pytest.mark.parametrize("name", ["3test", "2test", "stereo_res1_coupled", "six_ch_res2_4096", "three_ch_res2_misaligned", "res0_slab",
                                  "odd_dims_slab", "res2_alias_stereo", "two_pass_slab", "res0_3ch", "floor0_slab",
                                  "table_books_pair", "table_books_general", "table_books_b1", "equal_blocks_overrun"])
def test_slab_residue_sums_match_oracle(oracle, ogg_bytes, name):
    """The residue half of the host-written slabs against the oracle's IResidue.Decode (oracle/orc_residue.c), without a GPU: the
    chains, records and entries of a frame, walked here in the reference's order of additions, must give the oracle's residue
    vectors bit for bit -- for the pair walk (shipped files, coupled stereo, six channels), the quirk-B-1 bin walk and the general
    walk (Residue0, odd dimensions, aliasing stereo Residue2, two passes, three channels)."""
    import nvorbis_amd as nv
    from tests import synth_stream as ss
    from tests.test_gpu_parity import _open_headers
    if name in ogg_bytes:
        pk, _, _ = nv.demux_ogg(ogg_bytes[name])
        ids = list(range(3, 40)) + list(range(40, len(pk), max(1, (len(pk) - 40) // 12)))
    else:
        pk, _, _ = ss.filtered_stream(oracle, name, 40, 9)
        ids = range(3, len(pk))
    d = _open_headers(oracle, pk)
    s = nv.Stream(None, pk[0], pk[1], pk[2])
    try:
        lat = s.lattice_pool()
        vq = s.vq_pool()
        nch, b1 = s.channels, s.block1
        scratch = np.zeros(nch * b1, np.float32)
        frames = vectors = 0
        for i in ids:
            a, b, c, e = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            if oracle.L.orc_decode_packet_block(d, pk[i], len(pk[i]), scratch.ctypes.data, C.byref(a), C.byref(b), C.byref(c), C.byref(e)) != 1:
                continue
            s.drop_pending()
            s.push_packet(pk[i], -1, 0)
            if s.pending()[0] != 1 or int(s.pending_geometry()[-1][0]) == 0:
                continue
            words, _ = s.pending_slabs()
            h = parse_slab(words)
            n = e.value
            assert h["n"] == n
            pos, idx, anyx = np.zeros(16, np.int32), np.zeros(16, np.int32), C.c_int()
            ncall = oracle.L.orc_last_residue_calls(d, pos.ctypes.data, idx.ctypes.data, 16, C.byref(anyx))
            ref = np.zeros(nch * b1, np.float32)
            for k in range(ncall):
                bits = C.c_int()
                assert oracle.L.orc_residue_decode_at(d, int(idx[k]), pk[i], len(pk[i]), int(pos[k]), anyx.value, n, ref.ctypes.data,
                                                      C.byref(bits)) == 0
            got = _slab_residue_sums(words, h, lat, nch, n // 2, vq)
            want = ref.reshape(nch, b1)[:, :n // 2]
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (name, i, float(np.abs(got - want).max()))
            frames += 1
            vectors += h["nrec"]
        assert frames >= 10 and vectors > 100, (frames, vectors)
    finally:
        oracle.L.orc_close(d)
        s.close()
