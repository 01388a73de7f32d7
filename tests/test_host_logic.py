"""CPU tests of the product's host side: C-ABI surface, table builders, packet parser geometry."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    """libnvorbis_hip.so loads and exports every function include/nvorbis_hip.h declares."""
    import nvorbis_amd as nv
    from nvorbis_amd import native
    hdr = open(os.path.join(ROOT, "include", "nvorbis_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(nvh_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    handle = C.CDLL(nv.lib_path())
    for name in sorted(declared):
        assert hasattr(handle, name), "missing export: " + name
    assert declared == set(native.SIGNATURES), declared ^ set(native.SIGNATURES)
    assert b"gfx950" in nv.lib().nvh_version()


def test_comm_entry_points_check_their_arguments():
    """The native gather's entry points (nvh_comm_*) refuse bad arguments before they touch RCCL or a device."""
    import nvorbis_amd as nv
    from nvorbis_amd import native
    L = nv.lib()
    h = C.c_void_p()
    ident = (C.c_uint8 * 128)()
    assert L.nvh_comm_create(None, ident, 0, 1, C.byref(h)) == native.ERR_ARGUMENT and not h.value
    assert L.nvh_comm_unique_id(None) == native.ERR_ARGUMENT
    assert L.nvh_comm_info(None, None, None) == native.ERR_ARGUMENT
    one = (C.c_int64 * 1)(0)
    assert L.nvh_comm_allgather_i64(None, one, 1, one) == native.ERR_ARGUMENT
    assert L.nvh_comm_gather_pcm(None, None, 0, None, one, 0, 0) == native.ERR_ARGUMENT
    L.nvh_comm_destroy(None)  # a no-op


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the compute entry points fail loudly (NVH_ERR_NO_GPU), they never fall back."""
    import nvorbis_amd as nv
    from nvorbis_amd import native
    L = nv.lib()
    if L.nvh_device_count() > 0:
        pytest.skip("a GPU is visible")
    h = C.c_void_p()
    assert L.nvh_ctx_create(0, C.byref(h)) == native.ERR_NO_GPU
    with pytest.raises(nv.NvhError):
        nv.Context(0)
    data = open(os.path.join(ROOT, "tests", "golden", "1test.ogg"), "rb").read()
    pk, gr, fl = nv.demux_ogg(data)
    s = nv.Stream(None, pk[0], pk[1], pk[2])
    s.push_packet(pk[3], gr[3], fl[3])
    with pytest.raises(nv.NvhError) as ei:
        s.synth_host()
    assert ei.value.code == native.ERR_NO_GPU
    with pytest.raises(nv.NvhError):
        s.upload_batch()


@pytest.mark.parametrize("n", [64, 128, 256, 512, 1024, 2048, 4096, 8192])
def test_mdct_tables_match_oracle(oracle, n):
    """Host table builder (Mdct.cs:30-63 typing rules) == oracle, bit for bit."""
    import nvorbis_amd as nv
    a, b = np.zeros(n // 2, np.float32), np.zeros(n // 2, np.float32)
    c, br = np.zeros(n // 4, np.float32), np.zeros(n // 8, np.uint16)
    assert nv.lib().nvh_mdct_tables(n, a.ctypes.data, b.ctypes.data, c.ctypes.data, br.ctypes.data) == 0
    oa, ob, oc, obr = oracle.mdct_tables(n)
    for x, y in ((a, oa), (b, ob), (c, oc)):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert np.array_equal(br, obr)


def test_windows_match_oracle(oracle):
    import nvorbis_amd as nv
    L = nv.lib()
    for b0, b1 in ((256, 2048), (64, 8192), (512, 512)):
        for prev in (b0, b1):
            for nxt in (b0, b1):
                w = np.zeros(b1, np.float32)
                assert L.nvh_calc_window(prev, b1, nxt, w.ctypes.data) == 0
                assert np.array_equal(w.view(np.uint32), oracle.window(prev, b1, nxt).view(np.uint32))
                s, v, t = C.c_int(), C.c_int(), C.c_int()
                L.nvh_calc_overlap(prev, b1, nxt, C.byref(s), C.byref(v), C.byref(t))
                assert (s.value, v.value, t.value) == oracle.overlap(prev, b1, nxt)


def _parse_all(nv, packets, granules, flags):
    s = nv.Stream(None, packets[0], packets[1], packets[2])
    err = None
    try:
        for i in range(3, len(packets)):
            s.push_packet(packets[i], granules[i], flags[i])
        s.push_end()
    except nv.NvhError as e:
        err = e.code
    geo = s.pending_geometry()
    _, smp = s.pending()
    pos, emitted, eos = s.position()
    s.close()
    return geo, smp, pos, err


@pytest.mark.parametrize("name", ["1test", "2test", "3test", "issue6test"])
def test_parser_geometry_matches_oracle(oracle, ogg_bytes, name):
    """Frame lengths / positions are bit-exact: host parser geometry == oracle trace on the TestFiles."""
    import nvorbis_amd as nv
    pk, gr, fl = nv.demux_ogg(ogg_bytes[name])
    geo, smp, pos, err = _parse_all(nv, pk, gr.tolist(), fl.tolist())
    assert err is None
    pcm, info = oracle.decode_ogg(ogg_bytes[name], trace=True)
    tr = info["trace"]
    ok = tr[tr[:, 3] == 1]
    assert geo.shape[0] == ok.shape[0]
    assert np.array_equal(geo[:, 0], ok[:, 4])  # block size
    assert np.array_equal(geo[:, 1], ok[:, 0])  # start
    assert np.array_equal(geo[:, 3], ok[:, 2])  # total
    assert smp * info["channels"] == pcm.size
    assert geo[:, 5].sum() == smp
    assert pos == info["position"]
    # first packet emits nothing (StreamDecoder.cs:446-450)
    assert geo[0, 5] == 0


def test_parser_fuzz_truncation_and_bitflips_match_oracle(oracle, ogg_bytes):
    """Packet-time corruption is silent in the reference: truncated / bit-flipped
    packets must produce the same number of emitted samples and the same error status in the host
    parser as in the oracle.  (The PCM itself is compared on the GPU in test_gpu_parity.py.)"""
    import nvorbis_amd as nv
    rng = np.random.default_rng(2024)
    pk, gr, fl = nv.demux_ogg(ogg_bytes["3test"])
    gr, fl = gr.tolist(), fl.tolist()
    for trial in range(12):
        pk2 = list(pk[:3])
        sel = sorted(rng.choice(np.arange(3, len(pk)), size=60, replace=False).tolist())
        g2, f2 = gr[:3], fl[:3]
        for i in sel:
            p = bytearray(pk[i])
            mode = rng.integers(0, 4)
            if mode == 0 and len(p) > 2:
                p = p[: int(rng.integers(0, len(p)))]
            elif mode == 1:
                for _ in range(int(rng.integers(1, 4))):
                    j = int(rng.integers(0, len(p)))
                    p[j] ^= 1 << int(rng.integers(0, 8))
            elif mode == 2:
                p = bytearray()
            pk2.append(bytes(p))
            g2.append(-1)
            f2.append(0)
        geo, smp, pos, err = _parse_all(nv, pk2, g2, f2)
        try:
            pcm, info = oracle.decode_packets(pk2, g2, f2, trace=True)
            oerr = None
        except RuntimeError:
            oerr = True
        if oerr:
            assert err is not None
            continue
        assert err is None, (trial, err)
        assert smp * info["channels"] == pcm.size, trial
        ok = info["trace"][info["trace"][:, 3] == 1]
        assert geo[geo[:, 0] != 0].shape[0] == ok.shape[0]


@pytest.mark.parametrize("name", __import__("tests.synth_stream", fromlist=["CONFIG_NAMES"]).CONFIG_NAMES)
def test_parser_geometry_synthetic_configs(oracle, name):
    """Floor0, Residue0, 3/6 channels, several submaps, 64..8192 blocks (no shipped file has them): the host
    parser emits the same frame geometry / sample count as the oracle on random-bit packets."""
    import nvorbis_amd as nv
    from tests import synth_stream as ss
    for consistent in (True, False):
        pk, gr, fl = ss.filtered_stream(oracle, name, 60, 7, consistent_windows=consistent)
        geo, smp, pos, err = _parse_all(nv, pk, gr, fl)
        pcm, info = oracle.decode_packets(pk, gr, fl, trace=True)
        assert err is None
        assert smp * info["channels"] == pcm.size
        ok = info["trace"][info["trace"][:, 3] == 1]
        dec = geo[geo[:, 0] != 0]
        assert dec.shape[0] == ok.shape[0]
        assert np.array_equal(dec[:, 0], ok[:, 4]) and np.array_equal(dec[:, 1], ok[:, 0]) and np.array_equal(dec[:, 3], ok[:, 2])


def test_push_packets_equals_per_packet_push(ogg_bytes):
    """nvh_stream_push_packets (one FFI call per batch) leaves the parser in the same state as the per-packet loop,
    including the stop at the end-of-stream packet."""
    import nvorbis_amd as nv
    for name in ("1test", "issue6test"):
        pa = nv.demux_ogg_array(ogg_bytes[name])
        a = nv.Stream(None, pa[0], pa[1], pa[2])
        b = nv.Stream(None, pa[0], pa[1], pa[2])
        for i in range(3, len(pa)):
            if a.position()[2]:
                break
            a.push_packet(pa[i], int(pa.granules[i]), int(pa.flags[i]))
        nxt = 3
        while nxt < len(pa) and not b.position()[2]:
            took = b.push_packets(pa, nxt, 37)
            assert 0 < took <= 37
            nxt += took
        assert a.pending() == b.pending()
        assert a.position() == b.position()
        assert (a.pending_geometry() == b.pending_geometry()).all()
        a.close()
        b.close()


@pytest.mark.parametrize("name", ["1test", "2test", "3test", "issue6test"])
def test_packet_index_and_sample_counts(oracle, ogg_bytes, name):
    """Host-only geometry: nvh_stream_packet_sample_count == Mode.GetPacketSampleCount of the oracle (valid - start of
    every decodable packet, 0 for the others, StreamDecoder.cs:630-647); nvh_stream_index_packets == the state of a
    stream the same packets were pushed into one by one; reset makes the next packet a first packet again."""
    import ctypes as C
    import nvorbis_amd as nv
    pa = nv.demux_ogg_array(ogg_bytes[name])
    st = nv.Stream(None, pa[0], pa[1], pa[2])
    hdr = [pa[0], pa[1], pa[2]]
    blob = np.frombuffer(b"".join(hdr), dtype=np.uint8)
    offs = np.zeros(4, np.int64)
    offs[1:] = np.cumsum([len(p) for p in hdr])
    g3, f3, err = np.full(3, -1, np.int64), np.zeros(3, np.uint8), C.c_int(0)
    d = oracle.L.orc_open_packets(blob.ctypes.data, offs.ctypes.data, g3.ctypes.data, f3.ctypes.data, 3, C.byref(err))
    assert d
    try:
        planes = np.zeros(st.channels * st.block1, np.float32)
        for i in range(3, len(pa)):
            a, b, c, e = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            rc = oracle.L.orc_decode_packet_block(d, pa[i], len(pa[i]), planes.ctypes.data, C.byref(a), C.byref(b), C.byref(c), C.byref(e))
            want = (b.value - a.value) if rc == 1 else 0
            assert st.packet_sample_count(pa[i]) == want, i
            assert st.packet_sample_count(pa[i], is_resync=True) == 0
        assert st.packet_sample_count(b"") == 0 and st.packet_sample_count(b"\x01") == 0
        pos, em, state, total = st.index_packets(pa, 3)
        for i in range(3, len(pa)):
            if st.position()[2]:
                break
            st.push_packet(pa[i], int(pa.granules[i]), int(pa.flags[i]))
            p, e, eos = st.position()
            assert (p, e) == (int(pos[i - 3]), int(em[i - 3])), i
            assert st.position_state()[0] == bool(state[i - 3] & 4) and eos == bool(state[i - 3] & 8)
        if not st.position()[2]:
            st.push_end()
        assert st.position()[1] == total
        st.reset()  # ResetDecoder
        assert st.position() == (0, 0, False) and st.pending() == (0, 0) and st.position_state() == (False, 0)
        st.push_packet(pa[3], -1, 0)
        assert st.position()[1] == 0  # a first packet emits nothing again
    finally:
        st.close()
        oracle.L.orc_close(d)


def _ogg_pages(x):
    out, pos = [], 0
    while pos < len(x):
        nseg = x[pos + 26]
        tot = 27 + nseg + sum(x[pos + 27:pos + 27 + nseg])
        out.append(x[pos:pos + tot])
        pos += tot
    return out


def test_demux_chained_and_multiplexed_streams(ogg_bytes):
    """Logical streams of one physical file (Ogg/PageReader.cs:126-158): chained (files one after the other) and
    multiplexed (pages interleaved) containers deliver, per stream, exactly the packets, granule positions and flags the
    stream delivers on its own; a serial number that comes back after its end-of-stream page opens a new stream; junk
    between pages marks the next packet as a resync point and nothing else."""
    import nvorbis_amd as nv
    names = ["2test", "3test", "1test"]
    alone = {n: nv.demux_ogg_array(ogg_bytes[n]) for n in names}

    def same(a, b):
        return (len(a) == len(b) and np.array_equal(a.data[:a.offsets[-1]], b.data[:b.offsets[-1]]) and
                np.array_equal(a.offsets, b.offsets) and np.array_equal(a.granules, b.granules) and np.array_equal(a.flags, b.flags))

    chained = b"".join(ogg_bytes[n] for n in names)
    assert nv.ogg_stream_count(chained) == 3
    for k, n in enumerate(names):
        assert same(nv.demux_ogg_array(chained, k), alone[n])
    pa, pb = _ogg_pages(ogg_bytes["2test"]), _ogg_pages(ogg_bytes["3test"])
    mux = b"".join((pa[i] if i < len(pa) else b"") + (pb[i] if i < len(pb) else b"") for i in range(max(len(pa), len(pb))))
    assert nv.ogg_stream_count(mux) == 2
    assert same(nv.demux_ogg_array(mux, 0), alone["2test"]) and same(nv.demux_ogg_array(mux, 1), alone["3test"])
    twice = ogg_bytes["1test"] + ogg_bytes["1test"]  # the same serial number again after its end-of-stream page
    assert nv.ogg_stream_count(twice) == 2
    assert same(nv.demux_ogg_array(twice, 0), alone["1test"]) and same(nv.demux_ogg_array(twice, 1), alone["1test"])
    assert nv.ogg_stream_count(b"") == 0 and len(nv.demux_ogg_array(b"")) == 0
    with pytest.raises(nv.NvhError):
        nv.demux_ogg_array(mux, 2)
    junk = ogg_bytes["2test"] + b"\\x00garbage between streams" + ogg_bytes["3test"]
    second = nv.demux_ogg_array(junk, 1)
    assert len(second) == len(alone["3test"]) and np.array_equal(second.granules, alone["3test"].granules)
    assert second.flags[0] & 2 and np.array_equal(second.flags[1:], alone["3test"].flags[1:])


def _c_params(sig):
    sig = sig.strip()
    if sig in ("", "void"):
        return []
    out, depth, cur = [], 0, ""
    for ch in sig:
        if ch == "(":
            depth += 1
        if ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out


def test_csharp_pinvoke_declarations_follow_the_header():
    """csharp/NativeMethods.cs cannot be compiled in this image; what can be checked is that every [DllImport] it declares names an
    entry point of include/nvorbis_hip.h with the same number of parameters, pointer parameters where the header has pointers, and
    64-bit integers where the header has int64_t / size_t -- and that the C# sources only call entry points that are declared."""
    hdr = open(os.path.join(ROOT, "include", "nvorbis_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|void|const char\s*\*)\s*(nvh_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.S):
        protos[m.group(1)] = _c_params(re.sub(r"\s+", " ", m.group(2)))
    cs = open(os.path.join(ROOT, "csharp", "NativeMethods.cs")).read()
    decls = {}
    for m in re.finditer(r"\[DllImport\(Lib\)\]\s*public static extern (?:unsafe )?\w+ (nvh_[a-z0-9_]+)\(([^;]*?)\);", cs, flags=re.S):
        decls[m.group(1)] = _c_params(re.sub(r"\s+", " ", m.group(2)))
    assert len(decls) >= 45
    for name, cs_params in decls.items():
        assert name in protos, "NativeMethods.cs declares %s, which the header does not" % name
        c_params = protos[name]
        assert len(cs_params) == len(c_params), (name, cs_params, c_params)
        for cp, sp in zip(c_params, cs_params):
            is_ptr = "*" in cp
            cs_ptr = "*" in sp or sp.startswith(("out ", "ref ", "IntPtr ", "[Out] ")) or " IntPtr " in " " + sp
            assert is_ptr == cs_ptr or (not is_ptr and sp.startswith("UIntPtr ")), (name, cp, sp)
            if re.match(r"(const )?int64_t [a-z_]+$", cp):
                assert sp.startswith("long "), (name, cp, sp)
            if re.match(r"size_t [a-z_]+$", cp):
                assert sp.startswith("UIntPtr "), (name, cp, sp)
            if re.match(r"(const )?int [a-z_]+$", cp):
                assert sp.startswith("int "), (name, cp, sp)
    # every native call in the C# sources is declared
    for fn in ("GpuStreamDecoder.cs", "GpuFactory.cs", "GpuCorpusGather.cs"):
        src = open(os.path.join(ROOT, "csharp", fn)).read()
        for called in set(re.findall(r"NativeMethods\.(nvh_[a-z0-9_]+)\(", src)):
            assert called in decls, (fn, called)


def _spec_books(setup_pkt):
    """The codebook section of a setup header parsed by the spec-derived decoder (tests/vorbis_spec.py: written from the
    Vorbis I specification, follows neither the C# reference nor oracle/)."""
    from tests import vorbis_spec as vs
    r = vs.BitReader(setup_pkt)
    assert bytes(r.read(8) for _ in range(7)) == b"\x05vorbis"
    return [vs.Codebook.parse(r) for _ in range(r.read(8) + 1)]


def _tables(info_fn, tables_fn, handle, b, ok):
    v = [C.c_int(0) for _ in range(7)]
    assert info_fn(handle, b, *[C.byref(x) for x in v]) == ok
    dims, entries, map_type, prefix_bits, max_bits, n_prefix, n_overflow = [x.value for x in v]
    lengths = np.zeros(entries, np.int32)
    lookup = np.zeros(max(entries * dims, 1), np.float32)
    prefix = np.zeros(max(n_prefix, 1) * 5, np.int32)
    overflow = np.zeros(max(n_overflow, 1) * 5, np.int32)
    assert tables_fn(handle, b, lengths.ctypes.data, lookup.ctypes.data, prefix.ctypes.data, overflow.ctypes.data) == ok
    return dict(dims=dims, entries=entries, map_type=map_type, prefix_bits=prefix_bits, max_bits=max_bits, n_prefix=n_prefix,
                n_overflow=n_overflow, lengths=lengths, lookup=lookup[:entries * dims if map_type else 0],
                prefix=prefix[:max(n_prefix, 0) * 5].reshape(-1, 5), overflow=overflow[:max(n_overflow, 0) * 5].reshape(-1, 5))


@pytest.mark.parametrize("source", ["1test", "2test", "3test", "issue6test", "synthetic", "synthetic_floor0"])
def test_codebook_tables_product_vs_oracle_vs_spec(oracle, ogg_bytes, source):
    """Codebook.Init / InitLookupTable / Huffman.GenerateTable (Codebook.cs:59-283, Huffman.cs:15-76) compared DIRECTLY, not
    through decoded PCM: for every codebook of the four shipped files and of the synthetic setups (lookup type 2 and
    sequence_p books included) the product's lengths, VQ lookup table (bit patterns), prefix table and overflow list
    (nvh_stream_codebook_tables) equal the oracle's, node for node in list order (first-match semantics); and both agree with
    the spec-derived decoder, which shares no code or reading with either: codeword assignment by spec 3.2.1 ("lowest valued
    available codeword", bit-reversed because the stream is LSb-first) and the VQ unpack of spec 3.2.1 / 3.3 in double."""
    import nvorbis_amd as nv
    from tests import synth_stream as ss, vorbis_spec as vs
    if source.startswith("synthetic"):
        name = "floor0_stereo" if source.endswith("floor0") else "three_ch_res2_misaligned"
        try:
            cfg = ss.config(name)
        except Exception:
            pytest.skip("no synthetic configuration called %s" % name)
        headers = ss.make_stream(cfg, 1, 1)[0][:3]
    else:
        headers = nv.demux_ogg(ogg_bytes[source])[0][:3]
    st = nv.Stream(None, headers[0], headers[1], headers[2])
    d = oracle.open_headers(headers)
    L, O = nv.lib(), oracle.L
    vp = C.c_void_p
    O.orc_codebook_info.argtypes = [vp, C.c_int] + [C.POINTER(C.c_int)] * 7
    O.orc_codebook_tables.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    O.orc_book_count.argtypes = [vp]
    try:
        spec = _spec_books(headers[2])
        assert O.orc_book_count(d) == len(spec)
        kinds = set()
        for b in range(len(spec)):
            p = _tables(L.nvh_stream_codebook_info, L.nvh_stream_codebook_tables, st._h, b, 0)
            o = _tables(O.orc_codebook_info, O.orc_codebook_tables, d, b, 0)
            for k in ("dims", "entries", "map_type", "prefix_bits", "max_bits", "n_prefix", "n_overflow"):
                assert p[k] == o[k], (b, k, p[k], o[k])
            assert np.array_equal(p["lengths"], o["lengths"]), b
            assert np.array_equal(p["lookup"].view(np.uint32), o["lookup"].view(np.uint32)), b
            # unoccupied prefix slots hold a null node in the reference: compare occupancy, then the occupied nodes
            assert np.array_equal(p["prefix"][:, 0], o["prefix"][:, 0]), b
            occ = p["prefix"][:, 0] != 0
            assert np.array_equal(p["prefix"][occ], o["prefix"][occ]), b
            assert np.array_equal(p["overflow"], o["overflow"]), b
            # ---- against the specification ----
            sb = spec[b]
            assert (p["dims"], p["entries"], p["map_type"]) == (sb.dims, sb.entries, sb.lookup_type), b
            assert np.array_equal(np.maximum(p["lengths"], 0), np.maximum(np.asarray(sb.lengths, np.int32), 0)), b
            want = {(e, w[1], vs.bitrev(w[0], w[1])) for e, w in enumerate(sb.words) if w is not None}
            nodes = np.concatenate([p["prefix"][occ], p["overflow"]]) if p["n_prefix"] else np.zeros((0, 5), np.int32)
            got = {(int(r[1]), int(r[2]), int(r[3])) for r in nodes if r[0]}
            if len(want) > 1:
                assert got == want, (b, sorted(got ^ want)[:6])
                # a prefix slot is filled by the code whose `length` low bits equal the slot's (Huffman.cs:40-61)
                for slot in np.nonzero(occ)[0]:
                    _, val, ln, bits, mask = p["prefix"][slot]
                    assert (int(slot) & ((1 << ln) - 1)) == bits and mask == (1 << ln) - 1, (b, slot)
            if sb.lookup_type:
                kinds.add((sb.lookup_type, int(bool(sb.sequence_p))))
                tab = p["lookup"].reshape(sb.entries, sb.dims).astype(np.float64)
                ref = np.array([sb.vector(e) for e in range(sb.entries)], np.float64)
                assert np.abs(tab - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), b
        if source == "synthetic":
            assert {(1, 0), (1, 1), (2, 0)} <= kinds, kinds
    finally:
        O.orc_close(d)
        st.close()


def test_pcm_upper_bound_covers_what_the_decoder_emits(oracle, ogg_bytes):
    """corpus._pcm_upper_bound (the early arena of decode_files_to_device): container fields alone bound the decoded length of
    every shipped file -- issue6test emits more samples than its last granule position says -- and of corpus writer files; input
    that is not an Ogg Vorbis file gives no bound (the arena then waits for the index pass)."""
    from nvorbis_amd import corpus
    from tests import c5_corpus
    files = [ogg_bytes[k] for k in ("1test", "2test", "3test", "issue6test")]
    ws = c5_corpus.writer_setup()
    files += [c5_corpus.corpus_file(ws, i, 0.02) for i in (0, 1, 2)]
    for data in files:
        pcm, _ = oracle.decode_ogg(data)
        bound = corpus._pcm_upper_bound([data])
        assert bound is not None and pcm.size <= bound <= pcm.size + 2 * 2 * 8192 * 8 + 4096, (len(data), pcm.size, bound)
    assert corpus._pcm_upper_bound(files) == sum(corpus._pcm_upper_bound([f]) for f in files)
    assert corpus._pcm_upper_bound([b"junk" * 100]) is None
    assert corpus._pcm_upper_bound([files[0], b""]) is None


def test_index_pass_totals_equal_the_oracle_lengths(oracle, ogg_bytes):
    """corpus._index_pass (the sizing pass of decode_files_to_device: lacing-only page walk, packet geometry through host-only streams
    that a thread keeps per distinct header triple): every file's float count equals the length of the oracle's decode, its channel
    count the identification header's, its (packets, payload bytes) what the checked demultiplex finds -- on one thread and on four
    (streams reused across files and threads), files in any order, and with a file whose first page is damaged in between (the
    index then takes the checked demultiplex's word, ogg page CRC and all)."""
    from nvorbis_amd import corpus
    from nvorbis_amd.reader import demux_ogg_array
    from tests import c5_corpus
    ws = c5_corpus.writer_setup()
    good = [ogg_bytes[k] for k in ("1test", "2test", "3test", "issue6test")] + [c5_corpus.corpus_file(ws, i, 0.02) for i in range(6)]
    files = good + good[::-1] + good[4:]  # the writer files share their headers: the per-thread stream cache is hit
    want = []
    for data in files:
        pcm, info = oracle.decode_ogg(data)
        pa = demux_ogg_array(data)
        want.append((pcm.size, info["channels"], (len(pa), int(pa.offsets[-1]))))
    for workers in (1, 4):
        shape, totals, chans, errors = corpus._index_pass(files, workers)
        assert not errors
        assert [(t, c, sh) for t, c, sh in zip(totals, chans, shape)] == want, workers
    shape_f, totals_f, chans_f, errors = corpus._index_pass(files, 3, full_index=True)  # the round-5 index: checksums + packet copies
    assert not errors and (shape_f, totals_f, chans_f) == (shape, totals, chans)
    # not an Ogg file: an error for that file, the others unaffected
    shape, totals, chans, errors = corpus._index_pass([good[0], b"junk" * 64, good[5]], 2)
    assert [i for i, _ in errors] == [1] and totals[0] == want[0][0] and totals[2] == want[5][0]


def test_demux_in_one_call_equals_the_sizing_and_fill_calls(ogg_bytes):
    """reader.demux_ogg_array hands the library buffers sized from the file and gets the packets in one call; the two-call form
    (sizing call, then fill call: what a caller without a bound uses, and the fallback) yields the same arrays."""
    import ctypes as C

    import numpy as np

    from nvorbis_amd import native
    from nvorbis_amd.reader import demux_ogg_array
    L = native.lib()
    for name in ("1test", "2test", "3test", "issue6test"):
        data = ogg_bytes[name]
        for fn, forward in ((L.nvh_ogg_demux_stream, False), (L.nvh_ogg_demux_forward, True)):
            pa = demux_ogg_array(data, 0, forward)
            src = np.frombuffer(data, dtype=np.uint8)
            n, total = C.c_int(0), C.c_int64(0)
            assert fn(C.c_void_p(src.ctypes.data), len(data), 0, None, 0, None, None, None, 0, C.byref(n), C.byref(total), None) == native.OK
            pk = np.zeros(max(total.value, 1), np.uint8)
            offs, gran, flags = np.zeros(n.value + 1, np.int64), np.zeros(max(n.value, 1), np.int64), np.zeros(max(n.value, 1), np.uint8)
            assert fn(C.c_void_p(src.ctypes.data), len(data), 0, pk.ctypes.data, pk.size, offs.ctypes.data, gran.ctypes.data,
                      flags.ctypes.data, n.value, C.byref(n), C.byref(total), None) == native.OK
            assert len(pa) == n.value and pa.data.size == pk.size
            assert np.array_equal(pa.data, pk) and np.array_equal(pa.offsets, offs)
            assert np.array_equal(pa.granules[:n.value], gran[:n.value]) and np.array_equal(pa.flags[:n.value], flags[:n.value])


def test_lacing_only_index_equals_the_checked_demultiplex(ogg_bytes):
    """reader.index_ogg_array (nvh_ogg_index_packets; Ogg/PageReaderBase.cs:227-292 page header + lacing values,
    Ogg/PacketProvider.cs:324-438 packet rules): without the page checksums and the packets' bodies it yields the packet list of
    reader.demux_ogg_array -- same count, granule positions, end-of-stream / resync flags, the three headers whole, every other
    packet's first 8 bytes, the true payload size -- and Stream.index_packets over it the same geometry; a page with a damaged
    body passes the index and is found by the comparison the corpus pass makes (count, payload)."""
    import numpy as np

    from nvorbis_amd.reader import Stream, demux_ogg_array, index_ogg_array
    from tests import c5_corpus, ogg_py
    ws = c5_corpus.writer_setup()
    files = [ogg_bytes[k] for k in ("1test", "2test", "3test", "issue6test")] + [c5_corpus.corpus_file(ws, i, 0.05) for i in (0, 7, 19)]
    for data in files:
        pa = demux_ogg_array(data)
        qa, payload = index_ogg_array(data)
        n = len(pa)
        assert len(qa) == n and payload == int(pa.offsets[-1])
        assert np.array_equal(pa.granules[:n], qa.granules[:n]) and np.array_equal(pa.flags[:n], qa.flags[:n])
        assert all(pa[k] == qa[k] for k in range(3)) and all(pa[k][:8] == qa[k] for k in range(3, n))
        st = Stream(None, pa[0], pa[1], pa[2])
        try:
            a, b = st.index_packets(pa, 3), st.index_packets(qa, 3)
        finally:
            st.close()
        assert a[3] == b[3] and all(np.array_equal(x, y) for x, y in zip(a[:3], b[:3]))
        # a flipped body byte in the third-to-last page: the checksum refuses the page, the index cannot know
        pages = ogg_py.read_pages(data)
        if len(pages) >= 6:
            pg = pages[-3]
            bad = bytearray(data)
            bad[pg["offset"] + pg["length"] - 1] ^= 0x40
            bad = bytes(bad)
            qb, payload_b = index_ogg_array(bad)
            pb = demux_ogg_array(bad)
            assert (len(qb), payload_b) == (n, payload)                      # the index takes the page at its word
            assert (len(pb), int(pb.offsets[-1])) != (len(qb), payload_b)    # the checked list differs: the corpus pass re-indexes
    # chained streams: the index form follows the same stream bookkeeping
    two = files[0] + files[1]
    for k in (0, 1):
        pa = demux_ogg_array(two, k)
        qa, payload = index_ogg_array(two, k)
        assert len(pa) == len(qa) and payload == int(pa.offsets[-1])


def test_clmul_page_checksum_equals_the_table(tmp_path):
    """Ogg/Crc.cs:5-40 (polynomial 0x04c11db7, most significant bit first): the library checks pages by carry-less multiplication
    where the CPU has it (host_ogg.cpp: crc_clmul -- the corpus pass checks 3.3 GB of pages inside its timed region) and by the
    eight-byte table otherwise; tools/crc_check.cpp includes the source and compares the two on random buffers (lengths around
    the 16- and 64-byte block sizes, up to a maximal page, random initial values)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    if "pclmul" not in open("/proc/cpuinfo").read():
        pytest.skip("no pclmul on this CPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "crc_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "nvorbis_amd", "csrc"), os.path.join(root, "tools", "crc_check.cpp"), "-o", exe])
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("0 of "), out.stdout


def test_committed_profiles_belong_to_the_committed_sources():
    """profiles/traffic.json (the PMC passes behind bench.py's `roofline.traffic`) and the committed bench lines of the closing build
    carry the source hash of the library they were taken with (nvorbis_amd/build.py: every file under csrc/, the public header, the
    flags).  It must be the hash of the sources in the tree: numbers of another build are not this build's numbers.  (A change under
    csrc/ or include/ makes this fail until tools/round6_profiles.sh has run on the GPU box again -- that is the point.)"""
    import json
    from nvorbis_amd import build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = "nvh-src-hash=" + build.source_hash()
    traffic = json.load(open(os.path.join(root, "profiles", "traffic.json")))
    assert traffic["build"].endswith(want), (traffic["build"], want)
    assert "k_synth_group2" in traffic["kernels"]
    for name in ("r06_bench.json", "r06_bench_b.json", "r06_bench_driver.json"):
        line = json.loads(open(os.path.join(root, "profiles", name)).read().strip().splitlines()[-1])
        assert line["build"].endswith(want), (name, line["build"], want)
        assert line["roofline"]["traffic_build_matches"] is True and line["pcm_digest_ok"] is True
        # the line's own arithmetic: value = frames of the timed region / its duration; frac = algorithmic bytes / duration / peak
        c = line["config"]
        frames = c["frames_per_gpu"] * c["passes_per_step"] * line["steps"]
        assert abs(line["value"] - frames / c["timed_region_s"]) < 1e-6 * line["value"]
        r = line["roofline"]
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["peak"] == 8000.0
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
        assert r["algorithmic_bytes_per_launch"] == 2048 * 2 * (1024 * 4 + 1024 * 4)  # n/2 floats in, n/2 out per channel-frame
        assert 0.95 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.10  # PMC bytes per launch: no wasted re-reads
