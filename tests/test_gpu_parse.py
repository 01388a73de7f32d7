"""GPU packet parser (kernels_parse.hip): the descriptors k_parse produces lead to exactly the PCM
the host parser's descriptors lead to (which the rest of the suite pins to the oracle), for every stream shape inside
its limits; streams outside them are refused; packets the reference would throw on are reported."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _decode(nv, ctx, pk, gr, fl, gpu_parse, batch_frames, clip=True):
    dec = nv.StreamDecoder(ctx, pk, gr, fl, batch_frames=batch_frames, gpu_parse=gpu_parse)
    dec.ClipSamples = clip
    chunks = []
    buf = np.zeros(1 << 20, np.float32)
    buf = buf[: buf.size - buf.size % dec.Channels]
    while True:
        n = dec.Read(buf, 0, buf.size)
        if n == 0:
            break
        chunks.append(buf[:n].copy())
    dec.close()
    return np.concatenate(chunks) if chunks else np.zeros(0, np.float32)


@pytest.mark.parametrize("name", ["1test", "2test", "3test", "issue6test"])
@pytest.mark.parametrize("batch_frames", [5, 4096])
def test_files_gpu_parse_equals_oracle(oracle, gpu_ctx, ogg_bytes, name, batch_frames):
    import nvorbis_amd as nv
    ref, _ = oracle.decode_ogg(ogg_bytes[name])
    rd = nv.VorbisReader(ogg_bytes[name], ctx=gpu_ctx, batch_frames=batch_frames, gpu_parse=True)
    assert rd._dec._stream.pending() == (0, 0)
    got = rd.read_all()
    rd.close()
    assert got.size == ref.size and np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("name", ["mono_res0_small_blocks", "stereo_res1_coupled", "three_ch_res2_misaligned", "six_ch_res2_4096",
                                  "two_submaps", "equal_blocks_overrun", "mono_8192", "stereo_8192",
                                  "res0_slab", "odd_dims_slab", "res2_alias_stereo", "two_pass_slab", "res0_3ch",
                                  "table_books_pair", "table_books_general", "table_books_b1"])
def test_synthetic_shapes_gpu_parse_equals_oracle(oracle, gpu_ctx, name):
    """Residue0/1/2, 1-6 channels, several submaps, vector overrun, 64..8192 blocks, random-bit packets (so packets end in
    the middle of floors, class words and vectors all the time)."""
    import nvorbis_amd as nv
    from tests import synth_stream as ss
    pk, gr, fl = ss.filtered_stream(oracle, name, 160, 11, True)
    ref, _ = oracle.decode_packets(pk, gr, fl, clip=True)
    st = nv.Stream(gpu_ctx, pk[0], pk[1], pk[2])
    st.set_gpu_parse(True)  # raises if the shape were outside the GPU parser's limits
    st.close()
    for bf in (3, 64):
        got = _decode(nv, gpu_ctx, pk, gr, fl, True, bf)
        assert got.size == ref.size and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (name, bf)


def test_floor0_streams_are_refused(oracle, gpu_ctx):
    import nvorbis_amd as nv
    from nvorbis_amd import native
    from tests import synth_stream as ss
    pk, gr, fl = ss.filtered_stream(oracle, "floor0_stereo", 20, 5)
    st = nv.Stream(gpu_ctx, pk[0], pk[1], pk[2])
    with pytest.raises(native.NvhError) as e:
        st.set_gpu_parse(True)
    assert e.value.code == native.ERR_UNSUPPORTED
    st.close()


def _throwing_stream():
    from tests import synth_stream as ss
    cfg = ss.config("stereo_res1_coupled")
    old = cfg["books"][3]
    cfg["books"][3] = ss.IncompleteBook(old.bits, dims=old.dims, lookup=old.lookup, min_me=old.min_me, delta_me=old.delta_me,
                                        value_bits=old.value_bits, sequence_p=old.sequence_p, mults=old.mults)
    return ss.make_stream(cfg, 200, 1)  # seed 1: the first throwing packet is the third audio packet


def test_throwing_packet_keeps_the_rest_of_the_batch(gpu_ctx):
    """A packet that makes the managed decoder throw -- here the unassigned code of an incomplete Huffman tree without an
    overflow list (Codebook.cs:306, NullReferenceException) -- fails nvh_stream_push_packet on the host path with
    NVH_ERR_RUNTIME.  In GPU-parse mode k_parse finds it inside the look-ahead batch: the batch is parsed again on the
    host, nvh_stream_synth delivers the PCM of every other packet together with the same code, and
    nvh_stream_parse_errors says where in that PCM each exception belongs."""
    import nvorbis_amd as nv
    from nvorbis_amd import native
    pk, gr, fl = _throwing_stream()
    st = nv.Stream(gpu_ctx, pk[0], pk[1], pk[2])
    bad, before = [], []
    for i in range(3, 40):
        try:
            st.push_packet(pk[i], gr[i], fl[i])
        except native.NvhError as e:
            assert e.code == native.ERR_RUNTIME
            bad.append(i)
            before.append(st.pending()[1])
    assert bad and bad[0] > 3, "random packets never hit the unassigned code"
    ref = st.synth_host().copy()
    assert st.parse_errors == []
    st.close()
    st = nv.Stream(gpu_ctx, pk[0], pk[1], pk[2])
    st.set_gpu_parse(True)
    for i in range(3, 40):
        st.push_packet(pk[i], gr[i], fl[i])  # light parse: nothing to throw on yet
    got = st.synth_host().copy()
    assert [(e.code, at) for e, at in st.parse_errors] == [(native.ERR_RUNTIME, b * st.channels) for b in before]
    assert got.size == ref.size and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert st.pending() == (0, 0)
    # the stream goes on in GPU-parse mode with the state the host-parse path has at this point
    more = [i for i in range(40, 60)]
    for i in more:
        try:
            st.push_packet(pk[i], gr[i], fl[i])
        except native.NvhError:
            pass
    st.synth_host()
    st.close()


@pytest.mark.parametrize("gpu_parse", [False, True])
@pytest.mark.parametrize("batch_frames", [7, 64])
def test_exceptions_surface_where_the_reference_throws(oracle, gpu_ctx, gpu_parse, batch_frames):
    """Read() past throwing packets, one sample frame per call (so that no call loses samples it had already copied, as
    the reference's Read does when it throws): the same samples and the same exceptions at the same positions as the
    oracle's read loop, whichever parser runs and however far the look-ahead reaches."""
    import ctypes as C
    import nvorbis_amd as nv
    from nvorbis_amd import native
    pk, gr, fl = _throwing_stream()
    L = oracle.L
    blob = np.frombuffer(b"".join(pk), dtype=np.uint8)
    offs = np.zeros(len(pk) + 1, np.int64)
    offs[1:] = np.cumsum([len(p) for p in pk])
    g, f, err = np.asarray(gr, np.int64), np.asarray(fl, np.uint8), C.c_int(0)
    d = L.orc_open_packets(blob.ctypes.data, offs.ctypes.data, g.ctypes.data, f.ctypes.data, len(pk), C.byref(err))
    assert d
    buf = np.zeros(2, np.float32)
    ref, ref_err = [], []
    while True:
        n = L.orc_read_samples(d, buf.ctypes.data, 2, 0, 2)
        if n < 0:
            ref_err.append((len(ref), n))
            continue
        if n == 0:
            break
        ref.append(buf[:n].copy())
    L.orc_close(d)
    assert len(ref_err) > 50
    dec = nv.StreamDecoder(gpu_ctx, pk, gr, fl, batch_frames=batch_frames, gpu_parse=gpu_parse)
    got, got_err = [], []
    while True:
        try:
            n = dec.Read(buf, 0, 2)
        except native.NvhError as e:
            got_err.append((len(got), e.code))
            continue
        if n == 0:
            break
        got.append(buf[:n].copy())
    dec.close()
    assert got_err == ref_err
    a, b = np.concatenate(got), np.concatenate(ref)
    assert a.size == b.size and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("gpu_parse", [False, True])
def test_paired_emission_with_silent_channels(oracle, gpu_ctx, ogg_bytes, gpu_parse):
    """A steady long/long run in which now and then a channel has no floor (ExecuteChannel false, Mapping.cs:104-131).  The host
    marks paired-emission candidates from the geometry; for GPU-parsed batches the execute flags exist only on the device, so
    k_parse_links withdraws the candidates that involve such a frame and k_ola_compact takes those frames back
    (nvh_launch.hip: ola_all).  Both parsers, bit-exact against the oracle; batches of 512 are > 7/8 steady state, so the
    emission is on."""
    import nvorbis_amd as nv
    from tests import vorbis_encode as ve
    hdr = ve.shipped_headers(ogg_bytes["3test"])
    S = ve.setup_of(hdr)
    enc = ve.PacketEncoder(S)
    rng = np.random.default_rng(5)
    long_mode = next(i for i, (f, _) in enumerate(S.modes) if f)
    n = 1200
    pk = list(hdr)
    for i in range(n):
        # (a silent magnitude / angle channel of a coupled pair executes anyway when its partner does, Mapping.cs:112-119:
        # both silent is what switches the pair off)
        silent = (0, 1) if i % 97 == 40 else ((1,) if i % 151 == 75 else ())
        pk.append(enc.packet(rng, long_mode, 1, 1, silent=silent))
    gr = [-1, -1, -1] + ve.granules_for(S, np.ones(n, dtype=bool))
    fl = [0] * len(pk)
    ref, _ = oracle.decode_packets(pk, gr, fl, clip=True)
    for bf in (512, 100):
        got = _decode(nv, gpu_ctx, pk, gr, fl, gpu_parse, bf)
        assert got.size == ref.size and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (gpu_parse, bf)


@pytest.mark.parametrize("lanes", [1, 2, 8, 64])
def test_parse_lanes_of_a_context_do_not_change_the_pcm(oracle, ogg_bytes, lanes):
    """nvh_ctx_set_parse_lanes: the parser's launch shape a worker pool asks for (packets per wavefront; 1 = the wave-uniform
    kernel k_parse_slab_u, more = k_parse_slab's divergent lanes) -- shipped files and a random-bit stream whose packets end
    anywhere, both against the oracle; an argument that is not a power of two up to 64 is refused."""
    import nvorbis_amd as nv
    from nvorbis_amd import native
    from tests import synth_stream as ss
    ctx = nv.Context(0)
    try:
        for bad in (3, -1, 128):
            with pytest.raises(native.NvhError):
                ctx.set_parse_lanes(bad)
        ctx.set_parse_lanes(lanes)
        for name in ("3test", "issue6test"):
            ref, _ = oracle.decode_ogg(ogg_bytes[name])
            rd = nv.VorbisReader(ogg_bytes[name], ctx=ctx, batch_frames=300, gpu_parse=True)
            got = rd.read_all()
            rd.close()
            assert got.size == ref.size and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (name, lanes)
        pk, gr, fl = ss.filtered_stream(oracle, "stereo_res1_coupled", 160, 11, True)
        ref, _ = oracle.decode_packets(pk, gr, fl, clip=True)
        got = _decode(nv, ctx, pk, gr, fl, True, 64)
        assert got.size == ref.size and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), lanes
        ctx.set_parse_lanes(0)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["NVH_PARSE_LANES=8", "NVH_PARSE_LANES=32", "NVH_PARSE_LANES=16+NVH_NO_PARSE_SUB=1",
                                  "NVH_PARSE_LANES=8+NVH_PARSE_CUR=1", "NVH_PARSE_LANES=8+NVH_PARSE_CUR=0"])
def test_multi_packet_parser_forms_bit_exact(form):
    """Several packets per wavefront (kernels_parse.hip): by default the lean walk k_parse_slab_f (one cursor per lane, four
    positions per step, long codes through second-level tables; NVH_NO_PARSE_SUB: by scanning their groups), behind it the general
    body k_parse_slab_c over the frames the lean walk left -- packets that end inside the residue, faults, setups of the general bin
    walk -- and the tail kernel k_parse_slab_t; NVH_PARSE_CUR=1: the general body's cursor walk for every frame; NVH_PARSE_CUR=0:
    the lockstep nest of rounds 3-5 (k_parse_slab / _g).  NVH_PARSE_LANES forces the shape on every batch, whatever its size:
    this file (shipped files, truncated and random packets, faults) and the parity suite + the full-depth configs, GPU-parsed,
    are replayed in a child process under each form."""
    import os
    import subprocess
    import sys
    if os.environ.get("NVH_TEST_CHILD"):
        pytest.skip("already inside a replay")
    env = dict(os.environ)
    for t in form.split("+"):
        k, v = t.split("=")
        env[k] = v
    env["NVH_TEST_CHILD"] = "1"
    from tests.replay import run_children
    env_gp = dict(env)
    env_gp["NVH_GPU_PARSE"] = "1"
    # (three children side by side: this file, the parity suite GPU-parsed, the full-depth configs GPU-parsed)
    run_children([(["test_gpu_parse.py"], env, []), (["test_gpu_parity.py"], env_gp, []), (["test_full_depth.py"], env_gp, [])])
