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
                                  "two_submaps", "equal_blocks_overrun", "mono_8192"])
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


def test_throwing_packet_is_reported_at_synth(gpu_ctx):
    """A packet that makes the managed decoder throw -- here the unassigned code of an incomplete Huffman tree without an
    overflow list (Codebook.cs:306, NullReferenceException) -- fails nvh_stream_push_packet on the host path with
    NVH_ERR_RUNTIME; in GPU-parse mode the same code comes from nvh_stream_synth for the look-ahead batch."""
    import nvorbis_amd as nv
    from nvorbis_amd import native
    from tests import synth_stream as ss
    cfg = ss.config("stereo_res1_coupled")
    old = cfg["books"][3]
    cfg["books"][3] = ss.IncompleteBook(old.bits, dims=old.dims, lookup=old.lookup, min_me=old.min_me, delta_me=old.delta_me,
                                        value_bits=old.value_bits, sequence_p=old.sequence_p, mults=old.mults)
    pk, gr, fl = ss.make_stream(cfg, 200, 1)  # seed 1: the first throwing packet is the third audio packet
    st = nv.Stream(None, pk[0], pk[1], pk[2])
    bad = None
    for i in range(3, len(pk)):
        try:
            st.push_packet(pk[i], gr[i], fl[i])
        except native.NvhError as e:
            assert e.code == native.ERR_RUNTIME
            bad = i
            break
    st.close()
    assert bad is not None and bad > 3, "random packets never hit the unassigned code"
    st = nv.Stream(gpu_ctx, pk[0], pk[1], pk[2])
    st.set_gpu_parse(True)
    for i in range(3, bad + 1):
        st.push_packet(pk[i], gr[i], fl[i])  # light parse: nothing to throw on yet
    with pytest.raises(native.NvhError) as e:
        st.synth_host()
    assert e.value.code == native.ERR_RUNTIME
    # the batch was dropped; the stream object stays usable
    assert st.pending() == (0, 0)
    st.close()
