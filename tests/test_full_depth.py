"""GPU parity on the BASELINE.json workloads with FULL-DEPTH packets (SURVEY 8d generators, tests/vorbis_encode.py).

Random-byte packets end long before the residue does, so they only ever exercise the first few partitions; here every
packet is written by the structured encoder (chosen floor posts, a classification for every partition, a VQ entry for
every vector of every cascade stage) and the oracle's own residue trace is asserted to cover the spectrum.
All comparisons are bit-exact against the oracle, through the C ABI."""
import numpy as np
import pytest

from tests import ogg_py, vorbis_encode as ve

pytestmark = pytest.mark.gpu


def _decode_gpu(nv, ctx, pk, gr, fl, clip, batch_frames, gpu_parse=False):
    dec = nv.StreamDecoder(ctx, pk, gr, fl, batch_frames=batch_frames, gpu_parse=gpu_parse)
    dec.ClipSamples = clip
    chunks = []
    buf = np.zeros(1 << 22, np.float32)
    buf = buf[: buf.size - buf.size % dec.Channels]
    while True:
        n = dec.Read(buf, 0, buf.size)
        if n == 0:
            break
        chunks.append(buf[:n].copy())
    dec.close()
    return np.concatenate(chunks) if chunks else np.zeros(0, np.float32)


def _coverage(oracle, S, headers, packets, end_per_channel):
    d = oracle.open_headers(headers)
    try:
        fracs, stages = [], set()
        for p in packets:
            got = oracle.packet_coverage(d, p)
            assert got is not None
            bs, mask = got
            if bs != S.block1:
                continue
            m = mask[:, :end_per_channel]
            fracs.append(float((m != 0).mean()))
            for s in range(8):
                if (m & (1 << s)).any():
                    stages.add(s)
        return fracs, stages
    finally:
        oracle.L.orc_close(d)


def _assert_same(got, ref, what):
    assert got.size == ref.size, (what, got.size, ref.size)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (what, float(np.abs(got - ref).max()))


@pytest.mark.parametrize("psize", [48, 32])
def test_c4_six_channel_full_depth_bit_exact(oracle, gpu_ctx, ogg_bytes, psize):
    """BASELINE C4: 6 channels, 48 kHz, n = 4096, coupling [(0,2),(3,4)], Residue2 over 6 channels, end = 6 * 1536.
    psize 48 = the headline shape, psize 32 = quirk B-1 at six channels (Residue2.cs:25-27: offset /= channels with
    chPtr = 0, partitions not a multiple of the channel count -> the sequential path).  The oracle's residue trace of
    the very packets decoded shows >= 95 % of [0, 1536) per channel touched (psize 48) and all three cascade stages."""
    import nvorbis_amd as nv
    hdr = ve.c4_headers(ve.shipped_headers(ogg_bytes["3test"]), psize=psize)
    S = ve.setup_of(hdr)
    weights = [0] + [1] * 9  # class 0 of this setup carries no books
    pool = ve.packet_pool(S, 100 + psize, per_kind=40, class_weights=weights)
    fracs, stages = _coverage(oracle, S, hdr, pool[(True, 1, 1)], 1536)
    assert stages == {0, 1, 2}
    assert min(fracs) >= (0.95 if psize == 48 else 0.60), min(fracs)
    rng = np.random.default_rng(psize)
    # all-long (the C4 definition) and a mixed stream with short blocks and every window transition
    for kinds in (np.ones(1100, dtype=bool), ve.markov_kinds(rng, 300, 0.1, 0.3)):
        pk, gr = ve.stream_from_pool(S, hdr, pool, kinds, rng)
        fl = [0] * len(pk)
        for clip in (True, False):
            ref, info = oracle.decode_packets(pk, gr, fl, clip=clip)
            assert info["channels"] == 6
            for bf in (13, 1024):
                _assert_same(_decode_gpu(nv, gpu_ctx, pk, gr, fl, clip, bf), ref, (psize, clip, bf))
        _assert_same(_decode_gpu(nv, gpu_ctx, pk, gr, fl, True, 512, gpu_parse=True), oracle.decode_packets(pk, gr, fl)[0],
                     (psize, "gpu_parse"))


def test_c2_grand_full_size_bit_exact(oracle, gpu_ctx, ogg_bytes):
    """BASELINE C2, generator G-rand (numpy default_rng(20260928)): 4096 stereo long frames, n = 2048, Floor1 + Residue2
    on 3test.ogg's setup, every packet unique and full depth; one 4096-frame batch and small batches, host and GPU parser."""
    import nvorbis_amd as nv
    hdr = ve.shipped_headers(ogg_bytes["3test"])
    S = ve.setup_of(hdr)
    kinds = np.ones(4097, dtype=bool)
    pk, gr = ve.encode_stream(S, hdr, kinds, 20260928)
    fl = [0] * len(pk)
    fracs, stages = _coverage(oracle, S, hdr, pk[3:200], 1888 // 2)
    assert stages == {0, 1, 2} and np.mean(fracs) >= 0.88
    for clip in (True, False):
        ref, _ = oracle.decode_packets(pk, gr, fl, clip=clip)
        assert ref.size == (4096 * 1024 + 1024) * 2
        _assert_same(_decode_gpu(nv, gpu_ctx, pk, gr, fl, clip, 4096), ref, ("C2", clip, 4096))
    ref, _ = oracle.decode_packets(pk, gr, fl)
    _assert_same(_decode_gpu(nv, gpu_ctx, pk, gr, fl, True, 333), ref, ("C2", 333))
    _assert_same(_decode_gpu(nv, gpu_ctx, pk, gr, fl, True, 4096, gpu_parse=True), ref, ("C2", "gpu_parse"))


def test_c3_markov_8192_frames_bit_exact(oracle, gpu_ctx, ogg_bytes):
    """BASELINE C3: 8192 frames, block kinds from a 2-state Markov chain (P(L->S) = 0.03, P(S->L) = 0.12, seed 7) so all four
    long windows and the short window occur with lapping transitions; full-depth packets; output count = sum(valid - start)."""
    import nvorbis_amd as nv
    hdr = ve.shipped_headers(ogg_bytes["3test"])
    S = ve.setup_of(hdr)
    rng = np.random.default_rng(7)
    kinds = ve.markov_kinds(rng, 8192)
    assert 0.1 < 1.0 - kinds.mean() < 0.4
    pool = ve.packet_pool(S, 7, per_kind=96)
    pk, gr = ve.stream_from_pool(S, hdr, pool, kinds, rng)
    fl = [0] * len(pk)
    ref, info = oracle.decode_packets(pk, gr, fl, trace=True)
    tr = info["trace"]
    seen = {(int(r[4]), int(r[5])) for r in tr if r[3]}
    assert seen >= {(256, 0), (2048, 0), (2048, 1), (2048, 2), (2048, 3)}
    # every packet but the first emits valid - start; the provider runs dry at the end, so the last block's tail follows
    ok = [r for r in tr if r[3]]
    emitted = sum(int(r[1] - r[0]) for r in ok[1:]) + int(ok[-1][2] - ok[-1][1])
    assert ref.size == emitted * 2
    assert gr[-1] * 2 <= ref.size
    for bf in (1024, 4096):
        _assert_same(_decode_gpu(nv, gpu_ctx, pk, gr, fl, True, bf), ref, ("C3", bf))
    _assert_same(_decode_gpu(nv, gpu_ctx, pk, gr, fl, True, 2048, gpu_parse=True), ref, ("C3", "gpu_parse"))
    # the launch shapes a worker pool asks for (nvh_ctx_set_parse_lanes): several packets per wavefront, and -- block sizes mix --
    # the batch's frames handed to the parser longest packet first; 4096 frames at 8 and 2 lanes, 2048 at 8 (256 wavefronts)
    pool_ctx = nv.Context(0)
    try:
        for lanes, bf in ((8, 4096), (2, 4096), (8, 2048), (64, 8192)):
            pool_ctx.set_parse_lanes(lanes)
            _assert_same(_decode_gpu(nv, pool_ctx, pk, gr, fl, True, bf, gpu_parse=True), ref, ("C3", "gpu_parse lanes", lanes, bf))
    finally:
        pool_ctx.close()


def test_c5_corpus_world1_from_device_buffers(oracle, ogg_bytes):
    """BASELINE C5 at world size 1: files written by the corpus writer (Huffman-encoded side information, CRC-valid pages,
    Markov block kinds, seed = file index) + the shipped files, decoded file-parallel into ONE device arena and "gathered"
    from device memory (tools/corpus_transcode.py's path: transcode(..., to_host=False)); every file equals the oracle."""
    import torch
    from nvorbis_amd import corpus
    hdr = ve.shipped_headers(ogg_bytes["3test"])
    S = ve.setup_of(hdr)
    pool = ve.packet_pool(S, 5, per_kind=24)
    files = [ve.corpus_file(S, hdr, pool, i, scale=0.02) for i in range(28)] + [ogg_bytes[n] for n in ("1test", "2test", "3test", "issue6test")]
    out = corpus.transcode(files, rank=0, world=1, dist=None, device="cuda:0", gpu=0, workers=6, to_host=False)
    assert len(out) == len(files) and all(isinstance(o, torch.Tensor) and o.is_cuda for o in out)
    # one arena: consecutive files are adjacent in device memory
    assert all(out[i].data_ptr() + 4 * out[i].numel() == out[i + 1].data_ptr() for i in range(len(out) - 1))
    for i, data in enumerate(files):
        ref, _ = oracle.decode_ogg(data)
        got = out[i].cpu().numpy()
        _assert_same(got, ref, ("C5 file", i))
    # and the threaded host-destination path gives the same bytes
    host = corpus.decode_files_threaded(files[:8], device=0, workers=4, batch_frames=700)
    for i in range(8):
        _assert_same(host[i], out[i].cpu().numpy(), ("C5 host", i))


@pytest.mark.parametrize("scale,gpu_parse", [(0.1, False), (0.1, True), (1.0, False)])
def test_c5_corpus_1004_files_digests(scale, gpu_parse):
    """BASELINE C5 at world size 1, >= 1000 files: the SURVEY 8d corpus (tests/c5_corpus.py: 1000 writer files, lengths
    log-uniform 5-300 s x scale, seed = file index, + the 4 TestFiles) decoded file-parallel through the HIP path into one
    device arena (corpus.decode_files_to_device, what corpus.transcode(..., world=1, to_host=False) runs per rank); the
    SHA-256 of every file's PCM equals the oracle's committed digest (tests/golden/c5_digests_scale*.json, written by
    tools/corpus_c5.py --make-digests).  scale 0.1 (0.5-30 s per file, 270 k frames, 2.2 GB of PCM) runs always; the stated
    size, scale 1.0 (3.1 M frames, 21.6 GB of PCM in one device arena, about half a minute on the GPU box), runs too unless
    NVH_C5_SKIP_FULL=1 (and not in the toggle replays of test_fallback_kernel_paths_bit_exact).  gpu_parse: the way bench.py's
    corpus block runs it -- the GPU packet parser with the worker pool's launch shape (nvh_ctx_set_parse_lanes(8)), twice over
    worker contexts that are kept between the two jobs (keep_contexts)."""
    import os

    from nvorbis_amd import corpus
    from tests import c5_corpus
    if scale == 1.0 and (os.environ.get("NVH_C5_SKIP_FULL") or os.environ.get("NVH_TEST_CHILD")):
        pytest.skip("NVH_C5_SKIP_FULL / a toggle replay: the full-size corpus (21.6 GB of PCM) runs once, in the default mode")
    dig = c5_corpus.load_digests(scale)
    assert dig is not None, "tests/golden/c5_digests_scale%g.json is missing" % scale
    files = c5_corpus.build_files(scale)
    assert len(files) == 1004
    assert [c5_corpus.file_digest(f) for f in files] == [r[0] for r in dig["digests"]]  # the very files the oracle decoded
    for rep in range(2 if gpu_parse else 1):
        arena, views = corpus.decode_files_to_device(files, device=0, workers=16, gpu_parse=gpu_parse, keep_contexts=gpu_parse)
        assert int(arena.numel()) == dig["total_floats"]
        bad = [i for i, v in enumerate(views)
               if int(v.numel()) != dig["digests"][i][1] or c5_corpus.pcm_digest(v.cpu().numpy()) != dig["digests"][i][2]]
        assert not bad, (rep, bad[:10])
        del arena, views
    corpus.close_worker_contexts()


@pytest.mark.gpu
@pytest.mark.parametrize("gpu_parse", [False, True])
def test_corpus_pass_reindexes_a_file_with_a_damaged_page(gpu_parse):
    """The corpus pass sizes its arena from the lacing-only index (no page checksums); the decode pass demultiplexes with them.  A
    file one of whose pages fails its checksum decodes to another packet list than the index promised (the reference's reader
    drops the page and resynchronises, Ogg/PageReaderBase.cs:33-70): the pass finds out by count and payload, decodes that
    file from the checked list into a tensor of its own -- bit-exact against the oracle's decode of the damaged bytes -- and
    the other files' PCM is what it would have been."""
    import numpy as np

    from nvorbis_amd import corpus
    from tests import c5_corpus, ogg_py, oracle_py
    ws = c5_corpus.writer_setup()
    files = [c5_corpus.corpus_file(ws, i, 0.05) for i in range(6)]
    pages = ogg_py.read_pages(files[2])
    bad = bytearray(files[2])
    pg = pages[len(pages) // 2]
    bad[pg["offset"] + pg["length"] - 3] ^= 0x11
    files[2] = bytes(bad)
    orc = oracle_py.load()
    want = [orc.decode_ogg(f)[0] for f in files]
    t = {}
    arena, views = corpus.decode_files_to_device(files, device=0, workers=4, gpu_parse=gpu_parse, timings=t)
    assert t.get("files_reindexed") == [2]
    for i, (v, w) in enumerate(zip(views, want)):
        got = v.cpu().numpy()
        assert got.size == w.size and np.array_equal(got.view(np.uint32), w.view(np.uint32)), i
