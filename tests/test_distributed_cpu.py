"""Multi-rank path on CPU: file-parallel sharding + PCM gather with world_size 2 over gloo."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lpt_shards_balanced_and_complete():
    from nvorbis_amd.corpus import lpt_shards
    sizes = [100, 90, 80, 10, 10, 10, 5, 300]
    sh = lpt_shards(sizes, 3)
    assert sorted(i for s in sh for i in s) == list(range(len(sizes)))
    loads = [sum(sizes[i] for i in s) for s in sh]
    assert max(loads) == 300 and min(loads) >= 100 and sum(loads) == sum(sizes)
    assert lpt_shards(sizes, 3) == sh  # deterministic
    assert lpt_shards([], 2) == [[], []]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, files, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from nvorbis_amd.corpus import transcode
    from tests import oracle_py
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = oracle_py.load()
    out = transcode(files, lambda b: orc.decode_ogg(b)[0], rank, world, dist, "cpu")
    if rank == 0:
        q.put([o.tobytes() for o in out])
    dist.destroy_process_group()


def _arena_worker(rank, world, port, files, q):
    """As _worker, but the rank's PCM sits in one torch arena and the gather gets slices of it (the shape the GPU path
    has: decode_files_to_device + gather_pcm(to_host=False)); the payload must go out without a concatenation copy."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from nvorbis_amd import corpus
    from tests import oracle_py
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = oracle_py.load()
    mine = corpus.lpt_shards([len(f) for f in files], world)[rank]
    pcm = [orc.decode_ogg(files[i])[0] for i in mine]
    arena = torch.from_numpy(np.concatenate(pcm)) if pcm else torch.zeros(0)
    local, off = {}, 0
    for i, a in zip(mine, pcm):
        local[i] = arena[off:off + a.size]
        off += a.size
    flat = corpus._flat_payload([local[i] for i in sorted(local)], torch, torch.device("cpu"))
    assert flat.data_ptr() == arena.data_ptr() and flat.numel() == arena.numel()
    out = corpus.gather_pcm(local, len(files), rank, world, dist, "cpu", to_host=False)
    if rank == 0:
        assert all(isinstance(o, torch.Tensor) for o in out)
        q.put([o.numpy().tobytes() for o in out])
    dist.destroy_process_group()


def test_gloo_world2_gather_from_arena_slices(oracle, ogg_bytes):
    """The device-resident form of the gather (tensor slices of one arena in, tensors out), on CPU tensors over gloo."""
    import torch.multiprocessing as mp
    files = [ogg_bytes[n] for n in ("3test", "1test", "2test", "issue6test", "1test")]
    single = [oracle.decode_ogg(b)[0].tobytes() for b in files]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_arena_worker, args=(r, 2, port, files, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == single


def test_gloo_world2_gather_is_byte_identical(oracle, ogg_bytes):
    """Gathered PCM of a 2-rank run == single-process PCM, byte for byte.  The decoder behind
    the shard is the CPU oracle here (no GPU in this suite); the GPU suite runs the same code path with
    the HIP reader."""
    import torch.multiprocessing as mp
    files = [ogg_bytes[n] for n in ("1test", "2test", "3test", "1test", "issue6test", "2test")]
    single = [oracle.decode_ogg(b)[0].tobytes() for b in files]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, files, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == single


# ---- one stream across ranks: contiguous packet chunks with a one-packet lead-in ----

def _oracle_chunk_decoder(orc, pk, gr, fl):
    """decode_chunk_fn for decode_stream_sharded backed by the CPU oracle: a fresh decoder over the headers, the lead-in
    packet and the chunk's packets; the oracle has no way to stop without draining, so the drained tail is cut off at the
    sample count the plan states."""
    def fn(chunk, final):
        first, last = chunk["first"], chunk["last"]
        lo = first - 1 if first > 3 else 3
        sel = list(pk[:3]) + list(pk[lo:last])
        g = [-1, -1, -1] + list(gr[lo:last])
        f = [0, 0, 0] + list(fl[lo:last])
        pcm, info = orc.decode_packets(sel, g, f)
        want = (chunk["emitted1"] - chunk["emitted0"]) * info["channels"]
        assert pcm.size >= want, (first, last, pcm.size, want)
        return pcm[:want]
    return fn


@pytest.mark.parametrize("name", ["1test", "2test", "3test", "issue6test"])
@pytest.mark.parametrize("world", [2, 3, 7])
def test_stream_chunk_plan_and_serial_equivalence(oracle, ogg_bytes, name, world):
    """The chunk plan tiles the stream (contiguous packets, contiguous sample counts, the reference's total) and the
    chunks decoded independently -- lead-in packet first -- concatenate to the serial decode, byte for byte."""
    import nvorbis_amd as nv
    from nvorbis_amd.corpus import decode_stream_sharded, plan_stream_chunks
    pk, gr, fl = nv.demux_ogg(ogg_bytes[name])
    chunks = plan_stream_chunks(pk, gr, fl, world)
    assert chunks[0]["first"] == 3 and chunks[-1]["last"] == len(pk) and chunks[0]["emitted0"] == 0
    for a, b in zip(chunks, chunks[1:]):
        assert a["last"] == b["first"] and a["emitted1"] == b["emitted0"]
    serial, info = oracle.decode_ogg(ogg_bytes[name])
    assert chunks[-1]["emitted1"] * info["channels"] == serial.size
    assert len(chunks) == min(world, len(chunks))
    got = decode_stream_sharded(pk, gr, fl, decode_chunk_fn=_oracle_chunk_decoder(oracle, pk, gr, fl), world=1)
    # world=1 above only means "this process decodes every chunk"; the plan itself was made for `world` chunks
    assert got.tobytes() == serial.tobytes()
    parts = [_oracle_chunk_decoder(oracle, pk, gr, fl)(c, i == len(chunks) - 1) for i, c in enumerate(chunks)]
    assert np.concatenate(parts).tobytes() == serial.tobytes()


def _chunk_worker(rank, world, port, data, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import nvorbis_amd as nv
    from nvorbis_amd.corpus import decode_stream_sharded
    from tests import oracle_py
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = oracle_py.load()
    pk, gr, fl = nv.demux_ogg(data)
    out = decode_stream_sharded(pk, gr, fl, rank, world, dist, "cpu", decode_chunk_fn=_oracle_chunk_decoder(orc, pk, gr, fl))
    if rank == 0:
        q.put(out.tobytes())
    dist.destroy_process_group()


def test_gloo_world2_stream_chunks_byte_identical(oracle, ogg_bytes):
    """One stream decoded by two ranks (each its own chunk, PCM gathered to rank 0) == the serial decode."""
    import torch.multiprocessing as mp
    data = ogg_bytes["3test"]
    serial = oracle.decode_ogg(data)[0].tobytes()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chunk_worker, args=(r, 2, port, data, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == serial
