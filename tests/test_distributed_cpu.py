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
