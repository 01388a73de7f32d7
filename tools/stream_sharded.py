"""SURVEY 8e, second partitioning: ONE long stream decoded by all GPUs of a node, contiguous packet chunks with a
one-packet lead-in (nvorbis_amd.corpus.plan_stream_chunks / decode_stream_chunk / decode_stream_sharded).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         tools/stream_sharded.py [--file F.ogg] [--repeat 64] [--check]

Without --file the stream is tests/golden/3test.ogg with its audio packets repeated --repeat times (granules dropped, one
end-of-stream flag at the very end): a long stream with real packet statistics.  Every rank plans the cuts (light host
parse: packet type, mode number, window flags), decodes its own chunk and rank 0 gathers the PCM over RCCL / xGMI.
--check decodes the whole stream serially on rank 0 as well and compares byte for byte.  Rank 0 prints one JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import nvorbis_amd as nv
from nvorbis_amd import corpus

ap = argparse.ArgumentParser()
ap.add_argument("--file", type=str, default=None)
ap.add_argument("--repeat", type=int, default=64)
ap.add_argument("--check", action="store_true")
a = ap.parse_args()
rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(local)
dist = None
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
data = open(a.file or os.path.join(root, "tests", "golden", "3test.ogg"), "rb").read()
from nvorbis_amd.reader import PacketArray, demux_ogg_array
pa = demux_ogg_array(data)
if not a.file and a.repeat > 1:
    h_end, a_end = int(pa.offsets[3]), int(pa.offsets[len(pa)])
    lens = np.diff(pa.offsets)
    blob = np.concatenate([pa.data[:h_end]] + [pa.data[h_end:a_end]] * a.repeat)
    all_lens = np.concatenate([lens[:3]] + [lens[3:]] * a.repeat)
    offs = np.zeros(all_lens.size + 1, np.int64)
    offs[1:] = np.cumsum(all_lens)
    gran = np.full(all_lens.size, -1, np.int64)
    flags = np.zeros(all_lens.size, np.uint8)
    flags[-1] = 1
    pa = PacketArray(blob, offs, gran, flags)
pk, gr, fl = pa, pa.granules, pa.flags  # one contiguous buffer: packets are pushed a batch per library call
ctx = nv.Context(local)
if dist is not None:
    dist.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
out = corpus.decode_stream_sharded(pk, gr, fl, rank, world, dist, "cuda:%d" % local, ctx=ctx)
torch.cuda.synchronize()
t1 = time.perf_counter()
if rank == 0:
    res = {"packets": len(pk) - 3, "n_gpus": world, "seconds": t1 - t0, "frames_per_s": (len(pk) - 3) / (t1 - t0), "pcm_floats": int(out.size)}
    if a.check:
        t2 = time.perf_counter()
        serial, _ = corpus.decode_stream_chunk(ctx, pk, gr, fl, corpus.plan_stream_chunks(pk, gr, fl, 1, ctx)[0], True)
        res["serial_seconds"] = time.perf_counter() - t2
        res["identical"] = bool(serial.tobytes() == out.tobytes())
    print(json.dumps(res), flush=True)
if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
