"""BASELINE.json configs[4]: offline transcode of an .ogg corpus, file-parallel across the GPUs of one node.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         tools/corpus_transcode.py --files 1000 [--dir DIR] [--workers 16]

One process per GPU.  Files are sharded LPT-greedy by compressed size (no data-path collective), each rank decodes its
shard with a pool of host threads into one device arena (nvorbis_amd.corpus.decode_files_to_device: the overlap-add
kernels write every file's PCM at its final device address), and the only exchange is the final gather of the PCM to
rank 0 over RCCL / xGMI, device memory to device memory (all_gather of sample counts + grouped point-to-point payloads,
nvorbis_amd.corpus.gather_pcm).  Corpus: --dir DIR (*.ogg), or --synthetic (files written by
tests/vorbis_encode.py from 3test.ogg's setup, lengths log-uniform 5-300 s x --scale, seed = file index), else the four
shipped test files cycled to --files entries.  Rank 0 prints one JSON line."""
import argparse, glob, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before the HIP runtime initialises (nvorbis_amd.configure_process says why)
os.environ.setdefault("NVH_CORPUS_MALLOPT", "1")  # this process is a corpus job: the allocator settings of nvorbis_amd.corpus._tune_malloc (opt-in)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from nvorbis_amd import corpus

ap = argparse.ArgumentParser()
ap.add_argument("--files", type=int, default=1000)
ap.add_argument("--dir", type=str, default=None)
ap.add_argument("--workers", type=int, default=16)
ap.add_argument("--gpu-parse", action="store_true")
ap.add_argument("--synthetic", action="store_true")
ap.add_argument("--native-gather", type=str, default=None, metavar="ID_FILE",
                help="gather through the library's own RCCL entry points (nvh_comm_*, what a host without torch.distributed calls) "
                     "instead of torch.distributed; rank 0 writes the communicator id to ID_FILE, the other ranks read it there")
ap.add_argument("--self-p2p", action="store_true", help="with --native-gather: the root's own part goes through ncclSend / ncclRecv too")
ap.add_argument("--scale", type=float, default=0.05, help="length scale of the synthetic corpus (1.0 = 5-300 s per file)")
a = ap.parse_args()
rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
# development aid (no multi-GPU box at hand): NVH_BENCH_SHARE_GPU=1 puts every rank on device 0 and uses gloo, so that the
# shard + device-to-device gather code path runs with world > 1 on one GPU; not a scaling measurement
share_gpu = bool(os.environ.get("NVH_BENCH_SHARE_GPU"))
if share_gpu:
    local = 0
torch.cuda.set_device(local)
dist = None
# NVH_RCCL_WORLD1=1 (under torch.distributed.run --nproc-per-node 1): a one-rank RCCL process group, so that the library is
# initialised and the counts' all_gather + barriers run through it on a one-GPU box
if world > 1 or (os.environ.get("NVH_RCCL_WORLD1") and "RANK" in os.environ):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if share_gpu:
        dist.init_process_group(backend="gloo")
    else:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if a.dir:
    names = sorted(glob.glob(os.path.join(a.dir, "*.ogg")))
    files = [open(n, "rb").read() for n in names]
elif a.synthetic:
    from tests import vorbis_encode as ve
    hdr = ve.shipped_headers(open(os.path.join(root, "tests", "golden", "3test.ogg"), "rb").read())
    S = ve.setup_of(hdr)
    pool = ve.packet_pool(S, 5, per_kind=64)
    files = [ve.corpus_file(S, hdr, pool, i, scale=a.scale) for i in range(a.files)]
else:
    base = [open(os.path.join(root, "tests", "golden", n + ".ogg"), "rb").read() for n in ("1test", "2test", "3test", "issue6test")]
    files = [base[i % len(base)] for i in range(a.files)]
if dist is not None:
    dist.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
shards = corpus.lpt_shards([len(f) for f in files], world)
mine = shards[rank]
arena, views = corpus.decode_files_to_device([files[i] for i in mine], device=local, workers=a.workers, gpu_parse=a.gpu_parse)
t1 = time.perf_counter()
local_map = {i: v for i, v in zip(mine, views)}
if a.native_gather:
    import nvorbis_amd as nv
    nctx = nv.Context(local)
    if rank == 0:
        with open(a.native_gather + ".tmp", "wb") as f:
            f.write(nv.Comm.unique_id())
        os.replace(a.native_gather + ".tmp", a.native_gather)
    else:
        t_id = time.time()
        while not os.path.exists(a.native_gather):
            if time.time() - t_id > 120:
                raise SystemExit("no communicator id at %s" % a.native_gather)
            time.sleep(0.01)
    comm = nv.Comm(nctx, open(a.native_gather, "rb").read(), rank, world)
    t1 = time.perf_counter()  # (the communicator's set-up is not part of the gather)
    out = corpus.gather_pcm_native(local_map, len(files), comm, 0, nv.Comm.SELF_P2P if a.self_p2p else 0)
    backend_name = "rccl (nvh_comm_*)"
else:
    out = corpus.gather_pcm(local_map, len(files), rank, world, dist, "cuda:%d" % local, to_host=False)  # stays in HBM
    backend_name = dist.get_backend() if dist is not None else None
torch.cuda.synchronize()
t2 = time.perf_counter()
if rank == 0:
    samples = sum(int(o.numel()) for o in out)
    import hashlib
    hh = hashlib.sha256()
    for o in out:  # every file's PCM in file order, as gathered on rank 0
        hh.update(o.cpu().numpy().tobytes())
    print(json.dumps({"pcm_sha256": hh.hexdigest(), "files": len(files), "n_gpus": world, "backend": backend_name, "workers_per_gpu": a.workers, "decode_s": t1 - t0, "gather_s": t2 - t1,
                      "files_per_s": len(files) / (t2 - t0), "pcm_floats": int(samples),
                      "long_frame_equivalents_per_s": samples / 2 / 1024 / (t2 - t0)}), flush=True)
if a.native_gather:
    comm.close()
    nctx.close()
    if rank == 0 and os.path.exists(a.native_gather):
        os.remove(a.native_gather)
if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
