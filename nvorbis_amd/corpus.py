"""File-parallel corpus transcode across the GPUs of one node (BASELINE.json configs[4]).

Files / streams are independent units (per-stream state only, StreamDecoder.cs:35-39), so the corpus is
sharded across ranks with no data-path collective: one process per GPU, LPT-greedy by compressed
size.  The single exchange step is the final gather of the PCM to rank 0: an all_gather of per-file
sample counts followed by point-to-point transfers of the variable-sized payloads (xGMI is
point-to-point, each peer has one direct link to the root; a ring collective would be the wrong
shape).  Works with backend "nccl" (= RCCL, device tensors) and "gloo" (CPU tensors, used by the CPU
test suite).
"""
import numpy as np


def lpt_shards(sizes, world):
    """Longest-processing-time-first assignment of item indices to `world` ranks. Deterministic."""
    order = sorted(range(len(sizes)), key=lambda i: (-int(sizes[i]), i))
    load = [0] * world
    shards = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += int(sizes[i])
    for s in shards:
        s.sort()
    return shards


def gather_pcm(local, nfiles, rank, world, dist=None, device="cpu"):
    """Gather {file index -> float32 PCM array} from every rank onto rank 0.

    Returns a list of nfiles arrays on rank 0 (None elsewhere).  `dist` is torch.distributed (already
    initialised) or None for a single process.  `device` is where the exchange buffers live
    ("cuda:<n>" for RCCL over xGMI, "cpu" for gloo).
    """
    if dist is None or world == 1:
        return [local[i] for i in range(nfiles)]
    import torch
    counts = torch.zeros(nfiles, dtype=torch.int64, device=device)
    for i, a in local.items():
        counts[i] = a.size
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    owner = {}
    sizes = {}
    for r in range(world):
        c = all_counts[r].cpu().numpy()
        for i in np.nonzero(c)[0]:
            owner[int(i)] = r
            sizes[int(i)] = int(c[i])
    out = [None] * nfiles if rank == 0 else None
    # one flat payload per rank: deterministic order (ascending file index)
    mine = sorted(local.keys())
    flat = np.concatenate([local[i] for i in mine]) if mine else np.zeros(0, np.float32)
    if rank == 0:
        off = 0
        for i in mine:
            out[i] = flat[off:off + local[i].size].copy()
            off += local[i].size
        reqs, bufs = [], {}
        for r in range(1, world):
            idx = sorted(i for i, o in owner.items() if o == r)
            tot = sum(sizes[i] for i in idx)
            buf = torch.empty(max(tot, 1), dtype=torch.float32, device=device)
            bufs[r] = (buf, idx)
            if tot > 0:
                reqs.append(dist.irecv(buf[:tot], src=r))
        for q in reqs:
            q.wait()
        for r, (buf, idx) in bufs.items():
            host = buf.cpu().numpy()
            off = 0
            for i in idx:
                out[i] = host[off:off + sizes[i]].copy()
                off += sizes[i]
        for i in range(nfiles):
            if out[i] is None:
                out[i] = np.zeros(0, np.float32)  # files that produced no samples
    else:
        if flat.size > 0:
            t = torch.from_numpy(flat).to(device)
            dist.send(t, dst=0)
    dist.barrier()
    return out


def transcode(files, decode_fn, rank=0, world=1, dist=None, device="cpu"):
    """Decode `files` (list of bytes) file-parallel; rank 0 returns the list of PCM arrays in file order.

    decode_fn(bytes) -> float32 numpy PCM.  Gathered PCM is byte-identical to a single-rank run."""
    shards = lpt_shards([len(f) for f in files], world)
    local = {i: np.ascontiguousarray(decode_fn(files[i]), dtype=np.float32) for i in shards[rank]}
    return gather_pcm(local, len(files), rank, world, dist, device)
