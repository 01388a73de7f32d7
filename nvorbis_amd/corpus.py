"""File-parallel corpus transcode across the GPUs of one node (BASELINE.json configs[4]).

Files / streams are independent units (per-stream state only, StreamDecoder.cs:35-39), so the corpus is
sharded across ranks with no data-path collective: one process per GPU, LPT-greedy by compressed
size.  The single exchange step is the final gather of the PCM to rank 0: an all_gather of per-file
sample counts followed by point-to-point transfers of the variable-sized payloads (xGMI is
point-to-point, each peer has one direct link to the root; a ring collective would be the wrong
shape).  Works with backend "nccl" (= RCCL, device tensors) and "gloo" (CPU tensors, used by the CPU
test suite).
"""
import os

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


def gather_pcm(local, nfiles, rank, world, dist=None, device="cpu", to_host=True):
    """Gather {file index -> interleaved float32 PCM} from every rank onto rank 0.

    `local` values are numpy arrays (host) or torch tensors; tensors that already live on `device` are sent from where
    they are -- with backend "nccl" (= RCCL over xGMI) the PCM goes from the decoding GPU's memory to the root GPU's
    memory with no host bounce on either side.  Returns a list of nfiles arrays on rank 0 (None elsewhere): numpy arrays
    when to_host, else torch tensors on `device` (views of the receive buffers).  `dist` is torch.distributed (already
    initialised) or None for a single process.

    Exchange: one all_gather of per-file sample counts, then one flat payload per rank (ascending file index) sent
    point-to-point to the root, all transfers posted as one group (xGMI is point-to-point: every peer has its own
    link to the root, so the root receives from all of them at once; a ring collective would be the wrong shape)."""
    # (a process group of one rank still goes through the backend: the counts' all_gather and the barrier are then the whole
    # exchange -- tests/test_multi_rank_gpu.py initialises RCCL that way on a one-GPU box)
    import_torch = dist is not None
    if not import_torch:
        vals = [local.get(i) for i in range(nfiles)]
        if to_host:
            return [_to_numpy(v) for v in vals]
        return vals
    import torch
    dev = torch.device(device)

    def as_tensor(a):
        if isinstance(a, np.ndarray):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        return a.to(dev) if a.device != dev else a

    counts = torch.zeros(nfiles, dtype=torch.int64)
    for i, a in local.items():
        counts[i] = int(a.size if isinstance(a, np.ndarray) else a.numel())
    counts = counts.to(dev)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    owner, sizes = {}, {}
    for r in range(world):
        c = all_counts[r].cpu().numpy()
        for i in np.nonzero(c)[0]:
            owner[int(i)] = r
            sizes[int(i)] = int(c[i])
    mine = sorted(local.keys())
    parts = [as_tensor(local[i]).reshape(-1) for i in mine]
    out = [None] * nfiles if rank == 0 else None
    ops, bufs = [], {}
    if rank == 0:
        for i, t in zip(mine, parts):
            out[i] = t
        for r in range(1, world):
            idx = sorted(i for i, o in owner.items() if o == r)
            tot = sum(sizes[i] for i in idx)
            if tot > 0:
                buf = torch.empty(tot, dtype=torch.float32, device=dev)
                bufs[r] = (buf, idx)
                ops.append(dist.P2POp(dist.irecv, buf, r))
    else:
        flat = _flat_payload(parts, torch, dev)
        if flat.numel() > 0:
            ops.append(dist.P2POp(dist.isend, flat, 0))
    if ops:
        for q in dist.batch_isend_irecv(ops):  # one group: ncclGroupStart / ncclGroupEnd around all sends / receives
            q.wait()
    if rank == 0:
        for r, (buf, idx) in bufs.items():
            off = 0
            for i in idx:
                out[i] = buf[off:off + sizes[i]]
                off += sizes[i]
        for i in range(nfiles):
            if out[i] is None:
                out[i] = torch.zeros(0, dtype=torch.float32, device=dev)  # files that produced no samples
        if to_host:
            out = [_to_numpy(t) for t in out]
    dist.barrier()
    return out


def gather_pcm_native(local, nfiles, comm, root=0, flags=0):
    """gather_pcm through the library's own RCCL entry points (include/nvorbis_hip.h: nvh_comm_*; nvorbis_amd.Comm) instead
    of torch.distributed -- the exchange a host without Python performs ("Eight GPUs"), used here so that it
    is tested: an all-gather of the per-file float counts, one flat payload per rank (ascending file index) point to point to
    `root`.  `local`: {file index -> float32 torch tensor on this process's GPU}.  Returns, on the root, a list of nfiles
    device tensors (views of the receive buffer, files that produced nothing: empty), None elsewhere."""
    import torch
    mine = [0] * nfiles
    for i, a in local.items():
        mine[i] = int(a.numel())
    every = comm.allgather_i64(mine)  # [world][nfiles]
    totals = [sum(row) for row in every]
    idx = sorted(local.keys())
    dev = local[idx[0]].device if idx else torch.device("cuda", torch.cuda.current_device())
    flat = _flat_payload([local[i].reshape(-1) for i in idx], torch, dev)
    recv = torch.empty(sum(totals) if comm.rank == root else 0, dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)  # the payload was written on torch's stream; the gather runs on the context's
    comm.gather_pcm(flat.data_ptr() if flat.numel() else 0, int(flat.numel()), recv.data_ptr() if recv.numel() else 0, totals, root, flags)
    if comm.rank != root:
        return None
    out = [None] * nfiles
    off = 0
    for r in range(comm.world):
        for i in range(nfiles):
            if every[r][i] > 0:
                out[i] = recv[off:off + every[r][i]]
                off += every[r][i]
    for i in range(nfiles):
        if out[i] is None:
            out[i] = torch.zeros(0, dtype=torch.float32, device=dev)
    return out


def _to_numpy(v):
    if v is None:
        return np.zeros(0, np.float32)
    if isinstance(v, np.ndarray):
        return v
    return v.detach().cpu().numpy()


def _flat_payload(parts, torch, dev):
    """One contiguous tensor holding `parts` back to back; no copy when they already are (slices of one arena)."""
    parts = [p for p in parts if p.numel() > 0]
    if not parts:
        return torch.zeros(0, dtype=torch.float32, device=dev)
    base = parts[0]
    contiguous = True
    ptr = base.data_ptr()
    for p in parts:
        if p.data_ptr() != ptr or not p.is_contiguous():
            contiguous = False
            break
        ptr += p.numel() * 4
    if contiguous and len(parts) > 1:
        total = sum(p.numel() for p in parts)
        try:
            return torch.as_strided(base, (total,), (1,))  # the arena the slices were cut from
        except RuntimeError:
            pass
    return parts[0] if len(parts) == 1 else torch.cat(parts)


def transcode(files, decode_fn=None, rank=0, world=1, dist=None, device="cpu", gpu=0, workers=16, to_host=True):
    """Decode `files` (list of bytes) file-parallel; rank 0 returns the list of PCM arrays in file order.

    decode_fn(bytes) -> float32 numpy PCM decodes one file; None = this package's GPU path with a pool of `workers`
    host threads on HIP device `gpu`, every file's PCM written by the synthesis kernels straight into one device arena
    (decode_files_to_device) and gathered from there.  Gathered PCM is byte-identical to a single-rank run."""
    shards = lpt_shards([len(f) for f in files], world)
    mine = shards[rank]
    if decode_fn is None:
        arena, views = decode_files_to_device([files[i] for i in mine], device=gpu, workers=workers)
        local = {i: v for i, v in zip(mine, views)}
        if device == "cpu":  # host exchange (gloo): one read-back of the whole arena
            host = arena.cpu().numpy()
            off = 0
            for i, v in zip(mine, views):
                local[i] = host[off:off + v.numel()]
                off += v.numel()
    else:
        local = {i: np.ascontiguousarray(decode_fn(files[i]), dtype=np.float32) for i in mine}
    return gather_pcm(local, len(files), rank, world, dist, device, to_host)


def _decode_file_packets(st, pa, batch_frames, sink):
    """The ReadSamples loop of VorbisReader without its ring buffer: whole look-ahead batches handed to `sink(stream)`
    (which synthesises the pending batch wherever it wants the PCM)."""
    nxt = 3
    while True:
        if nxt < len(pa) and not st.position()[2]:
            nxt += st.push_packets(pa, nxt, batch_frames)
            last = nxt >= len(pa) or st.position()[2]
        else:
            last = True
        if last and not st.position()[2]:
            st.push_end()  # the provider ran dry: drains the carried tail (StreamDecoder.cs:352-356)
        if st.pending()[0]:
            sink(st)
        if last:
            break


_MALLOC_TUNED = [False]


def _tune_malloc():
    """Once per process, glibc only: file-sized blocks from the heap instead of from mmap.  The index pass allocates and frees
    buffers of the size of a file per file on every thread; with the allocator's defaults each of them is an mmap, a page fault
    per 4 KB and a munmap under the process's one address-space lock -- until the allocator has raised its own threshold, which
    takes a job: the first index pass of a process took twice (16 CPUs) to four times (8 CPUs) as long as the following ones.
    PROCESS-WIDE state (freed heap is no longer returned to the operating system), therefore OPT-IN: decode_files_to_device(...,
    tune_process=True) or NVH_CORPUS_MALLOPT=1 (bench.py and the corpus tools ask for it); it is logged once on stderr.
    NVH_CORPUS_NO_MALLOPT=1 leaves the allocator alone whoever asks."""
    if _MALLOC_TUNED[0] or os.environ.get("NVH_CORPUS_NO_MALLOPT"):
        return
    _MALLOC_TUNED[0] = True
    import sys
    sys.stderr.write("nvorbis_amd.corpus: glibc allocator tuned for this process (mallopt: M_MMAP_THRESHOLD 1 GiB, M_TRIM_THRESHOLD 2 GiB, "
                     "M_TOP_PAD 256 MiB) -- asked for by tune_process / NVH_CORPUS_MALLOPT=1\n")
    try:
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 1 << 30)         # M_MMAP_THRESHOLD
        libc.mallopt(-1, (1 << 31) - 1)   # M_TRIM_THRESHOLD: freed heap stays with the process
        libc.mallopt(-2, 256 << 20)       # M_TOP_PAD: the heap grows in large steps
    except Exception:
        pass


_WORKER_CONTEXTS = {}  # device -> [Context]: worker contexts kept between jobs (keep_contexts=True)
_CLOSERS = []  # threads that are closing the contexts of finished jobs


def wait_contexts_closed():
    """Block until the worker contexts of finished jobs have been released (they are closed behind the job's return).  Called at
    the start of the next job (its early arena is sized from the device memory that is free) and at interpreter exit (a closer
    still inside hipFree / hipHostFree must not race the runtime's teardown)."""
    while _CLOSERS:
        _CLOSERS.pop().join()


import atexit as _atexit  # noqa: E402
_atexit.register(wait_contexts_closed)


def close_worker_contexts():
    """Release the worker contexts a keep_contexts=True job left behind (streams, block pools, setup caches)."""
    for ctxs in _WORKER_CONTEXTS.values():
        for c in ctxs:
            c.close()
    _WORKER_CONTEXTS.clear()


PARSE_LANES_POOL = 32  # nvh_ctx_set_parse_lanes for the contexts of a worker pool of eight threads and more (see the header; round 6:
                       # 8 -> 32 with the lean walk of the multi-packet parser, whose wavefronts cost the same at 8 or 64 packets)


def _cpu_budget():
    """CPUs this process may really use: the cgroup's quota where there is one (a GPU box's container says 256 CPUs and has 16),
    else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def _warm_contexts(device, nthreads, sample, batch_frames, gpu_parse, parse_lanes, have):
    """Worker contexts made ready while the host-only index pass runs: each opens a stream on `sample` (a file of the job),
    parses and synthesises one look-ahead batch of it into a scratch buffer and closes the stream again -- which leaves behind
    what a cold context spends its first file on: the device's copy of the setup (codebook / Huffman / MDCT tables) and the
    pools of device and page-locked blocks, sized for a whole batch (hipMalloc / hipHostMalloc take milliseconds each and
    the runtime serialises them across threads).  Returns (thread, list) -- the list is filled when the thread has ended;
    entries stay None where something failed (the pool then makes that context itself).  `have`: contexts that exist already
    (keep_contexts) are passed through untouched."""
    import threading

    import torch

    from .reader import Context, Stream, demux_ogg_array
    out = [have[t] if (have is not None and t < len(have)) else None for t in range(nthreads)]

    def one(t, pa):
        ctx = None
        try:
            ctx = Context(device)
            ctx.set_parse_lanes(parse_lanes)
            st = Stream(ctx, pa[0], pa[1], pa[2])
            try:
                if gpu_parse:
                    try:
                        st.set_gpu_parse(True)
                    except Exception:
                        pass
                if len(pa) > 3:
                    st.push_packets(pa, 3, batch_frames)
                need = st.pending()[1] * st.channels
                if st.pending()[0] and need:
                    scratch = torch.empty(need, dtype=torch.float32, device="cuda:%d" % device)
                    st.synth_device(scratch.data_ptr(), need)
                    del scratch  # (closing the stream waits for its work)
            finally:
                st.close()
            out[t] = ctx
        except Exception:
            if ctx is not None:
                ctx.close()

    def run():
        try:
            pa = demux_ogg_array(sample)
        except Exception:
            return
        # a few threads, several contexts each: the runtime serialises most of what they do, and the index pass wants the cores
        todo = [t for t in range(nthreads) if out[t] is None]
        lanes_n = max(1, min(int(os.environ.get("NVH_CORPUS_WARM_THREADS", "4")), len(todo)))

        def some(k):
            for t in todo[k::lanes_n]:
                one(t, pa)

        ts = [threading.Thread(target=some, args=(k,), daemon=True) for k in range(lanes_n)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    th = threading.Thread(target=run, daemon=True)
    th.start()
    return th, out


def _pcm_upper_bound(files):
    """Floats the decode of `files` cannot exceed, from container fields alone: per file (granule position of its last page
    + two maximal blocks) x the channel count of its identification header -- or None when a file does not look like that
    (then the arena waits for the index pass).  Only used to start the arena's allocation early; the index pass decides."""
    total = 0
    for f in files:
        j = f.rfind(b"OggS")
        if j < 0 or j + 27 > len(f) or len(f) < 58 or f[:4] != b"OggS" or f[26] != 1 or f[28:35] != b"\x01vorbis":
            return None
        gran = int.from_bytes(f[j + 6:j + 14], "little", signed=True)
        ch = f[39]
        if gran < 0 or gran > (1 << 40) or ch == 0:
            return None
        total += (gran + 2 * 8192) * ch
    return total


def _early_arena(files, device):
    """(thread, box): box[0] becomes a float32 device tensor of _pcm_upper_bound(files) elements, allocated while the index
    pass runs (the runtime takes ~30 ms per GB of a fresh allocation: 0.6 s for the 21.6 GB of the corpus at its stated size) --
    or stays None: no bound, or a bound beyond half of the free device memory."""
    import threading

    import torch
    box = [None]

    def run():
        try:
            bound = _pcm_upper_bound(files)
            if not bound:
                return
            free, _ = torch.cuda.mem_get_info(device)
            if bound * 4 > free // 2:
                return
            box[0] = torch.empty(bound, dtype=torch.float32, device="cuda:%d" % device)
        except Exception:
            box[0] = None

    th = threading.Thread(target=run, daemon=True)
    th.start()
    return th, box


def _run_pool(n_items, workers, device, fn, need_ctx=True, keep_contexts=False, parse_lanes=0, contexts=None):
    """fn(index, ctx) for every index on a pool of host threads, one nvh_ctx (= one HIP stream) per thread (ctx is None with
    need_ctx=False: host-only work).  keep_contexts: the threads' contexts outlive the call and serve the next one -- a
    context's pools of device / page-locked blocks and its setup cache are warm after its first file, and creating them
    (hipMalloc / hipHostMalloc, serialised by the runtime) is most of a short job's decode pass."""
    import queue
    import threading

    from .reader import Context
    errors = []
    q = queue.Queue()
    for i in range(n_items):
        q.put(i)
    nthreads = max(1, min(workers, n_items))
    kept = _WORKER_CONTEXTS.setdefault(device, []) if (need_ctx and keep_contexts) else None

    def work(t):
        ctx = None
        if need_ctx:
            if kept is not None and t < len(kept) and kept[t] is not None:
                ctx = kept[t]
            elif contexts is not None and t < len(contexts) and contexts[t] is not None:
                ctx = contexts[t]  # made ready by _warm_contexts
                if kept is not None:
                    kept[t] = ctx
            else:
                ctx = Context(device)
                if kept is not None:
                    kept[t] = ctx
            ctx.set_parse_lanes(parse_lanes)
        try:
            while True:
                try:
                    i = q.get_nowait()
                except queue.Empty:
                    return
                try:
                    fn(i, ctx)
                except Exception as e:  # one bad file must not take the pool down; the caller sees it
                    errors.append((i, e))
        finally:
            if ctx is not None and kept is None:
                done_with.append(ctx)

    done_with = []
    if kept is not None:
        while len(kept) < nthreads:
            kept.append(None)
    threads = [threading.Thread(target=work, args=(t,), daemon=True) for t in range(nthreads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if contexts is not None and kept is None:  # warmed contexts no thread took (fewer items than threads)
        for t in range(nthreads, len(contexts)):
            if contexts[t] is not None:
                done_with.append(contexts[t])
    if done_with:
        # The job's results are complete; giving the workers' pools back to the runtime (hipFree / hipHostFree of every block,
        # serialised, ~0.2 s for 16 contexts) does not have to hold the caller up.
        def close_all(ctxs):
            for c in ctxs:
                try:
                    c.close()
                except Exception:
                    pass

        th = threading.Thread(target=close_all, args=(done_with,), daemon=True)
        th.start()
        _CLOSERS[:] = [t for t in _CLOSERS if t.is_alive()]  # (finished closers do not pile up)
        _CLOSERS.append(th)
    return errors


def _index_pass(files, workers, full_index=False):
    """The host-only sizing pass of decode_files_to_device over a pool of threads: (shape, totals, chans, errors) -- per file what
    the index saw of it (packets, payload bytes), the floats its decode emits, its channels; errors as _run_pool returns them."""
    from .reader import Stream, demux_ogg_array, index_ogg_array
    n = len(files)
    shape = [None] * n  # what the index saw of file i: (packets, payload bytes) -- the decode pass's full demultiplex must agree
    totals = [0] * n
    chans = [1] * n

    # Host-only streams of the index pass, per thread and per distinct header triple: nvh_stream_index_packets does not touch the
    # stream's state, and the files of a corpus mostly share their encoder's headers -- one open per thread instead of one per file
    # (five library calls per file become two: the pass is a pool of Python threads, and what bounds it is how often they hand
    # the interpreter lock to each other, not the calls' work: C5's 1004 files on the build box, 4 / 8 threads: 0.33 / 0.55 -> 0.22 / 0.33 s;
    # on a GPU box's 16 CPUs the pass is flat from six threads on -- tools/index_sweep.py: 0.35 s on one thread, 0.081 on six, 0.098 on sixteen).
    import threading
    tl = threading.local()
    opened, opened_lock = [], threading.Lock()

    def index_one(i, ctx):
        # Lacing-only index (reader.index_ogg_array / nvh_ogg_index_packets): page headers and lacing values give the packet list,
        # one byte per packet its block size, the page granule positions the length -- no checksum, no copy of the packets (1 % of the
        # file is read instead of all of it twice).  The checksums are the decode pass's, where the packets are needed anyway.
        def measure(pa):
            key = (pa[0], pa[1], pa[2])
            cache = getattr(tl, "streams", None)
            if cache is None:
                cache = tl.streams = {}
            st = cache.get(key)
            own = st is None
            if own:
                st = Stream(None, key[0], key[1], key[2])
                if len(cache) < 8:
                    cache[key] = st
                    own = False
                    with opened_lock:
                        opened.append(st)
            try:
                totals[i] = int(st.index_total(pa, 3)) * st.channels
                chans[i] = st.channels
            finally:
                if own:
                    st.close()

        if not full_index:
            try:
                pa, payload = index_ogg_array(files[i])
                measure(pa)
                shape[i] = (len(pa), payload)
                return
            except Exception:
                pass  # (a damaged header page, say: the checked demultiplex decides what the file is)
        pa = demux_ogg_array(files[i])
        measure(pa)
        shape[i] = (len(pa), int(pa.offsets[-1]))

    def close_index_streams():
        with opened_lock:
            sts, opened[:] = list(opened), []
        for st in sts:
            try:
                st.close()
            except Exception:
                pass

    errors = _run_pool(n, workers, 0, index_one, need_ctx=False)  # host-only: no GPU context per thread
    close_index_streams()
    return shape, totals, chans, errors


def decode_files_to_device(files, device=0, workers=16, batch_frames=4096, gpu_parse=False, keep_contexts=False, timings=None, tune_process=False):
    """Decode .ogg byte strings on ONE GPU into ONE device arena: returns (arena, views) with views[i] the interleaved
    float32 PCM of files[i] as a slice of `arena` (torch tensors on cuda:<device>), files back to back in list order.

    Two passes over the worker pool: a geometry-only index of every stream (nvh_stream_index_packets: packet type, mode
    number, window flags -- how many samples the serial decoder emits), which sizes the arena; then the decode, whose
    overlap-add kernels write each batch's PCM at its final address (nvh_stream_synth with a device destination).  No
    PCM crosses PCIe.  tune_process: see _tune_malloc (process-wide, opt-in)."""
    import torch

    from .reader import Context, Stream, demux_ogg_array, index_ogg_array
    n = len(files)
    full_index = bool(os.environ.get("NVH_CORPUS_FULL_INDEX"))  # A/B aid: the round-5 index (checksums + a copy of every packet)

    import time
    if tune_process or os.environ.get("NVH_CORPUS_MALLOPT"):  # process-wide allocator settings: opt-in (see _tune_malloc)
        _tune_malloc()
    wait_contexts_closed()  # the job before gives its workers' device / page-locked pools back first
    timing = bool(os.environ.get("NVH_CORPUS_TIMING"))
    if os.environ.get("NVH_CORPUS_BATCH"):  # A/B aid: packets per parse / synthesis batch
        batch_frames = int(os.environ["NVH_CORPUS_BATCH"])
    t_start = time.perf_counter()
    keep = keep_contexts or bool(os.environ.get("NVH_CORPUS_KEEP_CTX"))
    lanes = PARSE_LANES_POOL if (gpu_parse and min(workers, n) >= 8) else 0
    warm_thread, warm = None, None
    if n and not os.environ.get("NVH_CORPUS_NO_WARM"):  # the GPU side of the workers gets ready while the index pass keeps the CPUs busy
        sample = max(range(n), key=lambda i: len(files[i]))
        # (a pool of GPU-parse workers may be twice the CPUs -- its threads wait for the GPU most of the time --; the index pass
        # is host work on a few threads (below), with as many contexts as CPUs warmed beside it -- 32 index threads and 32 contexts on a
        # 16-CPU box made a fresh process's index pass 0.14 -> 0.49 s.  The pool's other threads make their contexts themselves.)
        budget = _cpu_budget()
        nwarm = max(1, min(workers, n))
        if not keep:
            nwarm = min(nwarm, max(8, budget))
        warm_thread, warm = _warm_contexts(device, nwarm, files[sample], batch_frames, gpu_parse, lanes,
                                           _WORKER_CONTEXTS.get(device) if keep else None)
    arena_thread, arena_box = (None, [None]) if (not n or os.environ.get("NVH_CORPUS_NO_WARM")) else _early_arena(files, device)
    # (host-only, no GPU context per thread; three eighths of the CPUs: beyond that its threads only wait for the interpreter lock,
    # and the context warm-up and the arena allocation beside it want cores too)
    shape, totals, chans, errors = _index_pass(files, min(workers, max(4, (3 * _cpu_budget()) // 8)), full_index)
    if arena_thread is not None:
        arena_thread.join()
    if errors:
        if warm_thread is not None:
            warm_thread.join()
            for c in warm:
                if c is not None and not (keep and c in _WORKER_CONTEXTS.get(device, [])):
                    c.close()
        raise RuntimeError("index failed for files %s: %r" % ([i for i, _ in errors], errors[0][1]))
    t_index = time.perf_counter()
    offs = np.zeros(n + 1, np.int64)
    offs[1:] = np.cumsum(totals)
    total = int(offs[-1])
    if arena_box[0] is not None and total >= 1 and total <= int(arena_box[0].numel()) <= total + max(total // 50, 1 << 20):
        # allocated during the index pass from an upper bound (two maximal blocks per file beyond its last granule position): taken
        # when at most 2 % (or 4 MB) of it stays unused -- the slice pins the whole block for the arena's lifetime
        arena = arena_box[0][:total]
    else:
        # short files (a 1-second stereo file's bound is ~37 % above its length), chained or damaged ones: the exact size, and the
        # early block goes back to the device first so that both do not sit in torch's cache next to the library's own pools
        if arena_box[0] is not None:
            arena_box[0] = None
            torch.cuda.empty_cache()
        arena = torch.empty(max(total, 1), dtype=torch.float32, device="cuda:%d" % device)
    torch.cuda.synchronize(device)
    base = arena.data_ptr()
    order = sorted(range(n), key=lambda i: (-len(files[i]), i))  # longest first: shorter tail

    phase = {"open": 0.0, "synth": 0.0, "push": 0.0, "close": 0.0}  # NVH_CORPUS_TIMING: seconds summed over the workers

    redo = []  # files whose checked demultiplex disagrees with the index (a damaged page): decoded again below, the slow way

    def decode_one(k, ctx):
        i = order[k]
        pa = demux_ogg_array(files[i])  # with the page checksums
        if shape[i] != (len(pa), int(pa.offsets[-1])):
            redo.append(i)
            return
        t0 = time.perf_counter()
        st = Stream(ctx, pa[0], pa[1], pa[2])
        try:
            if gpu_parse:
                try:
                    st.set_gpu_parse(True)
                except Exception:
                    pass
            pos = [int(offs[i])]
            t1 = time.perf_counter()
            in_sink = [0.0]

            def sink(s):
                ts = time.perf_counter()
                room = int(offs[i + 1]) - pos[0]
                need = s.pending()[1] * s.channels
                if need > room:
                    raise RuntimeError("file %d produces more than the %d floats its index says" % (i, totals[i]))
                pos[0] += s.synth_device(base + 4 * pos[0], room)
                in_sink[0] += time.perf_counter() - ts

            _decode_file_packets(st, pa, batch_frames, sink)
            if pos[0] != int(offs[i + 1]):
                raise RuntimeError("file %d produced %d floats, its index says %d" % (i, pos[0] - int(offs[i]), totals[i]))
        finally:
            t2 = time.perf_counter()
            st.close()
            if timing:
                t3 = time.perf_counter()
                phase["open"] += t1 - t0
                phase["synth"] += in_sink[0]
                phase["push"] += t2 - t1 - in_sink[0]
                phase["close"] += t3 - t2

    t_alloc = time.perf_counter()
    if warm_thread is not None:
        warm_thread.join()
    t_warm = time.perf_counter()
    errors = _run_pool(n, workers, device, decode_one, keep_contexts=keep, parse_lanes=lanes, contexts=warm)
    if errors:
        raise RuntimeError("decode failed for files %s: %r" % ([order[k] for k, _ in errors], errors[0][1]))
    if timings is not None:  # (a dict of the caller's: where the call's time went)
        timings.update({"index_s": t_index - t_start, "arena_s": t_alloc - t_index, "wait_for_warm_contexts_s": t_warm - t_alloc,
                        "decode_pass_s": time.perf_counter() - t_warm, "parse_lanes": lanes})
    if timing:
        import sys
        sys.stderr.write("decode_files_to_device: demux + index pass %.3f s, arena %.3f s, decode pass %.3f s (%d workers)\n" % (
            t_index - t_start, t_alloc - t_index, time.perf_counter() - t_alloc, workers))
        sys.stderr.write("decode_files_to_device: waited %.3f s for the warmed contexts; summed over the workers: stream open %.3f s, "
                         "push (+ upload and parse on the GPU in GPU-parse mode) %.3f s, synthesis %.3f s, close %.3f s\n" % (
                             t_warm - t_alloc, phase["open"], phase["push"], phase["synth"], phase["close"]))
    views = [arena[int(offs[i]):int(offs[i + 1])] for i in range(n)]
    if redo:
        # A page the index took at its word failed its checksum: the reference's reader drops it and resynchronises, the file's
        # packet list -- and with it its length -- is another.  Such a file gets a tensor of its own (views[i] is then not a slice
        # of the arena; the gather copes), decoded from the checked packet list.
        ctx = Context(device)
        try:
            for i in sorted(redo):
                pa = demux_ogg_array(files[i])
                st = Stream(None, pa[0], pa[1], pa[2])
                try:
                    tot = int(st.index_total(pa, 3)) * st.channels
                finally:
                    st.close()
                own = torch.empty(max(tot, 1), dtype=torch.float32, device="cuda:%d" % device)
                torch.cuda.synchronize(device)
                st = Stream(ctx, pa[0], pa[1], pa[2])
                try:
                    if gpu_parse:
                        try:
                            st.set_gpu_parse(True)
                        except Exception:
                            pass
                    pos = [0]

                    def sink(s_):
                        pos[0] += s_.synth_device(own.data_ptr() + 4 * pos[0], tot - pos[0])

                    _decode_file_packets(st, pa, batch_frames, sink)
                    if pos[0] != tot:
                        raise RuntimeError("file %d produced %d floats, its checked index says %d" % (i, pos[0], tot))
                finally:
                    st.close()
                ctx.synchronize()
                views[i] = own[:tot]
        finally:
            ctx.close()
        if timings is not None:
            timings["files_reindexed"] = sorted(redo)
    return arena, views


def decode_files_threaded(files, device=0, workers=16, batch_frames=4096, gpu_parse=False):
    """Decode a list of .ogg byte strings on ONE GPU with `workers` host threads; returns PCM arrays in file order.

    The bit-serial half of the decoder (Huffman / floor / residue side information, nvorbis_amd/csrc/host_parse.cpp)
    is sequential per stream and bounds a single stream at ~170 k frames/s per host core, far below what the GPU
    synthesises; files are independent, so a worker pool parses them side by side.  Each worker owns one nvh_ctx
    (= one HIP stream), so uploads, kernels and read-backs of different files overlap on the device.  ctypes
    releases the GIL for the duration of every library call and packets are pushed a batch per call
    (nvh_stream_push_packets), so the pool scales with cores.  gpu_parse=True moves the packet parse to the GPU as well
    (kernels_parse.hip): less host work per file, at ~0.8 ms of kernel latency per batch.  Results are byte-identical
    to a serial decode either way."""
    from .reader import Stream, demux_ogg_array
    out = [None] * len(files)
    order = sorted(range(len(files)), key=lambda i: (-len(files[i]), i))  # longest first: shorter tail

    def decode_one(k, ctx):
        i = order[k]
        out[i] = np.zeros(0, np.float32)
        pa = demux_ogg_array(files[i])
        st = Stream(ctx, pa[0], pa[1], pa[2])
        try:
            if gpu_parse:  # packets parsed by k_parse; shapes outside its limits keep the host parser
                try:
                    st.set_gpu_parse(True)
                except Exception:
                    pass
            chunks = []

            def sink(s):
                pcm = s.synth_host()
                if pcm.size:
                    chunks.append(pcm)

            _decode_file_packets(st, pa, batch_frames, sink)
            if chunks:
                out[i] = chunks[0] if len(chunks) == 1 else np.concatenate(chunks)
        finally:
            st.close()

    errors = _run_pool(len(files), workers, device, decode_one,
                       parse_lanes=PARSE_LANES_POOL if (gpu_parse and min(workers, len(files)) >= 8) else 0)
    if errors:
        raise RuntimeError("decode failed for files %s: %r" % ([order[k] for k, _ in errors], errors[0][1]))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# One stream across several GPUs: contiguous chunks of packets with a one-packet lead-in
# ---------------------------------------------------------------------------------------------------------------------
#
# Packets of a stream are independent given the setup; the only coupling between neighbours is the overlap-add of
# adjacent blocks (StreamDecoder.cs:440-445) and three integers of decoder state (_hasPosition, _currentPosition and
# the previous block's geometry).  A decoder that is given packet k-1 first emits nothing for it (the reference's
# "first packet" rule, :446-450 -- the same pre-roll SeekTo uses, :602-623) and from packet k on produces exactly the
# samples the serial decoder produces there.  So a stream is cut into `world` chunks; rank r decodes
# packets [first_r - 1, last_r) and keeps everything it emits; the concatenation is byte-identical to a serial decode.
# What has to be known per cut is integer geometry only: plan_stream_chunks runs the host parser over the packets
# (in its light mode when a GPU context is given: packet type, mode number and window flags only) and records the
# serial decoder's position state and emitted-sample count at every cut.


def _push_range(st, packets, granules, flags, lo, hi):
    """Push packets [lo, hi); stops at the stream's end-of-stream packet.  Returns the index after the last one pushed."""
    from .reader import PacketArray
    i = lo
    if isinstance(packets, PacketArray):
        while i < hi and not st.position()[2]:
            took = st.push_packets(packets, i, hi - i)
            if took == 0:
                break
            i += took
        return i
    while i < hi and not st.position()[2]:
        i += 1  # a packet that makes push_packet raise is consumed all the same (the reference's `packet.Done()`)
        st.push_packet(packets[i - 1], granules[i - 1], flags[i - 1])
    return i


def plan_stream_chunks(packets, granules, flags, world, ctx=None):
    """Cut points for decoding one logical stream (packets[0:3] = its headers) as `world` contiguous chunks.

    Returns a list of dicts, one per chunk (possibly fewer than `world` for very short streams):
      first, last   the chunk emits the samples the serial decoder emits while it processes packets [first, last)
      has_position, position   _hasPosition / _currentPosition of the serial decoder before packet `first`
      emitted0, emitted1       samples per channel the serial decoder has emitted before packet `first` / `last`
                               (emitted1 of the final chunk includes the end-of-stream drain or trim)
    A cut is only placed after a packet that decodes and whose own overlap does not reach into its tail (otherwise the
    lead-in packet alone would not reproduce the tail the next packet overlaps with); the host parser's geometry-only
    index of the stream (nvh_stream_index_packets: packet type, mode number, window flags) says so.  `ctx` is unused
    (the index is host work) and kept for callers that pass their context.
    """
    from .reader import PacketArray, Stream
    pa = PacketArray.from_list(packets, granules, flags)
    n = len(pa)
    st = Stream(None, pa[0], pa[1], pa[2])
    try:
        pos, em, state, total = st.index_packets(pa, 3)
    finally:
        st.close()
    world = max(1, int(world))
    cuts = []  # (first, has_position, position, emitted)
    prev = 3
    for r in range(1, world):
        k = max(3 + ((n - 3) * r) // world, prev + 1, 4)
        # the serial decoder stops pulling packets at _eosFound: no cut at or after that packet
        while k < n and not (state[k - 4] & 8) and not (state[k - 4] & 2):
            k += 1
        if k >= n or (state[k - 4] & 8):
            break
        cuts.append((k, bool(state[k - 4] & 4), int(pos[k - 4]), int(em[k - 4])))
        prev = k
    chunks = []
    bounds = [(3, False, 0, 0)] + cuts
    for i, (first, has, p, e0) in enumerate(bounds):
        last = bounds[i + 1][0] if i + 1 < len(bounds) else n
        e1 = bounds[i + 1][3] if i + 1 < len(bounds) else total
        chunks.append({"first": first, "last": last, "has_position": has, "position": p, "emitted0": e0, "emitted1": e1})
    return chunks


def decode_stream_chunk(ctx, packets, granules, flags, chunk, final, batch_frames=4096, gpu_parse=True, clip=True):
    """Decode one chunk of plan_stream_chunks on the GPU of `ctx`; returns (interleaved float32 PCM, has_clipped).

    The lead-in packet (chunk['first'] - 1) is pushed without granule or flags -- the serial decoder's position state at
    the cut already accounts for them -- then the state is set and the chunk's own packets follow.  Only the stream's
    final chunk ends with the provider running dry (push_end); the others simply stop: the tail of their last block is
    emitted by the next chunk."""
    from .reader import Stream
    from . import native
    st = Stream(ctx, packets[0], packets[1], packets[2])
    try:
        st.set_clip(clip)
        if gpu_parse:
            try:
                st.set_gpu_parse(True)
            except native.NvhError as e:
                if e.code != native.ERR_UNSUPPORTED:
                    raise
        first, last = chunk["first"], chunk["last"]
        if first > 3:
            st.push_packet(packets[first - 1], -1, 0)
        st.set_position_state(chunk["has_position"], chunk["position"])
        want = (chunk["emitted1"] - chunk["emitted0"]) * st.channels
        pcm = np.empty(want, dtype=np.float32)  # every batch is read back straight into its place
        got, cur = 0, first
        while True:
            hi = min(last, cur + batch_frames)
            cur2 = _push_range(st, packets, granules, flags, cur, hi)
            done = cur2 >= last or cur2 < hi or st.position()[2]
            cur = cur2
            if done and final and not st.position()[2]:
                st.push_end()
            if st.pending()[0]:
                need = st.pending()[1] * st.channels
                if got + need > want:
                    raise RuntimeError("chunk [%d, %d) produces more than the %d floats the plan says" % (first, last, want))
                got += st.synth_host(out=pcm[got:]).size
            if done:
                break
        if got != want:
            raise RuntimeError("chunk [%d, %d) produced %d floats, the plan says %d" % (first, last, got, want))
        return pcm, st.has_clipped()
    finally:
        st.close()


def decode_stream_sharded(packets, granules, flags, rank=0, world=1, dist=None, device="cpu", ctx=None, decode_chunk_fn=None,
                          batch_frames=4096):
    """Decode ONE logical stream with `world` ranks (one GPU each): rank r decodes chunk r of plan_stream_chunks, the PCM
    is gathered to rank 0 (gather_pcm: sample counts all_gather + point-to-point payloads) and concatenated there.
    Returns the interleaved PCM on rank 0 (None elsewhere); byte-identical to a serial decode.

    decode_chunk_fn(chunk, final) -> float32 PCM replaces the GPU decode (the CPU test suite passes the oracle)."""
    chunks = plan_stream_chunks(packets, granules, flags, world, ctx)
    local = {}
    for i in range(rank, len(chunks), world):  # fewer chunks than ranks for very short streams
        final = i == len(chunks) - 1
        if decode_chunk_fn is not None:
            pcm = decode_chunk_fn(chunks[i], final)
        else:
            pcm, _ = decode_stream_chunk(ctx, packets, granules, flags, chunks[i], final, batch_frames=batch_frames)
        local[i] = np.ascontiguousarray(pcm, dtype=np.float32)
    parts = gather_pcm(local, len(chunks), rank, world, dist, device)
    if parts is None:
        return None
    return np.concatenate(parts) if parts else np.zeros(0, np.float32)
