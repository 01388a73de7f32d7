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


def transcode(files, decode_fn=None, rank=0, world=1, dist=None, device="cpu", gpu=0, workers=16):
    """Decode `files` (list of bytes) file-parallel; rank 0 returns the list of PCM arrays in file order.

    decode_fn(bytes) -> float32 numpy PCM decodes one file; None = this package's GPU path with a pool of `workers`
    host threads on HIP device `gpu` (decode_files_threaded).  Gathered PCM is byte-identical to a single-rank run."""
    shards = lpt_shards([len(f) for f in files], world)
    mine = shards[rank]
    if decode_fn is None:
        pcm = decode_files_threaded([files[i] for i in mine], device=gpu, workers=workers)
        local = {i: np.ascontiguousarray(a, dtype=np.float32) for i, a in zip(mine, pcm)}
    else:
        local = {i: np.ascontiguousarray(decode_fn(files[i]), dtype=np.float32) for i in mine}
    return gather_pcm(local, len(files), rank, world, dist, device)


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
    import queue
    import threading

    from .reader import Context, Stream, demux_ogg_array

    def decode_one(data, ctx):
        # the ReadSamples loop of VorbisReader without its ring buffer: whole look-ahead batches straight into the
        # result (a file that fits one batch is returned without a single extra copy)
        pa = demux_ogg_array(data)
        st = Stream(ctx, pa[0], pa[1], pa[2])
        try:
            if gpu_parse:  # packets parsed by k_parse; shapes outside its limits keep the host parser
                try:
                    st.set_gpu_parse(True)
                except Exception:
                    pass
            chunks, nxt = [], 3
            while True:
                if nxt < len(pa) and not st.position()[2]:
                    nxt += st.push_packets(pa, nxt, batch_frames)
                    last = nxt >= len(pa) or st.position()[2]
                else:
                    last = True
                if last and not st.position()[2]:
                    st.push_end()  # the provider ran dry: drains the carried tail (StreamDecoder.cs:352-356)
                if st.pending()[0]:
                    pcm = st.synth_host()
                    if pcm.size:
                        chunks.append(pcm)
                if last:
                    break
            if not chunks:
                return np.zeros(0, np.float32)
            return chunks[0] if len(chunks) == 1 else np.concatenate(chunks)
        finally:
            st.close()

    out = [None] * len(files)
    errors = []
    order = sorted(range(len(files)), key=lambda i: (-len(files[i]), i))  # longest first: shorter tail
    q = queue.Queue()
    for i in order:
        q.put(i)

    def work():
        ctx = Context(device)
        try:
            while True:
                try:
                    i = q.get_nowait()
                except queue.Empty:
                    return
                try:
                    out[i] = decode_one(files[i], ctx)
                except Exception as e:  # one bad file must not take the pool down; the caller sees it
                    errors.append((i, e))
                    out[i] = np.zeros(0, np.float32)
        finally:
            ctx.close()

    threads = [threading.Thread(target=work, daemon=True) for _ in range(max(1, min(workers, len(files))))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise RuntimeError("decode failed for files %s: %r" % ([i for i, _ in errors], errors[0][1]))
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
        st.push_packet(packets[i], granules[i], flags[i])
        i += 1
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
