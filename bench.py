#!/usr/bin/env python3
"""Headline benchmark: decoded Vorbis frames/s for the synthesis hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): a batch of 4096 stereo 44.1 kHz long-block frames
(n = 2048, window long/long), Floor1 + Residue2 + coupling, setup and side information taken from the
254 long/long packets of TestFiles/3test.ogg (tests/golden/3test.ogg) tiled to 4096 ("G-real" of
SURVEY 8d).  A step is one pass of the whole hot path (residue adds -> inverse coupling + floor ->
IMDCT + window -> overlap-add + interleave + clip) over that batch, descriptors already resident in
HBM, PCM written to HBM.  Weak scaling: every rank owns one GPU and one such batch, no collective in
the data path.

What is resident when the clock starts is the packet parser's output: since round 4 the parser's own thread writes the per-frame
slabs the synthesis kernels fetch (nvorbis_amd/csrc/host_slab.cpp: Floor1 unwrap + segment lists, chain-major vector-write
records), so a pass is ALL the GPU work a once-synthesised batch needs -- the same launches the streaming reader runs for every
look-ahead batch (k_synth over the odd frames, k_synth_emit over the even ones); nothing is converted or prepared outside the
timed region.  After the timed region the PCM of every resident batch is hashed and compared with the digest the CPU oracle
produced for the same packets (tests/golden/bench_pcm_digests.json, tools/gen_bench_digests.py): `pcm_digest_ok`.

Launch: python bench.py --gpus N --steps K --warmup W.  For N > 1 the script starts its own ranks (it re-executes
itself under torch.distributed.run, one rank per GPU over RCCL) unless it already runs as a rank (WORLD_SIZE set).

A step is `passes_per_step` passes, chosen from a calibration run so that the timed region lasts at least ~2 s whatever
--steps says (a 20-step run of one 45 us pass each would time 0.9 ms of launch jitter, and an outside observer sampling GPU
activity would see nothing); the line states it in config, and `value` counts every pass.

Working set: the passes rotate over `--streams` decoder instances (default 3; own nvh_ctx / HIP stream each, as a corpus
transcoder's workers have) x R resident 4096-frame batches each, R chosen so that what the passes touch (slabs + the odd frames'
work planes + PCM, ~57 MB per batch) exceeds `--working-set-mib` (default 512 MiB, twice the 256 MiB Infinity Cache): `value`
and `roofline` are the HBM-resident regime.  The same loop over one batch per stream (working set ~170 MB, L3-resident -- what
rounds 1 and 2 reported) is in `roofline.l3_resident`.

`roofline`: the dominant kernel per LAUNCH by hipEvents on its own stream (one stream, outside the overlapped loop), algorithmic
bytes of that launch, HBM traffic from the committed PMC passes; `roofline.unfused`: the same loop in a child process with
NVH_NO_EMIT=1 (the overlap-add in k_ola_compact instead of inside k_synth).  `end_to_end`: the PCIe-inclusive rate of the
boundary, never `value`.

`configs`: the other BASELINE.json workloads on full-depth packets (C2 G-rand, C3 Markov 256/2048, C4 six channels n = 4096
psize 48), kernel-only: one stream by hipEvents, and the sustained rate over three streams -- parity-test cases timed for the
record, not the headline.
"""
import argparse
import datetime
import json
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before the HIP runtime initialises (nvorbis_amd.configure_process says why)
os.environ.setdefault("NVH_CORPUS_MALLOPT", "1")  # this process is a corpus job: the allocator settings of nvorbis_amd.corpus._tune_malloc (opt-in)
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FRAMES = 4096
BLOCK = 2048
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def ll_packets(nv, path):
    """Header packets + the long/long audio packets of an .ogg file (classified by the host parser)."""
    data = open(path, "rb").read()
    pk, gr, fl = nv.demux_ogg(data)
    s = nv.Stream(None, pk[0], pk[1], pk[2])
    ll = []
    for i in range(3, len(pk)):
        before = s.pending()[0]
        s.push_packet(pk[i], -1, 0)
        if s.pending()[0] == before + 1:
            g = s.pending_geometry()[-1]
            if g[0] == BLOCK and g[1] == 0 and g[2] == BLOCK // 2 and g[3] == BLOCK:
                ll.append(pk[i])
    s.close()
    return pk[:3], ll, s.channels


def cpu_baseline(headers, ll, seconds=12.0):
    """The CPU oracle (C restatement of the reference algorithm) on a bounded sample of the same workload: packets -> PCM
    including the bit parse, one core, then every host core with one independent stream per thread."""
    import ctypes as C
    import numpy as np
    from tests import oracle_py
    orc = oracle_py.load()
    L = orc.L
    nframes = 2048
    packets = list(headers) + [ll[i % len(ll)] for i in range(nframes + 1)]
    blob = np.frombuffer(b"".join(packets), dtype=np.uint8)
    offs = np.zeros(len(packets) + 1, np.int64)
    offs[1:] = np.cumsum([len(p) for p in packets])
    gr = np.full(len(packets), -1, np.int64)
    fl = np.zeros(len(packets), np.uint8)

    def decode_once(buf):
        # straight on the oracle's C entry points, large reads: the interpreter (and its lock) is out of the picture
        err = C.c_int(0)
        d = L.orc_open_packets(blob.ctypes.data, offs.ctypes.data, gr.ctypes.data, fl.ctypes.data, len(packets), C.byref(err))
        if not d:
            raise RuntimeError("oracle open failed: %d" % err.value)
        total = 0
        while True:
            n = L.orc_read_samples(d, buf.ctypes.data, buf.size, 0, buf.size)
            if n <= 0:
                break
            total += n
        L.orc_close(d)
        return total

    def worker(budget):
        buf = np.empty(1 << 20, np.float32)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            assert decode_once(buf) >= nframes * (BLOCK // 2) * 2  # + the drained tail of the last block
            n += nframes
        return n, time.perf_counter() - t0

    worker(0.2)  # warm tables
    frames, t_total = worker(seconds)
    out = {"value": frames / t_total, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d x %d stereo n=2048 LL frames of the bench workload (packets -> PCM incl. bit parse), %.1f s" % (
               frames // nframes, nframes, t_total)}
    import concurrent.futures as cf
    nthreads = max(1, os.cpu_count() or 1)
    try:  # a container's CPU quota (cgroup v2) is not visible in cpu_count(): more threads than that only wait
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            nthreads = max(1, min(nthreads, int(float(quota) / float(period) + 0.999)))
    except Exception:
        pass
    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(nthreads) as ex:
        done = sum(r[0] for r in ex.map(worker, [seconds / 2] * nthreads))
    wall = time.perf_counter() - t0
    out["all_cores"] = {"value": done / wall, "unit": "frames/s", "cores": nthreads,
                        "sample": "one such stream per thread, %d threads, %.1f s" % (nthreads, wall)}
    return out


def end_to_end(nv, ctx, headers, ll, frames=32768, rounds=16):
    """The PCIe-inclusive rate of the boundary on the same workload, one host thread (never `value`): packets
    in host memory -> parse -> kernels -> PCM in page-locked host memory, `frames` packets per look-ahead batch.  Host parser with
    a blocking read-back, and GPU packet parser (kernels_parse.hip) with the pipelined read-back (two batches outstanding)."""
    import numpy as np
    pk = [ll[(i + 1) % len(ll)] for i in range(frames)]
    offs = np.zeros(frames + 1, np.int64)
    offs[1:] = np.cumsum([len(p) for p in pk])
    pa = nv.PacketArray(np.frombuffer(b"".join(pk), np.uint8), offs, np.full(frames, -1, np.int64), np.zeros(frames, np.uint8))
    out = {"what": "packets in host memory -> PCM in page-locked host memory, one host thread, %d packets per batch; not `value`" % frames}
    st = nv.Stream(ctx, *headers)
    st.push_packet(ll[0], -1, 0)
    st.synth_host()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        assert st.push_packets(pa, 0, frames) == frames
        st.synth_host(pinned=True)
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    out["host_parser_frames_per_s"] = frames / best
    out["host_parser_kernels"] = [k for k in st.kernels() if k != "-"]
    st.close()
    st = nv.Stream(ctx, *headers)
    st.set_gpu_parse(True)
    st.push_packet(ll[0], -1, 0)
    st.synth_host()
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        outstanding = 0
        for _r in range(rounds):
            assert st.push_packets(pa, 0, frames) == frames
            st.synth_begin()
            outstanding += 1
            if outstanding == 2:
                st.synth_end()
                outstanding -= 1
        while outstanding:
            st.synth_end()
            outstanding -= 1
        dt = (time.perf_counter() - t0) / rounds
        best = dt if best is None or dt < best else best
    out["gpu_parser_pipelined_frames_per_s"] = frames / best
    # (a 32 768-packet batch of this setup: 32 packets per wavefront through the lean walk, the general body over the frames it
    # leaves -- none here --, the rest of the slab one wavefront per packet; nvh_launch.hip: batch_upload_gpu)
    out["gpu_parser_kernels"] = ["k_parse_fetch", "k_parse_slab_f", "k_parse_slab_c", "k_parse_slab_t", "k_parse_links"] + [k for k in st.kernels() if k != "-"]
    out["gpu_parser_pcm_GBps_over_pcie"] = frames * (BLOCK // 2) * 2 * 4 / best / 1e9
    st.close()
    return out


def copy_ceiling(torch, nv, ctx, ts, mib=1024, iters=20):
    """Measured HBM ceiling of this GPU: the library's float4 copy kernel (nvh_measure_copy) over buffers well past the
    256 MiB Infinity Cache; bytes read + bytes written per second, hipEvents on the context's stream."""
    import ctypes as C
    n = mib << 20
    src = torch.empty(n // 4, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    torch.cuda.synchronize()
    ms = C.c_float(0)
    nv.native.check(nv.lib().nvh_measure_copy(ctx._h, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), n, 3, C.byref(ms)), "nvh_measure_copy")
    nv.native.check(nv.lib().nvh_measure_copy(ctx._h, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), n, iters, C.byref(ms)), "nvh_measure_copy")
    ok = bool(torch.equal(src[:4096], dst[:4096]) and torch.equal(src[-4096:], dst[-4096:]))
    del src, dst
    return (2.0 * n * iters) / (ms.value * 1e-3) / 1e9 if ok and ms.value > 0 else None


def make_batches(nv, torch, ctx, headers, audio, ch, frames, count, seed_off=0):
    """`count` resident batches of `frames` frames each on one stream (= one HIP stream): (stream, [(batch, pcm)])."""
    stream = nv.Stream(ctx, headers[0], headers[1], headers[2])
    # priming frame (a first packet emits nothing, StreamDecoder.cs:446-450) goes through a batch of its own
    stream.push_packet(audio[seed_off % len(audio)], -1, 0)
    stream.synth_host()
    out, k = [], seed_off + 1
    for _ in range(count):
        while stream.pending()[0] < frames:
            stream.push_packet(audio[k % len(audio)], -1, 0)
            k += 1
        b = stream.upload_batch()
        pcm = torch.empty(max(b.samples * ch, 1), dtype=torch.float32, device="cuda")
        # synthesise it once right away: a resident batch overlaps its first frame with a snapshot of the stream's carried tail
        # taken at upload, and this launch leaves the batch's last block as the tail the NEXT upload snapshots -- the batches
        # of a stream then decode to exactly what the continuous stream decodes to (the digests of tests/golden compare that)
        b.synth(pcm.data_ptr(), pcm.numel())
        out.append((b, pcm))
    return stream, out


def check_digests(insts, seeds, no_check=False):
    """Hash the PCM every resident batch wrote last and compare with the CPU oracle's digest of the same packets
    (tests/golden/bench_pcm_digests.json; the oracle itself is not run here).  Returns (ok, batches compared)."""
    import hashlib
    path = os.path.join(ROOT, "tests", "golden", "bench_pcm_digests.json")
    want = json.load(open(path))["digests"]
    ok, n, total = True, 0, 0
    for (_, _, _, batches_k), seed in zip(insts, seeds):
        ref = want.get("seed%d" % seed) or []
        total += len(batches_k)
        for j, (b, pcm) in enumerate(batches_k):
            if j >= len(ref):
                break  # no committed digest for this batch: counted in `total`, not in `n` (the line shows both)
            got = hashlib.sha256(pcm.cpu().numpy().tobytes()).hexdigest()
            n += 1
            if got != ref[j]:
                ok = False
                sys.stderr.write("bench.py: PCM digest mismatch (seed %d, batch %d)\n" % (seed, j))
    if n < total:
        sys.stderr.write("bench.py: %d of %d timed batches have no committed digest (tools/gen_bench_digests.py)\n" % (total - n, total))
    return ok, n, total


def config_lines(nv, torch, ctx, root):
    """Kernel-only timings of the other BASELINE.json workloads on full-depth packets (tests/vorbis_encode.py writes them:
    a classification for every partition, a VQ entry for every vector of every cascade stage; SURVEY 8d generators)."""
    import numpy as np
    from tests import vorbis_encode as ve
    data = open(os.path.join(root, "tests", "golden", "3test.ogg"), "rb").read()
    hdr3 = ve.shipped_headers(data)
    S3 = ve.setup_of(hdr3)
    rng = np.random.default_rng(7)
    pool3 = ve.packet_pool(S3, 20260928, per_kind=256)
    out = {}

    def run(key, what, hdr, packets, frames):
        st = nv.Stream(ctx, hdr[0], hdr[1], hdr[2])
        audio = packets[3:]
        st.push_packet(audio[0], -1, 0)
        st.synth_host()
        k = 0
        while st.pending()[0] < frames and k < len(audio) - 1:
            st.push_packet(audio[1 + k], -1, 0)
            k += 1
        geo = st.pending_geometry()
        b = st.upload_batch()
        pcm = torch.empty(max(b.samples * st.channels, 1), dtype=torch.float32, device="cuda")
        b.time(pcm.data_ptr(), pcm.numel(), 5)
        iters = 40
        tot, km = b.time(pcm.data_ptr(), pcm.numel(), iters)
        names = b.kernels()
        live = [i for i in range(4) if names[i] != "-"]
        # per ch-frame n/2 * 4 B of spectrum in + (emitted samples) * 4 B of PCM out
        alg = st.channels * 4 * (sum(int(g[0]) // 2 for g in geo) + b.samples)
        pass_ms = tot / iters
        dom = max(live, key=lambda i: km[i])
        out[key] = {"workload": what, "frames": b.frames, "channels": st.channels, "frames_per_s_kernel_only": b.frames / (pass_ms * 1e-3),
                    "us_per_batch": pass_ms * 1e3, "kernels_us": {names[i]: km[i] * 1e3 for i in live},
                    "algorithmic_bytes_per_batch": alg, "frac_of_hbm_peak_pipeline": alg / (pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "frac_of_hbm_peak_dominant_kernel": alg / (km[dom] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "descriptor_bytes_per_frame": b.descriptor_bytes / max(b.frames, 1),
                    }
        # the same batch on three decoder instances (own HIP streams), passes rotating without synchronisation as in the
        # headline loop: what the GPU sustains when batches of independent streams follow each other (>= 0.5 s, L3-resident)
        extra = []
        for _ in range(2):
            c2 = nv.Context(ctx.device)
            s2 = nv.Stream(c2, hdr[0], hdr[1], hdr[2])
            s2.push_packet(audio[0], -1, 0)
            s2.synth_host()
            for j in range(k):
                s2.push_packet(audio[1 + j], -1, 0)
            b2 = s2.upload_batch()
            extra.append((c2, s2, b2, torch.empty_like(pcm)))
        ring = [(b, pcm)] + [(e[2], e[3]) for e in extra]

        def sync_all():
            ctx.synchronize()
            for e in extra:
                e[0].synchronize()

        def passes(n):
            for i in range(n):
                bb, pp = ring[i % 3]
                bb.synth(pp.data_ptr(), pp.numel())
            sync_all()

        passes(30)
        t0 = time.perf_counter()
        passes(300)
        n3 = max(300, int(0.6 / max(time.perf_counter() - t0, 1e-6) * 300))
        t0 = time.perf_counter()
        passes(n3)
        el3 = time.perf_counter() - t0
        out[key]["frames_per_s_3_streams"] = b.frames * n3 / el3
        out[key]["us_per_batch_3_streams"] = el3 / n3 * 1e6
        for c2, s2, b2, _ in extra:
            b2.free()
            s2.close()
            c2.close()
        b.free()
        st.close()

    p, _ = ve.stream_from_pool(S3, hdr3, pool3, np.ones(4200, dtype=bool), rng)
    run("C2_grand", "C2 G-rand: 4096 stereo n=2048 LL frames, every partition and cascade stage coded", hdr3, p, 4096)
    p, _ = ve.stream_from_pool(S3, hdr3, pool3, ve.markov_kinds(np.random.default_rng(7), 4700), rng)
    run("C3_markov", "C3: 4096 stereo frames, block kinds from the 256/2048 Markov chain (seed 7), full depth", hdr3, p, 4096)
    h4 = ve.c4_headers(hdr3, psize=48)
    S4 = ve.setup_of(h4)
    pool4 = ve.packet_pool(S4, 148, per_kind=128, class_weights=[0] + [1] * 9)
    p, _ = ve.stream_from_pool(S4, h4, pool4, np.ones(2100, dtype=bool), rng)
    run("C4_psize48", "C4: 2048 six-channel n=4096 frames, coupling [(0,2),(3,4)], Residue2 psize 48, full depth", h4, p, 2048)
    return out


def pin_to_gpu_numa(torch, local_rank):
    """Best effort: this process (and the worker threads it starts) onto the CPUs of the NUMA node the GPU hangs off, so that
    N ranks x W parser threads do not fight over one socket.  Returns a short description for the line."""
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
        if node < 0:
            return "numa node unknown (%s)" % bdf
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return "numa node %d has no allowed CPU" % node
        os.sched_setaffinity(0, cpus)
        return "numa node %d, %d CPUs" % (node, len(cpus))
    except Exception as e:  # containers without sysfs, torch builds without the PCI fields: stay where we are
        return "not pinned (%s)" % (repr(e)[:80])


def c5_block(nv, torch, dist, rank, world, local_rank, scale, workers, share_gpu):
    """BASELINE.json configs[4] inside the bench line: the 1004-file corpus (tests/c5_corpus.py, length scale `scale`) sharded
    file-parallel over the ranks (LPT by compressed size, no data-path collective), decoded by every rank into one device arena
    with the GPU packet parser, then the north star's ONE collective: the gather of the PCM to rank 0, device to device
    (nvorbis_amd.corpus.gather_pcm: all_gather of the counts + grouped point-to-point payloads over RCCL / xGMI).

    Outside the timed regions, and sized so that eight ranks on one host stay cheap: the shard plan comes from the compressed
    sizes in the committed digest file (`ogg_bytes`), so a rank BUILDS ONLY ITS OWN SHARD (child interpreters, no GPU); every
    rank checks its own files -- the .ogg bytes and the SHA-256 of the PCM against the oracle's committed digests
    (tests/golden/c5_digests_scale*.json), hashed by a few threads out of page-locked staging buffers -- BEFORE the gather; and
    the gather itself is checked on the device: two 64-bit word sums per file, taken by the owner before and by the root after
    (a checksum of checksums: the root never copies 21.6 GB to the host to hash it on one core)."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    from nvorbis_amd import corpus
    from tests import c5_corpus
    t_setup = time.perf_counter()
    dig = c5_corpus.load_digests(scale)
    nfiles = c5_corpus.n_files()
    cpus = len(os.sched_getaffinity(0))
    if dig is not None and "ogg_bytes" in dig:
        sizes = [int(x) for x in dig["ogg_bytes"]]
        shards = corpus.lpt_shards(sizes, world)
        mine = shards[rank]
        per_rank = max(1, cpus // (world if (world > 1 and cpus >= 2 * world) else 1))
        built = c5_corpus.build_subset(mine, scale, procs=min(8, per_rank))
    else:  # no committed sizes for this scale: every rank builds the list to learn them
        every = c5_corpus.build_files(scale)
        sizes = [len(f) for f in every]
        shards = corpus.lpt_shards(sizes, world)
        mine = shards[rank]
        built = {i: every[i] for i in mine}
        del every
    my_files = [built[i] for i in mine]
    input_ok = all(len(f) == sizes[i] for f, i in zip(my_files, mine))
    setup_s = time.perf_counter() - t_setup

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def over_ranks(x, op):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=getattr(dist.ReduceOp, op))
        return float(t.item())

    def word_sums(tensors):
        """[n, 2] int64 on the device: the sum of a file's 32-bit words and of its even-indexed words (wrapping)."""
        out = torch.zeros((len(tensors), 2), dtype=torch.int64, device=dev)
        for k, v in enumerate(tensors):
            if v.numel():
                w = v.reshape(-1).view(torch.int32)
                out[k, 0] = w.sum(dtype=torch.int64)
                out[k, 1] = w[::2].sum(dtype=torch.int64)
        return out

    dev = "cuda:%d" % local_rank
    barrier()
    t0 = time.perf_counter()
    passes = {}
    arena, views = corpus.decode_files_to_device(my_files, device=local_rank, workers=workers, gpu_parse=True, timings=passes)
    torch.cuda.synchronize()
    decode_mine = time.perf_counter() - t0
    decode_s, decode_min = over_ranks(decode_mine, "MAX"), over_ranks(decode_mine, "MIN")
    index_max = over_ranks(float(passes.get("index_s", 0.0)), "MAX")
    local_map = {i: v for i, v in zip(mine, views)}

    # ---- this rank's files against the oracle's digests (untimed) ----
    t_chk = time.perf_counter()
    ok_mine, checked_mine = 1.0, 0
    if dig is not None:
        rows = dig["digests"]
        nthr = max(1, min(8, cpus))
        longest = max([int(v.numel()) for v in views] + [1])
        stage = [torch.empty(longest, dtype=torch.float32).pin_memory() for _ in range(nthr)]
        free = list(range(nthr))

        def check(k):
            i, v = mine[k], views[k]
            want = rows[i]
            if c5_corpus.file_digest(my_files[k]) != want[0] or int(v.numel()) != want[1]:
                return False
            slot = free.pop()
            try:
                h = stage[slot][:v.numel()]
                h.copy_(v)
                return hashlib.sha256(h.numpy().view("uint8").data).hexdigest() == want[2]
            finally:
                free.append(slot)

        with ThreadPoolExecutor(nthr) as ex:
            res = list(ex.map(check, range(len(mine))))
        ok_mine, checked_mine = float(all(res)), len(res)
        del stage
    sums_mine = word_sums(views)
    check_s = time.perf_counter() - t_chk
    sha_ok = over_ranks(ok_mine, "MIN") == 1.0 if dig is not None else None
    checked = int(over_ranks(float(checked_mine), "SUM"))
    inputs_ok = over_ranks(float(input_ok), "MIN") == 1.0
    # every file's two sums as its owner saw them
    table = torch.zeros((nfiles, 2), dtype=torch.int64, device=dev)
    if mine:
        table[torch.tensor(mine, device=dev)] = sums_mine
    if dist is not None:
        dist.all_reduce(table)

    barrier()
    t1 = time.perf_counter()
    out = corpus.gather_pcm(local_map, nfiles, rank, world, dist, dev, to_host=False)  # stays in HBM
    torch.cuda.synchronize()
    gather_s = over_ranks(time.perf_counter() - t1, "MAX")
    block = None
    if rank == 0:
        floats = sum(int(o.numel()) for o in out)
        remote = floats - sum(int(v.numel()) for v in views)  # what crossed a link
        gathered_ok = bool(torch.equal(word_sums(out), table)) and (dig is None or [int(o.numel()) for o in out] == [r[1] for r in dig["digests"]])
        block = {"what": "C5: %d-file corpus at length scale %g, LPT shard over %d rank(s), GPU packet parser, one device arena per rank, "
                         "then the gather of all PCM to rank 0 (device to device)" % (nfiles, scale, world),
                 "files": nfiles, "scale": scale, "workers_per_rank": workers,
                 "decode_s": decode_s, "decode_s_min_rank": decode_min, "index_s_max_rank": index_max, "decode_passes_rank0": passes,
                 "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "gather_s": gather_s, "pcm_bytes": floats * 4, "gathered_bytes_over_links": remote * 4,
                 "gather_GBps": (remote * 4 / gather_s / 1e9) if (world > 1 and gather_s > 0) else None,
                 "gather_bound": "each sender's one direct xGMI link to the root, ~153 GB/s; the root receives from all of them at once",
                 "long_frame_equivalents_per_s": floats / 2 / 1024 / (decode_s + gather_s),
                 "pcm_sha256_ok": (sha_ok and inputs_ok and gathered_ok) if dig is not None else None, "files_checked": checked,
                 "check": {"per_rank_sha256_ok": sha_ok, "inputs_ok": inputs_ok, "gathered_word_sums_ok": gathered_ok,
                           "how": "every rank: SHA-256 of its own files' PCM against the committed oracle digests before the gather; "
                                  "root: two 64-bit word sums per gathered file against the owner's, on the device"},
                 "untimed_rank0": {"setup_s": setup_s, "check_s": check_s, "files_built": len(mine)},
                 "digests": os.path.relpath(c5_corpus.digest_path(scale), ROOT) if dig is not None else None,
                 "note": ("NVH_BENCH_SHARE_GPU: every rank on ONE GPU over gloo -- a code-path check, not a transfer rate" if share_gpu else None)}
    del out, views, arena
    barrier()
    return block


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the C2 G-rand / C3 / C4 kernel-only lines")
    ap.add_argument("--no-check", action="store_true", help="profiling builds with phases masked out produce garbage PCM")
    ap.add_argument("--c5-scale", type=float, default=1.0,
                    help="length scale of the corpus block (1.0 = BASELINE's stated size, the default: 3.3 GB of Ogg -> 21.6 GB of PCM, about "
                         "half a minute with its generation and SHA-256 check; 0.1: a tenth; 0: no block)")
    ap.add_argument("--c5-workers", type=int, default=0, help="parser threads per rank for the corpus block (0: twice the CPUs this rank may use, at most 32)")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent decoder instances (own nvh_ctx / HIP stream) the passes rotate over")
    ap.add_argument("--working-set-mib", type=float, default=512.0,
                    help="lower bound of what the rotating passes touch (resident batches per stream follow from it)")
    ap.add_argument("--min-timed-ms", type=float, default=2000.0, help="lower bound of the timed region (sets passes_per_step)")
    ap.add_argument("--no-unfused", action="store_true",
                    help="skip the short NVH_NO_EMIT=1 run (overlap-add in k_ola_compact) that roofline.unfused reports beside the fused kernels")
    args = ap.parse_args()

    import torch

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly with --gpus N: become N ranks (one per GPU of this node, RCCL over xGMI)
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus and not os.environ.get("NVH_BENCH_SHARE_GPU"):
            sys.stderr.write("bench.py --gpus %d: this node shows %d HIP device(s); the multi-GPU run needs %d\n" % (args.gpus, have, args.gpus))
            raise SystemExit(2)
        import socket
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])

    import nvorbis_amd as nv

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Development aid (the builder has no multi-GPU box): NVH_BENCH_SHARE_GPU=1 puts every rank on device 0 and rendezvouses
    # over gloo, so that the N > 1 code path -- barriers, MAX over ranks, rank-0 reporting -- can be exercised on one GPU.
    # The line it prints says so in `data`; it is not a scaling measurement.
    share_gpu = bool(os.environ.get("NVH_BENCH_SHARE_GPU"))
    if share_gpu:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: there is no CPU path to measure")
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; reporting n_gpus=%d (the ranks that actually run)\n" % (args.gpus, world, world))
    torch.cuda.set_device(local_rank)
    # N ranks on one host: each rank's parser threads (the corpus block) on the CPUs next to its GPU
    pinned = pin_to_gpu_numa(torch, local_rank) if (world > 1 and not share_gpu) else "not pinned (one rank)"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=300))
        else:
            # (a bounded collective timeout: a rank that falls out of the corpus gather turns into an error on the others within
            # minutes instead of a hang; the headline's timed region is over by then and the line is still printed)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=300))

    headers, ll, ch = ll_packets(nv, os.path.join(ROOT, "tests", "golden", "3test.ogg"))
    assert ch == 2 and len(ll) > 0

    # S decoder instances, each with its own HIP stream, each holding R resident 4096-frame batches (own descriptors, work
    # planes and PCM): a pass is the hot path over one batch; consecutive passes go to different instances, so the GPU
    # overlaps the tail of one batch's kernels with the head of the next one's (what a corpus transcoder does with
    # independent files), and a batch is revisited only after everything else -- > 2x the Infinity Cache -- went by.
    nin = max(1, args.streams)
    # descriptors / slabs + compact work planes + PCM, touched per pass; with paired emission (kernels_synth.hip: the even frames
    # overlap-add from registers) only the odd frames' planes are written and read
    fused = not os.environ.get("NVH_NO_EMIT")
    per_batch_bytes = FRAMES * (4400 + (ch * (BLOCK // 2) * 4) * (3 if fused else 4) // 2)
    reps = max(1, int(args.working_set_mib * (1 << 20) / (nin * per_batch_bytes) + 0.999))
    insts, seeds = [], []
    for k in range(nin):
        # the context's own HIP stream (hipStreamNonBlocking, created by nvh_ctx_create): never the legacy default stream, which
        # serialises against every other stream.  Three instances: sustained (>= 1.5 s) HBM-resident rates measured with paired
        # emission 135 / 155 / 135 / 148 / 148 M frames/s for 2 / 3 / 4 / 6 / 8 streams (round-3 sweep), without it
        # 129 / 132 / 124 / 129 / 129 M.
        ts = None
        ctx_k = nv.Context(local_rank)
        seeds.append(rank * 7 + k * 13)
        stream_k, batches_k = make_batches(nv, torch, ctx_k, headers, ll, ch, FRAMES, reps, seed_off=seeds[-1])
        for b, _ in batches_k:
            assert b.frames == FRAMES and b.samples == FRAMES * (BLOCK // 2), (b.frames, b.samples)
        insts.append((ts, ctx_k, stream_k, batches_k))
    _, ctx, stream, batches0 = insts[0]
    batch, pcm = batches0[0]
    cap = pcm.numel()
    touched = sum(b.descriptor_bytes + (3 if fused else 4) * b.samples * ch * 4 // 2 for _, _, _, bk in insts for b, _ in bk)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def make_order(r):
        return [insts[k][3][j] for j in range(r) for k in range(nin)]

    def timed(order, steps, warmup, min_ms):
        """`steps` steps of passes_per_step passes rotating over `order`; returns (elapsed s, passes, passes_per_step)."""
        pos = [0]

        def run_passes(n):
            i = pos[0]
            for _ in range(n):
                b, p = order[i % len(order)]
                b.synth(p.data_ptr(), cap)
                i += 1
            pos[0] = i

        run_passes(2 * len(order))  # calibration (untimed): how long one pass takes here
        barrier()
        t0 = time.perf_counter()
        run_passes(64)
        barrier()
        pass_ms = (time.perf_counter() - t0) / 64 * 1e3
        if dist is not None:
            t = torch.tensor([pass_ms], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            pass_ms = float(t.item())
        pps = max(1, int(1.35 * min_ms / max(steps, 1) / max(pass_ms, 1e-6) + 0.999))  # 35 % margin: the calibration passes are slower than steady state
        for _ in range(warmup):
            run_passes(pps)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            run_passes(pps)
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, steps * pps, pps

    # ---- the timed region of the headline: W warmup steps, exactly K steps, working set past the Infinity Cache ----
    elapsed, total_passes, passes_per_step = timed(make_order(reps), args.steps, args.warmup, args.min_timed_ms)
    # the same loop over one batch per stream (L3-resident, the regime of the round-1 / round-2 lines), shorter
    el_l3, passes_l3, _ = timed(make_order(1), max(args.steps // 4, 1), max(args.warmup // 4, 1), args.min_timed_ms / 4)

    # per-kernel durations, hipEvents on the launch stream (rank 0 reports): rotating over instance 0's batches one launch
    # at a time (HBM-resident when reps > 1) and repeated on one batch (L3-resident)
    names = None
    km = [0.0] * 4
    total_ms, rounds = 0.0, max(1, 48 // len(batches0))
    for b, p in batches0:
        b.time(p.data_ptr(), cap, 1)
    for _ in range(rounds):
        for b, p in batches0:
            t, k4 = b.time(p.data_ptr(), cap, 1)
            total_ms += t
            km = [x + y for x, y in zip(km, k4)]
    n_ev = rounds * len(batches0)
    km = [x / n_ev for x in km]
    total_ms /= n_ev
    total_l3, km_l3 = batch.time(pcm.data_ptr(), cap, 50)
    checksum = float(pcm.double().abs().sum().item())
    assert args.no_check or (checksum > 0 and bool(torch.isfinite(pcm).all().item()))
    # every resident batch's PCM against the CPU oracle's digest of the same packets: a line cannot come from kernels that write
    # wrong samples (every rank checks its own; rank 0 reports the AND)
    torch.cuda.synchronize()
    digest_ok, digest_n, digest_total = check_digests(insts, seeds)
    if dist is not None:
        t = torch.tensor([1.0 if digest_ok else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        digest_ok = bool(t.item() > 0.5)
    if not (digest_ok and digest_n > 0) and not args.no_check:
        raise SystemExit("bench.py: the PCM of the timed batches does not match the oracle's digests -- no line")

    # ---- BASELINE configs[4] beside the headline: the file-parallel corpus shard and the ONE collective of the north star, the
    # gather of the PCM (every rank takes part; rank 0 reports).  Never part of `value`.
    c5 = None
    if args.c5_scale > 0:
        # (GPU-parse workers wait for the GPU most of the time: twice the 16 cores a box's container has -- decode pass 0.48 -> 0.38 s)
        workers = args.c5_workers or max(1, min(32, 2 * len(os.sched_getaffinity(0))))
        try:
            c5 = c5_block(nv, torch, dist, rank, world, local_rank, args.c5_scale, workers, share_gpu)
            if c5 is not None:
                c5["cpu_pinning"] = pinned
        except Exception as e:  # never fatal for the headline, which is measured and verified above: the block says what failed
            sys.stderr.write("bench.py: corpus block failed on rank %d: %r\n" % (rank, e))
            c5 = {"error": repr(e)[:300], "rank": rank}

    if rank == 0:
        # the library says which kernel variant sits behind each timing slot ("-" = empty: only event overhead)
        names = batch.kernels()
        live = [k for k in range(4) if names[k] != "-"]
        dom = max(live, key=lambda k: km[k])
        # SURVEY 8d / DESIGN.md: algorithmic bytes of one ch-frame = n/2*4 B spectrum in + n/2*4 B PCM out = 4n B ... per
        # pipeline stage that means: every kernel of the chain moves one n/2-float vector in and one out per ch-frame
        alg_bytes = FRAMES * ch * 4 * BLOCK
        # paired emission: k_synth runs twice per pass, each launch over half of the batch's frames (the library times the two
        # together); per LAUNCH, like rocprofv3's average and the PMC traffic: half the bytes, half the duration
        launches = 2 if names[dom] in ("k_synth+k_synth_emit", "k_synth_group2", "k_synth_group4") else 1
        dom_ms = km[dom]
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (tools/profile_round.sh);
        # null when the committed profile does not cover the kernel that ran
        traffic, traffic_source, traffic_build, rocprof_us = None, None, None, None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                tk = tj.get("kernels", {})
                if launches == 2:
                    # paired emission: the pass is one launch over the odd frames (k_synth_tail when the last block is odd,
                    # else k_synth) and one of k_synth_emit over the even frames; per launch = their mean
                    if names[dom].startswith("k_synth_group"):
                        # frame groups: both launches are the same kernel (rocprofv3's row is the mean over the two)
                        traffic = tk.get(names[dom], {}).get("hbm_bytes")
                    else:
                        odd = tk.get("k_synth_tail") or tk.get("k_synth")
                        even = tk.get("k_synth_emit")
                        traffic = (odd["hbm_bytes"] + even["hbm_bytes"]) / 2 if odd and even else None
                else:
                    traffic = tk.get(names[dom], {}).get("hbm_bytes")
                traffic_build = tj.get("build")
                # per-kernel average durations of the committed rocprofv3 --kernel-trace pass (the same three-stream loop)
                rocprof_us = {k: round(v["avg_us"], 3) for k, v in tk.items() if k.startswith("k_synth") and "avg_us" in v}
                rocprof_us["source"] = tj.get("source")
                if traffic is not None:
                    traffic_source = "profiles/traffic.json: %s (committed rocprofv3 --pmc passes of this command, not measured in this run)" % tj.get("source", "?")
            except Exception:
                traffic = None
        ceiling = copy_ceiling(torch, nv, ctx, insts[0][0])
        # the same workload with the overlap-add left to k_ola_compact (NVH_NO_EMIT=1: the library reads its switches once, so
        # a child process), short: per-kernel durations of the unfused chain next to the fused kernel's
        unfused = None
        if fused and world == 1 and not args.no_unfused:
            import subprocess
            env = dict(os.environ)
            env["NVH_NO_EMIT"] = "1"
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", "40", "--warmup", "4", "--min-timed-ms", "400",
                                    "--streams", str(nin), "--working-set-mib", str(args.working_set_mib), "--no-configs",
                                    "--no-cpu-baseline", "--no-unfused"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                   text=True, timeout=600)
                uj = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                unfused = {"what": "NVH_NO_EMIT=1: every overlap-add in k_ola_compact (the path of GPU-parsed batches and of more than two channels); same loop, 40 steps",
                           "frames_per_s": uj["value"], "ms_per_pass": uj["config"]["ms_per_pass"], "kernels_ms": uj["kernels_ms"],
                           "frac": uj["roofline"]["frac"], "whole_pass_frac": uj["roofline"]["whole_pass_frac"]}
            except Exception as e:  # the headline does not depend on it
                unfused = {"error": repr(e)[:200]}
        step_ms = elapsed / total_passes * 1e3
        l3_ms = el_l3 / passes_l3 * 1e3
        out = {
            "metric": "decoded Vorbis frames/sec (44.1 kHz stereo long-block)",
            "value": world * FRAMES * total_passes / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (3test.ogg long/long packets tiled to %d resident batches of 4096 frames per GPU)" % (nin * reps) +
                    (" -- NVH_BENCH_SHARE_GPU: all %d ranks on ONE GPU over gloo, a code-path check, not a scaling number" % world if share_gpu else ""),
            "config": {"workload": "C2: 4096 stereo long-block (n=2048) frames, Floor1+Residue2+coupling, IMDCT+window+OLA",
                       "frames_per_gpu": FRAMES, "channels": ch, "block": BLOCK,
                       "parallelism": "frame-parallel x%d, %d HIP streams per GPU x %d resident batches each" % (world, nin, reps),
                       "passes_per_step": passes_per_step, "ms_per_pass": step_ms, "timed_region_s": elapsed,
                       "working_set_MiB": touched / (1 << 20),
                       "descriptor_bytes_per_frame": batch.descriptor_bytes / FRAMES},
            "kernels_ms": {names[k]: km[k] for k in live},
            "pipeline_ms_events": total_ms,
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "traffic_build_matches": (traffic_build == nv.native.build_id()) if traffic is not None else None,
                         "algorithmic_bytes_per_launch": alg_bytes // launches, "avg_launch_ms": dom_ms / launches,
                         "launches_per_pass": launches,
                         "kernel_scope": (("k_synth_group2 = residue + floor + inverse MDCT + window / overlap-add / clip / interleave of two consecutive frames per "
                                           "workgroup (the overlap inside a group on chip), two launches per pass (the groups with an odd index, then the even ones, "
                                           "which also emit the overlaps between groups), timed together by hipEvents on ONE stream"
                                           if names[dom].startswith("k_synth_group") else
                                           "k_synth+k_synth_emit = residue + floor + inverse MDCT + (paired emission) window / overlap-add / clip / interleave of the "
                                           "steady-state frames, two launches per pass (odd frames, then the emitting even frames), timed together, one stream")
                                          if fused else "k_synth = residue + floor + inverse MDCT; overlap-add in k_ola_compact"),
                         # how to read the numbers: `frac` / `achieved` / `avg_launch_ms` are ONE decoder instance (what a ReadSamples caller's
                         # kernels run at); `value` and `whole_pass_frac` are the timed loop over `streams` instances, whose launches overlap
                         "one_stream": {"pass_us": dom_ms * 1e3, "frames_per_s": FRAMES / (dom_ms * 1e-3), "frac": achieved / HBM_PEAK_GBPS},
                         "timed_loop": {"streams": nin, "pass_us": step_ms * 1e3, "frames_per_s": FRAMES / (step_ms * 1e-3),
                                        "frac": alg_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                         "rocprof_avg_us": rocprof_us,
                         "unfused": unfused,
                         "working_set_MiB": touched / (1 << 20), "regime": "HBM-resident: a batch is revisited after %.0f MiB went by (Infinity Cache: 256 MiB)" % (touched / (1 << 20)),
                         # the same bytes over one whole pass of the pipeline (all kernels, the batches of the streams overlapped)
                         "whole_pass_GBps": alg_bytes / (step_ms * 1e-3) / 1e9, "whole_pass_frac": alg_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "l3_resident": {"working_set_MiB": nin * per_batch_bytes / (1 << 20), "frames_per_s": world * FRAMES * passes_l3 / el_l3,
                                         "ms_per_pass": l3_ms, "kernels_ms": {names[k]: km_l3[k] for k in live},
                                         "frac": alg_bytes / (km_l3[dom] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                         "whole_pass_frac": alg_bytes / (l3_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                         "copy_ceiling_GBps": ceiling},
            "pcm_digest_ok": digest_ok, "pcm_digests_checked": digest_n, "pcm_batches_timed": digest_total,
            # what the ranks talked through: barriers / MAX of the timings in the headline, the PCM gather in `c5`
            "collective": ({"backend": dist.get_backend(), "world": dist.get_world_size(),
                            "library": "RCCL over xGMI (torch.distributed backend nccl)" if dist.get_backend() == "nccl" else "gloo (CPU): code-path check only"}
                           if dist is not None else {"backend": None, "world": 1}),
            "c5": c5,
            "decode_path": "resident input = the host packet parser's output (per-frame slabs, host_slab.cpp); a pass = every kernel a "
                           "once-synthesised batch needs (the streaming reader runs the same launches per look-ahead batch); no prepare / "
                           "conversion kernel exists for host-parsed batches",
            "build": nv.native.build_id(),
        }
        if world == 1 and not args.no_configs:
            out["configs"] = config_lines(nv, torch, ctx, ROOT)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["end_to_end"] = end_to_end(nv, ctx, headers, ll)
            except Exception as e:  # reported beside the headline, never part of it
                out["end_to_end"] = {"error": repr(e)[:200]}
            out["cpu_baseline"] = cpu_baseline(headers, ll)
        print(json.dumps(out), flush=True)
    for _, ctx_k, stream_k, batches_k in insts:
        for b, _ in batches_k:
            b.free()
        stream_k.close()
        ctx_k.close()
    if dist is not None:
        try:  # (after a failed corpus block the communicator may be gone: the line is out, leave quietly)
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:
            sys.stderr.write("bench.py: rank %d: %r at shutdown\n" % (rank, e))


if __name__ == "__main__":
    main()
