#!/usr/bin/env python3
"""Headline benchmark: decoded Vorbis frames/s for the synthesis hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): a batch of 4096 stereo 44.1 kHz long-block frames
(n = 2048, window long/long), Floor1 + Residue2 + coupling, setup and side information taken from the
254 long/long packets of TestFiles/3test.ogg (tests/golden/3test.ogg) tiled to 4096 ("G-real" of
SURVEY 8d).  A step is one pass of the whole hot path (residue adds -> inverse coupling + floor ->
IMDCT + window -> overlap-add + interleave + clip) over that batch, descriptors already resident in
HBM, PCM written to HBM.  Weak scaling: every rank owns one GPU and one such batch, no collective in
the data path.

Launch: python bench.py --gpus N --steps K --warmup W.  For N > 1 the script starts its own ranks (it re-executes
itself under torch.distributed.run, one rank per GPU over RCCL) unless it already runs as a rank (WORLD_SIZE set).

A step is `passes_per_step` passes over the batch, chosen from a calibration run so that the timed region lasts at least
~0.25 s whatever --steps says (a 20-step run of one 45 us pass each would time 0.9 ms of launch jitter); the line states
it in config, and `value` counts every pass.
"""
import argparse
import json
import os
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="profiling builds with phases masked out produce garbage PCM")
    ap.add_argument("--streams", type=int, default=2,
                    help="independent decoder instances (own nvh_ctx / HIP stream / resident batch) the passes rotate over")
    ap.add_argument("--min-timed-ms", type=float, default=250.0, help="lower bound of the timed region (sets passes_per_step)")
    args = ap.parse_args()

    import torch

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly with --gpus N: become N ranks (one per GPU of this node, RCCL over xGMI)
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
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
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: there is no CPU path to measure")
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; reporting n_gpus=%d (the ranks that actually run)\n" % (args.gpus, world, world))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    headers, ll, ch = ll_packets(nv, os.path.join(ROOT, "tests", "golden", "3test.ogg"))
    assert ch == 2 and len(ll) > 0

    # S independent decoder instances, each with its own HIP stream and its own resident 4096-frame batch: a
    # step is one pass of the hot path over one batch, and consecutive steps go to different instances, so the
    # GPU overlaps the tail of one batch's kernels with the head of the next one's (what a corpus transcoder
    # does with independent files).  Instance 0 also provides the serial per-kernel timings below.
    insts = []
    for k in range(max(1, args.streams)):
        ts = torch.cuda.Stream()  # never the legacy default stream: it serialises against every other stream
        ctx_k = nv.Context(local_rank)
        ctx_k.set_hip_stream(ts.cuda_stream)
        stream_k = nv.Stream(ctx_k, headers[0], headers[1], headers[2])
        # priming frame (a first packet emits nothing, StreamDecoder.cs:446-450) goes through a batch of its own
        stream_k.push_packet(ll[(rank * 7 + k * 13) % len(ll)], -1, 0)
        stream_k.synth_host()
        for i in range(FRAMES):
            stream_k.push_packet(ll[(i + rank * 7 + k * 13 + 1) % len(ll)], -1, 0)
        batch_k = stream_k.upload_batch()
        assert batch_k.frames == FRAMES and batch_k.samples == FRAMES * (BLOCK // 2), (batch_k.frames, batch_k.samples)
        pcm_k = torch.empty(batch_k.samples * ch, dtype=torch.float32, device="cuda")
        insts.append((ts, ctx_k, stream_k, batch_k, pcm_k))
    _, ctx, stream, batch, pcm = insts[0]
    cap = pcm.numel()
    nin = len(insts)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_passes(n):
        for i in range(n):
            insts[i % nin][3].synth(insts[i % nin][4].data_ptr(), cap)

    # calibration (untimed): how long one pass takes here, so that a step can be made of enough passes
    run_passes(8)
    barrier()
    t0 = time.perf_counter()
    run_passes(64)
    barrier()
    pass_ms = (time.perf_counter() - t0) / 64 * 1e3
    if dist is not None:
        t = torch.tensor([pass_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        pass_ms = float(t.item())
    passes_per_step = max(1, int(args.min_timed_ms / max(args.steps, 1) / max(pass_ms, 1e-6) + 0.999))

    for _ in range(args.warmup):
        run_passes(passes_per_step)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_passes(passes_per_step)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_passes = args.steps * passes_per_step

    # per-kernel durations, hipEvents on the launch stream (rank 0 reports)
    iters = 50
    total_ms, km = batch.time(pcm.data_ptr(), cap, iters)
    checksum = float(pcm.double().abs().sum().item())
    assert args.no_check or (checksum > 0 and bool(torch.isfinite(pcm).all().item()))

    if rank == 0:
        # the library says which kernel variant sits behind each timing slot ("-" = empty: only event overhead)
        names = batch.kernels()
        live = [k for k in range(4) if names[k] != "-"]
        dom = max(live, key=lambda k: km[k])
        # SURVEY 8d / DESIGN.md: algorithmic bytes of one ch-frame = n/2*4 B spectrum in + n/2*4 B PCM out = 4n B ... per
        # pipeline stage that means: every kernel of the chain moves one n/2-float vector in and one out per ch-frame
        alg_bytes = FRAMES * ch * 4 * BLOCK
        dom_ms = km[dom]
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (tools/profile_round.sh);
        # null when the committed profile does not cover the kernel that ran
        traffic, traffic_source = None, None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                traffic = tj.get("kernels", {}).get(names[dom], {}).get("hbm_bytes")
                if traffic is not None:
                    traffic_source = "profiles/traffic.json: %s (committed rocprofv3 --pmc passes of this command, not measured in this run)" % tj.get("source", "?")
            except Exception:
                traffic = None
        ceiling = copy_ceiling(torch, nv, ctx, insts[0][0])
        step_ms = elapsed / total_passes * 1e3
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
            "data": "synthetic (3test.ogg long/long packets tiled to 4096 frames per GPU, device resident)",
            "config": {"workload": "C2: 4096 stereo long-block (n=2048) frames, Floor1+Residue2+coupling, IMDCT+window+OLA",
                       "frames_per_gpu": FRAMES, "channels": ch, "block": BLOCK, "parallelism": "frame-parallel x%d, %d HIP streams per GPU" % (world, nin),
                       "passes_per_step": passes_per_step, "ms_per_pass": step_ms,
                       "descriptor_bytes_per_frame": batch.descriptor_bytes / FRAMES},
            "kernels_ms": {names[k]: km[k] for k in live},
            "pipeline_ms_events": total_ms / iters,
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": dom_ms,
                         # the same bytes over one whole pass of the pipeline (all kernels, the batches of the %d streams overlapped)
                         "whole_pass_GBps": alg_bytes / (step_ms * 1e-3) / 1e9, "whole_pass_frac": alg_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "copy_ceiling_GBps": ceiling},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(headers, ll)
        print(json.dumps(out), flush=True)
    for _, ctx_k, stream_k, batch_k, _ in insts:
        batch_k.free()
        stream_k.close()
        ctx_k.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
