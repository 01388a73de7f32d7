"""Where the C5 decode pass's time goes on one GPU, and what the parse launch shape / batch size / worker count do to it.

  python tools/c5_sweep.py                 (GPU box; the corpus is generated once and cached under /tmp)

Each case is a child process (the NVH_* switches are read once when the library is loaded): decode_files_to_device on the corpus at
--scale, GPU parser, with NVH_CORPUS_TIMING and NVH_TIME_UPLOAD; the child sums the per-batch host prep / enqueue / wait-for-k_parse
lines over all workers.  No digest check here (tools/corpus_c5.py and the suite do that)."""
import argparse
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before the HIP runtime initialises (nvorbis_amd.configure_process says why)
os.environ.setdefault("NVH_CORPUS_MALLOPT", "1")  # this process is a corpus job: the allocator settings of nvorbis_amd.corpus._tune_malloc (opt-in)
import pickle
import re
import subprocess
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CACHE = "/tmp/c5_files_%g.pkl"


def files_for(scale):
    p = CACHE % scale
    if os.path.exists(p):
        return pickle.load(open(p, "rb"))
    from tests import c5_corpus
    files = c5_corpus.build_files(scale)
    pickle.dump(files, open(p, "wb"), protocol=4)
    return files


def child(scale, workers, gpu_parse, reps):
    import torch

    from nvorbis_amd import corpus
    files = files_for(scale)
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        arena, views = corpus.decode_files_to_device(files, device=0, workers=workers, gpu_parse=gpu_parse)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        print("REP_S %.4f" % dt, flush=True)
        del arena, views
        if os.environ.get("NVH_SWEEP_EMPTY_CACHE"):  # every rep decodes into a freshly allocated arena
            torch.cuda.empty_cache()
    print("DECODE_S %.4f" % best, flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--host-parse", action="store_true")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--cases", default="")
    a = ap.parse_args()
    if a.child:
        child(a.scale, a.workers, not a.host_parse, a.reps)
        sys.exit(0)
    print("host cpus: %s" % os.cpu_count(), flush=True)
    t0 = time.time()
    files_for(a.scale)
    print("corpus ready in %.1f s" % (time.time() - t0), flush=True)
    # (workers, lanes, waves, batch, host_parse)
    cases = [(16, 0, 0, 0, 0), (16, 2, 0, 0, 0), (16, 4, 0, 0, 0), (16, 8, 0, 0, 0), (16, 16, 0, 0, 0),
             (32, 0, 0, 0, 0), (32, 4, 0, 0, 0), (32, 8, 0, 0, 0), (64, 8, 0, 0, 0),
             (16, 8, 0, 16384, 0), (32, 8, 0, 16384, 0), (16, 0, 0, 0, 1), (32, 0, 0, 0, 1), (64, 0, 0, 0, 1)]
    if a.cases:
        cases = [tuple(int(x) for x in c.split(",")) for c in a.cases.split(";")]
    for (w, lanes, waves, batch, hostp) in cases:
        env = dict(os.environ)
        env["NVH_CORPUS_TIMING"] = "1"
        env["NVH_TIME_UPLOAD"] = "1"
        if lanes:
            env["NVH_PARSE_LANES"] = str(lanes)
        if waves:
            env["NVH_PARSE_WAVES"] = str(waves)
        if batch:
            env["NVH_CORPUS_BATCH"] = str(batch)
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--scale", str(a.scale), "--workers", str(w), "--reps", str(a.reps)]
        if hostp:
            cmd.append("--host-parse")
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        m = re.findall(r"DECODE_S ([0-9.]+)", r.stdout)
        reps = re.findall(r"REP_S ([0-9.]+)", r.stdout)
        passes = re.findall(r"demux \+ index pass ([0-9.]+) s, arena ([0-9.]+) s, decode pass ([0-9.]+) s", r.stdout)
        up = re.findall(r"host prep ([0-9.]+) ms, enqueue ([0-9.]+) ms, wait for k_parse ([0-9.]+) ms", r.stdout)
        n = max(1, len(up))
        sums = [sum(float(u[k]) for u in up) for k in range(3)]
        print("workers %2d lanes %2s waves %2s batch %5s %s: decode_s %s (reps %s) | passes (index, arena, decode) %s | %d uploads: host prep %.0f ms, enqueue %.0f ms, "
              "wait %.0f ms summed over workers (per upload %.2f / %.2f / %.2f ms)" % (
                  w, lanes or "-", waves or "-", batch or "-", "host-parse" if hostp else "gpu-parse ", m[-1] if m else "FAILED", " ".join(reps),
                  " ".join("/".join(x) for x in passes) if passes else "-", len(up), sums[0], sums[1], sums[2], sums[0] / n, sums[1] / n, sums[2] / n), flush=True)
        for ln in r.stdout.splitlines():
            if "summed over the workers:" in ln and "decode_files_to_device: " in ln:
                print("    " + ln.split("decode_files_to_device: ")[1], flush=True)
        if not m:
            print(r.stdout[-1500:], flush=True)
