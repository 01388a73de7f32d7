"""BASELINE.json configs[4] on one GPU: offline transcode of a file corpus, end to end (file bytes in host memory ->
demux -> host parse -> H2D -> kernels -> D2H -> PCM in host memory), with a pool of host threads per GPU
(nvorbis_amd.corpus.decode_files_threaded).  The corpus is the four shipped test files cycled to `--files` entries.
The per-GPU shard of the 8-GPU form is exactly this; the only cross-GPU step is the final PCM gather
(nvorbis_amd.corpus.gather_pcm)."""
import argparse, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process: torch's)
import nvorbis_amd as nv
from nvorbis_amd import corpus

ap = argparse.ArgumentParser()
ap.add_argument("--files", type=int, default=1000)
ap.add_argument("--workers", type=str, default="1,4,16,32,64")
ap.add_argument("--batch-frames", type=int, default=4096)
ap.add_argument("--gpu-parse", action="store_true")
a = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = [open(os.path.join(root, "tests", "golden", n + ".ogg"), "rb").read() for n in ("1test", "2test", "3test", "issue6test")]
files = [base[i % len(base)] for i in range(a.files)]
ref = None
for w in [int(x) for x in a.workers.split(",")]:
    t0 = time.perf_counter()
    out = corpus.decode_files_threaded(files, device=0, workers=w, batch_frames=a.batch_frames, gpu_parse=a.gpu_parse)
    dt = time.perf_counter() - t0
    samples = sum(o.size for o in out) // 2
    if ref is None:
        ref = [out[i].copy() for i in range(len(base))]
    for i, o in enumerate(out):
        assert o.size == ref[i % len(base)].size and (o == ref[i % len(base)]).all(), "file %d differs" % i
    print("workers %3d: %6.2f s, %7.1f files/s, %6.2f M samples/s per channel, ~%5.0f k long-frame equivalents/s (1024 samples), %.1f MB/s compressed" % (
        w, dt, len(files) / dt, samples / dt / 1e6, samples / 1024 / dt / 1e3, sum(len(f) for f in files) / dt / 1e6), flush=True)
