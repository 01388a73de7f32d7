"""BASELINE.json configs[4] at its stated size on ONE GPU: 1000 writer files (+ the 4 TestFiles), every file's PCM checked.

  python tools/corpus_c5.py --make-digests --scale 0.1     (CPU, here: the oracle decodes every file; writes
                                                            tests/golden/c5_digests_scale0.1.json)
  python tools/corpus_c5.py --run --scale 1.0 [--workers 16] [--gpu-parse]     (GPU box)

--run decodes the corpus file-parallel through the HIP path into one device arena (nvorbis_amd.corpus.transcode at world
1, to_host=False: the path tools/corpus_transcode.py runs per rank), compares the SHA-256 of every file's PCM with the
oracle's committed digest and prints one JSON line (files/s, frames/s, mismatches).  The corpus definition lives in
tests/c5_corpus.py (lengths log-uniform 5-300 s x scale, seed = file index)."""
import argparse
import json
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # before the HIP runtime initialises (nvorbis_amd.configure_process says why)
os.environ.setdefault("NVH_CORPUS_MALLOPT", "1")  # this process is a corpus job: the allocator settings of nvorbis_amd.corpus._tune_malloc (opt-in)
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from tests import c5_corpus  # noqa: E402

_ws = None


def _digest_one(args):
    """(file index, scale) -> [file digest, PCM floats, PCM digest, channels] by the CPU oracle."""
    global _ws
    i, scale = args
    from tests import oracle_py
    if _ws is None:
        _ws = c5_corpus.writer_setup()
    data = c5_corpus.corpus_file(_ws, i, scale)
    pcm, info = oracle_py.load().decode_ogg(data)
    return [c5_corpus.file_digest(data), int(pcm.size), c5_corpus.pcm_digest(pcm), int(info["channels"])]


def make_digests(scale, procs):
    import multiprocessing as mp
    t0 = time.time()
    n = c5_corpus.n_files()
    with mp.get_context("fork").Pool(procs) as pool:
        rows = pool.map(_digest_one, [(i, scale) for i in range(n)], chunksize=4)
    out = {"scale": scale, "files": n, "pool_seed": c5_corpus.POOL_SEED, "pool_per_kind": c5_corpus.POOL_PER_KIND,
           "what": "oracle (oracle/, CPU restatement of NVorbis) PCM per file: [sha256(.ogg)[:16], floats, sha256(float32 PCM), channels]",
           "total_floats": sum(r[1] for r in rows), "digests": rows}
    with open(c5_corpus.digest_path(scale), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote %s: %d files, %.1f M floats, %.0f s" % (c5_corpus.digest_path(scale), n, out["total_floats"] / 1e6, time.time() - t0))


def run(scale, workers, gpu_parse):
    import torch

    from nvorbis_amd import corpus
    dig = c5_corpus.load_digests(scale)
    if dig is None:
        raise SystemExit("no committed digests for scale %g (tools/corpus_c5.py --make-digests --scale %g)" % (scale, scale))
    t0 = time.perf_counter()
    files = c5_corpus.build_files(scale)
    t_gen = time.perf_counter() - t0
    bad_files = [i for i, f in enumerate(files) if c5_corpus.file_digest(f) != dig["digests"][i][0]]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    arena, views = corpus.decode_files_to_device(files, device=0, workers=workers, gpu_parse=gpu_parse)
    torch.cuda.synchronize()
    t_dec = time.perf_counter() - t0
    t0 = time.perf_counter()
    mism = []
    for i, v in enumerate(views):
        row = dig["digests"][i]
        if int(v.numel()) != row[1] or c5_corpus.pcm_digest(v.cpu().numpy()) != row[2]:
            mism.append(i)
    t_chk = time.perf_counter() - t0
    floats = int(arena.numel())
    print(json.dumps({"config": "C5: %d-file corpus, lengths log-uniform 5-300 s x %g, one MI355X (world 1)" % (len(files), scale),
                      "files": len(files), "scale": scale, "workers": workers, "gpu_parse": bool(gpu_parse),
                      "ogg_bytes": sum(len(f) for f in files), "pcm_floats": floats,
                      "generate_s": t_gen, "decode_s": t_dec, "check_s": t_chk,
                      "files_per_s": len(files) / t_dec, "long_frame_equivalents_per_s": floats / 2 / 1024 / t_dec,
                      "input_mismatches": bad_files, "pcm_mismatches": mism,
                      "verdict": "every file's PCM SHA-256 equals the oracle's" if not mism and not bad_files else "MISMATCH"}), flush=True)
    return 0 if not mism and not bad_files else 1


def add_sizes(scale):
    """`ogg_bytes` of the digest file: the compressed size of every file, checked against the committed file digests."""
    path = c5_corpus.digest_path(scale)
    d = json.load(open(path))
    files = c5_corpus.build_files(scale)
    assert [c5_corpus.file_digest(f) for f in files] == [r[0] for r in d["digests"]]
    d["ogg_bytes"] = [len(f) for f in files]
    with open(path, "w") as fh:
        json.dump(d, fh, separators=(",", ":"))
    print("sizes entered:", path, sum(d["ogg_bytes"]), "bytes")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--make-digests", action="store_true")
    ap.add_argument("--add-sizes", action="store_true", help="enter every file's compressed size into the committed digest file (what an LPT shard needs, so that a rank builds its own shard only)")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--scale", type=float, default=0.1)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--procs", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--gpu-parse", action="store_true")
    a = ap.parse_args()
    if a.make_digests:
        make_digests(a.scale, a.procs)
    if a.add_sizes or a.make_digests:
        add_sizes(a.scale)
    if a.run:
        raise SystemExit(run(a.scale, a.workers, a.gpu_parse))
