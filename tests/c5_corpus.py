"""BASELINE.json configs[4] ("C5"): the 1000-file .ogg corpus of SURVEY 8d, as a deterministic function of `scale`.

The reference ships no corpus.  File i (seed = i) is written by the packet writer of tests/vorbis_encode.py -- the
inverse of the host parser: Huffman-encoded side information on 3test.ogg's setup headers, block kinds from the C3
Markov chain, laced into CRC-valid Ogg pages -- with a length log-uniform in 5 .. 300 s x scale; the four shipped
TestFiles close the list.  scale = 1.0 is the stated size (3.1 M frames, 25 GB of float PCM); tests/golden holds the
oracle's SHA-256 of every file's PCM for the scales listed in DIGEST_SCALES (written by tools/corpus_c5.py --make-digests,
so that the GPU test does not spend minutes in the oracle).  A digest entry also carries the hash of the .ogg bytes: the
test first proves that it decodes the very files the oracle decoded."""
import hashlib
import json
import os

import numpy as np

from tests import vorbis_encode as ve

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SHIPPED = ("1test", "2test", "3test", "issue6test")
N_WRITER_FILES = 1000
POOL_SEED, POOL_PER_KIND = 5, 64
DIGEST_SCALES = (0.1, 1.0)


def digest_path(scale):
    return os.path.join(GOLDEN, "c5_digests_scale%g.json" % scale)


def writer_setup():
    hdr = ve.shipped_headers(open(os.path.join(GOLDEN, "3test.ogg"), "rb").read())
    S = ve.setup_of(hdr)
    pool = ve.packet_pool(S, POOL_SEED, per_kind=POOL_PER_KIND)
    return S, hdr, pool


def corpus_file(ws, index, scale):
    """File `index` of the corpus: a writer file for index < N_WRITER_FILES, then the shipped TestFiles."""
    if index < N_WRITER_FILES:
        S, hdr, pool = ws
        return ve.corpus_file(S, hdr, pool, index, scale=scale)
    return open(os.path.join(GOLDEN, SHIPPED[index - N_WRITER_FILES] + ".ogg"), "rb").read()


def n_files():
    return N_WRITER_FILES + len(SHIPPED)


def build_files(scale, ws=None):
    ws = ws or writer_setup()
    return [corpus_file(ws, i, scale) for i in range(n_files())]


def build_subset(indices, scale, procs=0):
    """{index: bytes} of the corpus files `indices` -- what ONE rank of a sharded job needs.  procs > 1: written by that many
    child interpreters (python -m tests.c5_corpus --emit ...; the writer is a Python loop per packet, ~30 ms per file) into a
    scratch directory (tmpfs where there is one) and read back; children never import torch or touch the GPU."""
    indices = list(indices)
    if procs <= 1 or len(indices) < 4 * procs:
        ws = writer_setup()
        return {i: corpus_file(ws, i, scale) for i in indices}
    import shutil
    import subprocess
    import sys
    import tempfile
    tmp = tempfile.mkdtemp(prefix="nvh_c5_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None)
    try:
        # file lengths are log-uniform and independent of the index: strided lists balance
        jobs = []
        for k in range(procs):
            part = indices[k::procs]
            out = os.path.join(tmp, "part%d.bin" % k)
            jobs.append((part, out, subprocess.Popen([sys.executable, "-m", "tests.c5_corpus", "--emit", out, "--scale", repr(float(scale)),
                                                      "--indices", ",".join(map(str, part))], cwd=ROOT)))
        files = {}
        for part, out, pr in jobs:
            if pr.wait() != 0:
                raise RuntimeError("corpus writer child failed (%d)" % pr.returncode)
            lens = json.load(open(out + ".idx"))
            blob = open(out, "rb").read()
            off = 0
            for i, n in zip(part, lens):
                files[i] = blob[off:off + n]
                off += n
            assert off == len(blob)
        return files
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def file_digest(data):
    return hashlib.sha256(data).hexdigest()[:16]


def pcm_digest(pcm):
    a = np.ascontiguousarray(pcm, dtype=np.float32)
    return hashlib.sha256(a.view(np.uint8).data).hexdigest()


def load_digests(scale):
    p = digest_path(scale)
    if not os.path.exists(p):
        return None
    d = json.load(open(p))
    assert d["files"] == n_files() and abs(d["scale"] - scale) < 1e-12, (d["files"], d["scale"])
    return d


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--emit", required=True)
    ap.add_argument("--scale", type=float, required=True)
    ap.add_argument("--indices", required=True)
    a = ap.parse_args()
    ws_ = writer_setup()
    lens_ = []
    with open(a.emit, "wb") as fh:
        for i_ in map(int, a.indices.split(",")):
            d_ = corpus_file(ws_, i_, a.scale)
            fh.write(d_)
            lens_.append(len(d_))
    json.dump(lens_, open(a.emit + ".idx", "w"))
