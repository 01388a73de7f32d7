"""SHA-256 of the PCM the CPU oracle decodes for bench.py's resident batches -> tests/golden/bench_pcm_digests.json.

bench.py builds, per rank r and decoder instance k, a packet sequence from the long/long packets of tests/golden/3test.ogg:
a priming packet audio[s % L] (s = 7 r + 13 k), then batch j = the 4096 packets audio[(s + 1 + 4096 j + i) % L].  The oracle
(test infrastructure: oracle/, a CPU restatement of the reference) decodes the same sequence here; bench.py only hashes what
the GPU wrote after its timed region and compares with this file -- the oracle is not run by the bench.

  python tools/gen_bench_digests.py [--ranks 8] [--streams 3] [--batches 6]
"""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
from tests import oracle_py  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "bench_pcm_digests.json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--batches", type=int, default=6)
    a = ap.parse_args()
    import nvorbis_amd as nv
    headers, ll, ch = bench.ll_packets(nv, os.path.join(ROOT, "tests", "golden", "3test.ogg"))
    orc = oracle_py.load()
    per_batch = bench.FRAMES * (bench.BLOCK // 2) * ch
    out = {}
    for r in range(a.ranks):
        for k in range(a.streams):
            s = r * 7 + k * 13
            seq = [ll[(s + i) % len(ll)] for i in range(1 + bench.FRAMES * a.batches)]
            pk = list(headers) + seq
            pcm, _ = orc.decode_packets(pk, [-1] * len(pk), [0] * len(pk), clip=True, chunk=1 << 18)
            assert pcm.size >= per_batch * a.batches
            out["seed%d" % s] = [hashlib.sha256(pcm[j * per_batch:(j + 1) * per_batch].tobytes()).hexdigest() for j in range(a.batches)]
            print("seed", s, out["seed%d" % s][0][:16], flush=True)
    json.dump({"what": "sha256 of the float32 PCM (clipped, interleaved) the CPU oracle decodes for bench.py's resident batches: "
                       "key = seed offset 7 * rank + 13 * stream, value[j] = batch j (tools/gen_bench_digests.py)",
               "frames": bench.FRAMES, "block": bench.BLOCK, "channels": ch, "ll_packets": len(ll), "digests": out},
              open(OUT, "w"), indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
