"""The corpus pass's host-only index (nvorbis_amd.corpus._index_pass: lacing-only page walk + packet geometry of every file) against
its thread count, BASELINE C5 at its stated size, no GPU involved:   python tools/index_sweep.py [threads,threads,...]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import c5_corpus
from nvorbis_amd import corpus

counts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8,16").split(",")]
d = c5_corpus.build_subset(range(c5_corpus.n_files()), 1.0, procs=min(16, corpus._cpu_budget()))
files = [d[i] for i in range(len(d))]
corpus._tune_malloc()
corpus._index_pass(files[:64], 4)
print("cpu budget", corpus._cpu_budget(), "files", len(files), "%.2f GB" % (sum(map(len, files)) / 1e9), flush=True)
for w in counts:
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        shape, totals, chans, errors = corpus._index_pass(files, w)
        ts.append(time.perf_counter() - t)
        assert not errors and sum(totals) == 5402637680
    print("%2d threads: min %.3f s, median %.3f s" % (w, min(ts), sorted(ts)[2]), flush=True)
