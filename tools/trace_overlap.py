"""Kernel overlap in a rocprofv3 kernel trace (sqlite): per kernel name count / mean duration, the union of all kernel
intervals, and the time-weighted number of kernels in flight.  python tools/trace_overlap.py <trace.db> [t_from_frac]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = list(cur.execute("select name, start, end, queue_id, stream_id from kernels" if "stream_id" in cols else "select name, start, end, queue_id, 0 from kernels"))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
t0 = min(r[1] for r in rows); t1 = max(r[2] for r in rows)
cut = t0 + frac * (t1 - t0)
rows = [r for r in rows if r[1] >= cut]
by = {}
for n, s, e, q, st in rows:
    d = by.setdefault(n, [0, 0])
    d[0] += 1; d[1] += e - s
for n, (c, tot) in sorted(by.items(), key=lambda kv: -kv[1][1])[:8]:
    print("%-28s x%6d  mean %8.1f us  total %8.1f ms" % (n[:28], c, tot / c / 1e3, tot / 1e6))
ev = sorted([(s, 1) for _, s, e, _, _ in rows] + [(e, -1) for _, s, e, _, _ in rows])
busy = 0; depth = 0; last = ev[0][0]; area = 0
for t, d in ev:
    if depth > 0:
        busy += t - last; area += (t - last) * depth
    depth += d; last = t
span = max(r[2] for r in rows) - min(r[1] for r in rows)
print("span %.1f ms, some kernel running %.1f ms (%.0f %%), mean kernels in flight while busy %.2f, queues used %d" % (
    span / 1e6, busy / 1e6, 100.0 * busy / span, area / max(busy, 1), len(set(r[3] for r in rows))))
