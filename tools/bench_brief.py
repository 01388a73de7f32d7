#!/usr/bin/env python3
"""Print the interesting fields of bench.py's JSON line (stdin)."""
import json
import sys
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        d = json.loads(line)
        print(sys.argv[1] if len(sys.argv) > 1 else "", "frames/s=%.3e ms/step=%.4f" % (d["value"], d["ms_per_step"]),
              {k: round(v * 1e3, 1) for k, v in d["kernels_ms"].items()}, "frac=%.3f (%s)" % (d["roofline"]["frac"], d["roofline"]["kernel"]))
