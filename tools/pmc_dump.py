#!/usr/bin/env python3
"""Average every PMC counter per kernel from rocprofv3 rocpd sqlite outputs: pmc_dump.py <db> [<db> ...]"""
import sqlite3
import sys
from collections import defaultdict

acc = defaultdict(dict)
for path in sys.argv[1:]:
    cur = sqlite3.connect(path).cursor()
    for name, counter, avg, cnt in cur.execute(
            "select kernel_name, counter_name, avg(value), count(*) from counters_collection "
            "where kernel_name like 'k_%' group by kernel_name, counter_name"):
        acc[name][counter] = (avg, cnt)
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        print("    %-28s %16.1f  (n=%d)" % (c, acc[k][c][0], acc[k][c][1]))
