#!/usr/bin/env python
"""Kernel timeline of the LAST submission in a rocprofv3 --kernel-trace database (rocpd).  usage: python tools/timeline.py <db> [min_us]"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
rows = list(cur.execute("select name, start, end, grid_x, workgroup_x, queue_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "strings" in r[0]]
start = idx[-2] + 1 if len(idx) >= 2 else 0
# the submission begins after the previous one's gathers: first copy after the last gather before the last strings kernel
g = [i for i in range(start, idx[-1]) if "gather" in rows[i][0]]
if g: start = g[-1] + 1
t0 = rows[start][1]
for r in rows[start:]:
    d = (r[2] - r[1]) / 1e3
    if d >= min_us or "copy" not in r[0]:
        print(f"{(r[1]-t0)/1e6:9.3f} ms +{d/1e3:9.3f} ms  q{r[5]} grid {r[3]:9d} wg {r[4]:4d}  {r[0].split('(')[0][:48]}")
