#!/usr/bin/env python
"""Dispatch timeline of the last seconds of a rocprofv3 kernel trace (rocpd database): name, start relative to
the first row shown, duration and the idle gap in front of every dispatch.
    python tools/timeline.py <results.db> [rows=80] [skip_from_end=0]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 80
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
k = db.execute("select name, start, end from kernels order by start").fetchall()
k = k[len(k) - rows - skip: len(k) - skip]
t0, prev = k[0][1], k[0][1]
for name, s, e in k:
    short = name.split("(")[0].replace("void ", "").replace("tmd::", "")[:58]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:6.1f}  {short}")
    prev = e
