#!/usr/bin/env python
"""Kernel timeline of ONE short step() call from a rocprofv3 kernel trace (rocpd database):

    cd /tmp && rocprofv3 --kernel-trace -d /tmp/ct -- python $REPO/tools/short_call.py 20 8
    python tools/call_timeline.py /tmp/ct/<host>/<pid>_results.db [call index from the end, default 2]

Prints every dispatch between two observe_publish kernels (= one Integrator.step call) with its start relative to the
previous call's last kernel end, its duration and the gap to its predecessor."""
import sqlite3
import sys


def main(path, back=2):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "observe_publish" in r[0]]
    lo, hi = marks[-back - 1], marks[-back]
    t0 = rows[lo][2]
    prev = t0
    busy = 0
    for name, s, e in rows[lo + 1:hi + 1]:
        short = name.split("(")[0].replace("void ", "").replace("tmd::", "")
        if "list_pair_fast" in short:
            short = "list_pair_fast" + name[name.index("<"):name.index(">") + 1]
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev) / 1e3:6.1f}  {short[:70]}")
        busy += e - s
        prev = e
    print(f"call: {(prev - t0) / 1e3:.1f} us from the previous call's last kernel to this one's; kernels {busy / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
