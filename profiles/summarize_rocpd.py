#!/usr/bin/env python
"""Per-kernel summary (the `--stats` view) of a rocprofv3 rocpd database (`*_results.db`).

    python profiles/summarize_rocpd.py gpurun_out/prof/<host>/<pid>_results.db > profiles/<name>.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        "max(vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name order by 3 desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# source: {path}")
    print(f"# total kernel time: {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print("name,calls,total_ms,avg_us,min_us,max_us,percent,vgpr,sgpr,lds_bytes")
    for name, n, tot, avg, mn, mx, vg, sg, lds in rows:
        short = name if len(name) < 110 else name[:107] + "..."
        print(f"\"{short}\",{n},{tot/1e6:.3f},{avg/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f},{100*tot/total:.2f},{vg},{sg},{lds}")


if __name__ == "__main__":
    main(sys.argv[1])
