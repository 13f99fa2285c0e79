#!/usr/bin/env python
"""Per-kernel averages of the PMC counters stored in a rocprofv3 rocpd database.

    python profiles/summarize_pmc.py <results.db> [--min-us=X] [kernel-substring ...]
(--min-us drops dispatches shorter than X us, e.g. the early-exit launches of the rebuild chain)
Prints: kernel, dispatches, avg duration (us), then avg value per dispatch of every counter.
"""
import sqlite3
import sys
from collections import defaultdict


def main(path, filters, min_us=0.0):
    db = sqlite3.connect(path)
    cols = [d[1] for d in db.execute("pragma table_info(pmc_events)")]
    cname = "counter_name" if "counter_name" in cols else ("pmc_name" if "pmc_name" in cols else None)
    vname = "counter_value" if "counter_value" in cols else "value"
    if cname is None:
        cname = [c for c in cols if "name" in c and c != "name"][0]
    rows = db.execute(f"select name, dispatch_id, {cname}, {vname}, duration from pmc_events").fetchall()
    agg = defaultdict(lambda: defaultdict(float))
    ndisp = defaultdict(set)
    dur = defaultdict(float)
    for name, disp, c, v, d in rows:
        if filters and not any(f in name for f in filters):
            continue
        if d < min_us * 1e3:
            continue
        key = name.split("(")[0][-70:]
        agg[key][c] += v
        if disp not in ndisp[key]:
            ndisp[key].add(disp)
            dur[key] += d
    print(f"# source: {path}")
    for key in sorted(agg, key=lambda k: -dur[k]):
        n = len(ndisp[key])
        print(f"{key}: dispatches={n} avg_us={dur[key]/n/1e3:.2f}")
        for c, v in sorted(agg[key].items()):
            print(f"    {c:34s} {v/n:16.1f} per dispatch")


if __name__ == "__main__":
    args = sys.argv[2:]
    min_us = 0.0
    if args and args[0].startswith("--min-us="):
        min_us = float(args.pop(0).split("=")[1])
    main(sys.argv[1], args, min_us)
