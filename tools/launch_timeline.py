#!/usr/bin/env python
"""Pair-launch durations by position in a list's life, from a rocprofv3 kernel trace (rocpd database).

    rocprofv3 --kernel-trace -d /tmp/p -- python bench.py --steps 400 --warmup 100 --no-cpu-baseline --no-secondary
    python tools/launch_timeline.py /tmp/p/<host>/<pid>_results.db

The first pair launch after a list build writes the row padding and runs its tail checked (docs/history/round4.md, padded rows); this
prints what that launch costs against the ones that follow, and the gaps on the device between consecutive launches."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    age, by_age, gaps = None, {}, []
    prev_end = None
    for name, start, end in rows:
        if "build_list_kernel" in name and end - start > 50_000:  # a real build (early exits take a few us)
            age = 0
        elif "list_pair_fast_f32_kernel" in name and name.split("(")[0].rstrip().endswith((", 1>", ", 2>")):  # FUSED = 1 / 2
            if age is not None:
                by_age.setdefault(min(age, 12), []).append((end - start) / 1e3)
                age += 1
            if prev_end is not None:
                gaps.append((start - prev_end) / 1e3)
        prev_end = end
    print("launches after a build: mean / min / max duration of the fused pair + step launch (us), count")
    for a in sorted(by_age):
        v = by_age[a]
        print(f"  {a:2d}{'+' if a == 12 else ' '} {sum(v) / len(v):7.2f} {min(v):7.2f} {max(v):7.2f}  {len(v)}")
    if gaps:
        gaps.sort()
        print(f"idle on the device before a fused launch (us): median {gaps[len(gaps) // 2]:.2f}, 90 % {gaps[int(0.9 * len(gaps))]:.2f}, max {gaps[-1]:.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
