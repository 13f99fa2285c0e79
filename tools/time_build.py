#!/usr/bin/env python
"""Forced list rebuilds of the relaxed C3 box, for a kernel trace of the rebuild chain alone:

    cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/tb -- python $REPO/tools/time_build.py [n]
    python profiles/summarize_rocpd.py /tmp/tb/*/*_results.db | grep -i "build_list\|scan_place\|bin_members"

Also prints the wall-clock cost of a rebuild (forced rebuild + evaluation minus a steady evaluation) and a checksum of
the forces from the rebuilt list (variants that must not change the list agree on it)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from torchmd_amd import _lib as L  # noqa: E402
from torchmd_amd.integrator import Integrator  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
relax = int(os.environ.get("TIME_BUILD_RELAX", "300"))
dev = torch.device("cuda:0")
mol, par, system, forces, box = bench.build_system(32, dev, torch.float32, seed=1)
forces.compute(system.pos, system.box, system.forces)
if relax:
    Integrator(system, forces, 1.0, dev, gamma=10.0, T=300.0).step(relax)
eng = forces._engine(system.pos)
F = torch.zeros_like(system.pos)


def timed(fn, k):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / k * 1e6


def forced():
    L.check(eng.lib.tmdhip_invalidate_list(eng.ctx, 0))
    forces._evaluate(system.pos, system.box, F, False, True)


forced()
steady = timed(lambda: forces._evaluate(system.pos, system.box, F, False, True), 50)
reb = timed(forced, n)
print(f"TIMEBUILD lib {os.environ.get('TMDHIP_LIB', 'default')}: steady evaluation {steady:.1f} us, with a forced rebuild {reb:.1f} us -> rebuild "
      f"{reb - steady:.1f} us (host-synchronising re-plan included); entries {forces.stats(system.pos)['list_entries']}, "
      f"force checksum {float(F.double().abs().sum().item()):.6e}")
