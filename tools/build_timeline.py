#!/usr/bin/env python
"""Per-block timeline of the Verlet-list build on the relaxed C3 box (needs a GPU; TMDHIP_DEBUG_TIMELINE=1 is set here):
wave durations, and — per group of blocks that share a cycle counter — how many waves are alive over the launch."""
import ctypes as C
import os
import sys

os.environ["TMDHIP_DEBUG_TIMELINE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import build_system  # noqa: E402
from torchmd_amd import _lib as L  # noqa: E402
from torchmd_amd.integrator import Integrator  # noqa: E402

dev = torch.device("cuda:0")
mol, par, system, forces, box = build_system(32, dev, torch.float32, seed=1)
forces.compute(system.pos, system.box, system.forces)
Integrator(system, forces, 1.0, dev, gamma=10.0, T=300.0).step(600)
Integrator(system, forces, 1.0, dev, gamma=0.1, T=300.0).step(100)  # ends with device-side rebuilds in the MD loop
lib = L.load()
lib.tmdhip_debug_build_timeline.restype = C.c_int
lib.tmdhip_debug_build_timeline.argtypes = [C.c_void_p, C.c_size_t]
buf = np.zeros(4 * 20000, dtype=np.uint64)
nb = lib.tmdhip_debug_build_timeline(buf.ctypes.data, buf.nbytes)
print("blocks", nb)
r = buf[: 4 * nb].reshape(nb, 4)
t0, t1, xcc, nmax = r[:, 0].astype(np.int64), r[:, 1].astype(np.int64), r[:, 2].astype(np.int64), r[:, 3].astype(np.int64)
dur = t1 - t0
print(f"block duration cycles: mean {dur.mean():.0f} p5 {np.percentile(dur, 5):.0f} p50 {np.median(dur):.0f} p95 {np.percentile(dur, 95):.0f} max {dur.max()}")
print("corr(duration, longest list of the cell)", np.corrcoef(dur, nmax)[0, 1])
for x in range(8):
    m = xcc == x
    tend = t1[m].max()
    rec = m & (t0 > tend - 600000)
    s, e = t0[rec] - t0[rec].min(), t1[rec] - t0[rec].min()
    T = e.max()
    grid = np.linspace(0, T, 21)
    print(f"xcc {x}: {m.sum()} blocks, {rec.sum()} on the last block's clock; span {T} cycles; alive over time:",
          [int(((s <= b) & (e > b)).sum()) for b in grid])
    print("    entry-time percentiles / span:", [round(float(np.percentile(s, q)) / T, 2) for q in (5, 25, 50, 75, 95, 100)])
