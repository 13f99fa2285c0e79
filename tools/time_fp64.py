#!/usr/bin/env python
"""C3 water box in fp64 (generic list kernel) — step time for reference (needs a GPU)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import build_system  # noqa: E402
from torchmd_amd.integrator import Integrator  # noqa: E402

dev = torch.device("cuda:0")
mol, par, system, forces, box = build_system(32, dev, torch.float64, seed=1)
forces.compute(system.pos, system.box, system.forces)
Integrator(system, forces, 1.0, dev, gamma=10.0, T=300.0).step(300)
it = Integrator(system, forces, 1.0, dev, gamma=0.1, T=300.0)
it.step(50)
forces.enable_timing(system.pos, True, every=8)
forces.read_timing(system.pos)
torch.cuda.synchronize()
t0 = time.perf_counter()
ek, ep, T = it.step(500)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ms, n = forces.read_timing(system.pos)
print(f"fp64 C3: {dt / 500 * 1e6:.1f} us/step, pair kernel {ms / max(n, 1) * 1e3:.1f} us, T={T[0]:.1f} K")
