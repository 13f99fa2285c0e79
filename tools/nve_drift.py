#!/usr/bin/env python
"""Energy conservation of the full fp32 pipeline on the C3 water box (needs a GPU): equilibrate with
Langevin, then NVE for N steps at 0.5 fs (flexible TIP3P, reaction-field cutoff) and report the drift."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench import build_system  # noqa: E402
from torchmd_amd.integrator import Integrator  # noqa: E402

dev = torch.device("cuda:0")
mol, par, system, forces, box = build_system(32, dev, torch.float32, seed=1)
forces.compute(system.pos, system.box, system.forces)
Integrator(system, forces, 1.0, dev, gamma=10.0, T=300.0).step(2000)
Integrator(system, forces, 0.5, dev, gamma=1.0, T=300.0).step(2000)
nve = Integrator(system, forces, 0.5, dev)
e = []
r0 = forces.stats(system.pos)["n_rebuilds"]
for _ in range(int(os.environ.get("NVE_BLOCKS", "20"))):
    ek, ep, T = nve.step(200)
    e.append((ek[0] + ep[0], T[0]))
e = np.array(e)
n = mol.numAtoms
print(f"NVE {len(e)*200} steps x 0.5 fs, N={n}: Etot/N first {e[0,0]/n:.5f} last {e[-1,0]/n:.5f} kcal/mol, "
      f"drift {(e[-1,0]-e[0,0])/n/(len(e)*0.1):.2e} kcal/mol/atom/ps, rms fluct {e[:,0].std()/n:.2e}, T {e[0,1]:.1f} -> {e[-1,1]:.1f} K, "
      f"rebuilds {forces.stats(system.pos)['n_rebuilds'] - r0}")
