#!/usr/bin/env python
"""Context number for DESIGN.md: the UNMODIFIED reference (dense O(N^2) pair tensor) on this
container's CPU cores on the synthetic water recipe, a few sizes, fit t = c N^2 and extrapolate to C3.
Needs /root/reference (build container only); not used by tests, smoke or bench."""
import sys
import time

sys.path.insert(0, "/root/reference")
sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from torchmd.forces import Forces  # noqa: E402
from torchmd.integrator import Integrator, maxwell_boltzmann  # noqa: E402
from torchmd.systems import System  # noqa: E402

from torchmd_amd.builders import tip3p_box, water_forcefield  # noqa: E402
from torchmd_amd.parameters import Parameters  # noqa: E402

rows = []
for nside in (8, 10, 13):
    mol, pos, box = tip3p_box(nside, seed=1)
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=torch.float32)
    s = System(mol.numAtoms, 1, torch.float32, "cpu")
    s.set_positions(pos[:, :, None])
    s.set_box(box)
    torch.manual_seed(1)
    s.set_velocities(maxwell_boltzmann(par.masses, 300.0, 1))
    f = Forces(par, terms=terms, cutoff=9.0, rfa=True)
    it = Integrator(s, f, 1.0, "cpu", gamma=0.1, T=300.0)
    it.step(1)
    t0 = time.perf_counter()
    it.step(3)
    dt = (time.perf_counter() - t0) / 3
    rows.append((mol.numAtoms, dt))
    print(f"N={mol.numAtoms}: {dt:.3f} s/step ({torch.get_num_threads()} threads)", flush=True)
n = np.array([r[0] for r in rows], dtype=float)
t = np.array([r[1] for r in rows])
c = (t * n**2).sum() / (n**4).sum()
print(f"fit t = {c:.3e} N^2  ->  N=98304: {c * 98304**2:.1f} s/step = {86400e-6 / (c * 98304**2):.2e} ns/day "
      f"(the dense pair tensor would need {98304**2 / 2 * 2 * 8 / 1e9:.0f} GB)")
