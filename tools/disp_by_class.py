"""Largest displacement of the oxygens and of the hydrogens of the C3 water box n steps after a reference
point (the quantity a per-atom Verlet skin would be matched to): python tools/disp_by_class.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from torchmd_amd.integrator import Integrator

device = torch.device("cuda", 0)
mol, par, system, forces, box = bench.build_system(32, device, torch.float32, seed=1)
forces.compute(system.pos, system.box, system.forces)
Integrator(system, forces, bench.TIMESTEP_FS, device, gamma=10.0, T=300.0).step(1500)
integ = Integrator(system, forces, bench.TIMESTEP_FS, device, gamma=0.1, T=300.0)
integ.step(200)
isO = torch.tensor(np.asarray(par.masses).ravel() > 2.0, device=device)
rows = {}
for rep in range(12):
    ref = system.pos[0].clone()
    for n in range(1, 21):
        integ.step(1)
        d = (system.pos[0] - ref).norm(dim=1)
        rows.setdefault(n, []).append((d[isO].max().item(), d[~isO].max().item()))
print("steps  max|d| O     max|d| H    ratio")
for n in (4, 6, 8, 9, 10, 11, 12, 14, 16, 20):
    a = np.array(rows[n])
    print(f"{n:4d}   {a[:,0].mean():.3f}+-{a[:,0].std():.3f}  {a[:,1].mean():.3f}+-{a[:,1].std():.3f}  {(a[:,0]/a[:,1]).mean():.3f}")
