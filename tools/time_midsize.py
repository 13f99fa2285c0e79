#!/usr/bin/env python
"""Mid-size water boxes on the cell-list path (needs a GPU): us per MD step for tip3p_box(nside), nside from argv
(default 12 16 20 24 = 5 184 / 12 288 / 24 000 / 41 472 atoms), fp32, 9 A + reaction field, Langevin, 1 fs."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torchmd_amd.builders import tip3p_box, water_forcefield
from torchmd_amd.forces import Forces
from torchmd_amd.integrator import Integrator, maxwell_boltzmann
from torchmd_amd.parameters import Parameters
from torchmd_amd.systems import System

dev = torch.device("cuda:0")
terms = ["lj", "electrostatics", "bonds", "angles"]
skin = float(os.environ["MIDSIZE_SKIN"]) if os.environ.get("MIDSIZE_SKIN") else None  # (A/B: Verlet skin in A)
for nside in [int(a) for a in sys.argv[1:]] or [12, 16, 20, 24]:
    mol, pos, box = tip3p_box(nside, seed=0)
    par = Parameters(water_forcefield(mol), mol, terms, precision=torch.float32)
    s = System(mol.numAtoms, 1, torch.float32, dev)
    s.set_positions(pos[:, :, None]); s.set_box(box)
    torch.manual_seed(1)
    s.set_velocities(maxwell_boltzmann(par.masses, 300.0, 1))
    f = Forces(par, terms=terms, cutoff=9.0, rfa=True, skin_weights="mass", **({"skin": skin} if skin else {}))
    f.compute(s.pos, s.box, s.forces)
    Integrator(s, f, 1.0, dev, gamma=10.0, T=300.0).step(600)
    integ = Integrator(s, f, 1.0, dev, gamma=0.1, T=300.0)
    integ.step(200)
    st0 = f.stats(s.pos)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    integ.step(2000)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    st = f.stats(s.pos)
    print(f"{mol.numAtoms} atoms: {el / 2000 * 1e6:.1f} us/step = {2000 / el * 1e-6 * 86400:.0f} ns/day, algorithm {st['algorithm']}, "
          f"rebuilds {st['n_rebuilds'] - st0['n_rebuilds']}, chains skipped {st['chains_skipped'] - st0['chains_skipped']}, "
          f"steps made by the pair launch {st['steps_in_pair_launch'] - st0['steps_in_pair_launch']}", flush=True)
    f.close()
