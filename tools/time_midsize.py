#!/usr/bin/env python
"""Mid-size water boxes on the cell-list path (needs a GPU): us per MD step for tip3p_box(nside), fp32, 9 A + reaction
field, Langevin, 1 fs.

    python tools/time_midsize.py [--replicas R] [--steps K] [nside ...]      (default nside: 12 16 20 24 =
                                                                              5 184 / 12 288 / 24 000 / 41 472 atoms)

--replicas R: R replicas of the box in one context (different velocities); the figure is us per MD step of ALL of them.
TMDHIP_BATCH_REPLICAS=0 in the environment keeps the replica-by-replica loop (A/B)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torchmd_amd.builders import tip3p_box, water_forcefield
from torchmd_amd.forces import Forces
from torchmd_amd.integrator import Integrator, maxwell_boltzmann
from torchmd_amd.parameters import Parameters
from torchmd_amd.systems import System

ap = argparse.ArgumentParser()
ap.add_argument("--replicas", type=int, default=1)
ap.add_argument("--steps", type=int, default=2000)
ap.add_argument("nside", type=int, nargs="*", default=[12, 16, 20, 24])
a = ap.parse_args()
dev = torch.device("cuda:0")
terms = ["lj", "electrostatics", "bonds", "angles"]
skin = float(os.environ["MIDSIZE_SKIN"]) if os.environ.get("MIDSIZE_SKIN") else None  # (A/B: Verlet skin in A)
R = a.replicas
for nside in a.nside:
    mol, pos, box = tip3p_box(nside, seed=0)
    par = Parameters(water_forcefield(mol), mol, terms, precision=torch.float32)
    s = System(mol.numAtoms, R, torch.float32, dev)
    s.set_positions(pos[:, :, None].repeat(R, axis=2)); s.set_box(box)
    torch.manual_seed(1)
    s.set_velocities(maxwell_boltzmann(par.masses, 300.0, R))
    f = Forces(par, terms=terms, cutoff=9.0, rfa=True, skin_weights="mass", **({"skin": skin} if skin else {}))
    f.compute(s.pos, s.box, s.forces)
    Integrator(s, f, 1.0, dev, gamma=10.0, T=300.0).step(600)
    integ = Integrator(s, f, 1.0, dev, gamma=0.1, T=300.0)
    integ.step(200)
    st0 = f.stats(s.pos)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ek, ep, T = integ.step(a.steps)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    st = f.stats(s.pos)
    print(f"{mol.numAtoms} atoms x {R}: {el / a.steps * 1e6:.1f} us/step = {a.steps / el * 1e-6 * 86400:.0f} ns/day per replica, algorithm {st['algorithm']}, "
          f"rebuilds (replica 0) {st['n_rebuilds'] - st0['n_rebuilds']}, chains skipped {st['chains_skipped'] - st0['chains_skipped']}, "
          f"steps made by the pair launch {st['steps_in_pair_launch'] - st0['steps_in_pair_launch']}, batched launches "
          f"{st['batched_launches'] - st0['batched_launches']}, T {T.mean():.0f} K", flush=True)
    f.close()
