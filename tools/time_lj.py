#!/usr/bin/env python
"""Config C5 on ONE GPU: 10^6-atom Lennard-Jones (argon) box, cutoff 9 A, Langevin 85 K, 1 fs —
step time, list statistics and pair-kernel time (needs a GPU).

    python tools/time_lj.py [--nside 100] [--steps 300]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from torchmd_amd.builders import argon_forcefield, lj_box  # noqa: E402
from torchmd_amd.forces import Forces  # noqa: E402
from torchmd_amd.integrator import Integrator, maxwell_boltzmann  # noqa: E402
from torchmd_amd.parameters import Parameters  # noqa: E402
from torchmd_amd.systems import System  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nside", type=int, default=100)
    ap.add_argument("--steps", type=int, default=300)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    mol, pos, box = lj_box(args.nside, seed=0)
    par = Parameters(argon_forcefield(mol), mol, ["lj"], precision=torch.float32)
    s = System(mol.numAtoms, 1, torch.float32, dev)
    s.set_positions(pos[:, :, None])
    s.set_box(box)
    torch.manual_seed(1)
    s.set_velocities(maxwell_boltzmann(par.masses, 85.0, 1))
    f = Forces(par, terms=["lj"], cutoff=9.0)
    f.compute(s.pos, s.box, s.forces)
    integ = Integrator(s, f, 1.0, dev, gamma=1.0, T=85.0)
    integ.step(200)
    f.enable_timing(s.pos, True, every=16)
    f.read_timing(s.pos)
    r0 = f.stats(s.pos)["n_rebuilds"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ekin, epot, T = integ.step(args.steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms, n = f.read_timing(s.pos)
    st = f.stats(s.pos)
    pcut = f.count_pairs(s.pos, s.box)[0]
    print(f"N={mol.numAtoms} L={box[0]:.1f} A: {el / args.steps * 1e6:.1f} us/step = "
          f"{args.steps / el * 1e-6 * 86400:.1f} ns/day, T={T[0]:.1f} K, Epot/N={epot[0] / mol.numAtoms:.3f}, "
          f"pair kernel {ms / max(n, 1) * 1e3:.1f} us, P_cut={pcut} ({pcut / mol.numAtoms:.1f}/atom), "
          f"entries/atom {st['list_entries'] / mol.numAtoms:.1f}, rebuild every {args.steps / max(st['n_rebuilds'] - r0, 1):.1f} steps, "
          f"ncell {st['ncell']}, pair-interactions/s {pcut * args.steps / el:.3e}")


if __name__ == "__main__":
    main()
