#!/usr/bin/env python
"""True-reference CPU baseline at full C3 size: the reference's own `Forces` and
`Integrator` classes, unmodified except for ONE override — `_make_indeces` (torchmd/forces.py:348-357)
returns the non-excluded i<j pairs within cutoff + margin (cKDTree) instead of all N(N-1)/2 pairs, which
would need 77 GB at N = 98 304.  `compute()` still re-filters by `dist <= cutoff` (forces.py:266-269), so the
arithmetic is the reference's, bit for bit.  Times `Integrator.step` on this container's
cores; the candidate list is rebuilt outside the timed region (favourable to the CPU).

    python tools/ref_cpu_sparse.py [--nside 32] [--steps 3]  ->  profiles/r02_ref_cpu_sparse.json

Needs /root/reference (build container only); not used by tests, smoke or bench."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from torchmd.forces import Forces as RefForces  # noqa: E402
from torchmd.integrator import Integrator, maxwell_boltzmann  # noqa: E402
from torchmd.systems import System  # noqa: E402

from oracle import torchmd_oracle as orc  # noqa: E402  (candidate-pair search only)
from torchmd_amd.builders import tip3p_box, water_forcefield  # noqa: E402
from torchmd_amd.parameters import Parameters  # noqa: E402


class SparseForces(RefForces):
    """Reference Forces with a sparse candidate pair list; everything else inherited."""

    candidate_pairs = None

    def _make_indeces(self, natoms, excludepairs, device):
        return torch.as_tensor(self.candidate_pairs, dtype=torch.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nside", type=int, default=32)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--margin", type=float, default=0.6)
    args = ap.parse_args()
    mol, pos, box = tip3p_box(args.nside, seed=0)
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=torch.float32)
    s = System(mol.numAtoms, 1, torch.float32, "cpu")
    s.set_positions(pos[:, :, None])
    s.set_box(box)
    torch.manual_seed(1)
    s.set_velocities(maxwell_boltzmann(par.masses, 300.0, 1))
    t0 = time.perf_counter()
    SparseForces.candidate_pairs = orc.candidate_pairs(pos, box, 9.0 + args.margin, orc.exclusion_pairs(par))
    t_list = time.perf_counter() - t0
    f = SparseForces(par, terms=terms, cutoff=9.0, rfa=True)
    it = Integrator(s, f, 1.0, "cpu", gamma=0.1, T=300.0)
    f.compute(s.pos, s.box, s.forces)
    it.step(1)  # warm-up
    t0 = time.perf_counter()
    ekin, pot, T = it.step(args.steps)
    dt = (time.perf_counter() - t0) / args.steps
    out = {
        "what": "reference torchmd Forces+Integrator (only _make_indeces overridden: sparse candidate pairs), CPU",
        "natoms": int(mol.numAtoms),
        "candidate_pairs": int(len(SparseForces.candidate_pairs)),
        "threads": torch.get_num_threads(),
        "s_per_step": dt,
        "ns_per_day": 86400e-6 / dt,
        "candidate_list_build_s": t_list,
        "epot": float(pot[0]),
        "host": "build container (no GPU)",
    }
    print(json.dumps(out))
    with open(os.path.join(ROOT, "profiles", "r02_ref_cpu_sparse.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
