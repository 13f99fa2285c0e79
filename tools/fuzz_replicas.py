#!/usr/bin/env python
"""Random configurations of the replica batch on the cell-list path against the replica-by-replica loop (needs a GPU):

    python tools/fuzz_replicas.py [cases] [seed]

Each case draws a water box (10..14 molecules per edge), 2..6 replicas (or 17..18: two launches per step), thermostat or not,
LJ switch or not, call lengths, lanes per atom, jitter and velocities, and compares positions / velocities / forces of the
batched run with TMDHIP_BATCH_REPLICAS=0 bit for bit and the returned energies to 1e-12 / 2e-7.  Every third case also runs
TMDHIP_REPLICA_REBUILDS=together and checks the forces of its final state against a fresh evaluation (6e-4)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from torchmd_amd.builders import tip3p_box, water_forcefield
from torchmd_amd.forces import Forces
from torchmd_amd.integrator import Integrator
from torchmd_amd.parameters import Parameters
from torchmd_amd.systems import System

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev, dt = torch.device("cuda:0"), torch.float32
terms = ["lj", "electrostatics", "bonds", "angles"]
bad = 0
for case in range(cases):
    nside = int(rng.integers(10, 15))
    R = int(rng.choice([2, 3, 4, 5, 6, 17, 18], p=[0.2, 0.2, 0.2, 0.15, 0.1, 0.075, 0.075]))
    if R > 6:
        nside = 10
    langevin, switch = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    calls = [int(x) for x in rng.integers(1, 40, size=int(rng.integers(1, 4)))]
    lpa = int(rng.choice([8, 16, 32]))
    mol, pos0, box0 = tip3p_box(nside, seed=int(rng.integers(0, 100)))
    par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
    starts = np.stack([pos0 + 0.04 * (1 + r % 3) * rng.standard_normal(pos0.shape) for r in range(R)], axis=2)
    vels = torch.tensor(np.stack([0.02 * (1 + r % 4) * rng.standard_normal(pos0.shape) for r in range(R)]))
    kw = dict(cutoff=9.0, rfa=True, algorithm="celllist", **({"switch_dist": 7.5} if switch else {}))
    os.environ["TMDHIP_LPA"] = str(lpa)

    def run(batch, together=False):
        os.environ["TMDHIP_BATCH_REPLICAS"] = "1" if batch else "0"
        if together:
            os.environ["TMDHIP_REPLICA_REBUILDS"] = "together"
        else:
            os.environ.pop("TMDHIP_REPLICA_REBUILDS", None)
        s = System(mol.numAtoms, R, dt, dev)
        s.set_positions(starts)
        s.set_box(box0)
        s.set_velocities(vels)
        f = Forces(par, terms=terms, **kw)
        f.compute(s.pos, s.box, s.forces)
        torch.manual_seed(3)
        integ = Integrator(s, f, 1.0, dev, **(dict(gamma=0.5, T=300.0) if langevin else {}))
        out = [integ.step(k) for k in calls]
        st = f.stats(s.pos)
        res = (s.pos.clone(), s.vel.clone(), s.forces.clone(), out, st, integ.replays, s.box.clone())
        f.close()
        return res

    pb, vb, fb, ob, stb, rb, _ = run(True)
    ps, vs, fs, os_, sts, rs, _ = run(False)
    ok = torch.equal(pb, ps) and torch.equal(vb, vs) and torch.equal(fb, fs) and rb == rs and bool(torch.isfinite(pb).all())
    for a, b in zip(ob, os_):
        ok = ok and np.allclose(a[1], b[1], rtol=1e-12) and np.allclose(a[0], b[0], rtol=2e-7)
    tog = ""
    if case % 3 == 0:
        pt, vt, ft, ot, stt, rt, bx = run(True, together=True)
        os.environ.pop("TMDHIP_REPLICA_REBUILDS", None)
        fresh = Forces(par, terms=terms, **kw)
        F2 = torch.zeros_like(pt)
        fresh.compute(pt, bx, F2)
        fresh.close()
        err = (F2 - ft).abs().max().item()
        ok = ok and err < 6e-4 and stt["overflow"] == 0 and bool(torch.isfinite(pt).all())
        tog = f" together: max|dF| {err:.1e}, rebuilds {stt['n_rebuilds']} (default {stb['n_rebuilds']}), replays {rt}"
    bad += not ok
    print(f"case {case}: {mol.numAtoms} atoms x {R}, {'langevin' if langevin else 'nve'}{' switch' if switch else ''}, calls {calls}, lpa {lpa}: "
          f"{'OK' if ok else 'MISMATCH'} (batched launches {stb['batched_launches']}, rebuilds {stb['n_rebuilds']}, replays {rb}){tog}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
