#!/usr/bin/env python
"""Small-system throughput (needs a GPU): the reference's only published number is the tutorial's MD loop
on alanine dipeptide (688 atoms, fp32, cutoff 9 + switch 7.5 + reaction field, all 7 terms, Langevin 1 fs):
2.7 ns/day on a CUDA GPU (examples/tutorial.ipynb:737).  Same system and settings here, from
tests/golden/ala2.npz; also tests/water (291 atoms x 2 replicas, water_conf.yaml settings)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from _golden import GoldenParameters, load  # noqa: E402
from torchmd_amd.forces import Forces  # noqa: E402
from torchmd_amd.integrator import Integrator, maxwell_boltzmann  # noqa: E402
from torchmd_amd.systems import System  # noqa: E402


def run(name, terms, R, steps=4000, tutorial_loop=False, per_call=10, dtype=torch.float32, **kw):
    g = load(name)
    dev = torch.device("cuda:0")
    par = GoldenParameters(g, dtype)
    n = len(g["pos"])
    s = System(n, R, dtype, dev)
    s.set_positions(g["pos"][:, :, None])
    s.set_box(g["box"])
    torch.manual_seed(1)
    s.set_velocities(maxwell_boltzmann(par.masses, 300.0, R))
    f = Forces(par, terms=terms, **kw)
    f.compute(s.pos, s.box, s.forces)
    integ = Integrator(s, f, 1.0, dev, gamma=0.1, T=300.0)
    integ.step(500)
    wrapper = logw = None
    if tutorial_loop:  # what the reference's tutorial does every 10 steps besides stepping (tutorial.ipynb:747-748)
        import tempfile

        from torchmd_amd.utils import LogWriter
        from torchmd_amd.wrapper import Wrapper

        wrapper = Wrapper(n, g["mol_bonds"], dev)
        logw = LogWriter(tempfile.mkdtemp(), keys=("iter", "ns", "epot", "ekin", "etot", "T"))
        traj = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps // per_call):
        ek, ep, T = integ.step(per_call)  # energies read back every `per_call` steps (10 = the tutorial's output period)
        if tutorial_loop:
            wrapper.wrap(s.pos, s.box)
            traj.append(s.pos.detach().cpu().numpy().copy())
            logw.write_row({"iter": i * 10, "ns": 1e-6 * i * 10, "epot": ep[0], "ekin": ek[0], "etot": ep[0] + ek[0], "T": T[0]})
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if tutorial_loop:
        name += " (tutorial loop: wrap + host copy + CSV row every 10 steps)"
    if dtype == torch.float64:
        name += " fp64"
    print(f"{name} [step({per_call}) calls]: {n} atoms x {R} replicas, {el / steps * 1e6:.1f} us/step = {steps / el * 1e-6 * 86400:.0f} ns/day per replica, "
          f"T={T[0]:.0f} K, Epot={ep[0]:.1f}, algorithm={f.stats(s.pos)['algorithm']}")


if __name__ == "__main__":
    all7 = ["bonds", "angles", "dihedrals", "impropers", "1-4", "electrostatics", "lj"]
    run("ala2", all7, 1, cutoff=9.0, switch_dist=7.5, rfa=True)
    run("ala2", all7, 1, per_call=100, cutoff=9.0, switch_dist=7.5, rfa=True)  # (a call's fixed cost — ~50 us — spread over 100 steps)
    run("ala2", all7, 1, per_call=100, dtype=torch.float64, cutoff=9.0, switch_dist=7.5, rfa=True)  # (C2's precision)
    run("ala2", all7, 1, tutorial_loop=True, cutoff=9.0, switch_dist=7.5, rfa=True)
    for R in (2, 16, 64):  # replicas share every launch (batched all-pairs / bonded / integrator kernels)
        run("water291", ["lj", "bonds", "angles", "electrostatics"], R, steps=2000, cutoff=7.3)
    run("ala2", all7, 16, steps=2000, cutoff=9.0, switch_dist=7.5, rfa=True)
