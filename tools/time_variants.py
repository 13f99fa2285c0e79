#!/usr/bin/env python
"""The siblings of the headline launch at C3 (needs a GPU): us per MD step of step(2000) without / with the LJ switching
function (reference and exact force flavour), the cost of an evaluation with energies (`compute()`), of `step(1)` and
`step(10)` calls.  TMDHIP_LIB selects the library (A/B); VARIANTS_STEPS the length of the timed runs."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import build_system
from torchmd_amd.forces import Forces
from torchmd_amd.integrator import Integrator

dev = torch.device("cuda:0")
steps = int(os.environ.get("VARIANTS_STEPS", "2000"))
mol, par, system, forces, box = build_system(32, dev, torch.float32, seed=1)
forces.compute(system.pos, system.box, system.forces)
Integrator(system, forces, 1.0, dev, gamma=10.0, T=300.0).step(1500)
terms = ["lj", "electrostatics", "bonds", "angles"]


def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print("library:", os.environ.get("TMDHIP_LIB", "default"), flush=True)
for kw in (dict(), dict(switch_dist=7.5), dict(switch_dist=7.5, switch_mode="exact")):
    f = Forces(par, terms=terms, cutoff=9.0, rfa=True, skin_weights="mass", **kw)
    f.compute(system.pos, system.box, system.forces)
    it = Integrator(system, f, 1.0, dev, gamma=0.1, T=300.0)
    it.step(200)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ek, ep, T = it.step(steps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    F = torch.zeros_like(system.pos)
    ce = timed(lambda: f.compute(system.pos, system.box, F), 100)
    s1 = timed(lambda: it.step(1), 200)
    s10 = timed(lambda: it.step(10), 50) / 10
    print(f"{str(kw):50s} {dt / steps * 1e6:6.1f} us/step  compute()+energies {ce:6.1f} us  step(1) {s1:6.1f} us  "
          f"step(10) {s10:6.1f} us/step  T={T[0]:.1f} Epot={ep[0]:.1f}", flush=True)
    f.close()
