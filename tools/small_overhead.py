#!/usr/bin/env python
"""Where a short step(K) call of a SMALL all-pairs system spends its wall time (host-bound regime): host time inside
tmdhip_md_run and tmdhip_md_observe, the Python around them, and the GPU span.  python tools/small_overhead.py [ala2|water291] [R] [K]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from _golden import GoldenParameters, load  # noqa: E402
from torchmd_amd import _lib as L  # noqa: E402
from torchmd_amd.forces import Forces  # noqa: E402
from torchmd_amd.integrator import Integrator, maxwell_boltzmann  # noqa: E402
from torchmd_amd.systems import System  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ala2"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
g = load(name)
dev = torch.device("cuda:0")
par = GoldenParameters(g, torch.float32)
n = len(g["pos"])
s = System(n, R, torch.float32, dev)
s.set_positions(g["pos"][:, :, None])
s.set_box(g["box"])
torch.manual_seed(1)
s.set_velocities(maxwell_boltzmann(par.masses, 300.0, R))
if name == "ala2":
    f = Forces(par, terms=["bonds", "angles", "dihedrals", "impropers", "1-4", "electrostatics", "lj"], cutoff=9.0, switch_dist=7.5, rfa=True)
else:
    f = Forces(par, terms=["lj", "bonds", "angles", "electrostatics"], cutoff=7.3)
f.compute(s.pos, s.box, s.forces)
integ = Integrator(s, f, 1.0, dev, gamma=0.1, T=300.0)
integ.step(200)
lib = L.load()
tm = {"run": 0.0, "obs": 0.0}
orig_run, orig_obs = lib.tmdhip_md_run, lib.tmdhip_md_observe


class Timed:
    def __init__(self, fn, key):
        self.fn, self.key = fn, key

    def __call__(self, *a):
        t = time.perf_counter()
        r = self.fn(*a)
        tm[self.key] += time.perf_counter() - t
        return r


lib.tmdhip_md_run = Timed(orig_run, "run")
lib.tmdhip_md_observe = Timed(orig_obs, "obs")
rows = []
for c in range(60):
    tm["run"] = tm["obs"] = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    integ.step(K)
    t1 = time.perf_counter()
    rows.append(((t1 - t0) * 1e6, tm["run"] * 1e6, tm["obs"] * 1e6))
a = np.array(rows[10:])
print(f"{name} x {R}, step({K}): wall {a[:,0].mean():.1f} us per call = {a[:,0].mean()/K:.2f} us/step; host time inside tmdhip_md_run {a[:,1].mean():.1f} us, "
      f"inside tmdhip_md_observe (incl. the wait for the device) {a[:,2].mean():.1f} us, Python around them {a[:,0].mean()-a[:,1].mean()-a[:,2].mean():.1f} us")
