#!/usr/bin/env python
"""A few step(K) calls of a small all-pairs system for a kernel trace (tools/call_timeline.py reads it):
    cd /tmp && rocprofv3 --kernel-trace -d /tmp/sc -- python $REPO/tools/small_calls.py [ala2|water291|thrombin] [R] [K] [calls] [f32|f64]"""
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

name = sys.argv[1] if len(sys.argv) > 1 else "ala2"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 40
dt = torch.float64 if (len(sys.argv) > 5 and sys.argv[5] == "f64") else torch.float32
g = load(name)
dev = torch.device("cuda:0")
par = GoldenParameters(g, dt)
n = len(g["pos"])
s = System(n, R, dt, dev)
s.set_positions(g["pos"][:, :, None])
s.set_box(g["box"])
torch.manual_seed(1)
s.set_velocities(maxwell_boltzmann(par.masses, 300.0, R))
if name == "thrombin":  # 4 676-atom complex, no cutoff, no box: the all-pairs kernel at its largest
    f = Forces(par, terms=["bonds", "angles", "dihedrals", "impropers", "1-4", "electrostatics", "lj"])
    s.set_box([0.0, 0.0, 0.0])
elif name == "ala2":
    f = Forces(par, terms=["bonds", "angles", "dihedrals", "impropers", "1-4", "electrostatics", "lj"], cutoff=9.0, switch_dist=7.5, rfa=True)
else:
    f = Forces(par, terms=["lj", "bonds", "angles", "electrostatics"], cutoff=7.3)
f.compute(s.pos, s.box, s.forces)
integ = Integrator(s, f, 1.0, dev, gamma=0.1, T=300.0)
integ.step(200)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(calls):
    integ.step(K)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"{name} x {R} {'f64' if dt == torch.float64 else 'f32'}: step({K}) x {calls}: {el / (K * calls) * 1e6:.2f} us/step")
