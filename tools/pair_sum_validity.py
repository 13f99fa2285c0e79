#!/usr/bin/env python
"""Would a pair-sum validity test stretch the life of a Verlet list?  (Needs a GPU; an offline measurement on the bench's own
trajectory — nothing in the library does this.)

Today a list is rebuilt when ONE atom has moved further than its half skin: g_i = |x_i(t) - x_i(t0)| - s_i > 0.  The list
holds every pair with |r_ij(t0)| <= rc + s_i + s_j, so it stays complete as long as g_i + g_j <= 0 for every pair of atoms
in neighbouring cells — an atom may overrun its own limit while its neighbours have not used theirs.  Per cell A let G_A be
the largest g of its atoms; the test `G_A + max over the stencil cells B of G_B <= 0` (B = A included: conservative) is exact
enough and needs per-cell maxima only.  This script follows the C3 box (98 304 atoms, the engine's own per-atom,
velocity-dependent skins: s_i = min(0.8 w_i skin/2 + |v_i| 6 fs, 0.72 A)) from a build step on and records, per cycle, the
step at which each rule fires."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import torch.nn.functional as F
import bench
from torchmd_amd.integrator import Integrator, TIMEFACTOR

dev = torch.device("cuda", 0)
mol, par, system, forces, box = bench.build_system(32, dev, torch.float32, seed=1)
forces.compute(system.pos, system.box, system.forces)
Integrator(system, forces, bench.TIMESTEP_FS, dev, gamma=10.0, T=300.0).step(1500)
integ = Integrator(system, forces, bench.TIMESTEP_FS, dev, gamma=0.1, T=300.0)
integ.step(200)
skin = 1.2
w = torch.tensor(forces._skin_weight_array(), device=dev, dtype=torch.float32)
static = 0.5 * skin * w
cap = 0.5 * skin * float(w.max()) * 1.2
L = torch.tensor(box, device=dev, dtype=torch.float32)
nc = int(np.floor(box[0] / ((9.0 + 2 * cap) / 2)))  # the engine's grid: cell edge >= rlist / 2, stencil +-2
print(f"grid {nc}^3, skin cap {cap:.3f} A", flush=True)
rows = []
for cycle in range(int(os.environ.get("CYCLES", "24"))):
    ref = system.pos[0].clone()
    speed = system.vel[0].norm(dim=1) / TIMEFACTOR  # A per fs
    s = torch.clamp(0.8 * static + 6.0 * speed, max=cap)
    cell = torch.floor((ref - torch.floor(ref / L) * L) / L * nc).long().clamp_(0, nc - 1)
    cid = (cell[:, 0] * nc + cell[:, 1]) * nc + cell[:, 2]
    t_atom = t_pair = None
    for t in range(1, 40):
        integ.step(1)
        g = (system.pos[0] - ref).norm(dim=1) - s
        if t_atom is None and g.max().item() > 0:
            t_atom = t
        G = torch.full((nc ** 3,), -10.0, device=dev).scatter_reduce(0, cid, g, reduce="amax").view(1, 1, nc, nc, nc)
        Gp = F.pad(G, (2, 2, 2, 2, 2, 2), mode="circular")
        M = F.max_pool3d(Gp, kernel_size=5, stride=1)  # max over the 5^3 stencil (periodic)
        if (G + M).max().item() > 0:
            t_pair = t
            break
    rows.append((t_atom, t_pair))
    print(f"cycle {cycle}: single-atom rule fires at step {t_atom}, pair-sum rule at step {t_pair}", flush=True)
a = np.array(rows, dtype=float)
print(f"mean life of a list: single-atom rule {a[:, 0].mean() - 1:.2f} steps, pair-sum rule {a[:, 1].mean() - 1:.2f} steps "
      f"(+{(a[:, 1].mean() - 1) / (a[:, 0].mean() - 1) * 100 - 100:.0f} %)")
