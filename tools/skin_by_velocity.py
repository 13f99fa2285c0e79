"""Would a Verlet skin that depends on the atom's velocity at build time pay?  For a rebuild interval of T steps
every atom needs s_i >= max_{t<=T} |x_i(t) - x_i(0)|.  Model s_i = a |v_i(0)| T dt + b_class: for each a the
smallest b per class (O / H) that no atom violates, and the list size it implies (mean over pairs of
(rc + s_i + s_j)^3 relative to the uniform skin that survives the same T)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from torchmd_amd.integrator import Integrator, TIMEFACTOR

device = torch.device("cuda", 0)
mol, par, system, forces, box = bench.build_system(32, device, torch.float32, seed=1)
forces.compute(system.pos, system.box, system.forces)
Integrator(system, forces, bench.TIMESTEP_FS, device, gamma=10.0, T=300.0).step(1500)
integ = Integrator(system, forces, bench.TIMESTEP_FS, device, gamma=0.1, T=300.0)
integ.step(200)
isO = torch.tensor(np.asarray(par.masses).ravel() > 2.0, device=device)
rc = 9.0
for T in (9, 12, 15):
    res = {}
    for rep in range(6):
        ref = system.pos[0].clone()
        speed = system.vel[0].norm(dim=1) / TIMEFACTOR  # A per fs
        dmax = torch.zeros_like(speed)
        for n in range(T):
            integ.step(1)
            dmax = torch.maximum(dmax, (system.pos[0] - ref).norm(dim=1))
        for a in (0.0, 0.25, 0.5, 0.75, 1.0):
            pred = a * speed * T
            bO = (dmax - pred)[isO].max().item()
            bH = (dmax - pred)[~isO].max().item()
            s = pred + torch.where(isO, torch.full_like(pred, max(bO, 0.0)), torch.full_like(pred, max(bH, 0.0)))
            # mean over random pairs of (rc + s_i + s_j)^3
            idx = torch.randint(0, len(s), (400000, 2), device=device)
            vol = ((rc + s[idx[:, 0]] + s[idx[:, 1]]) ** 3).mean().item()
            res.setdefault(a, []).append((bO, bH, s.mean().item(), s.max().item(), vol))
    uni = None
    print(f"T = {T} steps")
    for a, rows in res.items():
        r = np.array(rows).mean(axis=0)
        if a == 0.0:
            uni = r[4]
        print(f"  a={a:4.2f}: floor O {r[0]:.3f} H {r[1]:.3f}  mean s {r[2]:.3f} max s {r[3]:.3f}  list volume {r[4]:8.1f} ({r[4]/uni:.3f} of per-class constant)")
print("reference: uniform skin 1.2 (s=0.6 for all): volume", (rc + 1.2) ** 3, "; per-class weights at skin 1.2:",
      (4 * (rc + 1.2) ** 3 + 4 * (rc + 0.6 + 0.173) ** 3 + (rc + 0.346) ** 3) / 9)
