import sys, time, torch
sys.path.insert(0, '.')
from bench import build_system
from torchmd_amd.forces import Forces
from torchmd_amd.integrator import Integrator
dev = torch.device("cuda:0")
mol, par, system, forces, box = build_system(32, dev, torch.float32, seed=1)
forces.compute(system.pos, system.box, system.forces)
Integrator(system, forces, 1.0, dev, gamma=10.0, T=300.0).step(1500)
for kw in (dict(), dict(switch_dist=7.5), dict(switch_dist=7.5, switch_mode="exact")):
    f = Forces(par, terms=["lj", "electrostatics", "bonds", "angles"], cutoff=9.0, rfa=True, **kw)
    f.compute(system.pos, system.box, system.forces)
    it = Integrator(system, f, 1.0, dev, gamma=0.1, T=300.0)
    it.step(200)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ek, ep, T = it.step(2000)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(kw, f"{dt/2000*1e6:.1f} us/step  T={T[0]:.1f} Epot={ep[0]:.1f}")
