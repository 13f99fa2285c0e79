import os, sys, time
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from _golden import GoldenParameters, load
from torchmd_amd.forces import Forces
from torchmd_amd.integrator import Integrator, maxwell_boltzmann
from torchmd_amd.systems import System
g = load("thrombin"); dev = torch.device("cuda:0")
par = GoldenParameters(g, torch.float32)
pos = np.asarray(g["pos"], dtype=np.float64)
terms = ["bonds", "angles", "dihedrals", "impropers", "1-4", "electrostatics", "lj"]
s = System(pos.shape[0], 1, torch.float32, dev); s.set_positions(pos[:, :, None]); s.set_box(np.zeros(3))
torch.manual_seed(1); s.set_velocities(maxwell_boltzmann(par.masses, 300.0, 1))
f = Forces(par, terms=terms, cutoff=9.0, algorithm="celllist")
f.compute(s.pos, s.box, s.forces)
integ = Integrator(s, f, 1.0, dev, gamma=1.0, T=300.0)
integ.step(300)
st0 = f.stats(s.pos)
torch.cuda.synchronize(); t0 = time.perf_counter(); integ.step(3000); torch.cuda.synchronize(); el = time.perf_counter() - t0
st = f.stats(s.pos)
print(f"thrombin 4676 atoms celllist: {el/3000*1e6:.1f} us/step, fused steps {st['steps_in_pair_launch']-st0['steps_in_pair_launch']}, rebuilds {st['n_rebuilds']-st0['n_rebuilds']}")
f.close()
# the reference's own use of this fixture: no cutoff (all pairs, tests/test_torchmd.py:297-466), fp32 and fp64
for dt in (torch.float32, torch.float64):
    par = GoldenParameters(g, dt)
    s = System(pos.shape[0], 1, dt, dev); s.set_positions(pos[:, :, None]); s.set_box(np.zeros(3))
    torch.manual_seed(1); s.set_velocities(maxwell_boltzmann(par.masses, 300.0, 1))
    f = Forces(par, terms=terms)
    f.compute(s.pos, s.box, s.forces)
    integ = Integrator(s, f, 1.0, dev, gamma=1.0, T=300.0)
    integ.step(100)
    torch.cuda.synchronize(); t0 = time.perf_counter(); integ.step(500); torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"thrombin 4676 atoms, no cutoff (all pairs), {'fp64' if dt == torch.float64 else 'fp32'}: {el/500*1e6:.1f} us/step, algorithm {f.stats(s.pos)['algorithm']}")
    f.close()
