import sys, time, torch
sys.path.insert(0, '/root/repo')
from bench import build_system
from torchmd_amd.forces import Forces
from torchmd_amd.integrator import Integrator
dev = torch.device("cuda:0")
mol, par, system, forces, box = build_system(32, dev, torch.float32, seed=1)
forces.compute(system.pos, system.box, system.forces)
integ = Integrator(system, forces, 1.0, dev, gamma=10.0, T=300.0); integ.step(600)
def timed(fn, n):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
F = torch.zeros_like(system.pos)
print("compute() with energies us:", timed(lambda: forces.compute(system.pos, system.box, F), 100))
print("step(1) us:", timed(lambda: integ.step(1), 200))
print("step(10) us/step:", timed(lambda: integ.step(10), 50)/10)
print(forces.compute(system.pos, system.box, F, returnDetails=True))
