"""Distribution of the per-step time of SHORT step() calls (the driver's bench line times step(20)):
python tools/short_call.py [steps_per_call] [calls] [time every n-th pair launch]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from torchmd_amd.integrator import Integrator

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 60
every = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # > 0: pair-kernel event timing on, every n-th launch
device = torch.device("cuda", 0)
mol, par, system, forces, box = bench.build_system(32, device, torch.float32, seed=1)
forces.compute(system.pos, system.box, system.forces)
Integrator(system, forces, bench.TIMESTEP_FS, device, gamma=10.0, T=300.0).step(1500)
integ = Integrator(system, forces, bench.TIMESTEP_FS, device, gamma=0.1, T=300.0)
integ.step(5)
if every:
    forces.enable_timing(system.pos, True, every=every)
rows = []
pause = float(os.environ.get("SHORT_CALL_PAUSE_MS", "0")) * 1e-3  # idle time in front of every call (clock ramp?)
for c in range(calls):
    r0 = forces.stats(system.pos)["n_rebuilds"]
    torch.cuda.synchronize()
    if pause:
        time.sleep(pause)
    t0 = time.perf_counter()
    integ.step(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if every:
        forces.read_timing(system.pos, reset=True)
    rows.append((dt / k * 1e6, forces.stats(system.pos)["n_rebuilds"] - r0))
a = np.array(rows)
print(f"step({k}) x {calls}: us/step mean {a[:,0].mean():.1f} min {a[:,0].min():.1f} max {a[:,0].max():.1f}; first call {a[0,0]:.1f}")
for nr in sorted(set(a[:, 1])):
    s = a[a[:, 1] == nr, 0]
    print(f"  {int(nr)} rebuilds: {len(s)} calls, mean {s.mean():.1f} us/step")
