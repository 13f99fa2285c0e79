"""Where a short step() call spends its wall time: GPU span (event before the call -> event behind it) against the
wall clock around the call, and the host time of the C entry points.  python tools/call_overhead.py [steps] [calls]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import bench
from torchmd_amd.integrator import Integrator

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
device = torch.device("cuda", 0)
mol, par, system, forces, box = bench.build_system(32, device, torch.float32, seed=1)
forces.compute(system.pos, system.box, system.forces)
Integrator(system, forces, bench.TIMESTEP_FS, device, gamma=10.0, T=300.0).step(1500)
integ = Integrator(system, forces, bench.TIMESTEP_FS, device, gamma=0.1, T=300.0)
integ.step(5)
# host time of tmdhip_md_run / tmdhip_md_observe
import torchmd_amd.forces as F
orig_run = F.Forces._md_run
tm = {"run": 0.0}
def timed_run(self, *a, **kw):
    t = time.perf_counter(); r = orig_run(self, *a, **kw); tm["run"] = time.perf_counter() - t; return r
F.Forces._md_run = timed_run
rows = []
for c in range(calls):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    integ.step(k)
    t1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    rows.append(((t1 - t0) * 1e6, e0.elapsed_time(e1) * 1e3, tm["run"] * 1e6))
a = np.array(rows[5:])
print(f"step({k}): wall {a[:,0].mean():.0f} us, GPU span (event -> event) {a[:,1].mean():.0f} us, host time inside tmdhip_md_run {a[:,2].mean():.0f} us")
print(f"  per step: wall {a[:,0].mean()/k:.1f}, span {a[:,1].mean()/k:.1f}")
