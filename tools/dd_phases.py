#!/usr/bin/env python
"""Where a domain-decomposition step spends its time (one rank, one brick exchanging its periodic images
with itself): wall-clock per phase with a device sync after each phase (needs a GPU)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from torchmd_amd import _lib as L  # noqa: E402
from torchmd_amd.builders import argon_forcefield, lj_box  # noqa: E402
from torchmd_amd.domain import DistTransport, DomainSet  # noqa: E402
from torchmd_amd.integrator import TIMEFACTOR, maxwell_boltzmann  # noqa: E402
from torchmd_amd.parameters import Parameters  # noqa: E402

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29513")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
mol, pos, box = lj_box(nside, seed=0)
par = Parameters(argon_forcefield(mol), mol, ["lj"], precision=torch.float32)
torch.manual_seed(1)
vel = maxwell_boltzmann(par.masses, 85.0, 1)[0].numpy()
A, B = par.get_AB()
ds = DomainSet(box, 1, dev, torch.float32, ["lj"], 9.0, A=A, B=B, skin=1.5, transport=DistTransport())
ds.scatter(pos, vel, par.charges.numpy(), par.mapped_atom_types.numpy(), par.masses.numpy().ravel())
ds.compute_forces()
ds.step(50, timestep_fs=1.0, gamma_ps=1.0, T=85.0, seed=3)
d = next(iter(ds.domains.values()))
lib = L.load()
code = L.dtype_code(torch.float32)
dt = 1.0 / TIMEFACTOR
acc = {}


def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return r


N = 100
st = lambda: torch.cuda.current_stream(dev).cuda_stream  # noqa: E731
counts = d.plan.send_counts


def dd_step(phases):
    L.check(lib.tmdhip_dd_step(code, d.nown, d.pos.data_ptr(), d.vel.data_ptr(), d.forces.data_ptr(), d.masses.data_ptr(),
                               0, dt, 0.0, 0, 0, phases, d.ref.data_ptr(), d.disp2.data_ptr(), st()))


for it in range(N):
    timed("dd_step (kick+drift+disp)", lambda: dd_step(3))
    due = timed("migration_check", ds._migration_due)
    if due:
        timed("migrate", ds.migrate)
        counts = d.plan.send_counts
    else:
        timed("pack", d.pack_halo)
        timed("all_to_all (in place)", lambda: ds.transport.all_to_all_into(d.halo_rows, d.send_buf, counts,
                                                                           ds._recv_counts["halo"]))
    timed("compute", ds.compute_forces)
print(f"own {d.nown} halo {d.local_pos.shape[1] - d.nown} migrations {ds.migrations}")
for k, v in acc.items():
    print(f"{k:28s} {v / N * 1e6:9.1f} us/step")
torch.cuda.synchronize()
t0 = time.perf_counter()
ds.step(N, timestep_fs=1.0)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"unsynchronised loop: {(time.perf_counter() - t0) / N * 1e6:.1f} us/step (host side of it: {(t1 - t0) / N * 1e6:.1f} us/step)")
if len(sys.argv) > 2:  # profile of one forced migration
    import cProfile
    import pstats

    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    ds.migrate()
    ds.compute_forces()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
d.forces_engine.close()
dist.destroy_process_group()
