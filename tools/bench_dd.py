#!/usr/bin/env python
"""Config C5: Lennard-Jones box, spatial domain decomposition with RCCL halo exchange, one rank per GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29512 tools/bench_dd.py --nside 100 --steps 200

Prints one JSON line (rank 0): us/step, ns/day, atoms and halo sizes per rank, migrations.
(With --nproc-per-node 1 the single brick exchanges its periodic images with itself through the same
RCCL all-to-all, which exercises the transport on one GPU.)
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from torchmd_amd.builders import argon_forcefield, lj_box  # noqa: E402
from torchmd_amd.domain import DistTransport, DomainSet  # noqa: E402
from torchmd_amd.integrator import maxwell_boltzmann  # noqa: E402
from torchmd_amd.parameters import Parameters  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nside", type=int, default=100)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--skin", type=float, default=2.5, help="halo skin in A (migration when an atom has moved skin/2)")
    args = ap.parse_args()
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    mol, pos, box = lj_box(args.nside, seed=0)
    par = Parameters(argon_forcefield(mol), mol, ["lj"], precision=torch.float32)
    torch.manual_seed(1)
    vel = maxwell_boltzmann(par.masses, 85.0, 1)[0].numpy()
    A, B = par.get_AB()
    ds = DomainSet(box, world, dev, torch.float32, ["lj"], 9.0, A=A, B=B, skin=args.skin, transport=DistTransport())
    ds.scatter(pos, vel, par.charges.numpy(), par.mapped_atom_types.numpy(), par.masses.numpy().ravel())
    ds.compute_forces()
    ds.step(args.warmup, timestep_fs=1.0, gamma_ps=1.0, T=85.0, seed=3)
    torch.cuda.synchronize()
    dist.barrier(device_ids=[local])
    m0 = ds.migrations
    mig_t = [0.0]
    plain_migrate = ds.migrate

    def timed_migrate():  # wall time of a migration incl. everything still queued in front of it
        torch.cuda.synchronize()
        t = time.perf_counter()
        plain_migrate()
        ds.compute_forces()
        torch.cuda.synchronize()
        mig_t[0] += time.perf_counter() - t

    if os.environ.get("DD_TIME_MIGRATIONS"):
        ds.migrate = timed_migrate
    t0 = time.perf_counter()
    ds.step(args.steps, timestep_fs=1.0, gamma_ps=1.0, T=85.0, seed=3)
    torch.cuda.synchronize()
    dist.barrier(device_ids=[local])
    el = time.perf_counter() - t0
    d = next(iter(ds.domains.values()))
    info = torch.tensor([d.nown, d.local_pos.shape[1] - d.nown], dtype=torch.float64, device=dev)
    allinfo = [torch.empty_like(info) for _ in range(world)]
    dist.all_gather(allinfo, info)
    if rank == 0:
        print(json.dumps({
            "workload": f"C5 LJ box {mol.numAtoms} atoms, L={box[0]:.1f} A, cutoff 9 A, Langevin 85 K, 1 fs; grid {ds.grid.dims}",
            "n_gpus": world, "us_per_step": el / args.steps * 1e6, "ns_per_day": args.steps / el * 1e-6 * 86400,
            "own_atoms": [int(x[0]) for x in allinfo], "halo_atoms": [int(x[1]) for x in allinfo],
            "migrations_in_timed_region": ds.migrations - m0, "steps": args.steps,
            "migration_ms_total": mig_t[0] * 1e3,
        }), flush=True)
    for dom in ds.domains.values():
        dom.forces_engine.close()
    dist.barrier(device_ids=[local])
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
