"""Helper of tests/test_gpu_domain.py::test_native_loop_over_rccl_ranks — run under torch.distributed.run with N
ranks, ONE GPU PER RANK: the spatial domain decomposition through the library's own RCCL communicator
(`tmdhip_comm_exchange` between real ranks, `tmdhip_dd_run` enqueuing the step loop from C).

Every rank builds the same 10 648-atom argon box, keeps its brick, and
  (i)  checks one grouped exchange of the library against torch.distributed.all_to_all_single,
  (ii) runs 40 NVE steps (hot start: atoms migrate between bricks);
rank 0 then gathers positions / velocities / forces by atom id and compares them with the single-domain integrator on
its own GPU (fp64: 1e-7 / 1e-7 / 1e-6).  Prints `DD-RANKS OK world=N migrations=M` on success."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    from torchmd_amd import _lib as L
    from torchmd_amd.builders import argon_forcefield, lj_box
    from torchmd_amd.domain import DistTransport, DomainSet
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    ngpu = torch.cuda.device_count()
    assert ngpu >= world, f"this check wants one GPU per rank ({world} ranks, {ngpu} GPUs)"
    torch.cuda.set_device(local)
    dev, dt = torch.device("cuda", local), torch.float64
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        mol, pos, box = lj_box(22, seed=6)
        par = Parameters(argon_forcefield(mol), mol, ["lj"], precision=dt)
        n = mol.numAtoms
        torch.manual_seed(3)
        vel = maxwell_boltzmann(par.masses, 4000.0, 1)[0].numpy()
        A, B = par.get_AB()
        tr = DistTransport()
        ds = DomainSet(box, world, dev, dt, ["lj"], 9.0, A=A, B=B, skin=1.0, transport=tr)
        ds.scatter(pos, vel, par.charges.numpy(), par.mapped_atom_types.numpy(), par.masses.numpy().ravel())
        assert tr.native() is not None, "the library's RCCL communicator could not be created"
        d = next(iter(ds.domains.values()))
        # (i) one grouped exchange of the library against torch's all_to_all_single
        sent = d.pack_halo().clone()
        want = tr.all_to_all(sent, d.plan.send_counts, ds._recv_counts["halo"])
        got = torch.zeros_like(want)
        sc = (C.c_int64 * world)(*d.plan.send_counts)
        rc = (C.c_int64 * world)(*ds._recv_counts["halo"])
        L.check(L.load().tmdhip_comm_exchange(tr.native(), L.dtype_code(dt), sent.data_ptr(), sc, got.data_ptr(), rc, 3,
                                              torch.cuda.current_stream(dev).cuda_stream))
        torch.cuda.synchronize()
        assert got.shape[0] > 0 and torch.equal(got, want), "tmdhip_comm_exchange != all_to_all_single"
        # (ii) trajectory through tmdhip_dd_run
        ds.compute_forces()
        ds.step(25, timestep_fs=2.0)
        ds.step(15, timestep_fs=2.0)
        rows = torch.cat([d.ids.to(dt)[:, None], d.pos + d.unwrap, d.vel, d.forces], dim=1).contiguous()
        counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev))
        counts = [int(c.item()) for c in counts]
        assert sum(counts) == n
        parts = [torch.zeros(c, 10, dtype=dt, device=dev) for c in counts]
        # (all_gather of ragged rows through padding)
        pad = torch.zeros(max(counts), 10, dtype=dt, device=dev)
        pad[: rows.shape[0]] = rows
        padded = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(padded, pad)
        for r in range(world):
            parts[r] = padded[r][: counts[r]]
        migrations = ds.migrations
        if rank == 0:
            allrows = torch.cat(parts)
            order = torch.argsort(allrows[:, 0])
            allrows = allrows[order]
            assert torch.equal(allrows[:, 0].long(), torch.arange(n, device=dev)), "atom ids lost or duplicated"
            s = System(n, 1, dt, dev)
            s.set_positions(pos[:, :, None])
            s.set_box(box)
            s.set_velocities(torch.tensor(vel)[None])
            f = Forces(par, terms=["lj"], cutoff=9.0)
            f.compute(s.pos, s.box, s.forces)
            Integrator(s, f, 2.0, dev).step(40)
            ep = (allrows[:, 1:4] - s.pos[0]).abs().max().item()
            ev = (allrows[:, 4:7] - s.vel[0]).abs().max().item()
            ef = (allrows[:, 7:10] - s.forces[0]).abs().max().item()
            assert ep < 1e-7 and ev < 1e-7 and ef < 1e-6, (ep, ev, ef)
            assert world == 1 or migrations >= 1
            print(f"DD-RANKS OK world={world} migrations={migrations} max|dx|={ep:.2e} max|dv|={ev:.2e} max|dF|={ef:.2e}", flush=True)
        d.forces_engine.close()
        tr.close()
        dist.barrier(device_ids=[local])
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
