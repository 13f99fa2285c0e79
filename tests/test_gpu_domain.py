"""Spatial domain decomposition (config C5) validated on ONE GPU: all ranks run inside one process
(`LocalTransport`), forces and a short trajectory must equal the single-domain periodic engine."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _system(nside, dtype):
    from torchmd_amd.builders import argon_forcefield, lj_box
    from torchmd_amd.parameters import Parameters

    mol, pos, box = lj_box(nside, seed=6)
    par = Parameters(argon_forcefield(mol), mol, ["lj"], precision=dtype)
    return mol, pos, box, par


@pytest.mark.parametrize("world,grid", [(8, None), (4, (2, 1, 2)), (2, None), (1, None)])
def test_forces_equal_single_domain(world, grid):
    from torchmd_amd.domain import DomainSet
    from torchmd_amd.forces import Forces

    dev, dt = torch.device("cuda:0"), torch.float64
    mol, pos, box, par = _system(22, dt)  # 10 648 atoms, L = 79.3 A
    rng = np.random.default_rng(0)
    pos = pos + rng.integers(-1, 2, size=pos.shape) * box  # scatter atoms over periodic images
    n = mol.numAtoms
    A, B = par.get_AB()
    ds = DomainSet(box, world, dev, dt, ["lj"], 9.0, A=A, B=B, skin=1.5, grid=grid)
    ds.scatter(pos, np.zeros_like(pos), par.charges.numpy(), par.mapped_atom_types.numpy(), par.masses.numpy().ravel())
    assert sum(d.nown for d in ds.domains.values()) == n
    ds.compute_forces()
    _, _, F = ds.gather(n)
    ref = Forces(par, terms=["lj"], cutoff=9.0)
    p = torch.tensor(pos, dtype=dt, device=dev)[None].contiguous()
    b = torch.diag(torch.tensor(box, dtype=dt, device=dev))[None].contiguous()
    Fr = torch.zeros_like(p)
    ref.compute(p, b, Fr)
    assert (F - Fr[0]).abs().max().item() < 1e-9
    halo = [d.local_pos.shape[1] - d.nown for d in ds.domains.values()]
    assert all(h > 0 for h in halo)
    # energy: halo atoms are passive, so a brick counts its own-own pairs fully and own-halo pairs half;
    # the bricks' energies add up to the single-domain total
    from torchmd_amd import _lib as L

    e_dd = sum(float(d.compute(want_energy=True)[0, L.ENERGY_SLOT["lj"]].item()) for d in ds.domains.values())
    e_ref = ref.compute(p, b, Fr, returnDetails=True)[0]["lj"]
    assert abs(e_dd - e_ref) <= 1e-9 * abs(e_ref)
    # and no force is computed on the halo atoms
    for d in ds.domains.values():
        assert d.local_forces[0, d.nown:].abs().max().item() == 0.0


def test_trajectory_equal_single_domain():
    """40 NVE steps (hot start so that atoms migrate between bricks): positions/velocities agree with
    the single-domain integrator to fp64 round-off."""
    from torchmd_amd.domain import DomainSet
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.systems import System

    dev, dt = torch.device("cuda:0"), torch.float64
    mol, pos, box, par = _system(22, dt)
    n = mol.numAtoms
    torch.manual_seed(3)
    vel = maxwell_boltzmann(par.masses, 4000.0, 1)[0].numpy()  # hot: ~0.05 A per step
    A, B = par.get_AB()
    ds = DomainSet(box, 8, dev, dt, ["lj"], 9.0, A=A, B=B, skin=1.0)
    ds.scatter(pos, vel, par.charges.numpy(), par.mapped_atom_types.numpy(), par.masses.numpy().ravel())
    ds.compute_forces()
    ds.step(40, timestep_fs=2.0)
    P, V, F = ds.gather(n)

    s = System(n, 1, dt, dev)
    s.set_positions(pos[:, :, None])
    s.set_box(box)
    s.set_velocities(torch.tensor(vel)[None])
    f = Forces(par, terms=["lj"], cutoff=9.0)
    f.compute(s.pos, s.box, s.forces)
    Integrator(s, f, 2.0, dev).step(40)
    assert ds.migrations >= 1
    assert (P - s.pos[0]).abs().max().item() < 1e-7
    assert (V - s.vel[0]).abs().max().item() < 1e-7
    assert (F - s.forces[0]).abs().max().item() < 1e-6


@pytest.mark.timeout(300)
def test_native_exchange_and_step_loop_single_rank():
    """The C-driven path (`tmdhip_dd_run`: fused kick/drift kernel, device-side halo pack, grouped RCCL
    send/recv into the halo rows, forces — all enqueued from C) on a 1-rank RCCL communicator, where the brick
    exchanges its periodic images with itself: (i) `tmdhip_comm_exchange` reproduces the torch all-to-all,
    (ii) a 40-step NVE trajectory with migrations equals the single-domain integrator, (iii) and equals the
    Python-driven loop over torch.distributed (and its torch-made migrations) to 1e-9."""
    import os

    import torch.distributed as dist

    from torchmd_amd import _lib as L
    from torchmd_amd.domain import DistTransport, DomainSet
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.systems import System

    dev, dt = torch.device("cuda:0"), torch.float64
    mol, pos, box, par = _system(22, dt)
    n = mol.numAtoms
    torch.manual_seed(3)
    vel = maxwell_boltzmann(par.masses, 4000.0, 1)[0].numpy()
    A, B = par.get_AB()
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:  # a port nobody listens on right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        results = {}
        for native in ("1", "0"):
            os.environ["TMDHIP_DD_NATIVE"] = native
            tr = DistTransport()
            ds = DomainSet(box, 1, dev, dt, ["lj"], 9.0, A=A, B=B, skin=1.0, transport=tr)
            ds.scatter(pos, vel, par.charges.numpy(), par.mapped_atom_types.numpy(), par.masses.numpy().ravel())
            assert (tr.native() is not None) == (native == "1")
            d = ds.domains[0]
            if native == "1":  # (i) one grouped exchange against torch's all_to_all_single
                import ctypes as C

                sent = d.pack_halo().clone()
                want = tr.all_to_all(sent, d.plan.send_counts, ds._recv_counts["halo"])
                got = torch.zeros_like(want)
                sc = (C.c_int64 * 1)(*d.plan.send_counts)
                rc = (C.c_int64 * 1)(*ds._recv_counts["halo"])
                L.check(L.load().tmdhip_comm_exchange(tr.native(), L.dtype_code(dt), sent.data_ptr(), sc, got.data_ptr(), rc, 3,
                                                      torch.cuda.current_stream(dev).cuda_stream))
                torch.cuda.synchronize()
                assert got.shape[0] > 0 and torch.equal(got, want)
            ds.compute_forces()
            ds.step(25, timestep_fs=2.0)
            ds.step(15, timestep_fs=2.0)
            results[native] = ds.gather(n) + (ds.migrations,)
            d.forces_engine.close()
            tr.close()
    finally:
        os.environ.pop("TMDHIP_DD_NATIVE", None)
        dist.destroy_process_group()
    P, V, F, mig = results["1"]
    assert mig >= 1
    s = System(n, 1, dt, dev)
    s.set_positions(pos[:, :, None])
    s.set_box(box)
    s.set_velocities(torch.tensor(vel)[None])
    f = Forces(par, terms=["lj"], cutoff=9.0)
    f.compute(s.pos, s.box, s.forces)
    Integrator(s, f, 2.0, dev).step(40)
    assert (P - s.pos[0]).abs().max().item() < 1e-7
    assert (V - s.vel[0]).abs().max().item() < 1e-7
    assert (F - s.forces[0]).abs().max().item() < 1e-6
    # the Python-driven loop migrates with torch operations and plans the engine's cell grid over the observed bounding
    # box; the library's migration (tmdhip_dd_migrate) plans it over brick + halo: same pairs, another summation order
    P0, V0, F0, mig0 = results["0"]
    assert mig0 == mig
    assert (P - P0).abs().max().item() < 1e-9 and (V - V0).abs().max().item() < 1e-9 and (F - F0).abs().max().item() < 1e-8


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_native_loop_at_world_2_4_8_on_one_gpu(world):
    """The library's own brick loop (`tmdhip_dd_run`: kick/drift, halo pack, exchange into the halo rows, forces, the
    asynchronous migration trigger with its max-reduction over the ranks) at world 2 / 4 / 8 on ONE GPU: one host
    thread and one stream per rank, the ranks exchange through the library's in-process communicator
    (`tmdhip_comm_create_local`: device copies ordered by events around a host barrier — the stand-in for RCCL where
    there is one GPU).  (i) one grouped exchange from `world` threads equals the reference all-to-all; (ii) 40 hot NVE
    steps with migrations equal the single-domain integrator (fp64: 1e-7 / 1e-7 / 1e-6) and (iii) the Python-driven
    loop over the same kernels (same migration decisions; 1e-9)."""
    import ctypes as C
    import threading

    from torchmd_amd import _lib as L
    from torchmd_amd.domain import DomainSet, LocalTransport
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.systems import System

    dev, dt = torch.device("cuda:0"), torch.float64
    mol, pos, box, par = _system(23, dt)  # (odd: brick faces pass through lattice planes, atoms change bricks)
    n = mol.numAtoms
    torch.manual_seed(3)
    vel = maxwell_boltzmann(par.masses, 4000.0, 1)[0].numpy()
    A, B = par.get_AB()
    results = {}
    for native in (True, False):
        tr = LocalTransport(world, native_threads=native)
        ds = DomainSet(box, world, dev, dt, ["lj"], 9.0, A=A, B=B, skin=1.0, transport=tr)
        ds.scatter(pos, vel, par.charges.numpy(), par.mapped_atom_types.numpy(), par.masses.numpy().ravel())
        if native:  # (i) tmdhip_comm_exchange from one thread per rank
            lib = L.load()
            comms, streams = tr.native(dev)
            sent = {r: d.pack_halo().clone() for r, d in ds.domains.items()}
            counts = {r: d.plan.send_counts for r, d in ds.domains.items()}
            want = ds._all_to_all("probe", sent, counts)
            got = {r: torch.zeros_like(want[r]) for r in ds.domains}
            torch.cuda.synchronize()
            rcs = {}

            def work(r):
                sc = (C.c_int64 * world)(*counts[r])
                rc = (C.c_int64 * world)(*[counts[src][r] for src in range(world)])
                with torch.cuda.device(dev):
                    rcs[r] = lib.tmdhip_comm_exchange(comms[r], L.dtype_code(dt), sent[r].data_ptr(), sc, got[r].data_ptr(),
                                                      rc, 3, streams[r].cuda_stream)

            ths = [threading.Thread(target=work, args=(r,)) for r in ds.domains]
            [th.start() for th in ths]
            [th.join() for th in ths]
            torch.cuda.synchronize()
            assert all(v == 0 for v in rcs.values()), rcs
            for r in ds.domains:
                assert got[r].shape[0] > 0 and torch.equal(got[r], want[r]), r
        ds.compute_forces()
        ds.step(25, timestep_fs=2.0)
        ds.step(15, timestep_fs=2.0)
        results[native] = ds.gather(n) + (ds.migrations,)
        for d in ds.domains.values():
            d.forces_engine.close()
        tr.close()
    P, V, F, mig = results[True]
    assert mig >= 1
    s = System(n, 1, dt, dev)
    s.set_positions(pos[:, :, None])
    s.set_box(box)
    s.set_velocities(torch.tensor(vel)[None])
    f = Forces(par, terms=["lj"], cutoff=9.0)
    f.compute(s.pos, s.box, s.forces)
    Integrator(s, f, 2.0, dev).step(40)
    ep, ev, ef = (P - s.pos[0]).abs().max().item(), (V - s.vel[0]).abs().max().item(), (F - s.forces[0]).abs().max().item()
    P0, V0, F0, mig0 = results[False]
    same = torch.equal(P, P0) and torch.equal(V, V0) and torch.equal(F, F0)
    print(f"world {world}: migrations {mig} (Python-driven loop: {mig0}), vs single domain max|dx| {ep:.2e} max|dv| {ev:.2e} "
          f"max|dF| {ef:.2e}; bit-identical to the Python-driven loop: {same}")
    assert ep < 1e-7 and ev < 1e-7 and ef < 1e-6
    assert (P - P0).abs().max().item() < 1e-9 and (V - V0).abs().max().item() < 1e-9 and (F - F0).abs().max().item() < 1e-8


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_native_loop_over_rccl_ranks(world):
    """`tmdhip_comm_exchange` and `tmdhip_dd_run` between REAL ranks, one GPU each (tests/_dd_ranks.py under
    torch.distributed.run): the library's grouped RCCL send/recv equals torch's all_to_all_single, and 40 NVE steps
    with migrations equal the single-domain integrator.  World sizes the box has no GPUs for are skipped (the
    1-GPU pool runs world = 1: the brick exchanges its periodic images with itself over RCCL)."""
    import os
    import socket
    import subprocess
    import sys

    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {torch.cuda.device_count()}")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("TMDHIP_DD_NATIVE", None)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(here, "_dd_ranks.py")],
                         capture_output=True, text=True, timeout=550, env=env)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    assert f"DD-RANKS OK world={world}" in res.stdout, res.stdout[-1500:]


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("langevin", [False, True])
def test_dd_step_equals_the_separate_integrator_kernels(dt, langevin):
    """`tmdhip_dd_step` (kick of the previous step + drift of this one in one launch) is bit-identical to
    tmdhip_[langevin_]second_vv followed by tmdhip_first_vv, and its displacement maximum is max |x - ref|^2;
    `tmdhip_halo_pack` equals index_select + shift."""
    from torchmd_amd import _lib as L

    lib, dev = L.load(), torch.device("cuda:0")
    code = L.dtype_code(dt)
    n = 5000
    g = torch.Generator(device="cpu").manual_seed(4)
    mk = lambda *shape: torch.randn(*shape, generator=g, dtype=torch.float64).to(dt).to(dev).contiguous()  # noqa: E731
    pos, vel, frc = mk(n, 3) * 10, mk(n, 3), mk(n, 3) * 5
    mass = (1.0 + 15.0 * torch.rand(n, generator=g, dtype=torch.float64)).to(dt).to(dev)
    vc = (0.01 * torch.rand(n, generator=g, dtype=torch.float64)).to(dt).to(dev)
    ref = (pos + 0.1 * mk(n, 3)).contiguous()
    tstep, gamma, seed, step = 0.02, 0.3, 1234, 77
    st = torch.cuda.current_stream(dev).cuda_stream
    p1, v1 = pos.clone(), vel.clone()
    if langevin:
        L.check(lib.tmdhip_langevin_second_vv(code, 1, n, v1.data_ptr(), frc.data_ptr(), mass.data_ptr(), vc.data_ptr(),
                                              tstep, gamma, seed, step, st))
    else:
        L.check(lib.tmdhip_second_vv(code, 1, n, v1.data_ptr(), frc.data_ptr(), mass.data_ptr(), tstep, st))
    L.check(lib.tmdhip_first_vv(code, 1, n, p1.data_ptr(), v1.data_ptr(), frc.data_ptr(), mass.data_ptr(), tstep, st))
    p2, v2 = pos.clone(), vel.clone()
    disp2 = torch.zeros(1, dtype=torch.float32, device=dev)
    L.check(lib.tmdhip_dd_step(code, n, p2.data_ptr(), v2.data_ptr(), frc.data_ptr(), mass.data_ptr(),
                               vc.data_ptr() if langevin else 0, tstep, gamma, seed, step, 3, ref.data_ptr(),
                               disp2.data_ptr(), st))
    torch.cuda.synchronize()
    assert torch.equal(p1, p2) and torch.equal(v1, v2)
    want = ((p2 - ref) ** 2).sum(dim=1).max().item()
    assert abs(disp2.item() - want) <= 2e-6 * want and disp2.item() >= want * (1 - 1e-6)
    # the two phases one at a time give the same state as both together
    p3, v3 = pos.clone(), vel.clone()
    for phases in (1, 2):
        L.check(lib.tmdhip_dd_step(code, n, p3.data_ptr(), v3.data_ptr(), frc.data_ptr(), mass.data_ptr(),
                                   vc.data_ptr() if langevin else 0, tstep, gamma, seed, step, phases, 0, 0, st))
    torch.cuda.synchronize()
    assert torch.equal(p3, p2) and torch.equal(v3, v2)
    # halo pack
    idx = torch.randint(0, n, (777,), generator=g).to(torch.int32).to(dev)
    shift = (mk(777, 3) * 3).contiguous()
    out = torch.empty(777, 3, dtype=dt, device=dev)
    L.check(lib.tmdhip_halo_pack(code, 777, p2.data_ptr(), idx.data_ptr(), shift.data_ptr(), out.data_ptr(), st))
    torch.cuda.synchronize()
    assert torch.equal(out, p2.index_select(0, idx.long()) + shift)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [1, 2, 8])
def test_brick_loop_variants_are_bit_identical(world, monkeypatch):
    """`tmdhip_dd_run` makes a brick step (a) as separate launches (TMDHIP_DD_FUSED=0: kick/drift, pack, displacement test,
    rebuild chain, pair kernel), (b) as three launches (=1: dd_own_kernel, dd_halo_kernel, the pair kernel with the chain
    left out while nobody is near its limit) or (c) with the owned atoms' update made by the step blocks of the previous
    pair launch (=2, the default; fp32 bricks on the lean kernel).  Same arithmetic, same rebuild and migration decisions:
    positions, velocities and forces must agree bit for bit — fp32, Langevin, hot enough for migrations (made by
    tmdhip_dd_migrate) and device-side list rebuilds inside the window; in-process ranks, one host thread each."""
    from torchmd_amd.domain import DomainSet, LocalTransport
    from torchmd_amd.integrator import maxwell_boltzmann

    dev, dt = torch.device("cuda:0"), torch.float32
    mol, pos, box, par = _system(23, dt)  # (odd: brick faces pass through lattice planes, atoms change bricks at once)
    n = mol.numAtoms
    torch.manual_seed(3)
    vel = maxwell_boltzmann(par.masses, 600.0, 1)[0].numpy()
    A, B = par.get_AB()
    monkeypatch.setenv("TMDHIP_DEBUG_CHAIN_MIN_ENTRIES", "1")
    out = {}
    for mode in ("2", "1", "0"):
        monkeypatch.setenv("TMDHIP_DD_FUSED", mode)
        tr = LocalTransport(world, native_threads=True)
        ds = DomainSet(box, world, dev, dt, ["lj"], 9.0, A=A, B=B, skin=1.0, transport=tr)
        ds.scatter(pos, vel, par.charges.numpy(), par.mapped_atom_types.numpy(), par.masses.numpy().ravel())
        ds.compute_forces()
        ds.step(45, timestep_fs=2.0, gamma_ps=1.0, T=600.0, seed=11)
        ds.step(36, timestep_fs=2.0, gamma_ps=1.0, T=600.0, seed=11)
        stats = [d.forces_engine.stats(d.local_pos) for d in ds.domains.values()]
        out[mode] = ds.gather(n) + (ds.migrations, sum(s["steps_in_pair_launch"] for s in stats),
                                    sum(s["n_rebuilds"] for s in stats), sum(s["chains_skipped"] for s in stats))
        for d in ds.domains.values():
            d.forces_engine.close()
        tr.close()
    P, V, F, mig, fused_steps, rebuilds, skipped = out["2"]
    assert torch.isfinite(P).all() and mig >= 2
    assert fused_steps > 40 * world and skipped > 20 * world, (fused_steps, skipped)
    for mode in ("1", "0"):
        P1, V1, F1, mig1, fs1, rb1, sk1 = out[mode]
        assert fs1 == 0 and mig1 == mig and rb1 == rebuilds, (mode, mig1, rb1)
        assert torch.equal(P, P1) and torch.equal(V, V1) and torch.equal(F, F1), mode
    assert out["1"][6] == skipped and out["0"][6] == 0


@pytest.mark.timeout(600)
def test_library_migration_equals_the_torch_migration_with_mixed_types():
    """`tmdhip_dd_migrate` against the torch migration of domain.py on a mixture: three atom types of which two share
    their LJ parameters (the engine merges them into one class: the type -> class map travels to the library), different
    masses, charges and a reaction field, so that every field of a migrating atom's state row matters.  World 4
    (2 x 1 x 2), in-process ranks; the library's loop with the library's migration against the same loop with
    TMDHIP_DD_MIGRATE=python: same migration decisions, same owners, positions / velocities / forces to 1e-9 (the two plan
    the engine's cell grid over different bounds: another summation order), and both equal the single-domain run."""
    import os

    from torchmd_amd.builders import Topology
    from torchmd_amd.domain import DomainSet, LocalTransport
    from torchmd_amd.forcefields.ff_yaml import YamlForceField
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev, dt = torch.device("cuda:0"), torch.float64
    rng = np.random.default_rng(12)
    nside, a = 21, 3.6  # odd: the faces of the 2 x 1 x 2 bricks pass THROUGH lattice planes, atoms cross them at once
    g = np.arange(nside)
    sites = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float64)
    pos = sites * a + a / 2 + rng.uniform(-0.3, 0.3, size=sites.shape)
    n = len(pos)
    box = np.array([nside * a] * 3)
    kinds = np.array(["A1", "A2", "B"], dtype=object)[rng.integers(0, 3, size=n)]
    ff = {
        "atomtypes": ["A1", "A2", "B"],
        "lj": {"A1": {"sigma": 3.345, "epsilon": 0.238}, "A2": {"sigma": 3.345, "epsilon": 0.238}, "B": {"sigma": 3.0, "epsilon": 0.15}},
        "electrostatics": {"A1": {"charge": 0.1}, "A2": {"charge": -0.1}, "B": {"charge": 0.0}},
        "masses": {"A1": 39.95, "A2": 20.0, "B": 30.0},
    }
    charge = np.array([ff["electrostatics"][k]["charge"] for k in kinds], dtype=np.float32)
    masses = np.array([ff["masses"][k] for k in kinds], dtype=np.float32)
    mol = Topology(atomtype=kinds, charge=charge, masses=masses)
    terms = ["lj", "electrostatics"]
    par = Parameters(YamlForceField(mol, ff), mol, terms, precision=dt)
    torch.manual_seed(5)
    vel = maxwell_boltzmann(par.masses, 3000.0, 1)[0].numpy()
    A, B = par.get_AB()
    out = {}
    growths = 0
    try:
        for how in ("native", "python", "native-tight"):
            os.environ["TMDHIP_DD_MIGRATE"] = how.split("-")[0]
            # third pass: capacity buffers without headroom — every brick that gains an atom, a halo row or a send row makes
            # tmdhip_dd_migrate return 2 (need_*); the caller grows its arrays and the call resumes where it stopped
            os.environ.pop("TMDHIP_DD_TEST_TIGHT_CAPS", None)
            if how == "native-tight":
                os.environ["TMDHIP_DD_TEST_TIGHT_CAPS"] = "1"
            tr = LocalTransport(4, native_threads=True)
            ds = DomainSet(box, 4, dev, dt, terms, 9.0, A=A, B=B, skin=1.0, grid=(2, 1, 2), transport=tr, rfa=True)
            ds.scatter(pos, vel, par.charges.numpy(), par.mapped_atom_types.numpy(), par.masses.numpy().ravel())
            eng = next(iter(ds.domains.values())).forces_engine._engine(next(iter(ds.domains.values())).local_pos)
            assert eng.type_map is not None and eng.ntypes == 2  # A1 and A2 are one LJ class
            owners0 = torch.zeros(n, dtype=torch.long, device=dev)
            for r, d in ds.domains.items():
                owners0[d.ids] = r
            ds.compute_forces()
            ds.step(30, timestep_fs=2.0)
            owners = torch.zeros(n, dtype=torch.long, device=dev)
            for r, d in ds.domains.items():
                owners[d.ids] = r
            out[how] = ds.gather(n) + (ds.migrations, owners.cpu())
            assert int((owners != owners0).sum()) >= 20  # atoms really changed bricks
            if how == "native-tight":
                growths = sum(getattr(d, "capacity_growths", 0) for d in ds.domains.values())
            for d in ds.domains.values():
                d.forces_engine.close()
            tr.close()
    finally:
        os.environ.pop("TMDHIP_DD_MIGRATE", None)
        os.environ.pop("TMDHIP_DD_TEST_TIGHT_CAPS", None)
    assert growths >= 3, growths
    for a, b in zip(out["native"][:3], out["native-tight"][:3]):  # resumed calls give the same state, bit for bit
        assert torch.equal(a, b)
    assert out["native"][3] == out["native-tight"][3] and torch.equal(out["native"][4], out["native-tight"][4])
    P, V, F, mig, own = out["native"]
    P0, V0, F0, mig0, own0 = out["python"]
    assert mig >= 3 and mig == mig0 and torch.equal(own, own0)
    assert (P - P0).abs().max().item() < 1e-9 and (V - V0).abs().max().item() < 1e-9 and (F - F0).abs().max().item() < 1e-8
    s = System(n, 1, dt, dev)
    s.set_positions(pos[:, :, None])
    s.set_box(box)
    s.set_velocities(torch.tensor(vel)[None])
    f = Forces(par, terms=terms, cutoff=9.0, rfa=True)
    f.compute(s.pos, s.box, s.forces)
    Integrator(s, f, 2.0, dev).step(30)
    assert (P - s.pos[0]).abs().max().item() < 1e-7 and (V - s.vel[0]).abs().max().item() < 1e-7
    assert (F - s.forces[0]).abs().max().item() < 1e-6


@pytest.mark.timeout(600)
def test_halo_overrun_is_recovered_from_the_saved_state():
    """A `check_every` far too large for the temperature: the first displacement that is looked at lies beyond the halo's
    half skin already, so the steps since the last migration ran with halo atoms missing.  The loop must notice (the
    measured value, not the extrapolation), go back to the state saved at the entry of `step` / at the last migration,
    halve `check_every` and repeat — in the Python-driven loop, in the library's loop over the in-process communicator
    (world 4: `tmdhip_dd_run` returns TMDHIP_DD_OVERRUN on every rank at the same iteration) and over RCCL (one rank) —
    and still arrive at the single-domain trajectory.  With `recover` off the same run raises."""
    import socket

    import torch.distributed as dist

    from torchmd_amd.domain import DistTransport, DomainSet, HaloOverrun, LocalTransport
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.systems import System

    dev, dt = torch.device("cuda:0"), torch.float64
    mol, pos, box, par = _system(23, dt)
    n = mol.numAtoms
    torch.manual_seed(3)
    vel = maxwell_boltzmann(par.masses, 4000.0, 1)[0].numpy()
    A, B = par.get_AB()
    s = System(n, 1, dt, dev)
    s.set_positions(pos[:, :, None])
    s.set_box(box)
    s.set_velocities(torch.tensor(vel)[None])
    f = Forces(par, terms=["lj"], cutoff=9.0)
    f.compute(s.pos, s.box, s.forces)
    Integrator(s, f, 2.0, dev).step(40)

    def run(transport, world, recover=True):
        ds = DomainSet(box, world, dev, dt, ["lj"], 9.0, A=A, B=B, skin=1.0, transport=transport)
        ds.check_every, ds.recover = 32, recover
        ds.scatter(pos, vel, par.charges.numpy(), par.mapped_atom_types.numpy(), par.masses.numpy().ravel())
        ds.compute_forces()
        try:
            ds.step(25, timestep_fs=2.0)
            ds.step(15, timestep_fs=2.0)
            moved = ds.verify_halo()
            return ds.gather(n), ds.recoveries, ds.check_every, ds.migrations, moved
        finally:
            for d in ds.domains.values():
                d.forces_engine.close()
            transport.close()

    with pytest.raises(HaloOverrun, match="half skin"):
        run(LocalTransport(4), 4, recover=False)
    legs = {"python loop": lambda: run(LocalTransport(4), 4), "library loop, world 4": lambda: run(LocalTransport(4, native_threads=True), 4)}
    for name, leg in legs.items():
        (P, V, F), rec, every, mig, moved = leg()
        ep, ev, ef = (P - s.pos[0]).abs().max().item(), (V - s.vel[0]).abs().max().item(), (F - s.forces[0]).abs().max().item()
        print(f"{name}: {rec} recoveries, check_every 32 -> {every}, {mig} migrations, max|dx| {ep:.2e} max|dv| {ev:.2e} max|dF| {ef:.2e}")
        assert rec >= 1 and every < 32 and mig >= 1 and moved <= 0.5
        assert ep < 1e-7 and ev < 1e-7 and ef < 1e-6, name

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        (P, V, F), rec, every, mig, moved = run(DistTransport(), 1)
    finally:
        dist.destroy_process_group()
    ep, ev, ef = (P - s.pos[0]).abs().max().item(), (V - s.vel[0]).abs().max().item(), (F - s.forces[0]).abs().max().item()
    print(f"library loop over RCCL, one rank: {rec} recoveries, check_every 32 -> {every}, {mig} migrations, max|dx| {ep:.2e}")
    assert rec >= 1 and every < 32 and mig >= 1
    assert ep < 1e-7 and ev < 1e-7 and ef < 1e-6
