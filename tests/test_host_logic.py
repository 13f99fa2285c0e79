"""CPU tests (no GPU): C-ABI surface, host-side logic of the Forces/Integrator mirror, topology
readers / Parameters against the reference (when /root/reference is present), replica fan-out over
gloo with world_size 2."""

import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from _golden import GoldenParameters, load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_exports_every_declared_symbol():
    """The library loads on a CPU-only box and exports exactly what include/tmdhip.h declares."""
    from torchmd_amd import _lib

    header = open(os.path.join(ROOT, "include", "tmdhip.h")).read()
    declared = set(re.findall(r"\b(tmdhip_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.tmdhip_abi_version() == _lib.ABI_VERSION
    # struct layouts agree with the C compiler's view of the header
    src = ('#include "tmdhip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu %d\\n", '
           "sizeof(tmdhip_nonbonded_desc), sizeof(tmdhip_bonded_desc), sizeof(tmdhip_stats), sizeof(tmdhip_md_desc), "
           "sizeof(tmdhip_dd_desc), sizeof(tmdhip_dd_brick), TMDHIP_DD_OVERRUN);return 0;}")
    exe = os.path.join(ROOT, "tests", ".sizeof_probe")
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    try:
        sizes = [int(x) for x in subprocess.run([exe], capture_output=True, check=True).stdout.split()]
    finally:
        os.remove(exe)
    assert sizes == [C.sizeof(_lib.NonbondedDesc), C.sizeof(_lib.BondedDesc), C.sizeof(_lib.Stats),
                     C.sizeof(_lib.MdDesc), C.sizeof(_lib.DdDesc), C.sizeof(_lib.DdBrick), _lib.DD_OVERRUN]


def test_capi_argument_validation_without_gpu():
    from torchmd_amd import _lib

    lib = _lib.load()
    ctx = C.c_void_p()
    d = _lib.NonbondedDesc()
    d.struct_size = 4  # wrong size -> ABI error before any HIP call
    assert lib.tmdhip_create(C.byref(ctx), C.byref(d)) < 0
    assert "size mismatch" in _lib.last_error()
    assert lib.tmdhip_first_vv(7, 1, 1, None, None, None, None, 0.1, None) < 0
    assert "dtype" in _lib.last_error()
    assert lib.tmdhip_compute_nonbonded(None, 0, None, None, None, None, 0, None) < 0
    assert lib.tmdhip_compute_bonded(None, _lib.ALL_REPLICAS, None, None, None, None, 0, None) < 0
    assert lib.tmdhip_update_atoms(None, 10, None, None, 0) < 0
    assert "null" in _lib.last_error()
    assert lib.tmdhip_md_run(None, None, None) < 0
    assert lib.tmdhip_check(None, 0, None) < 0
    assert lib.tmdhip_timing_enable(None, 1) < 0
    assert lib.tmdhip_set_skin_weights(None, None) < 0
    # domain-decomposition entry points
    assert lib.tmdhip_dd_step(9, 1, None, None, None, None, None, 0.1, 0.0, 0, 0, 3, None, None, None) < 0
    assert "dtype" in _lib.last_error()
    assert lib.tmdhip_dd_step(_lib.F32, 4, None, None, None, None, None, 0.1, 0.0, 0, 0, 0, None, None, None) < 0  # no phase
    assert lib.tmdhip_dd_step(_lib.F32, 0, None, None, None, None, None, 0.1, 0.0, 0, 0, 3, None, None, None) == 0  # nothing to do
    assert lib.tmdhip_halo_pack(_lib.F32, -1, None, None, None, None, None) < 0
    assert lib.tmdhip_halo_pack(_lib.F32, 0, None, None, None, None, None) == 0
    assert lib.tmdhip_halo_pack(_lib.F32, 5, None, None, None, None, None) < 0 and "null" in _lib.last_error()
    assert lib.tmdhip_comm_exchange(None, _lib.F32, None, None, None, None, 3, None) < 0
    assert lib.tmdhip_comm_create(None, b"", None, 0, 1) < 0
    assert lib.tmdhip_comm_unique_id(b"", None) < 0
    assert lib.tmdhip_dd_run(None, None, None, None, None) < 0
    assert lib.tmdhip_dd_reset(None) < 0
    assert lib.tmdhip_dd_migrate(None, None, None, None) < 0 and "null" in _lib.last_error()
    lib.tmdhip_comm_destroy(None)  # a no-op


def test_no_cpu_fallback():
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator
    from torchmd_amd.systems import System

    g = load("water291")
    par = GoldenParameters(g, torch.float32)
    f = Forces(par, terms=["lj", "electrostatics"], cutoff=7.3)
    s = System(291, 1, torch.float32, "cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        f.compute(s.pos, s.box, s.forces)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Integrator(s, f, 1.0, "cpu").step(1)


def test_forces_constructor_contract():
    from torchmd_amd.forces import Forces, build_exclusion_csr

    g = load("ala2")
    par = GoldenParameters(g, torch.float64)
    with pytest.raises(RuntimeError):
        Forces(par, terms=None)
    with pytest.raises(ValueError):
        Forces(par, terms=["bonds", "hbonds"])
    with pytest.raises(RuntimeError):
        Forces(par, terms=["1-4", "lj"])
    f = Forces(par, terms=["Bonds", "LJ", "Electrostatics"], cutoff=9, rfa=True, switch_dist=7.5)
    assert f.energies == ["bonds", "lj", "electrostatics"] and f.require_distances and f.natoms == 688
    assert par.A is not None and par.A.shape == (10, 10)  # T = 10 for C2
    assert Forces.terms == Forces.bonded + Forces.nonbonded
    off, idx = f._excl_csr
    assert off[-1] == len(idx) == 2 * 764  # 764 exclusion entries, stored in both directions
    assert f.ava_idx.shape == (235564, 2)  # P_all of C2
    # CSR rows are sorted and symmetric
    pairs = {(i, int(j)) for i in range(688) for j in idx[off[i]:off[i + 1]]}
    assert all((j, i) in pairs for i, j in pairs)
    assert all(np.all(np.diff(idx[off[i]:off[i + 1]]) > 0) for i in range(688))
    off2, idx2 = build_exclusion_csr(5, [[0, 1], [1, 0], [3, 2], [0, 1]])
    assert off2.tolist() == [0, 1, 2, 3, 4, 4] and idx2.tolist() == [1, 0, 3, 2]
    assert Forces(par, terms=["bonds"]).ava_idx is None


def test_system_setters():
    from torchmd_amd.systems import System

    s = System(4, 3, torch.float64, "cpu")
    s.set_positions(np.arange(12, dtype=np.float32).reshape(4, 3))
    assert torch.equal(s.pos[0], s.pos[2]) and s.pos.dtype == torch.float64
    s.set_positions(np.stack([np.full((4, 3), k, dtype=np.float32) for k in range(3)], axis=2))
    assert s.pos[1, 0, 0] == 1 and s.pos[2, 3, 2] == 2
    s.set_box(np.array([10.0, 11.0, 12.0]))
    assert torch.equal(s.box[2], torch.diag(torch.tensor([10.0, 11.0, 12.0], dtype=torch.float64)))
    s.set_box(np.array([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0], [7.0, 8.0, 9.0]]))
    assert torch.equal(s.box[1].diagonal(), torch.tensor([2.0, 5.0, 8.0], dtype=torch.float64))
    with pytest.raises(RuntimeError):
        s.set_positions(np.zeros((4, 2)))
    with pytest.raises(RuntimeError):
        s.set_velocities(torch.zeros(1, 4, 3))
    with pytest.raises(RuntimeError):
        s.set_box(np.zeros(2))
    with pytest.raises(RuntimeError):
        s.set_masses(torch.zeros(5))
    s.set_masses(torch.tensor([1.0, 2.0, 3.0, 4.0]))
    assert s.masses.shape == (4, 1) and s.natoms == 4 and s.nreplicas == 3


def test_kinetic_energy_helpers():
    """Reference tests/test_integrator.py:7-140 (single/multi replica, batches incl. empty ones)."""
    from torchmd_amd.integrator import BOLTZMAN, kinetic_energy, kinetic_to_temp, maxwell_boltzmann

    m = torch.tensor([[1.0], [2.0], [3.0]])
    v = torch.tensor([[[1.0, 0, 0], [0, 2.0, 0], [0, 0, 3.0]], [[0.0, 0, 0]] * 3])
    ke = kinetic_energy(m, v)
    assert ke.shape == (2, 1) and torch.allclose(ke[:, 0], torch.tensor([0.5 + 4.0 + 13.5, 0.0]))
    kb = kinetic_energy(m, v, batch=torch.tensor([0, 2, 2]))
    assert kb.shape == (2, 3) and torch.allclose(kb[0], torch.tensor([0.5, 0.0, 17.5]))
    with pytest.raises(ValueError):
        kinetic_energy(m, v[0])
    assert np.isclose(kinetic_to_temp(1.0, 10), 2.0 / (30 * BOLTZMAN))
    torch.manual_seed(0)
    vel = maxwell_boltzmann(torch.full((20000, 1), 12.0), 300.0, replicas=2)
    assert vel.shape == (2, 20000, 3)
    T = kinetic_to_temp(kinetic_energy(torch.full((20000, 1), 12.0), vel)[:, 0].numpy(), 20000)
    assert np.all(np.abs(T - 300) < 6)


def test_builders():
    from torchmd_amd.builders import lj_box, tip3p_box, water_forcefield
    from torchmd_amd.parameters import Parameters

    mol, pos, box = tip3p_box(4, seed=0)
    assert mol.numAtoms == 192 and pos.shape == (192, 3) and np.allclose(box, 4 * (1 / 0.0334) ** (1 / 3))
    d = np.linalg.norm(pos[1::3] - pos[0::3], axis=1)
    assert np.allclose(d, 0.9572, atol=1e-10)
    hoh = np.degrees(np.arccos(np.sum((pos[1::3] - pos[0::3]) * (pos[2::3] - pos[0::3]), axis=1) / 0.9572**2))
    assert np.allclose(hoh, 104.52, atol=1e-8)
    par = Parameters(water_forcefield(mol), mol, ["lj", "electrostatics", "bonds", "angles"])
    assert len(par.bond_params["idx"]) == 192 and len(par.angle_params["idx"]) == 64
    assert len(par.get_exclusions()) == 192 + 64
    assert abs(float(par.charges.sum())) < 1e-4
    mol2, pos2, box2 = tip3p_box(4, seed=0)
    assert np.array_equal(pos, pos2)  # deterministic
    mol3, pos3, box3 = lj_box(5)
    assert mol3.numAtoms == 125 and np.allclose(box3, 5 * (1 / 0.0213) ** (1 / 3))


def test_replica_slices():
    from torchmd_amd.replicas import replica_slice

    assert [list(replica_slice(8, r, 8)) for r in range(8)] == [[r] for r in range(8)]
    got = [list(replica_slice(10, r, 4)) for r in range(4)]
    assert got == [[0, 1, 2], [3, 4, 5], [6, 7], [8, 9]]
    with pytest.raises(ValueError):
        replica_slice(2, 0, 4)


_GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from torchmd_amd.replicas import ReplicaFanout
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
fan = ReplicaFanout(total_replicas=5)
assert fan.rank == rank and fan.world == 2 and list(fan.local) == ([0, 1, 2] if rank == 0 else [3, 4])
fan.check_same_topology(np.arange(10), np.ones(3))
try:
    fan.check_same_topology(np.arange(10) + rank)
    raise SystemExit("mismatch not detected")
except RuntimeError:
    pass
loc = list(fan.local)
obs = fan.gather_observables([10.0 + r for r in loc], [-100.0 - r for r in loc], [300.0 + r for r in loc])
assert obs.shape == (5, 3)
assert np.allclose(obs[:, 0], 10 + np.arange(5)) and np.allclose(obs[:, 1], -100 - np.arange(5)) and np.allclose(obs[:, 2], 300 + np.arange(5))
assert fan.max_over_ranks(1.0 + rank) == 2.0
fan.barrier()
dist.destroy_process_group()
print("ok", rank)
"""


def test_replica_fanout_gloo_world2(tmp_path):
    """N>1 path on CPU: two processes over gloo (the GPU run uses the same code over nccl = RCCL)."""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    procs = [
        subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT)
        for r in range(2)
    ]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


# ------------------------------------------------------------------------------------------------
# readers + Parameters vs the reference itself (build container only)
# ------------------------------------------------------------------------------------------------
def _ref_modules():
    sys.path.insert(0, "/root/reference")
    from torchmd.forcefields.ff_yaml import YamlForcefield
    from torchmd.parameters import Parameters as RefParameters

    return YamlForcefield, RefParameters


def _same_table(a, b):
    if a is None or b is None:
        return a is None and b is None or (not torch.is_tensor((a or b).get("idx")))
    ok = torch.equal(a["idx"], b["idx"])
    pa, pb = a["params"][a["map"][:, 1]], b["params"][b["map"][:, 1]]
    return ok and torch.equal(a["map"][:, 0], b["map"][:, 0]) and torch.equal(pa, pb)


@pytest.mark.needs_reference
@pytest.mark.parametrize("prec", [torch.float32, torch.float64])
def test_parameters_match_reference_water(prec):
    from torchmd_amd import io as tio
    from torchmd_amd.forcefields import YamlForceField
    from torchmd_amd.parameters import Parameters

    RefYaml, RefParameters = _ref_modules()
    d = "/root/reference/tests/water"
    mol = tio.read_psf(os.path.join(d, "structure.psf"))
    terms = ["lj", "bonds", "angles", "electrostatics"]
    mine = Parameters(YamlForceField(mol, os.path.join(d, "water_forcefield.yaml")), mol, terms, precision=prec)
    ref = RefParameters(RefYaml(mol, os.path.join(d, "water_forcefield.yaml")), mol, terms, precision=prec)
    assert torch.equal(mine.charges, ref.charges) and torch.equal(mine.masses, ref.masses)
    assert torch.equal(mine.mapped_atom_types, ref.mapped_atom_types)
    assert torch.equal(mine.nonbonded_params["params"], ref.nonbonded_params["params"])
    assert _same_table(mine.bond_params, ref.bond_params) and _same_table(mine.angle_params, ref.angle_params)
    A1, B1 = mine.get_AB()
    A2, B2 = ref.get_AB()
    assert torch.equal(A1, A2) and torch.equal(B1, B2)
    assert sorted(map(tuple, mine.get_exclusions())) == sorted(map(tuple, ref.get_exclusions()))


@pytest.mark.needs_reference
def test_parameters_match_reference_ala2():
    from torchmd_amd import io as tio
    from torchmd_amd.forcefields import ForceField, PrmtopForceField
    from torchmd_amd.parameters import Parameters

    _, RefParameters = _ref_modules()
    d = "/root/reference/tests/data/prod_alanine_dipeptide_amber"
    mol, top = tio.read_prmtop(os.path.join(d, "structure.prmtop"))
    ff = PrmtopForceField(mol, top)
    assert isinstance(ForceField.create(mol, os.path.join(d, "structure.prmtop")), PrmtopForceField)
    terms = ["bonds", "angles", "dihedrals", "impropers", "1-4", "electrostatics", "lj"]
    mine = Parameters(ff, mol, terms, precision=torch.float64)
    ref = RefParameters(ff, mol, terms, precision=torch.float64)
    for name in ("bond", "angle", "dihedral", "improper", "nonbonded_14"):
        assert _same_table(getattr(mine, name + "_params"), getattr(ref, name + "_params")), name
    assert torch.equal(mine.nonbonded_params["params"], ref.nonbonded_params["params"])
    g = load("ala2")
    assert np.array_equal(mine.charges.numpy(), g["par_charges"])
    assert np.array_equal(tio.read_namd_coor(os.path.join(d, "input.coor")), g["pos"])
    assert np.array_equal(tio.read_xsc(os.path.join(d, "input.xsc")), g["box"])


@pytest.mark.needs_reference
@pytest.mark.parametrize("folder", ["thrombin-ligand-amber", "benzamidine-amber", "ligand-amber"])
def test_parameters_match_reference_prmtop_fixtures(folder):
    """Own prmtop reader + `PrmtopForceField` + `Parameters` against the reference's `Parameters` fed with
    the same objects, on the reference's AMBER fixtures (protein + ligand complex, two ligands: impropers, many atom types)."""
    from torchmd_amd import io as tio
    from torchmd_amd.forcefields import PrmtopForceField
    from torchmd_amd.parameters import Parameters

    _, RefParameters = _ref_modules()
    path = os.path.join("/root/reference/tests/data", folder, "structure.prmtop")
    if not os.path.exists(path):
        pytest.skip(f"no {path}")
    mol, top = tio.read_prmtop(path)
    ff = PrmtopForceField(mol, top)
    terms = ["bonds", "angles", "dihedrals", "impropers", "1-4", "electrostatics", "lj"]
    mine = Parameters(ff, mol, terms, precision=torch.float64)
    ref = RefParameters(ff, mol, terms, precision=torch.float64)
    for name in ("bond", "angle", "dihedral", "improper", "nonbonded_14"):
        assert _same_table(getattr(mine, name + "_params"), getattr(ref, name + "_params")), name
    assert torch.equal(mine.nonbonded_params["params"], ref.nonbonded_params["params"])
    assert torch.equal(mine.mapped_atom_types, ref.mapped_atom_types)
    assert sorted(map(tuple, mine.get_exclusions())) == sorted(map(tuple, ref.get_exclusions()))


def _free_port():
    """A TCP port nobody listens on right now (fixed ports collide with sockets of an earlier run in TIME_WAIT)."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


_DD_WORKER = r"""
import os, sys, itertools
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from torchmd_amd.domain import BrickGrid, HaloPlan, DistTransport, factor_grid
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
assert factor_grid(8) == (2, 2, 2) and sorted(factor_grid(4)) == [1, 2, 2] and sorted(factor_grid(2)) == [1, 1, 2]
box = np.array([30.0, 24.0, 27.0])
halo = 6.5
rng = np.random.default_rng(11)
n = 3000
pos = torch.tensor(rng.uniform(-40, 70, size=(n, 3)))          # any periodic image
grid = BrickGrid(box, world)
owner, w = grid.owner(pos)
mine = torch.nonzero(owner == rank).flatten()
plan = HaloPlan(grid, rank, w[mine], halo)
tr = DistTransport().bind(torch.device("cpu"))
payload = torch.cat([plan.pack_positions(w[mine]), plan.pack(mine.double()[:, None])], dim=1)
recv_counts = tr.exchange_counts(plan.send_counts)
got = tr.all_to_all(payload, plan.send_counts, recv_counts)
# brute force: every periodic image of every atom inside my brick grown by the halo, minus the brick itself
lo, hi = grid.bounds(rank)
exp = []
for s in itertools.product((-1, 0, 1), repeat=3):
    img = w + torch.tensor(s, dtype=torch.float64) * torch.tensor(box)
    inside_ext = ((img >= lo - halo) & (img < hi + halo)).all(dim=1)
    inside = ((img >= lo) & (img < hi)).all(dim=1)
    sel = torch.nonzero(inside_ext & ~inside).flatten()
    exp.append(torch.cat([img[sel], sel.double()[:, None]], dim=1))
exp = torch.cat(exp)
key = lambda t: t[np.lexsort((t[:, 2].numpy(), t[:, 1].numpy(), t[:, 0].numpy(), t[:, 3].numpy()))]
assert got.shape == exp.shape, (got.shape, exp.shape)
assert torch.allclose(key(got), key(exp), atol=1e-12)
# the layout tmdhip_comm_exchange uses (csrc/domain.hip exchange_rows: one send and one recv per peer, offsets =
# running sums of the per-peer counts in rank order), replayed with point-to-point gloo ops: same rows as the
# all-to-all (a message to oneself is a local copy here; RCCL takes it inside the group)
p2p = torch.zeros_like(got)
so = ro = 0
ops = []
for p in range(world):
    ns, nr = plan.send_counts[p], recv_counts[p]
    if p == rank:
        p2p[ro: ro + nr] = payload[so: so + ns]
    else:
        if ns:
            ops.append(dist.P2POp(dist.isend, payload[so: so + ns].contiguous(), p))
        if nr:
            ops.append(dist.P2POp(dist.irecv, p2p[ro: ro + nr], p))
    so += ns
    ro += nr
for w_ in (dist.batch_isend_irecv(ops) if ops else []):
    w_.wait()
assert torch.equal(p2p, got)
assert tr.any_true(torch.tensor(rank == world - 1)) and not tr.any_true(torch.tensor(False))
assert float(tr.sum(torch.tensor([1.0 + rank]))) == world * (world + 1) / 2
dist.barrier()
dist.destroy_process_group()
print("ok", rank, got.shape[0], flush=True)
os._exit(0)  # gloo's point-to-point worker threads occasionally abort the interpreter's normal teardown
"""


@pytest.mark.parametrize("world", [2, 4])
def test_halo_exchange_gloo(tmp_path, world):
    """Exchange layer of the domain decomposition over gloo: the halo every rank receives equals the
    brute-force set of periodic images within `halo` of its brick (26 directed messages, incl. messages
    to itself across the periodic boundary when a dimension has one brick)."""
    script = tmp_path / "dd_worker.py"
    script.write_text(_DD_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(world))
    procs = [
        subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT)
        for r in range(world)
    ]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


@pytest.mark.parametrize("world", [2, 4])
def test_bench_c5_dry_gloo(world):
    """`bench.py --config c5 --gpus N --dry --backend gloo`: the N-brick bench path end to end without a GPU — DomainSet
    planning, DistTransport count / row exchanges, the migration trigger and migrations over gloo, the force engine
    stubbed (DryDomain).  The run checks itself (halo == brute-force image set before and after, every atom id exactly
    once, positions == x0 + v t although atoms changed bricks) and exits non-zero otherwise."""
    env = dict(os.environ, OMP_NUM_THREADS="2")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c5", "--gpus", str(world), "--backend", "gloo",
                          "--dry", "--nside", "24", "--steps", "40", "--warmup", "3"], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["dry"] is True and out["n_gpus"] == world and out["backend"] == "gloo"
    assert out["halo_equals_brute_force"] == [True, True] and out["atoms_accounted_for"] is True
    assert out["domains"]["migrations_in_timed_region"] >= 2 and out["max_abs_dx_vs_ballistic"] < 1e-9
    assert sorted(out["domains"]["grid"]) == ([1, 1, 2] if world == 2 else [1, 2, 2])


def test_merge_lj_types_preserves_every_pair_parameter():
    """Types with identical LJ rows are merged before the table goes to the device: every (atom, atom)
    pair must still see the same A and B."""
    import numpy as np
    import torch

    from _golden import GoldenParameters, load
    from torchmd_amd.forces import merge_lj_types

    for name, expect in (("ala2", 9), ("thrombin", 16), ("water291", 2)):
        par = GoldenParameters(load(name), torch.float64)
        A, B = (t.numpy() for t in par.get_AB())
        types = par.mapped_atom_types.numpy().astype(np.int64)
        A2, B2, t2, tmap = merge_lj_types(A, B, types)
        assert A2.shape[0] == expect
        sample = np.random.default_rng(0).integers(0, len(types), size=(2000, 2))
        i, j = sample[:, 0], sample[:, 1]
        assert np.array_equal(A[types[i], types[j]], A2[t2[i], t2[j]])
        assert np.array_equal(B[types[i], types[j]], B2[t2[i], t2[j]])
        if tmap is None:
            assert A2 is A and t2 is types
        else:
            assert np.array_equal(tmap[types], t2)


def test_external_plugin_config(tmp_path):
    """`torchmd_amd.external.External` (the reference's plugin hook, run.py:185-209): configuration file
    -> Parameters without a GPU, option validation, and the no-CPU-path contract."""
    import yaml

    from torchmd_amd.builders import TIP3P_FF
    from torchmd_amd.external import External

    psf = tmp_path / "w.psf"
    psf.write_text(
        "PSF\n\n       1 !NTITLE\n REMARKS one water\n\n       3 !NATOM\n"
        "       1 W    1    TIP3 OH2  OT    -0.834000       15.9994           0\n"
        "       2 W    1    TIP3 H1   HT     0.417000        1.0080           0\n"
        "       3 W    1    TIP3 H2   HT     0.417000        1.0080           0\n\n"
        "       3 !NBOND: bonds\n       1       2       1       3       2       3\n\n"
        "       1 !NTHETA: angles\n       2       1       3\n\n"
    )
    (tmp_path / "ff.yaml").write_text(yaml.safe_dump(TIP3P_FF))
    conf = tmp_path / "nb.yaml"
    conf.write_text(yaml.safe_dump({"topology": "w.psf", "forcefield": "ff.yaml", "terms": ["lj", "electrostatics"],
                                    "cutoff": 9.0, "rfa": True}))
    opts, par = External._from_file(str(conf))
    assert opts == {"terms": ["lj", "electrostatics"], "cutoff": 9.0, "rfa": True}
    assert par.natoms == 3 and abs(float(par.charges.sum())) < 1e-6
    assert sorted(map(tuple, par.get_exclusions(("bonds", "angles", "1-4")))) == [(0, 1), (0, 2), (0, 2), (1, 2)] or \
        len(par.get_exclusions(("bonds", "angles", "1-4"))) >= 3
    with pytest.raises(RuntimeError, match="ROCm device only"):
        External(str(conf), [0], device="cpu")
    (tmp_path / "bad.yaml").write_text(yaml.safe_dump({"forcefield": "ff.yaml"}))
    with pytest.raises(ValueError, match="topology"):
        External._from_file(str(tmp_path / "bad.yaml"))


def test_compat_install_registers_reference_module_names():
    """`torchmd_amd.compat.install()`: `from torchmd.forces import Forces` resolves to this package."""
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import torchmd_amd.compat as c\n"
        "names = c.install()\n"
        "from torchmd.forces import Forces\n"
        "from torchmd.integrator import Integrator, maxwell_boltzmann\n"
        "from torchmd.systems import System\n"
        "import torchmd_amd.forces, torchmd_amd.integrator\n"
        "assert Forces is torchmd_amd.forces.Forces and Integrator is torchmd_amd.integrator.Integrator\n"
        "assert 'torchmd.forces' in names and 'torchmd.parameters' in names\n"
        "print('ok')\n"
    ) % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr


def test_driver_external_hook(tmp_path):
    """`torchmd_amd.run.load_external` follows the reference's plugin protocol (run.py:185-209)."""
    import sys

    from torchmd_amd.run import load_external

    (tmp_path / "myplugin.py").write_text(
        "class External:\n"
        "    def __init__(self, file, embeddings, device=None, **kw):\n"
        "        self.file, self.embeddings, self.device, self.kw = file, embeddings, device, kw\n"
        "    def calculate(self, pos, box):\n"
        "        return pos.sum(dim=(1, 2)) * 0, pos * 0\n"
    )
    sys.path.insert(0, str(tmp_path))
    try:
        ext = load_external({"module": "myplugin", "file": "model.ckpt", "embeddings": [1, 8, 1], "scale": 2.0}, 3, "cpu")
    finally:
        sys.path.remove(str(tmp_path))
    assert ext.file == "model.ckpt" and ext.kw == {"scale": 2.0} and ext.device == "cpu"
    assert tuple(ext.embeddings.shape) == (3, 3)
    assert load_external(None, 1, "cpu") is None
    with pytest.raises(ValueError, match="module"):
        load_external({"file": "x"}, 1, "cpu")


def test_minimizers_on_a_mock_potential():
    """`torchmd_amd.minimizers` (mirror of the reference's `minimizers.py`) only need the duck type
    `compute(pos, box, forces, ...)`: an anisotropic harmonic well on CPU tensors has its minimum found by
    all three (L-BFGS-B, torch LBFGS, conjugate gradient)."""
    import types

    from torchmd_amd.minimizers import minimize_bfgs, minimize_cg, minimize_pytorch_bfgs

    torch.manual_seed(0)
    n = 7
    centre = torch.randn(1, n, 3, dtype=torch.float64)
    k = torch.tensor([1.0, 4.0, 0.5], dtype=torch.float64)

    class Well:
        def compute(self, pos, box, forces, returnDetails=False, explicit_forces=True, toNumpy=True,
                    calculateForces=True):
            d = pos - centre
            e = (0.5 * k * d * d).sum(dim=(1, 2))
            if forces is not None:
                with torch.no_grad():
                    forces[:] = -(k * d).detach()
            return [float(v) for v in e] if toNumpy else [v for v in e]

    def system():
        return types.SimpleNamespace(pos=centre + torch.randn(1, n, 3, dtype=torch.float64),
                                     box=torch.zeros(1, 3, 3, dtype=torch.float64),
                                     forces=torch.zeros(1, n, 3, dtype=torch.float64), nreplicas=1, natoms=n)

    s = system()
    res = minimize_bfgs(s, Well(), fmax=1e-6, steps=200)
    assert res.fun < 1e-10 and (s.pos - centre).abs().max() < 1e-5
    s = system()
    energies = minimize_pytorch_bfgs(s, Well(), steps=5, max_iter=20)
    assert energies.shape[0] == 1 and energies[0, -1] < 1e-8 and (s.pos - centre).abs().max() < 1e-3
    s = system()
    last = minimize_cg(s, Well(), steps=60, threshold=1e-2)  # (the line search resolves 1 % of its interval)
    assert last < 59 and (s.pos - centre).abs().max() < 2e-2 and s.forces.abs().max() < 1e-2
    assert minimize_bfgs(s, Well(), steps=0) is None and minimize_pytorch_bfgs(s, Well(), steps=0) is None
    two = types.SimpleNamespace(pos=torch.zeros(2, n, 3), box=None, forces=None)
    with pytest.raises(RuntimeError, match="replicas"):
        minimize_bfgs(two, Well())


def test_utils_mirror(tmp_path):
    """`torchmd_amd.utils` (reference `utils.py`): monitor CSV, option files, xyz export."""
    import argparse
    import csv

    import numpy as np

    from torchmd_amd.utils import LoadFromFile, LogWriter, save_argparse, xyz_writer

    log = LogWriter(str(tmp_path), ("iter", "epot"), header={"run": 1}, name="m.csv")
    log.write_row({"iter": 1, "epot": -2.5})
    log.f.close()
    lines = (tmp_path / "m.csv").read_text().splitlines()
    assert lines[0].startswith("# {") and lines[1] == "iter,epot,t"
    row = next(csv.DictReader(lines[1:]))
    assert row["iter"] == "1" and float(row["epot"]) == -2.5 and float(row["t"]) >= 0
    ap = argparse.ArgumentParser()
    ap.add_argument("--conf", type=open, action=LoadFromFile)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--name", default="x")
    (tmp_path / "c.yaml").write_text("steps: 9\nname: y\n")
    ns = ap.parse_args(["--conf", str(tmp_path / "c.yaml")])
    assert ns.steps == 9 and ns.name == "y"
    (tmp_path / "c.txt").write_text("steps=11\nname=z\n")
    ns = ap.parse_args(["--conf", str(tmp_path / "c.txt")])
    assert ns.steps == 11 and ns.name == "z"
    save_argparse(ns, str(tmp_path / "out.yaml"), exclude="conf")
    assert "steps: 11" in (tmp_path / "out.yaml").read_text()
    traj = np.arange(2 * 3 * 2, dtype=float).reshape(2, 3, 2)
    np.save(tmp_path / "t.npy", traj)
    xyz_writer(str(tmp_path / "t.npy"), str(tmp_path / "t.xyz"), ["O", "H"])
    out = (tmp_path / "t.xyz").read_text().splitlines()
    assert out[0] == "2" and out[2].startswith("O 0.0 2.0 4.0") and len(out) == 8


@pytest.mark.parametrize("world", [2, 8])
def test_bench_self_launch_gloo_world2(world):
    """`python bench.py --gpus N` outside torchrun starts its own N ranks (torch.distributed.run, 127.0.0.1)
    — the path the driver's multi-GPU scaling run takes (N = 1, 2, 4, 8 on one node); here with the gloo backend and no
    GPU work (--dry), at world 2 and at the full 8 ranks of a node.  Rank 0 prints ONE JSON line with n_gpus = N and every
    rank's observables gathered."""
    import json
    import subprocess

    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--dry",
                          "--steps", "5", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["dry"] is True and out["ranks_seen"] == list(range(world)) and out["steps"] == 5
    if world != 2:
        return
    # --gpus 1 needs no launcher (and no process group)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", "--steps", "3"], capture_output=True,
                         text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    assert json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])["n_gpus"] == 1


def test_skin_weight_rules():
    """Per-atom skin weights (host logic): by mass with exponent 0.45 and a floor of 0.2, none for single-mass
    systems, explicit arrays validated."""
    import numpy as np
    import torch

    from _golden import GoldenParameters, load
    from torchmd_amd.forces import Forces

    par = GoldenParameters(load("water291"), torch.float32)
    f = Forces(par, terms=["lj", "electrostatics"], cutoff=7.3)
    w = f._skin_weight_array()
    m = par.masses.numpy().ravel()
    assert w.shape == (291,) and w.max() == 1.0 and np.all(w[m < 2] == 1.0)
    assert np.allclose(w[m > 2], (m.min() / m[m > 2]) ** 0.45) and 0.27 < w[m > 2][0] < 0.30
    assert Forces(par, terms=["lj"], cutoff=7.3, skin_weights=None)._skin_weight_array() is None
    ones = Forces(par, terms=["lj"], cutoff=7.3, skin_weights=np.full(291, 0.5))._skin_weight_array()
    assert np.all(ones == 0.5)
    for bad in (np.full(290, 0.5), np.full(291, 1.5), np.zeros(291), "speed"):
        with pytest.raises(ValueError):
            Forces(par, terms=["lj"], cutoff=7.3, skin_weights=bad)._skin_weight_array()
    # one mass for every atom: nothing to weigh
    par1 = GoldenParameters(load("water291"), torch.float32)
    par1.masses = torch.full_like(par1.masses, 12.0)
    assert Forces(par1, terms=["lj"], cutoff=7.3)._skin_weight_array() is None


def test_bench_cpu_baseline_c5_sample_and_pmc_provenance():
    """`bench.py`'s CPU-baseline leg of config C5 runs the oracle on a bounded sample and scales by the atom ratio
    (stated in `sample`); the counter figures the bench line quotes come from a committed PMC file with its commit."""
    sys.path.insert(0, ROOT)
    import bench

    r = bench.cpu_baseline_c5(1_000_000, nside_sample=8, budget_s=1.0)
    assert r["kind"] == "port" and r["unit"] == "ns/day" and r["cores"] >= 1
    assert 0 < r["value"] < 10 and "512-atom" in r["sample"] and "atom ratio" in r["sample"]
    traffic, src, commit, valu = bench.pmc_traffic()
    assert traffic and traffic > 5e7 and src.startswith("profiles/") and commit and valu and valu > 1e6
    assert bench.TIMING_PASS_STEPS >= 64  # the dominant kernel is timed in a pass of its own behind the timed region


def test_launch_timeline_tool_on_a_synthetic_trace(tmp_path):
    """tools/launch_timeline.py groups the fused pair + step launches of a kernel trace by the number of launches since the
    last real list build (early-exit builds of a few microseconds do not count)."""
    import sqlite3

    db = tmp_path / "trace_results.db"
    con = sqlite3.connect(db)
    con.execute("create table kernels (name text, start integer, end integer)")
    fused = "void tmd::list_pair_fast_f32_kernel<8, true, true, false, false, 2>(int, HIP_vector_type<float, 4u> const*)"
    energy = "void tmd::list_pair_fast_f32_kernel<8, true, true, true, false, 0>(int, HIP_vector_type<float, 4u> const*)"
    build = "void tmd::build_list_kernel<float, false, true>(int, tmd::Vec<float>::T4 const*)"
    t, rows = 0, []

    def add(name, dur):
        nonlocal t
        rows.append((name, t, t + dur))
        t += dur + 100

    for _ in range(3):
        add(build, 160_000)
        add(fused, 50_000)
        for _ in range(4):
            add(build, 3_000)  # early exit
            add(fused, 44_000)
    add(energy, 52_000)
    con.executemany("insert into kernels values (?, ?, ?)", rows)
    con.commit()
    con.close()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "launch_timeline.py"), str(db)], capture_output=True,
                         text=True, check=True).stdout
    lines = {int(ln.split()[0].rstrip("+")): ln.split() for ln in out.splitlines() if ln.startswith("   ") or ln.startswith("  1")}
    assert float(lines[0][1]) == 50.0 and int(lines[0][4]) == 3
    assert float(lines[1][1]) == 44.0 and int(lines[4][4]) == 3 and 5 not in lines


def test_bench_caps_worker_threads_at_the_cpu_quota(tmp_path):
    """bench.py: worker threads of torch / numpy are capped at what the cgroup lets run at once (the GPU boxes: 256 hardware
    threads under a quota of 16 CPUs — 128 threads had every thread of the process throttled, the one pacing the GPU included).
    A fresh interpreter imports bench with no thread variables set: the environment it leaves and torch's thread count are within
    `usable_cpus()`, which is within the affinity mask; a caller's OMP_NUM_THREADS (torch.distributed.run sets 1) is respected."""
    code = ("import os, json, sys; sys.path.insert(0, %r); import bench, torch; "
            "print(json.dumps({'usable': bench.USABLE_CPUS, 'omp': os.environ['OMP_NUM_THREADS'], 'torch': torch.get_num_threads(), "
            "'aff': len(os.sched_getaffinity(0))}))" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    out = json.loads(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    assert 1 <= out["usable"] <= out["aff"]
    assert 1 <= int(out["omp"]) <= max(1, min(8, out["usable"] // 2)) and out["torch"] <= max(1, out["usable"])
    env["OMP_NUM_THREADS"] = "1"
    out = json.loads(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    assert out["omp"] == "1" and out["torch"] == 1
