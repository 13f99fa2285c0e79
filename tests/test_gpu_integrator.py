"""GPU tests of the integrator kernels: closed-form velocity-Verlet known answers (the reference's own
KATs, tests/test_integrator.py:310-511, re-stated for device tensors), the duck-typed `forces` contract,
Langevin statistics, the kinetic-energy reduction and a short NVE trajectory against the reference."""

import numpy as np
import pytest
import torch

from _golden import GoldenParameters, PREC, box_tensor, load, pos_tensor

pytestmark = pytest.mark.gpu

TIMEFACTOR = 48.88821
BOLTZMAN = 0.001987191


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


class ConstantForces:
    """Minimal duck type the Integrator needs (tests/test_integrator.py:155-158)."""

    def __init__(self, value, masses):
        self.value = value
        self.par = type("P", (), {"masses": masses})()
        self.calls = 0

    def compute(self, pos, box, forces):
        self.calls += 1
        forces[:] = self.value
        return [0.0] * pos.shape[0]


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("R", [1, 2])
def test_constant_force_closed_form(prec, R):
    from torchmd_amd.integrator import Integrator
    from torchmd_amd.systems import System

    dev, dt = _dev(), PREC[prec]
    n = 5
    s = System(n, R, dt, dev)
    m = torch.tensor([1.0, 2.0, 12.0, 16.0, 1.008], dtype=dt)
    s.set_masses(m)
    rng = np.random.default_rng(0)
    x0 = rng.normal(size=(R, n, 3))
    v0 = rng.normal(size=(R, n, 3)) * 0.1
    F = rng.normal(size=(R, n, 3))
    s.pos[:] = torch.tensor(x0, dtype=dt)
    s.set_velocities(torch.tensor(v0, dtype=dt))
    s.set_forces(F)
    ff = ConstantForces(torch.tensor(F, dtype=dt, device=dev), m)
    integ = Integrator(s, ff, timestep=1.0, device=dev, gamma=None, T=None)
    nsteps = 7
    ekin, pot, T = integ.step(niter=nsteps)
    assert ff.calls == nsteps
    t = nsteps * 1.0 / TIMEFACTOR
    a = F / m.numpy()[None, :, None]
    rtol = 1e-12 if prec == "f64" else 1e-5
    assert np.allclose(s.pos.cpu().numpy(), x0 + v0 * t + 0.5 * a * t * t, rtol=rtol, atol=rtol)
    assert np.allclose(s.vel.cpu().numpy(), v0 + a * t, rtol=rtol, atol=rtol)
    vel = v0 + a * t
    ek = 0.5 * (m.numpy()[None, :, None] * vel**2).sum(axis=(1, 2))
    assert np.allclose(ekin, ek, rtol=1e-5)
    assert np.allclose(T, 2.0 / (3.0 * n * BOLTZMAN) * ek, rtol=1e-5)
    assert pot == [0.0] * R and ekin.shape == (R,)


def test_masses_from_forces_par_and_batch():
    from torchmd_amd.integrator import Integrator, kinetic_energy
    from torchmd_amd.systems import System

    dev = _dev()
    s = System(4, 1, torch.float64, dev)  # (the reference's batch mode only works for one replica)
    m = torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=torch.float64)
    ff = ConstantForces(torch.zeros(1, 4, 3, dtype=torch.float64, device=dev), m)
    s.set_velocities(torch.ones(1, 4, 3, dtype=torch.float64))
    batch = torch.tensor([0, 0, 1, 1], device=dev)
    integ = Integrator(s, ff, 1.0, dev, batch=batch)
    assert integ.masses.shape == (4, 1) and integ.masses.device.type == "cuda"
    ekin, _, T = integ.step(1)
    assert np.allclose(ekin, [4.5, 10.5]) and T.shape == (2,)
    assert list(integ.natoms) == [2, 2]
    ke = kinetic_energy(integ.masses, s.vel)
    assert ke.shape == (1, 1) and torch.allclose(ke.cpu(), torch.tensor([[15.0]], dtype=torch.float64))
    with pytest.raises(ValueError):
        kinetic_energy(integ.masses, s.vel[0])


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_normal_stream_and_langevin(prec):
    """Philox/Box-Muller stream: moments, no repeats across steps; free-particle Langevin reaches T."""
    import ctypes as C

    from torchmd_amd import _lib as L
    from torchmd_amd.integrator import Integrator
    from torchmd_amd.systems import System

    dev, dt = _dev(), PREC[prec]
    lib = L.load()
    n = 3_000_000
    a = torch.empty(n, dtype=dt, device=dev)
    b = torch.empty(n, dtype=dt, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.tmdhip_normal_fill(L.dtype_code(dt), n, a.data_ptr(), 1234, 0, st))
    L.check(lib.tmdhip_normal_fill(L.dtype_code(dt), n, b.data_ptr(), 1234, 1, st))
    x = a.double().cpu().numpy()
    assert abs(x.mean()) < 3e-3 and abs(x.std() - 1) < 3e-3
    assert abs((x**3).mean()) < 1e-2 and abs((x**4).mean() - 3) < 3e-2
    assert abs(np.corrcoef(x, b.double().cpu().numpy())[0, 1]) < 3e-3
    assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 3e-3
    assert np.isfinite(x).all() and np.abs(x).max() > 4.5

    natoms = 20000
    s = System(natoms, 1, dt, dev)
    s.set_masses(torch.full((natoms,), 12.0, dtype=dt))
    zero = ConstantForces(torch.zeros(1, natoms, 3, dtype=dt, device=dev), torch.full((natoms,), 12.0))
    integ = Integrator(s, zero, timestep=4.0, device=dev, gamma=50.0, T=300.0)
    for _ in range(20):
        ekin, _, T = integ.step(50)
    # stationary state of the reference's update v += -g v dt + sqrt(2 g kT dt / m) xi  (integrator.py:72-74):
    # <v^2> = (kT/m) / (1 - g dt / 2); here g dt = 50/ps * 4 fs = 0.2 -> 333.3 K; sigma_T ~ 1.9 K
    assert abs(T[0] - 300.0 / (1 - 0.5 * 50.0 * 0.004)) < 8.0


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_nve_trajectory_vs_reference(prec):
    """5 NVE steps on tests/water (R=2) with Forces+Integrator on the GPU vs the reference trajectory."""
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator
    from torchmd_amd.systems import System

    g = load("water291")
    dev, dt = _dev(), PREC[prec]
    par = GoldenParameters(g, dt)
    terms = ["lj", "bonds", "angles", "electrostatics"]
    s = System(291, 2, dt, dev)
    s.set_positions(g["pos"][:, :, None])
    s.set_box(g["box"])
    s.set_velocities(torch.tensor(g["traj_vel0"]))
    f = Forces(par, terms=terms, cutoff=7.3, rfa=True)
    integ = Integrator(s, f, 1.0, dev, gamma=None, T=None)
    f.compute(s.pos, s.box, s.forces)
    ekin, pot, T = integ.step(niter=5)
    tol = 1e-9 if prec == "f64" else 2e-4
    assert np.abs(s.pos.cpu().numpy() - g[f"{prec}_traj_pos"]).max() < tol
    assert np.abs(s.vel.cpu().numpy() - g[f"{prec}_traj_vel"]).max() < tol
    assert np.abs(s.forces.cpu().numpy() - g[f"{prec}_traj_forces"]).max() < (1e-7 if prec == "f64" else 5e-3)
    assert np.allclose(ekin, g[f"{prec}_traj_ekin"], rtol=1e-5)
    assert np.allclose(pot, g[f"{prec}_traj_pot"], rtol=1e-5, atol=1e-3)
    assert np.allclose(T, g[f"{prec}_traj_T"], rtol=1e-5)


def test_energy_conservation_and_list_reuse():
    """NVE on the 5 184-atom water box through the cell-list path: total energy drift stays small over
    300 steps of 0.5 fs and the Verlet list is rebuilt only occasionally."""
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev = _dev()
    mol, pos, box = tip3p_box(12, seed=11)
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=torch.float64)
    s = System(mol.numAtoms, 1, torch.float64, dev)
    s.set_positions(pos[:, :, None])
    s.set_box(box)
    torch.manual_seed(0)
    f = Forces(par, terms=terms, cutoff=9.0, rfa=True, switch_dist=7.5, switch_mode="exact", algorithm="celllist")
    # relax the lattice start with strong friction, then switch the thermostat off
    s.set_velocities(maxwell_boltzmann(par.masses, 300, 1))
    f.compute(s.pos, s.box, s.forces)
    Integrator(s, f, 0.5, dev, gamma=20.0, T=300.0).step(400)
    nve = Integrator(s, f, 0.5, dev)
    e = []
    r0 = f.stats(s.pos)["n_rebuilds"]
    for _ in range(6):
        ekin, pot, T = nve.step(50)
        e.append(ekin[0] + pot[0])
    drift = abs(e[-1] - e[0]) / mol.numAtoms
    assert drift < 2e-3, (drift, e)  # kcal/mol per atom over 150 fs (reaction-field cutoff noise)
    rebuilds = f.stats(s.pos)["n_rebuilds"] - r0
    assert 1 <= rebuilds <= 60, rebuilds


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_wrapper_vs_oracle(prec):
    """Molecule wrapping (reference wrapper.py:8-30): waters + a few free ions + one 100-atom chain (the
    wave-per-group path), 2 replicas, against the oracle's restatement of the Python loop."""
    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import tip3p_box
    from torchmd_amd.wrapper import Wrapper

    dev, dt = _dev(), PREC[prec]
    mol, pos, box = tip3p_box(6, seed=9)  # 216 waters
    nw = mol.numAtoms
    nion, nchain = 5, 100
    n = nw + nion + nchain
    bonds = np.concatenate([mol.bonds, np.stack([nw + nion + np.arange(nchain - 1), nw + nion + np.arange(1, nchain)], 1)])
    rng = np.random.default_rng(3)
    extra = rng.uniform(0, box[0], size=(nion + nchain, 3))
    allpos = np.concatenate([pos, extra])
    # scatter molecules over several periodic images
    shift = np.zeros_like(allpos)
    molshift = rng.integers(-3, 4, size=(nw // 3, 3)) * box
    shift[:nw] = np.repeat(molshift, 3, axis=0)
    shift[nw:nw + nion] = rng.integers(-2, 3, size=(nion, 3)) * box
    shift[nw + nion:] = np.array([2, -1, 1]) * box
    p = torch.tensor(np.stack([allpos + shift, allpos - 0.37 * shift + 1.234]), dtype=dt)
    b = box_tensor(box, 2, dt)
    w = Wrapper(n, bonds, dev)
    assert w.ngroups == nw // 3 + nion + 1 and w.has_big and len(w.nongrouped) == nion and len(w.groups) == nw // 3 + 1
    ref = p.clone()
    orc.wrap_molecules(ref, b, [g.cpu() for g in w.groups], w.nongrouped.cpu())
    got = p.clone().to(dev)
    w.wrap(got, b.to(dev))
    diff = (got.cpu() - ref).abs()
    tol = 1e-9 if prec == "f64" else 2e-4
    # a group whose centre sits within rounding of a box face may legitimately land on the other image
    bad = (diff > tol).any(dim=2).sum().item()
    assert bad <= 3, bad
    com_ok = got.cpu()[:, :nw].reshape(2, -1, 3, 3).mean(dim=2)
    assert (com_ok >= -1e-3).all() and (com_ok <= box[0] + 1e-3).all()
    # all-zero box and wrapidx are no-ops (reference behaviour)
    z = p.clone().to(dev)
    w.wrap(z, torch.zeros(2, 3, 3, dtype=dt, device=dev))
    assert torch.equal(z.cpu(), p)
    w.wrap(z, b.to(dev), wrapidx=torch.tensor([0, 1, 2]))
    assert torch.equal(z.cpu(), p)


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("case", ["r2", "ions"])
def test_wrapper_vs_reference_golden(prec, case):
    """tmdhip_wrap equals the reference `Wrapper.wrap` (torchmd/wrapper.py:8-30) on the golden produced by the
    reference itself (tests/golden/wrap.npz): bit for bit in fp64 and fp32 (translations are whole box
    vectors; a group whose centre sits within rounding of a face could legitimately differ, none does here)."""
    from _golden import load
    from torchmd_amd.wrapper import Wrapper

    dev, dt = _dev(), PREC[prec]
    g = load("wrap")
    natoms = int(g[f"{case}_natoms"])
    pos = torch.tensor(g[f"{case}_{prec}_pos_in"], dtype=dt, device=dev)
    R = pos.shape[0]
    box = torch.zeros(R, 3, 3, dtype=dt, device=dev)
    for r in range(R):
        box[r] = torch.diag(torch.tensor(g[f"{case}_boxes"][:, r], dtype=dt))
    w = Wrapper(natoms, g["bonds"], dev)
    w.wrap(pos, box)
    assert np.array_equal(pos.cpu().numpy(), g[f"{case}_{prec}_pos_out"])


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_fused_md_run_equals_stepwise_loop(prec, monkeypatch):
    """tmdhip_md_run (fused half-kick / drift / displacement-test kernels, whole loop in C) reproduces the
    step-by-step Python loop (first_vv -> compute -> langevin_second_vv) bit for bit, incl. the noise.
    (Velocity-dependent skins off: only the fused loop knows velocities at a rebuild, so the two would rebuild
    at different steps and sum a list's entries in a different order.)"""
    monkeypatch.setenv("TMDHIP_VSKIN", "0")
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    class ZeroExternal:  # forces the Integrator onto its generic Python loop
        def calculate(self, pos, box):
            return torch.zeros(pos.shape[0], device=pos.device), torch.zeros_like(pos)

    dev, dt = _dev(), PREC[prec]
    mol, pos, box = tip3p_box(12, seed=21)
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
    torch.manual_seed(5)
    vel0 = maxwell_boltzmann(par.masses, 300, 2)
    out = []
    for ext in (None, ZeroExternal()):
        s = System(mol.numAtoms, 2, dt, dev)
        s.set_positions(pos[:, :, None])
        s.set_box(box)
        s.set_velocities(vel0)
        f = Forces(par, terms=terms, cutoff=9.0, rfa=True, external=ext)
        f.compute(s.pos, s.box, s.forces)
        torch.manual_seed(77)
        integ = Integrator(s, f, 1.0, dev, gamma=1.0, T=300.0)
        res = [integ.step(7), integ.step(1), integ.step(12)]
        assert f.stats(s.pos)["n_rebuilds"] >= 2
        out.append((s.pos.cpu(), s.vel.cpu(), s.forces.cpu(), res))
    (p0, v0, f0, r0), (p1, v1, f1, r1) = out
    assert torch.equal(p0, p1) and torch.equal(v0, v1) and torch.equal(f0, f1)
    for a, b in zip(r0, r1):
        assert np.allclose(a[0], b[0], rtol=1e-12) and np.allclose(a[1], b[1], rtol=1e-12)


def test_replica_batched_md_matches_single_replica_runs():
    """tmdhip_md_run batches the replicas of all-pairs systems into single launches: 25 NVE steps of
    R = 3 water replicas (different start coordinates) == three separate R = 1 runs."""
    import numpy as np

    from _golden import GoldenParameters, load
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator
    from torchmd_amd.systems import System

    g = load("water291")
    dev = torch.device("cuda:0")
    par = GoldenParameters(g, torch.float64)
    pos0 = np.asarray(g["pos"], dtype=np.float64).reshape(-1, 3)
    box0 = np.asarray(g["box"], dtype=np.float64).reshape(-1)[:3]
    n = pos0.shape[0]
    rng = np.random.default_rng(11)
    R = 3
    starts = [pos0 + 0.02 * r * rng.standard_normal(pos0.shape) for r in range(R)]
    vels = [0.01 * rng.standard_normal(pos0.shape) for r in range(R)]
    terms = ["bonds", "angles", "electrostatics", "lj"]

    def run(idx):
        k = len(idx)
        s = System(n, k, torch.float64, dev)
        s.set_positions(np.stack([starts[i] for i in idx], axis=2))
        s.set_box(np.stack([box0 for _ in idx], axis=1))
        s.set_velocities(torch.tensor(np.stack([vels[i] for i in idx])))
        f = Forces(par, terms=terms, cutoff=7.3, rfa=True)
        f.compute(s.pos, s.box, s.forces)
        integ = Integrator(s, f, 1.0, dev)
        ek, ep, T = integ.step(25)
        return s.pos.cpu().numpy(), np.asarray(ek), np.asarray(ep)

    pb, ekb, epb = run([0, 1, 2])
    for r in range(R):
        p1, ek1, ep1 = run([r])
        assert np.abs(pb[r] - p1[0]).max() < 1e-9, r
        assert abs(ekb[r] - ek1[0]) < 1e-9 * max(1.0, abs(ek1[0]))
        assert abs(epb[r] - ep1[0]) < 1e-9 * max(1.0, abs(ep1[0]))
    assert np.abs(pb[0] - pb[2]).max() > 1e-3


@pytest.mark.parametrize("case", ["langevin", "nve", "langevin-switch", "nve-boxes", "nve-f64", "nve-thrombin"])
def test_replicas_of_a_celllist_context_in_one_launch_are_bit_identical(case, monkeypatch):
    """The reference's batch axis on the cell-list path (systems.py:6-18, forces.py:105,116): fp32 contexts with several
    replicas make ONE pair + step launch per MD step for all of them (round 6; every replica keeps its own neighbour state,
    rebuild flags and pacing reports).  3 replicas of the 5 184-atom water box with different coordinates and velocities
    (`boxes`: and different boxes), 2 x 30 steps (device-side rebuilds inside, two calls that end in a launch with
    energies): positions, velocities and forces are bit-identical to the replica-by-replica loop of the same context
    (TMDHIP_BATCH_REPLICAS=0), the returned energies to 1e-12 / 2e-7 (sums of the same terms in another order), and — without
    thermostat, whose noise rows are numbered through the replicas of a context — to separate single-replica contexts.
    fp64 contexts keep the loop (`f64`: batch and single-replica runs agree all the same).  `thrombin`: a heavy topology
    (4 676-atom protein-ligand complex, all seven terms, open boundaries, cutoff 9 A) — the bonded force of every replica comes
    from the wave-per-atom bonded kernel in front of the launch, on the final step with its energies."""
    import numpy as np

    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev = torch.device("cuda:0")
    dt = torch.float64 if case.endswith("f64") else torch.float32
    langevin = case.startswith("langevin")
    rng = np.random.default_rng(5)
    if "thrombin" in case:
        g = load("thrombin")
        par = GoldenParameters(g, dt)
        pos0, box0 = np.asarray(g["pos"], dtype=np.float64), np.zeros(3)
        n = pos0.shape[0]
        terms = ["bonds", "angles", "dihedrals", "impropers", "1-4", "electrostatics", "lj"]
        R, kw, jitter, vscale = 2, dict(cutoff=9.0, algorithm="celllist"), 0.01, 0.01
    else:
        mol, pos0, box0 = tip3p_box(12, seed=3)
        n = mol.numAtoms
        terms = ["lj", "electrostatics", "bonds", "angles"]
        par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
        R, kw, jitter, vscale = 3, dict(cutoff=9.0, rfa=True, **({"switch_dist": 7.5} if "switch" in case else {})), 0.05, 0.02
    scale = [1.0, 1.0, 1.0] if "boxes" not in case else [1.0, 1.004, 0.997]
    starts = [(pos0 + jitter * rng.standard_normal(pos0.shape)) * scale[r] for r in range(R)]
    vels = [vscale * (1 + r) * rng.standard_normal(pos0.shape) for r in range(R)]
    monkeypatch.setenv("TMDHIP_LPA", "16")  # (a context picks its lanes per atom from the atoms that share a launch: pin it)

    def run(idx, batch):
        monkeypatch.setenv("TMDHIP_BATCH_REPLICAS", "1" if batch else "0")
        k = len(idx)
        s = System(n, k, dt, dev)
        s.set_positions(np.stack([starts[i] for i in idx], axis=2))
        s.set_box(np.stack([box0 * scale[i] for i in idx], axis=1))
        s.set_velocities(torch.tensor(np.stack([vels[i] for i in idx])))
        f = Forces(par, terms=terms, **kw)
        f.compute(s.pos, s.box, s.forces)
        torch.manual_seed(3)
        integ = Integrator(s, f, 1.0, dev, **(dict(gamma=0.5, T=300.0) if langevin else {}))
        out = [integ.step(30), integ.step(30)]
        st = f.stats(s.pos)
        res = (s.pos.clone(), s.vel.clone(), s.forces.clone(), out, st)
        f.close()
        return res

    pb, vb, fb, ob, stb = run(list(range(R)), True)
    ps, vs, fs, os_, sts = run(list(range(R)), False)
    assert stb["n_rebuilds"] >= 1 and sts["batched_launches"] == 0 and stb["algorithm"] == "celllist"
    if dt == torch.float32:
        assert stb["batched_launches"] >= 58 and stb["final_steps_in_pair_launch"] == 2, stb  # 2 x (29 interior + 1 final) launches
        assert stb["steps_in_pair_launch"] >= 56
    else:
        assert stb["batched_launches"] == 0
    assert torch.equal(pb, ps) and torch.equal(vb, vs) and torch.equal(fb, fs)
    for a, b in zip(ob, os_):
        assert np.allclose(a[1], b[1], rtol=1e-12) and np.allclose(a[0], b[0], rtol=2e-7) and np.allclose(a[2], b[2], rtol=2e-7)
    assert (pb[0] - pb[R - 1]).abs().max().item() > 1e-3  # the replicas really differ
    if not langevin:
        for r in range(R):
            p1, v1, f1, o1, _ = run([r], True)
            assert torch.equal(pb[r], p1[0]) and torch.equal(vb[r], v1[0]) and torch.equal(fb[r], f1[0]), r
            assert abs(ob[1][1][r] - o1[1][1][0]) <= 1e-12 * max(1.0, abs(o1[1][1][0]))


def test_replica_batched_langevin_equals_stepwise_loop():
    """Batched all-pairs MD (one launch per kernel for all replicas) with the Langevin thermostat against
    the step-by-step Python loop over the stateless integrator kernels: same noise rows (replica * natoms +
    atom), same trajectory."""
    import numpy as np

    from _golden import GoldenParameters, load
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator
    from torchmd_amd.systems import System

    class ZeroExternal:  # forces the Integrator onto its generic Python loop
        def calculate(self, pos, box):
            return torch.zeros(pos.shape[0], device=pos.device), torch.zeros_like(pos)

    g = load("water291")
    dev = torch.device("cuda:0")
    par = GoldenParameters(g, torch.float64)
    pos0 = np.asarray(g["pos"], dtype=np.float64).reshape(-1, 3)
    box0 = np.asarray(g["box"], dtype=np.float64).reshape(-1)[:3]
    n, R = pos0.shape[0], 3
    rng = np.random.default_rng(3)
    starts = np.stack([pos0 + 0.02 * r * rng.standard_normal(pos0.shape) for r in range(R)], axis=2)
    vel0 = torch.tensor(0.01 * rng.standard_normal((R, n, 3)))
    out = []
    for ext in (None, ZeroExternal()):
        s = System(n, R, torch.float64, dev)
        s.set_positions(starts)
        s.set_box(np.stack([box0 * (1 + 0.01 * r) for r in range(R)], axis=1))
        s.set_velocities(vel0)
        f = Forces(par, terms=["bonds", "angles", "electrostatics", "lj"], cutoff=7.3, rfa=True, external=ext)
        f.compute(s.pos, s.box, s.forces)
        torch.manual_seed(9)
        integ = Integrator(s, f, 1.0, dev, gamma=1.0, T=300.0)
        res = [integ.step(6), integ.step(1)]
        assert f.stats(s.pos)["algorithm"] == "allpairs"
        out.append((s.pos.cpu(), s.vel.cpu(), res))
    (p0, v0, r0), (p1, v1, r1) = out
    # the all-pairs kernel combines partial forces with float atomics: equality up to summation order
    assert (p0 - p1).abs().max().item() < 1e-10 and (v0 - v1).abs().max().item() < 1e-10
    for a, b in zip(r0, r1):
        assert np.allclose(a[0], b[0], rtol=1e-9) and np.allclose(a[1], b[1], rtol=1e-9)
    assert (p0[0] - p0[2]).abs().max().item() > 1e-3


@pytest.mark.parametrize("skin_weights", [None, "mass"])
def test_list_overflow_is_replayed_not_raised(monkeypatch, skin_weights):
    """A neighbour list that overflows during an Integrator.step() batch (a device-side rebuild finds more
    neighbours than the capacity sized at the first build) no longer invalidates the trajectory: the batch is
    rewound to its entry state (tmdhip_md_restore), the capacity grown, and the batch repeated with the same
    noise stream.  (Not bit-identical to a run with ample capacity: the replay rebuilds its list at the entry
    positions, i.e. at other steps, and the fp32 summation order follows the list.)  Checked: no exception, the
    lists grew, nothing is truncated at the end, the forces of the final state are those of a fresh evaluation,
    and the thermodynamic state matches the untroubled run."""
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev, dt = _dev(), torch.float32
    mol, pos, box = tip3p_box(14, seed=4)  # 8 232 atoms on a lattice: uniform neighbour counts at the start
    # expanded by 15 % (each molecule moved as a whole): ~290 neighbours per atom, after melting the largest
    # count exceeds the tight capacity (observed maximum rounded up to the list granule) by tens of entries
    com = pos.reshape(-1, 3, 3).mean(axis=1, keepdims=True)
    pos = (pos.reshape(-1, 3, 3) + 0.15 * com).reshape(-1, 3)
    box = box * 1.15
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=dt)

    monkeypatch.setenv("TMDHIP_LPA", "8")  # list granule of 32 entries per atom (a small system would get 256)

    def run(tight):
        if tight:
            monkeypatch.setenv("TMDHIP_DEBUG_LIST_SLACK", "0")
        else:
            monkeypatch.delenv("TMDHIP_DEBUG_LIST_SLACK", raising=False)
        s = System(mol.numAtoms, 1, dt, dev)
        s.set_positions(pos[:, :, None])
        s.set_box(box)
        torch.manual_seed(5)
        s.set_velocities(maxwell_boltzmann(par.masses, 300.0, 1))
        f = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist", skin_weights=skin_weights)
        f.compute(s.pos, s.box, s.forces)
        cap0 = f.stats(s.pos)["max_neighbours"]
        torch.manual_seed(6)
        integ = Integrator(s, f, 1.0, dev, gamma=1.0, T=300.0)
        out = integ.step(400)  # the lattice melts: the largest neighbour count grows by far more than 32
        st = f.stats(s.pos)
        fresh = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
        F2 = torch.zeros_like(s.pos)
        fresh.compute(s.pos, s.box, F2)
        ferr = (F2 - s.forces).abs().max().item()
        return out, cap0, st, ferr

    out_ref, cap_ref, st_ref, ferr_ref = run(False)
    out_t, cap_t, st_t, ferr_t = run(True)
    assert cap_t < cap_ref and st_t["max_neighbours"] > cap_t, (cap_t, cap_ref, st_t)  # the tight run had to grow its lists
    assert st_t["overflow"] == 0 and st_ref["overflow"] == 0
    print(f"list overflow replay: max|dF| vs a fresh evaluation {ferr_t:.3e} (tight lists) / {ferr_ref:.3e}")
    assert ferr_t < 6e-4 and ferr_ref < 6e-4  # (fp32, hot lattice start: FTOL_HOT of test_gpu_parity.py)
    assert abs(out_t[2][0] - out_ref[2][0]) < 15.0  # temperature (K)
    assert abs(out_t[1][0] - out_ref[1][0]) < 0.01 * abs(out_ref[1][0])  # potential energy


def test_rebuild_chain_left_out_and_violation_rewound(monkeypatch):
    """Chain skipping of `tmdhip_md_run` (the host leaves the five early-exit launches of the rebuild chain out
    while no atom is near its displacement limit).  (i) Regular operation on a small box (size gate lowered):
    chains are left out, no violation, same physics as with every chain in place.  (ii) With the "near" report
    disabled (fraction 2: no atom ever counts as near) the first atom to cross its limit does so in a step without
    a chain: the violation flag rewinds the batch, it is repeated with every chain, and forces are still those of
    a fresh evaluation."""
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev, dt = _dev(), torch.float32
    mol, pos, box = tip3p_box(16, seed=2)  # 12 288 atoms
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
    monkeypatch.setenv("TMDHIP_LPA", "8")
    monkeypatch.setenv("TMDHIP_DEBUG_CHAIN_MIN_ENTRIES", "1")

    def run(skip, near=None):
        monkeypatch.setenv("TMDHIP_CHAIN_SKIP", "1" if skip else "0")
        if near is None:
            monkeypatch.delenv("TMDHIP_DEBUG_CHAIN_NEAR", raising=False)
        else:
            monkeypatch.setenv("TMDHIP_DEBUG_CHAIN_NEAR", str(near))
        s = System(mol.numAtoms, 1, dt, dev)
        s.set_positions(pos[:, :, None])
        s.set_box(box)
        torch.manual_seed(5)
        s.set_velocities(maxwell_boltzmann(par.masses, 300.0, 1))
        f = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
        f.compute(s.pos, s.box, s.forces)
        torch.manual_seed(6)
        integ = Integrator(s, f, 1.0, dev, gamma=1.0, T=300.0)
        out = None
        for _ in range(3):
            out = integ.step(60)
        st = f.stats(s.pos)
        fresh = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
        F2 = torch.zeros_like(s.pos)
        fresh.compute(s.pos, s.box, F2)
        return out, st, (F2 - s.forces).abs().max().item()

    out0, st0, ferr0 = run(False)
    out1, st1, ferr1 = run(True)
    out2, st2, ferr2 = run(True, near=2.0)
    assert st0["chains_skipped"] == 0 and st1["chains_skipped"] > 60 and st2["chains_skipped"] > 0
    assert st0["n_rebuilds"] > 5 and st1["n_rebuilds"] > 5
    for st in (st0, st1, st2):
        assert st["overflow"] == 0
    print(f"chain skipping: max|dF| vs a fresh evaluation {ferr0:.3e} / {ferr1:.3e} / {ferr2:.3e}")
    assert ferr0 < 6e-4 and ferr1 < 6e-4 and ferr2 < 6e-4  # (fp32, hot lattice start)
    # regular skipping changes nothing but the launches that would have returned at once: same trajectory
    assert out1[1][0] == out0[1][0] and out1[2][0] == out0[2][0]
    # the rewound run rebuilt its lists at other steps: same physics, different rounding
    assert abs(out2[2][0] - out0[2][0]) < 25.0
    assert abs(out2[1][0] - out0[1][0]) < 0.02 * abs(out0[1][0])


def test_rebuild_chain_left_out_with_two_list_replicas(monkeypatch):
    """Chain skipping with two replicas on the cell-list path (each has its own list, flags and progress words):
    bit-identical to the run with every chain in place."""
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev, dt = _dev(), torch.float32
    mol, pos, box = tip3p_box(12, seed=3)  # 5 184 atoms
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
    monkeypatch.setenv("TMDHIP_DEBUG_CHAIN_MIN_ENTRIES", "1")
    rng = np.random.default_rng(0)
    pos2 = np.stack([pos, pos + rng.normal(0, 0.02, size=pos.shape)], axis=2)  # two slightly different replicas

    def run(skip):
        monkeypatch.setenv("TMDHIP_CHAIN_SKIP", "1" if skip else "0")
        s = System(mol.numAtoms, 2, dt, dev)
        s.set_positions(pos2)
        s.set_box(np.stack([box, box], axis=1))
        torch.manual_seed(5)
        s.set_velocities(maxwell_boltzmann(par.masses, 300.0, 2))
        f = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
        f.compute(s.pos, s.box, s.forces)
        torch.manual_seed(6)
        integ = Integrator(s, f, 1.0, dev, gamma=1.0, T=300.0)
        for _ in range(2):
            out = integ.step(50)
        st = [f.stats(s.pos, r) for r in range(2)]
        return out, st, s.pos.clone(), s.vel.clone()

    out0, st0, p0, v0 = run(False)
    out1, st1, p1, v1 = run(True)
    assert all(st["chains_skipped"] == 0 for st in st0) and all(st["chains_skipped"] > 30 for st in st1)
    assert all(st["n_rebuilds"] > 3 and st["overflow"] == 0 for st in st0 + st1)
    assert torch.equal(p0, p1) and torch.equal(v0, v1)
    assert (p0[0] - p0[1]).abs().max().item() > 1e-3


def test_aged_lists_hold_every_pair_inside_the_cutoff(monkeypatch):
    """Exactness of the skin policy over a trajectory: per-atom skins by mass, skins sized from the velocities at
    every rebuild and rebuild chains left out.  After every few MD steps the number of pairs inside the cutoff
    found through the CURRENT list (aged by several steps, no rebuild forced) must equal the oracle's count at
    those positions — a pair missing from a list would show — and the forces must be those of a fresh evaluation."""
    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev, dt = _dev(), torch.float32
    mol, pos, box = tip3p_box(16, seed=7)  # 12 288 atoms
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
    monkeypatch.setenv("TMDHIP_LPA", "8")
    monkeypatch.setenv("TMDHIP_DEBUG_CHAIN_MIN_ENTRIES", "1")
    monkeypatch.setenv("TMDHIP_CHAIN_SKIP", "1")
    monkeypatch.delenv("TMDHIP_VSKIN", raising=False)
    s = System(mol.numAtoms, 1, dt, dev)
    s.set_positions(pos[:, :, None])
    s.set_box(box)
    torch.manual_seed(11)
    s.set_velocities(maxwell_boltzmann(par.masses, 300.0, 1))
    f = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
    f.compute(s.pos, s.box, s.forces)
    integ = Integrator(s, f, 1.0, dev, gamma=1.0, T=300.0)
    integ.step(150)  # melt the lattice start
    excl = orc.exclusion_pairs(par)
    aged = 0
    worst = 0.0
    for k in range(12):
        r0 = f.stats(s.pos)["n_rebuilds"]
        integ.step(4)
        n_gpu = f.count_pairs(s.pos, s.box)[0]  # displacement test only: the list of the MD run is used as it is
        aged += f.stats(s.pos)["n_rebuilds"] == r0
        p = s.pos.detach().cpu()
        pairs = orc.candidate_pairs(p[0].double().numpy(), box, 9.3, excl)
        _, Fo, npairs = orc.compute(par, p, s.box.cpu(), terms, pairs=pairs, cutoff=9.0, rfa=True)
        assert n_gpu == npairs[0], (k, n_gpu, npairs)
        # forces of the MD run's last step (aged list, lean kernel + inline/separate bonded kernels) vs the oracle
        err = (s.forces.cpu() - Fo).abs().max().item()
        worst = max(worst, err)
        assert err < 6e-4, (k, err)  # (fp32, a melted lattice start: FTOL_HOT of test_gpu_parity.py)
    st = f.stats(s.pos)
    assert aged >= 3 and st["chains_skipped"] > 20 and st["overflow"] == 0
    print(f"rebuilds {st['n_rebuilds']}, chains skipped {st['chains_skipped']}, worst max|dF| vs the oracle {worst:.3e}")
    fresh = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist", skin_weights=None)
    F2 = torch.zeros_like(s.pos)
    fresh.compute(s.pos, s.box, F2)
    assert (F2 - s.forces).abs().max().item() < 2e-3


@pytest.mark.gpu
def test_fused_launch_timeout_falls_back_to_the_integrator_kernel(monkeypatch):
    """Fail-safe of the fused pair + step launch.  Its step blocks wait for force records written by pair blocks of
    the SAME launch, which relies on workgroups being dispatched in block order; the wait is bounded.  The test knob
    makes the step blocks of the 7th fused launch wait for a launch number nobody writes: they give up
    (F_STEP_TIMEOUT) without integrating, `tmdhip_md_observe` reports the batch invalid, Integrator.step rewinds it
    ONCE and repeats it with the separate integrator kernel (no fused launch, every rebuild chain in place).  The
    result must equal — bit for bit — an unfused run that also rebuilds its list in the first step of the call (what
    the rewind does), and the next step() call fuses again."""
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev, dt = _dev(), torch.float32
    mol, pos, box = tip3p_box(14, seed=4)  # 8 232 atoms
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
    monkeypatch.setenv("TMDHIP_LPA", "8")
    torch.manual_seed(3)
    vel0 = maxwell_boltzmann(par.masses, 300.0, 1)

    def run(knob):
        if knob:
            monkeypatch.setenv("TMDHIP_DEBUG_STEP_TIMEOUT", "7")
            monkeypatch.delenv("TMDHIP_FUSED_STEP", raising=False)
            monkeypatch.delenv("TMDHIP_CHAIN_SKIP", raising=False)
        else:
            monkeypatch.delenv("TMDHIP_DEBUG_STEP_TIMEOUT", raising=False)
            monkeypatch.setenv("TMDHIP_FUSED_STEP", "0")
            monkeypatch.setenv("TMDHIP_CHAIN_SKIP", "0")
        s = System(mol.numAtoms, 1, dt, dev)
        s.set_positions(pos[:, :, None])
        s.set_box(box)
        s.set_velocities(vel0)
        f = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
        f.compute(s.pos, s.box, s.forces)
        if not knob:
            f.invalidate_lists(s.pos)  # the rewound batch of the other run starts with a re-plan + rebuild as well
        torch.manual_seed(9)
        integ = Integrator(s, f, 1.0, dev, gamma=1.0, T=300.0)
        res = [integ.step(30)]
        mid = f.stats(s.pos)
        monkeypatch.delenv("TMDHIP_DEBUG_STEP_TIMEOUT", raising=False)
        res.append(integ.step(20))
        return s.pos.cpu(), s.vel.cpu(), s.forces.cpu(), res, mid, f.stats(s.pos)

    p1, v1, f1, r1, mid1, st1 = run(True)
    p0, v0, f0, r0, mid0, st0 = run(False)
    assert mid1["fused_step_timeouts"] == 1 and st1["fused_step_timeouts"] == 1, (mid1, st1)
    assert mid1["steps_in_pair_launch"] == 29  # the first attempt of the batch; its repetition made none
    assert st1["steps_in_pair_launch"] == 29 + 19  # the next call fuses again
    assert st0["fused_step_timeouts"] == 0 and st0["steps_in_pair_launch"] == 0
    assert torch.isfinite(p1).all()
    assert torch.equal(p1, p0) and torch.equal(v1, v0) and torch.equal(f1, f0)
    # (observables: the fused run's last launch sums the bonded and kinetic energies in its step blocks, the unfused one in
    # the bonded / kinetic-energy kernels — the same terms in another order)
    for a, b in zip(r1, r0):  # (Ekin, Epot, T): the potential energy to fp64 round-off — a dropped or doubled term would show —,
        for k, (x, y) in enumerate(zip(a, b)):  # Ekin and T to fp32 round-off
            assert np.allclose(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64), rtol=1e-12 if k == 1 else 2e-7,
                               atol=1e-9 if k == 1 else 0), (a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["water_langevin", "water_nve", "water_two_replicas", "lj_langevin", "water_counter_wraps",
                                  "water_32_lanes", "water_64_lanes", "thrombin"])
def test_step_blocks_of_the_pair_launch_are_bit_identical(case, monkeypatch):
    """Interior steps of tmdhip_md_run on the lean fp32 pair kernel are made by the pair launch itself ("step blocks"
    behind the pair blocks wait for the pair waves of their atoms: FusedStep in csrc/engine.h, pair_fast_f32.hip) instead of by an
    integrator launch.  Same device functions in the same order: positions, velocities, forces and energies equal
    those of the separate kernels (TMDHIP_FUSED_STEP=0) bit for bit — with velocity-dependent skins, rebuilds inside
    the window and chain skipping active (size gate opened).  Water = 8 lanes per atom (two pair blocks per step
    block) + inline bonded records + reaction field; the LJ box = 4 lanes per atom, no bonded terms.
    `water_counter_wraps`: the launch number the force records carry starts at 2^32 - 20 and wraps during the run
    (0 is skipped: it means "never written").  `thrombin`: a protein (4 676 atoms, all seven terms, open boundaries) —
    a heavy topology, whose bonded force is evaluated by bonded_wave_kernel in front of the pair launch into a buffer
    that the step blocks add.  (fp64 contexts keep the separate integrator kernel: their step blocks were measured
    slower in round 4 and removed in round 5.)"""
    from torchmd_amd.builders import argon_forcefield, lj_box, tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev, dt = _dev(), torch.float32
    monkeypatch.setenv("TMDHIP_DEBUG_CHAIN_MIN_ENTRIES", "1")
    if case == "water_counter_wraps":
        monkeypatch.setenv("TMDHIP_DEBUG_FUSED_GEN0", str(2**32 - 20))
    nrep = 2 if case == "water_two_replicas" else 1
    if case == "thrombin":
        g = load("thrombin")
        par = GoldenParameters(g, dt)
        pos, box = np.asarray(g["pos"], dtype=np.float64), np.zeros(3)
        terms = ["bonds", "angles", "dihedrals", "impropers", "1-4", "electrostatics", "lj"]
        kw = dict(cutoff=9.0)
    elif case.startswith("water"):
        mol, pos, box = tip3p_box(14, seed=4)  # 8 232 atoms
        terms = ["lj", "electrostatics", "bonds", "angles"]
        par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
        kw = dict(cutoff=9.0, rfa=True)
    else:
        mol, pos, box = lj_box(22, seed=4)  # 10 648 atoms
        terms = ["lj"]
        par = Parameters(argon_forcefield(mol), mol, terms, precision=dt)
        kw = dict(cutoff=9.0)
    gamma = None if case == "water_nve" else 1.0
    lanes = {"water_32_lanes": "32", "water_64_lanes": "64", "thrombin": "64"}.get(case, "8" if case.startswith("water") else "4")
    monkeypatch.setenv("TMDHIP_LPA", lanes)  # (32 / 64: what mid-size boxes get; 8 and 16 pair blocks per step block)
    torch.manual_seed(3)
    vel0 = maxwell_boltzmann(par.masses, 300.0, nrep)

    def run(fused):
        monkeypatch.setenv("TMDHIP_FUSED_STEP", "1" if fused else "0")
        s = System(pos.shape[0], nrep, dt, dev)
        s.set_positions(np.repeat(pos[:, :, None], nrep, axis=2))
        s.set_box(box)
        s.set_velocities(vel0)
        f = Forces(par, terms=terms, algorithm="celllist", **kw)
        f.compute(s.pos, s.box, s.forces)
        torch.manual_seed(9)  # (the noise stream's seed is drawn at construction)
        integ = Integrator(s, f, 1.0, dev, gamma=gamma, T=300.0 if gamma else None)
        res = [integ.step(2), integ.step(41), integ.step(1), integ.step(37)]
        return s.pos.cpu(), s.vel.cpu(), s.forces.cpu(), res, [f.stats(s.pos, r) for r in range(nrep)]

    p1, v1, f1, r1, st1 = run(True)
    p0, v0, f0, r0, st0 = run(False)
    for st in st1:
        assert st["algorithm"] == "celllist" and st["overflow"] == 0
        # every interior step: (2 - 1) + (41 - 1) + 0 + (37 - 1)
        assert st["steps_in_pair_launch"] == 77, st
        assert st["n_rebuilds"] >= 2 and st["chains_skipped"] > 20, st
    for st, su in zip(st1, st0):
        assert su["steps_in_pair_launch"] == 0
        assert st["n_rebuilds"] == su["n_rebuilds"] and st["chains_skipped"] == su["chains_skipped"]
    assert torch.isfinite(p1).all()
    assert torch.equal(p1, p0) and torch.equal(v1, v0) and torch.equal(f1, f0)
    # (observables: the fused run's last launch sums the bonded and kinetic energies in its FINAL step blocks, the unfused
    # one in the bonded / kinetic-energy kernels — the same fp64 terms in another order; Ekin and T are returned in fp32)
    for a, b in zip(r1, r0):  # (Ekin, Epot, T): the potential energy to fp64 round-off — a dropped or doubled term would show —,
        for k, (x, y) in enumerate(zip(a, b)):  # Ekin and T to fp32 round-off
            assert np.allclose(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64), rtol=1e-12 if k == 1 else 2e-7,
                               atol=1e-9 if k == 1 else 0), (a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["water_langevin", "water_nve", "lj_langevin", "thrombin"])
def test_final_step_of_a_call_in_the_pair_launch(case, monkeypatch):
    """The LAST step of a step() call (reference integrator.py:116-125: forces, second half kick, then the energies and
    the kinetic energy the call returns) is made by the last pair launch itself: FINAL step blocks (csrc/md_step.h) apply
    the kick, leave pair + bonded force in `forces`, and sum the bonded and kinetic energies — instead of a bonded
    kernel, a kick kernel and a kinetic-energy kernel behind the launch (TMDHIP_FUSED_FINAL=0 keeps those).  Same device
    functions in the same order: positions, velocities and forces equal bit for bit after calls of 1, 2, 19 and 7 steps;
    energies are the same fp64 terms summed in another order (1e-12), Ekin / T come back in fp32."""
    from torchmd_amd.builders import argon_forcefield, lj_box, tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev, dt = _dev(), torch.float32
    monkeypatch.setenv("TMDHIP_DEBUG_CHAIN_MIN_ENTRIES", "1")
    if case == "thrombin":
        g = load("thrombin")
        par = GoldenParameters(g, dt)
        pos, box = np.asarray(g["pos"], dtype=np.float64), np.zeros(3)
        terms = ["bonds", "angles", "dihedrals", "impropers", "1-4", "electrostatics", "lj"]
        kw = dict(cutoff=9.0)
    elif case.startswith("water"):
        mol, pos, box = tip3p_box(14, seed=4)  # 8 232 atoms
        terms = ["lj", "electrostatics", "bonds", "angles"]
        par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
        kw = dict(cutoff=9.0, rfa=True)
    else:
        mol, pos, box = lj_box(22, seed=4)  # 10 648 atoms
        terms = ["lj"]
        par = Parameters(argon_forcefield(mol), mol, terms, precision=dt)
        kw = dict(cutoff=9.0)
    gamma = None if case == "water_nve" else 1.0
    monkeypatch.setenv("TMDHIP_LPA", "64" if case == "thrombin" else ("8" if case.startswith("water") else "4"))
    torch.manual_seed(3)
    vel0 = maxwell_boltzmann(par.masses, 300.0, 1)

    def run(final):
        monkeypatch.setenv("TMDHIP_FUSED_FINAL", "1" if final else "0")
        s = System(pos.shape[0], 1, dt, dev)
        s.set_positions(pos[:, :, None])
        s.set_box(box)
        s.set_velocities(vel0)
        f = Forces(par, terms=terms, algorithm="celllist", **kw)
        f.compute(s.pos, s.box, s.forces)
        torch.manual_seed(9)
        integ = Integrator(s, f, 1.0, dev, gamma=gamma, T=300.0 if gamma else None)
        res, snaps = [], []
        for k in (1, 2, 19, 7):
            res.append(integ.step(k))
            snaps.append((s.pos.cpu().clone(), s.vel.cpu().clone(), s.forces.cpu().clone()))
        st = f.stats(s.pos)
        f.close()
        return res, snaps, st

    r1, s1, st1 = run(True)
    r0, s0, st0 = run(False)
    assert st1["overflow"] == 0 and st1["fused_step_timeouts"] == 0 and st1["steps_in_pair_launch"] == st0["steps_in_pair_launch"] == 25
    # the FINAL path really ran: one final launch per step() call with it, none without (tmdhip_stats, ABI 8)
    assert st1["final_steps_in_pair_launch"] == 4 and st0["final_steps_in_pair_launch"] == 0, (st1, st0)
    for (p1, v1, f1), (p0, v0, f0) in zip(s1, s0):
        assert torch.isfinite(p1).all()
        assert torch.equal(p1, p0) and torch.equal(v1, v0) and torch.equal(f1, f0)
    for a, b in zip(r1, r0):
        ek1, pot1, T1 = a
        ek0, pot0, T0 = b
        assert np.allclose(pot1, pot0, rtol=1e-12, atol=1e-9), (pot1, pot0)
        assert np.allclose(ek1, ek0, rtol=2e-7) and np.allclose(T1, T0, rtol=2e-7), (ek1, ek0)
    # the energies the fused call returns are those of a fresh evaluation at the final positions
    s = System(pos.shape[0], 1, dt, dev)
    s.pos.copy_(s1[-1][0].to(dev))
    s.set_box(box)
    f = Forces(par, terms=terms, algorithm="celllist", **kw)
    F = torch.zeros_like(s.pos)
    e = f.compute(s.pos, s.box, F)
    assert abs(e[0] - r1[-1][1][0]) <= 3e-5 * max(1.0, abs(e[0])), (e, r1[-1][1])
    assert (F.cpu() - s1[-1][2]).abs().max().item() < 6e-4
    f.close()


@pytest.mark.gpu
def test_wrong_continuation_hint_is_rewound(monkeypatch):
    """The first step of a step() call leaves its rebuild chain out when the positions tensor has not been written
    through torch since the previous call (tmdhip_md_desc::continuation, from the tensor's version counter).  The hint
    can be wrong — here half of the box is sheared by 0.9 A through `.data`, which does not bump the counter —
    and must then cost a rewind, not a wrong result: the displacement test of that first step raises F_VIOLATION, the
    batch is repeated with every chain in place, and the forces equal those of a fresh evaluation."""
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev, dt = _dev(), torch.float32
    monkeypatch.setenv("TMDHIP_DEBUG_CHAIN_MIN_ENTRIES", "1")
    mol, pos, box = tip3p_box(14, seed=6)
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
    s = System(mol.numAtoms, 1, dt, dev)
    s.set_positions(pos[:, :, None])
    s.set_box(box)
    torch.manual_seed(3)
    s.set_velocities(maxwell_boltzmann(par.masses, 300.0, 1))
    f = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
    f.compute(s.pos, s.box, s.forces)
    torch.manual_seed(9)
    integ = Integrator(s, f, 1.0, dev, gamma=1.0, T=300.0)
    integ.step(12)
    integ.step(12)  # (a call whose hint is right: continuation)
    # the hint is USED: one-step calls have no interior step, so any chain they leave out is their first step's (six
    # consecutive steps cannot all follow a near-limit report: the list lives ~9 steps)
    before = f.stats(s.pos)["chains_skipped"]
    for _ in range(6):
        integ.step(1)
    assert f.stats(s.pos)["chains_skipped"] > before
    skipped0 = f.stats(s.pos)["chains_skipped"]
    version = s.pos._version
    # shear the box by 0.9 A (more than any half skin) along the plane x = L/2: whole molecules move (by their oxygen's
    # side), so no bond is stretched and nothing overlaps, but pairs across the plane enter and leave the cutoff
    ox = s.pos[0, 0::3, 0]
    moved = (ox - torch.floor(ox / float(box[0])) * float(box[0])) > 0.5 * float(box[0])
    shift = torch.zeros_like(s.pos)
    shift[0, :, 1] = 0.9 * moved.repeat_interleave(3).to(dt)
    s.pos.data.add_(shift)
    assert s.pos._version == version  # the hint will say "nothing has moved"
    integ.step(8)
    st = f.stats(s.pos)
    assert st["overflow"] == 0 and st["chains_skipped"] > skipped0
    fresh = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
    F2 = torch.zeros_like(s.pos)
    fresh.compute(s.pos, s.box, F2)
    assert torch.isfinite(s.forces).all()
    assert (F2 - s.forces).abs().max().item() < 2e-3


def test_md_run_without_an_energy_buffer():
    """C-ABI callers may pass `energies_dev = NULL` to tmdhip_md_run (include/tmdhip.h): no energies, no report, and — since the
    call's first kernel takes the snapshot of the entry state and clears the energy buffer — nothing may be written through
    the null pointer.  Same trajectory as a call with the buffer."""
    import ctypes as C

    import numpy as np

    from torchmd_amd import _lib as L
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import TIMEFACTOR
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev = torch.device("cuda:0")
    mol, pos, box = tip3p_box(12, seed=8)
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=torch.float32)
    out = []
    for with_buffer in (True, False):
        s = System(mol.numAtoms, 1, torch.float32, dev)
        s.set_positions(pos[:, :, None])
        s.set_box(box)
        s.set_velocities(torch.tensor(0.01 * np.random.default_rng(1).standard_normal((1, mol.numAtoms, 3))))
        f = Forces(par, terms=terms, cutoff=9.0, rfa=True)
        f.compute(s.pos, s.box, s.forces)
        eng = f._engine(s.pos)
        masses = par.masses.to(dev, torch.float32).contiguous()
        boxes = np.ascontiguousarray(np.asarray(box, dtype=np.float64).reshape(1, 3))
        d = L.MdDesc()
        d.struct_size = C.sizeof(L.MdDesc)
        d.niter = 12
        d.pos_dev, d.vel_dev, d.forces_dev = s.pos.data_ptr(), s.vel.data_ptr(), s.forces.data_ptr()
        d.mass_dev = masses.data_ptr()
        d.vcoeff_dev = None
        d.box_host = boxes.ctypes.data_as(C.c_void_p)
        d.dt, d.gamma = 1.0 / TIMEFACTOR, 0.0
        d.seed, d.step0 = 1, 0
        d.energies_dev = eng.ebuf.data_ptr() if with_buffer else None
        d.continuation = 0
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        L.check(eng.lib.tmdhip_md_run(eng.ctx, C.byref(d), stream), "tmdhip_md_run")
        torch.cuda.synchronize()
        assert L.check(eng.lib.tmdhip_check(eng.ctx, 0, stream), "tmdhip_check") == 0
        out.append((s.pos.clone(), s.vel.clone()))
        f.close()
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    assert torch.isfinite(out[1][0]).all()


@pytest.mark.parametrize("trouble", ["overflow", "violation", "eighteen_replicas", "overflow_together", "violation_together"])
def test_replica_batch_recovers_like_the_loop(trouble, monkeypatch):
    """The recovery paths of `Integrator.step` with the replicas of a cell-list context in one launch (round 6).
    `overflow`: lists sized without slack overflow in a device-side rebuild of SOME replica -> the whole batch is rewound and
    repeated (tmdhip_md_restore); `violation`: with the "near" report disabled an atom crosses its limit in a step whose chain
    the host left out -> rewound and repeated with every chain.  Either way: no exception, nothing truncated at the end, the
    forces of every replica are those of a fresh evaluation at its final positions.  `eighteen_replicas`: more replicas than one
    launch holds (kBatchMax = 16: two launches per step), bit-identical to the replica loop.  `*_together`: the same trouble with
    the opt-in TMDHIP_REPLICA_REBUILDS=together (all replicas of a launch rebuild when one has to)."""
    import numpy as np

    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev, dt = torch.device("cuda:0"), torch.float32
    terms = ["lj", "electrostatics", "bonds", "angles"]
    monkeypatch.setenv("TMDHIP_DEBUG_CHAIN_MIN_ENTRIES", "1")
    if trouble.endswith("_together"):
        monkeypatch.setenv("TMDHIP_REPLICA_REBUILDS", "together")
        trouble = trouble.removesuffix("_together")
    if trouble == "overflow":
        mol, pos, box = tip3p_box(14, seed=4)
        com = pos.reshape(-1, 3, 3).mean(axis=1, keepdims=True)
        pos = (pos.reshape(-1, 3, 3) + 0.15 * com).reshape(-1, 3)  # expanded lattice: the lists grow as it melts
        box = box * 1.15
        monkeypatch.setenv("TMDHIP_LPA", "8")
        monkeypatch.setenv("TMDHIP_DEBUG_LIST_SLACK", "0")
        R, steps = 2, 300
    elif trouble == "violation":
        mol, pos, box = tip3p_box(12, seed=6)
        monkeypatch.setenv("TMDHIP_DEBUG_CHAIN_NEAR", "2.0")  # nobody ever counts as near a limit
        R, steps = 3, 60
    else:
        mol, pos, box = tip3p_box(12, seed=6)
        R, steps = 18, 24
    par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
    rng = np.random.default_rng(9)
    # (jitter of at most 0.09 A per coordinate: more than ~0.3 A on every atom of flexible water blows the replica up)
    starts = np.stack([pos + 0.03 * (r % 4) * rng.standard_normal(pos.shape) for r in range(R)], axis=2)

    def run(batch):
        monkeypatch.setenv("TMDHIP_BATCH_REPLICAS", "1" if batch else "0")
        s = System(mol.numAtoms, R, dt, dev)
        s.set_positions(starts)
        s.set_box(box)
        torch.manual_seed(5)
        s.set_velocities(maxwell_boltzmann(par.masses, 300.0, R))
        f = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
        f.compute(s.pos, s.box, s.forces)
        cap0 = f.stats(s.pos)["max_neighbours"]
        torch.manual_seed(6)
        integ = Integrator(s, f, 1.0, dev, gamma=1.0, T=300.0)
        out = integ.step(steps)
        sts = [f.stats(s.pos, r) for r in range(R)]
        res = (s.pos.clone(), s.vel.clone(), s.forces.clone(), out, sts, cap0, integ.replays)
        f.close()
        return res

    pb, vb, fb, ob, stb, cap0, replays = run(True)
    assert stb[0]["batched_launches"] > 0 and all(st["overflow"] == 0 for st in stb)
    if trouble == "eighteen_replicas":
        ps, vs, fs, os_, sts, _, replays_loop = run(False)
        # (two launches per step: 16 + 2 replicas; the jittered starts are hot enough for an atom to outrun the 75 % "near" margin
        # now and then: a rewind, if any, happens in both modes alike)
        assert replays == replays_loop and stb[0]["batched_launches"] >= 2 * (steps - 1)
        assert torch.equal(pb, ps) and torch.equal(vb, vs) and torch.equal(fb, fs)
        assert np.allclose(ob[1], os_[1], rtol=1e-12)
        return
    assert replays >= 1  # the batch was rewound and repeated
    if trouble == "overflow":
        assert max(st["max_neighbours"] for st in stb) > cap0
    # forces of the final state = a fresh evaluation's (fp32, hot start: FTOL_HOT of test_gpu_parity.py)
    monkeypatch.delenv("TMDHIP_DEBUG_LIST_SLACK", raising=False)
    fresh = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
    F2 = torch.zeros_like(pb)
    fresh.compute(pb, torch.stack([torch.diag(torch.tensor(box, dtype=dt, device=dev))] * R), F2)
    fresh.close()
    assert torch.isfinite(pb).all() and (F2 - fb).abs().max().item() < 6e-4
    assert all(100.0 < t < 3000.0 for t in ob[2])  # (jittered lattice starts are hot)


def test_replicas_rebuilding_together_is_an_opt_in_with_valid_lists(monkeypatch):
    """TMDHIP_REPLICA_REBUILDS=together (opt-in): every replica whose chain is in a batched launch rebuilds as soon as one of them
    has to — builds of several replicas cost what one costs.  The lists are as complete as lists built on time, but their entries
    come in another order, so a replica is no longer bit-identical to its run alone; checked here: the replicas rebuild in step
    (same rebuild count), nothing overflows or is rewound, the aged lists hold exactly the pairs inside the cutoff, and the
    forces of every replica's final state are those of a fresh evaluation."""
    import numpy as np

    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev, dt = torch.device("cuda:0"), torch.float32
    monkeypatch.setenv("TMDHIP_REPLICA_REBUILDS", "together")
    mol, pos, box = tip3p_box(12, seed=12)
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
    R = 4
    s = System(mol.numAtoms, R, dt, dev)
    s.set_positions(np.repeat(pos[:, :, None], R, axis=2))
    s.set_box(box)
    torch.manual_seed(2)
    s.set_velocities(maxwell_boltzmann(par.masses, 300.0, R))  # (different velocities: the replicas drift apart)
    f = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
    f.compute(s.pos, s.box, s.forces)
    integ = Integrator(s, f, 1.0, dev, gamma=1.0, T=300.0)
    integ.step(150)
    integ.step(151)
    sts = [f.stats(s.pos, r) for r in range(R)]
    assert integ.replays == 0 and all(st["overflow"] == 0 for st in sts) and sts[0]["batched_launches"] >= 299
    nreb = [st["n_rebuilds"] for st in sts]
    assert max(nreb) - min(nreb) <= 2 and min(nreb) >= 20, nreb  # in step with each other
    n_gpu = f.count_pairs(s.pos, s.box)  # through the run's own (aged) lists
    aged = [f.stats(s.pos, r)["n_rebuilds"] for r in range(R)] == nreb
    excl = orc.exclusion_pairs(par)
    p = s.pos.detach().cpu()
    for r in range(R):
        pairs = orc.candidate_pairs(p[r].double().numpy(), box, 9.4, excl)
        _, Fo, npairs = orc.compute(par, p[r:r + 1], s.box[r:r + 1].cpu(), terms, pairs=pairs, cutoff=9.0, rfa=True)
        assert n_gpu[r] == npairs[0], (r, aged)
        assert (s.forces[r].cpu() - Fo[0]).abs().max().item() < 6e-4  # (FTOL_HOT of test_gpu_parity.py)
