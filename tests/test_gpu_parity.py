"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the C ABI by
`torchmd_amd.forces.Forces`, against (a) golden outputs of the reference itself and (b) the CPU oracle
on the same seeded inputs.

Tolerances (BASELINE.json north_star): forces within 1e-4 kcal/mol/A in fp64 and 1e-2 in fp32.
The asserted bounds below are much tighter where the arithmetic allows it.
"""

import numpy as np
import pytest
import torch

from _golden import GoldenParameters, PREC, box_tensor, energies, load, pos_tensor

pytestmark = pytest.mark.gpu

# North-star bars: 1e-4 (fp64) / 1e-2 (fp32), and "forces within 1e-4" in its target sentence for the 100k-atom box.
# Asserted: fp64 1e-8; fp32 3e-4 — observed on MI355X (round 5, gpurun_out/r05_a/tests_rP.log): alanine dipeptide 6.7e-5,
# 5 184-atom water 5.6e-5, the 98 304-atom box 7.1e-5 (6.1e-5 with unwrapped coordinates): partial forces of up to a few
# hundred kcal/mol/A summed in list order in fp32, against the reference's pair order.
FTOL = {"f64": 1e-8, "f32": 3e-4}
# ... except on states with close contacts (an unrelaxed lattice start after tens of MD steps: |F| of several hundred,
# observed 2.0e-4 at 98 304 atoms): 6e-4
FTOL_HOT = {"f64": 1e-8, "f32": 6e-4}
# ... and the north star's own target sentence pinned where it speaks: the static 98 304-atom box, fp32 (observed 7e-5)
C3_STATIC_FTOL = {"f64": 1e-8, "f32": 1e-4}
ERTOL = {"f64": 1e-10, "f32": 2e-5}
EFAC = 3  # energies: relative tolerance ERTOL * EFAC (fp32: 6e-5; observed <= 2e-5)
ALL_TERMS = ["bonds", "angles", "dihedrals", "impropers", "1-4", "electrostatics", "lj"]


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def _run(par, pos, box, terms, R=1, prec="f64", **kw):
    from torchmd_amd.forces import Forces

    dev = _dev()
    dt = PREC[prec]
    f = Forces(par, terms=terms, **kw)
    p = pos_tensor(pos, R, dt, dev)
    b = box_tensor(box, R, dt, dev)
    F = torch.full_like(p, 7.0)  # must be overwritten, not accumulated
    pots = f.compute(p, b, F, returnDetails=True)
    return pots, F.cpu().numpy(), f, p, b


def _compare(g, tag, pots, F, terms, prec, R=1):
    for r in range(R):
        ref = energies(g, tag, r)
        for t in terms:
            if t == "1-4":
                assert pots[r][t] == 0.0
                continue
            scale = max(1.0, abs(ref[t]))
            assert abs(pots[r][t] - ref[t]) <= ERTOL[prec] * scale * EFAC, (tag, t, pots[r][t], ref[t])
        assert "external" in pots[r]
    err = np.abs(F - g[tag + "_forces"]).max()
    assert err <= FTOL[prec], (tag, err)
    return err


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("rfa", [False, True])
def test_water291(prec, rfa):
    """C1: tests/water, 291 atoms, R=2, cutoff 7.3 (box too small for cells -> all-pairs kernel)."""
    g = load("water291")
    par = GoldenParameters(g, PREC[prec])
    full = ["lj", "bonds", "angles", "electrostatics"]
    for label, terms in (("full", full), ("nb", ["lj", "electrostatics"])):
        pots, F, f, p, b = _run(par, g["pos"], g["box"], terms, R=2, prec=prec, cutoff=7.3, rfa=rfa)
        _compare(g, f"{prec}_{label}_rfa{int(rfa)}", pots, F, terms, prec, R=2)
        assert f.stats(p)["algorithm"] == "allpairs"


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_ala2_all_variants(prec):
    """C2: alanine dipeptide, all 7 terms, cutoff 9 / switch 7.5 / reaction field, periodic and box=0,
    plus no-cutoff Coulomb, the no-switch variant and the repulsion terms."""
    g = load("ala2")
    par = GoldenParameters(g, PREC[prec])
    sw = dict(cutoff=9.0, switch_dist=7.5, rfa=True)
    zero = np.zeros(3)
    worst = 0.0
    for label, terms in (("full", ALL_TERMS), ("nb", ["electrostatics", "lj"])):
        for tag, box, kw in (("pbc", g["box"], sw), ("box0", zero, sw), ("nocut", zero, {})):
            pots, F, *_ = _run(par, g["pos"], box, terms, prec=prec, **kw)
            worst = max(worst, _compare(g, f"{prec}_{label}_{tag}", pots, F, terms, prec))
    pots, F, *_ = _run(par, g["pos"], g["box"], ["electrostatics", "lj"], prec=prec, cutoff=9.0, rfa=True)
    _compare(g, f"{prec}_nb_pbc_noswitch", pots, F, ["electrostatics", "lj"], prec)
    for t in ("repulsion", "repulsioncg"):
        pots, F, *_ = _run(par, g["pos"], g["box"], [t], prec=prec, cutoff=9.0)
        _compare(g, f"{prec}_{t}_pbc", pots, F, [t], prec)
    print(f"ala2 {prec}: max |dF| over variants = {worst:.3e}")


def test_ala2_pair_count_matches_reference_filter():
    """The number of pairs passing `dist <= cutoff` is the reference's (decision arithmetic)."""
    from oracle import torchmd_oracle as orc

    g = load("ala2")
    for prec in ("f64", "f32"):
        par = GoldenParameters(g, PREC[prec])
        _, _, npairs = orc.compute(par, pos_tensor(g["pos"], 1, PREC[prec]), box_tensor(g["box"], 1, PREC[prec]),
                                   ["lj"], cutoff=9.0)
        _, _, f, p, b = _run(par, g["pos"], g["box"], ["lj"], prec=prec, cutoff=9.0)
        assert f.count_pairs(p, b) == npairs


def test_switch_modes():
    """Explicit forces keep upstream's switching quirk; explicit_forces=False gives -dE/dr."""
    from torchmd_amd.forces import Forces

    g = load("ala2")
    dev = _dev()
    par = GoldenParameters(g, torch.float64)
    p = pos_tensor(g["pos"], 1, torch.float64, dev)
    b = box_tensor(g["box"], 1, torch.float64, dev)
    f = Forces(par, terms=["lj"], cutoff=9.0, switch_dist=7.5)
    F_ref = torch.zeros_like(p)
    f.compute(p, b, F_ref)
    F_auto = torch.zeros_like(p)
    pg = p.clone().requires_grad_(True)
    f.compute(pg, b, F_auto, explicit_forces=False)
    assert (F_ref - F_auto).abs().max() > 1e-3  # the quirk is visible (SURVEY: 0.014 on this system)
    # finite-difference check of the exact mode on one coordinate
    h = 1e-4
    e = []
    for s in (+h, -h):
        q = p.clone()
        q[0, 5, 1] += s
        e.append(f.compute(q, b, None, calculateForces=False)[0])
    fd = -(e[0] - e[1]) / (2 * h)
    assert abs(fd - F_auto[0, 5, 1].item()) < 1e-5
    # differentiable potential
    pot = f.compute(pg, b, None, explicit_forces=False, toNumpy=False)
    (grad,) = torch.autograd.grad(pot.sum(), pg)
    assert torch.allclose(-grad, F_auto, atol=1e-12)


def test_vmap_over_compute_like_the_reference_test():
    """Reference tests/test_torchmd.py:552-605: `torch.vmap(forces.compute)` over a batch of positions with
    explicit_forces=False, calculateForces=False, toNumpy=False, then backward() — alanine dipeptide, all
    terms, no cutoff, fp64.  Epot per batch entry = the reference's own value (-1768.8915 for its ingestion of
    the fixture; the golden holds the value for these exact inputs) and -grad = the golden forces."""
    from torchmd_amd.forces import Forces

    g = load("ala2")
    dev = _dev()
    terms = ["bonds", "angles", "dihedrals", "impropers", "1-4", "electrostatics", "lj"]
    par = GoldenParameters(g, torch.float64)
    f = Forces(par, terms=terms, cutoff=None, switch_dist=7.5, rfa=False)
    pos = pos_tensor(g["pos"], 1, torch.float64, dev)  # [1, N, 3]
    box = box_tensor(np.zeros(3), 1, torch.float64, dev)
    batch = 4
    positions = torch.stack([pos] * batch, dim=0)  # [B, 1, N, 3]
    positions[2, 0, 7, 1] += 0.05  # one entry differs: results must not be broadcast copies
    positions.requires_grad = True
    epot = torch.vmap(f.compute, in_dims=(0,))(positions, box=box, forces=None, returnDetails=False,
                                               explicit_forces=False, calculateForces=False, toNumpy=False)
    epot.sum().backward()
    forces = -positions.grad
    assert epot.shape == (batch, 1) and forces.shape == positions.shape
    ref = sum(energies(g, "f64_full_nocut", 0).values())
    assert abs(ref + 1768.8915) < 1e-3  # the literal of the reference test, for its float32-ingested inputs
    assert abs(epot[0].item() - ref) < 1e-8 and abs(epot[1].item() - ref) < 1e-8 and abs(epot[3].item() - ref) < 1e-8
    assert abs(epot[2].item() - ref) > 1e-6
    assert np.abs(forces[0, 0].cpu().numpy() - g["f64_full_nocut_forces"][0]).max() < 1e-8
    assert np.abs(forces[2, 0].cpu().numpy() - g["f64_full_nocut_forces"][0]).max() > 1e-4


def test_thrombin_nocut_and_celllist():
    """4 676-atom non-periodic complex: no-cutoff all-pairs vs the reference golden, then the same
    system with a 9 A cutoff through the cell-list path vs the all-pairs kernel and the oracle."""
    from oracle import torchmd_oracle as orc

    g = load("thrombin")
    par = GoldenParameters(g, torch.float64)
    zero = np.zeros(3)
    terms = ["electrostatics", "lj"]
    pots, F, *_ = _run(par, g["pos"], zero, terms, prec="f64")
    _compare(g, "f64_nb_nocut", pots, F, terms, "f64")
    pots, F, *_ = _run(par, g["pos"], zero, ALL_TERMS, prec="f64")
    _compare(g, "f64_full_nocut", pots, F, ALL_TERMS, "f64")

    kw = dict(cutoff=9.0, rfa=True, switch_dist=7.5)
    pots_c, F_c, fc, p, b = _run(par, g["pos"], zero, terms, prec="f64", algorithm="celllist", **kw)
    pots_a, F_a, fa, _, _ = _run(par, g["pos"], zero, terms, prec="f64", algorithm="allpairs", **kw)
    assert fc.stats(p)["algorithm"] == "celllist" and fa.stats(p)["algorithm"] == "allpairs"
    assert np.abs(F_c - F_a).max() < 1e-9
    pairs = orc.candidate_pairs(g["pos"], zero, 9.5, orc.exclusion_pairs(par))
    po, Fo, npairs = orc.compute(par, pos_tensor(g["pos"], 1, torch.float64), box_tensor(zero, 1, torch.float64),
                                 terms, pairs=pairs, **kw)
    assert np.abs(F_c - Fo.numpy()).max() < 1e-8
    for t in terms:
        assert abs(pots_c[0][t] - po[0][t]) < 1e-8 * max(1, abs(po[0][t]))
    assert fc.count_pairs(p, b) == npairs


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_water_box_celllist_vs_oracle(prec):
    """Synthetic TIP3P box (12^3 molecules = 5 184 atoms, L = 37.3 A): cell-list path vs all-pairs
    kernel vs oracle, incl. the in-cutoff pair count and list rebuilds after moving atoms."""
    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.parameters import Parameters

    dev = _dev()
    dt = PREC[prec]
    mol, pos, box = tip3p_box(12, seed=3)
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
    kw = dict(cutoff=9.0, rfa=True)
    p32 = pos_tensor(pos, 1, dt)
    pairs = orc.candidate_pairs(pos, box, 9.6, orc.exclusion_pairs(par))
    po, Fo, npairs = orc.compute(par, p32, box_tensor(box, 1, dt), terms, pairs=pairs, **kw)
    res = {}
    for algo in ("celllist", "allpairs"):
        f = Forces(par, terms=terms, algorithm=algo, **kw)
        p, b = p32.to(dev), box_tensor(box, 1, dt, dev)
        F = torch.zeros_like(p)
        pots = f.compute(p, b, F, returnDetails=True)
        assert f.stats(p)["algorithm"] == algo
        assert f.count_pairs(p, b) == npairs, algo
        err = (F.cpu() - Fo).abs().max().item()
        assert err < FTOL[prec], (algo, err)
        for t in terms:
            assert abs(pots[0][t] - po[0][t]) <= ERTOL[prec] * EFAC * max(1, abs(po[0][t])), (algo, t)
        res[algo] = F.cpu()
    assert (res["celllist"] - res["allpairs"]).abs().max() < FTOL[prec]

    # move atoms: small displacement (no rebuild needed), then a large one (device-side rebuild)
    f = Forces(par, terms=["lj", "electrostatics"], algorithm="celllist", **kw)
    p, b = p32.to(dev), box_tensor(box, 1, dt, dev)
    F = torch.zeros_like(p)
    f.compute(p, b, F)
    rebuilds0 = f.stats(p)["n_rebuilds"]
    rng = np.random.default_rng(1)
    for scale, expect_rebuild in ((0.1, False), (1.5, True)):  # 0.1 A: below the smallest per-atom half skin
        d = torch.tensor(rng.uniform(-1, 1, size=pos.shape) * scale / np.sqrt(3), dtype=dt)
        pm = (p32[0] + d)[None].contiguous()
        f.compute(pm.to(dev), b, F)
        _, Fm, _ = orc.compute(par, pm, box_tensor(box, 1, dt), ["lj", "electrostatics"],
                               pairs=orc.candidate_pairs(pm[0].double().numpy(), box, 9.6, orc.exclusion_pairs(par)),
                               **kw)
        # random displacements create close contacts with huge forces: compare relative to |F|
        rel = ((F.cpu() - Fm).abs() / (1.0 + Fm.abs())).max().item()
        assert rel < (1e-4 if prec == "f32" else 1e-10), rel
        assert (f.stats(p)["n_rebuilds"] > rebuilds0) == expect_rebuild
        rebuilds0 = f.stats(p)["n_rebuilds"]
    # atoms translated by whole box vectors: same minimum-image geometry up to the rounding of the shifted
    # coordinates, so the list stays valid; forces and the in-cutoff pair count must be the ORACLE's at the
    # shifted positions (the reference never wraps: forces.py:360-365 sees such offsets after long runs)
    shift = torch.tensor(rng.integers(-2, 3, size=pos.shape), dtype=dt) * torch.tensor(box, dtype=dt)
    ps = (pm[0] + shift)[None].contiguous()
    F2 = torch.zeros_like(F)
    f.compute(ps.to(dev), b, F2)
    _, Fs, ns = orc.compute(par, ps, box_tensor(box, 1, dt), ["lj", "electrostatics"],
                            pairs=orc.candidate_pairs(ps[0].double().numpy(), box, 9.6, orc.exclusion_pairs(par)), **kw)
    # (fp32: the shifted coordinates are rounded at |x| ~ 100 A, the close contacts of the random displacement
    # amplify value-arithmetic differences to ~2e-4 relative; a flipped cutoff decision would show as >= 5e-3)
    assert ((F2.cpu() - Fs).abs() / (1.0 + Fs.abs())).max().item() < (5e-4 if prec == "f32" else 1e-10)
    assert f.count_pairs(ps.to(dev), b) == ns


@pytest.mark.parametrize("case", ["water12-f32", "water12-f64", "c3-f32", "c3-f64"])
@pytest.mark.parametrize("reach", [1, 3])
def test_image_offsets_vs_oracle(case, reach):
    """Unwrapped coordinates (the reference's System/Integrator never wrap: integrator.py:61-64): every atom is
    shifted by its own integer box vector in [-reach, reach]^3, so `round(d/box)` of forces.py:360-365 takes
    values up to 2*reach + 1.  The lean kernels fuse `d - box*k` while k*box is exact (|k| <= 2: coordinate extent
    below 2.4 box edges, case reach = 1 with images {0, 1}) and round the product separately beyond (reach = 3);
    see extent_note in csrc/engine.h.  Bar: in-cutoff pair count == the oracle's at the SAME shifted tensors
    (generic kernel), forces within the parity bound (lean kernel: one flipped pair is 0.05 kcal/mol/A with
    reaction field, 25x the fp32 bound)."""
    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.parameters import Parameters

    dev = _dev()
    name, prec = case.split("-")
    dt = PREC[prec]
    mol, pos, box = tip3p_box(12 if name == "water12" else 32, seed=11)
    terms = ["lj", "electrostatics"]
    par = Parameters(water_forcefield(mol), mol, terms + ["bonds", "angles"], precision=dt)
    rng = np.random.default_rng(reach)
    lo = 0 if reach == 1 else -reach  # reach 1: images {0, 1} only (extent 2 boxes: the fused form must hold)
    k = rng.integers(lo, reach + 1, size=pos.shape)
    p = (torch.tensor(pos, dtype=dt) + torch.tensor(k, dtype=dt) * torch.tensor(box, dtype=dt))[None].contiguous()
    b = box_tensor(box, 1, dt)
    pairs = orc.candidate_pairs(p[0].double().numpy(), box, 9.5, orc.exclusion_pairs(par))
    po, Fo, npairs = orc.compute(par, p, b, terms, pairs=pairs, cutoff=9.0, rfa=True)
    f = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
    F = torch.zeros(1, mol.numAtoms, 3, dtype=dt, device=dev)
    pots = f.compute(p.to(dev), b.to(dev), F, returnDetails=True)
    assert f.stats(p.to(dev))["algorithm"] == "celllist"
    n_gpu = f.count_pairs(p.to(dev), b.to(dev))
    err = (F.cpu() - Fo).abs().max().item()
    print(f"{case} reach {reach}: P_cut = {npairs[0]}, GPU count {n_gpu[0]}, max|dF| = {err:.3e}")
    assert n_gpu == npairs
    assert err < FTOL[prec]
    for t in terms:
        assert abs(pots[0][t] - po[0][t]) <= ERTOL[prec] * EFAC * max(1, abs(po[0][t])), t
    # the same through a second evaluation (list reused, displacement-test kernel writes the records)
    F2 = torch.zeros_like(F)
    f.compute(p.to(dev), b.to(dev), F2)
    assert torch.equal(F, F2)


def test_replicas_independent_lists():
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.parameters import Parameters

    dev = _dev()
    mol, pos, box = tip3p_box(12, seed=5)
    par = Parameters(water_forcefield(mol), mol, ["lj", "electrostatics"], precision=torch.float32)
    f = Forces(par, terms=["lj", "electrostatics"], cutoff=9.0, rfa=True, algorithm="celllist")
    rng = np.random.default_rng(0)
    p = torch.tensor(np.stack([pos, pos + rng.normal(scale=0.3, size=pos.shape)]), dtype=torch.float32, device=dev)
    b = box_tensor(box, 2, torch.float32, dev)
    F = torch.zeros_like(p)
    e2 = f.compute(p, b, F)
    F1 = torch.zeros_like(p[1:2])
    e1 = f.compute(p[1:2].contiguous(), b[1:2], F1)
    assert abs(e2[1] - e1[0]) < 1e-4 * abs(e1[0])
    # (the 0.3 A noise makes close contacts: |F| up to 1e9, and a two-replica context picks its lanes per atom from the atoms
    # that share a launch — another summation order than the one-replica context's: compare relative to the largest force)
    assert (F[1:2] - F1).abs().max().item() < 2e-6 * max(1.0, F1.abs().max().item())
    assert abs(e2[0] - e2[1]) > 1.0  # replicas really differ


def test_errors_and_api_surface():
    from torchmd_amd.forces import Forces

    g = load("water291")
    par = GoldenParameters(g, torch.float32)
    with pytest.raises(RuntimeError):
        Forces(par)
    with pytest.raises(ValueError):
        Forces(par, terms=["nope"])
    with pytest.raises(RuntimeError):
        Forces(par, terms=["1-4"])
    f = Forces(par, terms=["LJ", "Electrostatics"], cutoff=7.3)
    assert f.energies == ["lj", "electrostatics"] and f.natoms == 291
    assert f.ava_idx.shape == (41904, 2)  # P_all of tests/water
    dev = _dev()
    p = pos_tensor(g["pos"], 1, torch.float32, dev)
    b = box_tensor(g["box"], 1, torch.float32, dev)
    with pytest.raises(RuntimeError):
        f.compute(p, b, torch.zeros_like(p), explicit_forces=False)  # pos does not require grad
    out = f.compute(p, b, torch.zeros_like(p))
    assert isinstance(out, list) and isinstance(out[0], float)
    out = f.compute(p, b, torch.zeros_like(p), toNumpy=False)
    assert torch.is_tensor(out) and out.shape == (1,) and out.dtype == torch.float32

    class Ext:  # the reference's plugin hook (forces.py:321-326)
        def calculate(self, pos, box):
            return torch.full((pos.shape[0],), 2.5, device=pos.device), torch.ones_like(pos)

    fe = Forces(par, terms=["lj"], cutoff=7.3, external=Ext())
    F0, F1 = torch.zeros_like(p), torch.zeros_like(p)
    d0 = Forces(par, terms=["lj"], cutoff=7.3).compute(p, b, F0, returnDetails=True)
    d1 = fe.compute(p, b, F1, returnDetails=True)
    assert d1[0]["external"] == 2.5 and abs(d1[0]["lj"] - d0[0]["lj"]) < 1e-6
    assert torch.allclose(F1 - F0, torch.ones_like(F0), atol=1e-5)


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_lj_box_vs_oracle(prec):
    """C5-shaped case at a size the oracle handles: 22^3 = 10 648 argon atoms at liquid density, LJ only,
    one atom type, cell-list path vs oracle (forces, energy, in-cutoff pair count)."""
    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import argon_forcefield, lj_box
    from torchmd_amd.forces import Forces
    from torchmd_amd.parameters import Parameters

    dev, dt = _dev(), PREC[prec]
    mol, pos, box = lj_box(22, seed=2)
    par = Parameters(argon_forcefield(mol), mol, ["lj"], precision=dt)
    p = pos_tensor(pos, 1, dt)
    pairs = orc.candidate_pairs(pos, box, 9.6, None)
    po, Fo, npairs = orc.compute(par, p, box_tensor(box, 1, dt), ["lj"], pairs=pairs, cutoff=9.0, switch_dist=7.5)
    f = Forces(par, terms=["lj"], cutoff=9.0, switch_dist=7.5)
    F = torch.zeros(1, mol.numAtoms, 3, dtype=dt, device=dev)
    pots = f.compute(p.to(dev), box_tensor(box, 1, dt, dev), F, returnDetails=True)
    assert f.stats(p.to(dev))["algorithm"] == "celllist"
    assert (F.cpu() - Fo).abs().max().item() < FTOL[prec]
    assert abs(pots[0]["lj"] - po[0]["lj"]) <= ERTOL[prec] * EFAC * abs(po[0]["lj"])
    assert f.count_pairs(p.to(dev), box_tensor(box, 1, dt, dev)) == npairs


@pytest.mark.timeout(900)
@pytest.mark.parametrize("switch", [None, 7.5], ids=["noswitch", "switch7.5"])
@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_c3_full_size_vs_oracle(prec, switch):
    """Config C3 at full size (98 304 atoms; fp32 = the bench's precision, fp64 = the lean fp64 kernel), all four terms
    of the bench (lj, electrostatics, bonds, angles): HIP cell-list path vs the oracle with a sparse candidate list;
    bar on the static leg: max |dF| <= 1e-4 kcal/mol/A in fp32 — the north star's target sentence ("forces within 1e-4
    kcal/mol/A" at the 100k-atom box; its general fp32 bar is 1e-2; observed 7e-5) —, <= 1e-8 in fp64 (north star 1e-4),
    identical in-cutoff pair count, sum(F) ~ 0, energies within EFAC x 2e-5 (1e-10) relative.
    `switch7.5`: the same with the LJ switching function from 7.5 A (the reference's production settings,
    tests/prod_alanine_dipeptide_amber/conf.yaml:8-9; forces.py:399-413, the explicit-force flavour with its extra 1/r):
    the SWITCH variants of the lean kernels, alone and with the MD step in the same launch.
    Second leg: the state the bench times — ~60 Langevin steps through Integrator with the default gates (per-atom
    and velocity-dependent skins, rebuild chains left out by the pacing host: asserted active) — then the AGED
    list's in-cutoff pair count and the run's forces against the oracle at the final positions."""
    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev, dt = _dev(), PREC[prec]
    mol, pos, box = tip3p_box(32, seed=0)
    nb = ["lj", "electrostatics"]
    terms = nb + ["bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
    excl = orc.exclusion_pairs(par)
    s = System(mol.numAtoms, 1, dt, dev)
    s.set_positions(pos[:, :, None])
    s.set_box(box)
    okw = dict(cutoff=9.0, rfa=True, switch_dist=switch)
    f = Forces(par, terms=terms, **okw)
    pots = f.compute(s.pos, s.box, s.forces, returnDetails=True)
    n_gpu = f.count_pairs(s.pos, s.box)
    assert f.stats(s.pos)["algorithm"] == "celllist"
    assert s.forces.sum(dim=1).abs().max().item() < (0.5 if prec == "f32" else 1e-8)  # Newton's third law (sum over 98k atoms)
    p = s.pos.detach().cpu()
    pairs = orc.candidate_pairs(p[0].double().numpy(), box, 9.5, excl)
    po, Fo, npairs = orc.compute(par, p, s.box.cpu(), terms, pairs=pairs, **okw)
    err = (s.forces.cpu() - Fo).abs().max().item()
    F_only = torch.zeros_like(s.pos)  # the forces-only variant of the launch (what an MD step runs)
    f._evaluate(s.pos, s.box, F_only, False, True)
    err_only = (F_only.cpu() - Fo).abs().max().item()
    print(f"C3 full size {prec} switch {switch}: P_cut = {npairs[0]}, max|dF| = {err:.3e} (forces-only launch {err_only:.3e}), "
          + ", ".join(f"E_{t} = {pots[0][t]:.2f}" for t in terms))
    assert n_gpu == npairs
    assert max(err, err_only) < C3_STATIC_FTOL[prec]
    for t in terms:
        assert abs(pots[0][t] - po[0][t]) <= ERTOL[prec] * EFAC * max(1.0, abs(po[0][t])), (t, pots[0][t], po[0][t])

    # ---- the bench state: an MD run with every default gate, then the aged list against the oracle
    torch.manual_seed(1)
    s.set_velocities(maxwell_boltzmann(par.masses, 300.0, 1))
    integ = Integrator(s, f, 1.0, dev, gamma=0.1, T=300.0)
    integ.step(40)
    r0 = f.stats(s.pos)["n_rebuilds"]
    integ.step(21)
    st = f.stats(s.pos)
    assert st["chains_skipped"] > 5 and st["overflow"] == 0, st  # chain skipping is active at this list size
    assert st["n_rebuilds"] > r0  # ... and device-side rebuilds (velocity-dependent skins) happened meanwhile
    n_gpu = f.count_pairs(s.pos, s.box)  # through the run's own (aged) list: displacement test only
    aged = f.stats(s.pos)["n_rebuilds"] == st["n_rebuilds"]
    p = s.pos.detach().cpu()
    pairs = orc.candidate_pairs(p[0].double().numpy(), box, 9.3, excl)
    _, Fo, npairs = orc.compute(par, p, s.box.cpu(), terms, pairs=pairs, **okw)
    err = (s.forces.cpu() - Fo).abs().max().item()
    print(f"C3 {prec} switch {switch} after 61 MD steps: P_cut = {npairs[0]}, GPU count {n_gpu[0]} (list aged: {aged}), max|dF| = {err:.3e}, "
          f"chains skipped {st['chains_skipped']}, rebuilds {st['n_rebuilds']}")
    assert n_gpu == npairs
    assert err < FTOL_HOT[prec]  # (the lattice start after 61 steps: see FTOL_HOT)


def test_lj_million_atoms_properties():
    """Config C5 size on one GPU (10^6 argon atoms, L = 360.8 A): size-independent checks — sum of forces
    vanishes, the result is invariant under a rigid translation by a box vector, energy is extensive
    w.r.t. the 10 648-atom box at the same density, and the list statistics are sane."""
    from torchmd_amd.builders import argon_forcefield, lj_box
    from torchmd_amd.forces import Forces
    from torchmd_amd.parameters import Parameters

    dev = _dev()
    mol, pos, box = lj_box(100, seed=4)
    par = Parameters(argon_forcefield(mol), mol, ["lj"], precision=torch.float32)
    f = Forces(par, terms=["lj"], cutoff=9.0)
    p = torch.tensor(pos, dtype=torch.float32, device=dev)[None].contiguous()
    b = box_tensor(box, 1, torch.float32, dev)
    F = torch.zeros_like(p)
    e = f.compute(p, b, F)[0]
    st = f.stats(p)
    assert st["algorithm"] == "celllist" and st["overflow"] == 0
    assert abs(st["list_entries"] / 1e6 - 0.0213 * 4.18879 * 1000) < 10  # ~89 neighbours within 10 A (lattice: 82)
    assert F.sum(dim=1).abs().max().item() < 0.5
    shift = torch.tensor(box, dtype=torch.float32, device=dev) * torch.tensor([1.0, -1.0, 2.0], device=dev)
    F2 = torch.zeros_like(p)
    e2 = f.compute((p + shift).contiguous(), b, F2)[0]
    assert abs(e2 - e) < 2e-4 * abs(e)
    assert ((F2 - F).abs() / (1 + F.abs())).max().item() < 2e-2
    pcut = f.count_pairs(p, b)[0]
    assert abs(pcut / 1e6 - 0.5 * 0.0213 * 4.18879 * 729) < 6  # ~32.5 pairs/atom within 9 A (lattice: 36.8)
    assert -1.6 < e / 1e6 < -0.6  # cohesive LJ energy per atom (eps = 0.238 kcal/mol, jittered lattice: -1.01)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("nside", [100, 104])
def test_lj_million_atoms_vs_oracle(nside):
    """Config C5 size against the ORACLE (reference arithmetic: forces.py:260-319, 360-372, 381-415; argon parameters of
    /root/reference/tests/argon/argon_forcefield.yaml:8-18), on one GPU, after a few MD steps so that the state is the
    one the bench times (device-side rebuilds, the LJ-only loop of the lean kernel at 4 lanes per atom; 104^3 =
    1 124 864 atoms > 2^20: every iteration in the checked loop with the masked table offset).  The oracle cannot
    evaluate 3.3e7 pairs' worth of [P, 3] temporaries in seconds, so: (i) 20 000 random atoms — every candidate pair
    that touches one of them (cKDTree, 9.5 A) goes through the oracle, and the forces on the picked atoms (complete:
    all of their pairs are there) must equal the GPU's within the fp32 bar; (ii) 10^6-atom box only: the in-cutoff
    pair COUNT of the whole box by the oracle's decision arithmetic (pair_geometry + `dist <= cutoff`, fp32, in chunks)
    must equal the GPU's count exactly."""
    from scipy.spatial import cKDTree

    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import argon_forcefield, lj_box
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev, dt = _dev(), torch.float32
    mol, pos, box = lj_box(nside, seed=4)
    n = mol.numAtoms
    par = Parameters(argon_forcefield(mol), mol, ["lj"], precision=dt)
    s = System(n, 1, dt, dev)
    s.set_positions(pos[:, :, None])
    s.set_box(box)
    torch.manual_seed(2)
    s.set_velocities(maxwell_boltzmann(par.masses, 85.0, 1))
    f = Forces(par, terms=["lj"], cutoff=9.0)
    f.compute(s.pos, s.box, s.forces)
    Integrator(s, f, 1.0, dev, gamma=1.0, T=85.0).step(12)  # forces of the final positions are in s.forces
    st = f.stats(s.pos)
    assert st["algorithm"] == "celllist" and st["overflow"] == 0 and st["steps_in_pair_launch"] >= 10, st
    p = s.pos.detach().cpu()
    Fg = s.forces.detach().cpu()
    p64 = p[0].double().numpy()
    b = np.asarray(box, dtype=np.float64)
    w = p64 - np.floor(p64 / b) * b
    w = np.where(w >= b, w - b, w)
    tree = cKDTree(w, boxsize=b)
    pick = np.sort(np.random.default_rng(7).choice(n, 20000, replace=False))
    nb = tree.query_ball_point(w[pick], 9.5, workers=-1)
    i = np.repeat(pick, [len(x) for x in nb])
    j = np.concatenate([np.asarray(x, dtype=np.int64) for x in nb])
    keep = i != j
    lo, hi = np.minimum(i[keep], j[keep]), np.maximum(i[keep], j[keep])
    key = np.unique(lo * np.int64(n) + hi)  # pairs between two picked atoms were found twice; sorted (i asc, j asc)
    pairs = np.stack([key // n, key % n], axis=1)
    _, Fo, nin = orc.compute(par, p, s.box.cpu(), ["lj"], pairs=pairs, cutoff=9.0)
    err = (Fg[0, pick] - Fo[0, pick]).abs().max().item()
    print(f"{n} argon atoms after 12 MD steps: {len(pairs)} candidate pairs touch the 20 000 picked atoms, {nin[0]} inside the "
          f"cutoff; max|dF| on the picked atoms = {err:.3e} (max|F| = {Fg[0, pick].abs().max().item():.2f})")
    assert nin[0] > 600000
    assert err < FTOL["f32"]
    if nside == 100:
        n_gpu = f.count_pairs(s.pos, s.box)[0]
        allp = tree.query_pairs(9.05, output_type="ndarray")
        pt, bt = p[0], s.box.cpu()[0][torch.eye(3).bool()]
        total = 0
        for c in range(0, len(allp), 4_000_000):
            idx = torch.as_tensor(allp[c: c + 4_000_000].astype(np.int64))
            d, _, _ = orc.pair_geometry(pt, idx, bt)
            total += int((d <= 9.0).sum().item())
        print(f"P_cut of the whole box: oracle {total}, GPU {n_gpu}")
        assert n_gpu == total
    f.close()


def test_auto_falls_back_to_allpairs_in_small_boxes():
    """N >= 2048 with a cutoff selects the cell-list path, but with cutoff 10.5 A a 27.9 A box holds fewer than 5 cells of
    (cutoff + skin)/2 per edge: algorithm='auto' must switch to the all-pairs kernel by itself (also inside
    the fused Integrator loop) and still match the oracle."""
    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev = _dev()
    mol, pos, box = tip3p_box(9, seed=8)  # 2 187 atoms, L = 27.95 A
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=torch.float64)
    f = Forces(par, terms=terms, cutoff=10.5, rfa=True)
    s = System(mol.numAtoms, 1, torch.float64, dev)
    s.set_positions(pos[:, :, None])
    s.set_box(box)
    pots = f.compute(s.pos, s.box, s.forces, returnDetails=True)
    assert f.stats(s.pos)["algorithm"] == "allpairs"
    po, Fo, npairs = orc.compute(par, s.pos.cpu(), s.box.cpu(), terms, cutoff=10.5, rfa=True,
                                 pairs=orc.candidate_pairs(pos, box, 11.0, orc.exclusion_pairs(par)))
    assert (s.forces.cpu() - Fo).abs().max().item() < 1e-8
    assert f.count_pairs(s.pos, s.box) == npairs
    with pytest.raises(RuntimeError):
        Forces(par, terms=terms, cutoff=10.5, rfa=True, algorithm="celllist").compute(s.pos, s.box, s.forces)
    # fresh object, first use from inside the fused loop
    f2 = Forces(par, terms=terms, cutoff=10.5, rfa=True)
    s.forces.copy_(Fo.to(dev))
    ek, ep, T = Integrator(s, f2, 0.5, dev).step(3)
    assert f2.stats(s.pos)["algorithm"] == "allpairs" and np.isfinite(ep).all()


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize(
    "terms,kw",
    [
        (["lj"], dict(cutoff=9.0)),  # lean fp32 kernel, LJ only
        (["electrostatics"], dict(cutoff=9.0)),  # lean fp32 kernel, plain Coulomb
        (["electrostatics"], dict(cutoff=9.0, rfa=True)),  # lean fp32 kernel, reaction field only
        (["lj", "electrostatics"], dict(cutoff=9.0)),  # lean fp32 kernel, LJ + plain Coulomb
        (["lj", "electrostatics"], dict(cutoff=9.0, rfa=True, switch_dist=7.5)),  # lean kernels, SWITCH variant
        (["repulsion"], dict(cutoff=9.0)),
        (["repulsioncg", "electrostatics"], dict(cutoff=8.0, rfa=True, solventDielectric=60.0)),
    ],
)
def test_celllist_term_variants_vs_oracle(prec, terms, kw):
    """Every term combination on the cell-list path (the variants of the lean fp32 / fp64 kernels; the repulsion terms
    take the generic kernel) against the oracle on the 5 184-atom water box: forces, per-term energies, in-cutoff pair count."""
    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.parameters import Parameters

    dev, dt = _dev(), PREC[prec]
    mol, pos, box = tip3p_box(12, seed=13)
    par = Parameters(water_forcefield(mol), mol, ["lj", "electrostatics", "bonds", "angles"], precision=dt)
    p = pos_tensor(pos, 1, dt)
    okw = dict(kw)
    pairs = orc.candidate_pairs(pos, box, kw["cutoff"] + 0.6, orc.exclusion_pairs(par))
    po, Fo, npairs = orc.compute(par, p, box_tensor(box, 1, dt), terms, pairs=pairs, **okw)
    f = Forces(par, terms=terms, algorithm="celllist", **kw)
    pd, bd = p.to(dev), box_tensor(box, 1, dt, dev)
    F = torch.zeros_like(pd)
    f.compute(pd, bd, F)  # forces-only path first (this is what the lean fp32 kernel serves) ...
    F_noenergy = torch.zeros_like(pd)
    f._evaluate(pd, bd, F_noenergy, False, True)
    pots = f.compute(pd, bd, F, returnDetails=True)  # ... then with energies (the ENERGY variant)
    scale = 1.0 + Fo.abs()
    # fp32: ~400 partial forces of up to a few hundred kcal/mol/A per atom summed in list order, the
    # reference sums them in pair order (bar of the north star: 1e-2 absolute)
    assert ((F.cpu() - Fo).abs() / scale).max().item() < (1e-10 if prec == "f64" else 6e-5)
    assert ((F_noenergy.cpu() - Fo).abs() / scale).max().item() < (1e-10 if prec == "f64" else 6e-5)
    for t in terms:
        assert abs(pots[0][t] - po[0][t]) <= ERTOL[prec] * EFAC * max(1, abs(po[0][t])), t
    assert f.count_pairs(pd, bd) == npairs


@pytest.mark.parametrize("lpa", [1, 2, 4, 16, 32, 64])
def test_every_lanes_per_atom_variant(lpa, monkeypatch):
    """The list layout / pair kernels are templated on LPA (lanes per atom); the heuristic picks 4-64
    depending on N.  Force every instantiation on the same box (lean fp32 kernel and fp64 generic)."""
    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.parameters import Parameters

    monkeypatch.setenv("TMDHIP_LPA", str(lpa))
    dev = _dev()
    mol, pos, box = tip3p_box(12, seed=17)
    terms = ["lj", "electrostatics"]
    for prec in ("f32", "f64"):
        dt = PREC[prec]
        par = Parameters(water_forcefield(mol), mol, terms + ["bonds", "angles"], precision=dt)
        p = pos_tensor(pos, 1, dt)
        pairs = orc.candidate_pairs(pos, box, 9.6, orc.exclusion_pairs(par))
        po, Fo, npairs = orc.compute(par, p, box_tensor(box, 1, dt), terms, pairs=pairs, cutoff=9.0, rfa=True)
        f = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
        pd, bd = p.to(dev), box_tensor(box, 1, dt, dev)
        F = torch.zeros_like(pd)
        f._evaluate(pd, bd, F, False, True)
        # (one lane summing ~440 fp32 terms sequentially at LPA=1 rounds a little more than 8 lanes x 55)
        assert ((F.cpu() - Fo).abs() / (1 + Fo.abs())).max().item() < (1e-10 if prec == "f64" else 1e-4), (lpa, prec)
        e = f.compute(pd, bd, F, returnDetails=True)
        assert abs(e[0]["lj"] - po[0]["lj"]) <= ERTOL[prec] * EFAC * abs(po[0]["lj"])
        assert f.count_pairs(pd, bd) == npairs


@pytest.mark.parametrize("case", ["water-4", "water-8", "water-32", "lj-4", "thrombin-open"])
def test_padded_list_rows_are_bit_identical(case, monkeypatch):
    """Padded rows: the padding slots of every wave group point at a dummy record out of reach (written by the pair waves
    on their first launch after a list build) and the
    lean fp32 kernel runs every group unchecked.  A padding slot contributes exactly 0 and the real entries see the
    same arithmetic in the same order, so forces and energies must equal the unpadded list's (TMDHIP_PAD_ROWS=0) bit
    for bit — plain evaluations and along an MD trajectory (fused pair + step launches, device-side rebuilds).  Water:
    periodic, reaction field; the LJ box: the plain loop of the LJ-only variants; thrombin: open boundaries (dummies
    10^6 A out) and the switching-free protein terms."""
    from torchmd_amd.builders import argon_forcefield, lj_box, tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev, dt = _dev(), torch.float32
    kind, lanes = case.split("-")
    if kind == "thrombin":
        g = load("thrombin")
        par = GoldenParameters(g, dt)
        pos, box = np.asarray(g["pos"], dtype=np.float64), np.zeros(3)
        terms, kw, lanes = ALL_TERMS, dict(cutoff=9.0), "64"
    elif kind == "water":
        mol, pos, box = tip3p_box(14, seed=23)
        terms = ["lj", "electrostatics", "bonds", "angles"]
        par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
        kw = dict(cutoff=9.0, rfa=True)
    else:
        mol, pos, box = lj_box(22, seed=5)
        terms = ["lj"]
        par = Parameters(argon_forcefield(mol), mol, terms, precision=dt)
        kw = dict(cutoff=9.0)
    monkeypatch.setenv("TMDHIP_LPA", lanes)
    torch.manual_seed(5)
    vel0 = maxwell_boltzmann(par.masses, 300.0, 1)

    def run(pad):
        monkeypatch.setenv("TMDHIP_PAD_ROWS", pad)
        s = System(pos.shape[0], 1, dt, dev)
        s.set_positions(pos[:, :, None])
        s.set_box(box)
        s.set_velocities(vel0)
        f = Forces(par, terms=terms, algorithm="celllist", **kw)
        e0 = f.compute(s.pos, s.box, s.forces, returnDetails=True)
        F0 = s.forces.clone().cpu()
        torch.manual_seed(9)
        res = Integrator(s, f, 1.0, dev, gamma=1.0, T=300.0).step(60)
        st = f.stats(s.pos)
        out = (e0[0], F0, s.pos.cpu(), s.forces.cpu(), res, st["n_rebuilds"], st["list_entries"], f.count_pairs(s.pos, s.box))
        f.close()
        return out

    a, b = run("1"), run("0")
    # (energies are fp64 sums that the waves add into scratch rows with atomics: the order of the adds, hence the last bit of a
    # sum, is free from run to run — seen once in round 6: ...853525 against ...853524; forces and positions are not summed that way)
    for t in a[0]:
        assert abs(a[0][t] - b[0][t]) <= 1e-12 * max(1.0, abs(b[0][t])), (t, a[0], b[0])
    assert torch.isfinite(a[2]).all()
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    for k, (x, y) in enumerate(zip(a[4], b[4])):  # (Ekin, Epot, T)
        assert np.allclose(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64), rtol=1e-12 if k == 1 else 2e-7, atol=0)
    assert a[5] == b[5] and a[5] >= 2 and a[6] == b[6] and a[7] == b[7]


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_two_launch_binning_and_its_cell_overflow_fallback(prec, monkeypatch):
    """The cell binning is two launches (member arrays of 64 atoms per cell + scan and placement in one kernel) instead of
    four; same cell order, hence identical lists: forces and energies equal the four-launch binning's (TMDHIP_BIN2=0) bit
    for bit.  A cell that receives more than 64 atoms — here 150 weakly charged atoms inside one 2-A sphere of a larger
    box, electrostatics only so that nothing overflows numerically — raises F_CELLCAP: the replica falls back to the four
    launches and the result still equals the oracle's (without the fallback 86 atoms would be missing from the list)."""
    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.parameters import Parameters

    dev, dt = _dev(), PREC[prec]
    mol, pos, box = tip3p_box(12, seed=31)
    terms = ["lj", "electrostatics"]
    par = Parameters(water_forcefield(mol), mol, terms + ["bonds", "angles"], precision=dt)
    pd, bd = pos_tensor(pos, 1, dt, dev), box_tensor(box, 1, dt, dev)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TMDHIP_BIN2", mode)
        f = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
        F = torch.zeros_like(pd)
        e = f.compute(pd, bd, F, returnDetails=True)
        out[mode] = (e[0], F.clone().cpu(), f.count_pairs(pd, bd))
        f.close()
    assert torch.equal(out["1"][1], out["0"][1]) and out["1"][2] == out["0"][2]
    for t in terms:  # (fp64 energies are folded with atomics: the order of the terms shows in the last bits)
        assert abs(out["1"][0][t] - out["0"][0][t]) <= (0.0 if prec == "f32" else 1e-12 * abs(out["0"][0][t])), t
    # overflow: a dense cluster
    monkeypatch.setenv("TMDHIP_BIN2", "1")
    rng = np.random.default_rng(7)
    n = 4000
    L = 60.0
    p = rng.uniform(0, L, size=(n, 3))
    v = rng.normal(size=(150, 3))
    p[:150] = 30.0 + 2.0 * rng.uniform(0, 1, size=(150, 1)) ** (1 / 3) * v / np.linalg.norm(v, axis=1, keepdims=True)
    from torchmd_amd.builders import Topology
    from torchmd_amd.forcefields.ff_yaml import YamlForceField

    ff = {"atomtypes": ["X"], "lj": {"X": {"sigma": 3.0, "epsilon": 0.1}}, "electrostatics": {"X": {"charge": 0.0}}, "masses": {"X": 10.0}}
    q = rng.choice([-0.01, 0.01], size=n).astype(np.float32)
    m2 = Topology(atomtype=np.full(n, "X", dtype=object), charge=q, masses=np.full(n, 10.0, dtype=np.float32))
    par2 = Parameters(YamlForceField(m2, ff), m2, ["electrostatics"], precision=dt)
    b3 = np.array([L, L, L])
    pt = pos_tensor(p, 1, dt)
    pairs = orc.candidate_pairs(p, b3, 9.6, orc.exclusion_pairs(par2))
    po, Fo, npairs = orc.compute(par2, pt, box_tensor(b3, 1, dt), ["electrostatics"], pairs=pairs, cutoff=9.0, rfa=True)
    f = Forces(par2, terms=["electrostatics"], cutoff=9.0, rfa=True, algorithm="celllist")
    pd2, bd2 = pt.to(dev), box_tensor(b3, 1, dt, dev)
    F = torch.zeros_like(pd2)
    e = f.compute(pd2, bd2, F, returnDetails=True)
    assert f.count_pairs(pd2, bd2) == npairs
    tol = 1e-8 if prec == "f64" else 2e-3
    assert (F.cpu() - Fo).abs().max().item() <= tol * max(1.0, Fo.abs().max().item())
    assert abs(e[0]["electrostatics"] - po[0]["electrostatics"]) <= ERTOL[prec] * EFAC * max(1.0, abs(po[0]["electrostatics"]))


def test_streamed_list_reads_are_bit_identical(monkeypatch):
    """Lists that cannot live in the Infinity Cache are read with the non-temporal hint (kLmStream: a run-time choice
    between two load instructions in the lean fp32 kernel; by size, or TMDHIP_LIST_STREAM=0 / 1).  A cache hint must not
    change a single bit: forces, energies and a short fused trajectory with the hint forced on equal those with it off."""
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev, dt = _dev(), torch.float32
    mol, pos, box = tip3p_box(14, seed=2)
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
    monkeypatch.setenv("TMDHIP_LPA", "8")
    torch.manual_seed(1)
    vel0 = maxwell_boltzmann(par.masses, 300.0, 1)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("TMDHIP_LIST_STREAM", mode)
        s = System(mol.numAtoms, 1, dt, dev)
        s.set_positions(pos[:, :, None])
        s.set_box(box)
        s.set_velocities(vel0)
        f = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
        e0 = f.compute(s.pos, s.box, s.forces, returnDetails=True)
        F0 = s.forces.clone().cpu()
        torch.manual_seed(9)
        res = Integrator(s, f, 1.0, dev, gamma=1.0, T=300.0).step(30)
        out[mode] = (e0[0], F0, s.pos.cpu(), s.forces.cpu(), res)
        f.close()
    a, b = out["1"], out["0"]
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    # (energies: fp64 sums through atomics, free in their last bit from run to run — see test_padded_list_rows_are_bit_identical)
    for t in a[0]:
        assert abs(a[0][t] - b[0][t]) <= 1e-12 * max(1.0, abs(b[0][t])), t
    for k, (x, y) in enumerate(zip(a[4], b[4])):
        assert np.allclose(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64), rtol=1e-12 if k == 1 else 2e-7, atol=0)


def test_side_stream_equals_the_default_stream():
    """Everything is enqueued on the caller's stream (`torch.cuda.current_stream`).  A side stream of PyTorch is created
    non-blocking: it is NOT ordered behind the null stream, through which the library's set-up uploads go (parameters,
    exclusions, skins, bonded tables, zeroed flags).  Those uploads must be complete when the set-up calls return: a
    context created and used at once on a side stream gives the default stream's results bit for bit — forces, energies,
    a fused Langevin trajectory (cell-list path with per-atom skins and bonded terms) and the all-pairs path."""
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.integrator import Integrator, maxwell_boltzmann
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev, dt = _dev(), torch.float32
    mol, pos, box = tip3p_box(14, seed=2)
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
    torch.manual_seed(1)
    vel0 = maxwell_boltzmann(par.masses, 300.0, 1)
    g = load("ala2")
    gpar = GoldenParameters(g, torch.float64)

    def run():
        s = System(mol.numAtoms, 1, dt, dev)
        s.set_positions(pos[:, :, None])
        s.set_box(box)
        s.set_velocities(vel0)
        f = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
        e0 = f.compute(s.pos, s.box, s.forces, returnDetails=True)
        F0 = s.forces.clone()
        torch.manual_seed(9)  # (the integrator draws its noise key from torch's generator)
        res = Integrator(s, f, 1.0, dev, gamma=1.0, T=300.0).step(30)
        pots, F, *_ = _run(gpar, g["pos"], g["box"], ALL_TERMS, prec="f64", cutoff=9.0, switch_dist=7.5, rfa=True)
        torch.cuda.current_stream(dev).synchronize()
        out = (e0[0], F0.cpu(), s.pos.cpu(), s.forces.cpu(), res, pots[0], np.asarray(F))
        f.close()
        return out

    a = run()
    side = torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        b = run()
    torch.cuda.synchronize()
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    # (energies are folded with fp64 atomics, the all-pairs forces too: the order of the additions is not fixed)
    for k in a[0]:
        assert abs(a[0][k] - b[0][k]) <= 1e-12 * max(1.0, abs(a[0][k])), k
    for x, y in zip(a[4], b[4]):
        assert np.allclose(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64), rtol=1e-12, atol=0)
    for k in a[5]:
        assert abs(a[5][k] - b[5][k]) <= 1e-12 * max(1.0, abs(a[5][k])), k
    assert np.abs(a[6] - b[6]).max() < 1e-9


def test_box_change_and_capacity_growth():
    """(a) changing the box between calls re-plans the cell grid; (b) a denser configuration makes a
    device-side rebuild overflow the list capacity: tmdhip_check reports it, the capacity grows and the
    evaluation is repeated transparently."""
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.parameters import Parameters

    dev, dt = _dev(), torch.float64
    mol, pos, box = tip3p_box(12, seed=19)
    terms = ["lj", "electrostatics"]
    par = Parameters(water_forcefield(mol), mol, terms + ["bonds", "angles"], precision=dt)

    def fresh(p, b):
        F = torch.zeros_like(p)
        e = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist").compute(p, b, F)
        return e[0], F

    f = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
    p = pos_tensor(pos, 1, dt, dev)
    b = box_tensor(box, 1, dt, dev)
    F = torch.zeros_like(p)
    f.compute(p, b, F)
    ncell0 = f.stats(p)["ncell"]
    # (a) isotropic expansion by 30 %: new box tensor values -> new grid, same object
    p2, b2 = (p * 1.3).contiguous(), box_tensor(box * 1.3, 1, dt, dev)
    e2 = f.compute(p2, b2, F)[0]
    er, Fr = fresh(p2, b2)
    assert f.stats(p)["ncell"] != ncell0
    assert abs(e2 - er) < 1e-9 * abs(er) and (F - Fr).abs().max().item() < 1e-9
    # (b) open boundaries: compress the cluster so that neighbour counts exceed the capacity
    zero = box_tensor(np.zeros(3), 1, dt, dev)
    g = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
    g.compute(p, zero, F)
    cap0 = g.stats(p)["max_neighbours"]
    p3 = (p * 0.7).contiguous()  # density x 2.9 (atoms overlap: forces are huge but well defined)
    e3 = g.compute(p3, zero, F)[0]
    assert g.stats(p)["max_neighbours"] > cap0 and g.stats(p)["overflow"] == 0
    er, Fr = fresh(p3, zero)
    assert abs(e3 - er) < 1e-9 * abs(er)
    assert ((F - Fr).abs() / (1 + Fr.abs())).max().item() < 1e-9


@pytest.mark.parametrize("ntypes", [24, 32, 40])
def test_many_lj_classes_on_the_list_path(ntypes):
    """The lean fp32 list kernel keeps the LJ class of atom j in 5 bits of the list entry (<= 32 classes, LDS
    table rows of 32 x 8 B); more classes leave the field empty and the generic kernel reads the type array.
    Water box with the atoms re-labelled into `ntypes` artificial classes (random sigma/epsilon), LJ +
    reaction field, vs the oracle: forces, energies and the in-cutoff pair count."""
    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.parameters import Parameters

    dev, dt = _dev(), torch.float32
    mol, pos, box = tip3p_box(10, seed=2)  # 3 000 atoms
    terms = ["lj", "electrostatics"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
    rng = np.random.default_rng(ntypes)
    types = rng.integers(0, ntypes, size=mol.numAtoms)
    types[:ntypes] = np.arange(ntypes)  # every class occurs
    par.mapped_atom_types = torch.tensor(types, dtype=torch.int64)
    sig = rng.uniform(0.6, 1.7, size=ntypes)  # small: the re-labelled hydrogens sit 1.5-2 A from other molecules
    eps = rng.uniform(0.02, 0.2, size=ntypes)
    par.nonbonded_params = {"idx": [], "map": torch.tensor(np.stack([np.arange(mol.numAtoms), types], axis=1)),
                            "params": torch.tensor(np.stack([sig, eps], axis=1), dtype=par.nonbonded_params["params"].dtype)}
    kw = dict(cutoff=9.0, rfa=True)
    p32 = pos_tensor(pos, 1, dt)
    pairs = orc.candidate_pairs(pos, box, 9.6, orc.exclusion_pairs(par))
    po, Fo, npairs = orc.compute(par, p32, box_tensor(box, 1, dt), terms, pairs=pairs, **kw)
    f = Forces(par, terms=terms, algorithm="celllist", **kw)
    p, b = p32.to(dev), box_tensor(box, 1, dt, dev)
    F = torch.zeros_like(p)
    f._evaluate(p, b, F, False, True)  # forces only: the hot variant
    # the artificial classes put hydrogens with sizeable radii next to other molecules: forces reach 1e3-1e4,
    # so the fp32 comparison is relative to |F|
    scale = 1.0 + Fo.abs()
    assert ((F.cpu() - Fo).abs() / scale).max().item() < 2.5e-4
    pots = f.compute(p, b, F, returnDetails=True)
    assert ((F.cpu() - Fo).abs() / scale).max().item() < 2.5e-4
    for t in terms:
        assert abs(pots[0][t] - po[0][t]) <= ERTOL["f32"] * EFAC * max(1, abs(po[0][t])), t
    assert f.count_pairs(p, b) == npairs


def test_thrombin_fp32_open_boundaries_lean_kernel():
    """lean fp32 kernel on a non-periodic system with 15 atom types (LDS table, clamp-to-grid cells)."""
    g = load("thrombin")
    par = GoldenParameters(g, torch.float32)
    zero = np.zeros(3)
    terms = ["electrostatics", "lj"]
    kw = dict(cutoff=9.0, rfa=True)
    _, F_c, fc, p, b = _run(par, g["pos"], zero, terms, prec="f32", algorithm="celllist", **kw)
    _, F_a, fa, _, _ = _run(par, g["pos"], zero, terms, prec="f32", algorithm="allpairs", **kw)
    Fn = torch.zeros_like(p)
    fc._evaluate(p, b, Fn, False, True)  # forces only -> lean fp32 kernel
    print(f"thrombin fp32: cell list vs all pairs {np.abs(F_c - F_a).max():.3e}, lean kernel vs all pairs "
          f"{np.abs(Fn.cpu().numpy() - F_a).max():.3e} (max|F| {np.abs(F_a).max():.1f})")
    assert np.abs(F_c - F_a).max() < 2e-3 and np.abs(Fn.cpu().numpy() - F_a).max() < 2e-3
    assert fc.count_pairs(p, b) == fa.count_pairs(p, b)


@pytest.mark.parametrize("which", ["water291", "ala2"])
def test_replica_batch_matches_single_replica_calls(which):
    """All-pairs systems serve every replica with one launch per kernel (grid.z / grid.y = replica):
    R = 3 replicas with different coordinates AND different boxes must reproduce three independent
    single-replica evaluations (energies per term and forces), for the atom-centric (water) and the
    entry-parallel (alanine dipeptide) bonded paths."""
    from torchmd_amd.forces import Forces

    g = load(which)
    dev = _dev()
    par = GoldenParameters(g, torch.float64)
    terms = ["bonds", "angles", "electrostatics", "lj"] if which == "water291" else ALL_TERMS
    kw = dict(cutoff=7.3, rfa=True) if which == "water291" else dict(cutoff=9.0, rfa=True, switch_dist=7.5)
    rng = np.random.default_rng(5)
    pos0 = np.asarray(g["pos"], dtype=np.float64).reshape(-1, 3)
    box0 = np.asarray(g["box"], dtype=np.float64).reshape(-1)[:3]
    R = 3
    pos = np.stack([pos0 + 0.05 * r * rng.standard_normal(pos0.shape) for r in range(R)])
    boxes = np.stack([box0 * (1.0 + 0.01 * r) for r in range(R)])
    p = torch.tensor(pos, dtype=torch.float64, device=dev)
    b = torch.zeros(R, 3, 3, dtype=torch.float64, device=dev)
    for r in range(R):
        b[r] = torch.diag(torch.tensor(boxes[r], dtype=torch.float64))
    fb = Forces(par, terms=terms, **kw)
    Fb = torch.zeros_like(p)
    pots_b = fb.compute(p, b, Fb, returnDetails=True)
    assert fb.stats(p)["algorithm"] == "allpairs"
    for r in range(R):
        f1 = Forces(par, terms=terms, **kw)
        F1 = torch.zeros(1, p.shape[1], 3, dtype=torch.float64, device=dev)
        pots_1 = f1.compute(p[r : r + 1].contiguous(), b[r : r + 1].contiguous(), F1, returnDetails=True)
        for t in pots_1[0]:
            assert abs(pots_b[r][t] - pots_1[0][t]) <= 1e-9 * max(1.0, abs(pots_1[0][t])), (r, t)
        assert (Fb[r] - F1[0]).abs().max().item() <= 1e-9 * max(1.0, F1.abs().max().item()), r
    # the three replicas really differ
    assert abs(pots_b[0]["lj"] - pots_b[2]["lj"]) > 1e-3


@pytest.mark.parametrize("mode,lpa", [("reference", 0), ("exact", 0), ("reference", 4), ("reference", 16), ("reference", 32), ("reference", 64)])
def test_lean_kernel_switching_variants(mode, lpa, monkeypatch):
    """(`lpa` > 0: the same with the lanes per atom pinned — every LPA instantiation of the SWITCH and SWITCH + ENERGY variants, which
    evaluate the entries of a list word one at a time since round 6 — instead of the launcher's choice.)
    LJ switching in the lean list kernels (both force flavours, with and without energies) on the 5 184-atom water
    box: fp32 against fp64 (LJ only: the switched LJ force vanishes at the cutoff, so the handful of pairs whose
    cutoff decision differs between fp32 and fp64 coordinates does not matter) and BOTH against the oracle on the same
    tensors — `reference` = the oracle's explicit forces (forces.py:399-413 with the extra 1/r of 410-412), `exact` = its
    autograd forces (explicit_forces=False, forces.py:328-336), LJ only and LJ + reaction field, in-cutoff pair count
    equal."""
    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.parameters import Parameters

    if lpa:
        monkeypatch.setenv("TMDHIP_LPA", str(lpa))
    dev = _dev()
    mol, pos, box = tip3p_box(12, seed=17)
    terms = ["lj"]
    out = {}
    for prec in ("f64", "f32"):
        dt = PREC[prec]
        par = Parameters(water_forcefield(mol), mol, ["lj", "electrostatics", "bonds", "angles"], precision=dt)
        f = Forces(par, terms=terms, cutoff=9.0, switch_dist=7.5, algorithm="celllist", switch_mode=mode)
        p, b = pos_tensor(pos, 1, dt, dev), box_tensor(box, 1, dt, dev)
        F_only = torch.zeros_like(p)
        f._evaluate(p, b, F_only, False, True)  # forces-only variant
        F = torch.zeros_like(p)
        pots = f.compute(p, b, F, returnDetails=True)  # energy variant
        out[prec] = (F_only.double().cpu(), F.double().cpu(), pots[0])
    F64, F32 = out["f64"], out["f32"]
    scale = 1.0 + F64[1].abs()
    assert ((F64[0] - F64[1]).abs() / scale).max().item() < 1e-10
    assert ((F32[0] - F64[1]).abs() / scale).max().item() < 2e-4  # fp32 against fp64 arithmetic
    assert ((F32[1] - F64[1]).abs() / scale).max().item() < 2e-4
    for t in terms:
        assert abs(F32[2][t] - F64[2][t]) <= 2e-5 * 50 * max(1.0, abs(F64[2][t])), t
    # the switch really acts: energies differ from the unswitched ones
    par = Parameters(water_forcefield(mol), mol, ["lj", "electrostatics", "bonds", "angles"], precision=torch.float32)
    f0 = Forces(par, terms=terms, cutoff=9.0, algorithm="celllist")
    p, b = pos_tensor(pos, 1, torch.float32, dev), box_tensor(box, 1, torch.float32, dev)
    e0 = f0.compute(p, b, torch.zeros_like(p), returnDetails=True)[0]
    assert abs(e0["lj"] - F32[2]["lj"]) > 1.0
    # ---- the oracle leg: the same tensors through the reference arithmetic
    for prec in ("f64", "f32"):
        dt = PREC[prec]
        par = Parameters(water_forcefield(mol), mol, ["lj", "electrostatics", "bonds", "angles"], precision=dt)
        pairs = orc.candidate_pairs(pos, box, 9.6, orc.exclusion_pairs(par))
        for tt, kw in ((["lj"], dict(cutoff=9.0, switch_dist=7.5)), (["lj", "electrostatics"], dict(cutoff=9.0, switch_dist=7.5, rfa=True))):
            pc = pos_tensor(pos, 1, dt)
            if mode == "exact":
                pc.requires_grad_(True)
            po, Fo, npairs = orc.compute(par, pc, box_tensor(box, 1, dt), tt, pairs=pairs, explicit_forces=mode == "reference", **kw)
            f = Forces(par, terms=tt, algorithm="celllist", switch_mode=mode, **kw)
            p, b = pos_tensor(pos, 1, dt, dev), box_tensor(box, 1, dt, dev)
            F_only, F = torch.zeros_like(p), torch.zeros_like(p)
            f._evaluate(p, b, F_only, False, True)
            pots = f.compute(p, b, F, returnDetails=True)
            scale = 1.0 + Fo.abs()
            e_only = ((F_only.cpu() - Fo).abs() / scale).max().item()
            e_full = ((F.cpu() - Fo).abs() / scale).max().item()
            print(f"lean SWITCH {mode} {prec} {tt}: rel. dF {e_only:.2e} (forces only) {e_full:.2e} (with energies)")
            assert max(e_only, e_full) < (1e-10 if prec == "f64" else 6e-5), (prec, tt, e_only, e_full)
            for t in tt:
                assert abs(pots[0][t] - po[0][t]) <= ERTOL[prec] * EFAC * max(1.0, abs(po[0][t])), (prec, t)
            assert f.count_pairs(p, b) == npairs


def test_tiny_systems_and_odd_sizes():
    """Edge sizes: 1 atom (no pair at all), 2 atoms (one pair: analytic LJ), 3 water atoms (every pair
    excluded), 63 / 65 / 129 argon atoms (partial tiles of the all-pairs kernel) against the oracle, and
    several such replicas batched in one launch."""
    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import argon_forcefield, lj_box, tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.io import Topology
    from torchmd_amd.parameters import Parameters

    dev, dt = _dev(), torch.float64
    mol, pos, box = lj_box(6, seed=2)  # 216 argon atoms
    sig, eps = 3.345, 0.238
    for n in (1, 2, 63, 65, 129):
        sub = Topology(atomtype=np.array(["AR"] * n, dtype=object), charge=np.zeros(n), masses=np.full(n, 39.95))
        par = Parameters(argon_forcefield(sub), sub, ["lj"], precision=dt)
        p = pos[:n].copy()
        if n == 2:
            p[1] = p[0] + np.array([3.9, 0.3, -0.2])
        f = Forces(par, terms=["lj"], cutoff=9.0)
        R = 3
        pt = torch.tensor(np.stack([p + 0.01 * r for r in range(R)]), dtype=dt, device=dev)
        bt = box_tensor(box, R, dt, dev)
        F = torch.full_like(pt, 3.0)
        pots = f.compute(pt, bt, F, returnDetails=True)
        if n == 1:
            assert F.abs().max().item() == 0.0 and pots[0]["lj"] == 0.0
            continue
        po, Fo, _ = orc.compute(par, pt[:1].cpu(), bt[:1].cpu(), ["lj"], cutoff=9.0)
        assert (F[0].cpu() - Fo[0]).abs().max().item() < 1e-10
        assert abs(pots[0]["lj"] - po[0]["lj"]) < 1e-10 * max(1.0, abs(po[0]["lj"]))
        assert (F[1] - F[0]).abs().max().item() < 1e-9  # rigid shift of all atoms: same forces
        if n == 2:
            r = np.linalg.norm(p[1] - p[0])
            e = 4 * eps * ((sig / r) ** 12 - (sig / r) ** 6)
            assert abs(pots[0]["lj"] - e) < 1e-6  # (sigma/epsilon pass through float32 like the reference's)
    # one water molecule: all three pairs are excluded -> nonbonded terms vanish, bonded ones remain
    wmol, wpos, wbox = tip3p_box(1, seed=1)
    par = Parameters(water_forcefield(wmol), wmol, ["lj", "electrostatics", "bonds", "angles"], precision=dt)
    f = Forces(par, terms=["lj", "electrostatics", "bonds", "angles"], cutoff=9.0)
    pt, bt = pos_tensor(wpos, 1, dt, dev), torch.zeros(1, 3, 3, dtype=dt, device=dev)
    F = torch.zeros_like(pt)
    pots = f.compute(pt, bt, F, returnDetails=True)
    assert pots[0]["lj"] == 0.0 and pots[0]["electrostatics"] == 0.0
    assert F.sum(dim=1).abs().max().item() < 1e-9  # internal forces only


def test_minimum_image_known_geometry():
    """The reference's tiny periodic fixtures (tests/data/2watersperiodic, sodiumperiodic: two molecules
    15.1 A apart in a 19.3 A box, i.e. 4.2 A across the boundary): the periodic evaluation must equal the
    same molecules brought together by one box vector in an open box, and the oracle; two unit charges give
    Coulomb's law at the minimum-image distance."""
    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.io import Topology
    from torchmd_amd.parameters import Parameters

    dev, dt = _dev(), torch.float64
    box = np.array([19.339, 19.125, 19.140])
    mol, pos, _ = tip3p_box(2, seed=4)  # 8 waters: keep two
    mol2 = Topology(atomtype=mol.atomtype[:6], charge=mol.charge[:6], masses=mol.masses[:6],
                    bonds=mol.bonds[(mol.bonds < 6).all(axis=1)], angles=mol.angles[(mol.angles < 6).all(axis=1)])
    p = pos[:6].copy()
    p[:3] += np.array([2.0, 5.0, 5.0]) - p[0]
    p[3:] += np.array([17.1, 5.3, 4.8]) - p[3]
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol2), mol2, terms, precision=dt)
    f = Forces(par, terms=terms, cutoff=9.0, rfa=True)
    pt, bt = pos_tensor(p, 1, dt, dev), box_tensor(box, 1, dt, dev)
    F = torch.zeros_like(pt)
    e_per = f.compute(pt, bt, F, returnDetails=True)[0]
    q = p.copy()
    q[3:, 0] -= box[0]  # the image next to molecule 1
    Fo = torch.zeros_like(pt)
    e_open = f.compute(pos_tensor(q, 1, dt, dev), torch.zeros_like(bt), Fo, returnDetails=True)[0]
    for t in terms:
        assert abs(e_per[t] - e_open[t]) < 1e-10 * max(1.0, abs(e_open[t])), t
    assert (F - Fo).abs().max().item() < 1e-10
    assert abs(e_per["electrostatics"]) > 1e-3  # the molecules do interact across the boundary
    po, Fr, _ = orc.compute(par, pt.cpu(), bt.cpu(), terms, cutoff=9.0, rfa=True)
    assert (F.cpu() - Fr).abs().max().item() < 1e-10
    # two ions: plain Coulomb at the minimum-image distance
    ions = Topology(atomtype=np.array(["OT", "OT"], dtype=object), charge=np.array([1.0, 1.0]), masses=np.array([22.99, 22.99]))
    ipar = Parameters(water_forcefield(ions), ions, ["electrostatics"], precision=dt)
    ipos = np.array([[2.0, 5.0, 5.0], [17.1, 5.0, 5.0]])
    fi = Forces(ipar, terms=["electrostatics"], cutoff=9.0)
    Fi = torch.zeros(1, 2, 3, dtype=dt, device=dev)
    ei = fi.compute(pos_tensor(ipos, 1, dt, dev), bt, Fi, returnDetails=True)[0]["electrostatics"]
    r = box[0] - 15.1
    assert abs(ei - 332.06371307417066 / r) < 1e-9
    assert abs(Fi[0, 0, 0].item() - 332.06371307417066 / r**2) < 1e-9  # pushed apart: +x on the left ion


def test_more_than_2_pow_20_atoms_stay_on_the_lean_kernel():
    """1 124 864 argon atoms (> 2^20: the slot field of a list entry uses its bits 20..22, so the lean fp32 kernel
    runs every iteration in its checked loop, which masks the table offset): forces agree with the fp64 kernel's,
    vanish in the sum, and the energy per atom is that of the 10^6-atom box."""
    from torchmd_amd.builders import argon_forcefield, lj_box
    from torchmd_amd.forces import Forces
    from torchmd_amd.parameters import Parameters

    dev = _dev()
    mol, pos, box = lj_box(104, seed=4)
    assert mol.numAtoms > (1 << 20)
    res = {}
    for dt in (torch.float32, torch.float64):
        par = Parameters(argon_forcefield(mol), mol, ["lj"], precision=dt)
        f = Forces(par, terms=["lj"], cutoff=9.0)
        p = torch.tensor(pos, dtype=dt, device=dev)[None].contiguous()
        F = torch.zeros_like(p)
        e = f.compute(p, box_tensor(box, 1, dt, dev), F)[0]
        st = f.stats(p)
        assert st["algorithm"] == "celllist" and st["overflow"] == 0
        res[dt] = (e, F.double().cpu())
        f.close()
        del f, p, F
        torch.cuda.empty_cache()
    (e32, F32), (e64, F64) = res[torch.float32], res[torch.float64]
    assert F32.sum(dim=1).abs().max().item() < 0.5
    assert (F32 - F64).abs().max().item() < 1e-2
    assert abs(e32 - e64) < 2e-5 * abs(e64)
    assert -1.6 < e64 / mol.numAtoms < -0.6


def test_plain_evaluations_leave_the_rebuild_chain_out_and_recover(monkeypatch):
    """`compute()` with energies on the cell-list path (minimisers, callers that step the system themselves): the three
    launches of the rebuild chain are left out while the previous evaluation found no atom near its displacement limit
    (round 6, `tmdhip_compute`); an atom that crosses its limit all the same is caught by the flags the call reads back, and
    the evaluation is repeated with a fresh list.  Same positions in, same energies and forces out as with the chain in
    place on every evaluation (TMDHIP_SPEC_CHAIN=0) — small moves, a jump beyond the skin, small moves again."""
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.parameters import Parameters

    dev = _dev()
    mol, pos, box = tip3p_box(12, seed=21)
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=torch.float32)
    rng = np.random.default_rng(2)
    seq = [pos.copy()]
    for k in range(5):
        seq.append(seq[-1] + rng.normal(scale=0.01, size=pos.shape))
    jump = seq[-1].copy()
    jump[rng.choice(len(pos), 40, replace=False)] += rng.normal(scale=0.6, size=(40, 3))  # beyond the half skin of 0.6 A for many
    seq.append(jump)
    for k in range(4):
        seq.append(seq[-1] + rng.normal(scale=0.01, size=pos.shape))
    b = box_tensor(box, 1, torch.float32, dev)

    def run(spec):
        monkeypatch.setenv("TMDHIP_SPEC_CHAIN", "1" if spec else "0")
        f = Forces(par, terms=terms, cutoff=9.0, rfa=True, algorithm="celllist")
        out = []
        for x in seq:
            p = pos_tensor(x, 1, torch.float32, dev)
            F = torch.zeros_like(p)
            e = f.compute(p, b, F, returnDetails=True)
            out.append((e[0], F.cpu()))
        st = f.stats(p)
        f.close()
        return out, st

    a, sta = run(True)
    c, stc = run(False)
    # (evaluations 3-6 and the jump leave the chain out; the jump is caught and repeated, the evaluations behind it keep their chain)
    assert sta["chains_skipped"] >= 4 and stc["chains_skipped"] == 0, (sta, stc)
    assert sta["n_rebuilds"] >= 2  # the first list and the one behind the jump
    for k, ((ea, Fa), (ec, Fc)) in enumerate(zip(a, c)):
        for t in terms:
            assert abs(ea[t] - ec[t]) <= 1e-12 * max(1.0, abs(ec[t])), (k, t, ea[t], ec[t])
        assert torch.equal(Fa, Fc), k


@pytest.mark.parametrize("kw", [dict(cutoff=9.0, rfa=True), dict(cutoff=9.0, rfa=True, switch_dist=7.5), dict(cutoff=8.0)],
                         ids=["rf", "rf-switch", "coulomb"])
def test_plain_evaluation_with_the_bonded_terms_in_the_pair_launch(kw, monkeypatch):
    """`compute()` with energies on a cell-list context with a light topology (round 6, `tmdhip_compute`): the ENERGY variant of the
    lean pair launch carries evaluation-only step blocks that add the bonded force of their atoms and leave the bonded energies,
    and one kernel folds and reports — against the separate bonded / fold / report kernels (TMDHIP_FUSED_EVAL=0): forces bit for
    bit (the same device functions in the same order), energies to fp64 round-off; and both against the oracle."""
    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.parameters import Parameters

    dev = _dev()
    mol, pos, box = tip3p_box(14, seed=31)  # 8 232 atoms
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=torch.float32)
    rng = np.random.default_rng(4)
    seq = [pos, pos + rng.normal(scale=0.02, size=pos.shape), pos + rng.normal(scale=0.04, size=pos.shape)]
    b = box_tensor(box, 1, torch.float32, dev)
    out = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("TMDHIP_FUSED_EVAL", fused)
        f = Forces(par, terms=terms, algorithm="celllist", **kw)
        res = []
        for x in seq:
            p = pos_tensor(x, 1, torch.float32, dev)
            F = torch.full_like(p, 3.0)  # (must be overwritten)
            res.append((f.compute(p, b, F, returnDetails=True)[0], F.cpu()))
        out[fused] = res
        f.close()
    for (ea, Fa), (ec, Fc) in zip(out["1"], out["0"]):
        assert torch.equal(Fa, Fc)
        for t in terms:
            assert abs(ea[t] - ec[t]) <= 1e-12 * max(1.0, abs(ec[t])), (t, ea[t], ec[t])
    p = pos_tensor(seq[-1], 1, torch.float32)
    pairs = orc.candidate_pairs(seq[-1], box, kw["cutoff"] + 0.6, orc.exclusion_pairs(par))
    po, Fo, _ = orc.compute(par, p, box_tensor(box, 1, torch.float32), terms, pairs=pairs, **kw)
    ea, Fa = out["1"][-1]
    assert ((Fa - Fo).abs() / (1.0 + Fo.abs())).max().item() < 6e-5
    for t in terms:
        assert abs(ea[t] - po[0][t]) <= ERTOL["f32"] * EFAC * max(1.0, abs(po[0][t])), t


@pytest.mark.parametrize("lpa", [4, 8, 16, 32, 64])
def test_lean_kernel_energy_variant_at_every_lane_count(lpa, monkeypatch):
    """The ENERGY variant of the lean fp32 list kernel (an evaluation with per-term energies: `compute()`, the last step of a
    `step()` call) evaluates the entries of a list word one at a time since round 6 (five waves per SIMD): every lanes-per-atom
    instantiation on the 5 184-atom water box, LJ + reaction field + bonded terms, against the oracle — forces (with energies
    and forces only), per-term energies, in-cutoff pair count."""
    from oracle import torchmd_oracle as orc
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.parameters import Parameters

    monkeypatch.setenv("TMDHIP_LPA", str(lpa))
    dev, dt = _dev(), torch.float32
    mol, pos, box = tip3p_box(12, seed=23)
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=dt)
    kw = dict(cutoff=9.0, rfa=True)
    f = Forces(par, terms=terms, algorithm="celllist", **kw)
    p, b = pos_tensor(pos, 1, dt, dev), box_tensor(box, 1, dt, dev)
    F, F_only = torch.zeros_like(p), torch.zeros_like(p)
    pots = f.compute(p, b, F, returnDetails=True)
    f._evaluate(p, b, F_only, False, True)
    n_gpu = f.count_pairs(p, b)
    pairs = orc.candidate_pairs(pos, box, 9.6, orc.exclusion_pairs(par))
    po, Fo, npairs = orc.compute(par, pos_tensor(pos, 1, dt), box_tensor(box, 1, dt), terms, pairs=pairs, **kw)
    assert n_gpu == npairs
    assert max((F.cpu() - Fo).abs().max().item(), (F_only.cpu() - Fo).abs().max().item()) < 3e-4  # (FTOL of the fp32 list path)
    for t in terms:
        assert abs(pots[0][t] - po[0][t]) <= ERTOL["f32"] * EFAC * max(1.0, abs(po[0][t])), (t, pots[0][t], po[0][t])
