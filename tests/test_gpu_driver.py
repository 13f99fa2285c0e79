"""End-to-end run of the moleculekit-free driver (`torchmd_amd.run`) on the GPU: the reference's
tests/water example (`tests/water/water_conf.yaml`: PSF + PDB + YAML force field, 2 replicas, cutoff 7.3,
Langevin) with its input files regenerated from tests/golden/water291.npz."""

import os

import numpy as np
import pytest
import torch
import yaml

from _golden import load

pytestmark = pytest.mark.gpu


def _write_psf(path, g):
    n = len(g["atomtype"])
    with open(path, "w") as fh:
        fh.write("PSF\n\n       1 !NTITLE\n REMARKS regenerated from tests/golden/water291.npz\n\n")
        fh.write(f"{n:8d} !NATOM\n")
        for i in range(n):
            t = str(g["atomtype"][i])
            fh.write(f"{i + 1:8d} WT0  {i // 3 + 1:<4d} TIP3 {('OH2', 'H1', 'H2')[i % 3]:<4s} {t:<4s} "
                     f"{g['mol_charge'][i]:10.6f} {g['mol_masses'][i]:13.4f}           0\n")
        for tag, arr, per_line in (("!NBOND: bonds", g["mol_bonds"], 4), ("!NTHETA: angles", g["mol_angles"], 3)):
            fh.write(f"\n{len(arr):8d} {tag}\n")
            flat = (np.asarray(arr) + 1).reshape(len(arr), -1)
            for k in range(0, len(flat), per_line):
                fh.write("".join(f"{v:8d}" for row in flat[k:k + per_line] for v in row) + "\n")
        fh.write("\n       0 !NPHI: dihedrals\n\n       0 !NIMPHI: impropers\n\n")


def _write_pdb(path, g):
    b = g["box"]
    with open(path, "w") as fh:
        fh.write(f"CRYST1{b[0]:9.3f}{b[1]:9.3f}{b[2]:9.3f}  90.00  90.00  90.00 P 1           1\n")
        for i, (x, y, z) in enumerate(g["pos"]):
            name = ("OH2", "H1", "H2")[i % 3]
            fh.write(f"ATOM  {i + 1:5d} {name:<4s} TIP3W{i // 3 + 1:4d}    {x:8.3f}{y:8.3f}{z:8.3f}  0.00  0.00      WT0  {name[0]}\n")
        fh.write("END\n")


def test_water_conf_end_to_end(tmp_path):
    from torchmd_amd import io as tio
    from torchmd_amd import run as driver
    from torchmd_amd.builders import TIP3P_FF

    g = load("water291")
    psf, pdb, ff = tmp_path / "structure.psf", tmp_path / "structure.pdb", tmp_path / "water_forcefield.yaml"
    _write_psf(psf, g)
    _write_pdb(pdb, g)
    ff.write_text(yaml.safe_dump(TIP3P_FF))
    mol = tio.read_psf(str(psf))
    assert mol.numAtoms == 291 and np.array_equal(mol.bonds, g["mol_bonds"]) and np.array_equal(mol.angles, g["mol_angles"])
    xyz, box, _, _ = tio.read_pdb(str(pdb))
    assert np.allclose(xyz, g["pos"], atol=5e-4) and np.allclose(box, g["box"], atol=1e-3)

    conf = {  # tests/water/water_conf.yaml of the reference, shortened
        "structure": [str(psf), str(pdb)], "forcefield": str(ff), "forceterms": ["LJ", "Bonds", "Angles", "Electrostatics"],
        "cutoff": 7.3, "rfa": False, "replicas": 2, "precision": "single", "device": "cuda", "timestep": 1,
        "temperature": 300, "langevin_gamma": 0.1, "langevin_temperature": 300, "seed": 1, "steps": 200,
        "output_period": 50, "save_period": 100, "log_dir": str(tmp_path / "log"), "output": "output",
    }
    cpath = tmp_path / "conf.yaml"
    cpath.write_text(yaml.safe_dump(conf))
    driver.main(["--conf", str(cpath)])
    log = tmp_path / "log"
    for k in range(2):
        rows = (log / f"monitor_{k}.csv").read_text().strip().splitlines()
        assert rows[0].split(",") == ["iter", "ns", "epot", "ekin", "etot", "T", "t"] and len(rows) == 5
        last = dict(zip(rows[0].split(","), rows[-1].split(",")))
        assert int(last["iter"]) == 200 and 100 < float(last["T"]) < 1000 and np.isfinite(float(last["epot"]))
        traj = np.load(log / f"output_{k}.npy")
        assert traj.shape == (291, 3, 4) and np.isfinite(traj).all()
        # wrapped: every molecule's centre is inside the box
        com = traj[:, :, -1].reshape(97, 3, 3).mean(axis=1)
        assert (com > -1e-3).all() and (com < g["box"] + 1e-3).all()
    assert os.path.exists(log / "input.yaml")


def test_external_plugin_equals_forces():
    """The reference's plugin protocol `External(file, embeddings, device, **kw).calculate(pos, box) ->
    (energy[R], forces[R,N,3])`: nonbonded terms through the hook + bonded terms in `Forces` == all terms in
    `Forces` (tests/water, 2 replicas)."""
    import numpy as np

    from _golden import GoldenParameters, box_tensor, load, pos_tensor
    from torchmd_amd.external import External
    from torchmd_amd.forces import Forces

    g = load("water291")
    dev = torch.device("cuda:0")
    par = GoldenParameters(g, torch.float64)
    p = pos_tensor(g["pos"], 2, torch.float64, dev)
    p[1] += 0.01 * torch.randn_like(p[1])
    b = box_tensor(g["box"], 2, torch.float64, dev)
    full = Forces(par, terms=["bonds", "angles", "lj", "electrostatics"], cutoff=7.3, rfa=True)
    F_full = torch.zeros_like(p)
    e_full = full.compute(p, b, F_full, returnDetails=True)
    ext = External(par, None, device="cuda:0", terms=["lj", "electrostatics"], cutoff=7.3, rfa=True)
    energy, forces = ext.calculate(p, b)
    assert energy.shape == (2,) and forces.shape == p.shape
    for r in range(2):
        assert abs(energy[r].item() - (e_full[r]["lj"] + e_full[r]["electrostatics"])) < 1e-9 * abs(energy[r].item())
    split = Forces(par, terms=["bonds", "angles"], external=ext, cutoff=7.3)
    F_split = torch.zeros_like(p)
    e_split = split.compute(p, b, F_split, returnDetails=True)
    assert (F_split - F_full).abs().max().item() < 1e-9
    for r in range(2):
        assert abs(e_split[r]["external"] - energy[r].item()) < 1e-9
        assert abs(sum(e_split[r].values()) - sum(e_full[r].values())) < 1e-9 * abs(sum(e_full[r].values()))
    with pytest.raises(ValueError, match="nonbonded terms only"):
        External(par, None, device="cuda:0", terms=["bonds"])
    assert np.isfinite(forces.cpu().numpy()).all()


def test_minimizers_lower_the_energy_of_a_water_box():
    """The three minimisers of `torchmd_amd.minimizers` (reference `minimizers.py`) on the real engine: the
    jittered 648-atom water box relaxes (L-BFGS-B on explicit forces, torch LBFGS on the differentiable
    potential, conjugate gradient)."""
    from torchmd_amd.builders import tip3p_box, water_forcefield
    from torchmd_amd.forces import Forces
    from torchmd_amd.minimizers import minimize_bfgs, minimize_cg, minimize_pytorch_bfgs
    from torchmd_amd.parameters import Parameters
    from torchmd_amd.systems import System

    dev, dt = torch.device("cuda:0"), torch.float64
    mol, pos, box = tip3p_box(6, seed=8)
    terms = ["lj", "electrostatics", "bonds", "angles"]
    par = Parameters(water_forcefield(mol), mol, terms, precision=dt)

    def fresh():
        s = System(mol.numAtoms, 1, dt, dev)
        s.set_positions(pos[:, :, None])
        s.set_box(box)
        f = Forces(par, terms=terms, cutoff=9.0, rfa=True)
        return s, f, f.compute(s.pos, s.box, s.forces)[0]

    s, f, e0 = fresh()
    minimize_bfgs(s, f, fmax=0.5, steps=30)
    e1 = f.compute(s.pos, s.box, s.forces)[0]
    assert e1 < e0 - 50.0
    s, f, e0 = fresh()
    en = minimize_pytorch_bfgs(s, f, steps=2, max_iter=10)
    e2 = f.compute(s.pos, s.box, s.forces)[0]
    assert en.shape[0] == 1 and e2 < e0 - 50.0
    s, f, e0 = fresh()
    minimize_cg(s, f, steps=6)
    e3 = f.compute(s.pos, s.box, s.forces)[0]
    assert e3 < e0 - 50.0
