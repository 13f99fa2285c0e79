#!/usr/bin/env python
"""Generate tests/golden/*.npz by running THE REFERENCE ITSELF (imported read-only from
/root/reference) on its own fixtures.  Run in the build container only:

    python tests/golden/make_golden.py

Each file holds the *inputs* as plain arrays (so the parity tests need neither the reference nor its
data files on the GPU box) and the reference's *outputs* (per-term energies, forces, trajectory
end-points).  Topology ingestion uses this repo's readers (`torchmd_amd.io`, moleculekit/parmed are
not installable here); the reference's `Parameters`, `Forces`, `Integrator`, `System` do all maths.
"""

import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from torchmd.forces import Forces as RefForces  # noqa: E402
from torchmd.integrator import Integrator as RefIntegrator, maxwell_boltzmann  # noqa: E402
from torchmd.parameters import Parameters as RefParameters  # noqa: E402
from torchmd.systems import System as RefSystem  # noqa: E402

from torchmd_amd import io as tio  # noqa: E402
from torchmd_amd.forcefields import PrmtopForceField, YamlForceField  # noqa: E402

ALL_TERMS = ["bonds", "angles", "dihedrals", "impropers", "1-4", "electrostatics", "lj"]
PREC = {"f64": torch.double, "f32": torch.float}


def pack_parameters(par, out, prefix="par_"):
    """Flatten a reference Parameters object into npz-storable arrays."""
    out[prefix + "charges"] = par.charges.numpy()
    out[prefix + "masses"] = par.masses.numpy()
    out[prefix + "types"] = par.mapped_atom_types.numpy()
    if par.nonbonded_params is not None:
        out[prefix + "nonbonded_params"] = par.nonbonded_params["params"].numpy()
    for name in ("bond", "angle", "dihedral", "improper", "nonbonded_14"):
        tab = getattr(par, name + "_params")
        if tab is None or not torch.is_tensor(tab.get("idx")):
            continue
        out[prefix + name + "_idx"] = tab["idx"].numpy()
        out[prefix + name + "_map"] = tab["map"].numpy()
        out[prefix + name + "_params"] = tab["params"].numpy()


def run_reference(par, pos_np, box_np, terms, prec, R=1, explicit_forces=True, **kw):
    """One Forces.compute on the reference; returns (energy dicts, forces [R,N,3]).  explicit_forces=False: the
    reference's autograd forces (forces.py:328-336), positions passed with requires_grad as its own test does
    (tests/test_torchmd.py:496-517)."""
    n = pos_np.shape[0]
    system = RefSystem(n, R, PREC[prec], "cpu")
    system.set_positions(pos_np[:, :, None].astype(np.float64))
    system.set_box(np.asarray(box_np, dtype=np.float64))
    forces = RefForces(par, terms=terms, **kw)
    pos = system.pos if explicit_forces else system.pos.detach().requires_grad_(True)
    pots = forces.compute(pos, system.box, system.forces, returnDetails=True, explicit_forces=explicit_forces)
    return pots, system.forces.numpy().copy(), system


def store_case(out, tag, pots, forces):
    for r, p in enumerate(pots):
        for k, v in p.items():
            out[f"{tag}_E{r}_{k}"] = np.float64(v)
    out[f"{tag}_forces"] = forces


def water291():
    d = os.path.join(REF, "tests", "water")
    mol = tio.read_psf(os.path.join(d, "structure.psf"))
    xyz, box, _, _ = tio.read_pdb(os.path.join(d, "structure.pdb"))
    ff = YamlForceField(mol, os.path.join(d, "water_forcefield.yaml"))
    terms = ["lj", "bonds", "angles", "electrostatics"]  # tests/water/water_conf.yaml:6-10
    out = {"pos": xyz.astype(np.float64), "box": box.astype(np.float64)}
    out["atomtype"] = np.array([str(a) for a in mol.atomtype])
    out["mol_bonds"], out["mol_angles"] = mol.bonds, mol.angles
    out["mol_charge"], out["mol_masses"] = mol.charge, mol.masses
    for prec in ("f64", "f32"):
        par = RefParameters(ff, mol, terms, precision=PREC[prec], device="cpu")
        if prec == "f64":
            pack_parameters(par, out)
        for rfa in (False, True):
            for label, tt in (("full", terms), ("nb", ["lj", "electrostatics"])):
                pots, F, _ = run_reference(par, out["pos"], out["box"], tt, prec, R=2, cutoff=7.3, rfa=rfa)
                store_case(out, f"{prec}_{label}_rfa{int(rfa)}", pots, F)
        # short NVE trajectory (integrator.py:112-125): 5 steps of 1 fs from fixed velocities
        par = RefParameters(ff, mol, terms, precision=PREC[prec], device="cpu")
        _, _, system = run_reference(par, out["pos"], out["box"], terms, prec, R=2, cutoff=7.3, rfa=True)
        torch.manual_seed(7)
        vel = maxwell_boltzmann(par.masses, T=300, replicas=2)
        system.set_velocities(vel)
        if prec == "f64":
            out["traj_vel0"] = vel.numpy().astype(np.float64)
        system.set_velocities(torch.tensor(out["traj_vel0"]))
        forces = RefForces(par, terms=terms, cutoff=7.3, rfa=True)
        integ = RefIntegrator(system, forces, 1.0, "cpu", gamma=None, T=None)
        forces.compute(system.pos, system.box, system.forces)
        ekin, pot, T = integ.step(niter=5)
        out[f"{prec}_traj_pos"] = system.pos.numpy().copy()
        out[f"{prec}_traj_vel"] = system.vel.numpy().copy()
        out[f"{prec}_traj_forces"] = system.forces.numpy().copy()
        out[f"{prec}_traj_ekin"] = np.asarray(ekin, dtype=np.float64)
        out[f"{prec}_traj_pot"] = np.asarray(pot, dtype=np.float64)
        out[f"{prec}_traj_T"] = np.asarray(T, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "water291.npz"), **out)
    print("water291: fp64 rfa0", {k: float(v) for k, v in out.items() if k.startswith("f64_full_rfa0_E0")})


def ala2():
    d = os.path.join(REF, "tests", "data", "prod_alanine_dipeptide_amber")
    mol, top = tio.read_prmtop(os.path.join(d, "structure.prmtop"))
    xyz = tio.read_namd_coor(os.path.join(d, "input.coor"))
    box = tio.read_xsc(os.path.join(d, "input.xsc"))
    ff = PrmtopForceField(mol, top)
    out = {"pos": xyz, "box": box}
    out["atomtype"] = np.array([str(a) for a in mol.atomtype])
    for k in ("bonds", "angles", "dihedrals", "impropers"):
        out["mol_" + k] = getattr(mol, k)
    out["mol_charge"], out["mol_masses"] = mol.charge, mol.masses
    settings = dict(cutoff=9.0, switch_dist=7.5, rfa=True)  # tests/test_torchmd.py:370-373
    for prec in ("f64", "f32"):
        par = RefParameters(ff, mol, ALL_TERMS, precision=PREC[prec], device="cpu")
        if prec == "f64":
            pack_parameters(par, out)
        for label, tt in (("full", ALL_TERMS), ("nb", ["electrostatics", "lj"])):
            pots, F, _ = run_reference(par, xyz, box, tt, prec, **settings)
            store_case(out, f"{prec}_{label}_pbc", pots, F)
            pots, F, _ = run_reference(par, xyz, np.zeros(3), tt, prec, **settings)
            store_case(out, f"{prec}_{label}_box0", pots, F)
            pots, F, _ = run_reference(par, xyz, np.zeros(3), tt, prec)  # no cutoff, plain Coulomb
            store_case(out, f"{prec}_{label}_nocut", pots, F)
        # the autograd force flavour (explicit_forces=False: -dE/dr, without the switching quirk of forces.py:410-412)
        for label, tt in (("full", ALL_TERMS), ("nb", ["electrostatics", "lj"])):
            pots, F, _ = run_reference(par, xyz, box, tt, prec, explicit_forces=False, **settings)
            store_case(out, f"{prec}_{label}_pbc_autograd", pots, F)
        # exact (no switch) variant, and each optional pair term on its own
        pots, F, _ = run_reference(par, xyz, box, ["electrostatics", "lj"], prec, cutoff=9.0, rfa=True)
        store_case(out, f"{prec}_nb_pbc_noswitch", pots, F)
        pots, F, _ = run_reference(par, xyz, box, ["repulsion"], prec, cutoff=9.0)
        store_case(out, f"{prec}_repulsion_pbc", pots, F)
        pots, F, _ = run_reference(par, xyz, box, ["repulsioncg"], prec, cutoff=9.0)
        store_case(out, f"{prec}_repulsioncg_pbc", pots, F)
    # Known answers held by the reference's own tests: totals with box = 0
    tot_cut = sum(v for k, v in out.items() if k.startswith("f64_full_box0_E0_"))
    tot_nocut = sum(v for k, v in out.items() if k.startswith("f64_full_nocut_E0_"))
    print(f"ala2: Epot(cutoff9,switch7.5,rfa,box0) = {tot_cut:.6f}  [tests/test_torchmd.py:517 -> -1722.3569]")
    print(f"ala2: Epot(no cutoff)                  = {tot_nocut:.6f}  [tests/test_torchmd.py:605 -> -1768.8915]")
    print("ala2: fp32 pbc terms", {k[len('f32_full_pbc_E0_'):]: float(v) for k, v in out.items() if k.startswith("f32_full_pbc_E0_")})
    print("ala2: F[0] fp64 pbc", out["f64_full_pbc_forces"][0, 0])
    np.savez_compressed(os.path.join(HERE, "ala2.npz"), **out)


def thrombin():
    """4 676-atom non-periodic protein-ligand complex, no cutoff (tests/test_torchmd.py:297-466)."""
    d = os.path.join(REF, "tests", "data", "thrombin-ligand-amber")
    mol, top = tio.read_prmtop(os.path.join(d, "structure.prmtop"))
    xyz, _, _, _ = tio.read_pdb(os.path.join(d, "structure.pdb"))
    ff = PrmtopForceField(mol, top)
    out = {"pos": xyz.astype(np.float64), "box": np.zeros(3)}
    par = RefParameters(ff, mol, ALL_TERMS, precision=torch.double, device="cpu")
    pack_parameters(par, out)
    pots, F, _ = run_reference(par, out["pos"], out["box"], ["electrostatics", "lj"], "f64")
    store_case(out, "f64_nb_nocut", pots, F)
    pots, F, _ = run_reference(par, out["pos"], out["box"], ALL_TERMS, "f64")
    store_case(out, "f64_full_nocut", pots, F)
    keep = {k: v for k, v in out.items()}
    keep["pos"] = keep["pos"].astype(np.float32)  # PDB precision is 1e-3 A; float32 is lossless here
    for k in list(keep):
        if k.endswith("_forces"):
            keep[k] = keep[k].astype(np.float64)
    np.savez_compressed(os.path.join(HERE, "thrombin.npz"), **keep)
    print("thrombin:", {k: float(v) for k, v in out.items() if "_E0_" in k and k.startswith("f64_nb")})


def wrap():
    """Reference `Wrapper.wrap` (torchmd/wrapper.py:8-30) on the tests/water topology (97 bonded TIP3P groups),
    every molecule translated by random whole box vectors and then jittered, fp32 and fp64.  Case "r2": two
    replicas with different boxes, bonded groups only.  Case "ions": one replica with three free ions appended
    (the reference's free-atom branch, wrapper.py:28-30, only broadcasts for a single replica).  Inputs and the
    reference's wrapped positions are stored."""
    from torchmd.wrapper import Wrapper as RefWrapper

    d = os.path.join(REF, "tests", "water")
    mol = tio.read_psf(os.path.join(d, "structure.psf"))
    xyz, box, _, _ = tio.read_pdb(os.path.join(d, "structure.pdb"))
    rng = np.random.default_rng(11)
    nw = mol.numAtoms
    out = {"bonds": mol.bonds.astype(np.int64)}
    for case, R, nfree in (("r2", 2, 0), ("ions", 1, 3)):
        natoms = nw + nfree
        boxes = np.stack([box * (1.0 + 0.1 * r) * np.array([1.0, 0.93, 1.21]) for r in range(R)], axis=1)  # [3, R]
        pos0 = np.concatenate([xyz.astype(np.float64), rng.uniform(0, 1, (nfree, 3)) * box], axis=0)
        pos = np.repeat(pos0[None], R, axis=0)  # [R, N, 3]
        for r in range(R):
            shift_mol = rng.integers(-3, 4, size=(nw // 3, 3)).astype(np.float64) * boxes[:, r]
            pos[r, :nw] += np.repeat(shift_mol, 3, axis=0)
            pos[r, nw:] += rng.integers(-3, 4, size=(nfree, 3)).astype(np.float64) * boxes[:, r]
            pos[r] += rng.uniform(-0.4, 0.4, size=(natoms, 3)) * boxes[:, r]
        out[f"{case}_natoms"] = np.int64(natoms)
        out[f"{case}_boxes"] = boxes
        for prec in ("f64", "f32"):
            system = RefSystem(natoms, R, PREC[prec], "cpu")
            system.set_box(boxes)
            p = torch.tensor(pos, dtype=PREC[prec])
            out[f"{case}_{prec}_pos_in"] = p.numpy().copy()
            w = RefWrapper(natoms, mol.bonds, "cpu")
            w.wrap(p, system.box)
            out[f"{case}_{prec}_pos_out"] = p.numpy().copy()
            moved = np.abs(out[f"{case}_{prec}_pos_out"] - out[f"{case}_{prec}_pos_in"]).max()
            print(f"wrap {case} {prec}: {len(w.groups)} groups, {len(w.nongrouped)} free atoms, max translation {moved:.1f} A")
    np.savez_compressed(os.path.join(HERE, "wrap.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["water291", "ala2", "thrombin", "wrap"]
    for w in which:
        globals()[w]()
