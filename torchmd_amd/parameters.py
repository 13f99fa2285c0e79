"""Per-term index / parameter tensors — the *inputs* of the nonbonded hot path.

Field-compatible with the reference's `Parameters` (`torchmd/parameters.py:6-134`) so that either
class can be handed to `torchmd_amd.forces.Forces` (and to the reference's `Forces`):

    charges [N], masses [N,1], mapped_atom_types [N] (index into np.unique(type names)),
    nonbonded_params {"map","params"[T,2]=(sigma, eps)},
    bond_params / angle_params / dihedral_params / improper_params / nonbonded_14_params
        = {"idx": int64 [n,k], "map": int64 [m,2]=(row of idx, row of params), "params": [p,*]},
    A, B [T,T] Lennard-Jones tables (set by Forces via get_AB()).

This is one-time host set-up code (out of scope as a kernel); it exists so the
package is usable without the reference being importable (it is absent on the GPU box).
"""

from __future__ import annotations

from math import sqrt

import numpy as np
import torch


def lorentz_berthelot_AB(sigma: torch.Tensor, epsilon: torch.Tensor):
    """A = 4 eps_ij sigma_ij^12, B = 4 eps_ij sigma_ij^6 with sigma_ij = (s_i+s_j)/2,
    eps_ij = sqrt(e_i e_j)  (reference `calculate_AB`, `torchmd/parameters.py:449-457`)."""
    sig6 = (0.5 * (sigma[None, :] + sigma[:, None])) ** 6
    eps4 = torch.sqrt(epsilon[None, :] * epsilon[:, None]) * 4
    B = eps4 * sig6
    A = eps4 * sig6 * sig6
    return A, B


calculate_AB = lorentz_berthelot_AB  # reference name

# The reference builds every "params" table with `torch.tensor(list_of_python_floats)`, i.e. in
# torch's default float32, and only then casts to the run precision (parameters.py:150,170,192,
# 213,252,297 + precision_ 65-87).  Double-precision runs therefore use float32-rounded force-field
# constants; we reproduce that so both classes yield bit-identical tensors.
_PARAM_DTYPE = torch.float32


def _canonical_rows(rows: np.ndarray, ordered_by_ends: bool) -> np.ndarray:
    """Unique index tuples.  Bonds are stored with sorted ends; angles / dihedrals are flipped so
    that first < last (reference `make_bonds/make_angles/make_dihedrals`, parameters.py:159-214)."""
    rows = np.asarray(rows, dtype=np.int64)
    if rows.size == 0:
        return rows.reshape(0, rows.shape[-1] if rows.ndim == 2 else 0)
    if ordered_by_ends:
        flip = rows[:, 0] >= rows[:, -1]
        # reference keeps a row as-is only when first < last, otherwise reverses it
        rows = np.where(flip[:, None], rows[:, ::-1], rows)
    else:
        rows = np.sort(rows, axis=1)
    return np.unique(rows, axis=0)


class Parameters:
    def __init__(self, ff, mol, terms=None, precision=torch.float, device="cpu"):
        self.nonbonded_params = None
        self.bond_params = None
        self.charges = None
        self.masses = None
        self.mapped_atom_types = None
        self.angle_params = None
        self.dihedral_params = None
        self.nonbonded_14_params = None
        self.improper_params = None
        self.A = None
        self.B = None

        self.natoms = mol.numAtoms
        if terms is None:
            terms = ("bonds", "angles", "dihedrals", "impropers", "1-4", "lj")
        terms = [t.lower() for t in terms]
        self._build(ff, mol, terms)
        self.precision_(precision)
        self.to_(device)

    # ------------------------------------------------------------------ housekeeping
    _TABLES = (
        "nonbonded_params",
        "bond_params",
        "angle_params",
        "dihedral_params",
        "nonbonded_14_params",
        "improper_params",
    )

    def to_(self, device):
        self.charges = self.charges.to(device)
        self.masses = self.masses.to(device)
        for name in self._TABLES:
            tab = getattr(self, name)
            if tab is None:
                continue
            for k in ("idx", "map", "params"):
                if k in tab and torch.is_tensor(tab[k]):
                    tab[k] = tab[k].to(device)
        if self.mapped_atom_types is not None:
            self.mapped_atom_types = self.mapped_atom_types.to(device)
        if self.A is not None:
            self.A, self.B = self.A.to(device), self.B.to(device)
        self.device = device

    def precision_(self, precision):
        self.charges = self.charges.type(precision)
        self.masses = self.masses.type(precision)
        for name in self._TABLES:
            tab = getattr(self, name)
            if tab is not None and torch.is_tensor(tab.get("params")):
                tab["params"] = tab["params"].type(precision)

    # ------------------------------------------------------------------ exclusions
    def get_exclusions(self, types=("bonds", "angles", "1-4"), fullarray=False):
        """Excluded pairs: all bonds, the (0,2) ends of angles, the (0,3) ends of dihedrals
        (reference `get_exclusions`, parameters.py:89-107)."""
        pairs = []
        if self.bond_params is not None and "bonds" in types:
            pairs += self.bond_params["idx"].cpu().numpy().tolist()
        if self.angle_params is not None and "angles" in types:
            pairs += self.angle_params["idx"].cpu().numpy()[:, [0, 2]].tolist()
        if self.dihedral_params is not None and "1-4" in types:
            pairs += self.dihedral_params["idx"].cpu().numpy()[:, [0, 3]].tolist()
        if fullarray:
            full = np.zeros((self.natoms, self.natoms), dtype=bool)
            if len(pairs):
                p = np.asarray(pairs)
                full[p[:, 0], p[:, 1]] = True
                full[p[:, 1], p[:, 0]] = True
            return full
        return pairs

    def get_AB(self):
        p = self.nonbonded_params["params"]
        return lorentz_berthelot_AB(p[:, 0], p[:, 1])

    def get_AB_14(self):
        p = self.nonbonded_14_params["params"]
        return lorentz_berthelot_AB(p[:, 0], p[:, 1])

    # ------------------------------------------------------------------ builders
    def _build(self, ff, mol, terms):
        atomtype = np.asarray(mol.atomtype, dtype=object)
        uq, inverse = np.unique(atomtype, return_inverse=True)
        self.atomtypes = atomtype
        self.mapped_atom_types = torch.tensor(inverse, dtype=torch.int64)
        self.charges = torch.tensor(np.asarray(mol.charge).astype(np.float64))
        if mol.masses is not None:
            self.masses = torch.tensor(np.asarray(mol.masses)).to(torch.float32)[:, None]
        elif np.all(atomtype != "") and ff.prm is not None:
            self.masses = torch.tensor([ff.get_mass(a) for a in atomtype])[:, None]
        else:
            raise RuntimeError("No masses or atomtypes defined in the Molecule.")

        if any(t in terms for t in ("lj", "repulsion", "repulsioncg")):
            self.nonbonded_params = {
                "idx": [],
                "map": torch.tensor(np.stack([np.arange(len(atomtype)), inverse], axis=1)),
                "params": torch.tensor([list(ff.get_LJ(a)) for a in uq], dtype=_PARAM_DTYPE),
            }
        if "bonds" in terms and len(mol.bonds):
            self.bond_params = self._typed_table(
                _canonical_rows(mol.bonds, ordered_by_ends=False), atomtype, ff.get_bond
            )
        if "angles" in terms and len(mol.angles):
            self.angle_params = self._typed_table(
                _canonical_rows(mol.angles, ordered_by_ends=True), atomtype, ff.get_angle
            )
        if "dihedrals" in terms and len(mol.dihedrals):
            self.dihedral_params = self._torsion_table(
                _canonical_rows(mol.dihedrals, ordered_by_ends=True), atomtype, ff
            )
        if "impropers" in terms and len(mol.impropers):
            self.improper_params = self._improper_table(mol, atomtype, ff)
        if "1-4" in terms and len(mol.dihedrals):
            self.nonbonded_14_params = self._one_four_table(mol, atomtype, ff)

    @staticmethod
    def _typed_table(rows, atomtype, getter):
        """One parameter row per distinct type tuple; `map[:,1]` points each instance at it."""
        slot, params, mp = {}, [], []
        for i, r in enumerate(rows):
            key = tuple(atomtype[r])
            if key not in slot:
                slot[key] = len(params)
                params.append(list(getter(*key)))
            mp.append((i, slot[key]))
        return {
            "idx": torch.tensor(rows.astype(np.int64)),
            "map": torch.tensor(mp, dtype=torch.int64),
            "params": torch.tensor(params, dtype=_PARAM_DTYPE),
        }

    @staticmethod
    def _torsion_table(rows, atomtype, ff):
        """Dihedrals may carry several Fourier terms: `map` has one row per (dihedral, term)."""
        slot, params, mp = {}, [], []
        for i, r in enumerate(rows):
            key = tuple(atomtype[r])
            if key not in slot:
                first = len(params)
                tt = ff.get_dihedral(*key)
                params.extend([list(t) for t in tt])
                slot[key] = list(range(first, first + len(tt)))
            mp.extend((i, p) for p in slot[key])
        return {
            "idx": torch.tensor(rows.astype(np.int64)),
            "map": torch.tensor(mp, dtype=torch.int64),
            "params": torch.tensor(params, dtype=_PARAM_DTYPE),
        }

    @staticmethod
    def _improper_table(mol, atomtype, ff):
        rows = np.unique(np.asarray(mol.impropers, dtype=np.int64), axis=0)
        bonds = _canonical_rows(mol.bonds, ordered_by_ends=False)
        nbr = {}
        for a, b in bonds:
            nbr.setdefault(int(a), set()).add(int(b))
            nbr.setdefault(int(b), set()).add(int(a))
        slot, params, mp = {}, [], []
        for i, r in enumerate(rows):
            key = tuple(atomtype[r])
            try:
                prm = ff.get_improper(*key)
            except Exception:
                # centre = the atom bonded to the three others (reference parameters.py:233-239, 466-469)
                centre = next(int(a) for a in r if len(nbr.get(int(a), set()) & set(int(x) for x in r)) == 3)
                rest = sorted(int(a) for a in r if int(a) != centre)
                order = [rest[0], rest[1], centre, rest[2]]
                key = tuple(atomtype[order])
                prm = ff.get_improper(*key)
            if key not in slot:
                slot[key] = len(params)
                params.append(list(prm))
            mp.append((i, slot[key]))
        return {
            "idx": torch.tensor(rows),
            "map": torch.tensor(mp, dtype=torch.int64),
            "params": torch.tensor(params, dtype=_PARAM_DTYPE),
        }

    def _one_four_table(self, mol, atomtype, ff):
        """Scaled 1-4 pairs: dihedral ends that are not already bond/angle exclusions, one entry per
        distinct (first,last) pair, params = (A, B, scnb, scee) (reference `make_14`,
        parameters.py:255-299)."""
        rows = _canonical_rows(mol.dihedrals, ordered_by_ends=True)
        table = {"idx": [], "map": [], "params": []}
        excluded = set()
        for a, b in self.get_exclusions(types=("bonds", "angles")):
            excluded.add((int(a), int(b)))
            excluded.add((int(b), int(a)))
        keep = np.array([(int(r[0]), int(r[3])) not in excluded for r in rows], dtype=bool)
        rows = rows[keep]
        if not len(rows):
            return table
        _, first = np.unique(rows[:, [0, 3]], axis=0, return_index=True)
        rows = rows[first]
        slot, params, mp = {}, [], []
        for i, r in enumerate(rows):
            key = tuple(atomtype[r])
            if key[::-1] in slot:
                key = key[::-1]
            if key not in slot:
                scnb, scee, s1, e1, s4, e4 = ff.get_14(*key)
                sig6 = (0.5 * (s1 + s4)) ** 6
                eps4 = 4 * sqrt(e1 * e4)
                slot[key] = len(params)
                params.append([eps4 * sig6 * sig6, eps4 * sig6, scnb, scee])
            mp.append((i, slot[key]))
        table["idx"] = torch.tensor(rows[:, [0, 3]].astype(np.int64))
        table["map"] = torch.tensor(mp, dtype=torch.int64)
        table["params"] = torch.tensor(params, dtype=_PARAM_DTYPE)
        return table
