"""AMBER prmtop force-field backend.

The reference reads AMBER parameters through parmed (`torchmd/forcefields/ff_parmed.py:46-129`,
`AmberParameterSet.from_structure`), keyed by AMBER atom-type *names*.  parmed is not available
here, so this backend derives the same per-type tables straight from the prmtop sections
(recipe validated in SURVEY.md §8(c)):

* LJ: per-type sigma/epsilon from the diagonal `A_ii, B_ii` (`ATOM_TYPE_INDEX`,
  `NONBONDED_PARM_INDEX`, `LENNARD_JONES_ACOEF/BCOEF`): rmin/2 = (2A/B)^(1/6)/2,
  eps = B^2/(4A), sigma = 2 * rmin/2 * 2^(-1/6); `A_ii < 1e-10` -> (0, 0) (TIP3P hydrogens).
  The 1-4 variants equal the normal ones in AMBER.
* bonds / angles: (k, eq) of the first instance of each type-name tuple.
* dihedrals: every distinct (phi_k, phase, per) seen for a type-name quad, in file order,
  plus scee/scnb of the first term; impropers = rows with negative 4th index.
"""

from __future__ import annotations

import numpy as np

from ..io import AmberPrmtop, read_prmtop
from .forcefield import ForceFieldBase


class PrmtopForceField(ForceFieldBase):
    def __init__(self, mol, prm):
        if isinstance(prm, AmberPrmtop):
            top = prm
        else:
            _, top = read_prmtop(prm)
        self.mol = mol
        self.prm = top  # truthy, mirrors ParmedForcefield.prm
        natom, ntypes = (int(v) for v in top.ints("POINTERS")[:2])
        names = np.array(top.strs("AMBER_ATOM_TYPE")[:natom], dtype=object)
        self._names = names

        # ---- LJ per type name
        tidx = top.ints("ATOM_TYPE_INDEX")[:natom]
        nbidx = top.ints("NONBONDED_PARM_INDEX")
        acoef, bcoef = top.floats("LENNARD_JONES_ACOEF"), top.floats("LENNARD_JONES_BCOEF")
        self._lj = {}
        self._charge = {}
        self._mass = {}
        charges = top.floats("CHARGE")[:natom] / 18.2223
        masses = top.floats("MASS")[:natom]
        for i in range(natom):
            nm = names[i]
            if nm in self._lj:
                continue
            t = int(tidx[i])
            k = int(nbidx[ntypes * (t - 1) + t - 1]) - 1
            a, b = float(acoef[k]), float(bcoef[k])
            if a < 1e-10 or b < 1e-10:
                sigma, eps = 0.0, 0.0
            else:
                factor = 2.0 * a / b
                rmin_half = factor ** (1.0 / 6.0) * 0.5
                eps = b / (2.0 * factor)
                sigma = 2.0 * rmin_half * 2.0 ** (-1.0 / 6.0)
            self._lj[nm] = (sigma, eps)
            self._charge[nm] = float(charges[i])
            self._mass[nm] = float(masses[i])

        def rows(flags, width):
            parts = [top.ints(f).reshape(-1, width + 1) for f in flags if len(top.ints(f))]
            return np.concatenate(parts, axis=0) if parts else np.zeros((0, width + 1), dtype=np.int64)

        # ---- bonds
        bk, breq = top.floats("BOND_FORCE_CONSTANT"), top.floats("BOND_EQUIL_VALUE")
        self._bonds = {}
        for i3, j3, ty in rows(("BONDS_INC_HYDROGEN", "BONDS_WITHOUT_HYDROGEN"), 2):
            key = (names[i3 // 3], names[j3 // 3])
            val = (float(bk[ty - 1]), float(breq[ty - 1]))
            self._bonds.setdefault(key, val)
            self._bonds.setdefault(key[::-1], val)

        # ---- angles
        ak, aeq = top.floats("ANGLE_FORCE_CONSTANT"), top.floats("ANGLE_EQUIL_VALUE")
        self._angles = {}
        for i3, j3, k3, ty in rows(("ANGLES_INC_HYDROGEN", "ANGLES_WITHOUT_HYDROGEN"), 3):
            key = (names[i3 // 3], names[j3 // 3], names[k3 // 3])
            val = (float(ak[ty - 1]), float(aeq[ty - 1]))
            self._angles.setdefault(key, val)
            self._angles.setdefault(key[::-1], val)

        # ---- torsions
        dk = top.floats("DIHEDRAL_FORCE_CONSTANT")
        dper = top.floats("DIHEDRAL_PERIODICITY")
        dph = top.floats("DIHEDRAL_PHASE")
        scee = top.floats("SCEE_SCALE_FACTOR")
        scnb = top.floats("SCNB_SCALE_FACTOR")
        self._dihedrals = {}  # key -> {"terms": [[k, phase, per], ...], "scee": , "scnb": }
        self._impropers = {}
        for i3, j3, k3, l3, ty in rows(("DIHEDRALS_INC_HYDROGEN", "DIHEDRALS_WITHOUT_HYDROGEN"), 4):
            key = (names[abs(i3) // 3], names[abs(j3) // 3], names[abs(k3) // 3], names[abs(l3) // 3])
            term = [float(dk[ty - 1]), float(dph[ty - 1]), int(round(float(dper[ty - 1])))]
            if l3 < 0:
                self._impropers.setdefault(key, tuple(term))
                continue
            entry = self._dihedrals.get(key)
            if entry is None:
                entry = self._dihedrals.get(key[::-1])
            if entry is None:
                entry = {
                    "terms": [],
                    "scee": float(scee[ty - 1]) if len(scee) else 1.2,
                    "scnb": float(scnb[ty - 1]) if len(scnb) else 2.0,
                }
                self._dihedrals[key] = entry
            if term not in entry["terms"]:
                entry["terms"].append(term)

    # -- ForceFieldBase -----------------------------------------------------------------
    def get_atom_types(self):
        return np.unique(self._names)

    def get_charge(self, at):
        return self._charge[at]

    def get_mass(self, at):
        return self._mass[at]

    def get_LJ(self, at):
        return self._lj[at]

    def get_bond(self, at1, at2):
        return self._bonds[(at1, at2)]

    def get_angle(self, at1, at2, at3):
        return self._angles[(at1, at2, at3)]

    def _dihedral_entry(self, at1, at2, at3, at4):
        for key in ((at1, at2, at3, at4), (at4, at3, at2, at1)):
            if key in self._dihedrals:
                return self._dihedrals[key]
        raise RuntimeError(f"Could not find dihedral parameters for ({at1}, {at2}, {at3}, {at4})")

    def get_dihedral(self, at1, at2, at3, at4):
        return [list(t) for t in self._dihedral_entry(at1, at2, at3, at4)["terms"]]

    def get_14(self, at1, at2, at3, at4):
        e = self._dihedral_entry(at1, at2, at3, at4)
        s1, e1 = self._lj[at1]
        s4, e4 = self._lj[at4]
        return e["scnb"], e["scee"], s1, e1, s4, e4

    def get_improper(self, at1, at2, at3, at4):
        from itertools import permutations

        types = (at1, at2, at3, at4)
        for p in permutations(range(4)):
            if p[2] != 2:
                continue
            key = tuple(types[i] for i in p)
            if key in self._impropers:
                return self._impropers[key]
        raise RuntimeError(f"Could not find improper parameters for key {types}")
