"""YAML force-field backend (same file format as the reference's `ff_yaml.py`, e.g.
`tests/water/water_forcefield.yaml`): top-level sections `atomtypes, bonds, angles, dihedrals,
impropers, lj, electrostatics, masses`, keys are `TYPE` or `(T1, T2, ...)`, `X` is a wildcard.

Matching rule (reference `torchmd/forcefields/ff_yaml.py:13-52`): exact patterns win over
wildcard patterns (fewest `X` first); bonds/angles/dihedrals may match reversed; impropers may
match any permutation that keeps the centre (3rd position) fixed.
"""

from __future__ import annotations

from itertools import permutations, product
from math import radians

import numpy as np
import yaml

from .forcefield import ForceFieldBase

_IMPROPER_PERMS = [p for p in permutations(range(4)) if p[2] == 2]


def _key(types):
    types = [str(t) for t in types]
    return types[0] if len(types) == 1 else "(" + ", ".join(types) + ")"


def _wildcard_patterns(types):
    """All X-substituted variants of `types`, grouped by number of wildcards (ascending)."""
    n = len(types)
    masks = sorted(product((False, True), repeat=n), key=lambda m: (sum(m), m))
    return [(sum(m), tuple("X" if w else t for t, w in zip(types, m))) for m in masks]


class YamlForceField(ForceFieldBase):
    def __init__(self, mol, prm):
        self.mol = mol
        if isinstance(prm, dict):
            self.prm = prm
        else:
            with open(prm, "r") as fh:
                self.prm = yaml.safe_load(fh)

    # -- generic lookup -----------------------------------------------------------------
    def get_parameters(self, term, atomtypes):
        atomtypes = [str(a) for a in atomtypes]
        orders = [tuple(atomtypes)]
        if term in ("bonds", "angles", "dihedrals"):
            orders.append(tuple(reversed(atomtypes)))
        elif term == "impropers":
            orders += [tuple(atomtypes[i] for i in p) for p in _IMPROPER_PERMS]
        cands = []
        for rank, order in enumerate(orders):
            for nx, pat in _wildcard_patterns(order):
                cands.append((nx, rank, pat))
        cands.sort(key=lambda c: c[0])  # stable: fewest wildcards first, then order of `orders`
        table = self.prm[term]
        for _, _, pat in cands:
            k = _key(pat)
            if k in table:
                return table[k]
        raise RuntimeError(f"{np.array(atomtypes)} doesn't have {term} information in the FF")

    # -- ForceFieldBase -----------------------------------------------------------------
    def get_atom_types(self):
        return np.unique(self.prm["atomtypes"])

    def get_charge(self, at):
        return self.get_parameters("electrostatics", [at])["charge"]

    def get_mass(self, at):
        return self.prm["masses"][at]

    def get_LJ(self, at):
        p = self.get_parameters("lj", [at])
        return p["sigma"], p["epsilon"]

    def get_bond(self, at1, at2):
        p = self.get_parameters("bonds", [at1, at2])
        return p["k0"], p["req"]

    def get_angle(self, at1, at2, at3):
        p = self.get_parameters("angles", [at1, at2, at3])
        return p["k0"], radians(p["theta0"])

    def _torsion_terms(self, p):
        return [[t["phi_k"], radians(t["phase"]), t["per"]] for t in p["terms"]]

    def get_dihedral(self, at1, at2, at3, at4):
        return self._torsion_terms(self.get_parameters("dihedrals", [at1, at2, at3, at4]))

    def get_14(self, at1, at2, at3, at4):
        p = self.get_parameters("dihedrals", [at1, at2, at3, at4])
        lj1 = self.get_parameters("lj", [at1])
        lj4 = self.get_parameters("lj", [at4])
        return (
            p.get("scnb", 1),
            p.get("scee", 1),
            lj1["sigma14"],
            lj1["epsilon14"],
            lj4["sigma14"],
            lj4["epsilon14"],
        )

    def get_improper(self, at1, at2, at3, at4):
        p = self.get_parameters("impropers", [at1, at2, at3, at4])
        return p["phi_k"], radians(p["phase"]), p["per"]
