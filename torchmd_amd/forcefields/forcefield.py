"""Force-field lookup interface (host-side, set-up time only).

Mirrors the lookup surface `Parameters` needs from the reference's `_ForceFieldBase`
(`torchmd/forcefields/forcefield.py:5-43`) and the `ForceField.create(mol, prm)` factory
(`forcefield.py:46-62`).  The parmed backend of the reference is replaced by a direct AMBER
prmtop backend (`ff_prmtop.py`) because parmed is not available in this image.
"""

from __future__ import annotations

import os


class ForceFieldBase:
    """Per-atom-type parameter lookup.  Units: kcal/mol, Angstrom, radians, g/mol, e."""

    prm = None  # truthy object when masses can be looked up by type (reference parameters.py:117)

    def get_atom_types(self):
        raise NotImplementedError

    def get_charge(self, at):
        raise NotImplementedError

    def get_mass(self, at):
        raise NotImplementedError

    def get_LJ(self, at):
        """-> (sigma, epsilon)"""
        raise NotImplementedError

    def get_bond(self, at1, at2):
        """-> (k0, req)"""
        raise NotImplementedError

    def get_angle(self, at1, at2, at3):
        """-> (k0, theta0 [rad])"""
        raise NotImplementedError

    def get_dihedral(self, at1, at2, at3, at4):
        """-> list of [phi_k, phase [rad], periodicity]"""
        raise NotImplementedError

    def get_14(self, at1, at2, at3, at4):
        """-> (scnb, scee, sigma14_1, eps14_1, sigma14_4, eps14_4)"""
        raise NotImplementedError

    def get_improper(self, at1, at2, at3, at4):
        """-> (phi_k, phase [rad], periodicity)  (periodicity 0 = harmonic)"""
        raise NotImplementedError


class ForceField:
    """Factory with the reference's calling convention: `ForceField.create(mol, prm)`."""

    @staticmethod
    def create(mol, prm):
        from .ff_prmtop import PrmtopForceField
        from .ff_yaml import YamlForceField

        if isinstance(prm, str):
            ext = os.path.splitext(prm)[-1].lower()
            if ext in (".yaml", ".yml"):
                return YamlForceField(mol, prm)
            if ext in (".prmtop", ".parm7"):
                return PrmtopForceField(mol, prm)
            raise RuntimeError(
                f"Unsupported force-field file '{prm}': this build reads .yaml/.yml and AMBER .prmtop "
                "(CHARMM .prm / .frcmod need parmed, which is not part of this environment)"
            )
        if isinstance(prm, ForceFieldBase):
            return prm
        raise RuntimeError("prm must be a file name or a ForceFieldBase instance")
