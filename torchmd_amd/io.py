"""Topology / coordinate readers that feed the nonbonded hot path.

The reference gets its `mol` object from moleculekit (`torchmd/run.py:158-175`), which is not
installable here.  `Parameters` (reference `torchmd/parameters.py:109-134`) only needs a duck-typed
object with `numAtoms, atomtype, charge, masses, bonds, angles, dihedrals, impropers, coords, box,
element`, so this module provides `Topology` plus minimal readers for the formats the reference's
example configs use: X-PLOR PSF, PDB (CRYST1 + ATOM/HETATM), AMBER
prmtop, NAMD binary .coor and .xsc.
"""

from __future__ import annotations

import re
import struct
from dataclasses import dataclass, field

import numpy as np


@dataclass
class Topology:
    """Molecule-like container (duck type of `moleculekit.Molecule` as used by
    reference `torchmd/parameters.py:109-134` and `torchmd/run.py:211-216`)."""

    atomtype: np.ndarray  # object/str [N]
    charge: np.ndarray  # float [N]
    masses: np.ndarray  # float [N]
    bonds: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), dtype=np.int64))
    angles: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), dtype=np.int64))
    dihedrals: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), dtype=np.int64))
    impropers: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), dtype=np.int64))
    coords: np.ndarray | None = None  # [N,3,F] float32, Angstrom
    box: np.ndarray | None = None  # [3,F]
    element: np.ndarray | None = None
    name: np.ndarray | None = None

    @property
    def numAtoms(self) -> int:
        return len(self.atomtype)

    @property
    def numFrames(self) -> int:
        return 0 if self.coords is None else self.coords.shape[2]


# --------------------------------------------------------------------------- PSF
def _psf_section(lines, tag, width):
    """Return the integer table that follows the `!TAG` header line, reshaped to [-1, width]."""
    for k, ln in enumerate(lines):
        if tag in ln:
            count = int(ln.split()[0])
            vals = []
            j = k + 1
            while len(vals) < count * width and j < len(lines):
                vals.extend(int(t) for t in lines[j].split())
                j += 1
            arr = np.asarray(vals[: count * width], dtype=np.int64).reshape(-1, width)
            return arr - 1  # PSF indices are 1-based
    return np.zeros((0, width), dtype=np.int64)


def read_psf(path: str) -> Topology:
    """X-PLOR/CHARMM PSF: !NATOM rows (id seg resid resname name type charge mass ...),
    then !NBOND / !NTHETA / !NPHI / !NIMPHI 1-based index lists
    (layout as in reference fixture `tests/water/structure.psf:14-20,307,382`)."""
    with open(path) as fh:
        lines = fh.read().splitlines()
    start = next(k for k, ln in enumerate(lines) if "!NATOM" in ln)
    natom = int(lines[start].split()[0])
    types, charges, masses, names = [], [], [], []
    for ln in lines[start + 1 : start + 1 + natom]:
        t = ln.split()
        names.append(t[4])
        types.append(t[5])
        charges.append(float(t[6]))
        masses.append(float(t[7]))
    return Topology(
        atomtype=np.array(types, dtype=object),
        charge=np.array(charges, dtype=np.float32),
        masses=np.array(masses, dtype=np.float32),
        bonds=_psf_section(lines, "!NBOND", 2),
        angles=_psf_section(lines, "!NTHETA", 3),
        dihedrals=_psf_section(lines, "!NPHI", 4),
        impropers=_psf_section(lines, "!NIMPHI", 4),
        name=np.array(names, dtype=object),
    )


# --------------------------------------------------------------------------- PDB
def read_pdb(path: str):
    """Returns (coords [N,3] float32, box [3] float32 or zeros, names, elements) of the first MODEL."""
    xyz, names, elems = [], [], []
    box = np.zeros(3, dtype=np.float32)
    with open(path) as fh:
        for ln in fh:
            rec = ln[:6]
            if rec == "CRYST1":
                box = np.array([float(ln[6:15]), float(ln[15:24]), float(ln[24:33])], dtype=np.float32)
            elif rec in ("ATOM  ", "HETATM"):
                xyz.append((float(ln[30:38]), float(ln[38:46]), float(ln[46:54])))
                names.append(ln[12:16].strip())
                elems.append(ln[76:78].strip() if len(ln) >= 78 else "")
            elif rec == "ENDMDL":
                break
    return (
        np.asarray(xyz, dtype=np.float32),
        box,
        np.array(names, dtype=object),
        np.array(elems, dtype=object),
    )


# --------------------------------------------------------------------------- NAMD coor / xsc
def read_namd_coor(path: str) -> np.ndarray:
    """NAMD binary coordinates: int32 N followed by N*3 float64 (little endian)."""
    with open(path, "rb") as fh:
        raw = fh.read()
    (n,) = struct.unpack("<i", raw[:4])
    xyz = np.frombuffer(raw, dtype="<f8", count=3 * n, offset=4).reshape(n, 3)
    return xyz.astype(np.float64)


def read_xsc(path: str) -> np.ndarray:
    """NAMD/ACEMD extended system file: last non-comment line = step a_x a_y a_z b_x b_y b_z c_x ...;
    returns the orthorhombic diagonal (fields 1, 5, 9)."""
    with open(path) as fh:
        rows = [ln for ln in fh.read().splitlines() if ln.strip() and not ln.startswith("#")]
    t = rows[-1].split()
    return np.array([float(t[1]), float(t[5]), float(t[9])], dtype=np.float64)


# --------------------------------------------------------------------------- AMBER prmtop
_FMT = re.compile(r"%FORMAT\(\s*(\d+)\s*([aAiIeEfF])\s*(\d+)(?:\.(\d+))?\s*\)")


class AmberPrmtop:
    """Raw section access to an AMBER7 prmtop (`%FLAG` / `%FORMAT` fixed-width records)."""

    def __init__(self, path: str):
        self.sections: dict[str, list] = {}
        with open(path) as fh:
            lines = fh.read().splitlines()
        k = 0
        while k < len(lines):
            ln = lines[k]
            if ln.startswith("%FLAG"):
                flag = ln.split()[1]
                k += 1
                while not lines[k].startswith("%FORMAT"):
                    k += 1
                m = _FMT.match(lines[k].strip())
                kind, width = m.group(2).lower(), int(m.group(3))
                k += 1
                vals: list = []
                while k < len(lines) and not lines[k].startswith("%"):
                    row = lines[k]
                    for c in range(0, len(row), width):
                        tok = row[c : c + width]
                        if not tok.strip() and kind != "a":
                            continue
                        if kind == "a":
                            if c + width <= len(row) or tok.strip():
                                vals.append(tok.strip())
                        elif kind == "i":
                            vals.append(int(tok))
                        else:
                            vals.append(float(tok))
                    k += 1
                self.sections[flag] = vals
            else:
                k += 1

    def ints(self, flag):
        return np.asarray(self.sections.get(flag, []), dtype=np.int64)

    def floats(self, flag):
        return np.asarray(self.sections.get(flag, []), dtype=np.float64)

    def strs(self, flag):
        return list(self.sections.get(flag, []))


AMBER_CHARGE_SCALE = 18.2223  # prmtop charges are e * 18.2223 (sqrt of the AMBER Coulomb constant)


def read_prmtop(path: str):
    """Returns (Topology, AmberPrmtop).  Index triplets/quads are stored as coordinate offsets
    (atom index * 3); a negative 4th dihedral index marks an improper, a negative 3rd index marks
    a torsion whose 1-4 pair must be skipped (both are stripped with abs())."""
    top = AmberPrmtop(path)
    natom = int(top.ints("POINTERS")[0])
    types = [t for t in top.strs("AMBER_ATOM_TYPE")][:natom]
    names = [t for t in top.strs("ATOM_NAME")][:natom]

    def table(flags, width):
        parts = [top.ints(f).reshape(-1, width + 1) for f in flags if len(top.ints(f))]
        if not parts:
            return np.zeros((0, width + 1), dtype=np.int64)
        return np.concatenate(parts, axis=0)

    b = table(("BONDS_INC_HYDROGEN", "BONDS_WITHOUT_HYDROGEN"), 2)
    a = table(("ANGLES_INC_HYDROGEN", "ANGLES_WITHOUT_HYDROGEN"), 3)
    d = table(("DIHEDRALS_INC_HYDROGEN", "DIHEDRALS_WITHOUT_HYDROGEN"), 4)
    improper_rows = d[:, 3] < 0
    quad = np.abs(d[:, :4]) // 3
    mol = Topology(
        atomtype=np.array(types, dtype=object),
        charge=(top.floats("CHARGE")[:natom] / AMBER_CHARGE_SCALE).astype(np.float32),
        masses=top.floats("MASS")[:natom].astype(np.float32),
        bonds=b[:, :2] // 3,
        angles=a[:, :3] // 3,
        dihedrals=quad[~improper_rows],
        impropers=quad[improper_rows],
        name=np.array(names, dtype=object),
    )
    return mol, top
