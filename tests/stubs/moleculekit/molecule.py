"""`Molecule(file | [files])` + `.read(file)` with the attributes the reference's driver and `Parameters` read
(`torchmd/run.py:158-216`, `torchmd/parameters.py:109-134`): numAtoms, atomtype, charge, masses, bonds, angles, dihedrals,
impropers, coords [N, 3, 1] float32, box [3, 1], crystalinfo {a, b, c}, element, name."""
import os

import numpy as np

from torchmd_amd import io as tio

_TOPOLOGY_FIELDS = ("atomtype", "charge", "masses", "bonds", "angles", "dihedrals", "impropers", "name", "element")


class Molecule:
    def __init__(self, files=None):
        self.atomtype = np.zeros(0, dtype=object)
        self.charge = np.zeros(0, dtype=np.float32)
        self.masses = np.zeros(0, dtype=np.float32)
        self.bonds = np.zeros((0, 2), dtype=np.int64)
        self.angles = np.zeros((0, 3), dtype=np.int64)
        self.dihedrals = np.zeros((0, 4), dtype=np.int64)
        self.impropers = np.zeros((0, 4), dtype=np.int64)
        self.name = self.element = None
        self.coords = None
        self.box = np.zeros((3, 1), dtype=np.float32)
        self.crystalinfo = None
        if files is not None:
            for f in (files if isinstance(files, (list, tuple)) else [files]):
                self.read(f)

    @property
    def numAtoms(self):
        return len(self.atomtype) if len(self.atomtype) else (0 if self.coords is None else self.coords.shape[0])

    @property
    def numFrames(self):
        return 0 if self.coords is None else self.coords.shape[2]

    def _take_topology(self, top):
        for k in _TOPOLOGY_FIELDS:
            v = getattr(top, k, None)
            if v is not None:
                setattr(self, k, v)

    def read(self, path):
        ext = os.path.splitext(str(path))[1].lower()
        if ext == ".psf":
            self._take_topology(tio.read_psf(path))
        elif ext in (".prmtop", ".parm7"):
            self._take_topology(tio.read_prmtop(path)[0])
        elif ext == ".pdb":
            xyz, box, names, elems = tio.read_pdb(path)
            self.coords = np.ascontiguousarray(xyz[:, :, None].astype(np.float32))
            self.crystalinfo = {"a": float(box[0]), "b": float(box[1]), "c": float(box[2]), "alpha": 90.0, "beta": 90.0, "gamma": 90.0}
            if np.any(box != 0):
                self.box = box.reshape(3, 1).astype(np.float32)
            if self.name is None:
                self.name = names
            if self.element is None:
                self.element = elems
        elif ext == ".coor":
            self.coords = np.ascontiguousarray(tio.read_namd_coor(path)[:, :, None].astype(np.float32))
        elif ext == ".xsc":
            self.box = np.asarray(tio.read_xsc(path), dtype=np.float32).reshape(3, 1)
        else:
            raise ValueError(f"moleculekit stub: unsupported file '{path}'")
        if self.element is None and len(self.atomtype):
            src = self.name if self.name is not None else self.atomtype
            self.element = np.array([str(n)[:1] for n in src], dtype=object)
