"""Helpers shared by the tests: load tests/golden/*.npz and rebuild a `Parameters`-shaped object."""

import os
from types import SimpleNamespace

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PREC = {"f64": torch.float64, "f32": torch.float32}


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))


class GoldenParameters:
    """Duck type of reference `torchmd.parameters.Parameters` rebuilt from stored arrays."""

    def __init__(self, g, precision=torch.float64, device="cpu", prefix="par_"):
        def t(key, dtype=None):
            if prefix + key not in g:
                return None
            x = torch.tensor(g[prefix + key])
            return x.to(dtype) if dtype is not None else x

        self.charges = t("charges", precision)
        self.masses = t("masses", precision)
        self.mapped_atom_types = t("types")
        self.natoms = len(self.charges)
        self.device = device
        nb = t("nonbonded_params", precision)
        self.nonbonded_params = None if nb is None else {"params": nb}
        for name in ("bond", "angle", "dihedral", "improper", "nonbonded_14"):
            idx = t(name + "_idx")
            if idx is None:
                setattr(self, name + "_params", None)
            else:
                setattr(
                    self,
                    name + "_params",
                    {"idx": idx, "map": t(name + "_map"), "params": t(name + "_params", precision)},
                )
        self.A = self.B = None
        self.to_(device)

    def to_(self, device):
        self.charges = self.charges.to(device)
        self.masses = self.masses.to(device)
        self.mapped_atom_types = self.mapped_atom_types.to(device)
        for name in ("nonbonded", "bond", "angle", "dihedral", "improper", "nonbonded_14"):
            tab = getattr(self, name + "_params")
            if tab is not None:
                for k in tab:
                    tab[k] = tab[k].to(device)
        self.device = device

    def get_AB(self):
        s, e = self.nonbonded_params["params"][:, 0], self.nonbonded_params["params"][:, 1]
        sig6 = (0.5 * (s + s[:, None])) ** 6
        eps4 = torch.sqrt(e * e[:, None]) * 4
        return eps4 * sig6 * sig6, eps4 * sig6

    def get_exclusions(self, types=("bonds", "angles", "1-4"), fullarray=False):
        ex = []
        if self.bond_params is not None and "bonds" in types:
            ex += self.bond_params["idx"].cpu().numpy().tolist()
        if self.angle_params is not None and "angles" in types:
            ex += self.angle_params["idx"].cpu().numpy()[:, [0, 2]].tolist()
        if self.dihedral_params is not None and "1-4" in types:
            ex += self.dihedral_params["idx"].cpu().numpy()[:, [0, 3]].tolist()
        return ex


def energies(g, tag, replica=0):
    pre = f"{tag}_E{replica}_"
    return {k[len(pre):]: float(v) for k, v in g.items() if k.startswith(pre)}


def box_tensor(box3, R, dtype, device="cpu"):
    b = torch.zeros(R, 3, 3, dtype=dtype, device=device)
    for r in range(R):
        b[r].diagonal().copy_(torch.as_tensor(box3, dtype=dtype))
    return b


def pos_tensor(pos, R, dtype, device="cpu"):
    p = torch.as_tensor(np.asarray(pos, dtype=np.float64)).to(dtype)
    return p[None].repeat(R, 1, 1).contiguous().to(device)
