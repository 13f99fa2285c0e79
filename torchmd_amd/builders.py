"""Synthetic benchmark systems: the TIP3P water box of config C3 and the
Lennard-Jones (argon) box of config C5, as `Topology` + force-field objects that go through the same
`Parameters` -> `Forces` -> `Integrator` path as a real input.
"""

from __future__ import annotations

import numpy as np

from .forcefields import YamlForceField
from .io import Topology

# flexible TIP3P exactly as the reference's tests/water/water_forcefield.yaml (note: hydrogens carry LJ)
TIP3P_FF = {
    "atomtypes": ["OT", "HT"],
    "bonds": {"(OT, HT)": {"k0": 450.0, "req": 0.9572}, "(HT, HT)": {"k0": 0.0, "req": 1.5139}},
    "angles": {"(HT, OT, HT)": {"k0": 55.0, "theta0": 104.52}},
    "lj": {
        "OT": {"sigma": 3.150574226831496, "epsilon": -0.1521},
        "HT": {"sigma": 0.40001352444501237, "epsilon": -0.046},
    },
    "electrostatics": {"OT": {"charge": -0.834}, "HT": {"charge": 0.417}},
    "masses": {"OT": 15.9994, "HT": 1.008},
}

# argon as the reference's tests/argon/argon_forcefield.yaml:8-18
ARGON_FF = {
    "atomtypes": ["AR"],
    "bonds": {"(AR, AR)": {"k0": 0, "req": 0}},
    "lj": {"AR": {"sigma": 3.345, "epsilon": 0.238}},
    "electrostatics": {"AR": {"charge": 0.0}},
    "masses": {"AR": 39.95},
}


def _random_rotations(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack(
        [
            np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], axis=1),
            np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], axis=1),
            np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=1),
        ],
        axis=1,
    )


def tip3p_box(nside=32, seed=0, density=0.0334, jitter=0.2):
    """nside^3 TIP3P molecules on a cubic lattice (spacing (1/density)^(1/3) A), oxygen at the
    lattice site + spacing/2 + U(-jitter, jitter), rigid geometry (r_OH 0.9572 A, HOH 104.52 deg) with
    uniformly random orientation; atom order O,H1,H2; bonds O-H1, O-H2, H1-H2 and angle H1-O-H2 as in
    tests/water/structure.psf.  nside=32 -> N = 98 304, L = 99.365 A (config C3).
    Returns (Topology, pos [N,3] float64, box [3] float64)."""
    rng = np.random.default_rng(seed)
    a = (1.0 / density) ** (1.0 / 3.0)
    L = a * nside
    g = np.arange(nside)
    sites = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float64)
    nmol = len(sites)
    oxy = sites * a + a / 2 + rng.uniform(-jitter, jitter, size=(nmol, 3))
    r, th = 0.9572, np.deg2rad(104.52)
    h1 = np.array([r * np.sin(th / 2), 0.0, r * np.cos(th / 2)])
    h2 = np.array([-r * np.sin(th / 2), 0.0, r * np.cos(th / 2)])
    rot = _random_rotations(rng, nmol)
    pos = np.empty((nmol, 3, 3))
    pos[:, 0] = oxy
    pos[:, 1] = oxy + rot @ h1
    pos[:, 2] = oxy + rot @ h2
    pos = pos.reshape(-1, 3)
    base = 3 * np.arange(nmol)
    bonds = np.stack([np.stack([base, base + 1], 1), np.stack([base, base + 2], 1), np.stack([base + 1, base + 2], 1)], 1)
    angles = np.stack([base + 1, base, base + 2], axis=1)
    mol = Topology(
        atomtype=np.tile(np.array(["OT", "HT", "HT"], dtype=object), nmol),
        charge=np.tile(np.array([-0.834, 0.417, 0.417], dtype=np.float32), nmol),
        masses=np.tile(np.array([15.9994, 1.008, 1.008], dtype=np.float32), nmol),
        bonds=bonds.reshape(-1, 2).astype(np.int64),
        angles=angles.astype(np.int64),
    )
    return mol, pos, np.array([L, L, L])


def lj_box(nside=100, seed=0, density=0.0213, jitter=0.3):
    """nside^3 argon atoms on a jittered simple-cubic lattice at liquid density (config C5:
    nside=100 -> 1e6 atoms, L = 360.8 A).  Returns (Topology, pos, box)."""
    rng = np.random.default_rng(seed)
    a = (1.0 / density) ** (1.0 / 3.0)
    L = a * nside
    g = np.arange(nside)
    sites = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float64)
    pos = sites * a + a / 2 + rng.uniform(-jitter, jitter, size=sites.shape)
    n = len(pos)
    mol = Topology(
        atomtype=np.full(n, "AR", dtype=object),
        charge=np.zeros(n, dtype=np.float32),
        masses=np.full(n, 39.95, dtype=np.float32),
    )
    return mol, pos, np.array([L, L, L])


def water_forcefield(mol):
    return YamlForceField(mol, TIP3P_FF)


def argon_forcefield(mol):
    return YamlForceField(mol, ARGON_FF)
