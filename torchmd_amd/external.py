"""Zero-patch boundary: the reference's own plugin hook.

`torchmd/run.py:185-209` builds `importlib.import_module(args.external["module"]).External(file,
embeddings, device=device, **rest)` and `Forces.compute` adds what `External.calculate(pos, box)` returns —
`(energy[R], forces[R,N,3])` — to its own terms (`torchmd/forces.py:321-326`).  This class puts the HIP
nonbonded engine behind that hook, so an UNMODIFIED reference installation runs its pair terms on the
MI355X: keep the bonded terms in `forceterms` and move the nonbonded ones here, e.g.

    forceterms: [bonds, angles, dihedrals, impropers, 1-4]
    external:
      module: torchmd_amd.external
      file: nonbonded.yaml        # see below
      embeddings: [0]             # unused (the hook requires the key)

with `nonbonded.yaml`:

    topology: structure.prmtop     # or .psf
    forcefield: structure.prmtop   # or a .yaml force field
    terms: [lj, electrostatics]
    cutoff: 9.0
    rfa: true
    switch_dist: 7.5               # optional
    exclusions: [bonds, angles, 1-4]

The energy shows up under the reference's `"external"` key.  Programmatic use: pass a `Parameters` object
instead of the file name, options as keyword arguments.
"""

from __future__ import annotations

import os

import torch
import yaml

from .forces import Forces

_OPTIONS = ("terms", "cutoff", "rfa", "switch_dist", "solventDielectric", "exclusions", "skin", "algorithm",
            "switch_mode")


class External:
    def __init__(self, file, embeddings=None, device="cuda", **options):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("torchmd_amd.external.External runs on a ROCm device only (device='cuda')")
        if isinstance(file, (str, os.PathLike)):
            conf, parameters = self._from_file(os.fspath(file))
            conf.update(options)
        else:
            conf, parameters = dict(options), file
        unknown = sorted(set(conf) - set(_OPTIONS))
        if unknown:
            raise ValueError(f"unknown option(s) for torchmd_amd.external.External: {unknown}")
        terms = [t.lower() for t in conf.pop("terms", ("lj", "electrostatics"))]
        bad = [t for t in terms if t not in Forces.nonbonded]
        if bad:
            raise ValueError(f"External evaluates nonbonded terms only {Forces.nonbonded}, got {bad}")
        if "exclusions" in conf:
            conf["exclusions"] = tuple(conf["exclusions"])
        self.terms = terms
        self.forces = Forces(parameters, terms=terms, **conf)
        self._buf = None

    @staticmethod
    def _from_file(path):
        from .forcefields import ForceField
        from .io import read_prmtop, read_psf
        from .parameters import Parameters

        with open(path) as fh:
            conf = yaml.safe_load(fh) or {}
        base = os.path.dirname(os.path.abspath(path))
        try:
            top, ff = conf.pop("topology"), conf.pop("forcefield")
        except KeyError as e:
            raise ValueError(f"{path}: missing key {e.args[0]!r} (needs 'topology' and 'forcefield')") from None
        top, ff = (p if os.path.isabs(p) else os.path.join(base, p) for p in (top, ff))
        ext = os.path.splitext(top)[-1].lower()
        if ext == ".psf":
            mol = read_psf(top)
        elif ext in (".prmtop", ".parm7"):
            mol = read_prmtop(top)
            mol = mol[0] if isinstance(mol, tuple) else mol
        else:
            raise ValueError(f"{path}: unsupported topology '{top}' (.psf, .prmtop)")
        wanted = [t.lower() for t in conf.get("terms", ("lj", "electrostatics"))]
        par = Parameters(ForceField.create(mol, ff), mol, terms=wanted + ["bonds", "angles", "dihedrals", "1-4"])
        return conf, par

    def calculate(self, pos, box):
        """`(energy[R], forces[R,N,3])` of the configured nonbonded terms, on `pos`'s device and dtype."""
        if pos.device.type != "cuda":
            raise RuntimeError("External.calculate needs ROCm device tensors")
        p = pos.detach()
        if p.dtype not in (torch.float32, torch.float64):
            p = p.float()
        if not p.is_contiguous():
            p = p.contiguous()
        if self._buf is None or self._buf.shape != p.shape or self._buf.dtype != p.dtype or self._buf.device != p.device:
            self._buf = torch.empty_like(p)
        ebuf = self.forces._evaluate(p, box.to(p.dtype), self._buf, True, True)
        energy = self.forces.total_energy_from(ebuf, None).to(pos.dtype)
        return energy, self._buf.to(pos.dtype)
