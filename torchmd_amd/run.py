"""MD driver with the reference's configuration surface (`torchmd/run.py:30-144`: same option names,
`--conf file.yaml`), built on this package's readers instead of moleculekit:

    python -m torchmd_amd.run --conf tests/water/water_conf.yaml [--device cuda:0] [--steps N]

Inputs: `structure: [file.psf, file.pdb]` (CHARMM) or `topology: file.prmtop` + `coordinates:
file.coor|.pdb` + `extended_system: file.xsc` (AMBER), `forcefield: *.yaml | *.prmtop`.
Outputs like the reference (`run.py:230-291`): `monitor_{k}.csv` (iter, ns, epot, ekin, etot, T, t),
`{output}_{k}.npy` trajectory `[N,3,frames]`, `input.yaml` echo.  Frames are staged through a pinned
host ring with asynchronous copies instead of a blocking `.cpu()` per period.
"""

from __future__ import annotations

import argparse
import csv
import os
import time

import numpy as np
import torch
import yaml

from . import io as tio
from .forcefields import ForceField
from .forces import Forces
from .integrator import Integrator, maxwell_boltzmann
from .minimizers import minimize_bfgs
from .utils import LogWriter
from .parameters import Parameters
from .systems import System
from .wrapper import Wrapper

FS2NS = 1e-6
PRECISION = {"single": torch.float, "double": torch.double}

DEFAULTS = dict(
    timestep=1.0, temperature=300.0, langevin_temperature=0.0, langevin_gamma=0.1, device="cuda:0",
    structure=None, topology=None, coordinates=None, forcefield=None, seed=1, output_period=10,
    save_period=0, steps=10000, log_dir="./", output="output", forceterms=["LJ"], cutoff=None,
    switch_dist=None, precision="single", external=None, rfa=False, replicas=1, extended_system=None,
    minimize=None, exclusions=("bonds", "angles", "1-4"),
)


def get_args(arguments=None):
    ap = argparse.ArgumentParser(description="TorchMD on MI355X (torchmd_amd)")
    ap.add_argument("--conf", default=None, help="YAML configuration file (same keys as the options)")
    for key, val in DEFAULTS.items():
        opt = "--" + key.replace("_", "-")
        if isinstance(val, bool):
            ap.add_argument(opt, dest=key, action="store_true", default=None)
        elif key in ("forceterms", "structure"):
            ap.add_argument(opt, dest=key, nargs="+", default=None)
        elif key in ("external", "exclusions"):
            continue
        else:
            ap.add_argument(opt, dest=key, default=None, type=type(val) if val is not None else str)
    ns = ap.parse_args(arguments)
    cfg = dict(DEFAULTS)
    if ns.conf:
        with open(ns.conf) as fh:
            cfg.update({k: v for k, v in (yaml.safe_load(fh) or {}).items()})
    cfg.update({k: v for k, v in vars(ns).items() if k != "conf" and v is not None})
    args = argparse.Namespace(**cfg)
    for k in ("cutoff", "switch_dist", "timestep", "temperature", "langevin_temperature", "langevin_gamma"):
        if getattr(args, k) is not None:
            setattr(args, k, float(getattr(args, k)))
    for k in ("steps", "output_period", "save_period", "replicas", "seed"):
        setattr(args, k, int(getattr(args, k)))
    if isinstance(args.forceterms, str):
        args.forceterms = [args.forceterms]
    if args.forceterms is None:
        args.forceterms = []
    if str(args.device) == "cuda":
        args.device = "cuda:0"
    if args.steps % args.output_period != 0:
        raise ValueError("Steps must be multiple of output-period.")
    if args.save_period == 0:
        args.save_period = 10 * args.output_period
    if args.save_period % args.output_period != 0:
        raise ValueError("save-period must be multiple of output-period.")
    os.makedirs(args.log_dir, exist_ok=True)
    with open(os.path.join(args.log_dir, "input.yaml"), "w") as fh:
        yaml.safe_dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(args).items()}, fh)
    return args


def load_molecule(args):
    """Topology + coordinates + box from the file kinds the reference's configs use."""
    files = []
    for f in (args.topology, args.structure, args.coordinates, args.extended_system):
        if f is None:
            continue
        files += list(f) if isinstance(f, (list, tuple)) else [f]
    mol, coords, box, prmtop = None, None, np.zeros(3), None
    for f in files:
        ext = os.path.splitext(f)[1].lower()
        if ext == ".psf":
            mol = tio.read_psf(f)
        elif ext in (".prmtop", ".parm7"):
            mol, prmtop = tio.read_prmtop(f)
        elif ext == ".pdb":
            xyz, pbox, names, elems = tio.read_pdb(f)
            coords = xyz.astype(np.float64)
            if np.any(pbox != 0):
                box = pbox.astype(np.float64)
            if mol is not None and mol.element is None:
                mol.element = elems
        elif ext == ".coor":
            coords = tio.read_namd_coor(f)
        elif ext == ".xsc":
            box = tio.read_xsc(f)
        else:
            raise RuntimeError(f"unsupported input file '{f}'")
    if mol is None or coords is None:
        raise RuntimeError("need a topology (.psf / .prmtop) and coordinates (.pdb / .coor)")
    if len(coords) != mol.numAtoms:
        raise RuntimeError("coordinate and topology atom counts differ")
    mol.coords = coords[:, :, None].astype(np.float32)
    mol.box = np.asarray(box, dtype=np.float64)
    if mol.element is None:
        mol.element = np.array([str(n)[:1] for n in (mol.name if mol.name is not None else mol.atomtype)], dtype=object)
    return mol, prmtop


def load_external(conf, replicas, device):
    """The reference's plugin hook (`torchmd/run.py:185-209`): `conf = {module, file, embeddings, ...}` ->
    `import_module(module).External(file, embeddings, device=device, **rest)`; `embeddings` is a list or the
    name of a `.npy` file, repeated per replica.  The object's `calculate(pos, box)` must return
    `(energy[R], forces[R,N,3])` (consumed by `Forces.compute`, reference `forces.py:321-326`)."""
    if conf is None:
        return None
    import importlib

    conf = dict(conf)
    try:
        module, file = conf.pop("module"), conf.pop("file")
    except KeyError as e:
        raise ValueError(f"external: missing key {e.args[0]!r} (needs 'module' and 'file')") from None
    emb = conf.pop("embeddings", None)
    if isinstance(emb, str):
        emb = np.load(emb).astype(int)
    embeddings = None if emb is None else torch.tensor(emb).repeat(replicas, 1)
    return importlib.import_module(module).External(file, embeddings, device=device, **conf)


def setup(args):
    torch.manual_seed(args.seed)
    device = torch.device(args.device)
    mol, prmtop = load_molecule(args)
    precision = PRECISION[args.precision]
    ff_src = args.forcefield
    if prmtop is not None and (ff_src is None or str(ff_src).endswith((".prmtop", ".parm7"))):
        from .forcefields import PrmtopForceField

        ff = PrmtopForceField(mol, prmtop)
    else:
        ff = ForceField.create(mol, ff_src)
    terms = args.forceterms if args.forceterms else ["bonds", "angles", "dihedrals", "impropers", "1-4", "electrostatics", "lj"]
    print("Force terms: ", terms)
    parameters = Parameters(ff, mol, terms, precision=precision, device=device)  # (on the device, as run.py:181-183 does)
    external = load_external(args.external, args.replicas, device)
    system = System(mol.numAtoms, args.replicas, precision, device)
    system.set_positions(mol.coords)
    system.set_box(mol.box)
    system.set_velocities(maxwell_boltzmann(parameters.masses, args.temperature, args.replicas))
    forces = Forces(parameters, terms=terms, external=external, cutoff=args.cutoff, rfa=args.rfa,
                    switch_dist=args.switch_dist, exclusions=tuple(args.exclusions))
    return mol, system, forces


class FrameStager:
    """Trajectory frames leave the device through a pinned host ring with asynchronous copies on a side
    stream; the step loop never blocks on them (the reference does a blocking `.cpu()` each period and
    re-saves the whole trajectory with np.save, `run.py:267-274`)."""

    def __init__(self, system, nframes):
        R, N = system.pos.shape[0], system.pos.shape[1]
        self.buf = torch.empty((nframes, R, N, 3), dtype=system.pos.dtype).pin_memory()
        self.stream = torch.cuda.Stream(device=system.pos.device)
        self.events = []
        self.count = 0

    def push(self, pos):
        snap = pos.detach().clone()  # the integrator keeps mutating pos
        ready = torch.cuda.Event()   # recorded AFTER the clone: the side stream's copy must see the snapshot
        ready.record(torch.cuda.current_stream(pos.device))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            self.buf[self.count].copy_(snap, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.stream)
        snap.record_stream(self.stream)
        self.events.append(done)
        self.count += 1

    def frames(self, replica):
        """[N,3,frames] numpy array of the frames copied so far (reference layout)."""
        for ev in self.events:
            ev.synchronize()
        return np.ascontiguousarray(self.buf[: self.count, replica].numpy().transpose(1, 2, 0))


def dynamics(args, mol, system, forces):
    torch.manual_seed(args.seed)
    device = torch.device(args.device)
    integrator = Integrator(system, forces, args.timestep, device, gamma=args.langevin_gamma,
                            T=args.langevin_temperature)
    wrapper = Wrapper(mol.numAtoms, mol.bonds if len(mol.bonds) else None, device)
    nper = args.steps // args.output_period
    stager = FrameStager(system, nper)
    logs = [LogWriter(args.log_dir, ("iter", "ns", "epot", "ekin", "etot", "T"), name=f"monitor_{k}.csv")
            for k in range(args.replicas)]
    if args.minimize is not None:
        minimize_bfgs(system, forces, steps=int(args.minimize))
    forces.compute(system.pos, system.box, system.forces)
    name, ext = os.path.splitext(args.output)
    t0 = time.time()
    for i in range(1, nper + 1):
        Ekin, Epot, T = integrator.step(niter=args.output_period)
        wrapper.wrap(system.pos, system.box)
        stager.push(system.pos)
        for k in range(args.replicas):
            if (i * args.output_period) % args.save_period == 0 or i == nper:
                np.save(os.path.join(args.log_dir, f"{name}_{k}{ext or '.npy'}"), stager.frames(k))
            logs[k].write_row({"iter": i * args.output_period, "ns": FS2NS * i * args.output_period * args.timestep,
                               "epot": Epot[k], "ekin": float(Ekin[k]), "etot": Epot[k] + float(Ekin[k]),
                               "T": float(T[k])})
    wall = time.time() - t0
    print(f"{args.steps} steps in {wall:.2f} s = {args.steps * args.timestep * FS2NS / wall * 86400:.1f} ns/day per replica")
    return stager


def main(arguments=None):
    args = get_args(arguments)
    mol, system, forces = setup(args)
    dynamics(args, mol, system, forces)


if __name__ == "__main__":
    main()
