#!/usr/bin/env python
"""Micro-timing of the force path on the C3 water box (needs a GPU):
   forced neighbour-list rebuild vs steady-state force evaluation, for a few skins.

    python tools/time_kernels.py [--nside 32] [--skins 0.6 0.8 1.0 1.5]
"""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from bench import build_system  # noqa: E402
from torchmd_amd import _lib as L  # noqa: E402
from torchmd_amd.forces import Forces  # noqa: E402
from torchmd_amd.integrator import Integrator  # noqa: E402


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nside", type=int, default=32)
    ap.add_argument("--skins", type=float, nargs="*", default=[0.6, 0.8, 1.0, 1.5])
    ap.add_argument("--relax", type=int, default=600)
    ap.add_argument("--no-md", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    mol, par, system, forces, box = build_system(args.nside, dev, torch.float32, seed=1)
    forces.compute(system.pos, system.box, system.forces)
    Integrator(system, forces, 1.0, dev, gamma=10.0, T=300.0).step(args.relax)
    for skin in args.skins:
        f = Forces(par, terms=["lj", "electrostatics"], cutoff=9.0, rfa=True, skin=skin)
        F = torch.zeros_like(system.pos)
        f._evaluate(system.pos, system.box, F, False, True)
        eng = f._engine(system.pos)
        steady = timed(lambda: f._evaluate(system.pos, system.box, F, False, True), 200)

        def forced():
            L.check(eng.lib.tmdhip_invalidate_list(eng.ctx, 0))
            f._evaluate(system.pos, system.box, F, False, True)

        rebuild = timed(forced, 30)
        f.enable_timing(system.pos, True)
        for _ in range(50):
            f._evaluate(system.pos, system.box, F, False, True)
        ms, n = f.read_timing(system.pos)
        st = f.stats(system.pos)
        print(f"skin {skin:4.2f}: steady force call {steady:7.1f} us | with forced rebuild {rebuild:7.1f} us "
              f"(rebuild chain ~{rebuild - steady:6.1f} us) | pair kernel {ms / n * 1e3:6.1f} us | "
              f"entries/atom {st['list_entries'] / mol.numAtoms:6.1f} cap {st['max_neighbours']} ncell {st['ncell']}")
        if args.no_md:
            f.close()
            continue
        # dynamics: steps per rebuild at this skin
        s2 = type(system)(mol.numAtoms, 1, torch.float32, dev)
        s2.pos[:] = system.pos
        s2.vel[:] = system.vel
        s2.box[:] = system.box
        f2 = Forces(par, terms=["lj", "electrostatics", "bonds", "angles"], cutoff=9.0, rfa=True, skin=skin)
        f2.compute(s2.pos, s2.box, s2.forces)
        integ = Integrator(s2, f2, 1.0, dev, gamma=0.1, T=300.0)
        integ.step(100)
        r0 = f2.stats(s2.pos)["n_rebuilds"]
        t = timed(lambda: integ.step(400), 1) / 400
        r1 = f2.stats(s2.pos)["n_rebuilds"]
        print(f"           MD: {t:7.1f} us/step, {400 / max(r1 - r0, 1):5.1f} steps per rebuild")
        f.close()
        f2.close()


if __name__ == "__main__":
    main()
