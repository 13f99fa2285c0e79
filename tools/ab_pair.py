#!/usr/bin/env python
"""A/B timing of differently built libraries on the relaxed C3 box (needs a GPU).

    python -m torchmd_amd._build -DTMD_EXP=2 --out=torchmd_amd/lib/exp/libtmdhip_e2.so
    python tools/ab_pair.py torchmd_amd/lib/exp/libtmdhip_e0.so torchmd_amd/lib/exp/libtmdhip_e2.so [--rounds 3]

Every library runs in its own process (TMDHIP_LIB), alternating over `--rounds` so that clock / box drift hits all of
them alike.  Per run: pair-kernel time from HIP events attached to every 4th launch of a 1 200-step MD run, the wall
time per step of that run, the list build time (forced rebuilds), and a checksum of the forces after a fixed
trajectory (variants that must not change results have to agree on it)."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(args):
    sys.path.insert(0, ROOT)
    import torch

    from bench import build_system
    from torchmd_amd import _lib as L
    from torchmd_amd.integrator import Integrator

    dev = torch.device("cuda:0")
    mol, par, system, forces, box = build_system(args.nside, dev, torch.float32, seed=1)
    forces.compute(system.pos, system.box, system.forces)
    if args.static:  # no dynamics (debug builds with wrong results): the pair kernel on the start configuration only
        from torchmd_amd.forces import Forces
        forces = Forces(par, terms=["lj", "electrostatics"], cutoff=9.0, rfa=True)  # (no bonded kernel adding into F)
        F = torch.zeros_like(system.pos)
        forces._evaluate(system.pos, system.box, F, False, True)
        forces.enable_timing(system.pos, True, every=1)
        forces.read_timing(system.pos, reset=True)
        for _ in range(200):
            forces._evaluate(system.pos, system.box, F, False, True)
        ms, n = forces.read_timing(system.pos, reset=True)
        if args.dump:  # debug builds that store a wave timeline in the force buffer
            import numpy as np
            np.save(args.dump, F.detach().cpu().numpy())
        print("ABRESULT " + json.dumps({"pair_us": ms / max(n, 1) * 1e3, "step_us": 0.0, "eval_us": 0.0, "eval_rebuild_us": 0.0,
                                        "rebuilds": 0, "entries": int(forces.stats(system.pos)["list_entries"]),
                                        "chains_skipped": 0, "force_checksum": float(F.double().abs().sum().item())}), flush=True)
        return
    Integrator(system, forces, 1.0, dev, gamma=10.0, T=300.0).step(args.relax)
    integ = Integrator(system, forces, 1.0, dev, gamma=0.1, T=300.0)
    integ.step(200)
    csum = float(system.forces.double().abs().sum().item())  # after relax + 200 steps: same trajectory in every variant
    out = {"lib": os.environ.get("TMDHIP_LIB", "default"), "force_checksum": csum}
    forces.enable_timing(system.pos, True, every=4)
    forces.read_timing(system.pos, reset=True)
    st0 = forces.stats(system.pos)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    integ.step(args.steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms, n = forces.read_timing(system.pos, reset=True)
    forces.enable_timing(system.pos, False)
    st1 = forces.stats(system.pos)
    out.update(pair_us=ms / max(n, 1) * 1e3, step_us=el / args.steps * 1e6, launches=int(n),
               rebuilds=int(st1["n_rebuilds"] - st0["n_rebuilds"]), entries=int(st1["list_entries"]),
               chains_skipped=int(st1["chains_skipped"] - st0["chains_skipped"]))
    # list build: forced rebuild + evaluation minus a steady evaluation (plain compute path)
    eng = forces._engine(system.pos)
    F = torch.zeros_like(system.pos)

    def timed(fn, k):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / k * 1e6

    steady = timed(lambda: forces._evaluate(system.pos, system.box, F, False, True), 100)

    def forced():
        L.check(eng.lib.tmdhip_invalidate_list(eng.ctx, 0))
        forces._evaluate(system.pos, system.box, F, False, True)

    out.update(eval_us=steady, eval_rebuild_us=timed(forced, 20))
    print("ABRESULT " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=1200)
    ap.add_argument("--relax", type=int, default=600)
    ap.add_argument("--nside", type=int, default=32)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--static", action="store_true", help="time the pair kernel on the start configuration, no MD")
    ap.add_argument("--dump", default=None, help="with --static: save the force buffer of the last evaluation (.npy)")
    ap.add_argument("--env", action="append", default=[], help="NAME=VALUE for every child (repeatable)")
    args = ap.parse_args()
    if args.child:
        return child(args)
    res = {}
    for rnd in range(args.rounds):
        for lib in args.libs:
            env = dict(os.environ)
            if lib != "default":
                env["TMDHIP_LIB"] = os.path.abspath(lib)
            for kv in args.env:
                k, v = kv.split("=", 1)
                env[k] = v
            p = subprocess.run(["timeout", "90", sys.executable, os.path.abspath(__file__), "--child", "--steps", str(args.steps), "--relax",
                                str(args.relax), "--nside", str(args.nside)] + (["--static"] if args.static else [])
                               + (["--dump", args.dump] if args.dump else []),
                               env=env, capture_output=True, text=True)
            line = next((l for l in p.stdout.splitlines() if l.startswith("ABRESULT ")), None)
            if line is None:
                print(f"{lib}: FAILED rc={p.returncode}\n{p.stdout[-2000:]}\n{p.stderr[-2000:]}", flush=True)
                continue
            r = json.loads(line[len("ABRESULT "):])
            res.setdefault(lib, []).append(r)
            print(f"round {rnd} {os.path.basename(lib):28s} pair {r['pair_us']:6.2f} us  step {r['step_us']:6.2f} us  "
                  f"eval {r['eval_us']:6.1f}  eval+rebuild {r['eval_rebuild_us']:6.1f}  rebuilds {r['rebuilds']}  "
                  f"entries {r['entries']}  skipped {r['chains_skipped']}  checksum {r['force_checksum']:.6e}", flush=True)
    print("--- best of rounds ---")
    for lib, rs in res.items():
        print(f"{os.path.basename(lib):28s} pair {min(r['pair_us'] for r in rs):6.2f} us  step {min(r['step_us'] for r in rs):6.2f} us  "
              f"rebuild chain {min(r['eval_rebuild_us'] - r['eval_us'] for r in rs):6.1f} us")


if __name__ == "__main__":
    main()
