#!/usr/bin/env python
"""Soak of the three parity checks that failed on ONE box of the pool in round 4 and nowhere else (DESIGN "an anomaly
that did not reproduce"): is it a bad box or a race in the library?

    python tools/soak.py [--inproc 200] [--fresh 20] [--out gpurun_out/soak]

* first `tools/ubench/sanity` (plain HIP, nothing of libtmdhip: known-answer fp32 atomics from all XCDs, a copy
  pattern, an ALU chain, an LDS transpose) — a box that fails THAT is a bad box;
* then the three checks, mirrored from tests/test_gpu_parity.py with their CPU references computed once:
    thrombin   4 676-atom complex: all-pairs no-cutoff (fp64) vs the reference's golden, all seven terms vs the golden,
               cell-list vs all-pairs vs the oracle at 9 A / switch / reaction field, in-cutoff pair count
    repulsion  5 184-atom water box, `repulsion` term, fp32, cell-list path vs the oracle (generic kernel, float atomics)
    vmap       torch.vmap(forces.compute) over 4 copies of alanine dipeptide + backward, fp64, vs the golden
  `--inproc N` times each in ONE process (contexts created and destroyed N times), once with TMDHIP_DEBUG_POISON=1
  (fresh device buffers filled with 0xFF) and once without, and `--fresh M` times as fresh processes (half with poison).
  Every process prints the box header; a failing iteration dumps its inputs and both outputs to <out>_fail_*.npz.
"""
import argparse
import json
import os
import subprocess
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def box_header():
    import torch

    p = torch.cuda.get_device_properties(0)
    host = os.uname().nodename
    return (f"box: host {host}, device {p.name}, {p.multi_processor_count} CUs, {p.total_memory / 2**30:.0f} GiB, "
            f"arch {getattr(p, 'gcnArchName', '?')}, host cores {os.cpu_count()}, poison {os.environ.get('TMDHIP_DEBUG_POISON', '0')}")


class Refs:
    """CPU references of the three checks (oracle results are cached in a file: fresh processes reuse them)."""

    def __init__(self, cache):
        import numpy as np
        import torch
        from _golden import GoldenParameters, box_tensor, load, pos_tensor
        from oracle import torchmd_oracle as orc
        from torchmd_amd.builders import tip3p_box, water_forcefield
        from torchmd_amd.parameters import Parameters

        self.g_thr = load("thrombin")
        self.par_thr = GoldenParameters(self.g_thr, torch.float64)
        self.g_ala = load("ala2")
        self.par_ala = GoldenParameters(self.g_ala, torch.float64)
        mol, pos, box = tip3p_box(12, seed=13)
        self.w_pos, self.w_box = pos, box
        self.par_w = Parameters(water_forcefield(mol), mol, ["lj", "electrostatics", "bonds", "angles"], precision=torch.float32)
        if cache and os.path.exists(cache):
            c = dict(np.load(cache))
        else:
            zero = np.zeros(3)
            kw = dict(cutoff=9.0, rfa=True, switch_dist=7.5)
            pairs = orc.candidate_pairs(self.g_thr["pos"], zero, 9.5, orc.exclusion_pairs(self.par_thr))
            po, Fo, npairs = orc.compute(self.par_thr, pos_tensor(self.g_thr["pos"], 1, torch.float64), box_tensor(zero, 1, torch.float64),
                                         ["electrostatics", "lj"], pairs=pairs, **kw)
            c = {"thr_F": Fo.numpy(), "thr_E": np.array([po[0]["electrostatics"], po[0]["lj"]]), "thr_n": np.array(npairs)}
            pairs = orc.candidate_pairs(pos, box, 9.6, orc.exclusion_pairs(self.par_w))
            po, Fo, npairs = orc.compute(self.par_w, pos_tensor(pos, 1, torch.float32), box_tensor(box, 1, torch.float32), ["repulsion"],
                                         pairs=pairs, cutoff=9.0)
            c.update({"rep_F": Fo.numpy(), "rep_E": np.array([po[0]["repulsion"]]), "rep_n": np.array(npairs)})
            if cache:
                np.savez(cache, **c)
        self.c = c


def dump(out, case, it, **arrays):
    import numpy as np

    path = f"{out}_fail_{case}_{os.getpid()}_{it}.npz"
    np.savez(path, **{k: np.asarray(v) for k, v in arrays.items()})
    return path


def case_thrombin(R, out, it):
    import numpy as np
    import torch
    from _golden import PREC, box_tensor, energies, pos_tensor
    from torchmd_amd.forces import Forces

    dev = torch.device("cuda:0")
    g, par = R.g_thr, R.par_thr
    zero = np.zeros(3)
    all_terms = ["bonds", "angles", "dihedrals", "impropers", "1-4", "electrostatics", "lj"]
    nb = ["electrostatics", "lj"]
    bad = []

    def run(terms, **kw):
        f = Forces(par, terms=terms, **kw)
        p = pos_tensor(g["pos"], 1, torch.float64, dev)
        b = box_tensor(zero, 1, torch.float64, dev)
        F = torch.full_like(p, 7.0)
        pots = f.compute(p, b, F, returnDetails=True)
        return pots, F.cpu().numpy(), f, p, b

    for tag, terms in (("f64_nb_nocut", nb), ("f64_full_nocut", all_terms)):
        pots, F, f, *_ = run(terms)
        err = float(np.abs(F - g[tag + "_forces"]).max())
        ref = energies(g, tag, 0)
        eerr = max(abs(pots[0][t] - ref[t]) / max(1.0, abs(ref[t])) for t in terms if t != "1-4")
        if not (err <= 1e-8 and eerr <= 3e-10):
            bad.append((tag, err, eerr, dump(out, "thrombin_" + tag, it, pos=g["pos"], F_gpu=F, F_ref=g[tag + "_forces"])))
        f.close()
    kw = dict(cutoff=9.0, rfa=True, switch_dist=7.5)
    pots_c, F_c, fc, p, b = run(nb, algorithm="celllist", **kw)
    pots_a, F_a, fa, _, _ = run(nb, algorithm="allpairs", **kw)
    e_ca = float(np.abs(F_c - F_a).max())
    e_co = float(np.abs(F_c - R.c["thr_F"]).max())
    e_ao = float(np.abs(F_a - R.c["thr_F"]).max())
    ee = max(abs(pots_c[0][t] - R.c["thr_E"][k]) / max(1, abs(R.c["thr_E"][k])) for k, t in enumerate(nb))
    npairs = fc.count_pairs(p, b)
    if not (e_ca < 1e-9 and e_co < 1e-8 and ee < 1e-8 and list(npairs) == [int(x) for x in R.c["thr_n"]]):
        bad.append(("celllist/allpairs/oracle", e_ca, e_co, e_ao, ee, list(npairs),
                    dump(out, "thrombin_cell", it, pos=g["pos"], F_cell=F_c, F_allpairs=F_a, F_oracle=R.c["thr_F"])))
    fc.close()
    fa.close()
    return bad


def case_repulsion(R, out, it):
    import numpy as np
    import torch
    from _golden import box_tensor, pos_tensor
    from torchmd_amd.forces import Forces

    dev = torch.device("cuda:0")
    f = Forces(R.par_w, terms=["repulsion"], algorithm="celllist", cutoff=9.0)
    pd, bd = pos_tensor(R.w_pos, 1, torch.float32, dev), box_tensor(R.w_box, 1, torch.float32, dev)
    F = torch.zeros_like(pd)
    f.compute(pd, bd, F)
    F2 = torch.zeros_like(pd)
    f._evaluate(pd, bd, F2, False, True)
    pots = f.compute(pd, bd, F, returnDetails=True)
    Fo = R.c["rep_F"]
    scale = 1.0 + np.abs(Fo)
    e1 = float((np.abs(F.cpu().numpy() - Fo) / scale).max())
    e2 = float((np.abs(F2.cpu().numpy() - Fo) / scale).max())
    ee = abs(pots[0]["repulsion"] - R.c["rep_E"][0]) / max(1, abs(R.c["rep_E"][0]))
    n = f.count_pairs(pd, bd)
    bad = []
    if not (e1 < 6e-5 and e2 < 6e-5 and ee <= 6e-5 and list(n) == [int(x) for x in R.c["rep_n"]]):
        bad.append((e1, e2, ee, list(n), dump(out, "repulsion", it, pos=R.w_pos, F_gpu=F.cpu().numpy(), F_noenergy=F2.cpu().numpy(), F_oracle=Fo)))
    f.close()
    return bad


def case_vmap(R, out, it):
    import numpy as np
    import torch
    from _golden import box_tensor, energies, pos_tensor
    from torchmd_amd.forces import Forces

    dev = torch.device("cuda:0")
    g = R.g_ala
    terms = ["bonds", "angles", "dihedrals", "impropers", "1-4", "electrostatics", "lj"]
    f = Forces(R.par_ala, terms=terms, cutoff=None, switch_dist=7.5, rfa=False)
    pos = pos_tensor(g["pos"], 1, torch.float64, dev)
    box = box_tensor(np.zeros(3), 1, torch.float64, dev)
    positions = torch.stack([pos] * 4, dim=0)
    positions[2, 0, 7, 1] += 0.05
    positions.requires_grad = True
    epot = torch.vmap(f.compute, in_dims=(0,))(positions, box=box, forces=None, returnDetails=False, explicit_forces=False,
                                               calculateForces=False, toNumpy=False)
    epot.sum().backward()
    forces = (-positions.grad).cpu().numpy()
    ref = sum(energies(g, "f64_full_nocut", 0).values())
    e = epot.detach().cpu().numpy().ravel()
    ok = (abs(e[0] - ref) < 1e-8 and abs(e[1] - ref) < 1e-8 and abs(e[3] - ref) < 1e-8 and abs(e[2] - ref) > 1e-6
          and np.abs(forces[0, 0] - g["f64_full_nocut_forces"][0]).max() < 1e-8
          and np.abs(forces[1, 0] - g["f64_full_nocut_forces"][0]).max() < 1e-8
          and np.abs(forces[3, 0] - g["f64_full_nocut_forces"][0]).max() < 1e-8)
    bad = []
    if not ok:
        bad.append((list(e - ref), dump(out, "vmap", it, pos=g["pos"], epot=e, ref=ref, forces=forces, F_ref=g["f64_full_nocut_forces"])))
    f.close()
    return bad


CASES = {"thrombin": case_thrombin, "repulsion": case_repulsion, "vmap": case_vmap}


def child(args):
    import torch

    print(box_header(), flush=True)
    R = Refs(args.cache)
    res = {c: {"runs": 0, "failures": 0, "errors": []} for c in CASES}
    t0 = time.perf_counter()
    for it in range(args.iters):
        for name, fn in CASES.items():
            res[name]["runs"] += 1
            try:
                bad = fn(R, args.out, it)
            except Exception:  # noqa: BLE001  (a crash of the library call is a failure of the iteration too)
                bad = [traceback.format_exc(limit=3)]
            if bad:
                res[name]["failures"] += 1
                if len(res[name]["errors"]) < 5:
                    res[name]["errors"].append(repr(bad)[:600])
                print(f"FAIL {name} iteration {it}: {repr(bad)[:300]}", flush=True)
    torch.cuda.synchronize()
    res["seconds"] = time.perf_counter() - t0
    res["poison"] = os.environ.get("TMDHIP_DEBUG_POISON", "0")
    res["iters"] = args.iters
    print("SOAKRESULT " + json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inproc", type=int, default=200)
    ap.add_argument("--fresh", type=int, default=20)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "soak"))
    ap.add_argument("--cache", default="/tmp/soak_refs.npz")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--iters", type=int, default=1)
    args = ap.parse_args()
    if args.child:
        return child(args)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    log = open(args.out + ".log", "w")

    def say(s):
        print(s, flush=True)
        log.write(s + "\n")
        log.flush()

    say(f"# tools/soak.py --inproc {args.inproc} --fresh {args.fresh}  ({time.strftime('%Y-%m-%d %H:%M:%S')})")
    sanity = os.path.join(ROOT, "tools", "ubench", "sanity")
    if os.path.exists(sanity):
        r = subprocess.run([sanity, "3"], capture_output=True, text=True)
        say("## library-independent sanity (tools/ubench/sanity.hip): exit code %d" % r.returncode)
        say(r.stdout.strip())
    else:
        say("## tools/ubench/sanity not built (hipcc --offload-arch=gfx950 -O3 tools/ubench/sanity.hip -o tools/ubench/sanity)")
    totals = {}

    def run_child(iters, poison, label):
        env = dict(os.environ)
        env.pop("TMDHIP_DEBUG_POISON", None)
        if poison:
            env["TMDHIP_DEBUG_POISON"] = "1"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--iters", str(iters), "--out", args.out, "--cache", args.cache],
                           capture_output=True, text=True, env=env)
        lines = r.stdout.strip().splitlines()
        hdr = next((l for l in lines if l.startswith("box:")), "box: ?")
        resl = next((l for l in lines if l.startswith("SOAKRESULT ")), None)
        for l in lines:
            if l.startswith("FAIL"):
                say("   " + l)
        if resl is None:
            say(f"{label}: process died (exit {r.returncode}): {r.stderr.strip()[-400:]}")
            totals.setdefault("died", 0)
            totals["died"] += 1
            return
        res = json.loads(resl[len("SOAKRESULT "):])
        say(f"{label}: {hdr} | " + ", ".join(f"{c} {res[c]['failures']}/{res[c]['runs']} failed" for c in CASES) + f" | {res['seconds']:.1f} s")
        for c in CASES:
            t = totals.setdefault(c, [0, 0])
            t[0] += res[c]["failures"]
            t[1] += res[c]["runs"]

    say("## in one process")
    run_child(1, False, "warm-up (references cached)")
    for poison in (False, True):
        run_child(args.inproc, poison, f"in-process x{args.inproc}, poison {int(poison)}")
    say("## fresh processes")
    for k in range(args.fresh):
        run_child(1, k % 2 == 1, f"fresh {k:2d}, poison {k % 2}")
    say("## totals: " + ", ".join(f"{c} {v[0]} failures in {v[1]} runs" for c, v in totals.items() if c != "died")
        + (f", {totals['died']} processes died" if totals.get("died") else ""))
    log.close()


if __name__ == "__main__":
    main()
