#!/usr/bin/env python
"""Child process of tests/test_gpu_reference_driver.py: the REFERENCE'S OWN driver — `torchmd/run.py:30-291`
(`get_args`, `setup`, `dynamics`), imported unchanged from a reference checkout — run on the MI355X classes.

    python tests/reference_driver_main.py --ref /path/to/reference --case water|ala2 --log-dir DIR

What is replaced, and nothing else: `moleculekit` (not in this image: tests/stubs/moleculekit, file readers only) and
the hot-path modules `torchmd.forces / integrator / systems / wrapper` (`torchmd_amd.compat.install()`); case `ala2`
also `torchmd.forcefields` (the reference's AMBER backend needs parmed).  `Parameters`, `utils`, `minimizers` and
`run.py` itself are the reference's.  Prints one JSON line (the last line of stdout)."""
import argparse
import csv
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def monitor_rows(log_dir, k):
    with open(os.path.join(log_dir, f"monitor_{k}.csv")) as fh:
        return [{a: float(b) for a, b in row.items()} for row in csv.DictReader(fh)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", required=True)
    ap.add_argument("--case", default="water", choices=["water", "ala2"])
    ap.add_argument("--log-dir", required=True)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--mirror", action="store_true", help="this package's own driver (torchmd_amd.run) instead, same options")
    a = ap.parse_args()
    ref = os.path.abspath(a.ref)
    sys.path[:0] = [os.path.join(HERE, "stubs"), ref, ROOT]
    os.chdir(ref)  # the reference's configuration files hold paths relative to its root
    import torch

    out = {"case": a.case, "driver": "torchmd_amd.run" if a.mirror else "reference torchmd/run.py"}
    if a.mirror:
        from torchmd_amd import run as drv
    else:
        from torchmd_amd import compat

        out["replaced"] = compat.install(extra=("forcefields",) if a.case == "ala2" else ())
        import torchmd.run as drv

        assert os.path.abspath(drv.__file__).startswith(ref), drv.__file__
        import torchmd.forces, torchmd.integrator, torchmd.parameters, torchmd.utils  # noqa: E401

        assert torchmd.forces.__name__ == "torchmd_amd.forces" and torchmd.integrator.__name__ == "torchmd_amd.integrator"
        assert os.path.abspath(torchmd.parameters.__file__).startswith(ref)  # (the reference's own Parameters)
        out["driver_file"] = os.path.relpath(drv.__file__, ref)
    if a.case == "water":
        argv = ["--conf", "tests/water/water_conf.yaml", "--device", "cuda:0", "--log-dir", a.log_dir, "--steps", str(a.steps),
                "--output-period", "50", "--save-period", "50"]
    else:
        argv = ["--conf", "tests/prod_alanine_dipeptide_amber/conf.yaml", "--device", "cuda:0", "--log-dir", a.log_dir,
                "--steps", str(a.steps), "--output-period", "50", "--save-period", "100", "--minimize", "40",
                # (the file leaves `forceterms:` empty, which the reference's own Forces rejects, forces.py:39-42)
                "--forceterms", "bonds", "angles", "dihedrals", "impropers", "1-4", "electrostatics", "lj"]
    args = drv.get_args(argv)
    mol, system, forces = drv.setup(args)
    assert type(forces).__module__ == "torchmd_amd.forces" and type(system).__module__ == "torchmd_amd.systems"
    e0 = forces.compute(system.pos, system.box, system.forces.clone(), returnDetails=True)
    out["epot_step0_terms"] = e0
    out["epot_step0"] = [float(sum(d.values())) for d in e0]
    out["stats"] = {k: v for k, v in forces.stats(system.pos).items() if k in ("algorithm", "n_compute")}
    drv.dynamics(args, mol, system, forces)
    torch.cuda.synchronize()
    out["monitor"] = [monitor_rows(a.log_dir, k) for k in range(args.replicas)]
    out["trajectory_shape"] = list(__import__("numpy").load(os.path.join(a.log_dir, f"{args.output}_0.npy")).shape)
    out["native_library"] = [ln.split()[-1] for ln in open("/proc/self/maps") if "libtmdhip" in ln][:1]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
