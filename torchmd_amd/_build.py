"""Build the gfx950 shared library `torchmd_amd/lib/libtmdhip.so` with hipcc (in-tree, so the
binary travels with the repo snapshot to the GPU box).

    python -m torchmd_amd._build [--force]
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIBPATH = os.path.join(LIBDIR, "libtmdhip.so")
SOURCES = ["nonbonded.hip", "bonded.hip", "integrator.hip", "domain.hip"]
HEADERS = ["common.h", "pair_math.h", "rng.h", "bonded_math.h", os.path.join("..", "..", "include", "tmdhip.h")]
ARCH = "gfx950"
FLAGS = ["-fno-slp-vectorize"]  # the SLP vectoriser packs the pair kernel into v_pk_* ops + v_mov transposes: measured slower


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def is_stale() -> bool:
    if not os.path.exists(LIBPATH):
        return True
    built = os.path.getmtime(LIBPATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > built for d in deps if os.path.exists(d))


def build_library(force: bool = False, verbose: bool = False, extra_flags=(), out: str | None = None) -> str:
    """Compile the library.  `extra_flags` / `out`: developer knobs for A/B builds (see TMDHIP_LIB in _lib.py)."""
    target = out or LIBPATH
    if not force and not out and not is_stale():
        return LIBPATH
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [
        _hipcc(),
        "-O3",
        "-std=c++17",
        f"--offload-arch={ARCH}",
        "-fPIC",
        "-shared",
        "-Wall",
        "-Wno-unused-function",
        *FLAGS,
        *extra_flags,
        *[os.path.join(CSRC, s) for s in SOURCES],
        "-o",
        target + ".tmp",
    ]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{res.stdout}\n{res.stderr}")
    os.replace(target + ".tmp", target)
    return target


if __name__ == "__main__":
    extra = [a for a in sys.argv[1:] if a.startswith(("-f", "-m", "-D"))]
    out = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--out=")), None)
    print(build_library(force="--force" in sys.argv, verbose=True, extra_flags=extra, out=out))
