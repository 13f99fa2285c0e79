"""Build the gfx950 shared library `torchmd_amd/lib/libtmdhip.so` with hipcc (in-tree, so the
binary travels with the repo snapshot to the GPU box).

    python -m torchmd_amd._build [--force]

Every translation unit of `csrc/` is compiled to an object of its own (in parallel; only the stale ones) and
the objects are linked into the library.
"""

from __future__ import annotations

import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIBPATH = os.path.join(LIBDIR, "libtmdhip.so")
SOURCES = ["context.hip", "list_build.hip", "pair_generic.hip", "pair_fast_f32.hip", "pair_fast_f32_batch.hip", "pair_lean_f64.hip", "md_loop.hip",
           "bonded.hip", "integrator.hip", "domain.hip", "dd_migrate.hip"]
HEADERS = ["common.h", "pair_math.h", "rng.h", "bonded_math.h", "engine.h", "md_step.h", "pair_fast_kernel.h", "dd_comm.h",
           os.path.join("..", "..", "include", "tmdhip.h")]
ARCH = "gfx950"
FLAGS = ["-fno-slp-vectorize"]  # the SLP vectoriser packs the pair kernel into v_pk_* ops + v_mov transposes: measured slower


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _newest_header() -> float:
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))


def is_stale() -> bool:
    if not os.path.exists(LIBPATH):
        return True
    built = os.path.getmtime(LIBPATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > built for d in deps if os.path.exists(d))


def _compile_one(hipcc, src, obj, extra_flags, verbose):
    cmd = [hipcc, "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-c", "-Wall", "-Wno-unused-function",
           *FLAGS, *extra_flags, src, "-o", obj + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed on {os.path.basename(src)}:\n{res.stdout}\n{res.stderr}")
    os.replace(obj + ".tmp", obj)
    return res.stderr


def build_library(force: bool = False, verbose: bool = False, extra_flags=(), out: str | None = None) -> str:
    """Compile the library.  `extra_flags` / `out`: developer knobs for A/B builds (see TMDHIP_LIB in _lib.py)."""
    target = out or LIBPATH
    if not force and not out and not is_stale():
        return LIBPATH
    hipcc = _hipcc()
    # objects of an A/B build (other flags) live in a directory of their own
    tag = hashlib.sha1(" ".join(extra_flags).encode()).hexdigest()[:8] if extra_flags else "default"
    objdir = os.path.join(OBJDIR, tag)
    os.makedirs(objdir, exist_ok=True)
    hdr = _newest_header()
    jobs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr):
            jobs.append((src, obj))
    workers = max(1, min(len(jobs), os.cpu_count() or 1))
    if jobs:
        with concurrent.futures.ThreadPoolExecutor(workers) as pool:
            futs = [pool.submit(_compile_one, hipcc, src, obj, list(extra_flags), verbose) for src, obj in jobs]
            for f in futs:
                warn = f.result()
                if verbose and warn.strip():
                    print(warn, file=sys.stderr)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-fPIC", "-shared", *objs, "-o", target + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"link failed:\n{res.stdout}\n{res.stderr}")
    os.replace(target + ".tmp", target)
    return target


if __name__ == "__main__":
    extra = [a for a in sys.argv[1:] if a.startswith(("-f", "-m", "-D"))]
    out = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--out=")), None)
    print(build_library(force="--force" in sys.argv, verbose=True, extra_flags=extra, out=out))
