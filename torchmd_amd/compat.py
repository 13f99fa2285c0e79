"""Make `from torchmd.forces import Forces` (etc.) resolve to the MI355X classes.

`torchmd/run.py:5-12` and user scripts import the hot-path classes by module path.  `install()` registers
this package's modules under the reference's names in `sys.modules`, so such code runs unchanged:

    import torchmd_amd.compat; torchmd_amd.compat.install()
    from torchmd.forces import Forces              # -> torchmd_amd.forces.Forces
    from torchmd.integrator import Integrator      # -> torchmd_amd.integrator.Integrator

If the reference package is importable, only the hot-path modules are replaced and everything else
(`torchmd.parameters`, `torchmd.forcefields`, ...) stays the reference's; otherwise a `torchmd` namespace
made of this package's mirrors is created.
"""

from __future__ import annotations

import importlib
import sys
import types

HOT_PATH = ("forces", "integrator", "systems", "wrapper")
MIRRORS = ("parameters", "forcefields", "run", "minimizers", "utils")


def install(everything: bool = False):
    """Returns the list of `torchmd.*` module names that now point at this package."""
    try:
        ref = importlib.import_module("torchmd")
        have_ref = not getattr(ref, "__torchmd_amd_shim__", False)
    except ImportError:
        ref, have_ref = None, False
    if ref is None:
        ref = types.ModuleType("torchmd")
        ref.__path__ = []  # mark as a package
        ref.__torchmd_amd_shim__ = True
        sys.modules["torchmd"] = ref
    names = list(HOT_PATH) + (list(MIRRORS) if (everything or not have_ref) else [])
    done = []
    for name in names:
        mod = importlib.import_module(f"torchmd_amd.{name}")
        sys.modules[f"torchmd.{name}"] = mod
        setattr(ref, name, mod)
        done.append(f"torchmd.{name}")
    return done
