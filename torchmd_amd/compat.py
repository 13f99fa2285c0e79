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


def install(everything: bool = False, extra=()):
    """Returns the list of `torchmd.*` module names that now point at this package.  `extra`: further mirrors to
    register although the reference is importable — e.g. ("forcefields",) where the reference's AMBER backend needs
    parmed (`torchmd/forcefields/ff_parmed.py:23`) and this package reads the prmtop natively."""
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
    names = list(HOT_PATH) + (list(MIRRORS) if (everything or not have_ref) else [m for m in extra if m in MIRRORS])
    done = []
    for name in names:
        mod = importlib.import_module(f"torchmd_amd.{name}")
        sys.modules[f"torchmd.{name}"] = mod
        setattr(ref, name, mod)
        done.append(f"torchmd.{name}")
        if hasattr(mod, "__path__"):  # a package (forcefields): its submodules under the reference's names too
            import pkgutil

            for sub in pkgutil.iter_modules(mod.__path__):
                sm = importlib.import_module(f"torchmd_amd.{name}.{sub.name}")
                sys.modules[f"torchmd.{name}.{sub.name}"] = sm
                done.append(f"torchmd.{name}.{sub.name}")
    return done
