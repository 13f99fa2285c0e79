"""torchmd_amd — MI355X-native (gfx950) nonbonded force/energy + integrator hot path for TorchMD.

Public surface mirrors the reference package for this path:
    torchmd_amd.forces.Forces, torchmd_amd.integrator.Integrator, torchmd_amd.systems.System,
    torchmd_amd.parameters.Parameters, torchmd_amd.forcefields.ForceField
backed by hand-written HIP kernels in `torchmd_amd/lib/libtmdhip.so` (C ABI: include/tmdhip.h).
"""

from .forces import Forces
from .integrator import Integrator, kinetic_energy, kinetic_to_temp, maxwell_boltzmann
from .parameters import Parameters
from .systems import System

__all__ = [
    "Forces",
    "Integrator",
    "Parameters",
    "System",
    "kinetic_energy",
    "kinetic_to_temp",
    "maxwell_boltzmann",
]
__version__ = "0.1.0"
