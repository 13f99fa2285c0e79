"""State container for the MD hot path.

Same public surface and tensor layout as the reference `System` (`torchmd/systems.py:5-98`):
`pos, vel, forces [R,N,3]` contiguous, `box [R,3,3]` (only the diagonal is used by the force
path, reference `torchmd/forces.py:118`), `masses [N,1]`; dtype = `precision`, device = `device`.
The integrator and force kernels update these buffers in place, so callers (run.py, minimizers)
keep seeing the tensors they hold references to.
"""

from __future__ import annotations

import numpy as np
import torch


class System:
    _FIELDS = ("box", "pos", "vel", "forces", "masses")

    def __init__(self, natoms, nreplicas, precision, device):
        self.box = torch.zeros(nreplicas, 3, 3, dtype=precision, device=device)
        self.pos = torch.zeros(nreplicas, natoms, 3, dtype=precision, device=device)
        self.vel = torch.zeros(nreplicas, natoms, 3, dtype=precision, device=device)
        self.forces = torch.zeros(nreplicas, natoms, 3, dtype=precision, device=device)
        self.masses = torch.zeros(natoms, 1, dtype=precision, device=device)

    @property
    def natoms(self):
        return self.pos.shape[1]

    @property
    def nreplicas(self):
        return self.pos.shape[0]

    def to_(self, device):
        for f in self._FIELDS:
            setattr(self, f, getattr(self, f).to(device))

    def precision_(self, precision):
        for f in self._FIELDS:
            setattr(self, f, getattr(self, f).type(precision))

    # ---- setters (argument conventions of reference systems.py:42-98) -----------------
    def set_positions(self, pos):
        """`pos`: [N,3] or [N,3,F] (numpy or tensor); frame f goes to replica f, a single frame is
        broadcast to all replicas."""
        if pos.shape[1] != 3:
            raise RuntimeError(
                "Positions shape must be (natoms, 3, 1) or (natoms, 3, nreplicas) "
                f"but were given {pos.shape} instead"
            )
        pos = torch.as_tensor(np.asarray(pos) if isinstance(pos, np.ndarray) else pos.detach())
        pos = pos.to(dtype=self.pos.dtype, device=self.pos.device)
        if pos.ndim == 2:
            pos = pos[:, :, None]
        frames = pos.permute(2, 0, 1)
        if self.nreplicas > 1 and frames.shape[0] != self.nreplicas:
            frames = frames[:1].expand(self.nreplicas, -1, -1)
        self.pos[:] = frames

    def set_velocities(self, vel):
        if tuple(vel.shape) != (self.nreplicas, self.natoms, 3):
            raise RuntimeError("Velocities shape must be (nreplicas, natoms, 3)")
        self.vel[:] = torch.as_tensor(vel).detach().to(dtype=self.vel.dtype, device=self.vel.device)

    def set_box(self, box):
        """`box`: [3] or [3,F] box edge lengths; written to the diagonal of `self.box[r]`."""
        box = np.asarray(box.detach().cpu() if torch.is_tensor(box) else box)
        if box.ndim == 1:
            if len(box) != 3:
                raise RuntimeError("Box must have at least 3 elements")
            box = box[:, None]
        if box.shape[0] != 3:
            raise RuntimeError("Box shape must be (3, 1) or (3, nreplicas)")
        box = box.T  # [F,3]
        if self.nreplicas > 1 and box.shape[0] != self.nreplicas:
            box = np.repeat(box[:1], self.nreplicas, axis=0)
        diag = torch.as_tensor(np.ascontiguousarray(box), dtype=self.box.dtype, device=self.box.device)
        for r in range(diag.shape[0]):
            self.box[r].diagonal().copy_(diag[r])

    def set_forces(self, forces):
        if tuple(forces.shape) != (self.nreplicas, self.natoms, 3):
            raise RuntimeError("Forces shape must be (nreplicas, natoms, 3)")
        self.forces[:] = torch.as_tensor(forces).to(dtype=self.forces.dtype, device=self.forces.device)

    def set_masses(self, masses):
        if tuple(masses.shape) != (self.natoms,):
            raise RuntimeError("Masses shape must be (natoms,)")
        self.masses[:, 0] = torch.as_tensor(masses).detach().to(
            dtype=self.masses.dtype, device=self.masses.device
        )
