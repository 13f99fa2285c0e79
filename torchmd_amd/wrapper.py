"""Periodic wrapping of whole molecules — mirror of the reference `Wrapper` (`torchmd/wrapper.py`).

`Wrapper(natoms, bonds, device)`; `.wrap(pos, box, wrapidx=None)` moves every bonded group (connected
component of the bond graph) by whole box vectors so that its centre (unweighted mean of its atoms)
lies in [0, box); atoms without bonds are wrapped individually.  The reference does this with a Python
loop over the groups (`wrapper.py:22-25`); here it is one HIP kernel (`tmdhip_wrap`).
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L


def calculate_molecule_groups(natoms, bonds):
    """Connected components of the bond graph as a CSR (offsets, members), members ascending."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components

    if bonds is not None and len(bonds):
        b = np.asarray(bonds, dtype=np.int64).reshape(-1, 2)
        adj = coo_matrix((np.ones(len(b), dtype=np.int8), (b[:, 0], b[:, 1])), shape=(natoms, natoms))
        ngroups, label = connected_components(adj, directed=False)
    else:
        ngroups, label = natoms, np.arange(natoms)
    order = np.argsort(label, kind="stable").astype(np.int32)
    counts = np.bincount(label, minlength=ngroups)
    offsets = np.zeros(ngroups + 1, dtype=np.int32)
    offsets[1:] = np.cumsum(counts)
    return offsets, order


class Wrapper:
    def __init__(self, natoms, bonds, device):
        self.natoms = natoms
        off, mem = calculate_molecule_groups(natoms, bonds)
        self.ngroups = len(off) - 1
        self.has_big = bool(np.any(np.diff(off) > 64))
        self._off = torch.as_tensor(off, device=device)
        self._mem = torch.as_tensor(mem, device=device)
        # reference attributes: list of index tensors for groups > 1 atom, tensor of single atoms
        sizes = np.diff(off)
        self.nongrouped = torch.as_tensor(mem[off[:-1][sizes == 1]].astype(np.int64), device=device)
        self._sizes = sizes

    @property
    def groups(self):
        off, mem = self._off.cpu().numpy(), self._mem.cpu().numpy()
        return [torch.as_tensor(mem[off[g]:off[g + 1]].astype(np.int64), device=self._mem.device)
                for g in range(self.ngroups) if self._sizes[g] > 1]

    def wrap(self, pos, box, wrapidx=None):
        if wrapidx is not None:
            # The reference re-binds `pos` to a centred *copy* before wrapping (wrapper.py:16-19), so the
            # caller's tensor is left untouched in that mode; reproduced as is.
            return
        L.require_device_tensor(pos, "pos")
        L.require_device_tensor(box, "box")
        if not pos.is_contiguous() or not box.is_contiguous():
            raise RuntimeError("pos and box must be contiguous")
        if self._off.device != pos.device:
            self._off, self._mem = self._off.to(pos.device), self._mem.to(pos.device)
        lib = L.load()
        bx = box if box.dtype == pos.dtype else box.to(pos.dtype)
        with torch.cuda.device(pos.device):
            L.check(
                lib.tmdhip_wrap(
                    L.dtype_code(pos.dtype), pos.shape[0], pos.shape[1], pos.data_ptr(), bx.data_ptr(), self.ngroups,
                    self._off.data_ptr(), self._mem.data_ptr(), 1 if self.has_big else 0,
                    C.c_void_p(torch.cuda.current_stream(pos.device).cuda_stream),
                ),
                "tmdhip_wrap",
            )
