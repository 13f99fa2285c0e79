"""`Forces` — host-side mirror of the reference force/energy engine, backed by HIP kernels.

Same constructor, attributes, `compute()` signature, return types and error behaviour as the
reference class (`torchmd/forces.py:7-346`), so `torchmd/run.py`, the minimizers and the
`Integrator` can use it unchanged.  All arithmetic runs in `libtmdhip.so` (see include/tmdhip.h):

* nonbonded block (forces.py:260-319)  -> tmdhip_compute_nonbonded  (tiled all-pairs kernel, or cell
  list + Verlet list + list pair kernel instead of the reference's dense [P,2] pair tensor)
* bonded block (forces.py:122-258)      -> tmdhip_compute_bonded
* the `external` plugin hook (forces.py:321-326) is kept as is (a torch module on the device).

There is no CPU path: tensors must live on a ROCm device, otherwise `compute()` raises.
"""

from __future__ import annotations

import atexit
import ctypes as C
import sys
import weakref

import numpy as np
import torch

from . import _lib as L


def _np_real(t, dtype):
    return np.ascontiguousarray(t.detach().to("cpu", dtype).numpy())


def build_exclusion_csr(natoms, pairs):
    """Symmetric, per-row sorted, duplicate-free CSR of the excluded pairs returned by
    `Parameters.get_exclusions` (reference parameters.py:89-107).  This replaces the N x N boolean
    matrix of `_make_indeces` (forces.py:348-357)."""
    pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
    pairs = pairs[pairs[:, 0] != pairs[:, 1]]
    if len(pairs) == 0:
        return np.zeros(natoms + 1, dtype=np.int32), np.zeros(0, dtype=np.int32)
    if pairs.min() < 0 or pairs.max() >= natoms:
        raise ValueError("exclusion index out of range")
    both = np.concatenate([pairs, pairs[:, ::-1]], axis=0)
    key = np.unique(both[:, 0] * np.int64(natoms) + both[:, 1])
    rows = (key // natoms).astype(np.int64)
    cols = (key % natoms).astype(np.int32)
    offsets = np.zeros(natoms + 1, dtype=np.int64)
    np.add.at(offsets, rows + 1, 1)
    offsets = np.cumsum(offsets).astype(np.int32)
    return offsets, cols


_LIVE_ENGINES = weakref.WeakSet()


@atexit.register
def _close_all_engines():
    # release device contexts while the HIP runtime (and any attached profiler) is still fully alive
    for eng in list(_LIVE_ENGINES):
        eng.close()


def merge_lj_types(A, B, types):
    """Merge atom types that share their LJ parameters (most type names of a protein force field do).
    The pair kernels keep the [T,T] table in LDS, so a smaller T means more resident waves.  Rows of
    (A | B) identical  <=>  same sigma and epsilon  <=>  identical columns.  Returns the compacted tables,
    the remapped per-atom types and the type -> class map (None when nothing was merged)."""
    _, first, inverse = np.unique(np.concatenate([A, B], axis=1), axis=0, return_index=True, return_inverse=True)
    inverse = np.asarray(inverse).reshape(-1)
    if len(first) == A.shape[0]:
        return A, B, types, None
    A2 = np.ascontiguousarray(A[np.ix_(first, first)])
    B2 = np.ascontiguousarray(B[np.ix_(first, first)])
    return A2, B2, inverse[types], inverse


class _Engine:
    """One tmdhip context = (device, dtype, nreplicas) instance of a Forces object."""

    def __init__(self, owner: "Forces", device: torch.device, dtype: torch.dtype, nreplicas: int, exact: bool):
        lib = L.load()
        self.lib = lib
        self.ctx = C.c_void_p()
        self.nreplicas = nreplicas
        self.dtype = dtype
        self.type_map = None  # type id -> LJ class when classes were merged
        par = owner.par
        n = owner.natoms
        keep = []  # numpy arrays that must outlive the create call

        def ptr(a):
            keep.append(a)
            return a.ctypes.data_as(C.c_void_p)

        d = L.NonbondedDesc()
        d.struct_size = C.sizeof(L.NonbondedDesc)
        d.dtype = L.dtype_code(dtype)
        d.natoms = n
        d.nreplicas = nreplicas
        d.device = device.index if device.index is not None else torch.cuda.current_device()
        terms = 0
        for t in owner.energies:
            terms |= L.TERM_BIT.get(t, 0)
        d.terms = terms
        need_tab = terms & (L.TERM_LJ | L.TERM_REPULSION | L.TERM_REPULSIONCG)
        if need_tab:
            A, B = owner._lj_tables()
            A_np, B_np = _np_real(A, dtype), _np_real(B, dtype)
            types = par.mapped_atom_types.detach().cpu().numpy().astype(np.int64)
            A_np, B_np, types, self.type_map = merge_lj_types(A_np, B_np, types)
            self.ntypes = int(A_np.shape[0])
            d.ntypes = self.ntypes
            d.types_host = ptr(np.ascontiguousarray(types.astype(np.int32)))
            d.lj_A_host = ptr(A_np)
            d.lj_B_host = ptr(B_np)
        else:
            self.ntypes = 1
            d.ntypes = 1
            d.types_host = ptr(np.zeros(n, dtype=np.int32))
        d.charges_host = ptr(_np_real(par.charges, dtype))
        off, idx = owner._excl_csr
        d.excl_offsets_host = ptr(off)
        d.excl_index_host = ptr(idx if len(idx) else np.zeros(1, dtype=np.int32))
        d.rfa = 1 if owner.rfa else 0
        d.cutoff = float(owner.cutoff) if owner.cutoff is not None else 0.0
        d.switch_dist = float(owner.switch_dist) if owner.switch_dist is not None else 0.0
        d.solvent_dielectric = float(owner.solventDielectric)
        d.switch_mode = L.SWITCH_EXACT if exact else L.SWITCH_REFERENCE
        d.algorithm = {"auto": L.ALGO_AUTO, "allpairs": L.ALGO_ALLPAIRS, "celllist": L.ALGO_CELLLIST}[owner.algorithm]
        d.skin = float(owner.skin) if owner.skin else 0.0
        L.check(lib.tmdhip_create(C.byref(self.ctx), C.byref(d)), "tmdhip_create")

        b = L.BondedDesc()
        b.struct_size = C.sizeof(L.BondedDesc)
        have = False
        en = owner.energies

        def table(tab, width):
            idx = np.ascontiguousarray(tab["idx"].detach().cpu().numpy().astype(np.int32)).reshape(-1, width)
            mp = tab["map"].detach().cpu().numpy()
            prm = _np_real(tab["params"], dtype)
            return idx, mp, prm

        if "bonds" in en and par.bond_params is not None:
            idx, mp, prm = table(par.bond_params, 2)
            b.nbonds = len(idx)
            b.bond_idx_host = ptr(idx)
            b.bond_prm_host = ptr(np.ascontiguousarray(prm[mp[:, 1]]))
            b.bonds_use_cutoff = 1 if owner.cutoff is not None else 0
            have = True
        if "angles" in en and par.angle_params is not None:
            idx, mp, prm = table(par.angle_params, 3)
            b.nangles = len(idx)
            b.angle_idx_host = ptr(idx)
            b.angle_prm_host = ptr(np.ascontiguousarray(prm[mp[:, 1]]))
            have = True
        if "dihedrals" in en and par.dihedral_params is not None:
            idx, mp, prm = table(par.dihedral_params, 4)
            order = np.argsort(mp[:, 0], kind="stable")
            b.ndihedrals = len(idx)
            b.dihedral_idx_host = ptr(idx)
            b.ndihedral_terms = len(mp)
            b.dihedral_term_of_host = ptr(np.ascontiguousarray(mp[order, 0].astype(np.int32)))
            b.dihedral_prm_host = ptr(np.ascontiguousarray(prm[mp[order, 1]]))
            have = True
        if "impropers" in en and par.improper_params is not None:
            idx, mp, prm = table(par.improper_params, 4)
            order = np.argsort(mp[:, 0], kind="stable")
            b.nimpropers = len(idx)
            b.improper_idx_host = ptr(idx)
            b.nimproper_terms = len(mp)
            b.improper_term_of_host = ptr(np.ascontiguousarray(mp[order, 0].astype(np.int32)))
            b.improper_prm_host = ptr(np.ascontiguousarray(prm[mp[order, 1]]))
            have = True
        p14 = par.nonbonded_14_params
        if "1-4" in en and p14 is not None and torch.is_tensor(p14.get("idx")) and len(p14["idx"]):
            idx, mp, prm = table(p14, 2)
            b.n14 = len(idx)
            b.pair14_idx_host = ptr(idx)
            b.pair14_prm_host = ptr(np.ascontiguousarray(prm[mp[:, 1]]))
            t14 = 0
            if "lj" in en:
                t14 |= L.TERM_LJ
            if "electrostatics" in en:
                t14 |= L.TERM_ELECTROSTATICS
            b.terms14 = t14
            have = have or bool(t14)
        self.has_bonded = have
        if have:
            L.check(lib.tmdhip_set_bonded(self.ctx, C.byref(b)), "tmdhip_set_bonded")
        self.has_nonbonded = terms != 0
        st = L.Stats()
        L.check(lib.tmdhip_get_stats(self.ctx, 0, C.byref(st)), "tmdhip_get_stats")
        # the nonbonded kernels *store* forces when asked to (the cell-list pair kernel owns every atom
        # exactly once; the all-pairs path zero-fills internally), which saves a zero-fill pass here
        self.stores_forces = self.has_nonbonded
        w = owner._skin_weight_array()
        if w is not None and st.algorithm == L.ALGO_CELLLIST:
            L.check(lib.tmdhip_set_skin_weights(self.ctx, ptr(_np_real(torch.as_tensor(w), dtype))), "tmdhip_set_skin_weights")
        # per-term energies [R, NENERGY] followed by the kinetic energies [R] in ONE buffer, so that
        # Integrator.step reads everything back with a single device-to-host copy
        self.comb = torch.zeros(nreplicas * (L.NENERGY + 1), dtype=torch.float64, device=device)
        self.ebuf = self.comb[: nreplicas * L.NENERGY].view(nreplicas, L.NENERGY)
        self.kebuf = self.comb[nreplicas * L.NENERGY:]
        _LIVE_ENGINES.add(self)
        del keep

    def close(self):
        if self.ctx:
            self.lib.tmdhip_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        # never call into the HIP runtime during interpreter teardown (module globals may be gone by then: `sys` is None)
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


class Forces:
    """
    Parameters (as the reference, torchmd/forces.py:27-37)
    ----------
    cutoff : float
        Only calculate LJ, electrostatics and bond energies for atoms closer than this threshold.
    rfa : bool
        Reaction-field approximation for electrostatics up to the cutoff.
    solventDielectric : float
        Dielectric used by `rfa`.
    switch_dist : float
        Start of the LJ switching function.

    Extra keyword-only knobs of this implementation
    ----------
    skin : float          Verlet-list skin in Angstrom (default 1.2; the largest pair skin when per-atom skins are on)
    skin_weights          "mass" (default) | None | array [natoms] in (0, 1]: per-atom share of the skin.  Atom i may
                          move w_i * skin / 2 before the list is rebuilt and pair (i, j) is listed within
                          cutoff + (w_i + w_j) * skin / 2 — exactly as safe as one skin for all, but slow (heavy)
                          atoms stop paying for the room the fast ones need.  "mass": w_i = (m_min / m_i)^0.45,
                          at least 0.2 (measured on flexible TIP3P: the oxygens' largest displacement is 0.28 of
                          the hydrogens'); systems with a single mass keep the uniform skin.
    algorithm : str       "auto" | "allpairs" | "celllist"
    switch_mode : str     "reference" (upstream's explicit switching force, extra 1/r,
                          forces.py:410-412) | "exact" (-dE/dr)
    """

    bonded = ["bonds", "angles", "dihedrals", "impropers", "1-4"]
    nonbonded = ["electrostatics", "lj", "repulsion", "repulsioncg"]
    terms = bonded + nonbonded

    def __init__(
        self,
        parameters,
        terms=None,
        external=None,
        cutoff=None,
        rfa=False,
        solventDielectric=78.5,
        switch_dist=None,
        exclusions=("bonds", "angles", "1-4"),
        *,
        skin=None,
        skin_weights="mass",
        algorithm="auto",
        switch_mode="reference",
    ):
        self.par = parameters
        if terms is None:
            raise RuntimeError(
                'Set force terms or leave empty brackets [].\nAvailable options: "bonds", "angles", "dihedrals", '
                '"impropers", "1-4", "electrostatics", "lj", "repulsion", "repulsioncg".'
            )
        if self.par.nonbonded_params is not None and "lj" in terms:
            self.par.A, self.par.B = self.par.get_AB()

        self.energies = [ene.lower() for ene in terms]
        for et in self.energies:
            if et not in Forces.terms:
                raise ValueError(f"Force term {et} is not implemented.")
        if "1-4" in self.energies and "dihedrals" not in self.energies:
            raise RuntimeError("You cannot enable 1-4 interactions without enabling dihedrals")
        if algorithm not in ("auto", "allpairs", "celllist"):
            raise ValueError("algorithm must be 'auto', 'allpairs' or 'celllist'")
        if switch_mode not in ("reference", "exact"):
            raise ValueError("switch_mode must be 'reference' or 'exact'")
        if rfa and cutoff is None:
            raise RuntimeError("rfa=True needs a cutoff")
        # the library encodes "no cutoff" / "no switching" as 0: reject the values that would be mistaken for
        # "unset" (the reference would filter out every pair with cutoff = 0 and switch from r = 0)
        if cutoff is not None and not cutoff > 0:
            raise ValueError("cutoff must be positive (use None for no cutoff)")
        if switch_dist is not None and not switch_dist > 0:
            raise ValueError("switch_dist must be positive (use None for no switching)")
        if switch_dist is not None and cutoff is not None and not switch_dist < cutoff:
            raise ValueError("switch_dist must be smaller than cutoff")

        self.natoms = len(parameters.masses)
        self.require_distances = any(f in self.nonbonded for f in self.energies)
        self.external = external
        self.cutoff = cutoff
        self.rfa = rfa
        self.solventDielectric = solventDielectric
        self.switch_dist = switch_dist
        self.exclusions = tuple(exclusions)
        self.skin = skin  # None -> library default
        self.skin_weights = skin_weights
        self.algorithm = algorithm
        self.switch_mode = switch_mode
        self._excl_csr = build_exclusion_csr(
            self.natoms, parameters.get_exclusions(exclusions) if self.require_distances else []
        )
        self._engines = {}
        self._box_cache = None
        self._ava_idx = None

    def _skin_weight_array(self):
        """Per-atom skin weights in (0, 1] as a numpy array, or None for the uniform skin."""
        sw = self.skin_weights
        if sw is None:
            return None
        if isinstance(sw, str):
            if sw != "mass":
                raise ValueError("skin_weights must be 'mass', None or an array of per-atom weights")
            m = np.asarray(self.par.masses.detach().cpu().numpy(), dtype=np.float64).ravel()
            if len(m) != self.natoms or not (m > 0).all() or m.min() == m.max():
                return None
            return np.maximum((m.min() / m) ** 0.45, 0.2)
        w = np.asarray(sw, dtype=np.float64).ravel()
        if len(w) != self.natoms or not ((w > 0) & (w <= 1)).all():
            raise ValueError("skin_weights: one weight in (0, 1] per atom")
        return w

    def update_atoms(self, parameters, nactive=None):
        """Atomic systems only (no bonded terms, no exclusions): swap in a new atom set — `parameters`
        with the same LJ table but other atoms (count, types, charges) — keeping the device contexts and
        their buffers; atoms with index >= `nactive` become passive (they act on the others but get no
        neighbour list and zero force).  Used by the domain decomposition at every atom migration."""
        if any(t in self.energies for t in self.bonded) or len(self._excl_csr[1]):
            raise RuntimeError("update_atoms is only available for atomic systems")
        self.par = parameters
        self.natoms = len(parameters.masses)
        self._excl_csr = (np.zeros(self.natoms + 1, dtype=np.int32), np.zeros(0, dtype=np.int32))
        self._ava_idx = None
        types0 = parameters.mapped_atom_types.detach().cpu().numpy().astype(np.int64)
        for eng in self._engines.values():
            types = np.ascontiguousarray((eng.type_map[types0] if eng.type_map is not None else types0).astype(np.int32))
            if types.max(initial=0) >= eng.ntypes:
                raise RuntimeError("update_atoms: atom type outside the context's LJ table")
            q = _np_real(parameters.charges, eng.dtype)
            L.check(
                eng.lib.tmdhip_update_atoms(eng.ctx, self.natoms, types.ctypes.data_as(C.c_void_p),
                                            q.ctypes.data_as(C.c_void_p), int(nactive) if nactive else 0),
                "tmdhip_update_atoms",
            )
            eng.ebuf.zero_()
        self._nactive = nactive

    def _atoms_swapped(self, natoms, nactive):
        """Host-side bookkeeping of an atom swap the library has made itself (tmdhip_dd_migrate wrote the contexts'
        per-atom arrays on the device): what `update_atoms` does around its C call."""
        self.natoms = int(natoms)
        self._excl_csr = (np.zeros(self.natoms + 1, dtype=np.int32), np.zeros(0, dtype=np.int32))
        self._ava_idx = None
        self._nactive = nactive
        for eng in self._engines.values():
            eng.ebuf.zero_()

    def close(self):
        """Release the device contexts (they are re-created on the next compute())."""
        for eng in self._engines.values():
            eng.close()
        self._engines = {}

    # ------------------------------------------------------------------ reference attributes
    @property
    def ava_idx(self):
        """Dense [P,2] list of non-excluded i<j pairs (reference `_make_indeces`, forces.py:348-357).
        Only materialised on request — the kernels never use it — and only for small systems."""
        if not self.require_distances:
            return None
        if self._ava_idx is None:
            if self.natoms > 20000:
                raise MemoryError("ava_idx is O(N^2); not available for natoms > 20000")
            i, j = np.triu_indices(self.natoms, k=1)
            off, idx = self._excl_csr
            rows = np.repeat(np.arange(self.natoms), np.diff(off))
            excl_keys = rows.astype(np.int64) * self.natoms + idx
            keep = ~np.isin(i.astype(np.int64) * self.natoms + j, excl_keys)
            self._ava_idx = torch.tensor(np.stack([i[keep], j[keep]], axis=1)).to(self.par.device)
        return self._ava_idx

    def _lj_tables(self):
        A, B = getattr(self.par, "A", None), getattr(self.par, "B", None)
        if A is None or B is None:
            if self.par.nonbonded_params is None:
                raise RuntimeError("LJ/repulsion terms requested but the parameters hold no nonbonded table")
            A, B = self.par.get_AB()
        return A, B

    # ------------------------------------------------------------------ engine plumbing
    def _engine(self, pos, exact=None):
        """`exact` selects the switching-force flavour: the explicit path keeps upstream's formula,
        the autograd path (explicit_forces=False) is -dE/dr by construction (forces.py:328-336)."""
        if exact is None:
            exact = self.switch_mode == "exact"
        exact = bool(exact) and self.switch_dist is not None
        key = (pos.device.index, pos.dtype, pos.shape[0], exact)
        eng = self._engines.get(key)
        if eng is None:
            with torch.cuda.device(pos.device):
                eng = _Engine(self, pos.device, pos.dtype, pos.shape[0], exact)
            self._engines[key] = eng
        return eng

    def _host_box(self, box):
        """Box diagonals on the host, [R,3] float64.  Re-read only when the tensor changed (a read is a
        device sync; the reference syncs on `torch.all(box == 0)` every call, forces.py:361)."""
        # keyed on the tensor OBJECT (weak reference) and its version counter: a temporary freed and
        # re-allocated at the same address is a different object, so it can never hit a stale entry.
        # (In-place edits through `.data` do not bump the version: pass a new tensor or use set_box.)
        cache = self._box_cache
        if cache is not None and cache[0]() is box and cache[1] == box._version:
            return cache[2]
        diag = torch.diagonal(box.detach(), dim1=-2, dim2=-1).to("cpu", torch.float64).contiguous().numpy()
        host = np.ascontiguousarray(diag.reshape(-1, 3))
        self._box_cache = (weakref.ref(box), box._version, host)
        return host

    def _launch(self, eng, pos, box, forces, want_energy, want_forces, count_pairs=False):
        """Enqueue bonded + nonbonded kernels of every replica on the current stream."""
        lib = eng.lib
        hbox = self._host_box(box)
        stream = C.c_void_p(torch.cuda.current_stream(pos.device).cuda_stream)
        flags = (L.WANT_ENERGY if want_energy else 0) | (L.WANT_FORCES if want_forces else 0)
        R, N = pos.shape[0], pos.shape[1]
        if want_energy:
            eng.ebuf.zero_()
        if want_forces and not eng.stores_forces:
            forces.zero_()
        # one call for all replicas (the reference's `for i in range(nsystems)` loop, forces.py:116, runs
        # inside the library: batched launches for all-pairs systems, per-replica lists otherwise)
        boxes = np.ascontiguousarray(
            np.stack([hbox[min(r, len(hbox) - 1)] for r in range(R)]).astype(np.float64)
        )
        bx = boxes.ctypes.data_as(C.POINTER(C.c_double))
        p = C.c_void_p(pos.data_ptr())
        f = C.c_void_p(forces.data_ptr()) if want_forces else C.c_void_p()
        e = C.c_void_p(eng.ebuf.data_ptr())
        # nonbonded first: on the cell-list path it overwrites `forces`, the bonded kernels then add
        if eng.has_nonbonded:
            nbflags = flags | (L.COUNT_PAIRS if count_pairs else 0)
            if eng.stores_forces:
                nbflags |= L.OVERWRITE_FORCES
            L.check(
                lib.tmdhip_compute_nonbonded(eng.ctx, L.ALL_REPLICAS, p, bx, f, e, nbflags, stream),
                "tmdhip_compute_nonbonded",
            )
        if eng.has_bonded:
            L.check(
                lib.tmdhip_compute_bonded(eng.ctx, L.ALL_REPLICAS, p, bx, f, e, flags, stream),
                "tmdhip_compute_bonded",
            )

    def _verify(self, eng, pos):
        """Host-visible validity check (neighbour-list capacity). True = results valid."""
        stream = C.c_void_p(torch.cuda.current_stream(pos.device).cuda_stream)
        ok = True
        for r in range(pos.shape[0]):
            rc = L.check(eng.lib.tmdhip_check(eng.ctx, r, stream), "tmdhip_check")
            ok = ok and rc == 0
        return ok

    def _evaluate(self, pos, box, forces, want_energy, want_forces, count_pairs=False, exact=None):
        """Zero + fill `forces`, return the per-term energy buffer [R, NENERGY] (float64, device)."""
        L.require_device_tensor(pos, "pos")
        L.require_device_tensor(box, "box")
        if pos.dim() != 3 or pos.shape[2] != 3 or pos.shape[1] != self.natoms:
            raise RuntimeError(f"pos must have shape (nreplicas, {self.natoms}, 3), got {tuple(pos.shape)}")
        p = pos.detach()
        if not p.is_contiguous():
            p = p.contiguous()
        target = None
        if want_forces:
            L.require_device_tensor(forces, "forces")
            if forces.dtype != pos.dtype or forces.shape != pos.shape:
                raise RuntimeError("forces must have the dtype and shape of pos")
            if not forces.is_contiguous():
                target, forces = forces, torch.empty_like(p)
        eng = self._engine(p, exact)
        with torch.cuda.device(p.device):
            for _ in range(4):
                self._launch(eng, p, box, forces, want_energy, want_forces, count_pairs)
                if not (want_energy or count_pairs) or self._verify(eng, p):
                    break
            else:
                raise RuntimeError("neighbour list kept overflowing; increase `skin` capacity")
        if target is not None:
            target.copy_(forces)
        return eng.ebuf

    def _evaluate_sync(self, pos, box, forces, exact=None):
        """One energy (+ force) evaluation through `tmdhip_compute`: a single C call and a single host
        synchronisation.  Fills `forces` (None: energies only) and returns the per-term energies as a host
        array [R, NENERGY] (float64)."""
        L.require_device_tensor(pos, "pos")
        L.require_device_tensor(box, "box")
        if pos.dim() != 3 or pos.shape[2] != 3 or pos.shape[1] != self.natoms:
            raise RuntimeError(f"pos must have shape (nreplicas, {self.natoms}, 3), got {tuple(pos.shape)}")
        p = pos.detach()
        if not p.is_contiguous():
            p = p.contiguous()
        target = None
        if forces is not None:
            L.require_device_tensor(forces, "forces")
            if forces.dtype != pos.dtype or forces.shape != pos.shape:
                raise RuntimeError("forces must have the dtype and shape of pos")
            if not forces.is_contiguous():
                target, forces = forces, torch.empty_like(p)
        eng = self._engine(p, exact)
        hbox = self._host_box(box)
        R = p.shape[0]
        boxes = hbox if len(hbox) == R else np.ascontiguousarray(np.stack([hbox[min(r, len(hbox) - 1)] for r in range(R)]))
        out = np.empty((R, L.NENERGY), dtype=np.float64)
        with torch.cuda.device(p.device):
            stream = C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream)
            for _ in range(4):
                rc = L.check(
                    eng.lib.tmdhip_compute(
                        eng.ctx, C.c_void_p(p.data_ptr()), boxes.ctypes.data_as(C.POINTER(C.c_double)),
                        C.c_void_p(forces.data_ptr()) if forces is not None else C.c_void_p(),
                        out.ctypes.data_as(C.POINTER(C.c_double)), stream,
                    ),
                    "tmdhip_compute",
                )
                if rc == 0:
                    break
            else:
                raise RuntimeError("neighbour list kept overflowing; increase `skin` capacity")
        if target is not None:
            target.copy_(forces)
        return out

    # ------------------------------------------------------------------ public API
    def compute(
        self,
        pos,
        box,
        forces,
        returnDetails=False,
        explicit_forces=True,
        toNumpy=True,
        calculateForces=True,
    ):
        if _is_batched(pos):  # called under torch.vmap (reference tests/test_torchmd.py:590-598)
            return self._compute_vmapped(pos, box, forces, returnDetails, explicit_forces, toNumpy, calculateForces)
        if calculateForces:
            if not explicit_forces and not pos.requires_grad:
                raise RuntimeError(
                    "The positions passed don't require gradients. Please use pos.detach().requires_grad_(True) "
                    "before passing."
                )
        nsystems = pos.shape[0]
        want_forces = calculateForces and forces is not None
        if forces is not None and not want_forces:
            forces.zero_()  # the reference zeroes `forces` whenever it is given (forces.py:113-114)
        # The reference's potential is a torch expression of `pos`, so a tensor result can always be
        # back-propagated, also with calculateForces=False (its vmap test does exactly that).  Here the
        # gradient is -F from the kernels: evaluate the forces whenever a differentiable result is due.
        differentiable = (not toNumpy) and (not returnDetails) and pos.requires_grad and torch.is_grad_enabled()
        scratch = None
        if forces is None and ((calculateForces and not explicit_forces and pos.requires_grad) or differentiable):
            scratch = torch.zeros_like(pos.detach())
        exact = True if ((calculateForces and not explicit_forces) or (differentiable and not want_forces)) else None
        ehost = self._evaluate_sync(pos, box, forces if want_forces else scratch, exact=exact)

        ext_ene = None
        if self.external:
            ext_ene, ext_force = self.external.calculate(pos, box)
            if want_forces and explicit_forces:
                forces += ext_force
            elif want_forces and not explicit_forces:
                # Deviation from upstream, on purpose: with explicit_forces=False the reference overwrites
                # `forces` with -grad of the summed potential (forces.py:328-336), so an external force only
                # survives there if `ext_ene` carries an autograd graph back to `pos`.  Here the external
                # force the plugin returns is always added, exactly as on the explicit path.
                forces += ext_force.detach() if ext_force is not None else 0

        # per-term energies in the order of the reference dict: self.energies ..., then "external"
        names = list(self.energies) + ["external"]
        cols = np.zeros((nsystems, len(names)), dtype=np.float64)
        for k, name in enumerate(self.energies):
            slot = L.ENERGY_SLOT.get(name)
            if slot is not None:  # "1-4" has no slot: it accumulates into lj/electrostatics (forces.py:216,232)
                cols[:, k] = ehost[:, slot]
        if ext_ene is not None:
            cols[:, -1] = torch.as_tensor(ext_ene).detach().to("cpu", torch.float64).reshape(nsystems).numpy()

        if not returnDetails:
            tot = cols.sum(axis=1)
            if toNumpy:
                return [float(v) for v in tot]
            tot = torch.as_tensor(tot, device=pos.device).to(pos.dtype)
            # whenever a force buffer was filled the potential is differentiable w.r.t. `pos` (in the reference it
            # always is a torch expression of pos): backward() hands out -F, decoupled from the caller's tensor
            if differentiable and (want_forces or scratch is not None):
                fsrc = forces if want_forces else scratch
                tot = _PotentialWithGrad.apply(pos, tot, fsrc.detach().clone())
            return tot
        if toNumpy:
            return [{n: float(v) for n, v in zip(names, row)} for row in cols]
        tcols = torch.as_tensor(cols, device=pos.device).to(pos.dtype)
        return [{n: tcols[s, k : k + 1].clone() for k, n in enumerate(names)} for s in range(nsystems)]

    def _compute_vmapped(self, pos, box, forces, returnDetails, explicit_forces, toNumpy, calculateForces):
        """`torch.vmap(forces.compute)`: the batch dimension becomes extra replicas of ONE evaluation (the
        kernels serve [R,N,3] anyway), and the result is handed back to vmap as a batched tensor.  Supports what
        the reference's use needs: a tensor result (toNumpy=False, returnDetails=False), `forces=None`."""
        F = torch._C._functorch
        if toNumpy or returnDetails:
            raise RuntimeError("compute() under torch.vmap returns tensors: pass toNumpy=False, returnDetails=False")
        if forces is not None:
            raise RuntimeError("compute() under torch.vmap does not fill a `forces` tensor: pass forces=None and "
                               "differentiate the returned potential")
        level, bdim = F.maybe_get_level(pos), F.maybe_get_bdim(pos)
        p = F.get_unwrapped(pos)
        if _is_batched(p):
            raise RuntimeError("nested torch.vmap over compute() is not supported")
        p = p.movedim(bdim, 0)  # [B, R, N, 3]
        B, R = p.shape[0], p.shape[1]
        flat = p.reshape(B * R, p.shape[2], 3)
        if _is_batched(box):
            b = F.get_unwrapped(box).movedim(F.maybe_get_bdim(box), 0)
        else:
            b = box.unsqueeze(0).expand(B, *box.shape)
        bflat = b.reshape(B * R, 3, 3).contiguous()
        tot = self.compute(flat, bflat, None, False, explicit_forces, False, calculateForces)  # [B * R]
        return F._add_batch_dim(tot.reshape(B, R), 0, level)

    # used by Integrator: no host synchronisation unless energies are requested
    def _compute_async(self, pos, box, forces, want_energy):
        ebuf = self._evaluate(pos, box, forces, want_energy, True)
        if self.external:
            ext_ene, ext_force = self.external.calculate(pos, box)
            forces += ext_force
            if want_energy:
                return ebuf, torch.as_tensor(ext_ene, device=pos.device).detach().to(torch.float64).reshape(-1)
        return (ebuf, None) if want_energy else (None, None)

    def _md_run(self, system, masses, vcoeff, dt, gamma, seed, step0, niter, restore=False):
        """Integrator fast path: `niter` MD steps enqueued by one C call; returns the energy buffer of the
        last step (device, [R, NENERGY]).  `restore`: first rewind to the state at the entry of the previous
        call (tmdhip_md_restore) — the replay after a neighbour-list validity failure."""
        pos = system.pos
        L.require_device_tensor(pos, "systems.pos")
        if pos.shape[1] != self.natoms:
            raise RuntimeError(f"systems.pos must have {self.natoms} atoms")
        eng = self._engine(pos)
        hbox = self._host_box(system.box)
        R = pos.shape[0]
        # the descriptor and the per-replica box array are kept between calls (a 20-step call is short enough for
        # the Python in front of its first launch to show)
        cache = getattr(eng, "_md_cache", None)
        if cache is None or cache[0] is not hbox or cache[1] != R:
            boxes = np.ascontiguousarray(np.stack([hbox[min(r, len(hbox) - 1)] for r in range(R)]).astype(np.float64))
            d = L.MdDesc()
            d.struct_size = C.sizeof(L.MdDesc)
            d.box_host = boxes.ctypes.data_as(C.c_void_p)
            cache = eng._md_cache = (hbox, R, boxes, d)
        d = cache[3]
        d.niter = int(niter)
        d.pos_dev, d.vel_dev, d.forces_dev = pos.data_ptr(), system.vel.data_ptr(), system.forces.data_ptr()
        d.mass_dev = masses.data_ptr()
        d.vcoeff_dev = vcoeff.data_ptr() if vcoeff is not None else None
        d.dt, d.gamma = float(dt), float(gamma)
        d.seed, d.step0 = int(seed), int(step0)
        d.energies_dev = eng.ebuf.data_ptr()  # zeroed by the library
        # has anybody written the positions through torch since the previous call returned?  (tensor version counters;
        # the library's own kernels write through raw pointers and do not count.)  Only a hint, see tmdhip_md_desc.
        key = (pos.data_ptr(), pos._version, id(hbox))
        d.continuation = 1 if (not restore and getattr(eng, "_md_key", None) == key) else 0
        stream = C.c_void_p(torch.cuda.current_stream(pos.device).cuda_stream)
        if restore:
            L.check(eng.lib.tmdhip_md_restore(eng.ctx, C.byref(d), stream), "tmdhip_md_restore")
        L.check(eng.lib.tmdhip_md_run(eng.ctx, C.byref(d), stream), "tmdhip_md_run")
        eng._md_key = (pos.data_ptr(), pos._version, id(hbox))
        return eng.ebuf

    def energy_columns(self):
        return [L.ENERGY_SLOT[n] for n in self.energies if n in L.ENERGY_SLOT]

    def total_energy_from(self, ebuf, ext):
        cols = [L.ENERGY_SLOT[n] for n in self.energies if n in L.ENERGY_SLOT]
        tot = ebuf[:, cols].sum(dim=1) if cols else torch.zeros(ebuf.shape[0], dtype=torch.float64, device=ebuf.device)
        if ext is not None:
            tot = tot + ext
        return tot

    def count_pairs(self, pos, box):
        """Number of non-excluded i<j pairs with r <= cutoff per replica (the P_cut of the
        pair-interactions/s metric)."""
        self._evaluate(pos, box, None, False, False, count_pairs=True)
        return [self.stats(pos, r)["pairs_in_cutoff"] for r in range(pos.shape[0])]

    def invalidate_lists(self, pos):
        """Drop the neighbour lists of every replica: the next evaluation re-plans the grid and rebuilds them
        (`tmdhip_invalidate_list`; after positions were changed out of band, and by tests)."""
        eng = self._engine(pos.detach())
        for r in range(pos.shape[0]):
            L.check(eng.lib.tmdhip_invalidate_list(eng.ctx, r), "tmdhip_invalidate_list")

    def stats(self, pos, replica=0):
        eng = self._engine(pos.detach())
        st = L.Stats()
        L.check(eng.lib.tmdhip_get_stats(eng.ctx, replica, C.byref(st)), "tmdhip_get_stats")
        return {
            "n_compute": st.n_compute,
            "n_rebuilds": st.n_rebuilds,
            "list_entries": st.list_entries,
            "pairs_in_cutoff": st.pairs_in_cutoff,
            "algorithm": {L.ALGO_ALLPAIRS: "allpairs", L.ALGO_CELLLIST: "celllist"}.get(st.algorithm, "?"),
            "max_neighbours": st.max_neighbours,
            "overflow": st.overflow,
            "ncell": tuple(st.ncell),
            "skin": st.skin,
            "final_steps_in_pair_launch": st.final_steps_in_pair_launch,
            "batched_launches": st.batched_launches,
            "chains_skipped": int(st.chains_skipped),
            "steps_in_pair_launch": int(st.steps_in_pair_launch),
            "fused_step_timeouts": int(st.fused_step_timeouts),
        }

    def enable_timing(self, pos, on=True, every=1, limit=0, skip=0, interior_only=False):
        """HIP events around every `every`-th launch of the list pair kernel (an event pair costs 3-6 us of stream
        time), at most `limit` of them (0: no limit), after passing over the first `skip` (<= 7) launches;
        `interior_only`: the launches that also return energies (another variant of the kernel: the last step of a
        `step()` call, `compute()`) are neither timed nor counted."""
        eng = self._engine(pos.detach())
        code = (min(max(1, int(every)), 0xFFFF) | (min(max(0, int(limit)), 0xFFF) << 16) | (min(max(0, int(skip)), 7) << 28)) if on else 0
        if on and interior_only:
            code |= 1 << 31
        L.check(eng.lib.tmdhip_timing_enable(eng.ctx, C.c_int(code - (1 << 32) if code >= (1 << 31) else code)))

    def read_timing(self, pos, reset=True):
        """(total ms, launches) of the list pair kernel measured with HIP events on the launch stream."""
        eng = self._engine(pos.detach())
        ms, n = C.c_double(), C.c_int64()
        L.check(eng.lib.tmdhip_timing_read(eng.ctx, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value


def _is_batched(t) -> bool:
    return torch.is_tensor(t) and torch._C._functorch.is_batchedtensor(t)


class _PotentialWithGrad(torch.autograd.Function):
    """Makes the returned potential differentiable w.r.t. `pos` for `explicit_forces=False` callers
    (reference forces.py:328-336 derives forces with autograd; here dE/dpos = -F from the kernels).
    Written in the functorch-compatible style (setup_context) so that it can be applied while a
    `torch.vmap` level is active."""

    generate_vmap_rule = True

    @staticmethod
    def forward(pos, energy, forces):
        return energy.clone()

    @staticmethod
    def setup_context(ctx, inputs, output):
        ctx.save_for_backward(inputs[2])

    @staticmethod
    def backward(ctx, grad_out):
        (forces,) = ctx.saved_tensors
        return -forces * grad_out.reshape(-1, 1, 1).to(forces.dtype), None, None
