"""Spatial domain decomposition with halo exchange (10^6-atom LJ box on
8 GPUs).  The reference has no counterpart (single process, single device); this is new work behind
the same kernels.

Scheme ("full-list ownership": no force return path)
  * the periodic box is cut into px x py x pz bricks, one rank (= one GPU, one process) per brick;
    2 x 2 x 2 bricks have exactly 7 distinct neighbour ranks = the 7 xGMI links of an MI355X;
  * a rank OWNS the atoms whose wrapped position lies in its brick and integrates only those;
  * every step it receives the positions of the HALO atoms — all atoms (periodic images included)
    within `cutoff + skin` of its brick — from the owning ranks, already shifted to the image that
    lies next to the brick, and evaluates the forces on its own atoms from own + halo atoms with the
    ordinary open-boundary cell-list engine.  Each pair that straddles a face is computed on both
    sides, so forces never travel; only positions do (12 B per halo atom per step);
  * when any atom has moved more than skin/2 since the last migration, atoms are re-assigned to bricks
    (all-to-all of state), the halo plan is rebuilt and the engine re-created for the new local set.

Per step: ONE all-to-all of positions (26 directed messages per rank, packed into one
`all_to_all_single` that writes straight into the halo rows of the engine's position buffer) + every
`check_every` steps one 4-byte all-reduce for the migration trigger, read back asynchronously and acted on
at the next check (no host synchronisation in the step loop).  Device work per step besides the force
evaluation: `tmdhip_dd_step` (kick of the previous step + drift of this one + displacement maximum, one
launch) and `tmdhip_halo_pack` (one launch) — `csrc/domain.hip`.  Over RCCL the payload is ~0.56 MB per
rank and step at C5: latency-, not bandwidth-bound.

Scope of this version: atomic systems (no bonded terms; types/charges travel with the atoms) — LJ,
repulsion and electrostatic terms.  Molecules with bonds need molecule-aware ownership (next).

`LocalTransport` runs all ranks inside one process (used to validate the decomposition against the
single-domain engine on one GPU); `DistTransport` is the torch.distributed one (nccl = RCCL on GPUs,
gloo in the CPU tests of the exchange layer).
"""

from __future__ import annotations

import itertools
from types import SimpleNamespace

import numpy as np
import torch

DIRECTIONS = [d for d in itertools.product((-1, 0, 1), repeat=3) if d != (0, 0, 0)]  # 26 neighbours


def factor_grid(world: int):
    """px*py*pz = world, as cubic as possible (8 -> 2x2x2, 4 -> 2x2x1, 2 -> 2x1x1)."""
    best = None
    for px in range(1, world + 1):
        if world % px:
            continue
        for py in range(1, world // px + 1):
            if (world // px) % py:
                continue
            pz = world // px // py
            key = (max(px, py, pz) - min(px, py, pz), -px, -py)
            if best is None or key < best[0]:
                best = (key, (px, py, pz))
    return best[1]


class BrickGrid:
    def __init__(self, box, world, grid=None):
        self.box = torch.as_tensor(box, dtype=torch.float64).reshape(3)
        if not bool((self.box > 0).all()):
            raise ValueError("domain decomposition needs a periodic box")
        self.dims = tuple(grid) if grid is not None else factor_grid(world)
        if int(np.prod(self.dims)) != world:
            raise ValueError(f"grid {self.dims} does not match world size {world}")
        self.world = world
        self.edge = self.box / torch.tensor(self.dims, dtype=torch.float64)

    def coords(self, rank):
        px, py, pz = self.dims
        return (rank // (py * pz), (rank // pz) % py, rank % pz)

    def rank_of(self, c):
        px, py, pz = self.dims
        return ((c[0] % px) * py + (c[1] % py)) * pz + (c[2] % pz)

    def bounds(self, rank):
        c = torch.tensor(self.coords(rank), dtype=torch.float64)
        lo = c * self.edge
        return lo, lo + self.edge

    def owner(self, pos):
        """Owning rank of every position (any periodic image), and the wrapped positions."""
        box = self.box.to(pos.device, pos.dtype)
        w = pos - torch.floor(pos / box) * box
        w = torch.where(w >= box, w - box, w)
        dims = torch.tensor(self.dims, device=pos.device)
        c = torch.minimum((w / self.edge.to(pos.device, pos.dtype)).floor().long(), dims - 1).clamp_(min=0)
        px, py, pz = self.dims
        return (c[:, 0] * py + c[:, 1]) * pz + c[:, 2], w


class HaloPlan:
    """For one rank: which owned atoms go to which neighbour (26 directed messages) and with which
    periodic shift, valid until the next migration."""

    def __init__(self, grid: BrickGrid, rank: int, wrapped_pos: torch.Tensor, halo: float):
        if bool((grid.edge < halo).any()):
            raise ValueError(f"bricks {grid.edge.tolist()} are thinner than the halo {halo}: use fewer ranks")
        dev, dt = wrapped_pos.device, wrapped_pos.dtype
        lo, hi = (t.to(dev, dt) for t in grid.bounds(rank))
        me = grid.coords(rank)
        near_lo = wrapped_pos < lo + halo  # [n,3]
        near_hi = wrapped_pos >= hi - halo
        # the 26 directed messages, ordered by destination rank so that one all-to-all carries them
        order = sorted(range(len(DIRECTIONS)), key=lambda q: (grid.rank_of(tuple(me[k] + DIRECTIONS[q][k] for k in range(3))), q))
        self.dest = [grid.rank_of(tuple(me[k] + DIRECTIONS[q][k] for k in range(3))) for q in order]
        shifts = torch.zeros(len(order), 3, dtype=torch.float64)
        for m_, q in enumerate(order):
            for k in range(3):
                if DIRECTIONS[q][k] == -1 and me[k] == 0:
                    shifts[m_, k] = grid.box[k]  # crossing the lower global face: receiver sees us at +L
                elif DIRECTIONS[q][k] == 1 and me[k] == grid.dims[k] - 1:
                    shifts[m_, k] = -grid.box[k]
        # membership of every atom in every message as ONE [26, n] mask and ONE nonzero (a single host
        # synchronisation instead of 26): row m of the result lists, in atom order, the atoms of message m
        true = torch.ones(wrapped_pos.shape[0], dtype=torch.bool, device=dev)
        axis = [(near_lo[:, k], true, near_hi[:, k]) for k in range(3)]  # indexed by direction component + 1
        masks = torch.stack([axis[0][DIRECTIONS[q][0] + 1] & axis[1][DIRECTIONS[q][1] + 1] & axis[2][DIRECTIONS[q][2] + 1]
                             for q in order])
        nz = torch.nonzero(masks)
        self.send_index = nz[:, 1].contiguous()
        per_message = torch.bincount(nz[:, 0], minlength=len(order)).tolist()
        self.shift = [shifts[m_].to(dev, dt) for m_ in range(len(order))]
        self.send_shift = shifts.to(dev, dt)[nz[:, 0]].contiguous()
        self.index = list(torch.split(self.send_index, per_message))
        counts = [0] * grid.world
        for dst, cnt in zip(self.dest, per_message):
            counts[dst] += cnt
        self.send_counts = counts

    def pack(self, tensor):
        """Rows of `tensor` ([n_own, k]) in message order."""
        return tensor.index_select(0, self.send_index)

    def pack_positions(self, pos):
        return pos.index_select(0, self.send_index) + self.send_shift


# ------------------------------------------------------------------------------------------------
# transports
# ------------------------------------------------------------------------------------------------
class DistTransport:
    """torch.distributed: `nccl` (= RCCL over xGMI) with device tensors, `gloo` with CPU tensors."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def exchange_counts(self, send_counts):
        t = torch.tensor(send_counts, dtype=torch.long, device=self._dev)
        out = torch.empty_like(t)
        self.dist.all_to_all_single(out, t, group=self.group)
        return out.tolist()

    def bind(self, device):
        self._dev = device
        return self

    def native(self):
        """The library's own RCCL communicator over the same ranks (device tensors only): lets the halo exchange
        and the whole step loop be enqueued from C (`tmdhip_dd_run`).  Rank 0 draws the id, this group's
        broadcast distributes it.  None when it cannot be created (CPU tensors / gloo, TMDHIP_DD_NATIVE=0)."""
        if getattr(self, "_native", False) is not False:
            return self._native
        self._native = None
        import ctypes as C
        import os

        from . import _lib as L

        if self._dev.type != "cuda" or os.environ.get("TMDHIP_DD_NATIVE", "1") == "0":
            return None
        lib = L.load()
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        path = (cand if os.path.exists(cand) else "").encode()
        # rank 0 draws the id; a failure there (no librccl to open) must not leave the others waiting in the broadcast:
        # the id travels with a leading success byte and every rank falls back to the torch.distributed transport
        ident = (C.c_ubyte * L.COMM_ID_BYTES)()
        ok = 1
        if self.rank == 0:
            try:
                L.check(lib.tmdhip_comm_unique_id(path, ident), "tmdhip_comm_unique_id")
            except RuntimeError:
                ok = 0
        t = torch.tensor([ok] + list(ident), dtype=torch.uint8, device=self._dev)
        self.dist.broadcast(t, src=self.dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                            group=self.group)
        got = t.cpu().tolist()
        if not got[0]:
            return None
        ident = (C.c_ubyte * L.COMM_ID_BYTES)(*got[1:])
        handle = C.c_void_p()
        made = 1
        with torch.cuda.device(self._dev):
            try:
                L.check(lib.tmdhip_comm_create(C.byref(handle), path, ident, self.rank, self.world), "tmdhip_comm_create")
            except RuntimeError:
                made = 0
        # (a communicator only some ranks could create is useless: agree on the minimum)
        flag = torch.tensor([made], dtype=torch.int32, device=self._dev)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.group)
        if not int(flag.item()):
            if made:
                lib.tmdhip_comm_destroy(handle)
            return None
        self._native = handle
        return handle

    def close(self):
        if getattr(self, "_native", None):
            from . import _lib as L

            L.load().tmdhip_comm_destroy(self._native)
        self._native = False

    def all_to_all(self, send, send_counts, recv_counts):
        out = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        self.dist.all_to_all_single(out, send.contiguous(), output_split_sizes=list(recv_counts),
                                    input_split_sizes=list(send_counts), group=self.group)
        return out

    def all_to_all_into(self, out, send, send_counts, recv_counts):
        """Rows of `send` (message order) -> rows of `out` (source-rank order), no intermediate tensor."""
        self.dist.all_to_all_single(out, send, output_split_sizes=list(recv_counts),
                                    input_split_sizes=list(send_counts), group=self.group)

    def max_(self, t: torch.Tensor) -> torch.Tensor:
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return t

    def any_true(self, flag: torch.Tensor) -> bool:
        t = flag.to(torch.int32).reshape(1).clone()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return bool(t.item())

    def sum(self, t: torch.Tensor) -> torch.Tensor:
        t = t.clone()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t


class LocalTransport:
    """All ranks in one process (round-robin driven by DomainSet): the same calls, served from a shared
    mailbox.  Used to validate the decomposition on a single GPU.

    `native_threads=True`: between migrations every brick's step loop is enqueued from C (`tmdhip_dd_run`) by a host
    thread of its own, on a stream of its own, and the bricks exchange their halo rows through the library's
    in-process communicator (`tmdhip_comm_create_local`: device copies ordered by events, a host barrier instead of
    RCCL) — the code path of a multi-GPU run at world 2 / 4 / 8 on a one-GPU box."""

    def __init__(self, world, native_threads=False):
        self.world = world
        self.box = {}
        self.native_threads = native_threads
        self._hub = None
        self._comms = self._streams = None

    def native(self, device):
        """(communicators, streams), one per rank, created on first use."""
        import ctypes as C

        from . import _lib as L

        if self._comms is None:
            lib = L.load()
            hub = C.c_void_p()
            with torch.cuda.device(device):
                L.check(lib.tmdhip_local_hub_create(C.byref(hub), self.world), "tmdhip_local_hub_create")
                comms = []
                for r in range(self.world):
                    h = C.c_void_p()
                    L.check(lib.tmdhip_comm_create_local(C.byref(h), hub, r), "tmdhip_comm_create_local")
                    comms.append(h)
                self._streams = [torch.cuda.Stream(device) for _ in range(self.world)]
            self._hub, self._comms = hub, comms
        return self._comms, self._streams

    def close(self):
        if self._comms is not None:
            from . import _lib as L

            lib = L.load()
            for h in self._comms:
                lib.tmdhip_comm_destroy(h)
            lib.tmdhip_local_hub_destroy(self._hub)
        self._hub = self._comms = self._streams = None

    def post(self, key, rank, value):
        self.box.setdefault(key, {})[rank] = value

    def collect(self, key):
        vals = self.box.pop(key)
        return [vals[r] for r in range(self.world)]


# ------------------------------------------------------------------------------------------------
# one rank's domain
# ------------------------------------------------------------------------------------------------
class DomainReplay(RuntimeError):
    """The steps since the last saved state are invalid; `DomainSet.step` goes back to that state and repeats them."""


class HaloOverrun(DomainReplay):
    """An atom moved further than half the halo skin before a migration was requested (halo atoms were missing)."""


class ListInvalid(DomainReplay):
    """A brick's neighbour list overflowed or outlived its skin during the batch (its capacity has been grown)."""


def _local_parameters(charges, types, masses, A, B):
    """Minimal `Parameters`-shaped object for a set of atoms without bonded terms."""
    par = SimpleNamespace()
    par.charges, par.masses, par.mapped_atom_types = charges, masses.reshape(-1, 1), types
    par.nonbonded_params = {"params": None} if A is not None else None
    par.bond_params = par.angle_params = par.dihedral_params = par.improper_params = None
    par.nonbonded_14_params = None
    par.A, par.B = A, B
    par.device = "cpu"
    par.get_AB = lambda: (A, B)
    par.get_exclusions = lambda types=(), fullarray=False: []
    return par


class Domain:
    """State and engine of one brick.  `ids` are global atom indices (for gathering / tests)."""

    def __init__(self, grid, rank, device, dtype, terms, cutoff, skin, A, B, engine_kwargs):
        self.grid, self.rank, self.device, self.dtype = grid, rank, device, dtype
        self.terms, self.cutoff, self.skin = terms, float(cutoff), float(skin)
        self.halo = self.cutoff + self.skin
        self.A, self.B = A, B
        self.engine_kwargs = engine_kwargs
        self.forces_engine = None

    # -- state ------------------------------------------------------------------------------
    def adopt(self, ids, pos, vel, charges, types, masses):
        """Take ownership of a set of atoms.  Own positions are kept in the WRAPPED frame (the frame the
        halo images live in) and are integrated in place inside the engine's local position buffer, so
        a step moves no own-atom data around; `unwrap` restores the caller's periodic image on gather."""
        self._bufs = None  # (fresh tensors: the capacity buffers of a native migration are history)
        self.ids, self.vel = ids, vel.contiguous()
        self.charges, self.types, self.masses = charges.contiguous(), types.long().contiguous(), masses.contiguous()
        self.nown = len(ids)
        _, w = self.grid.owner(pos)
        self.unwrap = (pos - w).contiguous()
        self._own_init = w.contiguous()
        self.pos = self._own_init  # re-pointed into local_pos once the halo size is known
        self.ref = w.clone()
        self.plan = HaloPlan(self.grid, self.rank, w, self.halo)
        self.send_index32 = self.plan.send_index.to(torch.int32).contiguous()
        self.send_shift = self.plan.send_shift.contiguous()
        self.send_buf = torch.empty(len(self.send_index32), 3, dtype=self.dtype, device=self.device)
        self.disp2 = torch.zeros(1, dtype=torch.float32, device=self.device)  # max |x - ref|^2 since this migration
        self.vcoeff_unit = torch.sqrt(1.0 / self.masses).contiguous()  # x sqrt(2 gamma kB T dt) at run time
        self._vc_key = None  # (the scaled copy belongs to the previous atom set)

    def state_rows(self):
        """[n_own, 10] float64 rows for migration: id, pos(3), vel(3), charge, type, mass."""
        return torch.cat([self.ids.double()[:, None], (self.pos + self.unwrap).double(), self.vel.double(),
                          self.charges.double()[:, None], self.types.double()[:, None],
                          self.masses.double()[:, None]], dim=1)

    def from_rows(self, rows):
        dt = self.dtype
        self.adopt(rows[:, 0].long(), rows[:, 1:4].to(dt), rows[:, 4:7].to(dt), rows[:, 7].to(dt),
                   rows[:, 8].long(), rows[:, 9].to(dt))

    def max_displacement(self):
        d2 = ((self.pos - self.ref) ** 2).sum(dim=1)
        return torch.sqrt(d2.max()) if len(d2) else torch.zeros((), dtype=self.dtype, device=self.device)

    # -- halo -------------------------------------------------------------------------------
    def pack_halo(self):
        """Positions of all outgoing messages (wrapped frame + image shift) -> `send_buf`, one launch."""
        from . import _lib as L

        if len(self.send_index32):
            L.check(L.load().tmdhip_halo_pack(L.dtype_code(self.dtype), len(self.send_index32), self.pos.data_ptr(),
                                              self.send_index32.data_ptr(), self.send_shift.data_ptr(),
                                              self.send_buf.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream))
        return self.send_buf

    @property
    def halo_rows(self):
        """The halo part of the engine's position buffer (contiguous [nhalo, 3] view): the exchange writes here."""
        return self.local_pos[0, self.nown:]

    def halo_payload(self, static: bool):
        """Rows to send: positions (wrapped frame + image shift), plus charge/type on (re)builds."""
        if not static:
            return self.pack_halo()
        p = self.plan.pack_positions(self.pos)
        extra = torch.stack([self.charges, self.types.to(self.dtype)], dim=1)
        return torch.cat([p, self.plan.pack(extra)], dim=1)

    def set_halo(self, rows, static: bool):
        if static:
            self.halo_charges = rows[:, 3].contiguous()
            self.halo_types = rows[:, 4].long()
            self._build_engine(rows.shape[0])
            self.local_pos = torch.empty(1, self.nown + rows.shape[0], 3, dtype=self.dtype, device=self.device)
            self.local_pos[0, : self.nown] = self._own_init
            self.pos = self.local_pos[0, : self.nown]  # the integrator now updates the engine's buffer in place
        self.local_pos[0, self.nown:] = rows[:, :3]

    def _build_engine(self, nhalo):
        from .forces import Forces

        q = torch.cat([self.charges, self.halo_charges]).cpu()
        t = torch.cat([self.types, self.halo_types]).cpu()
        m = torch.cat([self.masses, torch.ones(nhalo, dtype=self.dtype, device=self.device)]).cpu()
        par = _local_parameters(q, t, m, self.A, self.B)
        n = self.nown + nhalo
        if self.forces_engine is None:
            kw = dict(skin_weights=None)  # halo atoms carry a dummy mass: one skin for all
            kw.update(self.engine_kwargs)
            self.forces_engine = Forces(par, terms=self.terms, cutoff=self.cutoff, **kw)
            # create the context now, then mark the halo atoms passive
            self.forces_engine._engine(torch.empty(1, n, 3, dtype=self.dtype, device=self.device))
        # the device context and its buffers survive migrations: only the atom set is swapped; halo atoms
        # (index >= nown) are passive: they get no neighbour list, so neither forces nor energy are
        # computed on them (own-halo pairs then count half in the energy, as they should)
        self.forces_engine.update_atoms(par, nactive=self.nown)
        self.local_forces = torch.zeros(1, n, 3, dtype=self.dtype, device=self.device)
        self.zero_box = torch.zeros(1, 3, 3, dtype=self.dtype, device=self.device)

    # -- migration in the library (tmdhip_dd_migrate) ---------------------------------------------
    _OWN_ARRAYS = ("ids", "unwrap", "vel", "charge", "type", "mass", "ref")

    def _alloc_buffers(self, cap_own, cap_rows, cap_send):
        dev, dt = self.device, self.dtype
        z = lambda *shape, dtype=dt: torch.zeros(*shape, dtype=dtype, device=dev)  # noqa: E731
        return dict(cap_own=cap_own, cap_rows=cap_rows, cap_send=cap_send, ids=z(cap_own, dtype=torch.int64), pos=z(cap_rows, 3),
                    unwrap=z(cap_own, 3), vel=z(cap_own, 3), charge=z(cap_own), type=z(cap_own, dtype=torch.int32), mass=z(cap_own),
                    ref=z(cap_own, 3), send_index=z(cap_send, dtype=torch.int32), send_shift=z(cap_send, 3), send_buf=z(cap_send, 3),
                    forces=z(cap_rows, 3))

    def _to_buffers(self):
        """Move the brick's state into capacity buffers (first native migration, or after a Python one)."""
        n = self.nown
        nrows = self.local_pos.shape[1] if getattr(self, "local_pos", None) is not None else n
        nsend = len(self.send_index32)
        import os

        if os.environ.get("TMDHIP_DD_TEST_TIGHT_CAPS"):  # test knob: no headroom, so that the "capacity too small" paths run
            b = self._alloc_buffers(n + 1, nrows + 1, nsend + 1)
        else:
            b = self._alloc_buffers(int(1.2 * n) + 4096, int(1.2 * nrows) + 8192, int(1.25 * nsend) + 4096)
        b["ids"][:n], b["pos"][:n], b["unwrap"][:n], b["vel"][:n] = self.ids, self.pos, self.unwrap, self.vel
        b["charge"][:n], b["type"][:n], b["mass"][:n], b["ref"][:n] = self.charges, self.types.to(torch.int32), self.masses, self.ref
        self._bufs = b

    def _grow_buffers(self, need_own, need_rows, need_send):
        """A capacity reported too small by tmdhip_dd_migrate: new buffers, the owned rows copied over."""
        o = self._bufs
        n = min(self.nown, o["cap_own"])
        import os

        tight = bool(os.environ.get("TMDHIP_DD_TEST_TIGHT_CAPS"))
        grow = (lambda need: need + 1) if tight else (lambda need: int(1.25 * need) + 1024)  # noqa: E731
        b = self._alloc_buffers(max(o["cap_own"], grow(need_own)), max(o["cap_rows"], grow(need_rows)), max(o["cap_send"], grow(need_send)))
        self.capacity_growths = getattr(self, "capacity_growths", 0) + 1
        for k in self._OWN_ARRAYS + ("pos",):
            b[k][:n] = o[k][:n]
        self._bufs = b

    def _point_at_buffers(self, nown, nhalo, nsend, send_counts, recv_counts):
        """The brick's fields as views of the capacity buffers after a native migration."""
        b = self._bufs
        n = nown + nhalo
        self.nown = nown
        self.ids, self.unwrap, self.vel = b["ids"][:nown], b["unwrap"][:nown], b["vel"][:nown]
        self.charges, self.types, self.masses, self.ref = b["charge"][:nown], b["type"][:nown], b["mass"][:nown], b["ref"][:nown]
        self.local_pos = b["pos"][:n].view(1, n, 3)
        self.local_forces = b["forces"][:n].view(1, n, 3)
        self.pos = self.local_pos[0, :nown]
        self._own_init = self.pos
        self.send_index32, self.send_shift, self.send_buf = b["send_index"][:nsend], b["send_shift"][:nsend], b["send_buf"][:nsend]
        self.plan = SimpleNamespace(send_counts=list(send_counts))
        self.recv_counts = list(recv_counts)
        self.vcoeff_unit = torch.sqrt(1.0 / self.masses).contiguous()
        self._vc_key = None  # (the scaled copy belongs to the previous atom set)
        if getattr(self, "zero_box", None) is None:  # (the same tensor every time: Forces caches its host copy by object)
            self.zero_box = torch.zeros(1, 3, 3, dtype=self.dtype, device=self.device)
        self.forces_engine._atoms_swapped(n, nown)

    def migrate_native(self, comm, stream):
        """This rank's part of a migration, in the library (every rank of the communicator calls it).  `stream`: the
        raw HIP stream everything is enqueued on."""
        import ctypes as C

        from . import _lib as L

        if getattr(self, "_bufs", None) is None:
            self._to_buffers()
        eng = self.forces_engine._engine(self.local_pos)
        world = self.grid.world
        sc, rc = (C.c_int64 * world)(), (C.c_int64 * world)()
        tmap = eng.type_map
        tm = np.ascontiguousarray(tmap, dtype=np.int32) if tmap is not None else None
        nown_in = self.nown
        for _ in range(8):
            b = self._bufs
            br = L.DdBrick(
                struct_size=C.sizeof(L.DdBrick), dtype=L.dtype_code(self.dtype), rank=self.rank, world=world,
                dims=(C.c_int32 * 3)(*self.grid.dims), ntypes_map=len(tm) if tm is not None else 0,
                box=(C.c_double * 3)(*[float(x) for x in self.grid.box]), halo=self.halo,
                cap_own=b["cap_own"], cap_rows=b["cap_rows"], cap_send=b["cap_send"], nown=nown_in,
                ids_dev=b["ids"].data_ptr(), pos_dev=b["pos"].data_ptr(), unwrap_dev=b["unwrap"].data_ptr(),
                vel_dev=b["vel"].data_ptr(), charge_dev=b["charge"].data_ptr(), type_dev=b["type"].data_ptr(),
                mass_dev=b["mass"].data_ptr(), ref_dev=b["ref"].data_ptr(), disp2_dev=self.disp2.data_ptr(),
                send_index_dev=b["send_index"].data_ptr(), send_shift_dev=b["send_shift"].data_ptr(),
                send_counts_host=C.addressof(sc), recv_counts_host=C.addressof(rc),
                type_map_host=tm.ctypes.data if tm is not None else None,
            )
            with torch.cuda.device(self.device):
                rcode = L.check(L.load().tmdhip_dd_migrate(eng.ctx, comm, C.byref(br), stream), "tmdhip_dd_migrate")
            if rcode == 0:
                self._point_at_buffers(int(br.nown), int(br.nhalo), int(br.nsend), list(sc), list(rc))
                return
            # a capacity was too small (nothing is lost: the call resumes where it stopped)
            if br.need_send > 0:
                self.nown = nown_in = int(br.nown)  # the owned rows are in place already and must be kept
            else:
                self.nown = 0  # the owned rows are still in the library's scratch
            self._grow_buffers(int(br.need_own), int(br.need_rows), int(br.need_send))
        raise RuntimeError("tmdhip_dd_migrate: capacities did not settle")

    def compute(self, want_energy=False):
        """Forces on the owned atoms from own + halo atoms (open boundaries: images are explicit).
        Energy: own-own pairs count fully, own-halo pairs half (the other half is the neighbour's)."""
        e = self.forces_engine._evaluate(self.local_pos, self.zero_box, self.local_forces, want_energy, True)
        self.forces = self.local_forces[0, : self.nown]
        return e


class _DryEngine:
    """Stand-in for `Forces` in a dry run (no device, no forces): what DomainSet asks of an engine."""

    def _engine(self, pos):
        return None

    def _verify(self, eng, pos):
        return True

    def close(self):
        pass


class DryDomain(Domain):
    """A brick without a force engine, on CPU tensors: `bench.py --config c5 --gpus N --dry --backend gloo` drives the
    planning (ownership, halo plans), the count / row exchanges, the migration trigger and the migrations of a
    `DomainSet` over gloo with it — the N > 1 bench path executed end to end where no GPU is.  Forces are zero (atoms
    move ballistically, which is what makes them cross brick faces); the halo rows of every exchange can be checked
    against the brute-force set of periodic images (`halo_matches_brute_force`)."""

    def _build_engine(self, nhalo):
        n = self.nown + nhalo
        self.forces_engine = _DryEngine()
        self.local_forces = torch.zeros(1, n, 3, dtype=self.dtype, device=self.device)
        self.zero_box = torch.zeros(1, 3, 3, dtype=self.dtype, device=self.device)

    def pack_halo(self):
        if len(self.send_index32):
            torch.add(self.pos[self.send_index32.long()], self.send_shift, out=self.send_buf)
        return self.send_buf

    def compute(self, want_energy=False):
        self.forces = self.local_forces[0, : self.nown]
        return None

    def halo_matches_brute_force(self, all_wrapped):
        """`all_wrapped`: the wrapped positions of EVERY atom of the box (gathered by the caller): the rows this brick
        holds behind its own atoms must be exactly the periodic images that lie within `halo` of the brick."""
        import itertools

        lo, hi = self.grid.bounds(self.rank)
        box = torch.as_tensor(self.grid.box, dtype=all_wrapped.dtype)
        exp = []
        for sft in itertools.product((-1, 0, 1), repeat=3):
            img = all_wrapped + torch.tensor(sft, dtype=all_wrapped.dtype) * box
            ext = ((img >= lo - self.halo) & (img < hi + self.halo)).all(dim=1)
            own = ((img >= lo) & (img < hi)).all(dim=1)
            exp.append(img[ext & ~own])
        exp = torch.cat(exp)
        got = self.halo_rows.to(all_wrapped.dtype)
        if got.shape != exp.shape:
            return False
        key = lambda t: t[np.lexsort((t[:, 2].numpy(), t[:, 1].numpy(), t[:, 0].numpy()))]  # noqa: E731
        return bool(torch.allclose(key(got), key(exp), atol=1e-9))


class DomainSet:
    """Drives the domains of this process: one (`DistTransport`) or all of them (`LocalTransport`)."""

    def __init__(self, box, world, device, dtype, terms, cutoff, A=None, B=None, skin=1.5, grid=None, transport=None,
                 dry=False, **engine_kwargs):
        self.dry = bool(dry)  # CPU tensors, no force engine (DryDomain): the plumbing of an N-rank run without a GPU
        self.grid = BrickGrid(box, world, grid)
        self.device, self.dtype = torch.device(device), dtype
        self.transport = transport if transport is not None else LocalTransport(world)
        self.local = isinstance(self.transport, LocalTransport)
        ranks = range(world) if self.local else [self.transport.rank]
        if not self.local:
            self.transport.bind(self.device)
        kind = DryDomain if self.dry else Domain
        mk = lambda r: kind(self.grid, r, self.device, dtype, terms, cutoff, skin, A, B, engine_kwargs)  # noqa: E731
        self.domains = {r: mk(r) for r in ranks}
        self.migrations = 0
        self._recv_counts = {}
        self._nstep = 0
        self._since_migration = 0
        self.check_every = 4  # steps between migration-trigger collectives
        self._pending = None  # (event, pinned host value, step since migration) of the last displacement read-back
        self._host_ring, self._ring_pos = None, 0
        # recovery: the atoms' state is saved at the entry of `step` and after every migration (6 small copies); when a
        # measured displacement turns out to lie beyond the halo's half skin, or a brick's list turns out invalid, the
        # steps since then are repeated from that state — after a halo overrun with half the `check_every`
        self.recover = True
        self.max_recoveries = 6  # per `step` call
        self.recoveries = 0  # (total, for reports)
        self._saved = None
        # a halo overrun halves `check_every`; `regrow_after` clean migrations in a row give one step of it back, up to
        # the value it had before the first overrun (one hot transient must not cost an all-reduce per step for ever)
        self.regrow_after = 16
        self._check_every_ceiling = None
        self._clean_migrations = 0

    # -- setup: every rank holds the same global arrays and keeps its brick ---------------------
    def scatter(self, pos, vel, charges, types, masses):
        pos = torch.as_tensor(pos, dtype=self.dtype, device=self.device)
        owner, _ = self.grid.owner(pos)
        for r, dom in self.domains.items():
            sel = torch.nonzero(owner == r).flatten()
            f = lambda x, dt=self.dtype: torch.as_tensor(x, device=self.device).to(dt)[sel]  # noqa: E731
            dom.adopt(sel, pos[sel], f(vel), f(charges), torch.as_tensor(types, device=self.device).long()[sel],
                      f(masses))
        self._exchange(static=True)

    # -- communication --------------------------------------------------------------------------
    def _all_to_all(self, key, payloads, counts):
        """payloads/counts: {rank: tensor/list}; returns {rank: received rows}."""
        if not self.local:
            r = self.transport.rank
            if key not in self._recv_counts:
                self._recv_counts[key] = self.transport.exchange_counts(counts[r])
            return {r: self.transport.all_to_all(payloads[r], counts[r], self._recv_counts[key])}
        out = {}
        for dst in range(self.grid.world):
            parts = []
            for src in range(self.grid.world):
                off = sum(counts[src][:dst])
                parts.append(payloads[src][off: off + counts[src][dst]])
            out[dst] = torch.cat(parts, dim=0)
        return out

    def _exchange(self, static):
        counts = {r: d.plan.send_counts for r, d in self.domains.items()}
        if static:  # after a (re)distribution: positions + charge + type, engines re-created for the new atom sets
            self._recv_counts = {}
            payloads = {r: d.halo_payload(True) for r, d in self.domains.items()}
            got = self._all_to_all("halo", payloads, counts)
            for r, d in self.domains.items():
                d.set_halo(got[r], True)
            return
        # every step: pack on the device, receive in place
        if not self.local:
            r, d = next(iter(self.domains.items()))
            self.transport.all_to_all_into(d.halo_rows, d.pack_halo(), counts[r], self._recv_counts["halo"])
            return
        sent = {r: d.pack_halo() for r, d in self.domains.items()}
        for dst, d in self.domains.items():
            out, o = d.halo_rows, 0
            for src in range(self.grid.world):
                off, cnt = sum(counts[src][:dst]), counts[src][dst]
                out[o: o + cnt] = sent[src][off: off + cnt]
                o += cnt

    def _any(self, flags):
        if self.local:
            return any(bool(f) for f in flags.values())
        return self.transport.any_true(next(iter(flags.values())))

    def _migration_due(self):
        """Checked every `check_every` steps without stalling the step loop: the running maximum of the squared
        displacement (kept on the device by `tmdhip_dd_step`) is max-reduced over the ranks and copied to pinned
        host memory asynchronously; the value is looked at one check later.  A migration is requested early
        enough that the halo stays complete until the decision after that one: the displacement measured at
        step `at` (since the last migration) plus twice its average growth over the 2 x `check_every` steps
        that pass until then must stay below skin/2."""
        self._since_migration += 1
        k = self.check_every
        if self._since_migration % k:
            return False
        due = False
        limit = 0.5 * next(iter(self.domains.values())).skin

        def measured(host):
            """What was measured must never have crossed the limit already (the extrapolation is only a prediction:
            hot atoms, a larger check_every, a changed time step) — halo atoms would be missing, forces wrong."""
            moved = float(host.item()) ** 0.5
            if moved > limit:
                self._pending = None
                raise HaloOverrun(f"domain decomposition: an atom moved {moved:.3f} A since the last migration, beyond "
                                  f"the halo's half skin of {limit:.3f} A, before a migration was requested; the forces "
                                  "of the last steps are invalid (use a larger halo skin or a smaller check_every)")
            return moved

        if self._pending is not None:
            ev, host, at = self._pending
            ev.synchronize()  # recorded check_every steps ago: long done
            ahead = 1.0 + 2.0 * (self._since_migration + k - at) / at
            due = measured(host) * ahead > limit
        if not due:
            first_check = self._pending is None  # first boundary after a migration: nothing measured yet
            if self.local:
                t = torch.stack([d.disp2[0] for d in self.domains.values()]).max().reshape(1)
            else:  # reduced in place: the flag becomes the maximum over all ranks, which is what it is used for
                t = self.transport.max_(next(iter(self.domains.values())).disp2)
            if self._host_ring is None:  # two pinned slots and events, reused alternately
                if self.device.type == "cuda":
                    self._host_ring = [(torch.empty(1, dtype=torch.float32, pin_memory=True), torch.cuda.Event())
                                       for _ in range(2)]
                else:  # (dry run on CPU tensors: nothing is asynchronous)
                    done = SimpleNamespace(synchronize=lambda: None, record=lambda *a: None)
                    self._host_ring = [(torch.empty(1, dtype=torch.float32), done) for _ in range(2)]
            self._ring_pos ^= 1
            host, ev = self._host_ring[self._ring_pos]
            host.copy_(t, non_blocking=True)
            ev.record(torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None)
            self._pending = (ev, host, self._since_migration)
            if first_check:
                # one synchronous look, so that a migration can be requested now already instead of two periods
                # after the last one (the value is the maximum over all ranks: every rank decides alike)
                ev.synchronize()
                due = measured(host) * (1.0 + 2.0 * k / self._since_migration) > limit
        return due

    def _native_migration(self):
        """Migration in the library (tmdhip_dd_migrate) where the library's communicator exists: over RCCL (one brick
        per process) or in-process (one host thread per brick).  TMDHIP_DD_MIGRATE=python keeps the torch version."""
        import os
        import threading

        if os.environ.get("TMDHIP_DD_MIGRATE", "native") == "python" or self.device.type != "cuda":
            return False
        if not self.local:
            comm = self.transport.native()
            if comm is None:
                return False
            d = next(iter(self.domains.values()))
            d.migrate_native(comm, torch.cuda.current_stream(self.device).cuda_stream)
            self._recv_counts["halo"] = d.recv_counts
            return True
        if not self.transport.native_threads:
            return False
        comms, streams = self.transport.native(self.device)
        torch.cuda.synchronize(self.device)
        errors = {}

        def work(r):
            try:
                with torch.cuda.device(self.device), torch.cuda.stream(streams[r]):
                    self.domains[r].migrate_native(comms[r], streams[r].cuda_stream)
            except Exception as exc:  # noqa: BLE001
                errors[r] = repr(exc)

        threads = [threading.Thread(target=work, args=(r,)) for r in self.domains]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        torch.cuda.synchronize(self.device)
        if errors:
            if self.local:
                self.transport.close()  # (a broken hub is not reused: the next call creates a new one)
            raise RuntimeError(f"tmdhip_dd_migrate failed on ranks {sorted(errors)}: {next(iter(errors.values()))}")
        return True

    # -- saved state / recovery -----------------------------------------------------------------------
    def _save_state(self, remaining, first):
        """The atoms of every brick (global id, position in the caller's periodic image, velocity, charge, type, mass)
        with the step counter and what is left of the running `step` call (`first`: the phases of its next iteration)."""
        rows = {r: (d.ids.clone(), d.pos + d.unwrap, d.vel.clone(), d.charges.clone(), d.types.clone(), d.masses.clone())
                for r, d in self.domains.items()}
        self._saved = (rows, self._nstep, remaining, first)

    def _restore_state(self):
        """Back to the saved state: its atoms, a migration from there (owners, halo, engines, reference positions),
        forces.  Returns what was left of the `step` call at that point."""
        rows, nstep, remaining, first = self._saved
        for r, d in self.domains.items():
            ids, pos, vel, q, t, m = rows[r]
            d.adopt(ids.clone(), pos.clone(), vel.clone(), q.clone(), t.clone(), m.clone())
        self._nstep = nstep
        self.migrate(verify=False)  # (adopt has set the reference positions: nothing to verify)
        self.compute_forces()
        return remaining, first

    def verify_halo(self):
        """The displacement since the last migration as it is NOW (maximum over all ranks; one host synchronisation).
        The step loop looks at a value measured `check_every` steps earlier and extrapolates; this is the exact test,
        made at every migration and available to a caller that wants the last steps of a batch certified."""
        doms = list(self.domains.values())
        if not doms or getattr(doms[0], "disp2", None) is None:
            return 0.0
        if self.local:
            t = torch.stack([d.disp2[0] for d in doms]).max()
        else:
            t = self.transport.max_(doms[0].disp2.clone())[0]
        moved = float(t.item()) ** 0.5
        limit = 0.5 * doms[0].skin
        if moved > limit:
            raise HaloOverrun(f"domain decomposition: an atom has moved {moved:.3f} A since the last migration, beyond the "
                              f"halo's half skin of {limit:.3f} A; the forces of the last steps are invalid")
        return moved

    def migrate(self, verify=True):
        """Re-assign atoms to bricks, rebuild halo plans and engines."""
        if verify and self.recover:
            self.verify_halo()
            # the lists of the steps since the last checkpoint, BEFORE the migration resets them (a new atom set forces a
            # rebuild and would absorb an overflow / outlived-skin report): a state saved after this migration is then one
            # whose forces were computed from complete lists (round 4's advisor; migrate() synchronises with the host anyway)
            self._lists_valid()
            self._clean_migrations += 1
            if self._check_every_ceiling and self._clean_migrations >= self.regrow_after and \
                    self.check_every < self._check_every_ceiling:
                self.check_every += 1
                self._clean_migrations = 0
        if self._native_migration():
            self.migrations += 1
            self._since_migration = 0
            self._pending = None
            return
        payloads, counts = {}, {}
        for r, d in self.domains.items():
            owner, _ = self.grid.owner(d.pos)
            order = torch.argsort(owner, stable=True)
            payloads[r] = d.state_rows()[order]
            counts[r] = torch.bincount(owner, minlength=self.grid.world).tolist()
        self._recv_counts.pop("migrate", None)
        got = self._all_to_all("migrate", payloads, counts)
        for r, d in self.domains.items():
            rows = got[r]
            rows = rows[torch.argsort(rows[:, 0], stable=True)]  # deterministic local order: by global id
            d.from_rows(rows)
        self.migrations += 1
        self._since_migration = 0
        self._pending = None
        if not self.local and self.transport.native():
            from . import _lib as L

            L.check(L.load().tmdhip_dd_reset(self.transport.native()))
        if self.local and getattr(self.transport, "_comms", None):
            from . import _lib as L

            for h in self.transport._comms:
                L.check(L.load().tmdhip_dd_reset(h))
        self._exchange(static=True)

    # -- dynamics -------------------------------------------------------------------------------
    def compute_forces(self):
        for d in self.domains.values():
            d.compute()

    def step(self, niter, timestep_fs, gamma_ps=None, T=None, seed=0):
        """Velocity Verlet (+ Langevin) over the decomposed system; forces must be current on entry."""
        from . import _lib as L
        from .integrator import BOLTZMAN, PICOSEC2TIMEU, TIMEFACTOR

        lib = L.load()
        dt = timestep_fs / TIMEFACTOR
        code = L.dtype_code(self.dtype)
        gamma = gamma_ps / PICOSEC2TIMEU if gamma_ps is not None else 0.0
        stream = lambda: torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else None  # noqa: E731
        vnoise = float(np.sqrt(2.0 * gamma * BOLTZMAN * T * dt)) if T else 0.0

        def dd_step(d, phases):
            """phases 1: second half kick of the step that just got its forces; 2: first half of the next; 3: both."""
            if self.dry:  # forces are zero: no kick; the drift and the running displacement maximum in torch
                if phases & 2:
                    d.pos.add_(d.vel, alpha=dt)
                    if d.nown:
                        d.disp2[0] = max(float(d.disp2[0]), float(((d.pos - d.ref) ** 2).sum(dim=1).max()))
                return
            vc = 0
            if T and (phases & 1):
                if getattr(d, "_vc_key", None) != (vnoise, d.nown):
                    d._vc = (d.vcoeff_unit * vnoise).contiguous()
                    d._vc_key = (vnoise, d.nown)
                vc = d._vc.data_ptr()
            L.check(lib.tmdhip_dd_step(code, d.nown, d.pos.data_ptr(), d.vel.data_ptr(), d.forces.data_ptr(),
                                       d.masses.data_ptr(), vc, dt, gamma, seed + 7919 * d.rank, max(self._nstep - 1, 0),
                                       phases, d.ref.data_ptr(), d.disp2.data_ptr(), stream()))

        comm = None if (self.local or self.dry) else self.transport.native()

        def run(remaining, first):
            if comm is not None:
                return self._step_native(comm, remaining, first, dt, gamma, vnoise if T else None, seed)
            if self.local and self.transport.native_threads and self.device.type == "cuda":
                return self._step_native_threads(remaining, first, dt, gamma, vnoise if T else None, seed)
            for it in range(remaining):
                for d in self.domains.values():
                    dd_step(d, first if it == 0 else 3)
                migrated = self._migration_due()
                if migrated:
                    self.migrate()
                else:
                    self._exchange(static=False)
                self.compute_forces()
                self._nstep += 1
                if migrated and self.recover:
                    self._save_state(remaining - it - 1, 3)
            if remaining > 0 or first == 3:  # (nothing was drifted otherwise: a lone second half kick would be applied twice)
                for d in self.domains.values():
                    dd_step(d, 1)
            if remaining > 0:
                self._lists_valid()

        remaining, first = niter, 2
        if self.recover and niter > 0:
            self._save_state(remaining, first)
        recovered = 0
        while True:
            try:
                run(remaining, first)
                if self.recover and niter > 0:
                    # the exact test on the way out: a batch that ends between two looks of the loop is certified too, and
                    # the state saved at the entry of the next call is a valid one
                    self.verify_halo()
                return
            except DomainReplay as exc:
                # every rank gets here at the same iteration: the displacement is a maximum over the ranks, and a
                # brick's invalid list is reported to all of them (`_lists_valid`)
                halo = isinstance(exc, HaloOverrun)
                if not self.recover or self._saved is None or recovered >= self.max_recoveries or \
                        (halo and self.check_every == 1):
                    raise
                recovered += 1
                self.recoveries += 1
                if halo:
                    if self._check_every_ceiling is None:
                        self._check_every_ceiling = self.check_every
                    self.check_every = max(1, self.check_every // 2)
                    self._clean_migrations = 0
                remaining, first = self._restore_state()

    def _dd_desc(self, d, recv_counts, remaining, first, dt, gamma, vnoise, seed):
        """The `tmdhip_dd_desc` of one brick (+ the ctypes arrays it points into, which must outlive the call)."""
        import ctypes as C

        from . import _lib as L

        eng = d.forces_engine._engine(d.local_pos)
        if not eng.stores_forces:
            raise RuntimeError("domain decomposition needs the cell-list engine (brick too small)")
        vc = 0
        if vnoise is not None:
            if getattr(d, "_vc_key", None) != (vnoise, d.nown):
                d._vc = (d.vcoeff_unit * vnoise).contiguous()
                d._vc_key = (vnoise, d.nown)
            vc = d._vc.data_ptr()
        sc = (C.c_int64 * self.grid.world)(*d.plan.send_counts)
        rc_ = (C.c_int64 * self.grid.world)(*recv_counts)
        desc = L.DdDesc(
            struct_size=C.sizeof(L.DdDesc), dtype=L.dtype_code(self.dtype), niter=remaining, first_phases=first,
            check_every=self.check_every, nown=d.nown, nhalo=d.local_pos.shape[1] - d.nown,
            pos_dev=d.local_pos.data_ptr(), vel_dev=d.vel.data_ptr(), forces_dev=d.local_forces.data_ptr(),
            mass_dev=d.masses.data_ptr(), vcoeff_dev=vc, ref_dev=d.ref.data_ptr(), disp2_dev=d.disp2.data_ptr(),
            dt=dt, gamma=gamma, seed=seed + 7919 * d.rank, step0=self._nstep, nsend=len(d.send_index32),
            send_index_dev=d.send_index32.data_ptr(), send_shift_dev=d.send_shift.data_ptr(),
            send_buf_dev=d.send_buf.data_ptr(), send_counts_host=C.addressof(sc), recv_counts_host=C.addressof(rc_),
            skin=d.skin, since_migration=self._since_migration,
        )
        return eng, desc, (sc, rc_)

    def _lists_valid(self):
        """The neighbour lists of the batch that just ended, on every rank (a verdict all ranks share)."""
        from . import _lib as L

        bad = False
        for d in self.domains.values():
            if getattr(d, "forces_engine", None) is None:
                continue
            bad = bad or not d.forces_engine._verify(d.forces_engine._engine(d.local_pos), d.local_pos)
        if not self.local:  # (every rank must take the same way: the verdict is the OR over the ranks)
            bad = self.transport.any_true(torch.tensor([1.0 if bad else 0.0], device=self.device))
        if bad:
            raise ListInvalid(f"a neighbour list of a brick became invalid during the batch ({L.last_error()}): "
                              "repeat the batch")

    def _step_native_threads(self, niter, first, dt, gamma, vnoise, seed):
        """All bricks in this process, each brick's loop enqueued from C by a host thread of its own over the
        in-process communicator (`LocalTransport(native_threads=True)`); Python takes over for a migration."""
        import ctypes as C
        import threading

        from . import _lib as L

        lib = L.load()
        comms, streams = self.transport.native(self.device)
        world = self.grid.world
        remaining = niter
        while True:
            jobs = {}
            for r, d in self.domains.items():
                recv_counts = [self.domains[src].plan.send_counts[r] for src in range(world)]
                jobs[r] = self._dd_desc(d, recv_counts, remaining, first, dt, gamma, vnoise, seed)
            torch.cuda.synchronize(self.device)  # the ranks' streams do not order themselves behind the default stream
            results = {}

            def work(r):
                eng, desc, _keep = jobs[r]
                done = C.c_int32(0)
                try:
                    with torch.cuda.device(self.device):
                        rc = lib.tmdhip_dd_run(eng.ctx, comms[r], C.byref(desc), C.byref(done), streams[r].cuda_stream)
                    results[r] = (rc, done.value, L.last_error() if rc < 0 or rc == L.DD_OVERRUN else "")
                except Exception as exc:  # noqa: BLE001
                    results[r] = (-99, 0, repr(exc))

            threads = [threading.Thread(target=work, args=(r,)) for r in self.domains]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
            torch.cuda.synchronize(self.device)
            bad = {r: v for r, v in results.items() if v[0] < 0}
            if bad:
                self.transport.close()  # (the hub is broken for good: the next use creates a new one with new communicators)
                raise RuntimeError(f"tmdhip_dd_run failed on ranks {sorted(bad)}: {next(iter(bad.values()))[2]}")
            rcs, dones = {v[0] for v in results.values()}, {v[1] for v in results.values()}
            if len(rcs) != 1 or len(dones) != 1:
                raise RuntimeError(f"the ranks of the brick grid disagree about the migration: {results}")
            rc, done = rcs.pop(), dones.pop()
            for d in self.domains.values():
                d.forces = d.local_forces[0, : d.nown]
            if rc == L.DD_OVERRUN:
                raise HaloOverrun(next(iter(results.values()))[2])
            self._nstep += done
            self._since_migration += done + (1 if rc == 1 else 0)
            remaining -= done
            if rc == 0:
                break
            self.migrate()  # the iteration that has already drifted: new bricks, new halo, then its forces
            self.compute_forces()
            self._nstep += 1
            remaining -= 1
            first = 3
            if self.recover:
                self._save_state(remaining, first)
        self._lists_valid()

    def _step_native(self, comm, niter, first, dt, gamma, vnoise, seed):
        """`niter` iterations enqueued from C (`tmdhip_dd_run`: RCCL send/recv on the compute stream); Python
        only takes over for a migration."""
        import ctypes as C

        from . import _lib as L

        lib = L.load()
        d = next(iter(self.domains.values()))
        remaining = niter
        done = C.c_int32(0)
        while True:
            eng, desc, _keep = self._dd_desc(d, self._recv_counts["halo"], remaining, first, dt, gamma, vnoise, seed)
            with torch.cuda.device(self.device):
                rc = L.check(lib.tmdhip_dd_run(eng.ctx, comm, C.byref(desc), C.byref(done),
                                               torch.cuda.current_stream(self.device).cuda_stream), "tmdhip_dd_run")
            d.forces = d.local_forces[0, : d.nown]
            if rc == L.DD_OVERRUN:
                raise HaloOverrun(L.last_error())
            self._nstep += done.value
            self._since_migration += done.value + (1 if rc == 1 else 0)
            remaining -= done.value
            if rc == 0:
                break
            self.migrate()  # the iteration that has already drifted: new bricks, new halo, then its forces
            self.compute_forces()
            self._nstep += 1
            remaining -= 1
            first = 3
            if self.recover:
                self._save_state(remaining, first)
        self._lists_valid()

    # -- gathering (tests / output) ---------------------------------------------------------------
    def gather(self, natoms):
        """Global [N,3] positions, velocities and forces ordered by atom id (LocalTransport only)."""
        pos = torch.zeros(natoms, 3, dtype=self.dtype, device=self.device)
        vel, frc = torch.zeros_like(pos), torch.zeros_like(pos)
        for d in self.domains.values():
            pos[d.ids], vel[d.ids], frc[d.ids] = d.pos + d.unwrap, d.vel, d.forces
        return pos, vel, frc
