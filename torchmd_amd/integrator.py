"""Velocity-Verlet / Langevin integrator on HIP kernels.

Mirror of the reference module (`torchmd/integrator.py`): same constants, helper functions,
`Integrator(systems, forces, timestep, device, gamma=None, T=None, batch=None)` constructor and
`step(niter) -> (Ekin, pot, T)` contract.  Each iteration is

    tmdhip_first_vv  ->  forces.compute  ->  tmdhip_langevin_second_vv | tmdhip_second_vv

(integrator.py:115-120).  With this package's `Forces` the loop enqueues everything asynchronously
and only reads energies back after the last iteration (the reference returns the energies of the last
`compute()` only, integrator.py:125); any other object with a `.compute(pos, box, forces)` method
(the duck type shown by tests/test_integrator.py:155-158) is called as in the reference.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L

TIMEFACTOR = 48.88821
BOLTZMAN = 0.001987191
PICOSEC2TIMEU = 1000.0 / TIMEFACTOR


def kinetic_energy(masses, vel, batch=None):
    """Kinetic energy per replica (nreplicas, 1), or per replica and atom group (nreplicas, nbatches)
    when `batch` (natoms,) assigns atoms to groups — reference integrator.py:8-43.  Analysis helper
    on torch tensors (any device); `Integrator.step` uses the fused HIP reduction instead."""
    if vel.dim() != 3:
        raise ValueError(f"vel must be 3D (nreplicas, natoms, 3), got {vel.dim()}D")
    per_atom = 0.5 * masses * torch.sum(vel * vel, dim=2, keepdim=True)
    if batch is None:
        return torch.sum(per_atom, dim=1)
    nbatch = int(torch.max(batch).item() + 1)
    out = torch.zeros(vel.shape[0], nbatch, device=vel.device, dtype=vel.dtype)
    out.index_add_(1, batch, per_atom[:, :, 0])
    return out


def maxwell_boltzmann(masses, T, replicas=1):
    """Velocities ~ N(0, sqrt(kB T / m)) per replica (reference integrator.py:46-54)."""
    natoms = len(masses)
    scale = torch.sqrt(T * BOLTZMAN / masses)
    return torch.stack([scale * torch.randn((natoms, 3)).type_as(masses) for _ in range(replicas)], dim=0)


def kinetic_to_temp(Ekin, natoms):
    return 2.0 / (3.0 * natoms * BOLTZMAN) * Ekin


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _first_VV(pos, vel, force, mass, dt):
    """pos += vel*dt + 0.5*(F/m)*dt^2 ; vel += 0.5*dt*(F/m)   (integrator.py:61-64), one kernel."""
    lib = L.load()
    for name, t in (("pos", pos), ("vel", vel), ("force", force), ("mass", mass)):
        L.require_device_tensor(t, name)
    R, N = pos.shape[0], pos.shape[1]
    with torch.cuda.device(pos.device):
        L.check(
            lib.tmdhip_first_vv(
                L.dtype_code(pos.dtype), R, N, pos.data_ptr(), vel.data_ptr(), force.data_ptr(), mass.data_ptr(),
                float(dt), _stream(pos.device),
            ),
            "tmdhip_first_vv",
        )


def _second_VV(vel, force, mass, dt):
    """vel += 0.5*dt*(F/m)   (integrator.py:67-69)."""
    lib = L.load()
    for name, t in (("vel", vel), ("force", force), ("mass", mass)):
        L.require_device_tensor(t, name)
    R, N = vel.shape[0], vel.shape[1]
    with torch.cuda.device(vel.device):
        L.check(
            lib.tmdhip_second_vv(
                L.dtype_code(vel.dtype), R, N, vel.data_ptr(), force.data_ptr(), mass.data_ptr(), float(dt),
                _stream(vel.device),
            ),
            "tmdhip_second_vv",
        )


_REPLAY_FAILED = ("Integrator.step(): the batch of steps was rewound and repeated once and failed again; the trajectory "
                  "since the previous step() call is invalid (restart from the last saved state).  The library says: ")
# (the step-by-step loop over a duck-typed / external force has no saved entry state: nothing was rewound)
_LIST_INVALID = ("Integrator.step(): a neighbour list overflowed or outlived its skin during this call; the trajectory since "
                 "the previous step() call is invalid (restart from the last saved state; the list capacity has been grown)")


class Integrator:
    def __init__(self, systems, forces, timestep, device, gamma=None, T=None, batch=None):
        self.dt = timestep / TIMEFACTOR
        self.systems = systems
        self.forces = forces
        self.device = device
        if gamma is not None:
            gamma = gamma / PICOSEC2TIMEU
        self.gamma = gamma
        self.T = T
        if torch.any(systems.masses != 0):
            self.masses = systems.masses
        else:
            self.masses = torch.as_tensor(self.forces.par.masses).detach().clone()
            self.masses = self.masses.to(device=device, dtype=systems.pos.dtype).view(-1, 1)
        self.masses = self.masses.contiguous()
        if T:
            if gamma is None:
                raise RuntimeError("Langevin temperature T requires a friction gamma")
            self.vcoeff = torch.sqrt(2.0 * gamma / self.masses * BOLTZMAN * T * self.dt).to(device).contiguous()
        self.batch = batch
        if batch is not None:
            self.natoms = torch.bincount(batch).cpu().numpy()
        else:
            self.natoms = len(self.masses)
        # noise stream: seeded from torch's global generator so torch.manual_seed() reproduces runs
        self._seed = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
        self._nstep = 0
        self._ke = None
        self.replays = 0  # batches that were rewound and repeated (list validity failure / step-block time-out)

    def _check_layout(self):
        s = self.systems
        for name in ("pos", "vel", "forces"):
            t = getattr(s, name)
            L.require_device_tensor(t, f"systems.{name}")
            if not t.is_contiguous():
                raise RuntimeError(f"systems.{name} must be contiguous (it is updated in place by the HIP kernels)")
            if t.dtype != s.pos.dtype:
                raise RuntimeError("systems.pos/vel/forces must share one dtype")
        if self.masses.device != s.pos.device or self.masses.dtype != s.pos.dtype:
            self.masses = self.masses.to(device=s.pos.device, dtype=s.pos.dtype).contiguous()
            if self.T:
                self.vcoeff = self.vcoeff.to(device=s.pos.device, dtype=s.pos.dtype).contiguous()

    def step(self, niter=1):
        from .forces import Forces

        lib = L.load()
        s = self.systems
        self._check_layout()
        dev = s.pos.device
        code = L.dtype_code(s.pos.dtype)
        R, N = s.pos.shape[0], s.pos.shape[1]
        fast = isinstance(self.forces, Forces)
        fused = fast and not self.forces.external and niter > 0
        with torch.cuda.device(dev):
            return self._step_body(lib, s, dev, code, R, N, fast, fused, niter, replay=False)

    def _step_body(self, lib, s, dev, code, R, N, fast, fused, niter, replay):

        pot = None
        ebuf = ext = None
        if fused:
            # whole loop enqueued from C (tmdhip_md_run): fused half-kick/drift/displacement-test
            # kernels, no Python or ctypes work per step
            step0 = self._nstep - niter if replay else self._nstep
            ebuf = self.forces._md_run(
                s, self.masses, self.vcoeff if self.T else None, self.dt,
                float(self.gamma) if self.T else 0.0, self._seed, step0, niter, restore=replay,
            )
            if not replay:
                self._nstep += niter
        for it in range(0 if not fused else niter, niter):
            st = _stream(dev)
            L.check(
                lib.tmdhip_first_vv(code, R, N, s.pos.data_ptr(), s.vel.data_ptr(), s.forces.data_ptr(),
                                    self.masses.data_ptr(), self.dt, st),
                "tmdhip_first_vv",
            )
            if fast:
                ebuf, ext = self.forces._compute_async(s.pos, s.box, s.forces, want_energy=(it == niter - 1))
            else:
                pot = self.forces.compute(s.pos, s.box, s.forces)
            if self.T:
                L.check(
                    lib.tmdhip_langevin_second_vv(code, R, N, s.vel.data_ptr(), s.forces.data_ptr(),
                                                  self.masses.data_ptr(), self.vcoeff.data_ptr(), self.dt,
                                                  float(self.gamma), self._seed, self._nstep, st),
                    "tmdhip_langevin_second_vv",
                )
            else:
                L.check(
                    lib.tmdhip_second_vv(code, R, N, s.vel.data_ptr(), s.forces.data_ptr(),
                                         self.masses.data_ptr(), self.dt, st),
                    "tmdhip_second_vv",
                )
            self._nstep += 1

        eng = self.forces._engine(s.pos) if (fast and niter > 0) else None
        if fused and self.batch is None:
            # kinetic energy + energies of the last step + neighbour-list validity: ONE C call, ONE read-back,
            # ONE host synchronisation (tmdhip_md_observe)
            obs = np.empty((R, L.NENERGY + 1), dtype=np.float64)
            rc = L.check(
                # (AFTER_RUN: nothing has touched the velocities since the tmdhip_md_run a few lines up returned)
                lib.tmdhip_md_observe(eng.ctx, s.vel.data_ptr(), self.masses.data_ptr(), ebuf.data_ptr(),
                                      obs.ctypes.data_as(C.POINTER(C.c_double)), L.OBSERVE_AFTER_RUN, _stream(dev)),
                "tmdhip_md_observe",
            )
            if rc != 0:
                # a neighbour list was truncated or outlived its skin between two scheduled rebuilds, or a step block
                # of a fused pair + step launch timed out: rewind to the entry state (saved by tmdhip_md_run) and
                # repeat the batch with the rebuild chain on every step (and, after a time-out, the separate
                # integrator kernel); the noise stream is counter based, so it is the same trajectory
                if not replay:
                    self.replays += 1
                    return self._step_body(lib, s, dev, code, R, N, fast, fused, niter, replay=True)
                raise RuntimeError(_REPLAY_FAILED + L.last_error())
            cols = self.forces.energy_columns()
            tot = obs[:, cols].sum(axis=1) if cols else np.zeros(R)
            pot = [float(v) for v in tot]
            Ekin = obs[:, L.NENERGY].copy()
            Ekin = Ekin.astype(np.dtype("float32") if s.pos.dtype == torch.float32 else np.float64)
            return Ekin, pot, kinetic_to_temp(Ekin, self.natoms)
        if self.batch is None:
            if eng is not None:
                kebuf = eng.kebuf  # shares one buffer with the energies: a single read-back below
            else:
                if self._ke is None or self._ke.shape[0] != R or self._ke.device != dev:
                    self._ke = torch.zeros(R, dtype=torch.float64, device=dev)
                kebuf = self._ke
            L.check(
                lib.tmdhip_kinetic_energy(code, R, N, s.vel.data_ptr(), self.masses.data_ptr(),
                                          kebuf.data_ptr(), _stream(dev)),
                "tmdhip_kinetic_energy",
            )
            ke = kebuf
        else:
            ke = kinetic_energy(self.masses, s.vel, self.batch).flatten().to(torch.float64)
        if eng is not None:
            if self.batch is None and ebuf is eng.ebuf:
                host = eng.comb.cpu().numpy()  # the only synchronising call of step()
                e = host[: R * L.NENERGY].reshape(R, L.NENERGY)
                Ekin = host[R * L.NENERGY:].copy()
                cols = self.forces.energy_columns()
                tot = e[:, cols].sum(axis=1) if cols else np.zeros(R)
                if ext is not None:
                    tot = tot + ext.cpu().numpy()
                pot = [float(v) for v in tot]
            else:
                tot = self.forces.total_energy_from(ebuf, ext)
                host = torch.cat([ke.flatten(), tot]).cpu().numpy()
                Ekin, pot = host[: ke.numel()], [float(v) for v in host[ke.numel():]]
            if not self.forces._verify(eng, s.pos):
                why = L.last_error()  # (tmdhip_check's verdict, set by judge_flags a moment ago)
                if fused and not replay:  # (batch mode of the fused loop: same rewind as above)
                    self.replays += 1
                    return self._step_body(lib, s, dev, code, R, N, fast, fused, niter, replay=True)
                raise RuntimeError((_REPLAY_FAILED + why) if (fused and replay) else (_LIST_INVALID + ": " + why))
        else:
            Ekin = ke.flatten().cpu().numpy()
        Ekin = Ekin.astype(np.dtype("float32") if s.pos.dtype == torch.float32 else np.float64)
        T = kinetic_to_temp(Ekin, self.natoms)
        return Ekin, pot, T
