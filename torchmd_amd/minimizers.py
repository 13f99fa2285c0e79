"""Energy minimisers driving `Forces.compute` (mirror of the reference module `torchmd/minimizers.py`:
same function names, arguments and effect on `system.pos`).  They are callers of the hot path, not part
of it: every energy/force evaluation is one `forces.compute(pos, box, forces)` on the device; the
optimisation logic runs on the host (scipy L-BFGS-B), in torch (LBFGS on the differentiable potential)
or as a few tensor operations per line-search point (conjugate gradient).
"""

from __future__ import annotations

import logging

import numpy as np
import torch

logger = logging.getLogger(__name__)

_GOLDEN = 0.618033988749895  # (sqrt(5) - 1) / 2


def minimize_bfgs(system, forces, fmax=0.5, steps=1000):
    """scipy L-BFGS-B on the potential of the single replica (reference minimizers.py:8-51):
    `gtol = fmax`, `maxiter = steps`; the minimum is written to `system.pos`."""
    from scipy.optimize import minimize

    if steps == 0:
        return None
    if system.pos.shape[0] != 1:
        raise RuntimeError("System minimization currently doesn't support replicas")
    n = system.pos.shape[1]
    count = [0]

    def fun(x):
        system.pos[:] = torch.as_tensor(x.reshape(1, n, 3), dtype=system.pos.dtype, device=system.pos.device)
        e = forces.compute(system.pos, system.box, system.forces)[0]
        g = -system.forces.detach().cpu().numpy().astype(np.float64)[0]
        logger.info("%4d   % 3.6f   % 3.6f", count[0], e, np.max(np.linalg.norm(g, axis=1)))
        count[0] += 1
        return float(e), g.reshape(-1)

    x0 = system.pos.detach().cpu().numpy().astype(np.float64).reshape(-1)
    res = minimize(fun, x0, method="L-BFGS-B", jac=True, options={"gtol": fmax, "maxiter": steps, "disp": False})
    system.pos[:] = torch.as_tensor(res.x.reshape(1, n, 3), dtype=system.pos.dtype, device=system.pos.device)
    return res


def minimize_pytorch_bfgs(system, calculator, steps=10, max_iter=20, tolerance_change=1e-9):
    """`torch.optim.LBFGS` on the summed potential of all replicas, through the differentiable
    `compute(..., toNumpy=False)` (reference minimizers.py:54-96).  Returns the energies seen, shape
    [nreplicas, evaluations]."""
    if steps == 0:
        return None
    x = system.pos.detach().clone().requires_grad_(True)
    opt = torch.optim.LBFGS([x], max_iter=max_iter, tolerance_change=tolerance_change)
    seen = []

    def closure():
        opt.zero_grad()
        pots = calculator.compute(x, system.box, system.forces, explicit_forces=False, toNumpy=False)
        pots = torch.stack([p.reshape(()) for p in pots]) if isinstance(pots, (list, tuple)) else pots.reshape(-1)
        seen.append(pots.detach().cpu().numpy())
        total = pots.sum()
        if x.grad is None and total.requires_grad:
            total.backward()
        elif not total.requires_grad:  # calculator without autograd support: use its explicit forces
            x.grad = -system.forces.detach().clone()
        return total

    for _ in range(steps):
        opt.step(closure)
    with torch.no_grad():
        system.pos[:] = x.detach()
    return np.stack(seen, axis=1)


def _energy_forces(forces, system, pos):
    e = forces.compute(pos, system.box, system.forces)[0]
    return float(e), system.forces.detach()[0].clone()


def _line_minimum(forces, system, start, direction, u0, max_disp=1.0, tol=1e-2):
    """Golden-section search for the minimum of U(start + a * direction), a in [0, max_disp / max |d_i|]
    (no atom moves further than `max_disp` Angstrom per line search; reference minimizers.py:108-262)."""
    dmax = float(torch.sqrt((direction**2).sum(dim=1).max()))
    if dmax == 0.0:
        return start, u0
    lo, hi = 0.0, max_disp / dmax
    width0 = hi - lo

    def energy(a):
        return _energy_forces(forces, system, (start + a * direction)[None])[0]

    a1, a2 = hi - _GOLDEN * (hi - lo), lo + _GOLDEN * (hi - lo)
    u1, u2 = energy(a1), energy(a2)
    best_a, best_u = 0.0, u0
    while (hi - lo) > tol * width0:
        if u1 < u2:
            hi, a2, u2 = a2, a1, u1
            a1 = hi - _GOLDEN * (hi - lo)
            u1 = energy(a1)
        else:
            lo, a1, u1 = a1, a2, u2
            a2 = lo + _GOLDEN * (hi - lo)
            u2 = energy(a2)
        for a, u in ((a1, u1), (a2, u2)):
            if u < best_u:
                best_a, best_u = a, u
    return start + best_a * direction, best_u


def minimize_cg(system, forces, steps=1000, start_step: int = 0, threshold=None):
    """Fletcher-Reeves conjugate gradient with a golden-section line search (reference
    minimizers.py:264-310).  Returns the index of the last step taken; stops early when the largest force
    component drops below `threshold`."""
    if system.pos.shape[0] != 1:
        raise RuntimeError("System minimization currently doesn't support replicas")
    pos = system.pos.detach()[0].clone()
    u, frc = _energy_forces(forces, system, pos[None])
    direction = frc.clone()
    fdf = float((frc**2).sum())
    last = start_step
    for step in range(start_step, steps):
        last = step
        pos, u = _line_minimum(forces, system, pos, direction, u)
        u, frc = _energy_forces(forces, system, pos[None])
        new_fdf = float((frc**2).sum())
        beta = new_fdf / fdf if fdf > 0 else 0.0
        fdf = new_fdf
        direction = frc + beta * direction
        fmax = float(frc.abs().max())
        logger.info("%12d %14.4f %16.4f", step, u, fmax)
        if threshold is not None and fmax < threshold:
            break
    with torch.no_grad():
        system.pos[0] = pos
    forces.compute(system.pos, system.box, system.forces)
    return last
