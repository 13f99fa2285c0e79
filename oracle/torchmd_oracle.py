"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

A plain CPU restatement (torch CPU tensor ops, numpy for index work) of the reference TorchMD hot
path: the nonbonded + bonded force/energy evaluation of `torchmd/forces.py` and the velocity-Verlet /
Langevin step of `torchmd/integrator.py`.  Every function cites the reference lines it follows.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this
module, and only as the checker — never as the thing that is measured or shipped.  The product
(`torchmd_amd`) never imports it.

Parity status: PINNED.  `tests/golden/make_golden.py` runs the *reference itself* (imported from
/root/reference in the build container) on the reference's own fixtures and stores inputs+outputs in
`tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this oracle against those files and
against the literals the reference's tests / tutorial hold.

Why a restatement and not the reference: the reference materialises every i<j pair as a dense
[P,2] tensor (`forces.py:348-357`), which cannot be built for N ~ 1e5.  The oracle takes the pair
list as an *argument*; because the reference re-filters by `dist <= cutoff` each call
(`forces.py:266-269`), any superset of the in-cutoff pairs in the same (i asc, j asc) order gives
bit-identical results (measured max |dF| = 0.0).  The arithmetic below uses the same
torch ops in the same order as the reference so fp32 cut-off decisions and sums match bitwise.
"""

from __future__ import annotations

from math import pi

import numpy as np
import torch

# reference forces.py:375-378 evaluates this from scipy.constants (CODATA 2018, scipy 1.15.3):
#   1/(4 pi eps0) * e^2 / angstrom * N_A / (kilo*calorie)
ELEC_FACTOR = 332.06371307417066
TIMEFACTOR = 48.88821  # integrator.py:4
BOLTZMAN = 0.001987191  # integrator.py:5
PICOSEC2TIMEU = 1000.0 / TIMEFACTOR  # integrator.py:77

BONDED = ("bonds", "angles", "dihedrals", "impropers", "1-4")
NONBONDED = ("electrostatics", "lj", "repulsion", "repulsioncg")


# ----------------------------------------------------------------------------- pair lists
def exclusion_pairs(par, types=("bonds", "angles", "1-4")):
    """[E,2] int64 excluded pairs (unordered, may contain duplicates) — parameters.py:89-107."""
    ex = par.get_exclusions(types)
    return np.asarray(ex, dtype=np.int64).reshape(-1, 2)


def _pair_keys(pairs, n):
    lo = np.minimum(pairs[:, 0], pairs[:, 1])
    hi = np.maximum(pairs[:, 0], pairs[:, 1])
    return lo * np.int64(n) + hi


def all_pairs(natoms, excl=None):
    """Dense pair list, i<j, row-major, minus exclusions (forces.py:348-357)."""
    i, j = np.triu_indices(natoms, k=1)
    pairs = np.stack([i, j], axis=1).astype(np.int64)
    if excl is not None and len(excl):
        keep = ~np.isin(_pair_keys(pairs, natoms), _pair_keys(excl, natoms))
        pairs = pairs[keep]
    return pairs


def candidate_pairs(pos, box, rlist, excl=None):
    """Sparse superset of the in-cutoff pairs: every non-excluded i<j with minimum-image distance
    <= rlist (float64 geometry), sorted (i asc, j asc) like the dense list.  `box` all-zero = open
    boundaries.  Uses scipy's cKDTree; only the *ordering and completeness* matter because the
    evaluation re-filters by the real cutoff."""
    from scipy.spatial import cKDTree

    pos = np.asarray(pos, dtype=np.float64)
    box = np.asarray(box, dtype=np.float64).reshape(-1)[:3] if np.ndim(box) == 1 else np.diag(np.asarray(box, dtype=np.float64))
    n = len(pos)
    if np.all(box == 0):
        tree = cKDTree(pos)
    else:
        wrapped = pos - np.floor(pos / box) * box
        wrapped = np.where(wrapped >= box, wrapped - box, wrapped)
        tree = cKDTree(wrapped, boxsize=box)
    pairs = tree.query_pairs(rlist, output_type="ndarray").astype(np.int64)
    pairs = np.stack([pairs.min(axis=1), pairs.max(axis=1)], axis=1)
    if excl is not None and len(excl):
        keep = ~np.isin(_pair_keys(pairs, n), _pair_keys(excl, n))
        pairs = pairs[keep]
    order = np.argsort(pairs[:, 0] * np.int64(n) + pairs[:, 1], kind="stable")
    return pairs[order]


# ----------------------------------------------------------------------------- geometry
def min_image(d, box):
    """forces.py:360-365 — skipped when box is None or all zero; torch.round = half-to-even."""
    if box is None or torch.all(box == 0):
        return d
    return d - box.unsqueeze(0) * torch.round(d / box.unsqueeze(0))


def pair_geometry(pos, idx, box):
    """forces.py:368-372 -> (dist [P], unit [P,3], vec [P,3]); vec = pos[i] - pos[j]."""
    vec = min_image(pos[idx[:, 0]] - pos[idx[:, 1]], box)
    dist = torch.norm(vec, dim=1)
    unit = vec / dist.unsqueeze(1)
    return dist, unit, vec


# ----------------------------------------------------------------------------- pair potentials
def lj_core(dist, aa, bb, scale, switch_dist, cutoff):
    """forces.py:390-415.  NOTE the explicit-force switching term carries an extra 1/dist
    (`pot*switch_deriv/dist`, line 410-412) — reproduced on purpose."""
    rinv1 = 1 / dist
    rinv6 = rinv1**6
    rinv12 = rinv6 * rinv6
    pot = ((aa * rinv12) - (bb * rinv6)) / scale
    force = (-12 * aa * rinv12 + 6 * bb * rinv6) * rinv1 / scale
    if switch_dist is not None and cutoff is not None:
        mask = dist > switch_dist
        t = (dist[mask] - switch_dist) / (cutoff - switch_dist)
        sw = 1 + t * t * t * (-10 + t * (15 - t * 6))
        dsw = t * t * (-30 + t * (60 - t * 30)) / (cutoff - switch_dist)
        force[mask] = sw * force[mask] + pot[mask] * dsw / dist[mask]
        pot[mask] = pot[mask] * sw
    return pot, force


def lj(dist, idx, types, A, B, switch_dist, cutoff):
    """forces.py:381-387."""
    t = types[idx]
    return lj_core(dist, A[t[:, 0], t[:, 1]], B[t[:, 0], t[:, 1]], 1, switch_dist, cutoff)


def repulsion(dist, idx, types, A):
    """forces.py:418-433."""
    t = types[idx]
    aa = A[t[:, 0], t[:, 1]]
    rinv1 = 1 / dist
    rinv6 = rinv1**6
    rinv12 = rinv6 * rinv6
    return (aa * rinv12) / 1, (-12 * aa * rinv12) * rinv1 / 1


def repulsion_cg(dist, idx, types, B):
    """forces.py:436-450."""
    t = types[idx]
    coef = B[t[:, 0], t[:, 1]]
    rinv1 = 1 / dist
    rinv6 = rinv1**6
    return (coef * rinv6) / 1, (-6 * coef * rinv6) * rinv1 / 1


def electrostatics(dist, idx, charges, scale=1, cutoff=None, rfa=False, solventDielectric=78.5):
    """forces.py:453-491 (plain Coulomb or reaction field)."""
    if rfa:
        denom = (2 * solventDielectric) + 1
        krf = (1 / cutoff**3) * (solventDielectric - 1) / denom
        crf = (1 / cutoff) * (3 * solventDielectric) / denom
        common = ELEC_FACTOR * charges[idx[:, 0]] * charges[idx[:, 1]] / scale
        dist2 = dist**2
        pot = common * ((1 / dist) + krf * dist2 - crf)
        force = common * (2 * krf * dist - 1 / dist2)
    else:
        pot = ELEC_FACTOR * charges[idx[:, 0]] * charges[idx[:, 1]] / dist / scale
        force = -pot / dist
    return pot, force


# ----------------------------------------------------------------------------- bonded terms
def bonds(dist, prm):
    """forces.py:494-503."""
    x = dist - prm[:, 1]
    return prm[:, 0] * (x**2), 2 * prm[:, 0] * x


def angles(r21, r23, prm):
    """forces.py:506-539."""
    k0, theta0 = prm[:, 0], prm[:, 1]
    dot = torch.sum(r23 * r21, dim=1)
    n23 = 1 / torch.norm(r23, dim=1)
    n21 = 1 / torch.norm(r21, dim=1)
    cos_t = torch.clamp(dot * n21 * n23, -1, 1)
    theta = torch.acos(cos_t)
    dth = theta - theta0
    pot = k0 * dth * dth
    sin_t = torch.sqrt(1.0 - cos_t * cos_t)
    coef = torch.zeros_like(sin_t)
    nz = sin_t != 0
    coef[nz] = -2.0 * k0[nz] * dth[nz] / sin_t[nz]
    f0 = coef[:, None] * (cos_t[:, None] * r21 * n21[:, None] - r23 * n23[:, None]) * n21[:, None]
    f2 = coef[:, None] * (cos_t[:, None] * r23 * n23[:, None] - r21 * n21[:, None]) * n23[:, None]
    return pot, (f0, -(f0 + f2), f2)


def torsions(r12, r23, r34, term_of, prm):
    """forces.py:542-605: `term_of[m]` = torsion that Fourier/harmonic term m belongs to."""
    cA = torch.cross(r12, r23, dim=1)
    cB = torch.cross(r23, r34, dim=1)
    cC = torch.cross(r23, cA, dim=1)
    nA, nB, nC = torch.norm(cA, dim=1), torch.norm(cB, dim=1), torch.norm(cC, dim=1)
    uB = cB / nB.unsqueeze(1)
    cosphi = torch.sum(cA * uB, dim=1) / nA
    sinphi = torch.sum(cC * uB, dim=1) / nC
    phi = -torch.atan2(sinphi, cosphi)
    nt = r12.shape[0]
    pot = torch.zeros(nt, dtype=r12.dtype)
    coeff = torch.zeros(nt, dtype=r12.dtype)
    k0, phi0, per = prm[:, 0], prm[:, 1], prm[:, 2]
    if torch.all(per > 0):  # AMBER
        ad = per * phi[term_of] - phi0
        pot = torch.scatter_add(pot, 0, term_of, k0 * (1 + torch.cos(ad)))
        coeff = torch.scatter_add(coeff, 0, term_of, -per * k0 * torch.sin(ad))
    else:  # CHARMM harmonic
        ad = phi[term_of] - phi0
        ad[ad < -pi] = ad[ad < -pi] + 2 * pi
        ad[ad > pi] = ad[ad > pi] - 2 * pi
        pot = torch.scatter_add(pot, 0, term_of, k0 * ad**2)
        coeff = torch.scatter_add(coeff, 0, term_of, 2 * k0 * ad)
    n23 = torch.norm(r23, dim=1)
    n23sq = n23**2
    ff0 = (-coeff * n23) / (nA**2)
    ff1 = torch.sum(r12 * r23, dim=1) / n23sq
    ff2 = torch.sum(r34 * r23, dim=1) / n23sq
    ff3 = (coeff * n23) / (nB**2)
    f0v = ff0.unsqueeze(1) * cA
    f3v = ff3.unsqueeze(1) * cB
    s = ff1.unsqueeze(1) * f0v - ff2.unsqueeze(1) * f3v
    return pot, (-f0v, f0v + s, f3v - s, -f3v)


# ----------------------------------------------------------------------------- Forces.compute
def compute(
    par,
    pos,
    box,
    terms,
    cutoff=None,
    rfa=False,
    solventDielectric=78.5,
    switch_dist=None,
    pairs=None,
    exclusions=("bonds", "angles", "1-4"),
    explicit_forces=True,
):
    """Explicit-force evaluation of one `Forces.compute(pos, box, forces, returnDetails=True)` call
    (forces.py:83-346) for CPU tensors.  `pos [R,N,3]`, `box [R,3,3]`; `pairs` = [P,2] int64 numpy
    or tensor (per replica list allowed) of non-excluded i<j candidates in (i,j) ascending order;
    None = dense all-pairs list.  Returns (list of per-term energy dicts (python floats), forces
    tensor [R,N,3], per-replica number of nonbonded pairs inside the cutoff).
    `explicit_forces=False` (forces.py:94-98, 328-336): `pos` must require gradients; no term scatters its
    analytic force, the forces are minus the autograd gradient of the summed energies — the flavour without
    the switching quirk of forces.py:410-412."""
    if not explicit_forces and not pos.requires_grad:
        raise RuntimeError("The positions passed don't require gradients. Please use pos.detach().requires_grad_(True) before passing.")
    terms = [t.lower() for t in terms]
    R, N = pos.shape[0], pos.shape[1]
    dt = pos.dtype
    if par.nonbonded_params is not None:
        # the reference fills par.A/par.B only when "lj" is requested (forces.py:45-46); the
        # repulsion terms then reuse tables left by an earlier Forces object — same values.
        A, B = par.get_AB()
    else:
        A, B = getattr(par, "A", None), getattr(par, "B", None)
    need_pairs = any(t in NONBONDED for t in terms)
    if need_pairs and pairs is None:
        pairs = all_pairs(N, exclusion_pairs(par, exclusions))
    forces = torch.zeros_like(pos)
    pots, npairs = [], []
    for r in range(R):
        spos = pos[r]
        sbox = box[r][torch.eye(3).bool()]
        pot = {t: torch.zeros(1, dtype=dt) for t in terms}
        F = forces[r]

        def scatter_pair(idx, unit, coef):
            if not explicit_forces:  # forces.py:140-143 etc.: every scatter sits under `if explicit_forces`
                return
            fv = unit * coef[:, None]
            F.index_add_(0, idx[:, 0], -fv)
            F.index_add_(0, idx[:, 1], fv)

        if "bonds" in terms and par.bond_params is not None:  # forces.py:122-143
            idx = par.bond_params["idx"]
            prm = par.bond_params["params"][par.bond_params["map"][:, 1]]
            d, u, _ = pair_geometry(spos, idx, sbox)
            if cutoff is not None:
                m = d <= cutoff
                d, u, idx, prm = d[m], u[m], idx[m], prm[m]
            E, fc = bonds(d, prm)
            pot["bonds"] = pot["bonds"] + E.sum()
            scatter_pair(idx, u, fc)
        if "angles" in terms and par.angle_params is not None:  # forces.py:145-161
            idx = par.angle_params["idx"]
            prm = par.angle_params["params"][par.angle_params["map"][:, 1]]
            _, _, r21 = pair_geometry(spos, idx[:, [0, 1]], sbox)
            _, _, r23 = pair_geometry(spos, idx[:, [2, 1]], sbox)
            E, ff = angles(r21, r23, prm)
            pot["angles"] = pot["angles"] + E.sum()
            for c in range(3 if explicit_forces else 0):
                F.index_add_(0, idx[:, c], ff[c])
        if "dihedrals" in terms and par.dihedral_params is not None:  # forces.py:163-183
            _torsion_block(spos, sbox, par.dihedral_params, pot, "dihedrals", F if explicit_forces else None)
        if "1-4" in terms and par.nonbonded_14_params is not None and len(par.nonbonded_14_params["idx"]):
            tab = par.nonbonded_14_params  # forces.py:185-236
            idx = tab["idx"]
            d, u, _ = pair_geometry(spos, idx, sbox)
            p = tab["params"][tab["map"][:, 1]]
            if "lj" in terms:
                E, fc = lj_core(d, p[:, 0], p[:, 1], p[:, 2], None, None)
                pot["lj"] = pot["lj"] + E.sum()
                scatter_pair(idx, u, fc)
            if "electrostatics" in terms:
                E, fc = electrostatics(d, idx, par.charges, p[:, 3], None, False, solventDielectric)
                pot["electrostatics"] = pot["electrostatics"] + E.sum()
                scatter_pair(idx, u, fc)
        if "impropers" in terms and par.improper_params is not None:  # forces.py:238-258
            _torsion_block(spos, sbox, par.improper_params, pot, "impropers", F if explicit_forces else None)

        nin = 0
        if need_pairs:  # forces.py:260-319
            pr = pairs[r] if isinstance(pairs, (list, tuple)) else pairs
            idx = torch.as_tensor(pr, dtype=torch.int64)
            if len(idx):
                d, u, _ = pair_geometry(spos, idx, sbox)
                if cutoff is not None:
                    m = d <= cutoff
                    d, u, idx = d[m], u[m], idx[m]
                nin = int(len(idx))
                for t in terms:
                    if t == "electrostatics":
                        E, fc = electrostatics(d, idx, par.charges, 1, cutoff, rfa, solventDielectric)
                    elif t == "lj":
                        E, fc = lj(d, idx, par.mapped_atom_types, A, B, switch_dist, cutoff)
                    elif t == "repulsion":
                        E, fc = repulsion(d, idx, par.mapped_atom_types, A)
                    elif t == "repulsioncg":
                        E, fc = repulsion_cg(d, idx, par.mapped_atom_types, B)
                    else:
                        continue
                    pot[t] = pot[t] + E.sum()
                    scatter_pair(idx, u, fc)
        npairs.append(nin)
        pots.append(pot)
    if not explicit_forces:  # forces.py:328-336
        enesum = torch.zeros(1, dtype=dt)
        for pot in pots:
            for ene in pot:
                if pot[ene].requires_grad:
                    enesum = enesum + pot[ene]
        forces[:] = -torch.autograd.grad(enesum, pos, only_inputs=True, retain_graph=True)[0]
    return [{k: v.item() for k, v in pot.items()} for pot in pots], forces, npairs


def _torsion_block(spos, sbox, tab, pot, name, F):
    idx = tab["idx"]
    _, _, r12 = pair_geometry(spos, idx[:, [0, 1]], sbox)
    _, _, r23 = pair_geometry(spos, idx[:, [1, 2]], sbox)
    _, _, r34 = pair_geometry(spos, idx[:, [2, 3]], sbox)
    E, ff = torsions(r12, r23, r34, tab["map"][:, 0], tab["params"][tab["map"][:, 1]])
    pot[name] = pot[name] + E.sum()
    for c in range(4 if F is not None else 0):
        F.index_add_(0, idx[:, c], ff[c])


# ----------------------------------------------------------------------------- integrator
def first_vv(pos, vel, force, mass, dt):
    """integrator.py:61-64 (in place)."""
    accel = force / mass
    pos += vel * dt + 0.5 * accel * dt * dt
    vel += 0.5 * dt * accel


def second_vv(vel, force, mass, dt):
    """integrator.py:67-69 (in place)."""
    accel = force / mass
    vel += 0.5 * dt * accel


def langevin(vel, gamma, coeff, dt, noise):
    """integrator.py:72-74 with the N(0,1) draw passed in (in place)."""
    vel += -gamma * vel * dt + noise * coeff


def kinetic_energy(masses, vel):
    """integrator.py:8-31 (batch=None): [R,1]."""
    return torch.sum(0.5 * masses * torch.sum(vel * vel, dim=2, keepdim=True), dim=1)


def kinetic_to_temp(ekin, natoms):
    """integrator.py:57-58."""
    return 2.0 / (3.0 * natoms * BOLTZMAN) * ekin


def integrator_constants(timestep_fs, gamma_ps, T, masses):
    """integrator.py:84-104 -> (dt, gamma, vcoeff)."""
    dt = timestep_fs / TIMEFACTOR
    gamma = None if gamma_ps is None else gamma_ps / PICOSEC2TIMEU
    vcoeff = None
    if T:
        vcoeff = torch.sqrt(2.0 * gamma / masses * BOLTZMAN * T * dt)
    return dt, gamma, vcoeff


def md_step(par, pos, vel, forces, box, masses, dt, terms, gamma=None, vcoeff=None, noise=None, **kw):
    """One iteration of `Integrator.step` (integrator.py:115-120); returns per-term energies."""
    first_vv(pos, vel, forces, masses, dt)
    pots, f, npairs = compute(par, pos, box, terms, **kw)
    forces.copy_(f)
    if vcoeff is not None:
        langevin(vel, gamma, vcoeff, dt, noise if noise is not None else torch.randn_like(vel))
    second_vv(vel, forces, masses, dt)
    return pots, npairs


# ----------------------------------------------------------------------------- wrapper
def wrap_molecules(pos, box, groups, nongrouped):
    """wrapper.py:8-30 with wrapidx=None (in place): `groups` = list of index tensors, `nongrouped` =
    index tensor of atoms without bonds; pos [R,N,3], box [R,3,3]."""
    b = box[:, torch.eye(3).bool()]
    if torch.all(b == 0):
        return
    for group in groups:
        com = torch.sum(pos[:, group], dim=1) / len(group)
        offset = torch.floor(com / b) * b
        pos[:, group] -= offset.unsqueeze(1)
    if len(nongrouped):
        offset = torch.floor(pos[:, nongrouped] / b.unsqueeze(1)) * b.unsqueeze(1)
        pos[:, nongrouped] -= offset
