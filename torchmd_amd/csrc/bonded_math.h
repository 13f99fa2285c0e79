// Device arithmetic of the bonded terms (forces.py:122-258, 494-605), shared by the bonded kernels
// (bonded.hip) and by the MD-step kernels of md_loop.hip / pair_fast_f32.hip, which evaluates an atom's bonded force inline
// for light topologies.  Everything here is per-translation-unit (anonymous namespace, forceinline).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "pair_math.h"

namespace tmd {

template <typename R>
struct Box3 {
  R box[3], invbox[3];
};

// Per-atom record of the atom-centric kernels (light topologies): the entry word plus, for bonds and
// angles, the partner atoms and the two parameters inline, so that evaluating an atom's terms is
// "read my records -> read the partners' positions" instead of four dependent loads
// (atom_off -> atom_ent -> idx/prm -> pos).  a/b: bond = (partner, -); angle = the two other atoms in
// idx-column order.  Torsions and 1-4 pairs keep the table lookup through the term index in `ent`.
template <typename R>
struct AtomRec {
  unsigned ent;  // kind << 28 | role << 26 | term index; kNoRec = unused slot
  int a, b;
  R p0, p1;
};
constexpr unsigned kNoRec = 0xFFFFFFFFu;

template <typename R>
struct BondedArgs {
  const AtomRec<R> *arec;  // [natoms][arec_stride], null for heavy topologies
  int arec_stride;
  const int *atom_off, *atom_ent;
  const int *bond_idx;
  const R *bond_prm;
  const int *angle_idx;
  const R *angle_prm;
  const int *dih_idx, *dih_start;
  const R *dih_prm;
  const int *imp_idx, *imp_start;
  const R *imp_prm;
  const int *p14_idx;
  const R *p14_prm;
  const R *qs;
  int dih_amber, imp_amber;
  uint32_t terms14;
  R bond_r2max;
  Box3<R> b;
};

namespace {

enum Kind : unsigned { KBOND = 0, KANGLE = 1, KDIHEDRAL = 2, KIMPROPER = 3, KPAIR14 = 4 };
// entry = kind << 28 | role << 26 | term index (26 bits)
constexpr unsigned kIdxBits = 26;

template <typename R>
struct V3 {
  R x, y, z;
};

template <typename R>
__device__ __forceinline__ V3<R> wrapped_delta(const R *__restrict__ pos, int i, int j, const Box3<R> &b) {
  V3<R> d;
  d.x = min_image(pos[3 * i + 0] - pos[3 * j + 0], b.box[0], b.invbox[0]);
  d.y = min_image(pos[3 * i + 1] - pos[3 * j + 1], b.box[1], b.invbox[1]);
  d.z = min_image(pos[3 * i + 2] - pos[3 * j + 2], b.box[2], b.invbox[2]);
  return d;
}

__device__ __forceinline__ float dsqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double dsqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float dacos(float x) { return acosf(x); }
__device__ __forceinline__ double dacos(double x) { return acos(x); }
__device__ __forceinline__ float datan2(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ double datan2(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ void dsincos(float a, float *s, float *c) { sincosf(a, s, c); }
__device__ __forceinline__ void dsincos(double a, double *s, double *c) { sincos(a, s, c); }

// forces.py:122-143 + evaluate_bonds 494-503.  The reference drops bonds with dist > cutoff when a
// cutoff is set (same decision arithmetic as the nonbonded filter).  role 0 = first atom.
template <typename R>
__device__ __forceinline__ void bond_core(const BondedArgs<R> &A, const R *__restrict__ pos, int i, int j, R k0,
                                          R d0, int role, R &fx, R &fy, R &fz, double &e) {
#pragma clang fp contract(off)  // same rounding whether or not the energy is live (inline use)
  const V3<R> d = wrapped_delta(pos, i, j, A.b);
  const R r2 = norm2(d.x, d.y, d.z);
  if (!(r2 <= A.bond_r2max)) return;
  const R r = dsqrt(r2);
  const R x = r - d0;
  if (role == 0) e += (double)(k0 * x * x);
  const R fs = R(2) * k0 * x / r;  // unitvec * force_coeff ; F_i -= , F_j +=
  const R sgn = role == 0 ? R(-1) : R(1);
  fx += sgn * d.x * fs;
  fy += sgn * d.y * fs;
  fz += sgn * d.z * fs;
}

template <typename R>
__device__ __forceinline__ void bond_term(const BondedArgs<R> &A, const R *__restrict__ pos, int t, int role,
                                          R &fx, R &fy, R &fz, double &e) {
  bond_core<R>(A, pos, A.bond_idx[2 * t], A.bond_idx[2 * t + 1], A.bond_prm[2 * t], A.bond_prm[2 * t + 1], role, fx,
               fy, fz, e);
}

// forces.py:145-161 + evaluate_angles 506-539.  roles 0,1,2 = idx columns (1 = vertex)
template <typename R>
__device__ __forceinline__ void angle_core(const BondedArgs<R> &A, const R *__restrict__ pos, int a0, int a1,
                                           int a2, R k0, R th0, int role, R &fx, R &fy, R &fz, double &e) {
#pragma clang fp contract(off)  // same rounding whether or not the energy is live (inline use)
  const V3<R> r21 = wrapped_delta(pos, a0, a1, A.b);
  const V3<R> r23 = wrapped_delta(pos, a2, a1, A.b);
  const R dot = r23.x * r21.x + r23.y * r21.y + r23.z * r21.z;
  const R n21 = R(1) / dsqrt(r21.x * r21.x + r21.y * r21.y + r21.z * r21.z);
  const R n23 = R(1) / dsqrt(r23.x * r23.x + r23.y * r23.y + r23.z * r23.z);
  R cs = dot * n21 * n23;
  cs = cs < R(-1) ? R(-1) : (cs > R(1) ? R(1) : cs);
  const R dth = dacos(cs) - th0;
  if (role == 0) e += (double)(k0 * dth * dth);
  const R sn = dsqrt(R(1) - cs * cs);
  const R coef = sn != R(0) ? R(-2) * k0 * dth / sn : R(0);
  const R f0x = coef * (cs * r21.x * n21 - r23.x * n23) * n21;
  const R f0y = coef * (cs * r21.y * n21 - r23.y * n23) * n21;
  const R f0z = coef * (cs * r21.z * n21 - r23.z * n23) * n21;
  const R f2x = coef * (cs * r23.x * n23 - r21.x * n21) * n23;
  const R f2y = coef * (cs * r23.y * n23 - r21.y * n21) * n23;
  const R f2z = coef * (cs * r23.z * n23 - r21.z * n21) * n23;
  if (role == 0) {
    fx += f0x, fy += f0y, fz += f0z;
  } else if (role == 2) {
    fx += f2x, fy += f2y, fz += f2z;
  } else {
    fx -= f0x + f2x, fy -= f0y + f2y, fz -= f0z + f2z;
  }
}

template <typename R>
__device__ __forceinline__ void angle_term(const BondedArgs<R> &A, const R *__restrict__ pos, int t, int role,
                                           R &fx, R &fy, R &fz, double &e) {
  angle_core<R>(A, pos, A.angle_idx[3 * t], A.angle_idx[3 * t + 1], A.angle_idx[3 * t + 2], A.angle_prm[2 * t],
                A.angle_prm[2 * t + 1], role, fx, fy, fz, e);
}

// forces.py:163-183 / 238-258 + evaluate_torsion 542-605.  The terms of torsion t are rows
// [start[t], start[t+1]) of prm = (k0, phi0, per); `amber` mirrors `torch.all(per > 0)`.
template <typename R>
__device__ __forceinline__ void torsion_term(const int *__restrict__ idx, const int *__restrict__ start,
                                             const R *__restrict__ prm, int amber, const Box3<R> &b,
                                             const R *__restrict__ pos, int t, int role, R &fx, R &fy, R &fz,
                                             double &e) {
#pragma clang fp contract(off)  // same rounding whether or not the energy is live (inline use)
  const int i0 = idx[4 * t], i1 = idx[4 * t + 1], i2 = idx[4 * t + 2], i3 = idx[4 * t + 3];
  const V3<R> a = wrapped_delta(pos, i0, i1, b);  // r12
  const V3<R> m = wrapped_delta(pos, i1, i2, b);  // r23
  const V3<R> c = wrapped_delta(pos, i2, i3, b);  // r34
  // crossA = r12 x r23, crossB = r23 x r34, crossC = r23 x crossA
  const R Ax = a.y * m.z - a.z * m.y, Ay = a.z * m.x - a.x * m.z, Az = a.x * m.y - a.y * m.x;
  const R Bx = m.y * c.z - m.z * c.y, By = m.z * c.x - m.x * c.z, Bz = m.x * c.y - m.y * c.x;
  const R Cx = m.y * Az - m.z * Ay, Cy = m.z * Ax - m.x * Az, Cz = m.x * Ay - m.y * Ax;
  const R nA2 = Ax * Ax + Ay * Ay + Az * Az, nB2 = Bx * Bx + By * By + Bz * Bz;
  const R nA = dsqrt(nA2), nB = dsqrt(nB2), nC = dsqrt(Cx * Cx + Cy * Cy + Cz * Cz);
  const R ux = Bx / nB, uy = By / nB, uz = Bz / nB;
  const R cosphi = (Ax * ux + Ay * uy + Az * uz) / nA;
  const R sinphi = (Cx * ux + Cy * uy + Cz * uz) / nC;
  const R phi = -datan2(sinphi, cosphi);
  R pot = 0, coeff = 0;
  const R PI = R(3.14159265358979323846);
  for (int q = start[t]; q < start[t + 1]; ++q) {
    const R k0 = prm[3 * q], phi0 = prm[3 * q + 1], per = prm[3 * q + 2];
    if (amber) {
      R s, cc;
      dsincos(per * phi - phi0, &s, &cc);
      pot += k0 * (R(1) + cc);
      coeff += -per * k0 * s;
    } else {
      R ad = phi - phi0;
      if (ad < -PI) ad += R(2) * PI;
      else if (ad > PI) ad -= R(2) * PI;
      pot += k0 * ad * ad;
      coeff += R(2) * k0 * ad;
    }
  }
  if (role == 0) e += (double)pot;
  const R n23sq = m.x * m.x + m.y * m.y + m.z * m.z;
  const R n23 = dsqrt(n23sq);
  const R ff0 = (-coeff * n23) / nA2;
  const R ff1 = (a.x * m.x + a.y * m.y + a.z * m.z) / n23sq;
  const R ff2 = (c.x * m.x + c.y * m.y + c.z * m.z) / n23sq;
  const R ff3 = (coeff * n23) / nB2;
  const R f0x = ff0 * Ax, f0y = ff0 * Ay, f0z = ff0 * Az;
  const R f3x = ff3 * Bx, f3y = ff3 * By, f3z = ff3 * Bz;
  const R sx = ff1 * f0x - ff2 * f3x, sy = ff1 * f0y - ff2 * f3y, sz = ff1 * f0z - ff2 * f3z;
  if (role == 0) {
    fx -= f0x, fy -= f0y, fz -= f0z;
  } else if (role == 1) {
    fx += f0x + sx, fy += f0y + sy, fz += f0z + sz;
  } else if (role == 2) {
    fx += f3x - sx, fy += f3y - sy, fz += f3z - sz;
  } else {
    fx -= f3x, fy -= f3y, fz -= f3z;
  }
}

// forces.py:185-236: scaled 1-4 LJ (evaluate_LJ_internal with scale=scnb, no switch) and plain Coulomb
// with scale=scee, no cutoff.  prm = (A, B, scnb, scee); qs = q*sqrt(k_e).
template <typename R>
__device__ __forceinline__ void pair14_term(const BondedArgs<R> &A, const R *__restrict__ pos, int t, int role,
                                            R &fx, R &fy, R &fz, double &elj, double &eel) {
#pragma clang fp contract(off)  // same rounding whether or not the energy is live (inline use)
  const int i = A.p14_idx[2 * t], j = A.p14_idx[2 * t + 1];
  const V3<R> d = wrapped_delta(pos, i, j, A.b);
  const R r2 = norm2(d.x, d.y, d.z);
  const R rinv = R(1) / dsqrt(r2);
  const R rinv2 = rinv * rinv, rinv6 = rinv2 * rinv2 * rinv2;
  const R a = A.p14_prm[4 * t], bb = A.p14_prm[4 * t + 1], scnb = A.p14_prm[4 * t + 2], scee = A.p14_prm[4 * t + 3];
  R dEdr = 0;
  if (A.terms14 & TMDHIP_TERM_LJ) {
    if (role == 0) elj += (double)((a * rinv6 - bb) * rinv6 / scnb);
    dEdr += (R(-12) * a * rinv6 + R(6) * bb) * rinv6 * rinv / scnb;
  }
  if (A.terms14 & TMDHIP_TERM_ELECTROSTATICS) {
    const R ee = A.qs[i] * A.qs[j] * rinv / scee;
    if (role == 0) eel += (double)ee;
    dEdr -= ee * rinv;
  }
  const R fs = dEdr * rinv;
  const R sgn = role == 0 ? R(-1) : R(1);
  fx += sgn * d.x * fs;
  fy += sgn * d.y * fs;
  fz += sgn * d.z * fs;
}

// force on the atom that entry `ent` stands for (and the term's energy if that atom has role 0)
template <typename R>
__device__ __forceinline__ void eval_entry(const BondedArgs<R> &A, const R *__restrict__ pos, unsigned ent, R &fx,
                                           R &fy, R &fz, double *e) {
  const unsigned kind = ent >> 28;
  const int role = (int)((ent >> kIdxBits) & 3u);
  const int t = (int)(ent & ((1u << kIdxBits) - 1u));
  if (kind == KBOND) {
    bond_term<R>(A, pos, t, role, fx, fy, fz, e[TMDHIP_E_BONDS]);
  } else if (kind == KANGLE) {
    angle_term<R>(A, pos, t, role, fx, fy, fz, e[TMDHIP_E_ANGLES]);
  } else if (kind == KDIHEDRAL) {
    torsion_term<R>(A.dih_idx, A.dih_start, A.dih_prm, A.dih_amber, A.b, pos, t, role, fx, fy, fz,
                    e[TMDHIP_E_DIHEDRALS]);
  } else if (kind == KIMPROPER) {
    torsion_term<R>(A.imp_idx, A.imp_start, A.imp_prm, A.imp_amber, A.b, pos, t, role, fx, fy, fz,
                    e[TMDHIP_E_IMPROPERS]);
  } else {
    pair14_term<R>(A, pos, t, role, fx, fy, fz, e[TMDHIP_E_LJ], e[TMDHIP_E_ELECTROSTATICS]);
  }
}

// the same, from the per-atom record of atom `self` (bit-identical: same core functions, same order)
template <typename R>
__device__ __forceinline__ void eval_rec(const BondedArgs<R> &A, const R *__restrict__ pos, int self,
                                         const AtomRec<R> &rec, R &fx, R &fy, R &fz, double *e) {
  const unsigned kind = rec.ent >> 28;
  const int role = (int)((rec.ent >> kIdxBits) & 3u);
  if (kind == KBOND) {
    bond_core<R>(A, pos, role == 0 ? self : rec.a, role == 0 ? rec.a : self, rec.p0, rec.p1, role, fx, fy, fz,
                 e[TMDHIP_E_BONDS]);
  } else if (kind == KANGLE) {
    const int a0 = role == 0 ? self : rec.a;
    const int a1 = role == 1 ? self : (role == 0 ? rec.a : rec.b);
    const int a2 = role == 2 ? self : rec.b;
    angle_core<R>(A, pos, a0, a1, a2, rec.p0, rec.p1, role, fx, fy, fz, e[TMDHIP_E_ANGLES]);
  } else {
    eval_entry<R>(A, pos, rec.ent, fx, fy, fz, e);
  }
}

// All bonded terms of atom `self` (atom-centric scheme), evaluated by the FOUR adjacent lanes of the atom:
// lane `sub` takes records sub, sub + 4, ... (a record = load, then dependent partner-position loads: one
// thread walking an atom's records is a chain of 2 x records memory round trips — 9.9 us for the kernel at C3 —
// four lanes make it 2), then the three force components are added with a fixed butterfly:
// (r0 + r1) + (r2 + r3).  Must be called by all four lanes (active = the atom exists); every lane returns the sum.
constexpr int kQuad = 4;
template <typename R>
__device__ __forceinline__ void eval_atom_quad(const BondedArgs<R> &A, const R *__restrict__ pos, int self, int sub,
                                               bool active, R &fx, R &fy, R &fz, double *e) {
  if (active) {
    const AtomRec<R> *rec = A.arec + (size_t)self * A.arec_stride;
    for (int k = sub; k < A.arec_stride; k += kQuad) {
      const AtomRec<R> r = rec[k];
      if (r.ent == kNoRec) break;  // records are packed from the front
      eval_rec<R>(A, pos, self, r, fx, fy, fz, e);
    }
  }
#pragma unroll
  for (int o = 1; o < kQuad; o <<= 1) {
    fx += __shfl_xor(fx, o, 64);
    fy += __shfl_xor(fy, o, 64);
    fz += __shfl_xor(fz, o, 64);
  }
}

// per-term energies of a wave go to the wave's scratch row (pair_math.h: energy_row)
__device__ __forceinline__ void wave_energy(double e, double *dst) {
  const double s = wave_sum(e);
  if ((threadIdx.x & 63) == 0 && s != 0.0) unsafeAtomicAdd(dst, s);  // dst: this wave's scratch row
}


__device__ __forceinline__ void flush_energies(const double *e, double *scratch) {
  double *energies = energy_row(scratch);
  wave_energy(e[TMDHIP_E_BONDS], energies + TMDHIP_E_BONDS);
  wave_energy(e[TMDHIP_E_ANGLES], energies + TMDHIP_E_ANGLES);
  wave_energy(e[TMDHIP_E_DIHEDRALS], energies + TMDHIP_E_DIHEDRALS);
  wave_energy(e[TMDHIP_E_IMPROPERS], energies + TMDHIP_E_IMPROPERS);
  wave_energy(e[TMDHIP_E_LJ], energies + TMDHIP_E_LJ);
  wave_energy(e[TMDHIP_E_ELECTROSTATICS], energies + TMDHIP_E_ELECTROSTATICS);
}

}  // namespace
}  // namespace tmd
