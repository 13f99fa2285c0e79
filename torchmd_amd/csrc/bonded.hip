// Bonded terms of Forces.compute on gfx950: harmonic bonds and angles, AMBER / CHARMM torsions
// (dihedrals and impropers) and scaled 1-4 pairs.
//
// Reference semantics: torchmd/forces.py:122-258 (term blocks) and 494-605 (evaluate_bonds,
// evaluate_angles, evaluate_torsion).  These are O(N) gather/scatter kernels: one thread per
// instance, forces combined with hardware float atomics, energies reduced per wave and added with
// one double atomic per wave.
#include <hip/hip_runtime.h>

#include <cmath>
#include <limits>
#include <vector>

#include "common.h"
#include "pair_math.h"

using namespace tmd;

namespace {

struct DevArr {
  void *p = nullptr;
  template <typename T>
  int upload(const T *host, size_t count) {
    release();
    if (count == 0) return 0;
    TMD_HIP(hipMalloc(&p, sizeof(T) * count));
    TMD_HIP(hipMemcpy(p, host, sizeof(T) * count, hipMemcpyHostToDevice));
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
  }
  template <typename T>
  T *as() const {
    return reinterpret_cast<T *>(p);
  }
};

struct Bonded {
  int nbonds = 0, nangles = 0, ndih = 0, nimp = 0, n14 = 0;
  int dih_amber = 1, imp_amber = 1;
  uint32_t terms14 = 0;
  int bonds_use_cutoff = 0;
  DevArr bond_idx, bond_prm, angle_idx, angle_prm;
  DevArr dih_idx, dih_start, dih_prm, imp_idx, imp_start, imp_prm;
  DevArr p14_idx, p14_prm;
  void release() {
    for (DevArr *a : {&bond_idx, &bond_prm, &angle_idx, &angle_prm, &dih_idx, &dih_start, &dih_prm, &imp_idx,
                      &imp_start, &imp_prm, &p14_idx, &p14_prm})
      a->release();
  }
};

template <typename R>
struct Box3 {
  R box[3], invbox[3];
};

template <typename R>
__device__ __forceinline__ void wrapped_delta(const R *__restrict__ pos, int i, int j, const Box3<R> &b, R &dx,
                                              R &dy, R &dz) {
  dx = min_image(pos[3 * i + 0] - pos[3 * j + 0], b.box[0], b.invbox[0]);
  dy = min_image(pos[3 * i + 1] - pos[3 * j + 1], b.box[1], b.invbox[1]);
  dz = min_image(pos[3 * i + 2] - pos[3 * j + 2], b.box[2], b.invbox[2]);
}

template <typename R>
__device__ __forceinline__ void add3(R *__restrict__ f, int i, R x, R y, R z) {
  unsafeAtomicAdd(&f[3 * i + 0], x);
  unsafeAtomicAdd(&f[3 * i + 1], y);
  unsafeAtomicAdd(&f[3 * i + 2], z);
}

__device__ __forceinline__ void wave_energy(double e, double *dst) {
  const double s = wave_sum(e);
  if ((threadIdx.x & 63) == 0 && s != 0.0) unsafeAtomicAdd(dst, s);
}

__device__ __forceinline__ float dsqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double dsqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float dacos(float x) { return acosf(x); }
__device__ __forceinline__ double dacos(double x) { return acos(x); }
__device__ __forceinline__ float datan2(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ double datan2(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ void dsincos(float a, float *s, float *c) { sincosf(a, s, c); }
__device__ __forceinline__ void dsincos(double a, double *s, double *c) { sincos(a, s, c); }

// forces.py:122-143 + evaluate_bonds 494-503.  r2max: the reference drops bonds with dist > cutoff
// when a cutoff is set (same decision arithmetic as the nonbonded filter).
template <typename R>
__global__ void bonds_kernel(int n, const int *__restrict__ idx, const R *__restrict__ prm,
                             const R *__restrict__ pos, Box3<R> b, R r2max, R *__restrict__ forces,
                             double *__restrict__ energy, int want_e) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double e = 0;
  if (t < n) {
    const int i = idx[2 * t], j = idx[2 * t + 1];
    R dx, dy, dz;
    wrapped_delta(pos, i, j, b, dx, dy, dz);
    const R r2 = norm2(dx, dy, dz);
    if (r2 <= r2max) {
      const R r = dsqrt(r2);
      const R k0 = prm[2 * t], d0 = prm[2 * t + 1];
      const R x = r - d0;
      e = (double)(k0 * x * x);
      if (forces) {
        const R fs = R(2) * k0 * x / r;  // unitvec * force_coeff
        add3(forces, i, -dx * fs, -dy * fs, -dz * fs);
        add3(forces, j, dx * fs, dy * fs, dz * fs);
      }
    }
  }
  if (want_e) wave_energy(e, energy);
}

// forces.py:145-161 + evaluate_angles 506-539
template <typename R>
__global__ void angles_kernel(int n, const int *__restrict__ idx, const R *__restrict__ prm,
                              const R *__restrict__ pos, Box3<R> b, R *__restrict__ forces,
                              double *__restrict__ energy, int want_e) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double e = 0;
  if (t < n) {
    const int a0 = idx[3 * t], a1 = idx[3 * t + 1], a2 = idx[3 * t + 2];
    R x21, y21, z21, x23, y23, z23;
    wrapped_delta(pos, a0, a1, b, x21, y21, z21);
    wrapped_delta(pos, a2, a1, b, x23, y23, z23);
    const R k0 = prm[2 * t], th0 = prm[2 * t + 1];
    const R dot = x23 * x21 + y23 * y21 + z23 * z21;
    const R n21 = R(1) / dsqrt(x21 * x21 + y21 * y21 + z21 * z21);
    const R n23 = R(1) / dsqrt(x23 * x23 + y23 * y23 + z23 * z23);
    R cs = dot * n21 * n23;
    cs = cs < R(-1) ? R(-1) : (cs > R(1) ? R(1) : cs);
    const R th = dacos(cs);
    const R dth = th - th0;
    e = (double)(k0 * dth * dth);
    if (forces) {
      const R sn = dsqrt(R(1) - cs * cs);
      const R coef = sn != R(0) ? R(-2) * k0 * dth / sn : R(0);
      const R f0x = coef * (cs * x21 * n21 - x23 * n23) * n21;
      const R f0y = coef * (cs * y21 * n21 - y23 * n23) * n21;
      const R f0z = coef * (cs * z21 * n21 - z23 * n23) * n21;
      const R f2x = coef * (cs * x23 * n23 - x21 * n21) * n23;
      const R f2y = coef * (cs * y23 * n23 - y21 * n21) * n23;
      const R f2z = coef * (cs * z23 * n23 - z21 * n21) * n23;
      add3(forces, a0, f0x, f0y, f0z);
      add3(forces, a2, f2x, f2y, f2z);
      add3(forces, a1, -(f0x + f2x), -(f0y + f2y), -(f0z + f2z));
    }
  }
  if (want_e) wave_energy(e, energy);
}

// forces.py:163-183 / 238-258 + evaluate_torsion 542-605.  One thread per torsion; its terms are
// rows [start[t], start[t+1]) of prm = (k0, phi0, per).  `amber` mirrors `torch.all(per > 0)`.
template <typename R>
__global__ void torsions_kernel(int n, const int *__restrict__ idx, const int *__restrict__ start,
                                const R *__restrict__ prm, int amber, const R *__restrict__ pos, Box3<R> b,
                                R *__restrict__ forces, double *__restrict__ energy, int want_e) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double e = 0;
  if (t < n) {
    const int i0 = idx[4 * t], i1 = idx[4 * t + 1], i2 = idx[4 * t + 2], i3 = idx[4 * t + 3];
    R ax, ay, az, bx, by, bz, cx, cy, cz;  // r12, r23, r34
    wrapped_delta(pos, i0, i1, b, ax, ay, az);
    wrapped_delta(pos, i1, i2, b, bx, by, bz);
    wrapped_delta(pos, i2, i3, b, cx, cy, cz);
    // crossA = r12 x r23, crossB = r23 x r34, crossC = r23 x crossA
    const R Ax = ay * bz - az * by, Ay = az * bx - ax * bz, Az = ax * by - ay * bx;
    const R Bx = by * cz - bz * cy, By = bz * cx - bx * cz, Bz = bx * cy - by * cx;
    const R Cx = by * Az - bz * Ay, Cy = bz * Ax - bx * Az, Cz = bx * Ay - by * Ax;
    const R nA2 = Ax * Ax + Ay * Ay + Az * Az, nB2 = Bx * Bx + By * By + Bz * Bz;
    const R nA = dsqrt(nA2), nB = dsqrt(nB2), nC = dsqrt(Cx * Cx + Cy * Cy + Cz * Cz);
    const R ux = Bx / nB, uy = By / nB, uz = Bz / nB;
    const R cosphi = (Ax * ux + Ay * uy + Az * uz) / nA;
    const R sinphi = (Cx * ux + Cy * uy + Cz * uz) / nC;
    const R phi = -datan2(sinphi, cosphi);
    R pot = 0, coeff = 0;
    const R PI = R(3.14159265358979323846);
    for (int m = start[t]; m < start[t + 1]; ++m) {
      const R k0 = prm[3 * m], phi0 = prm[3 * m + 1], per = prm[3 * m + 2];
      if (amber) {
        R s, c;
        dsincos(per * phi - phi0, &s, &c);
        pot += k0 * (R(1) + c);
        coeff += -per * k0 * s;
      } else {
        R ad = phi - phi0;
        if (ad < -PI) ad += R(2) * PI;
        else if (ad > PI) ad -= R(2) * PI;
        pot += k0 * ad * ad;
        coeff += R(2) * k0 * ad;
      }
    }
    e = (double)pot;
    if (forces) {
      const R n23sq = bx * bx + by * by + bz * bz;
      const R n23 = dsqrt(n23sq);
      const R ff0 = (-coeff * n23) / nA2;
      const R ff1 = (ax * bx + ay * by + az * bz) / n23sq;
      const R ff2 = (cx * bx + cy * by + cz * bz) / n23sq;
      const R ff3 = (coeff * n23) / nB2;
      const R f0x = ff0 * Ax, f0y = ff0 * Ay, f0z = ff0 * Az;
      const R f3x = ff3 * Bx, f3y = ff3 * By, f3z = ff3 * Bz;
      const R sx = ff1 * f0x - ff2 * f3x, sy = ff1 * f0y - ff2 * f3y, sz = ff1 * f0z - ff2 * f3z;
      add3(forces, i0, -f0x, -f0y, -f0z);
      add3(forces, i1, f0x + sx, f0y + sy, f0z + sz);
      add3(forces, i2, f3x - sx, f3y - sy, f3z - sz);
      add3(forces, i3, -f3x, -f3y, -f3z);
    }
  }
  if (want_e) wave_energy(e, energy);
}

// forces.py:185-236: scaled 1-4 LJ (evaluate_LJ_internal with scale=scnb, no switch) and plain
// Coulomb with scale=scee, no cutoff.  prm = (A, B, scnb, scee); qs = q*sqrt(k_e).
template <typename R>
__global__ void pairs14_kernel(int n, const int *__restrict__ idx, const R *__restrict__ prm,
                               const R *__restrict__ qs, uint32_t terms, const R *__restrict__ pos, Box3<R> b,
                               R *__restrict__ forces, double *__restrict__ e_lj, double *__restrict__ e_el,
                               int want_e) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double elj = 0, eel = 0;
  if (t < n) {
    const int i = idx[2 * t], j = idx[2 * t + 1];
    R dx, dy, dz;
    wrapped_delta(pos, i, j, b, dx, dy, dz);
    const R r2 = norm2(dx, dy, dz);
    const R rinv = R(1) / dsqrt(r2);
    const R rinv2 = rinv * rinv, rinv6 = rinv2 * rinv2 * rinv2;
    const R A = prm[4 * t], B = prm[4 * t + 1], scnb = prm[4 * t + 2], scee = prm[4 * t + 3];
    R dEdr = 0;
    if (terms & TMDHIP_TERM_LJ) {
      elj = (double)((A * rinv6 - B) * rinv6 / scnb);
      dEdr += (R(-12) * A * rinv6 + R(6) * B) * rinv6 * rinv / scnb;
    }
    if (terms & TMDHIP_TERM_ELECTROSTATICS) {
      const R ee = qs[i] * qs[j] * rinv / scee;
      eel = (double)ee;
      dEdr -= ee * rinv;
    }
    if (forces) {
      const R fs = dEdr * rinv;
      add3(forces, i, -dx * fs, -dy * fs, -dz * fs);
      add3(forces, j, dx * fs, dy * fs, dz * fs);
    }
  }
  if (want_e) {
    wave_energy(elj, e_lj);
    wave_energy(eel, e_el);
  }
}

template <typename R>
int upload_terms(const int32_t *term_of, const void *prm, int nterms, int ntors, DevArr &start, DevArr &dprm,
                 int &amber) {
  std::vector<int> st(ntors + 1, 0);
  for (int m = 0; m < nterms; ++m) {
    if (term_of[m] < 0 || term_of[m] >= ntors) return fail("tmdhip_set_bonded: torsion term index out of range");
    if (m && term_of[m] < term_of[m - 1]) return fail("tmdhip_set_bonded: torsion terms must be grouped in ascending order");
    st[term_of[m] + 1]++;
  }
  for (int t = 0; t < ntors; ++t) st[t + 1] += st[t];
  TMD_TRY(start.upload(st.data(), st.size()));
  const R *p = (const R *)prm;
  amber = 1;
  for (int m = 0; m < nterms; ++m)
    if (!(p[3 * m + 2] > 0)) amber = 0;
  TMD_TRY(dprm.upload(p, (size_t)3 * nterms));
  return 0;
}

}  // namespace

namespace tmd {
// accessors implemented in nonbonded.hip (the ctx layout is private to that file)
void *&ctx_bonded_slot(tmdhip_ctx *ctx);
const tmdhip_nonbonded_desc &ctx_desc(const tmdhip_ctx *ctx);
const void *ctx_scaled_charges(const tmdhip_ctx *ctx);
int ctx_nreplicas(const tmdhip_ctx *ctx);

void bonded_release(tmdhip_ctx *ctx) {
  Bonded *b = (Bonded *)ctx_bonded_slot(ctx);
  if (b) {
    b->release();
    delete b;
  }
  ctx_bonded_slot(ctx) = nullptr;
}
}  // namespace tmd

namespace {

template <typename R>
int set_bonded(tmdhip_ctx *ctx, Bonded *b, const tmdhip_bonded_desc *d) {
  const int n = ctx_desc(ctx).natoms;
  auto check = [&](const int32_t *idx, int count, int width) {
    for (int k = 0; k < count * width; ++k)
      if (idx[k] < 0 || idx[k] >= n) return false;
    return true;
  };
  if (d->nbonds) {
    if (!check(d->bond_idx_host, d->nbonds, 2)) return fail("tmdhip_set_bonded: bond index out of range");
    TMD_TRY(b->bond_idx.upload(d->bond_idx_host, (size_t)2 * d->nbonds));
    TMD_TRY(b->bond_prm.upload((const R *)d->bond_prm_host, (size_t)2 * d->nbonds));
  }
  if (d->nangles) {
    if (!check(d->angle_idx_host, d->nangles, 3)) return fail("tmdhip_set_bonded: angle index out of range");
    TMD_TRY(b->angle_idx.upload(d->angle_idx_host, (size_t)3 * d->nangles));
    TMD_TRY(b->angle_prm.upload((const R *)d->angle_prm_host, (size_t)2 * d->nangles));
  }
  if (d->ndihedrals) {
    if (!check(d->dihedral_idx_host, d->ndihedrals, 4)) return fail("tmdhip_set_bonded: dihedral index out of range");
    TMD_TRY(b->dih_idx.upload(d->dihedral_idx_host, (size_t)4 * d->ndihedrals));
    TMD_TRY((upload_terms<R>(d->dihedral_term_of_host, d->dihedral_prm_host, d->ndihedral_terms, d->ndihedrals,
                             b->dih_start, b->dih_prm, b->dih_amber)));
  }
  if (d->nimpropers) {
    if (!check(d->improper_idx_host, d->nimpropers, 4)) return fail("tmdhip_set_bonded: improper index out of range");
    TMD_TRY(b->imp_idx.upload(d->improper_idx_host, (size_t)4 * d->nimpropers));
    TMD_TRY((upload_terms<R>(d->improper_term_of_host, d->improper_prm_host, d->nimproper_terms, d->nimpropers,
                             b->imp_start, b->imp_prm, b->imp_amber)));
  }
  if (d->n14) {
    if (!check(d->pair14_idx_host, d->n14, 2)) return fail("tmdhip_set_bonded: 1-4 index out of range");
    TMD_TRY(b->p14_idx.upload(d->pair14_idx_host, (size_t)2 * d->n14));
    TMD_TRY(b->p14_prm.upload((const R *)d->pair14_prm_host, (size_t)4 * d->n14));
  }
  b->nbonds = d->nbonds;
  b->nangles = d->nangles;
  b->ndih = d->ndihedrals;
  b->nimp = d->nimpropers;
  b->n14 = d->n14;
  b->terms14 = d->terms14;
  b->bonds_use_cutoff = d->bonds_use_cutoff;
  return 0;
}

template <typename R>
R host_r2max(double cutoff) {
  if (!(cutoff > 0)) return std::numeric_limits<R>::infinity();
  const R c = (R)cutoff;
  R r2 = c * c;
  while (std::sqrt(r2) <= c) r2 = std::nextafter(r2, std::numeric_limits<R>::infinity());
  while (std::sqrt(r2) > c) r2 = std::nextafter(r2, (R)0);
  return r2;
}

template <typename R>
int run_bonded(tmdhip_ctx *ctx, const Bonded *b, const void *pos_v, const double *box, void *forces_v,
               double *en, int flags, hipStream_t st) {
  const R *pos = (const R *)pos_v;
  R *forces = (flags & TMDHIP_WANT_FORCES) ? (R *)forces_v : nullptr;
  const int we = (flags & TMDHIP_WANT_ENERGY) ? 1 : 0;
  Box3<R> bx;
  const bool allzero = box[0] == 0 && box[1] == 0 && box[2] == 0;
  for (int k = 0; k < 3; ++k) {
    bx.box[k] = (R)box[k];
    bx.invbox[k] = (!allzero && bx.box[k] != R(0)) ? R(1) / bx.box[k] : R(0);
  }
  const int T = 128;
  if (b->nbonds) {
    const R r2max = b->bonds_use_cutoff ? host_r2max<R>(ctx_desc(ctx).cutoff) : std::numeric_limits<R>::infinity();
    hipLaunchKernelGGL((bonds_kernel<R>), dim3((b->nbonds + T - 1) / T), dim3(T), 0, st, b->nbonds,
                       b->bond_idx.as<int>(), b->bond_prm.as<R>(), pos, bx, r2max, forces, en + TMDHIP_E_BONDS, we);
  }
  if (b->nangles)
    hipLaunchKernelGGL((angles_kernel<R>), dim3((b->nangles + T - 1) / T), dim3(T), 0, st, b->nangles,
                       b->angle_idx.as<int>(), b->angle_prm.as<R>(), pos, bx, forces, en + TMDHIP_E_ANGLES, we);
  if (b->ndih)
    hipLaunchKernelGGL((torsions_kernel<R>), dim3((b->ndih + T - 1) / T), dim3(T), 0, st, b->ndih,
                       b->dih_idx.as<int>(), b->dih_start.as<int>(), b->dih_prm.as<R>(), b->dih_amber, pos, bx,
                       forces, en + TMDHIP_E_DIHEDRALS, we);
  if (b->n14 && b->terms14)
    hipLaunchKernelGGL((pairs14_kernel<R>), dim3((b->n14 + T - 1) / T), dim3(T), 0, st, b->n14,
                       b->p14_idx.as<int>(), b->p14_prm.as<R>(), (const R *)ctx_scaled_charges(ctx), b->terms14, pos,
                       bx, forces, en + TMDHIP_E_LJ, en + TMDHIP_E_ELECTROSTATICS, we);
  if (b->nimp)
    hipLaunchKernelGGL((torsions_kernel<R>), dim3((b->nimp + T - 1) / T), dim3(T), 0, st, b->nimp,
                       b->imp_idx.as<int>(), b->imp_start.as<int>(), b->imp_prm.as<R>(), b->imp_amber, pos, bx,
                       forces, en + TMDHIP_E_IMPROPERS, we);
  TMD_HIP(hipGetLastError());
  return 0;
}

}  // namespace

extern "C" {

int tmdhip_set_bonded(tmdhip_ctx *ctx, const tmdhip_bonded_desc *desc) {
  if (!ctx || !desc) return fail("tmdhip_set_bonded: null argument");
  if (desc->struct_size != (int32_t)sizeof(tmdhip_bonded_desc))
    return fail("tmdhip_set_bonded: tmdhip_bonded_desc size mismatch (ABI)");
  tmd::bonded_release(ctx);
  Bonded *b = new Bonded();
  const int rc = ctx_desc(ctx).dtype == TMDHIP_F32 ? set_bonded<float>(ctx, b, desc) : set_bonded<double>(ctx, b, desc);
  if (rc) {
    b->release();
    delete b;
    return rc;
  }
  ctx_bonded_slot(ctx) = b;
  return 0;
}

int tmdhip_compute_bonded(tmdhip_ctx *ctx, int replica, const void *pos_dev, const double *box_host,
                          void *forces_dev, double *energies_dev, int flags, void *stream) {
  if (!ctx || !pos_dev || !box_host) return fail("tmdhip_compute_bonded: null argument");
  if (replica < 0 || replica >= ctx_nreplicas(ctx)) return fail("tmdhip_compute_bonded: bad replica index");
  if ((flags & TMDHIP_WANT_FORCES) && !forces_dev) return fail("tmdhip_compute_bonded: forces requested without a buffer");
  if ((flags & TMDHIP_WANT_ENERGY) && !energies_dev) return fail("tmdhip_compute_bonded: energies requested without a buffer");
  const Bonded *b = (const Bonded *)ctx_bonded_slot(ctx);
  if (!b) return 0;
  hipStream_t st = (hipStream_t)stream;
  return ctx_desc(ctx).dtype == TMDHIP_F32
             ? run_bonded<float>(ctx, b, pos_dev, box_host, forces_dev, energies_dev, flags, st)
             : run_bonded<double>(ctx, b, pos_dev, box_host, forces_dev, energies_dev, flags, st);
}

}  // extern "C"
