// Bonded terms of Forces.compute on gfx950: harmonic bonds and angles, AMBER / CHARMM torsions
// (dihedrals and impropers) and scaled 1-4 pairs — ONE atom-centric kernel.
//
// Reference semantics: torchmd/forces.py:122-258 (term blocks) and 494-605 (evaluate_bonds,
// evaluate_angles, evaluate_torsion).  The reference evaluates each term once and scatters its forces
// with index_add_.  On the GPU a term-parallel kernel needs ~6-12 float atomics per term (measured:
// 29 us per step for the 98 304 bonds + 32 768 angles of the water box, atomics-bound).  Here the
// topology is inverted once on the host into a per-atom list of (term, role) entries and each thread
// owns ONE atom: it re-evaluates the few terms that atom takes part in (a bond is evaluated twice, an
// angle three times, a torsion four times — all O(N) and cheap) and accumulates its own force in
// registers.  No atomics, no zero-fill dependency, bit-reproducible summation order, one launch.
// Energies are taken from the role-0 evaluation of each term only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

#include "common.h"
#include "pair_math.h"
#include "bonded_math.h"

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

// topologies where no atom takes part in more terms than this use the single atom-centric kernel
constexpr int kAtomCentricLimit = 8;


struct Bonded {
  int natoms = 0;
  int nentries = 0;
  int max_entries_per_atom = 0;
  int dih_amber = 1, imp_amber = 1;
  uint32_t terms14 = 0;
  int bonds_use_cutoff = 0;
  DevArr atom_off, atom_ent;
  DevArr arec;  // AtomRec<R>[natoms][max_entries_per_atom] for light topologies
  DevArr bond_idx, bond_prm, angle_idx, angle_prm;
  DevArr dih_idx, dih_start, dih_prm, imp_idx, imp_start, imp_prm;
  DevArr p14_idx, p14_prm;
  void release() {
    for (DevArr *a : {&atom_off, &atom_ent, &arec, &bond_idx, &bond_prm, &angle_idx, &angle_prm, &dih_idx, &dih_start,
                      &dih_prm, &imp_idx, &imp_start, &imp_prm, &p14_idx, &p14_prm})
      a->release();
  }
};

// replica batch: blockIdx.y = replica; boxes[y] = {box[3], 1/box[3]} (null: single replica, box in A.b)
template <typename R>
__device__ __forceinline__ void replica_view(BondedArgs<R> &A, const R *__restrict__ &pos, R *__restrict__ &forces,
                                             double *__restrict__ &energies, const R *__restrict__ boxes,
                                             int natoms) {
  if (!boxes) return;
  const int rep = blockIdx.y;
  pos += (size_t)rep * 3 * natoms;
  if (forces) forces += (size_t)rep * 3 * natoms;
  if (energies) energies += (size_t)rep * kEnergySlots * kEnergyStride;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    A.b.box[k] = boxes[6 * rep + k];
    A.b.invbox[k] = boxes[6 * rep + 3 + k];
  }
}

// (1) light topologies (every atom in a handful of terms: water, ions): four lanes per atom, one launch
template <typename R>
__global__ __launch_bounds__(256) void bonded_atom_kernel(int natoms, BondedArgs<R> A, const R *__restrict__ pos,
                                                          R *__restrict__ forces, double *__restrict__ energies,
                                                          int want_e, const R *__restrict__ boxes) {
  replica_view(A, pos, forces, energies, boxes, natoms);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int a = t / kQuad, sub = t % kQuad;
  R fx = 0, fy = 0, fz = 0;
  double e[TMDHIP_NENERGY] = {0, 0, 0, 0, 0, 0, 0, 0};
  eval_atom_quad<R>(A, pos, a, sub, a < natoms, fx, fy, fz, e);
  if (a < natoms && sub == 0 && forces) {
    if (want_e & 2) {  // TMDHIP_OVERWRITE_FORCES: the bonded force alone (tmdhip_md_run hands it to the pair launch's step blocks)
      forces[3 * a + 0] = fx;
      forces[3 * a + 1] = fy;
      forces[3 * a + 2] = fz;
    } else {
      forces[3 * a + 0] += fx;
      forces[3 * a + 1] += fy;
      forces[3 * a + 2] += fz;
    }
  }
  if (want_e & 1) flush_energies(e, energies);
}

// (2) heavy topologies (proteins: an atom sits in dozens of torsions; one thread walking them is a chain
// of dependent global loads, measured 102 us for alanine dipeptide's 688 atoms): one WAVE per atom, the
// lanes take the atom's (term, role) entries, the three force components are reduced across the wave with
// a fixed butterfly.  One launch, no atomics on the forces, bit-reproducible.
template <typename R>
__global__ __launch_bounds__(256) void bonded_wave_kernel(int natoms, BondedArgs<R> A, const R *__restrict__ pos,
                                                          R *__restrict__ forces, double *__restrict__ energies,
                                                          int want_e, const R *__restrict__ boxes) {
  replica_view(A, pos, forces, energies, boxes, natoms);
  const int lane = threadIdx.x & 63;
  const int a = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  R fx = 0, fy = 0, fz = 0;
  double e[TMDHIP_NENERGY] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (a < natoms)
    for (int q = A.atom_off[a] + lane, qe = A.atom_off[a + 1]; q < qe; q += 64)
      eval_entry<R>(A, pos, (unsigned)A.atom_ent[q], fx, fy, fz, e);
  fx = wave_sum(fx);
  fy = wave_sum(fy);
  fz = wave_sum(fz);
  if (lane == 0 && a < natoms && forces) {
    if (want_e & 2) {
      forces[3 * a + 0] = fx;
      forces[3 * a + 1] = fy;
      forces[3 * a + 2] = fz;
    } else {
      forces[3 * a + 0] += fx;
      forces[3 * a + 1] += fy;
      forces[3 * a + 2] += fz;
    }
  }
  if (want_e & 1) flush_energies(e, energies);
}

template <typename R>
int upload_terms(const int32_t *term_of, const void *prm, int nterms, int ntors, DevArr &start, DevArr &dprm,
                 int &amber) {
  std::vector<int> st(ntors + 1, 0);
  for (int m = 0; m < nterms; ++m) {
    if (term_of[m] < 0 || term_of[m] >= ntors) return fail("tmdhip_set_bonded: torsion term index out of range");
    if (m && term_of[m] < term_of[m - 1])
      return fail("tmdhip_set_bonded: torsion terms must be grouped in ascending order");
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
// accessors implemented in context.hip (the ctx layout is private to the nonbonded engine, engine.h)
void *&ctx_bonded_slot(tmdhip_ctx *ctx);
const tmdhip_nonbonded_desc &ctx_desc(const tmdhip_ctx *ctx);
const void *ctx_scaled_charges(const tmdhip_ctx *ctx);
int ctx_nreplicas(const tmdhip_ctx *ctx);
double *ctx_energy_scratch(const tmdhip_ctx *ctx);
int fold_energies(tmdhip_ctx *ctx, double *energies, hipStream_t st, int nrep);
const void *set_boxes(tmdhip_ctx *ctx, const double *box_host, hipStream_t st);

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
  b->natoms = n;
  std::vector<std::vector<unsigned>> per_atom(n);
  auto add = [&](const int32_t *idx, int count, int width, unsigned kind, const char *what) -> int {
    if (count < 0 || (count > 0 && !idx)) return fail(std::string("tmdhip_set_bonded: bad ") + what + " table");
    if ((unsigned)count >= (1u << kIdxBits)) return fail(std::string("tmdhip_set_bonded: too many ") + what);
    for (int t = 0; t < count; ++t)
      for (int r = 0; r < width; ++r) {
        const int a = idx[t * width + r];
        if (a < 0 || a >= n) return fail(std::string("tmdhip_set_bonded: ") + what + " index out of range");
        per_atom[a].push_back((kind << 28) | ((unsigned)r << kIdxBits) | (unsigned)t);
      }
    return 0;
  };
  // bonds with k0 == 0 contribute exactly zero energy and force (e.g. the H-H "bond" of TIP3P in
  // tests/water/water_forcefield.yaml): they are kept in the tables but get no per-atom entries
  if (d->nbonds) {
    const R *prm = (const R *)d->bond_prm_host;
    std::vector<int32_t> live_idx;
    std::vector<int> live_t;
    if (d->nbonds > 0 && (!d->bond_idx_host || !prm)) return fail("tmdhip_set_bonded: bad bond table");
    for (int t = 0; t < d->nbonds; ++t) {
      for (int r = 0; r < 2; ++r) {
        const int a = d->bond_idx_host[2 * t + r];
        if (a < 0 || a >= n) return fail("tmdhip_set_bonded: bond index out of range");
      }
      if (prm[2 * t] == R(0)) continue;
      for (int r = 0; r < 2; ++r)
        per_atom[d->bond_idx_host[2 * t + r]].push_back((KBOND << 28) | ((unsigned)r << kIdxBits) | (unsigned)t);
    }
    TMD_TRY(b->bond_idx.upload(d->bond_idx_host, (size_t)2 * d->nbonds));
    TMD_TRY(b->bond_prm.upload(prm, (size_t)2 * d->nbonds));
  }
  if (d->nangles) {
    TMD_TRY(add(d->angle_idx_host, d->nangles, 3, KANGLE, "angle"));
    TMD_TRY(b->angle_idx.upload(d->angle_idx_host, (size_t)3 * d->nangles));
    TMD_TRY(b->angle_prm.upload((const R *)d->angle_prm_host, (size_t)2 * d->nangles));
  }
  if (d->ndihedrals) {
    TMD_TRY(add(d->dihedral_idx_host, d->ndihedrals, 4, KDIHEDRAL, "dihedral"));
    TMD_TRY(b->dih_idx.upload(d->dihedral_idx_host, (size_t)4 * d->ndihedrals));
    TMD_TRY((upload_terms<R>(d->dihedral_term_of_host, d->dihedral_prm_host, d->ndihedral_terms, d->ndihedrals,
                             b->dih_start, b->dih_prm, b->dih_amber)));
  }
  if (d->nimpropers) {
    TMD_TRY(add(d->improper_idx_host, d->nimpropers, 4, KIMPROPER, "improper"));
    TMD_TRY(b->imp_idx.upload(d->improper_idx_host, (size_t)4 * d->nimpropers));
    TMD_TRY((upload_terms<R>(d->improper_term_of_host, d->improper_prm_host, d->nimproper_terms, d->nimpropers,
                             b->imp_start, b->imp_prm, b->imp_amber)));
  }
  if (d->n14 && d->terms14) {
    TMD_TRY(add(d->pair14_idx_host, d->n14, 2, KPAIR14, "1-4 pair"));
    TMD_TRY(b->p14_idx.upload(d->pair14_idx_host, (size_t)2 * d->n14));
    TMD_TRY(b->p14_prm.upload((const R *)d->pair14_prm_host, (size_t)4 * d->n14));
  }
  std::vector<int> off(n + 1, 0);
  for (int a = 0; a < n; ++a) off[a + 1] = off[a] + (int)per_atom[a].size();
  std::vector<int> ent((size_t)off[n] + 1);
  for (int a = 0; a < n; ++a) std::copy(per_atom[a].begin(), per_atom[a].end(), ent.begin() + off[a]);
  b->nentries = off[n];
  for (int a = 0; a < n; ++a) b->max_entries_per_atom = std::max(b->max_entries_per_atom, off[a + 1] - off[a]);
  TMD_TRY(b->atom_off.upload(off.data(), off.size()));
  TMD_TRY(b->atom_ent.upload(ent.data(), ent.size()));
  if (b->max_entries_per_atom <= kAtomCentricLimit && b->nentries > 0) {
    // per-atom records with the bond / angle partners and parameters inline (bonded_math.h: AtomRec)
    const int K = b->max_entries_per_atom;
    std::vector<AtomRec<R>> recs((size_t)n * K);
    for (auto &r : recs) r = AtomRec<R>{kNoRec, 0, 0, R(0), R(0)};
    const R *bprm = (const R *)d->bond_prm_host, *aprm = (const R *)d->angle_prm_host;
    for (int a = 0; a < n; ++a)
      for (size_t k = 0; k < per_atom[a].size(); ++k) {
        const unsigned e = per_atom[a][k];
        const unsigned kind = e >> 28;
        const int role = (int)((e >> kIdxBits) & 3u), t = (int)(e & ((1u << kIdxBits) - 1u));
        AtomRec<R> r{e, 0, 0, R(0), R(0)};
        if (kind == KBOND) {
          r.a = d->bond_idx_host[2 * t + (role == 0 ? 1 : 0)];
          r.p0 = bprm[2 * t], r.p1 = bprm[2 * t + 1];
        } else if (kind == KANGLE) {
          const int32_t *ix = d->angle_idx_host + 3 * t;
          r.a = role == 0 ? ix[1] : ix[0];
          r.b = role == 2 ? ix[1] : ix[2];
          r.p0 = aprm[2 * t], r.p1 = aprm[2 * t + 1];
        }
        recs[(size_t)a * K + k] = r;
      }
    TMD_TRY(b->arec.upload(recs.data(), recs.size()));
  }
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
void fill_args(tmdhip_ctx *ctx, const Bonded *b, const double *box, BondedArgs<R> &A) {
  A.arec = b->arec.as<AtomRec<R>>();
  A.arec_stride = b->max_entries_per_atom;
  A.atom_off = b->atom_off.as<int>();
  A.atom_ent = b->atom_ent.as<int>();
  A.bond_idx = b->bond_idx.as<int>();
  A.bond_prm = b->bond_prm.as<R>();
  A.angle_idx = b->angle_idx.as<int>();
  A.angle_prm = b->angle_prm.as<R>();
  A.dih_idx = b->dih_idx.as<int>();
  A.dih_start = b->dih_start.as<int>();
  A.dih_prm = b->dih_prm.as<R>();
  A.imp_idx = b->imp_idx.as<int>();
  A.imp_start = b->imp_start.as<int>();
  A.imp_prm = b->imp_prm.as<R>();
  A.p14_idx = b->p14_idx.as<int>();
  A.p14_prm = b->p14_prm.as<R>();
  A.qs = (const R *)ctx_scaled_charges(ctx);
  A.dih_amber = b->dih_amber;
  A.imp_amber = b->imp_amber;
  A.terms14 = b->terms14;
  A.bond_r2max = b->bonds_use_cutoff ? host_r2max<R>(ctx_desc(ctx).cutoff) : std::numeric_limits<R>::infinity();
  const bool allzero = box[0] == 0 && box[1] == 0 && box[2] == 0;
  for (int k = 0; k < 3; ++k) {
    A.b.box[k] = (R)box[k];
    A.b.invbox[k] = (!allzero && A.b.box[k] != R(0)) ? R(1) / A.b.box[k] : R(0);
  }
}

template <typename R>
int run_bonded(tmdhip_ctx *ctx, Bonded *b, const void *pos_v, const double *box, void *forces_v, double *en,
               int flags, hipStream_t st, int nrep) {
  if (b->nentries == 0) return 0;
  const R *boxes = nullptr;
  if (nrep > 1) {
    boxes = (const R *)set_boxes(ctx, box, st);
    if (!boxes) return fail("could not upload the replica boxes");
  }
  BondedArgs<R> A;
  fill_args<R>(ctx, b, box, A);
  R *forces = (flags & TMDHIP_WANT_FORCES) ? (R *)forces_v : nullptr;
  const int we = ((flags & TMDHIP_WANT_ENERGY) ? 1 : 0) | ((flags & TMDHIP_OVERWRITE_FORCES) ? 2 : 0);  // kernel mode bits
  const int n = b->natoms;
  if (b->max_entries_per_atom <= kAtomCentricLimit) {
    hipLaunchKernelGGL((bonded_atom_kernel<R>), dim3((kQuad * n + 255) / 256, nrep), dim3(256), 0, st, n, A,
                       (const R *)pos_v, forces, ctx_energy_scratch(ctx), we, boxes);
  } else {
    hipLaunchKernelGGL((bonded_wave_kernel<R>), dim3((n + 3) / 4, nrep), dim3(256), 0, st, n, A, (const R *)pos_v,
                       forces, ctx_energy_scratch(ctx), we, boxes);
  }
  TMD_HIP(hipGetLastError());
  if (we & 1) TMD_TRY(fold_energies(ctx, en, st, nrep));
  return 0;
}

}  // namespace

namespace tmd {
// Arguments for evaluating the bonded force of an atom inline in the MD-step kernels (md_loop.hip, pair_fast_f32.hip).
// 0: no bonded terms; 1: light topology (thread per atom, per-atom records); 2: heavy (wave per atom).
int bonded_inline_args(tmdhip_ctx *ctx, const double *box, BondedArgs<float> &A) {
  const Bonded *b = (const Bonded *)ctx_bonded_slot(ctx);
  if (!b || b->nentries == 0) return 0;
  fill_args<float>(ctx, b, box, A);
  return b->max_entries_per_atom <= kAtomCentricLimit ? 1 : 2;
}
int bonded_inline_args(tmdhip_ctx *ctx, const double *box, BondedArgs<double> &A) {
  const Bonded *b = (const Bonded *)ctx_bonded_slot(ctx);
  if (!b || b->nentries == 0) return 0;
  fill_args<double>(ctx, b, box, A);
  return b->max_entries_per_atom <= kAtomCentricLimit ? 1 : 2;
}
}  // namespace tmd

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
  (void)hipStreamSynchronize(nullptr);  // (null-stream uploads complete before a non-blocking stream of the caller uses them)
  return 0;
}

int tmdhip_compute_bonded(tmdhip_ctx *ctx, int replica, const void *pos_dev, const double *box_host,
                          void *forces_dev, double *energies_dev, int flags, void *stream) {
  if (!ctx || !pos_dev || !box_host) return fail("tmdhip_compute_bonded: null argument");
  if (replica != TMDHIP_ALL_REPLICAS && (replica < 0 || replica >= ctx_nreplicas(ctx)))
    return fail("tmdhip_compute_bonded: bad replica index");
  if ((flags & TMDHIP_WANT_FORCES) && !forces_dev)
    return fail("tmdhip_compute_bonded: forces requested without a buffer");
  if ((flags & TMDHIP_WANT_ENERGY) && !energies_dev)
    return fail("tmdhip_compute_bonded: energies requested without a buffer");
  Bonded *b = (Bonded *)ctx_bonded_slot(ctx);
  hipStream_t st = (hipStream_t)stream;
  const bool overwrite = (flags & TMDHIP_WANT_FORCES) && (flags & TMDHIP_OVERWRITE_FORCES);
  if (!b || b->nentries == 0) {
    // no bonded terms: "the bonded force alone" is zero (TMDHIP_OVERWRITE_FORCES), an accumulating call changes nothing
    if (overwrite) {
      const size_t nrep0 = replica == TMDHIP_ALL_REPLICAS ? (size_t)ctx_nreplicas(ctx) : 1;
      const size_t esz = ctx_desc(ctx).dtype == TMDHIP_F32 ? 4 : 8;
      TMD_HIP(hipMemsetAsync(forces_dev, 0, esz * 3 * (size_t)ctx_desc(ctx).natoms * nrep0, st));
    }
    return 0;
  }
  // all replicas ([R][N][3] positions/forces, [R][8] energies, [R][3] boxes): one launch with grid.y = R
  const int nrep = replica == TMDHIP_ALL_REPLICAS ? ctx_nreplicas(ctx) : 1;
  return ctx_desc(ctx).dtype == TMDHIP_F32
             ? run_bonded<float>(ctx, b, pos_dev, box_host, forces_dev, energies_dev, flags, st, nrep)
             : run_bonded<double>(ctx, b, pos_dev, box_host, forces_dev, energies_dev, flags, st, nrep);
}

}  // extern "C"
