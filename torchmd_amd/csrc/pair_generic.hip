// Generic pair kernels of the nonbonded engine for gfx950 (MI355X) and the dispatch between all pair kernels.
//
// Reference semantics: torchmd/forces.py:260-319 (nonbonded block of Forces.compute) with 348-357 (pair set = all i<j
// minus exclusions), 360-372 (minimum image, distances), 76-81 (cutoff filter) and 381-491 (pair potentials; pair_math.h).
// The reference evaluates a dense [P,2] pair tensor every step; here each unique pair is evaluated from both of its
// atoms (no atomics on the list path, no j-force reduction).
//   K4  allpairs_kernel        tiled O(N^2) for small systems and contexts without a cutoff
//   K3  list_pair_kernel       any term mix over the Verlet list (repulsion terms, pair counting, > 32 LJ classes);
//                              LJ and/or electrostatics go to the lean kernels of pair_fast_f32.hip / pair_lean_f64.hip
#include "engine.h"

namespace tmd {

// ---- K4: tiled all-pairs ----------------------------------------------------------------------
// grid = (ceil(N/64), nsplit); one wave per block.  Lane = one i atom, the j range of this block is
// streamed through LDS in tiles of 64 (broadcast reads).  Every (i,j) with i != j is evaluated from
// i's side only, so forces need no cross-lane reduction; blocks with different j ranges combine
// through one atomic add per atom.
constexpr int kAllpairsLdsTypes = 16;  // LJ classes whose table the all-pairs kernel stages in LDS (2 / 4 KB)
template <typename R, bool ENERGY>
__global__ __launch_bounds__(64) void allpairs_kernel(
    int n, const R *__restrict__ pos, const R *__restrict__ qs, const int *__restrict__ types, int ntypes,
    const typename Vec<R>::T2 *__restrict__ tab, const int *__restrict__ excl_off,
    const int *__restrict__ excl_idx, PairConsts<R> c, int jchunk, R *__restrict__ forces,
    double *__restrict__ energies, unsigned long long *__restrict__ paircount, const R *__restrict__ boxes,
    int nsplit, BondedArgs<R> B) {
  using R4 = typename Vec<R>::T4;
  __shared__ R4 sj[64];
  __shared__ int st[64];
  if (boxes) {  // replica batch: blockIdx.z = replica, boxes[z] = {box[3], 1/box[3]} (see set_boxes)
    const int rep = blockIdx.z;
    pos += (size_t)rep * 3 * n;
    if (forces) forces += (size_t)rep * 3 * n;
    if (ENERGY) energies += (size_t)rep * kEnergySlots * kEnergyStride;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      c.box[k] = boxes[6 * rep + k];
      c.invbox[k] = boxes[6 * rep + 3 + k];
    }
  }
  const int lane = threadIdx.x;
  if ((int)blockIdx.y >= nsplit) {
    // rows of the grid beyond the pair blocks: the bonded terms of heavy topologies ride on this launch
    // (small systems are launch-bound).  One wave per atom like bonded_wave_kernel; the force joins the
    // pair blocks' partial sums with atomics.
    const int a = ((int)blockIdx.y - nsplit) * (int)gridDim.x + (int)blockIdx.x;
    if (a >= n) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      B.b.box[k] = c.box[k];
      B.b.invbox[k] = c.invbox[k];
    }
    R bx = 0, by = 0, bz = 0;
    double e[TMDHIP_NENERGY] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = B.atom_off[a] + lane, qe = B.atom_off[a + 1]; q < qe; q += 64)
      eval_entry<R>(B, pos, (unsigned)B.atom_ent[q], bx, by, bz, e);
    bx = wave_sum(bx);
    by = wave_sum(by);
    bz = wave_sum(bz);
    if (lane == 0 && forces) {
      unsafeAtomicAdd(&forces[3 * a + 0], bx);
      unsafeAtomicAdd(&forces[3 * a + 1], by);
      unsafeAtomicAdd(&forces[3 * a + 2], bz);
    }
    if (ENERGY) flush_energies(e, energies);
    return;
  }
  const int i = blockIdx.x * 64 + lane;
  const bool active = i < n;
  const int jbeg = blockIdx.y * jchunk;
  const int jend = min(n, jbeg + jchunk);

  // Round 5: small systems run ~1 000 of these waves for a few hundred atoms — 64 x 16 pairs each — so a wave's life is
  // its chain of memory round trips, not its arithmetic (688-atom alanine dipeptide: 12 us per launch).  The chain was:
  // own atom + exclusion offsets -> the exclusion row scanned entry by entry up to the wave's j range (one dependent load
  // each: ~20 for a solute atom) -> barrier -> the j tile -> per hit a load of the LJ table from global memory.  Now:
  // ONE batch holds the own atom, the exclusion offsets and the first j tile; a second one the first eight entries of
  // the exclusion row (tested against every tile from registers; longer rows read the rest per tile); the LJ table is
  // staged in LDS (up to kAllpairsLdsTypes classes).  Same arithmetic on the same values: results are unchanged.
  constexpr int kExReg = 8;
  __shared__ typename Vec<R>::T2 s_tab[kAllpairsLdsTypes * kAllpairsLdsTypes];
  const bool tab_in_lds = ntypes <= kAllpairsLdsTypes;
  R xi = 0, yi = 0, zi = 0, qi = 0;
  int trow = 0;
  int ebeg = 0, eend = 0;
  R4 v0;  // this lane's record of the first tile
  v0.x = v0.y = v0.z = v0.w = R(0);
  int t0 = 0;
  {
    const int jl = jbeg + lane;
    if (jl < jend) {
      v0.x = pos[3 * jl + 0];
      v0.y = pos[3 * jl + 1];
      v0.z = pos[3 * jl + 2];
      v0.w = qs[jl];
      t0 = types[jl];
    }
  }
  if (active) {
    xi = pos[3 * i + 0];
    yi = pos[3 * i + 1];
    zi = pos[3 * i + 2];
    qi = qs[i];
    trow = types[i] * ntypes;
    ebeg = excl_off[i];
    eend = excl_off[i + 1];
  }
  if (tab_in_lds)
    for (int t = lane; t < ntypes * ntypes; t += 64) s_tab[t] = tab[t];
  int ex[kExReg];
#pragma unroll
  for (int k = 0; k < kExReg; ++k) ex[k] = (ebeg + k < eend) ? excl_idx[ebeg + k] : -1;
  R fx = 0, fy = 0, fz = 0;
  R en[4] = {0, 0, 0, 0};
  unsigned long long cnt = 0;

  for (int j0 = jbeg; j0 < jend; j0 += 64) {
    __syncthreads();
    const int jl = j0 + lane;
    if (j0 == jbeg) {
      sj[lane] = v0;
      st[lane] = t0;
    } else if (jl < jend) {
      R4 v;
      v.x = pos[3 * jl + 0];
      v.y = pos[3 * jl + 1];
      v.z = pos[3 * jl + 2];
      v.w = qs[jl];
      sj[lane] = v;
      st[lane] = types[jl];
    }
    __syncthreads();
    // exclusion mask of this tile for atom i: the row's first entries from registers, the rest (long rows: proteins) read here
    unsigned long long skip = 0;
#pragma unroll
    for (int k = 0; k < kExReg; ++k) {
      const unsigned d = (unsigned)(ex[k] - j0);
      if (d < 64u) skip |= 1ull << d;
    }
    for (int e = ebeg + kExReg; e < eend; ++e) {
      const int x = excl_idx[e];
      if (x >= j0 + 64) break;  // (rows are sorted)
      if (x >= j0) skip |= 1ull << (x - j0);
    }
    if (i >= j0 && i < j0 + 64) skip |= 1ull << (i - j0);
    const int tile = min(64, jend - j0);
    for (int k = 0; k < tile; ++k) {
      const R4 pj = sj[k];
      const R dx = min_image(xi - pj.x, c.box[0], c.invbox[0]);
      const R dy = min_image(yi - pj.y, c.box[1], c.invbox[1]);
      const R dz = min_image(zi - pj.z, c.box[2], c.invbox[2]);
      const R r2 = norm2(dx, dy, dz);
      const bool hit = active && !((skip >> k) & 1ull) && (r2 <= c.r2max);
      if (hit) {
        const typename Vec<R>::T2 ab = tab_in_lds ? s_tab[trow + st[k]] : tab[trow + st[k]];
        const R fs = pair_terms<R, ENERGY>(c, r2, qi * pj.w, ab.x, ab.y, en);
        fx -= dx * fs;
        fy -= dy * fs;
        fz -= dz * fs;
        if (j0 + k > i) ++cnt;
      }
    }
  }
  if (active && forces) {
    unsafeAtomicAdd(&forces[3 * i + 0], fx);
    unsafeAtomicAdd(&forces[3 * i + 1], fy);
    unsafeAtomicAdd(&forces[3 * i + 2], fz);
  }
  if (ENERGY) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double s = wave_sum((double)en[t]);
      if (lane == 0 && s != 0.0) unsafeAtomicAdd(&energy_row(energies)[t], 0.5 * s);
    }
  }
  if (paircount) {
    const unsigned long long s = wave_sum(cnt);
    if (lane == 0 && s) atomicAdd(paircount, s);
  }
}

// ---- K3: list pair kernel ------------------------------------------------------------------------
// LPA lanes cooperate on one atom (strided over its list), APW = 64/LPA atoms per wave.
// The neighbour stream is read with one coalesced 256-B load per wave and iteration; UNROLL
// iterations are issued together so that their index loads and the dependent position gathers
// overlap (memory-level parallelism), and the pair maths is predicated instead of branched.
// FAST = 1 is the branch-free specialisation for LJ + reaction-field electrostatics without
// switching (the water benchmark); FAST = 0 takes every option from PairConsts at run time.
template <typename R>
__device__ __forceinline__ R pair_fast_lj_rf(const PairConsts<R> &c, R r2, R qq, R A, R B) {
  const R rinv = fast_rsqrt(r2);
  const R rinv2 = rinv * rinv;
  const R rinv6 = rinv2 * rinv2 * rinv2;
  // (dE_lj/dr + dE_rf/dr) / r
  return (R(-12) * A * rinv6 + R(6) * B) * rinv6 * rinv2 + qq * (R(2) * c.krf - rinv2 * rinv);
}

template <typename R, bool ENERGY, int LPA, int FAST>
__global__ __launch_bounds__(256) void list_pair_kernel(
    int n, const typename Vec<R>::T4 *__restrict__ sorted, const int *__restrict__ stype,
    const int *__restrict__ order, int ntypes, const typename Vec<R>::T2 *__restrict__ tab,
    const unsigned *__restrict__ nlist, const int *__restrict__ nneigh, int maxn, PairConsts<R> c,
    R *__restrict__ forces, int overwrite, double *__restrict__ energies,
    unsigned long long *__restrict__ paircount, unsigned *publish, unsigned publish_value) {
  using R4 = typename Vec<R>::T4;
  using R2 = typename Vec<R>::T2;
  constexpr int APW = 64 / LPA;
  constexpr int UNROLL = 4;
  // tells the host (host-mapped word) that everything enqueued before this launch has completed
  if (publish && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(publish, publish_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  extern __shared__ __align__(16) unsigned char smem[];
  R2 *stab = reinterpret_cast<R2 *>(smem);
  for (int t = threadIdx.x; t < ntypes * ntypes; t += blockDim.x) stab[t] = tab[t];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int a = wave * APW + lane / LPA;
  const int sub = lane % LPA;
  const bool active = a < n;
  const int aself = active ? a : 0;
  R4 pi;
  pi.x = pi.y = pi.z = pi.w = 0;
  int nn = 0, trow = 0;
  if (active) {
    pi = sorted[a];
    nn = nneigh[a];
    trow = stype[a] * ntypes;
  }
  int nmax = nn;
#pragma unroll
  for (int o = 32; o >= LPA; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
  const int nkk = (nmax + LPA - 1) / LPA;
  const unsigned *row = nlist + (size_t)wave * maxn * APW + lane * 4;  // + (kk / 4) * 256 + kk % 4

  R fx = 0, fy = 0, fz = 0;
  R en[4] = {0, 0, 0, 0};
  unsigned cnt = 0;
  for (int kk0 = 0; kk0 < nkk; kk0 += UNROLL) {
    unsigned entry[UNROLL];
    R4 pj[UNROLL];
    bool valid[UNROLL];
    int jdx[UNROLL], tj[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) entry[u] = (kk0 + u < nkk) ? row[(size_t)(kk0 >> 2) * 256 + u] : 0u;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      valid[u] = (kk0 + u) * LPA + sub < nn;
      jdx[u] = valid[u] ? (int)((entry[u] & kEntryOffMask) >> 4) : aself;
      pj[u] = sorted[jdx[u]];
      tj[u] = !valid[u] ? 0 : (ntypes <= kEntryTypes ? (int)(entry[u] >> kEntryTypeShift) : stype[jdx[u]]);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const R dx = min_image(pi.x - pj[u].x, c.box[0], c.invbox[0]);
      const R dy = min_image(pi.y - pj[u].y, c.box[1], c.invbox[1]);
      const R dz = min_image(pi.z - pj[u].z, c.box[2], c.invbox[2]);
      const R r2 = norm2(dx, dy, dz);
      const bool hit = valid[u] && (r2 <= c.r2max);
      const R2 ab = stab[trow + tj[u]];
      const R r2s = hit ? r2 : R(1);
      R fs;
      if (FAST == 1 && !ENERGY) {
        fs = pair_fast_lj_rf<R>(c, r2s, pi.w * pj[u].w, ab.x, ab.y);
      } else {
        R e4[4] = {0, 0, 0, 0};
        fs = pair_terms<R, ENERGY>(c, r2s, pi.w * pj[u].w, ab.x, ab.y, e4);
        if (ENERGY) {
#pragma unroll
          for (int t = 0; t < 4; ++t) en[t] += hit ? e4[t] : R(0);
        }
      }
      fs = hit ? fs : R(0);
      fx -= dx * fs;
      fy -= dy * fs;
      fz -= dz * fs;
      cnt += hit ? 1u : 0u;
    }
  }
#pragma unroll
  for (int o = LPA >> 1; o > 0; o >>= 1) {
    fx += __shfl_xor(fx, o, 64);
    fy += __shfl_xor(fy, o, 64);
    fz += __shfl_xor(fz, o, 64);
  }
  if (active && sub == 0 && forces) {
    const int oi = order[a];
    if (overwrite) {
      forces[3 * oi + 0] = fx;
      forces[3 * oi + 1] = fy;
      forces[3 * oi + 2] = fz;
    } else {
      forces[3 * oi + 0] += fx;
      forces[3 * oi + 1] += fy;
      forces[3 * oi + 2] += fz;
    }
  }
  if (ENERGY) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double s = wave_sum((double)en[t]);
      if (lane == 0 && s != 0.0) unsafeAtomicAdd(&energy_row(energies)[t], 0.5 * s);
    }
  }
  if (paircount) {
    const unsigned long long s = wave_sum((unsigned long long)cnt);
    if (lane == 0 && s) atomicAdd(paircount, s);
  }
}

__global__ void halve_count_kernel(unsigned long long *c) { *c >>= 1; }

int halve_pair_count(unsigned long long *count_dev, hipStream_t st) {
  hipLaunchKernelGGL(halve_count_kernel, dim3(1), dim3(1), 0, st, count_dev);
  TMD_HIP(hipGetLastError());
  return 0;
}

template <typename R>
int launch_allpairs(tmdhip_ctx *ctx, const void *pos, const double *box, void *forces, double *energies,
                    int flags, unsigned long long *paircount, hipStream_t st, int nrep,
                    const BondedArgs<R> *bonded) {
  // nrep > 1: pos/forces/energies/box are the arrays of all replicas ([nrep][n][3], [nrep][8], [nrep][3])
  // and one launch (grid.z = replica) serves them all — small systems are launch-bound
  const int n = ctx->d.natoms;
  const PairConsts<R> c = make_consts<R>(ctx, box);
  const R *boxes = nullptr;
  if (nrep > 1) {
    boxes = (const R *)tmd::set_boxes(ctx, box, st);
    if (!boxes) return fail("could not upload the replica boxes");
  }
  if ((flags & TMDHIP_OVERWRITE_FORCES) && (flags & TMDHIP_WANT_FORCES) && !(flags & kForcesZeroed))
    TMD_HIP(hipMemsetAsync(forces, 0, sizeof(R) * 3 * (size_t)n * nrep, st));  // partial sums are combined with atomics
  const int nb = (n + 63) / 64;
  // split the j range so that ~1024 waves are in flight even for a few hundred atoms (each block then
  // walks a short j range; the partial forces are combined with one atomic per atom and split)
  // (round 5: the wave target grows with the work — n^2 x replicas pairs / 512, 1 024 .. 4 096 waves: a batch of 16
  // alanine-dipeptide replicas ran 880 waves of 144 iterations each, one per SIMD, nothing to hide a wave's latency
  // behind.  Measured, us per MD step at 1 024 / 2 048 / 4 096 / 8 192 waves (profiles/r05_allpairs_waves.txt): ala2 x 16
  // 71.1 / 43.3 / 37.8 / 44.2, tests/water x 64 46.0 / 29.5 / 23.0 / 25.7, x 16 18.5 / 15.3 / 15.2 / 15.6.  TMDHIP_ALLPAIRS_WAVES overrides.)
  static const int waves_env = [] { const char *e = std::getenv("TMDHIP_ALLPAIRS_WAVES"); return e ? std::atoi(e) : 0; }();
  const double pairs = (double)n * (double)n * (double)nrep;
  const int want_waves = waves_env > 0 ? waves_env : (int)std::min(4096.0, std::max(1024.0, pairs / 512.0));
  int nsplit = std::max(1, std::min((n + 15) / 16, want_waves / std::max(nb * nrep, 1)));
  int jchunk = ((n + nsplit - 1) / nsplit + 15) / 16 * 16;
  nsplit = (n + jchunk - 1) / jchunk;
  // `bonded` (heavy topologies, MD loop): n more one-wave blocks evaluate the bonded terms in the same launch
  dim3 grid(nb, nsplit + (bonded ? (n + nb - 1) / nb : 0), nrep);
  const BondedArgs<R> B = bonded ? *bonded : BondedArgs<R>{};
  R *f = (flags & TMDHIP_WANT_FORCES) ? (R *)forces : nullptr;
  using R2 = typename Vec<R>::T2;
  if (flags & TMDHIP_WANT_ENERGY)
    hipLaunchKernelGGL((allpairs_kernel<R, true>), grid, dim3(64), 0, st, n, (const R *)pos,
                       ctx->qs.as<R>(), ctx->types.as<int>(), ctx->d.ntypes, ctx->tab.as<R2>(),
                       ctx->excl_off.as<int>(), ctx->excl_idx.as<int>(), c, jchunk, f, ctx->escratch.as<double>(),
                       paircount, boxes, nsplit, B);
  else
    hipLaunchKernelGGL((allpairs_kernel<R, false>), grid, dim3(64), 0, st, n, (const R *)pos,
                       ctx->qs.as<R>(), ctx->types.as<int>(), ctx->d.ntypes, ctx->tab.as<R2>(),
                       ctx->excl_off.as<int>(), ctx->excl_idx.as<int>(), c, jchunk, f, nullptr, paircount, boxes, nsplit,
                       B);
  TMD_HIP(hipGetLastError());
  if (flags & TMDHIP_WANT_ENERGY) TMD_TRY(tmd::fold_energies(ctx, energies, st, nrep));
  return 0;
}

template <typename R, bool ENERGY>
int launch_list_pair(tmdhip_ctx *ctx, Replica &rp, const PairConsts<R> &c, R *f, int overwrite, double *energies,
                     unsigned long long *paircount, hipStream_t st, hipEvent_t e0, hipEvent_t e1, int lmode,
                     const FusedLaunchT<R> *fl, bool fold) {
  using R4 = typename Vec<R>::T4;
  using R2 = typename Vec<R>::T2;
  const int n = ctx->d.natoms;
  const int apw = rp.lg.apw;
  const int waves = (n + apw - 1) / apw;
  const int blocks = (waves + 3) / 4;
  const size_t shmem = (size_t)ctx->d.ntypes * ctx->d.ntypes * sizeof(R2);
  // the lean fp32 kernel covers LJ (with or without switching) and/or electrostatics (reaction field or plain Coulomb)
  const bool only_lj_el = c.terms != 0 && (c.terms & ~(TMDHIP_TERM_LJ | TMDHIP_TERM_ELECTROSTATICS)) == 0;
  const bool fast = only_lj_el;  // (switching, if any, acts on the LJ term and is a kernel variant)
  const bool lean = fast && !paircount && ctx->d.ntypes <= kEntryTypes;  // (the entry's type field holds 32 LJ classes)
  if constexpr (std::is_same<R, float>::value) {
    if (lean && (f || ENERGY || fl)) {  // lean fp32 kernel (n > 2^20: every iteration in its checked loop)
      TMD_TRY(launch_pair_fast_f32<ENERGY>(ctx, rp, c, f, overwrite, st, e0, e1, lmode, fl));
      if (ENERGY && fold) TMD_TRY(tmd::fold_energies(ctx, energies, st, 1));
      return 0;
    }
  }
  if constexpr (std::is_same<R, double>::value) {
    if (lean && !fl && (f || ENERGY)) {  // lean fp64 kernel (same conditions as the fp32 one; never fused)
      TMD_TRY(launch_pair_lean_f64<ENERGY>(ctx, rp, c, f, overwrite, st, e0, e1, lmode));
      if (ENERGY && fold) TMD_TRY(tmd::fold_energies(ctx, energies, st, 1));
      return 0;
    }
  }
  if (fl) return fail("fused MD step: the context does not run a lean pair kernel");
#define TMD_LAUNCH(L, F)                                                                                \
  launch_with_events(list_pair_kernel<R, ENERGY, L, F>, dim3(blocks), dim3(256), shmem, st, e0, e1, n,  \
                     rp.sorted.as<R4>(), rp.stype.as<int>(), rp.order.as<int>(), ctx->d.ntypes,          \
                     ctx->tab.as<R2>(), rp.nlist.as<unsigned>(), rp.nneigh.as<int>(), rp.lg.maxn, c, f,  \
                     overwrite, ctx->escratch.as<double>(), paircount, rp.pub_ptr, rp.pub_val)
  // the generic kernel's branch-free FAST=1 body hard-codes LJ + electrostatics (krf = 0: plain Coulomb)
  const bool fast_generic =
      fast && !c.switch_on && !ENERGY && c.terms == (TMDHIP_TERM_LJ | TMDHIP_TERM_ELECTROSTATICS);
#define TMD_LAUNCH_LPA(L)     \
  if (fast_generic) {         \
    TMD_LAUNCH(L, 1);         \
  } else {                    \
    TMD_LAUNCH(L, 0);         \
  }
  switch (rp.lg.lpa) {
    case 1: TMD_LAUNCH_LPA(1); break;
    case 2: TMD_LAUNCH_LPA(2); break;
    case 4: TMD_LAUNCH_LPA(4); break;
    case 8: TMD_LAUNCH_LPA(8); break;
    case 16: TMD_LAUNCH_LPA(16); break;
    case 32: TMD_LAUNCH_LPA(32); break;
    default: TMD_LAUNCH_LPA(64); break;
  }
#undef TMD_LAUNCH_LPA
#undef TMD_LAUNCH
  TMD_HIP(hipGetLastError());
  if (ENERGY && fold) TMD_TRY(tmd::fold_energies(ctx, energies, st, 1));
  return 0;
}

template int launch_allpairs<float>(tmdhip_ctx *, const void *, const double *, void *, double *, int, unsigned long long *,
                                    hipStream_t, int, const BondedArgs<float> *);
template int launch_allpairs<double>(tmdhip_ctx *, const void *, const double *, void *, double *, int, unsigned long long *,
                                     hipStream_t, int, const BondedArgs<double> *);
#define TMD_INSTANTIATE_LLP(R, E)                                                                                          \
  template int launch_list_pair<R, E>(tmdhip_ctx *, Replica &, const PairConsts<R> &, R *, int, double *, unsigned long long *, \
                                      hipStream_t, hipEvent_t, hipEvent_t, int, const FusedLaunchT<R> *, bool)
TMD_INSTANTIATE_LLP(float, true);
TMD_INSTANTIATE_LLP(float, false);
TMD_INSTANTIATE_LLP(double, true);
TMD_INSTANTIATE_LLP(double, false);
#undef TMD_INSTANTIATE_LLP

}  // namespace tmd
