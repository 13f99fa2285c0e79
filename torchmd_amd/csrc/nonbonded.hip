// Nonbonded force/energy engine for gfx950 (MI355X): tiled all-pairs kernel, cell binning,
// Verlet-list build and the list pair kernel, plus the C-ABI context that owns their buffers.
//
// Reference semantics: torchmd/forces.py:260-319 (nonbonded block of Forces.compute) with
// 348-357 (pair set = all i<j minus exclusions), 360-372 (minimum image, distances), 76-81
// (cutoff filter) and 381-491 (pair potentials).  The reference evaluates a dense [P,2] pair tensor
// every step; here the same pair set is produced by an O(N) cell list + Verlet list with a skin,
// and each unique pair is evaluated from both of its atoms (full list, no atomics, no j-force
// reduction) — a gather/stream kernel bound by HBM/L2 traffic of the neighbour indices and by the
// fp32 vector ALU, not an MFMA problem.
//
// Data layout in HBM (per replica):
//   sorted_xyzq  real4[N]   positions in cell-sorted order + scaled charge q*sqrt(k_e)
//   sorted_type  int32[N]
//   order        int32[N]   cell-sorted slot -> original atom index
//   nlist        uint32[G * maxn * APW]   G = ceil(N/APW) wave groups, APW = 64/LPA atoms per wave;
//                entry k of the a-th atom of group g belongs to lane l = a*LPA + k%LPA, iteration kk = k/LPA,
//                and lives at  g*maxn*APW + ((kk/4)*64 + l)*4 + kk%4 : a lane's entries of four consecutive
//                iterations are one 16-byte word, so one wave-wide dwordx4 load reads 1 KB of contiguous list.
//                (Consecutive entries in ADJACENT LANES matter: candidates arrive in cell-sorted order, so the
//                lanes of an atom gather runs of consecutive records, which the texture path serves faster —
//                tools/ubench/gather_rate.hip.  Giving each lane four consecutive entries instead would make
//                the build's store address three instructions but costs the pair kernel 46 -> 52 us.)
//                entry = type_j << 27 | j << 4 (j = sorted slot)
//   nneigh       int32[N]
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <type_traits>
#include <vector>

#include "common.h"
#include "pair_math.h"
#include "bonded_math.h"
#include "rng.h"

namespace tmd {

std::string &last_error() {
  static thread_local std::string s;
  return s;
}
int fail(const std::string &msg) {
  last_error() = msg;
  return -1;
}

// =============================================================================================
// device kernels
// =============================================================================================

// ---- K4: tiled all-pairs ----------------------------------------------------------------------
// grid = (ceil(N/64), nsplit); one wave per block.  Lane = one i atom, the j range of this block is
// streamed through LDS in tiles of 64 (broadcast reads).  Every (i,j) with i != j is evaluated from
// i's side only, so forces need no cross-lane reduction; blocks with different j ranges combine
// through one atomic add per atom.
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

  R xi = 0, yi = 0, zi = 0, qi = 0;
  int trow = 0;
  int e = 0, eend = 0;
  if (active) {
    xi = pos[3 * i + 0];
    yi = pos[3 * i + 1];
    zi = pos[3 * i + 2];
    qi = qs[i];
    trow = types[i] * ntypes;
    e = excl_off[i];
    eend = excl_off[i + 1];
    while (e < eend && excl_idx[e] < jbeg) ++e;
  }
  R fx = 0, fy = 0, fz = 0;
  R en[4] = {0, 0, 0, 0};
  unsigned long long cnt = 0;

  for (int j0 = jbeg; j0 < jend; j0 += 64) {
    __syncthreads();
    const int jl = j0 + lane;
    if (jl < jend) {
      R4 v;
      v.x = pos[3 * jl + 0];
      v.y = pos[3 * jl + 1];
      v.z = pos[3 * jl + 2];
      v.w = qs[jl];
      sj[lane] = v;
      st[lane] = types[jl];
    }
    __syncthreads();
    // exclusion mask of this tile for atom i (rows of the CSR are sorted)
    unsigned long long skip = 0;
    while (e < eend && excl_idx[e] < j0 + 64) {
      skip |= 1ull << (excl_idx[e] - j0);
      ++e;
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
        const typename Vec<R>::T2 ab = tab[trow + st[k]];
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

// ---- cell grid ----------------------------------------------------------------------------------
struct Grid {
  int nc[3];
  int m;          // stencil half-width in cells (1..3)
  signed char zreach[7][7];  // per (x, y) stencil row: largest |z offset| whose cell can hold an atom within
                             // rlist of the home cell, -1 if none (index = offset + m)
  int periodic;   // 1: wrap cell coordinates, 0: clamp (open boundaries)
  double origin[3];
  double inv_edge[3];  // cells per Angstrom
};

template <typename R>
__device__ __forceinline__ int cell_coord(R x, const Grid &g, int d) {
  double f = ((double)x - g.origin[d]) * g.inv_edge[d];
  int nc = g.nc[d];
  if (g.periodic) {
    f -= floor(f / nc) * nc;
    int cidx = (int)f;
    return cidx >= nc ? nc - 1 : (cidx < 0 ? 0 : cidx);
  }
  int cidx = (int)floor(f);
  return cidx < 0 ? 0 : (cidx >= nc ? nc - 1 : cidx);
}

// wave-wide mask of lanes with a <= b (ordered), written straight to an SGPR pair by v_cmp
__device__ __forceinline__ unsigned long long wave_mask_le(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 5 /* FCMP_OLE */); }
__device__ __forceinline__ unsigned long long wave_mask_le(double a, double b) { return __builtin_amdgcn_fcmp(a, b, 5 /* FCMP_OLE */); }

// position folded into [0, box) (identity for box edge 0 = open boundary)
template <typename R>
__device__ __forceinline__ R wrap_into_box(R x, R box, R invbox) {
  return x - floor(x * invbox) * box;
}

// Device-side list bookkeeping of one replica: int flags[F_COUNT].
//   F_REBUILD0/1  rebuild requested in the step with parity 0/1.  The check of a step with parity p may only
//                 SET flags[p] and CLEAR flags[p^1]; every other kernel of that step only reads flags[p].
//   F_MAXN        largest neighbour count seen by a build (> capacity: a list was truncated)
//   F_NREBUILD    rebuild counter
//   F_VIOLATION   a rebuild was requested in a step whose rebuild chain the host had not enqueued (see
//                 ListCheck::skipped): the forces since then are invalid, the caller rewinds and repeats
enum { F_REBUILD0 = 0, F_REBUILD1 = 1, F_MAXN = 2, F_NREBUILD = 3, F_VIOLATION = 4, F_COUNT = 5 };

// Displacement test that drives the rebuilds: the list (cutoff + skin) is valid while no atom has moved
// further than skin/2 from `ref`; the test runs on the device (in the fused integrator kernel, or in
// check_displacement_kernel for plain evaluations) and every kernel of the rebuild chain returns at once unless
// the flag of its step is set, so the host never has to look.
template <typename R>
struct ListCheck {
  const R *ref;  // positions at the last list build, original atom order [3N]
  R hard2;       // (skin/2)^2
  const R *hs2;  // per-atom (half skin)^2, original atom order [N], or null: `hard2` for every atom
  int *flags;
  int parity;
  // Chain skipping (tmdhip_md_run on large systems).  The five launches of the rebuild chain return at once on
  // ~8 of 9 steps and still cost ~1.6 us each; the host leaves them out for a step when it knows that in the
  // step before no atom had used up more than `near_frac2` of its (squared) limit.  It learns that from host-
  // mapped memory: every atom beyond that fraction stores `seq` into *near_host, and the pair kernel of the
  // same step publishes `seq` as progress.  Should an atom nevertheless cross its limit in a step without a
  // chain (`skipped`), F_VIOLATION makes the caller rewind the batch and repeat it with every chain in place.
  unsigned *near_host;  // null: no reporting
  unsigned seq;
  R near_frac2;
  int skipped;
  int *ext;  // coordinate extent of everything ever stored into sorted_xyzq (see extent_note)
};

// Coordinate extent of the positions the pair kernels gather: int keys of {min x, y, z, max x, y, z} (float order
// = signed int order of the key).  The lean pair kernels fuse the minimum image as fma(-k, box, d), which equals
// the reference's separately rounded `d - box*round(d/box)` (forces.py:360-365) only while k*box is exact, i.e.
// |k| <= 2 (or a power of two): guaranteed while every coordinate difference is below 2.5 box edges.  The
// reference never wraps positions (integrator.py:61-64), so atoms may drift many boxes apart; every kernel that
// writes sorted_xyzq widens this extent, and a pair kernel that finds it beyond kExtentExactFrac box edges takes
// its loop copy with the product rounded separately.  The bounds only widen (reset: tmdhip_invalidate_list, a new
// box); after the first pass no lane is outside them and the cost is six compares per atom.
constexpr float kExtentExactFrac = 2.4f;
constexpr int kExtentEmpty[6] = {0x7F800000, 0x7F800000, 0x7F800000,                  // keys of +inf
                                 (int)0x807FFFFFu, (int)0x807FFFFFu, (int)0x807FFFFFu};  // keys of -inf
__device__ __forceinline__ int extent_key(float x) {
  const int i = __float_as_int(x);
  return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float extent_unkey(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }
template <typename R>
__device__ __forceinline__ void extent_note(int *ext, R x, R y, R z) {
  if (!ext) return;
  // (fp64 positions: the float cast moves a bound by half an ulp of fp32 at most, nothing against the 0.1-box slack)
  const int k[3] = {extent_key((float)x), extent_key((float)y), extent_key((float)z)};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (k[d] < ext[d]) atomicMin(&ext[d], k[d]);
    if (k[d] > ext[3 + d]) atomicMax(&ext[3 + d], k[d]);
  }
}
// true when some coordinate difference may reach 2.5 box edges (wave-uniform: scalar loads)
template <typename R>
__device__ __forceinline__ bool extent_needs_exact_image(const int *__restrict__ ext, const R *box) {
  if (!ext) return false;
  bool exact = false;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float span = extent_unkey(ext[3 + d]) - extent_unkey(ext[d]);  // -inf while nothing was noted
    exact = exact || ((float)box[d] > 0.f && span > kExtentExactFrac * (float)box[d]);
  }
  return exact;
}

// (rx, ry, rz) = position - reference position of one atom
template <typename R>
__device__ __forceinline__ void list_check_point(const ListCheck<R> &k, const PairConsts<R> &c, R rx, R ry, R rz,
                                                 R h2) {
  const R dx = min_image(rx, c.box[0], c.invbox[0]);
  const R dy = min_image(ry, c.box[1], c.invbox[1]);
  const R dz = min_image(rz, c.box[2], c.invbox[2]);
  const R d2 = dx * dx + dy * dy + dz * dz;
  if (!(d2 <= h2)) {  // NaN positions also force a rebuild
    k.flags[F_REBUILD0 + k.parity] = 1;
    if (k.skipped) k.flags[F_VIOLATION] = 1;
    if (k.near_host) k.near_host[2] = k.seq;  // "this step rebuilds": its successor needs no chain either
  }
  if (k.near_host && !(d2 <= h2 * k.near_frac2)) *k.near_host = k.seq;  // host-mapped: only the few fast atoms store
}
// squared displacement atom i may reach before the list has to be rebuilt
template <typename R>
__device__ __forceinline__ R list_check_limit(const ListCheck<R> &k, int i) {
  return k.hs2 ? k.hs2[i] : k.hard2;
}
template <typename R>
__device__ __forceinline__ void list_check_atom(const ListCheck<R> &k, const PairConsts<R> &c, int i, R px, R py, R pz) {
  list_check_point<R>(k, c, px - k.ref[3 * i + 0], py - k.ref[3 * i + 1], pz - k.ref[3 * i + 2], list_check_limit(k, i));
}

// thread 0 of the check of a step: the other parity's request is history
__device__ __forceinline__ void list_check_clear(int *flags, int parity) { flags[F_REBUILD0 + (parity ^ 1)] = 0; }

template <typename R>
__global__ void check_displacement_kernel(int n, const R *__restrict__ pos, ListCheck<R> k, PairConsts<R> c, int force,
                                          const int *__restrict__ inv, const R *__restrict__ qs,
                                          typename Vec<R>::T4 *__restrict__ sorted) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    list_check_clear(k.flags, k.parity);
    if (force) k.flags[F_REBUILD0 + k.parity] = 1;
  }
  if (i >= n || force) return;  // (forced: place_sorted_kernel writes the records and notes the extent)
  const R x = pos[3 * i + 0], y = pos[3 * i + 1], z = pos[3 * i + 2];
  list_check_atom<R>(k, c, i, x, y, z);
  extent_note<R>(k.ext, x, y, z);
  // callers of a plain evaluation hand in arbitrary new positions: refresh the cell-sorted copy the pair kernel
  // reads in the same pass (on a rebuild place_sorted_kernel rewrites it in the new order; the MD loop's
  // integrator kernel keeps the copy current itself)
  typename Vec<R>::T4 rec;  // one full 16/32-byte store (partial writes of a record are slower)
  rec.x = x;
  rec.y = y;
  rec.z = z;
  rec.w = qs[i];
  sorted[inv[i]] = rec;
}

template <typename R>
__global__ void bin_count_kernel(int n, const R *__restrict__ pos, Grid g, int *__restrict__ cell_of,
                                 int *__restrict__ slot, int *__restrict__ count, const int *flag) {
  if (*flag == 0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int cx = cell_coord(pos[3 * i + 0], g, 0);
  const int cy = cell_coord(pos[3 * i + 1], g, 1);
  const int cz = cell_coord(pos[3 * i + 2], g, 2);
  const int cidx = (cx * g.nc[1] + cy) * g.nc[2] + cz;
  cell_of[i] = cidx;
  slot[i] = atomicAdd(&count[cidx], 1);
}

// single block of 1024 threads; cell_start[ncell] = n afterwards; counts are zeroed for the next rebuild
__global__ __launch_bounds__(1024) void scan_cells_kernel(int ncell, int *__restrict__ count,
                                                          int *__restrict__ cell_start, const int *flag) {
  if (*flag == 0) return;
  __shared__ int wsum[16];
  __shared__ int carry;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < ncell; base += 1024) {
    const int idx = base + t;
    const int v = idx < ncell ? count[idx] : 0;
    int inc = v;  // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(inc, o, 64);
      if (lane >= o) inc += up;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int woff = carry;
    for (int k = 0; k < w; ++k) woff += wsum[k];
    if (idx < ncell) {
      cell_start[idx] = woff + inc - v;
      count[idx] = 0;
    }
    __syncthreads();
    if (t == 1023) carry = woff + inc;
    __syncthreads();
  }
  if (t == 0) cell_start[ncell] = carry;
}

__global__ void fill_cells_kernel(int n, const int *__restrict__ cell_of, const int *__restrict__ slot,
                                  const int *__restrict__ cell_start, int *__restrict__ order_tmp,
                                  const int *flag) {
  if (*flag == 0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  order_tmp[cell_start[cell_of[i]] + slot[i]] = i;
}

// deterministic order inside a cell (rank by original index) and, with the final position known, the
// cell-sorted copies the pair kernel reads: {x, y, z, q*sqrt(k)}, type, inverse permutation, and the
// reference positions of the displacement test
template <typename R>
struct PlaceArgs {
  const int *cell_of, *cell_start, *order_tmp;
  const R *pos, *qs;
  const int *types;
  int *order, *inv;
  typename Vec<R>::T4 *sorted;
  int *stype;
  R *ref;
  const R *half_skin;
  R *sorted_hs;
  const R *vel;
  R vs_floor, vs_time, vs_cap;
  R *hs2_dyn;
  int *ext;
};

// atom at position `a` of the unsorted cell order -> its final slot (rank by original index inside the cell) and every
// per-slot copy the pair kernels read
template <typename R>
__device__ __forceinline__ void place_atom(const PlaceArgs<R> &P, int a) {
  const int me = P.order_tmp[a];
  const int cidx = P.cell_of[me];
  const int s = P.cell_start[cidx], e = P.cell_start[cidx + 1];
  int rank = 0;
  for (int k = s; k < e; ++k) rank += P.order_tmp[k] < me;
  const int dst = s + rank;
  P.order[dst] = me;
  P.inv[me] = dst;
  typename Vec<R>::T4 v;
  v.x = P.pos[3 * me + 0];
  v.y = P.pos[3 * me + 1];
  v.z = P.pos[3 * me + 2];
  v.w = P.qs[me];
  P.sorted[dst] = v;
  extent_note<R>(P.ext, v.x, v.y, v.z);
  P.stype[dst] = P.types[me];
  if (P.half_skin) {
    // this list's half skin of the atom: its static share, or — inside an MD run, where the velocity is known —
    // a reduced floor plus the distance it covers in `vs_time` at its present speed, capped at vs_cap times the
    // largest static share (the cells are sized for that).  Any choice is safe: the displacement test uses the
    // same number (hs2_dyn); a good choice lets fast atoms go further before they force a rebuild while slow
    // ones keep short lists.
    R h = P.half_skin[me];
    if (P.vel) {
      const R vx = P.vel[3 * me + 0], vy = P.vel[3 * me + 1], vz = P.vel[3 * me + 2];
      h = min(P.vs_floor * h + P.vs_time * sqrt(vx * vx + vy * vy + vz * vz), P.vs_cap);
    }
    P.sorted_hs[dst] = h;
    if (P.hs2_dyn) P.hs2_dyn[me] = h * h;
  }
  P.ref[3 * me + 0] = v.x;
  P.ref[3 * me + 1] = v.y;
  P.ref[3 * me + 2] = v.z;
}

template <typename R>
__global__ void place_sorted_kernel(int n, PlaceArgs<R> P, const int *flag) {
  if (*flag == 0) return;
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  place_atom<R>(P, a);
}

// Binning of a small system in ONE launch of one block: count (LDS atomics), scan, fill and place — the work of
// bin_count / scan_cells / fill_cells / place_sorted.  On the ~10 of 11 steps without a rebuild the chain then costs
// two early-exit launches instead of five (~1.7 us each).  A rebuild on one CU is slower than the four parallel
// launches, which sets the size limit — measured, water boxes, us per MD step without / with: 5 184 atoms 27.2 / 23.8,
// 12 288 atoms 34.6 / 36.1, 41 472 atoms 41.8 / 67.1.
constexpr int kPrepSmallMaxCells = 4096;
constexpr int kPrepSmallMaxAtoms = 8192;
template <typename R>
__global__ __launch_bounds__(1024) void prep_small_kernel(int n, const R *__restrict__ pos, Grid g, int ncell,
                                                          int *__restrict__ cell_of, int *__restrict__ slot,
                                                          int *__restrict__ cell_start, int *__restrict__ order_tmp,
                                                          PlaceArgs<R> P, const int *flag) {
  if (*flag == 0) return;
  __shared__ int s_count[kPrepSmallMaxCells];
  __shared__ int s_start[kPrepSmallMaxCells + 1];
  __shared__ int wsum[16];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int c = t; c < ncell; c += 1024) s_count[c] = 0;
  __syncthreads();
  for (int i = t; i < n; i += 1024) {
    const int cx = cell_coord(pos[3 * i + 0], g, 0);
    const int cy = cell_coord(pos[3 * i + 1], g, 1);
    const int cz = cell_coord(pos[3 * i + 2], g, 2);
    const int cidx = (cx * g.nc[1] + cy) * g.nc[2] + cz;
    cell_of[i] = cidx;
    slot[i] = atomicAdd(&s_count[cidx], 1);
  }
  __syncthreads();
  // exclusive scan: thread t owns cells 4t .. 4t+3
  int v[4], mine = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = 4 * t + k;
    v[k] = c < ncell ? s_count[c] : 0;
    mine += v[k];
  }
  int inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int run = inc - mine;
  for (int k = 0; k < w; ++k) run += wsum[k];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = 4 * t + k;
    if (c < ncell) {
      s_start[c] = run;
      cell_start[c] = run;
    }
    run += v[k];
  }
  if (t == 0) {
    s_start[ncell] = n;
    cell_start[ncell] = n;
  }
  __syncthreads();
  for (int i = t; i < n; i += 1024) order_tmp[s_start[cell_of[i]] + slot[i]] = i;
  __threadfence_block();
  __syncthreads();  // order_tmp and cell_start are complete for the whole block
  for (int a = t; a < n; a += 1024) place_atom<R>(P, a);
}

// list entry = type_j << 27 | j << 4 (j = cell-sorted slot, 23 bits): `entry & kEntryOffMask` is the byte offset
// of atom j's float4 record, `entry >> 24` the byte offset of type j in an 8-byte-stride LDS table row (for
// n <= 2^20; larger systems mask it).  Contexts with more than kEntryTypes LJ classes leave the type field 0 (kernels read stype[j]).
constexpr unsigned kEntryOffMask = 0x07FFFFF0u;  // byte offset of atom j's float4 record
constexpr int kEntryTypes = 32;                  // LJ classes that fit the entry's type field
constexpr int kEntryTypeShift = 27;
constexpr float kR2Floor = 1.0e-2f;  // (0.1 A)^2: keeps 1/r^14 finite for the self entries that pad a column

struct ListGeom {
  int lpa;        // lanes per atom in the pair kernel (power of two, 1..64)
  int apw;        // atoms per wave = 64 / lpa
  int maxn;       // capacity per atom (multiple of lpa)
  int lpa_shift;  // log2(lpa)
};

__device__ __forceinline__ size_t list_slot(const ListGeom &lg, int a, int k) {
  const int apw_shift = 6 - lg.lpa_shift;
  const int g = a >> apw_shift, ain = a & (lg.apw - 1);
  const int kk = k >> lg.lpa_shift, l = (ain << lg.lpa_shift) + (k & (lg.lpa - 1));
  return ((size_t)g * lg.maxn << apw_shift) + ((size_t)(((kk >> 2) << 6) + l) << 2) + (kk & 3);
}

// ---- K2: Verlet list build ---------------------------------------------------------------------
// One wave per cell.  The candidates (all atoms of the (2m+1)^3 stencil cells, which are contiguous
// runs of the cell-sorted arrays) are streamed through the 64 lanes with coalesced loads; for every
// chunk of 64 candidates the wave loops over the atoms i of its cell (wave-uniform data), tests
// |d|^2 <= rlist^2 and the exclusions, and appends the hits of atom i with a ballot / prefix-popcount
// compaction.  Entry order per atom is fixed by the stencil order -> lists are bit-reproducible.
// WSKIN: per-atom skins — pair (i, j) is listed when |d| <= cutoff + s_i + s_j (s = the atom's half skin: the
// displacement it may reach before a rebuild, see ListCheck), instead of cutoff + skin for every pair.
template <typename R, bool LOOP, bool WSKIN>
__global__ __launch_bounds__(64) void build_list_kernel(
    int n, const typename Vec<R>::T4 *__restrict__ sorted, const R *__restrict__ sorted_hs,
    const int *__restrict__ stype,
    const int *__restrict__ order, const int *__restrict__ cell_start, Grid g, PairConsts<R> c, R rlist2, R rcut,
    const int *__restrict__ excl_off, const int *__restrict__ excl_idx, ListGeom lg,
    unsigned *__restrict__ nlist, int *__restrict__ nneigh, int *__restrict__ status, const int *flag,
    int ncell, int nactive, int type_in_entry, unsigned long long *dbg, int split) {
  if (*flag == 0) return;
  const unsigned long long dbg_t0 = dbg ? __builtin_readcyclecounter() : 0ull;  // TMDHIP_DEBUG_TIMELINE (tools/build_timeline.py)
  using R4 = typename Vec<R>::T4;
  // the whole list as a bounds-checked buffer (< 2^30 entries): an out-of-range store is dropped
  const __amdgpu_buffer_rsrc_t nrsrc = __builtin_amdgcn_make_buffer_rsrc(
      nlist, 0, (int)((((size_t)n + lg.apw - 1) / lg.apw) * (size_t)lg.maxn * lg.apw * 4u), 0x00020000);
#if TMD_EXP & (1 << 22)  // EXPERIMENT: 8 KB of LDS per block = 20 blocks per CU = 5 waves per SIMD resident, the rest dispatched as waves finish
  __shared__ int occ_pad[900];
  if (n < 0) occ_pad[threadIdx.x] = n, nlist[0] = occ_pad[(threadIdx.x + 1) & 63];
#elif TMD_EXP & (1 << 23)  // EXPERIMENT: 6.6 KB = 24 blocks per CU = 6 waves per SIMD
  __shared__ int occ_pad[500];
  if (n < 0) occ_pad[threadIdx.x] = n, nlist[0] = occ_pad[(threadIdx.x + 1) & 63];
#endif
  __shared__ int seg_start[128];
  __shared__ int seg_prefix[129];
  __shared__ int seg_code[128];  // periodic image of the stencil cell: 2 bits per axis, 0:-L 1:0 2:+L
  const int lane = threadIdx.x;
  int wmax = 0;
  unsigned long long dbg_work = 0;  // candidates x atoms over the block's cells (debug timeline only)
  // LOOP: the grid is capped and a block walks several cells (a launch that returns at once on the steps
  // without a rebuild still costs time proportional to its block count: the 343k cells of the 10^6-atom
  // LJ box = 100 us per step).  Systems with fewer cells keep one cell per block (no loop: faster code).
  // split > 1 (few cells: the one-wave-per-cell grid would leave most SIMDs idle — 729 cells on 1 024 SIMDs at 12 288
  // atoms, 109 us per build): `split` blocks share a cell, each builds the lists of its share of the cell's atoms
  // from the same candidates (never together with LOOP)
  int cell = LOOP ? (int)blockIdx.x : (int)blockIdx.x / split;
  const int part = LOOP ? 0 : (int)blockIdx.x % split;
  do {
  int cs = cell_start[cell], ce = cell_start[cell + 1];
  if (cell == 0 && part == 0 && lane == 0) status[1] += 1;  // flags[F_NREBUILD]
  if (!LOOP && split > 1) {  // this block's atoms of the cell (multiples of 4: whole batches)
    const int per = ((ce - cs + split - 1) / split + 3) & ~3;
    cs = min(cs + part * per, ce);
    ce = min(cs + per, ce);
  }
  if (cs == ce) continue;
  __syncthreads();  // LDS tables of the previous cell are no longer read
  const int cz = cell % g.nc[2], cy = (cell / g.nc[2]) % g.nc[1], cx = cell / (g.nc[2] * g.nc[1]);
  // stencil segments: a segment is a run of cells along z (contiguous in the cell-sorted arrays) of one
  // (x, y) stencil row, clipped to the cells that can hold an atom within rlist of this cell
  // (g.zreach) and split where it crosses the periodic boundary: <= 2 pieces per row, <= 98 segments.
  // Lane handles segments `lane` and `lane + 64` (segment = 2 * row + piece).
  const int w = 2 * g.m + 1, nrows = w * w;  // <= 49
  int cnt2[2], st2[2], code2[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int sidx = lane + 64 * h;
    const int row = sidx >> 1, piece = sidx & 1;
    int count = 0, start = 0, code = 1 | (1 << 2) | (1 << 4);
    if (row < nrows) {
      const int ox = row / w - g.m, oy = row % w - g.m;
      const int zr = g.zreach[ox + g.m][oy + g.m];  // -1: no cell of this row is in reach
      int x = cx + ox, y = cy + oy;
      bool ok = zr >= 0;
      int codexy = 1 | (1 << 2);
      if (g.periodic) {
        codexy = (x < 0 ? 0 : (x >= g.nc[0] ? 2 : 1)) | ((y < 0 ? 0 : (y >= g.nc[1] ? 2 : 1)) << 2);
        x = (x + g.nc[0]) % g.nc[0];
        y = (y + g.nc[1]) % g.nc[1];
      } else {
        ok = ok && x >= 0 && x < g.nc[0] && y >= 0 && y < g.nc[1];
      }
      if (ok) {
        const int zlo = cz - zr, zhi = cz + zr, nz = g.nc[2];
        // piece 0: the part inside [0, nz); piece 1: the part that wraps (below 0 or beyond nz-1)
        int a = max(zlo, 0), b = min(zhi, nz - 1), zc = 1;
        if (piece == 1) {
          if (!g.periodic) {
            a = 1, b = 0;
          } else if (zlo < 0) {
            a = zlo + nz, b = nz - 1, zc = 0;
          } else if (zhi >= nz) {
            a = 0, b = zhi - nz, zc = 2;
          } else {
            a = 1, b = 0;
          }
        }
        if (a <= b) {
          const int base = (x * g.nc[1] + y) * nz;
          start = cell_start[base + a];
          count = cell_start[base + b + 1] - start;
          code = codexy | (zc << 4);
        }
      }
    }
    cnt2[h] = count;
    st2[h] = start;
    code2[h] = code;
  }
  // exclusive prefix over the 128 slots
  int inc0 = cnt2[0], inc1 = cnt2[1];
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t0 = __shfl_up(inc0, o, 64), t1 = __shfl_up(inc1, o, 64);
    if (lane >= o) {
      inc0 += t0;
      inc1 += t1;
    }
  }
  const int tot0 = __shfl(inc0, 63, 64);
  const int ncand = tot0 + __shfl(inc1, 63, 64);
  seg_start[lane] = st2[0];
  seg_start[lane + 64] = st2[1];
  seg_code[lane] = code2[0];
  seg_code[lane + 64] = code2[1];
  seg_prefix[lane] = inc0 - cnt2[0];
  seg_prefix[lane + 64] = tot0 + inc1 - cnt2[1];
  if (lane == 0) seg_prefix[128] = ncand;
  if (dbg) dbg_work += (unsigned long long)ncand * (unsigned long long)(ce - cs);
  __syncthreads();

  // per-atom data of the i block staged in LDS as two 16-byte records that the inner loop reads with
  // wave-uniform (broadcast) LDS loads: rec0 = {wrapped xyz, cutoff + own half skin (WSKIN)},
  // rec1 = {own original index, first two excluded partners, word offset of the atom's list row} (so
  // "j == i" is just one more exclusion; longer exclusion rows spill to global reads).  PMC showed this loop limited by the
  // scalar unit (one SALU per CU, shared by the 4 SIMDs) as much as by VALU, hence: LDS addresses and
  // the row offset live in VGPRs, the per-atom hit counters move with readlane/writelane, and the
  // rare long-exclusion path is hoisted out as a separate loop version.
  constexpr int EXS = 3;
  __shared__ R4 s_rec0[64];
  __shared__ int4 s_rec1[64];
  __shared__ int s_eb[64], s_more[64];
  __shared__ unsigned s_bm[64];
  __shared__ __align__(16) int s_cnt[64];
  const int apw_shift = 6 - lg.lpa_shift;
  const unsigned kmask = (unsigned)lg.lpa - 1u;
  // entry k of a row sits at byte ((k / (4 LPA)) << 10) + ((k % LPA) << 4) + (((k / LPA) % 4) << 2)  (list_slot); the
  // masks live in VGPRs (an SGPR operand halves the VALU rate)
  unsigned vmask_hi, vmask_lo;
  asm("v_mov_b32 %0, %1" : "=v"(vmask_hi) : "s"(~((4u << lg.lpa_shift) - 1u)));
  asm("v_mov_b32 %0, %1" : "=v"(vmask_lo) : "s"(kmask));
  unsigned vmaxn1;
  asm("v_mov_b32 %0, %1" : "=v"(vmaxn1) : "s"((unsigned)lg.maxn - 1u));
  const unsigned sh_hi = 8u - (unsigned)lg.lpa_shift;  // (k / (4 LPA)) << 10 == (k & ~(4 LPA - 1)) << (10 - 2 - lpa_shift)
  for (int ib = cs; ib < ce; ib += 64) {  // blocks of up to 64 atoms i of this cell (usually one)
    const int iend = min(ib + 64, ce);
    const int ni = iend - ib;
    __syncthreads();
    int long_rows = 0;
    if (lane < ni) {
      const int a = ib + lane;
      R4 p = sorted[a];
      p.x = wrap_into_box(p.x, c.box[0], c.invbox[0]);
      p.y = wrap_into_box(p.y, c.box[1], c.invbox[1]);
      p.z = wrap_into_box(p.z, c.box[2], c.invbox[2]);
      const unsigned rowoff = (((unsigned)(a >> apw_shift) * (unsigned)lg.maxn) << apw_shift) +
                              ((unsigned)(a & (lg.apw - 1)) << (lg.lpa_shift + 2));
      p.w = WSKIN ? rcut + sorted_hs[a] : R(0);
      const int oi = order[a];
      // passive atoms (original index >= nactive: halo images of a domain) get no list: parked out of reach
      if (oi >= nactive) p.x = (R)-1e18;
      const int eb = excl_off[oi], ne = excl_off[oi + 1] - eb;
      s_rec0[lane] = p;
      s_eb[lane] = eb + (EXS - 1);
      s_more[lane] = max(ne - (EXS - 1), 0);
      int4 ex;
      ex.x = oi;
      ex.y = 0 < ne ? excl_idx[eb + 0] : -1;
      ex.z = 1 < ne ? excl_idx[eb + 1] : -1;
      ex.w = (int)(rowoff * 4u);  // byte offset of the atom's list row
      s_rec1[lane] = ex;
      long_rows = ne > EXS - 1;
    } else {
      // dummy atoms that pad the last batch of four: parked out of reach (never a hit, never a store)
      R4 p;
      p.x = (R)-1e18;
      p.y = p.z = p.w = R(0);
      s_rec0[lane] = p;
      s_rec1[lane] = make_int4(-1, -1, -1, 0);
    }
    const bool any_long = __ballot(long_rows) != 0ull;
    s_bm[lane] = 0u;
    __syncthreads();
    s_cnt[lane] = 0;  // neighbour count of atom ib + lane (lives in LDS: one broadcast read + one
                      // same-value write per iteration instead of cross-lane register traffic)
    // Bitmap (2 048 bits, key = original index mod 2048) of everything some atom of this block excludes — itself and
    // its first two excluded partners.  A candidate whose bit is clear is excluded by nobody here: the three index
    // compares per (atom, chunk) are only made for batches that meet a flagged candidate (for water: the chunks of the
    // own and the adjacent cells, ~15 %).
    if (lane < ni) {
      const int4 ex = s_rec1[lane];
      atomicOr(&s_bm[((unsigned)ex.x & 2047u) >> 5], 1u << ((unsigned)ex.x & 31u));
      if (ex.y >= 0) atomicOr(&s_bm[((unsigned)ex.y & 2047u) >> 5], 1u << ((unsigned)ex.y & 31u));
      if (ex.z >= 0) atomicOr(&s_bm[((unsigned)ex.z & 2047u) >> 5], 1u << ((unsigned)ex.z & 31u));
    }
    __syncthreads();
    int seg = 0;      // segment of this lane's candidate; q grows by 64 per chunk so it only moves forward
    // candidate stream, software-pipelined: the three global loads of chunk q0 + 64 are issued before
    // chunk q0 is processed, so their latency overlaps the i loop instead of stalling the wave at the
    // top of every chunk (the build is latency-bound: PMC showed VALU busy 57 %)
    R4 nx_p;
    R nx_hs = 0;
    int nx_j = cs, nx_code = 0, nx_order = 0, nx_type = 0;
    bool nx_valid = false;
    auto fetch = [&](int q0) {
      const int q = q0 + lane;
      nx_valid = q < ncand;
      nx_j = cs;
      if (nx_valid) {  // last s with seg_prefix[s] <= q
        while (seg_prefix[seg + 1] <= q) ++seg;
        nx_j = seg_start[seg] + (q - seg_prefix[seg]);
      }
      nx_code = seg_code[seg];
      nx_p = sorted[nx_j];
      nx_order = order[nx_j];
      nx_type = stype[nx_j];
      if constexpr (WSKIN) nx_hs = sorted_hs[nx_j];
    };
    fetch(0);
    for (int q0 = 0; q0 < ncand; q0 += 64) {
      R4 pj = nx_p;
      const R sj = nx_hs;
      const int j = nx_j, code = nx_code;
      const bool valid = nx_valid;
      const unsigned oj = (unsigned)nx_order;
      const unsigned entry = ((unsigned)j << 4) | (type_in_entry ? (unsigned)nx_type << kEntryTypeShift : 0u);
      if (q0 + 64 < ncand) fetch(q0 + 64);
      // candidate position as the periodic image that lies next to this cell: the i loop then needs
      // no minimum-image arithmetic (the list criterion has the skin as slack, so it need not reproduce
      // the reference's rounding; the pair kernel's cutoff test does).  Lanes past the end of the
      // candidate list are parked far away so that they can never hit.
      pj.x = wrap_into_box(pj.x, c.box[0], c.invbox[0]) + (R)((code & 3) - 1) * c.box[0];
      pj.y = wrap_into_box(pj.y, c.box[1], c.invbox[1]) + (R)(((code >> 2) & 3) - 1) * c.box[1];
      pj.z = wrap_into_box(pj.z, c.box[2], c.invbox[2]) + (R)(((code >> 4) & 3) - 1) * c.box[2];
      if (!valid) pj.x = (R)1e18;
      // exclusions, compaction and store of the hits of atom t (mask = lanes whose candidate is in range)
      auto handle = [&](int t, unsigned roff, const R4 &pi, unsigned long long mask) {
        const int4 ex = *reinterpret_cast<const int4 *>(reinterpret_cast<const char *>(s_rec1) + roff);
        const int base = s_cnt[t];
        // wave-wide masks (SGPR pairs) instead of per-lane booleans: the compares write the masks
        // directly, they are combined on the scalar unit, and the prefix count is two v_mbcnt
        mask &= ~(__builtin_amdgcn_uicmp((unsigned)ex.x, oj, 32 /* eq */) |
                  __builtin_amdgcn_uicmp((unsigned)ex.y, oj, 32) |
                  __builtin_amdgcn_uicmp((unsigned)ex.z, oj, 32));
        if (any_long) {  // wave-uniform, rare (atoms with more than EXS-1 exclusions: proteins)
          const int more = s_more[t], eb = s_eb[t];
          for (int e = 0; e < more; ++e) mask &= ~__builtin_amdgcn_uicmp((unsigned)excl_idx[eb + e], oj, 32);
        }
        const unsigned k = (unsigned)base + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                      __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
        if (__builtin_amdgcn_inverse_ballot_w64(mask) && (k < (unsigned)lg.maxn)) {
          const unsigned rowoff = (unsigned)ex.w >> 2;
          const unsigned kk = k >> lg.lpa_shift;
          nlist[rowoff + ((kk >> 2) << 8) + ((k & kmask) << 2) + (kk & 3u)] = entry;
        }
        s_cnt[t] = base + (int)__popcll(mask);  // every lane writes the same value
      };
      auto rec0 = [&](unsigned roff) -> R4 {
        return *reinterpret_cast<const R4 *>(reinterpret_cast<const char *>(s_rec0) + roff * (unsigned)(sizeof(R4) / 16));
      };
      auto in_range = [&](const R4 &pi) -> unsigned long long {
        const R dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
        if constexpr (WSKIN) {
          const R reach = pi.w + sj;  // cutoff + s_i + s_j
          return wave_mask_le(dx * dx + dy * dy + dz * dz, reach * reach);
        }
        return wave_mask_le(dx * dx + dy * dy + dz * dz, rlist2);
      };
      // LDS byte offset of the current i record, kept in a VGPR on purpose (see above).  Two-level test:
      // the distance masks of four atoms are computed together (independent LDS reads and arithmetic
      // chains), the expensive part only runs for (atom, chunk) combinations with at least one hit
      // (a chunk is ~one z-column of the stencil, so for a given atom many chunks are out of reach).
      unsigned recoff;
      asm volatile("v_mov_b32 %0, 0" : "=v"(recoff));
      int t = 0;
      // Batches of four atoms with ONE branch (any hit at all?) and none inside: a taken scalar branch costs more than
      // the arithmetic it skips (the per-block timeline gives ~150 cycles per (atom, chunk) combination against ~70
      // of VALU work).  The four hit counters travel as one 16-byte LDS word each way, and a lane without a hit
      // stores to an out-of-range offset of a bounds-checked buffer (dropped by the hardware) instead of leaving
      // exec.  Atoms with long exclusion rows (proteins) keep the branching path.
      // The last batch is padded with parked dummy atoms (staged above), so there is no scalar remainder loop.
      if (!any_long) {
        // candidates that somebody in this block excludes (see s_bm); lanes past the end never hit anyway
        const unsigned bmw = s_bm[(oj & 2047u) >> 5];
        const unsigned long long special = __builtin_amdgcn_uicmp((bmw >> (oj & 31u)) & 1u, 0u, 33 /* ne */);
        for (; t < ni; t += 4, recoff += 64u) {
          const R4 p0 = rec0(recoff), p1 = rec0(recoff + 16u), p2 = rec0(recoff + 32u), p3 = rec0(recoff + 48u);
          unsigned long long m[4] = {in_range(p0), in_range(p1), in_range(p2), in_range(p3)};
          const unsigned long long any = m[0] | m[1] | m[2] | m[3];
          if (!any) continue;
          const int4 base4 = *reinterpret_cast<const int4 *>(&s_cnt[t]);
          const int base[4] = {base4.x, base4.y, base4.z, base4.w};
          int4 ex[4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            ex[u] = *reinterpret_cast<const int4 *>(reinterpret_cast<const char *>(s_rec1) + recoff + 16u * u);
          if (any & special) {  // rare: a flagged candidate is in range of one of the four
#pragma unroll
            for (int u = 0; u < 4; ++u)
              m[u] &= ~(__builtin_amdgcn_uicmp((unsigned)ex[u].x, oj, 32 /* eq */) | __builtin_amdgcn_uicmp((unsigned)ex[u].y, oj, 32) |
                        __builtin_amdgcn_uicmp((unsigned)ex[u].z, oj, 32));
          }
          int cnt[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            // slot of this lane's hit, clamped to the row's last one: a row that overflows is reported through F_MAXN
            // and its list thrown away (the caller grows the capacity and rebuilds), so what lands there is never used
            const unsigned k = min((unsigned)base[u] + __builtin_amdgcn_mbcnt_hi((unsigned)(m[u] >> 32),
                                                                                __builtin_amdgcn_mbcnt_lo((unsigned)m[u], 0u)),
                                   vmaxn1);
            // byte offset of entry k in the row: iteration kk = k / LPA, lane part k % LPA (list_slot's layout)
            unsigned posb = (unsigned)ex[u].w + ((k & vmask_hi) << sh_hi);
            posb += (k & vmask_lo) << 4;
            posb += __builtin_amdgcn_ubfe(k, (unsigned)lg.lpa_shift, 2u) << 2;
            // only the lanes with a hit store: exec = the hit mask for the one instruction (every lane of the block is
            // active here); a v_cndmask on an out-of-range offset would cost a half-rate VALU slot instead
            asm volatile("s_mov_b64 exec, %2\n\tbuffer_store_dword %0, %1, %3, 0 offen\n\ts_mov_b64 exec, -1"
                         :: "v"(entry), "v"(posb), "s"(m[u]), "s"(nrsrc) : "memory");
            cnt[u] = base[u] + (int)__popcll(m[u]);
          }
          *reinterpret_cast<int4 *>(&s_cnt[t]) = make_int4(cnt[0], cnt[1], cnt[2], cnt[3]);  // every lane writes the same values
        }
      }
      for (; t + 4 <= ni; t += 4, recoff += 64u) {  // (cells with long exclusion rows: the branching path)
        const R4 p0 = rec0(recoff), p1 = rec0(recoff + 16u), p2 = rec0(recoff + 32u), p3 = rec0(recoff + 48u);
        const unsigned long long m0 = in_range(p0), m1 = in_range(p1), m2 = in_range(p2), m3 = in_range(p3);
        if (m0) handle(t, recoff, p0, m0);
        if (m1) handle(t + 1, recoff + 16u, p1, m1);
        if (m2) handle(t + 2, recoff + 32u, p2, m2);
        if (m3) handle(t + 3, recoff + 48u, p3, m3);
      }
      for (; t < ni; ++t, recoff += 16u) {
        const R4 p0 = rec0(recoff);
        const unsigned long long m0 = in_range(p0);
        if (m0) handle(t, recoff, p0, m0);
      }
    }
    __syncthreads();
    const int mycnt = s_cnt[lane];
    if (lane < ni) nneigh[ib + lane] = min(mycnt, lg.maxn);
    wmax = max(wmax, lane < ni ? mycnt : 0);
  }
  } while (LOOP && (cell += gridDim.x) < ncell);
  if (dbg && lane == 0) {  // per block: entry / exit cycle counters, XCC id, candidates x atoms of its (last) cell
    unsigned long long *o = dbg + 4 * (size_t)blockIdx.x;
    o[0] = dbg_t0;
    o[1] = __builtin_readcyclecounter();
    // XCC id | HW_ID (wave 0-3, SIMD 4-5, pipe 6-7, CU 8-11, SH 12, SE 13-15) << 8; longest list | work << 32
    o[2] = (unsigned long long)__builtin_amdgcn_s_getreg((6 << 11) | 20) |
           ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 8);
    o[3] = (unsigned long long)wmax | (dbg_work << 32);
  }
  // flags[2] = largest neighbour count ever seen; > maxn means a list was truncated (overflow)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o, 64));
  if (lane == 0 && wmax > 0) atomicMax(status, wmax);
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

// ---- K3f: lean fp32 specialisation of the list pair kernel ---------------------------------------
// Issue-rate measurements on gfx950 (tools/ubench/valu_rates.hip, 8 waves per SIMD): a plain fp32 / integer
// VALU op (v_fma_f32, v_mul, v_add, v_and, v_mov, v_cndmask) retires in ~2.3 cycles per wave, a PACKED op
// (v_pk_fma/mul/add_f32) in ~4.3 — packing two entries into one instruction buys no ALU throughput on this
// chip — 32-bit shifts and v_mul_u32_u24 run at half rate (4.2), v_cmp costs 5.3, v_rsq/v_rcp 8.2.  The first
// version of this kernel evaluated entries two at a time on float2 vectors: 60 packed ops + 47 v_mov
// (transposes of {pj[u].x, pj[u+1].x} into register pairs) per 4 entries = ~155 cycles per entry.  This
// version is plain scalar code on the natural float4 record: ~32 full-rate ops + 1 v_cmp + 1 v_rsq per
// entry (~85 cycles), no transposes, no shifts:
//   entry = type << 27 | j << 4      -> gather offset = entry & 0x07FFFFF0 (one v_and), LDS table address
//                                       = (type_i << 8) | entry >> 24 (one SDWA v_or; table rows of 32 x 8 B)
//   minimum image by the magic-number trick (3 ops per component, bit-exact, see min_image_magic)
//   force scale factored as  rinv2 * ((a12 rinv6 + b6) rinv6 - qq rinv) + qq 2 krf   (9 ops)
// Same decision arithmetic (bit-exact) as pair_math.h.  Terms: LJ and/or electrostatics (plain Coulomb or
// reaction field), optionally the LJ switching function (SWITCH) and the per-term energies (ENERGY);
// repulsion terms, fp64, more than 32 LJ classes and pair counting take list_pair_kernel.
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// k = round-half-even(d / box) by the magic-number trick: fma(d, 1/box, 1.5*2^23) - 1.5*2^23 is exact
// round-to-nearest-even for |d/box| < 2^22 (v_rndne_f32 would be a fourth instruction).  It differs from
// rndne(fl(d*invbox)) only when d/box lies within one rounding error of a half-integer, i.e. the wrapped |d| ~
// box/2 >= cutoff, where the pair is rejected either way (same argument as for d*invbox vs d/box in pair_math.h).
// EXACT = false fuses the product into the subtraction: identical to the reference's separately rounded
// `d - box*round(d/box)` (forces.py:360-365) whenever k*box is representable — |k| <= 2 — which the kernel
// establishes from the coordinate extent (extent_needs_exact_image); EXACT = true rounds the product first
// (one more instruction per component) and holds for any image offset.
template <bool EXACT>
__device__ __forceinline__ float min_image_magic(float d, float box, float invbox) {
#pragma clang fp contract(off)
  const float magic = 12582912.0f;
  const float t = __builtin_fmaf(d, invbox, magic);
  const float k = t - magic;
  if (EXACT) {
    const float p = box * k;
    return d - p;
  }
  return __builtin_fmaf(-k, box, d);
}

using exact_image = std::integral_constant<bool, true>;
using fused_image = std::integral_constant<bool, false>;

// kernel experiments (A/B builds: python -m torchmd_amd._build -DTMD_EXP=<bits> --out=...; see tools/ab_pair.py)
#ifndef TMD_EXP
#define TMD_EXP 0
#endif

// What the loop below is shaped by (gfx950, tools/ubench/valu_detail.hip + body_bisect.hip, 6 waves per SIMD; cycles
// per wave-instruction per SIMD at 2.4 GHz):
//   plain fp32 / integer VALU with VGPR, inline-constant or 32-bit-literal operands     2.1 - 2.35
//   ANY SGPR operand (VOP2 src0, VOP3 src0/src2: v_fma/v_mul/v_sub/v_fmac)              4.05   <- half rate
//   v_cmp (VCC or SGPR pair) 4.1, v_cndmask with an SGPR/VCC mask 4.1, SDWA forms 4.1, v_mov_b64 4.1,
//   VOP3-only integer ops (v_perm, v_bfe, v_alignbit, v_and_or, v_lshl_or) 4.1
//   v_rsq_f32: 8.1 back to back, ~10 in bursts of four, ~19 when it stands alone among plain instructions
//   VGPR bank conflicts: none measurable (only three sources in ONE bank cost 4.1)
// The compiler keeps every uniform value (box, 1/box, r2max) in SGPRs — 24 of the 36 v_fma of a 4-entry group read
// one — rotates the prefetched list words with v_mov_b64, advances the list pointer with a 64-bit VALU add and puts
// the list load IN FRONT of the gathers, where every wait for a gather (vmcnt retires in order) also waits for the
// list stream from the Infinity Cache.  Hence: loop constants laundered into VGPRs; the cutoff test as arithmetic
// (v_fma with clamp + v_mul) where no energy is wanted; the four v_rsq of a group issued back to back; list words
// through a raw buffer with a SCALAR running offset, requested behind the gathers issued in the same breath; and the
// unchecked groups software-pipelined over two register sets (gathers of group g+1 in flight while g is evaluated).
#if (TMD_EXP & 4) && (TMD_EXP & 32)
#define TMD_FAST_WAVES 8
#elif TMD_EXP & 4
#define TMD_FAST_WAVES 6
#elif TMD_EXP & 512
#define TMD_FAST_WAVES 4
#elif TMD_EXP & 16  // (bit 64: the pipeline in half-word stages, see below: 72 registers at 7 waves, 64 + spills at 8)
#define TMD_FAST_WAVES 7
#elif TMD_EXP & 32
#define TMD_FAST_WAVES 8
#else
#define TMD_FAST_WAVES 5
#endif
#if TMD_EXP & 1024
#define TMD_FAST_THREADS 768
#elif TMD_EXP & 2048
#define TMD_FAST_THREADS 512
#else
#define TMD_FAST_THREADS 256
#endif
// ---- the MD step inside the pair launch (FUSED variants; tmdhip_md_run, interior steps) -----------------------
// Between two force evaluations an MD step is per-atom work on the force just computed: second half kick of step
// `it` (+ thermostat), first half kick and drift of step it+1, the displacement test, the new record of the
// cell-sorted copy.  As a kernel of its own that is 8.8 us at C3 (22 us at 10^6 atoms): a chain of memory round
// trips (order -> bonded records -> partner positions -> update) with one wave per SIMD and nothing to hide it behind.
// A FUSED launch appends "step blocks" to the grid.  Workgroups are dispatched in order, so a step block starts when
// every pair block has been dispatched — in the slots the launch's last, partial round of pair blocks leaves idle —
// and does everything that does not need the new forces (bonded records of its 64 atoms, noise, loads) while the
// last pair blocks are still gathering; then every lane waits for the force record of ITS atom (the pair wave stores
// {force, launch number} as one 16-byte word) and updates.  Behind the last pair block only one load-update-store
// round remains, and the stored force array, its reload and one launch per step go away.
// Pair blocks never wait for anything, so the wait cannot deadlock; it is bounded all the same.
// Other blocks still read the positions of this launch, so the new ones go to the OTHER position buffer and the
// OTHER cell-sorted copy (the host swaps the two after every fused launch).  Same device functions in the same
// order as md_step_bonded_kernel / md_step_kernel: trajectories are bit-identical to the separate kernels.
struct FusedStatic;  // what does not change from launch to launch (device memory; defined with the MD-step kernels)
struct FusedStep {   // what does (kernel argument)
  const float *pos_in;  // positions of this launch's forces, original atom order (partners of the bonded terms)
  float *pos_out;       // drifted positions
  float4 *sorted_out;   // their cell-sorted records
  float4 *fsort;        // {pair force, launch number} per atom, cell-sorted order (pair blocks write, step blocks watch)
  unsigned gen;         // number of this launch (never 0)
  int bonded;           // FusedStatic::has_bonded (0 none, 1 inline records, 2 from FusedStatic::fbond)
  int nstep_blocks;     // step blocks at the end of the grid (a multiple of 8, like the pair blocks)
  uint64_t noise_step;
  unsigned *near_host;  // chain skipping: report words of the NEXT step's displacement test (null: none)
  unsigned seq;
  int parity;           // of the next step
};
template <bool LANGEVIN, int APB>
__device__ __forceinline__ void fused_step_blocks(const FusedStatic *__restrict__ fst, const FusedStep &fs,
                                                  const PairConsts<float> &c, int n, const float4 *__restrict__ sorted,
                                                  const int *__restrict__ order, int j, int npair, float *s_lds);

#if TMD_EXP & (1 << 21)
#define TMD_STEP_POLL_SLEEP 16
#else
#define TMD_STEP_POLL_SLEEP 4
#endif
constexpr int kAuxDeviceScope = 16;  // sc1 of a gfx942/950 buffer access: coherent across the XCDs' L2s
// lmode bits (list bookkeeping duties of the launch's first thread)
constexpr int kLmViolation = 1;  // the chain of this step was left out and its displacement test ran in the previous
                                 // launch's epilogue, which could not know that: a rebuild request found now = F_VIOLATION
constexpr int kLmParity = 2;     // parity of this step

template <int LPA, bool LJ, bool ELEC, bool ENERGY, bool SWITCH, int FUSED = 0>
// (LJ-only systems — liquid argon, short lists of ~90 entries — run the plain loop at one wave more per SIMD: 10^6 atoms
// 175.5 -> 168.5 us/step; with charges the pipelined loop at 5 waves wins, section 6c)
__global__ __launch_bounds__(TMD_FAST_THREADS, (!ELEC && TMD_FAST_WAVES == 5) ? 6 : TMD_FAST_WAVES) void list_pair_fast_f32_kernel(
    int n, const float4 *__restrict__ sorted, const int *__restrict__ stype, const int *__restrict__ order,
    int ntypes, const float2 *__restrict__ tab, const unsigned *__restrict__ nlist,
    const int *__restrict__ nneigh, int maxn, PairConsts<float> c, float *__restrict__ forces, int overwrite,
    double *__restrict__ energies, unsigned *publish, unsigned publish_value, const int *__restrict__ ext,
    int *lflags, int lmode, const FusedStatic *__restrict__ fst, FusedStep fstep) {
  constexpr int APW = 64 / LPA;
  constexpr int UNROLL = 4;
  static_assert(!(FUSED && ENERGY), "the fused step is for interior steps");
  static_assert(!FUSED || TMD_FAST_THREADS == 256, "step blocks are four waves");
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // tells the host (host-mapped word) that everything enqueued before this launch has completed
    if (publish) __hip_atomic_store(publish, publish_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (lflags) {
      const int parity = (lmode & kLmParity) ? 1 : 0;
      if ((lmode & kLmViolation) && lflags[F_REBUILD0 + parity] != 0) lflags[F_VIOLATION] = 1;
      // the epilogue's test (parity ^ 1) is the next step's: this step's request is history (list_check_clear)
      if (FUSED) lflags[F_REBUILD0 + parity] = 0;
    }
  }
  __shared__ __align__(16) float2 stab[kEntryTypes * kEntryTypes];  // row of type i: 32 x {-12 A, 6 B}
  const int lane = threadIdx.x & 63;
  // pair blocks of the launch (FUSED: step blocks follow them)
  const unsigned npair = FUSED ? gridDim.x - (unsigned)fstep.nstep_blocks : gridDim.x;
  if (FUSED && blockIdx.x >= npair) {
    fused_step_blocks<FUSED == 2, TMD_FAST_THREADS / LPA>(fst, fstep, c, n, sorted, order, (int)(blockIdx.x - npair),
                                                          (int)npair, reinterpret_cast<float *>(stab));
    return;
  }
  // XCD-aware block order: consecutive block ids go to the 8 XCDs round-robin, so block b works on
  // chunk (b % 8) * npair/8 + b / 8 — every XCD (own L2) gets a contiguous eighth of the cell-sorted
  // atoms and gathers neighbours from that region only.  npair is a multiple of 8; the surplus
  // blocks of the last eighths have nothing to do.
  const int blk = (int)((blockIdx.x & 7u) * (npair >> 3) + (blockIdx.x >> 3));
  if (blk * (int)(blockDim.x >> 6) * APW >= n) return;  // (block-uniform: nobody is left waiting at the barrier below)
#if TMD_EXP & (1 << 20)  // EXPERIMENT: pair waves issue ahead of step-block waves
  if (FUSED) __builtin_amdgcn_s_setprio(2);
#endif
  const int wave = __builtin_amdgcn_readfirstlane(blk * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6));
#if TMD_EXP & 512  // DEBUG (wrong results, timing only): i-pairs — a lane group evaluates the list of atom 2p for atoms 2p, 2p+1
  const int a = (wave * APW + lane / LPA) * 2;
  if (wave * APW * 2 >= n) return;
#else
  const int a = wave * APW + lane / LPA;
#endif
  const int sub = lane % LPA;
  const bool active = a < n;

  // ---- prologue: every load a wave needs before its first gather is requested HERE, in one batch, and only then
  // is the LJ table staged (a wave's life used to begin with three dependent memory round trips — table, then
  // atom record / list length, then the first list word — 5 500 of its ~40 000 cycles)
  // list words of this wave: group G (iterations 4G .. 4G+3 of all 64 lanes) is the 1 KB at byte G * 1024; rows are
  // padded, and reads past the buffer's end return 0
#if TMD_EXP & 512
  const unsigned *wrow = nlist + (size_t)(2 * wave) * maxn * APW;
  const __amdgpu_buffer_rsrc_t lrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(wrow), 0, 2 * maxn * APW * 4 + 4096, 0x00020000);
  const unsigned lvoff = (unsigned)((lane / LPA) / (APW / 2)) * (unsigned)(maxn * APW * 4) +
                         (unsigned)((((lane / LPA) * 2) % APW) * LPA + sub) * 16u;
#else
  const unsigned *wrow = nlist + (size_t)wave * maxn * APW;
  const __amdgpu_buffer_rsrc_t lrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(wrow), 0, maxn * APW * 4 + 4096, 0x00020000);
  const unsigned lvoff = (unsigned)lane * 16u;
#endif
#if TMD_EXP & 8192  // DEBUG (wrong results, timing only): the list stream from a 4 KB window per wave (cache resident)
  auto list_word = [&](int g) { return __builtin_amdgcn_raw_buffer_load_b128(lrsrc, lvoff, (g & 3) * 1024, 0); };
#elif TMD_EXP & 262144  // EXPERIMENT: list stream with the nt (slc) hint, so that it does not displace sorted_xyzq from L2
  auto list_word = [&](int g) { return __builtin_amdgcn_raw_buffer_load_b128(lrsrc, lvoff, g * 1024, 2); };
#else
  auto list_word = [&](int g) { return __builtin_amdgcn_raw_buffer_load_b128(lrsrc, lvoff, g * 1024, 0); };
#endif
  v4u word = list_word(0);  // list word of the next group to be gathered (in flight)
  float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
  int nn = 0, oi = 0;
  unsigned trow = 0;  // byte offset of this atom's row of the LDS table
  if (active) {
    pi = sorted[a];
    nn = nneigh[a];
    trow = (unsigned)stype[a] << 8;
    oi = order[a];
  }
#if TMD_EXP & 512
  float4 pi2 = active ? sorted[a + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned trow2 = active ? (unsigned)stype[a + 1] << 8 : 0u;
  const int oi2 = active ? order[a + 1] : 0;
  float gx = 0.f, gy = 0.f, gz = 0.f;
#endif
  // (only the rows of existing classes are ever read: ntypes x 32 entries instead of 32 x 32 — at 10^6 LJ atoms
  // with 64 atoms per block the full table was 15 625 x 8 KB of staging)
  for (int t = threadIdx.x; t < ntypes * kEntryTypes; t += blockDim.x) {
    const int ti = t >> 5, tj = t & 31;
    float2 ab = make_float2(0.f, 0.f);
    if (tj < ntypes) ab = tab[ti * ntypes + tj];
    stab[t] = make_float2(-12.0f * ab.x, 6.0f * ab.y);
  }
  __syncthreads();

  const int myiters = (nn - sub + LPA - 1) / LPA;  // entries kk < myiters are real for this lane
  int itmax = myiters, itmin = myiters;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    itmax = max(itmax, __shfl_xor(itmax, o, 64));
    itmin = min(itmin, __shfl_xor(itmin, o, 64));
  }
  const int nkk = __builtin_amdgcn_readfirstlane(itmax);
  // iterations every lane has entries for.  The unchecked loop takes the table offset as `entry >> 24`, which needs
  // the slot's bits 20..22 to be zero: systems of more than 2^20 atoms run all their iterations in the checked
  // loop, which masks the offset
  const int nfull = n > (1 << 20) ? 0 : __builtin_amdgcn_readfirstlane(itmin) / UNROLL * UNROLL;
  // bounds-checked raw buffer over sorted_xyzq: lanes past the end of their list read whatever the
  // (uninitialised) padding entry points at — out-of-range offsets return 0 instead of faulting — and
  // are discarded by `valid`
  const __amdgpu_buffer_rsrc_t srsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(sorted), 0, n * 16, 0x00020000);
  const char *tbase = reinterpret_cast<const char *>(stab);
  const float two_krf = 2.0f * c.krf;
  const float qi2k = pi.w * two_krf;
  const float sw_ir = c.inv_switch_range, sw_t0 = -c.switch_dist * c.inv_switch_range;
  auto in_vgpr = [](float sv) {  // a uniform value the compiler can no longer keep in an SGPR
    float v;
    asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(sv));
    return v;
  };
  const float vbx = in_vgpr(c.box[0]), vby = in_vgpr(c.box[1]), vbz = in_vgpr(c.box[2]);
  const float vibx = in_vgpr(c.invbox[0]), viby = in_vgpr(c.invbox[1]), vibz = in_vgpr(c.invbox[2]);
  const float vr2max = in_vgpr(c.r2max);
  // cutoff test as arithmetic: step = clamp((r2max' - r2) * 2^100, 0, 1) with r2max' the successor of r2max is exactly
  // 1 for r2 <= r2max and 0 beyond
  const float cut_h = in_vgpr(-1.2676506e30f);  // -2^100
  const float cut_c0 = in_vgpr(__int_as_float(__float_as_int(c.r2max) + 1) * 1.2676506e30f);

  float fx = 0.f, fy = 0.f, fz = 0.f;
  float e_lj = 0.f, e_el = 0.f;  // per-lane fp32 partial sums (~55 pairs), reduced in fp64

  using checked_t = std::integral_constant<bool, false>;
  using unchecked_t = std::integral_constant<bool, true>;
  // one group = this lane's 4 entries of iterations kk0 .. kk0+3 (one 16-byte list word) and their 4 gathered records;
  // tab[u] = byte offset of entry u's {-12 A, 6 B} in the LDS table (row of type i | 8 x type j)
  // (a stage = NU entries of a lane: 4 = one whole list word, 2 = half of one)
  auto group = [&](auto image, auto unchecked, const auto &tab, const auto &raw, int kk0) {
    constexpr bool EXACT = decltype(image)::value;
    constexpr bool UNCHECKED = decltype(unchecked)::value;
    constexpr bool ARITH_CUT = UNCHECKED && !ENERGY;
    constexpr int NU = (int)std::extent<std::remove_reference_t<decltype(tab)>>::value;
    static_assert(NU == 4 || NU == 2, "stage size");
#if TMD_EXP & 65536  // DEBUG (wrong results, timing only): the memory side alone — one add per gathered record
    if (UNCHECKED) {
#pragma unroll
      for (int u = 0; u < NU; ++u) fx += __uint_as_float(raw[u].x ^ raw[u].y ^ raw[u].z ^ raw[u].w ^ tab[u]);
      return;
    }
#endif
    float dx[NU], dy[NU], dz[NU], r2[NU], rinv[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      dx[u] = min_image_magic<EXACT>(pi.x - __uint_as_float(raw[u].x), vbx, vibx);
      dy[u] = min_image_magic<EXACT>(pi.y - __uint_as_float(raw[u].y), vby, viby);
      dz[u] = min_image_magic<EXACT>(pi.z - __uint_as_float(raw[u].z), vbz, vibz);
      r2[u] = norm2(dx[u], dy[u], dz[u]);
    }
    if constexpr (NU == 4) {
      asm("v_rsq_f32 %0, %4\n\tv_rsq_f32 %1, %5\n\tv_rsq_f32 %2, %6\n\tv_rsq_f32 %3, %7"
          : "=&v"(rinv[0]), "=&v"(rinv[1]), "=&v"(rinv[2]), "=&v"(rinv[3])
          : "v"(r2[0]), "v"(r2[1]), "v"(r2[2]), "v"(r2[3]));
    } else {
      asm("v_rsq_f32 %0, %2\n\tv_rsq_f32 %1, %3" : "=&v"(rinv[0]), "=&v"(rinv[1]) : "v"(r2[0]), "v"(r2[1]));
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const bool valid = UNCHECKED || (kk0 + u < myiters);  // padding words are garbage
      const float pjw = __uint_as_float(raw[u].w);
      const bool hit = valid && (r2[u] <= vr2max);
      const float rinv2 = rinv[u] * rinv[u];
      const float rinv6 = rinv2 * rinv2 * rinv2;
      float fs;  // (dE/dr) / r; rejected entries may produce inf/NaN here, the select below discards them
      float2 ab = make_float2(0.f, 0.f);  // (-12 A, 6 B)
      if (LJ) ab = *reinterpret_cast<const float2 *>(tbase + tab[u]);
      // E_lj = (A r^-6 - B) r^-6 from the force coefficients (energy / switching variants only)
      auto elj_of = [&](float r6) { return __builtin_fmaf(ab.x * (-1.0f / 12.0f), r6, ab.y * (-1.0f / 6.0f)) * r6; };
      if (LJ && !SWITCH && ELEC) {
        const float qq = pi.w * pjw;
        const float p = __builtin_fmaf(ab.x, rinv6, ab.y) * rinv6;  // (a12 rinv6 + b6) rinv6
        const float g = __builtin_fmaf(-qq, rinv[u], p);
        fs = __builtin_fmaf(rinv2, g, qi2k * pjw);
        if (ENERGY) e_lj += hit ? elj_of(rinv6) : 0.f;
      } else {
        fs = 0.f;
        float sw = 1.f;  // switching function S(r) of the LJ term (forces.py:402-412), 1 below switch_dist
        if (LJ) {
          fs = __builtin_fmaf(ab.x, rinv6, ab.y) * (rinv6 * rinv2);
          if (SWITCH) {
            // t = (r - r_s)/(r_c - r_s) clamped at 0: S = 1 + t^3 (-10 + t (15 - 6 t)),
            // S' = t^2 (-30 + t (60 - 30 t)) / (r_c - r_s);  (dE/dr)/r = S f + E S' x, x = 1/r (exact) or
            // 1/r^2 (the reference's explicit-force expression divides the switching term by r once more)
            const float r = r2[u] * rinv[u];
            const float t = fmaxf(__builtin_fmaf(r, sw_ir, sw_t0), 0.f);
            const float t2 = t * t;
            const float pp = __builtin_fmaf(t, __builtin_fmaf(t, -6.f, 15.f), -10.f);
            sw = __builtin_fmaf(t2 * t, pp, 1.f);
            const float dq = __builtin_fmaf(t, __builtin_fmaf(t, -30.f * sw_ir, 60.f * sw_ir), -30.f * sw_ir);
            const float elj = elj_of(rinv6);
            const float x = c.switch_reference_mode ? rinv2 : rinv[u];
            fs = __builtin_fmaf(sw, fs, elj * (t2 * dq) * x);
          }
          if (ENERGY) e_lj += hit ? sw * elj_of(rinv6) : 0.f;
        }
        if (ELEC) fs += (pi.w * pjw) * (two_krf - rinv2 * rinv[u]);
      }
      if (ENERGY && ELEC) e_el += hit ? (pi.w * pjw) * (rinv[u] + c.krf * r2[u] - c.crf) : 0.f;  // krf = crf = 0: plain Coulomb
      if (ARITH_CUT) {
        float step;
        asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(step) : "v"(r2[u]), "v"(cut_h), "v"(cut_c0));
        fs *= step;  // (every entry of the unchecked loop is a real pair, not a padding word: fs is finite)
      } else {
        fs = hit ? fs : 0.f;
      }
      fx = __builtin_fmaf(-dx[u], fs, fx);
      fy = __builtin_fmaf(-dy[u], fs, fy);
      fz = __builtin_fmaf(-dz[u], fs, fz);
    }
#if TMD_EXP & 512
    {
      float ex_[NU], ey_[NU], ez_[NU], q2[NU], qinv[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        ex_[u] = min_image_magic<EXACT>(pi2.x - __uint_as_float(raw[u].x), vbx, vibx);
        ey_[u] = min_image_magic<EXACT>(pi2.y - __uint_as_float(raw[u].y), vby, viby);
        ez_[u] = min_image_magic<EXACT>(pi2.z - __uint_as_float(raw[u].z), vbz, vibz);
        q2[u] = norm2(ex_[u], ey_[u], ez_[u]);
      }
      asm("v_rsq_f32 %0, %4\n\tv_rsq_f32 %1, %5\n\tv_rsq_f32 %2, %6\n\tv_rsq_f32 %3, %7"
          : "=&v"(qinv[0]), "=&v"(qinv[1]), "=&v"(qinv[2]), "=&v"(qinv[3])
          : "v"(q2[0]), "v"(q2[1]), "v"(q2[2]), "v"(q2[3]));
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const float pjw = __uint_as_float(raw[u].w);
        const float rinv2 = qinv[u] * qinv[u];
        const float rinv6 = rinv2 * rinv2 * rinv2;
        const float2 ab = *reinterpret_cast<const float2 *>(tbase + ((tab[u] & 0xFFu) | trow2));
        const float qq = pi2.w * pjw;
        const float p = __builtin_fmaf(ab.x, rinv6, ab.y) * rinv6;
        const float g2 = __builtin_fmaf(-qq, qinv[u], p);
        float fs = __builtin_fmaf(rinv2, g2, (pi2.w * two_krf) * pjw);
        float step;
        asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(step) : "v"(q2[u]), "v"(cut_h), "v"(cut_c0));
        fs *= step;
        gx = __builtin_fmaf(-ex_[u], fs, gx);
        gy = __builtin_fmaf(-ey_[u], fs, gy);
        gz = __builtin_fmaf(-ez_[u], fs, gz);
      }
    }
#endif
  };

  static_assert(UNROLL == 4, "one dwordx4 of list per lane and group");
  // issue the 4 gathers of the group whose list word is `w` and form its table offsets (unchecked: n <= 2^20, bits
  // 24..27 of an entry are zero; checked: padding words are garbage, the offset is masked)
  auto issue = [&](auto unchecked, const v4u &w, v4u (&raw)[UNROLL], unsigned (&tab)[UNROLL]) {
    const unsigned entry[UNROLL] = {w.x, w.y, w.z, w.w};
#if TMD_EXP & 16384  // DEBUG (wrong results, timing only): no gathers at all, records made up from the entry
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      raw[u] = (v4u){(entry[u] & 0x7F0u) | 0x41000000u, (entry[u] & 0x3F0u) | 0x41100000u, (entry[u] & 0x5F0u) | 0x41200000u, 0x3ECCCCCDu};
#elif TMD_EXP & 256  // DEBUG (wrong results, timing only): half the gathers, the other records made up from them
#pragma unroll
    for (int u = 0; u < UNROLL; u += 2) {
      raw[u] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, entry[u] & kEntryOffMask, 0, 0);
      raw[u + 1] = raw[u] ^ (v4u){entry[u + 1] & 0x3000u, entry[u + 1] & 0x5000u, entry[u + 1] & 0x6000u, 0u};
    }
#else
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) raw[u] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, entry[u] & kEntryOffMask, 0, 0);
#endif
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) tab[u] = trow | (decltype(unchecked)::value ? entry[u] >> 24 : (entry[u] >> 24) & 0xF8u);
  };
  const int gall = (nkk + UNROLL - 1) / UNROLL;  // groups of this wave
  int g = 0;                                     // next group to evaluate; `word` = its list word
  // A list word is requested AFTER the gathers issued in the same breath (see the head comment); sched_barrier pins
  // that order against the compiler's preference.
  auto checked_loop = [&](auto image) {  // per-lane validity; not pipelined (the tail is short)
    for (; g < gall; ++g) {
      v4u raw[UNROLL];
      unsigned tab[UNROLL];
      issue(checked_t{}, word, raw, tab);
      __builtin_amdgcn_sched_barrier(0);
      word = list_word(g + 1);
      __builtin_amdgcn_sched_barrier(0);
      group(image, checked_t{}, tab, raw, g * UNROLL);
    }
  };
  if (extent_needs_exact_image(ext, c.box)) {  // wave-uniform, rare: atoms more than 2.4 box edges apart
    checked_loop(exact_image{});
  } else {
    const int gfull = nfull / UNROLL;  // groups in which every lane has real entries: no validity test
    // Software pipeline over the unchecked groups: the gathers of group g+1 (and the list word of g+2) are requested
    // before group g is evaluated, into the other register set; a wave then waits for memory once per group, for
    // requests it made a whole group's arithmetic earlier (counters of the unpipelined loop: 44 % of a wave's cycles
    // in s_waitcnt, 27 % issuing — at the ~5 cycles per instruction a wave can issue by itself, six such waves do
    // not fill the VALU pipe).  94 VGPRs: five waves per SIMD.
    if constexpr (ENERGY || SWITCH || !ELEC || (TMD_EXP & 4)) {  // (the variants with more live values keep the plain loop: no spills at 5 waves)
      for (; g < gfull; ++g) {
        v4u raw[UNROLL];
        unsigned tab[UNROLL];
        issue(unchecked_t{}, word, raw, tab);
        __builtin_amdgcn_sched_barrier(0);
        word = list_word(g + 1);
        __builtin_amdgcn_sched_barrier(0);
        group(fused_image{}, unchecked_t{}, tab, raw, g * UNROLL);
      }
    }
#if TMD_EXP & 64
    // EXPERIMENT (measured, not the default): the pipeline in stages of TWO entries (half a list word) keeps the two
    // register sets + temporaries within 72 registers — seven waves per SIMD instead of five — and is SLOWER: 46.2 us
    // against 43.7 at C3 (eight waves, 64 registers + spills: 54 us).  More resident waves mean more neighbourhoods
    // competing for the 32 KB L1 of the CU, not more hidden latency.
    else if (gfull > 0) {
      auto issue2 = [&](unsigned e0, unsigned e1, v4u (&raw)[2], unsigned (&tab)[2]) {
        raw[0] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, e0 & kEntryOffMask, 0, 0);
        raw[1] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, e1 & kEntryOffMask, 0, 0);
        tab[0] = trow | (e0 >> 24);
        tab[1] = trow | (e1 >> 24);
      };
      v4u ra[2], rb[2];
      unsigned ta[2], tb[2];
      // `word` = list word g (arrives first), w1 = word g+1, w2 = word g+2 (requested in iteration g)
      v4u w1 = list_word(1), w2;
      issue2(word.x, word.y, ra, ta);
      for (; g < gfull; ++g) {
        issue2(word.z, word.w, rb, tb);
        __builtin_amdgcn_sched_barrier(0);
        w2 = list_word(g + 2);
        __builtin_amdgcn_sched_barrier(0);
        group(fused_image{}, unchecked_t{}, ta, ra, g * UNROLL);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 1 < gfull) issue2(w1.x, w1.y, ra, ta);  // (wave-uniform)
        __builtin_amdgcn_sched_barrier(0);
        group(fused_image{}, unchecked_t{}, tb, rb, g * UNROLL + 2);
        __builtin_amdgcn_sched_barrier(0);
        word = w1;
        w1 = w2;
      }
      // `word` = list word gfull for the tail; w1 (word gfull + 1) is requested again there
    }
#elif TMD_EXP & 128
    // EXPERIMENT (measured, not the default: 44.5 us against 42.6-43.8, 96 registers + 12 bytes of spills).
    // Software pipeline over the unchecked groups: while group g is evaluated the gathers of g+1 are in flight in the
    // other register set, and the list words of g+2 AND g+3 behind them.  A list word comes from the Infinity Cache /
    // HBM: ~1 500 cycles even on a quiet chip (a loop that does nothing but wait for its next list word takes 32 us)
    // — more than one group's arithmetic (~1 200 cycles), so with the word requested only one group ahead every
    // iteration ended up waiting for it: 1 985 cycles per group and wave, whatever was done to the gathers or the
    // arithmetic.  Two groups ahead it has two evaluations to arrive.
    else if (gfull > 0) {
      v4u ra[UNROLL], rb[UNROLL];
      unsigned ta[UNROLL], tb[UNROLL];
      issue(unchecked_t{}, word, ra, ta);  // G(0)
      __builtin_amdgcn_sched_barrier(0);
      v4u w1 = list_word(1), w2 = list_word(2);
      __builtin_amdgcn_sched_barrier(0);
      while (true) {  // entry: set a = G(g) in flight, w1 = W(g+1), w2 = W(g+2) in flight
        if (g + 1 >= gfull) {
          group(fused_image{}, unchecked_t{}, ta, ra, g * UNROLL);
          g += 1;
          break;
        }
        issue(unchecked_t{}, w1, rb, tb);  // G(g+1)
        __builtin_amdgcn_sched_barrier(0);
        w1 = list_word(g + 3);
        __builtin_amdgcn_sched_barrier(0);
        group(fused_image{}, unchecked_t{}, ta, ra, g * UNROLL);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 2 >= gfull) {
          group(fused_image{}, unchecked_t{}, tb, rb, (g + 1) * UNROLL);
          g += 2;
          { const v4u t = w1; w1 = w2; w2 = t; }  // (w2 = W(g) after the increment below is what the tail wants)
          break;
        }
        issue(unchecked_t{}, w2, ra, ta);  // G(g+2)
        __builtin_amdgcn_sched_barrier(0);
        w2 = list_word(g + 4);
        __builtin_amdgcn_sched_barrier(0);
        group(fused_image{}, unchecked_t{}, tb, rb, (g + 1) * UNROLL);
        __builtin_amdgcn_sched_barrier(0);
        g += 2;
      }
      word = w1;  // list word of group g (requested long ago); the tail requests g+1 again itself
    }
#else
    else if (gfull > 0) {
      v4u ra[UNROLL], rb[UNROLL];
      unsigned ta[UNROLL], tb[UNROLL];
      issue(unchecked_t{}, word, ra, ta);
      __builtin_amdgcn_sched_barrier(0);
      word = list_word(1);
      __builtin_amdgcn_sched_barrier(0);
      while (true) {
        if (g + 1 >= gfull) {
          group(fused_image{}, unchecked_t{}, ta, ra, g * UNROLL);
          g += 1;
          break;
        }
        issue(unchecked_t{}, word, rb, tb);
        __builtin_amdgcn_sched_barrier(0);
        word = list_word(g + 2);
        __builtin_amdgcn_sched_barrier(0);
        group(fused_image{}, unchecked_t{}, ta, ra, g * UNROLL);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 2 >= gfull) {
          group(fused_image{}, unchecked_t{}, tb, rb, (g + 1) * UNROLL);
          g += 2;
          break;
        }
        issue(unchecked_t{}, word, ra, ta);
        __builtin_amdgcn_sched_barrier(0);
        word = list_word(g + 3);
        __builtin_amdgcn_sched_barrier(0);
        group(fused_image{}, unchecked_t{}, tb, rb, (g + 1) * UNROLL);
        __builtin_amdgcn_sched_barrier(0);
        g += 2;
      }
    }
#endif
    checked_loop(fused_image{});  // tail
  }
  float sx = fx, sy = fy, sz = fz;
#pragma unroll
  for (int o = LPA >> 1; o > 0; o >>= 1) {
    sx += __shfl_xor(sx, o, 64);
    sy += __shfl_xor(sy, o, 64);
    sz += __shfl_xor(sz, o, 64);
  }
#if TMD_EXP & 512
  {
    float tx = gx, ty = gy, tz = gz;
#pragma unroll
    for (int o = LPA >> 1; o > 0; o >>= 1) {
      tx += __shfl_xor(tx, o, 64);
      ty += __shfl_xor(ty, o, 64);
      tz += __shfl_xor(tz, o, 64);
    }
    if (active && sub == 0 && forces) {
      forces[3 * oi2 + 0] = tx;
      forces[3 * oi2 + 1] = ty;
      forces[3 * oi2 + 2] = tz;
    }
  }
#endif
  if constexpr (FUSED != 0) {
    // The force record {fx, fy, fz, launch number} goes to the cell-sorted array the step blocks watch, as ONE 16-byte
    // store written through to device scope (sc1): the number in .w says the force beside it is this launch's.
    // (A flag per wave behind the stores cost a memory round trip more at the end of the launch; an agent-scope
    // release does it with buffer_wbl2, a write-back of the whole L2 per wave: 365 us per launch.)
    const __amdgpu_buffer_rsrc_t frsrc = __builtin_amdgcn_make_buffer_rsrc(fstep.fsort, 0, n * 16, 0x00020000);
    if (active && sub == 0)
      __builtin_amdgcn_raw_buffer_store_b128((v4u){__float_as_uint(sx), __float_as_uint(sy), __float_as_uint(sz), fstep.gen},
                                             frsrc, a * 16, 0, kAuxDeviceScope);
    return;
  }
  if (active && sub == 0 && forces) {
    if (overwrite) {
      forces[3 * oi + 0] = sx;
      forces[3 * oi + 1] = sy;
      forces[3 * oi + 2] = sz;
    } else {
      forces[3 * oi + 0] += sx;
      forces[3 * oi + 1] += sy;
      forces[3 * oi + 2] += sz;
    }
  }
  if (ENERGY) {  // every pair is listed from both atoms: half of the sum
    if (LJ) {
      const double s = wave_sum((double)e_lj);
      if (lane == 0 && s != 0.0) unsafeAtomicAdd(&energy_row(energies)[TMDHIP_E_LJ], 0.5 * s);
    }
    if (ELEC) {
      const double s = wave_sum((double)e_el);
      if (lane == 0 && s != 0.0) unsafeAtomicAdd(&energy_row(energies)[TMDHIP_E_ELECTROSTATICS], 0.5 * s);
    }
  }
}

// ---- K3d: the same lean kernel for fp64 contexts ----------------------------------------------------
// 32-byte records (two 16-byte gathers per entry), 16-byte table entries, half-rate arithmetic; 1/r from v_rsq_f64
// and two Newton steps.  Same entry format, list layout and decision arithmetic (min_image_magic's fp64 overload:
// magic number 1.5 * 2^52; norm2's fp64 order).
template <bool EXACT>
__device__ __forceinline__ double min_image_magic(double d, double box, double invbox) {
#pragma clang fp contract(off)
  const double magic = 6755399441055744.0;
  const double t = __builtin_fma(d, invbox, magic);
  const double k = t - magic;
  if (EXACT) {
    const double p = box * k;
    return d - p;
  }
  return __builtin_fma(-k, box, d);
}

template <int LPA, bool LJ, bool ELEC, bool ENERGY, bool SWITCH>
__global__ __launch_bounds__(256) void list_pair_lean_f64_kernel(
    int n, const double4 *__restrict__ sorted, const int *__restrict__ stype, const int *__restrict__ order,
    int ntypes, const double2 *__restrict__ tab, const unsigned *__restrict__ nlist,
    const int *__restrict__ nneigh, int maxn, PairConsts<double> c, double *__restrict__ forces, int overwrite,
    double *__restrict__ energies, unsigned *publish, unsigned publish_value, const int *__restrict__ ext) {
  constexpr int APW = 64 / LPA;
  constexpr int UNROLL = 4;
  // tells the host (host-mapped word) that everything enqueued before this launch has completed
  if (publish && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(publish, publish_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __shared__ __align__(16) double2 stab[kEntryTypes * kEntryTypes];  // row of type i: 32 x {-12 A, 6 B}
  for (int t = threadIdx.x; t < ntypes * kEntryTypes; t += blockDim.x) {  // rows of existing classes only
    const int ti = t >> 5, tj = t & 31;
    double2 ab = make_double2(0.0, 0.0);
    if (tj < ntypes) ab = tab[ti * ntypes + tj];
    stab[t] = make_double2(-12.0 * ab.x, 6.0 * ab.y);
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  // XCD-aware block order: consecutive block ids go to the 8 XCDs round-robin, so block b works on
  // chunk (b % 8) * gridDim.x/8 + b / 8 — every XCD (own L2) gets a contiguous eighth of the cell-sorted
  // atoms and gathers neighbours from that region only.  gridDim.x is a multiple of 8; the surplus
  // blocks of the last eighths have nothing to do.
  const int blk = (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3));
  if (blk * 4 * APW >= n) return;
  const int wave = blk * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int a = wave * APW + lane / LPA;
  const int sub = lane % LPA;
  const bool active = a < n;
  double4 pi = make_double4(0.0, 0.0, 0.0, 0.0);
  int nn = 0;
  unsigned trow = 0;  // byte offset of this atom's row of the LDS table
  if (active) {
    pi = sorted[a];
    nn = nneigh[a];
    trow = (unsigned)stype[a] << 9;  // rows of 32 x 16 B
  }
  const int myiters = (nn - sub + LPA - 1) / LPA;  // entries kk < myiters are real for this lane
  int itmax = myiters, itmin = myiters;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    itmax = max(itmax, __shfl_xor(itmax, o, 64));
    itmin = min(itmin, __shfl_xor(itmin, o, 64));
  }
  const int nkk = __builtin_amdgcn_readfirstlane(itmax);
  const int nfull = __builtin_amdgcn_readfirstlane(itmin) / UNROLL * UNROLL;  // iterations every lane has entries for
  // a lane's entries of iterations 4G .. 4G+3 are one 16-byte word at row4[G * 64]
  const v4u *row4 = reinterpret_cast<const v4u *>(nlist + (size_t)wave * maxn * APW) + lane;
  // bounds-checked raw buffer over sorted_xyzq: lanes past the end of their list read whatever the
  // (uninitialised) padding entry points at — out-of-range offsets return 0 instead of faulting — and
  // are discarded by `valid`
  const __amdgpu_buffer_rsrc_t srsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<double4 *>(sorted), 0, n * 32, 0x00020000);
  const char *tbase = reinterpret_cast<const char *>(stab);
  const double two_krf = 2.0 * c.krf;
  const double qi2k = pi.w * two_krf;
  const double sw_ir = c.inv_switch_range, sw_t0 = -c.switch_dist * c.inv_switch_range;
  const double bx = c.box[0], by = c.box[1], bz = c.box[2];
  const double ibx = c.invbox[0], iby = c.invbox[1], ibz = c.invbox[2];
  const double r2max = c.r2max;

  double fx = 0.0, fy = 0.0, fz = 0.0;
  double e_lj = 0.0, e_el = 0.0;

  auto body = [&](auto image, unsigned tofs, const v4u &lo, const v4u &hi, bool valid) {  // one list entry
    constexpr bool EXACT = decltype(image)::value;
    const double pjx = __hiloint2double((int)lo.y, (int)lo.x), pjy = __hiloint2double((int)lo.w, (int)lo.z);
    const double pjz = __hiloint2double((int)hi.y, (int)hi.x), pjw = __hiloint2double((int)hi.w, (int)hi.z);
    const double dx = min_image_magic<EXACT>(pi.x - pjx, bx, ibx);
    const double dy = min_image_magic<EXACT>(pi.y - pjy, by, iby);
    const double dz = min_image_magic<EXACT>(pi.z - pjz, bz, ibz);
    const double r2 = norm2(dx, dy, dz);
    const bool hit = valid && (r2 <= r2max);
    // 1/r: v_rsq_f64 (~2^-26 relative) + two Newton steps; rejected entries may produce inf/NaN, discarded below
    double rinv = __builtin_amdgcn_rsq(r2);
    rinv = rinv * __builtin_fma(-0.5 * r2 * rinv, rinv, 1.5);
    rinv = rinv * __builtin_fma(-0.5 * r2 * rinv, rinv, 1.5);
    const double rinv2 = rinv * rinv;
    const double rinv6 = rinv2 * rinv2 * rinv2;
    double fs;  // (dE/dr) / r
    double2 ab = make_double2(0.0, 0.0);  // (-12 A, 6 B)
    if (LJ) ab = *reinterpret_cast<const double2 *>(tbase + (trow | tofs));
    auto elj_of = [&](double r6) { return __builtin_fma(ab.x * (-1.0 / 12.0), r6, ab.y * (-1.0 / 6.0)) * r6; };
    if (LJ && !SWITCH && ELEC) {
      const double qq = pi.w * pjw;
      const double p = __builtin_fma(ab.x, rinv6, ab.y) * rinv6;
      const double g = __builtin_fma(-qq, rinv, p);
      fs = __builtin_fma(rinv2, g, qi2k * pjw);
      if (ENERGY) e_lj += hit ? elj_of(rinv6) : 0.0;
    } else {
      fs = 0.0;
      double sw = 1.0;
      if (LJ) {
        fs = __builtin_fma(ab.x, rinv6, ab.y) * (rinv6 * rinv2);
        if (SWITCH) {  // same polynomial as the fp32 kernel (forces.py:402-412)
          const double r = r2 * rinv;
          const double t = fmax(__builtin_fma(r, sw_ir, sw_t0), 0.0);
          const double t2 = t * t;
          const double pp = __builtin_fma(t, __builtin_fma(t, -6.0, 15.0), -10.0);
          sw = __builtin_fma(t2 * t, pp, 1.0);
          const double dq = __builtin_fma(t, __builtin_fma(t, -30.0 * sw_ir, 60.0 * sw_ir), -30.0 * sw_ir);
          const double elj = elj_of(rinv6);
          const double x = c.switch_reference_mode ? rinv2 : rinv;
          fs = __builtin_fma(sw, fs, elj * (t2 * dq) * x);
        }
        if (ENERGY) e_lj += hit ? sw * elj_of(rinv6) : 0.0;
      }
      if (ELEC) fs += (pi.w * pjw) * (two_krf - rinv2 * rinv);
    }
    if (ENERGY && ELEC) e_el += hit ? (pi.w * pjw) * (rinv + c.krf * r2 - c.crf) : 0.0;
    fs = hit ? fs : 0.0;
    fx = __builtin_fma(-dx, fs, fx);
    fy = __builtin_fma(-dy, fs, fy);
    fz = __builtin_fma(-dz, fs, fz);
  };

  static_assert(UNROLL == 4, "one dwordx4 of list per lane and group");
  // index words are fetched two groups (8 entries per lane, 2 KB per wave) ahead of their use
  v4u nxa = row4[0], nxb = row4[64];  // rows are padded: always readable
  int kk0 = 0;
  auto checked_loop = [&](auto image) {  // per-lane validity
    for (; kk0 < nkk; kk0 += UNROLL) {
      const v4u cur = nxa;
      nxa = nxb;
      nxb = row4[(size_t)((kk0 >> 2) + 2) * 64];
      const unsigned entry[UNROLL] = {cur.x, cur.y, cur.z, cur.w};
      v4u lo[UNROLL], hi[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const unsigned off = (entry[u] & kEntryOffMask) << 1;
        lo[u] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, off, 0, 0);
        hi[u] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, off + 16u, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) body(image, (entry[u] >> 23) & 0x1F0u, lo[u], hi[u], kk0 + u < myiters);  // padding words are garbage
    }
  };
  if (extent_needs_exact_image(ext, c.box)) {  // wave-uniform, rare: atoms more than 2.4 box edges apart
    checked_loop(exact_image{});
  } else {
    for (; kk0 < nfull; kk0 += UNROLL) {  // every lane has real entries here: no validity test
      const v4u cur = nxa;
      nxa = nxb;
      nxb = row4[(size_t)((kk0 >> 2) + 2) * 64];
      const unsigned entry[UNROLL] = {cur.x, cur.y, cur.z, cur.w};
      v4u lo[UNROLL], hi[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {  // 32-byte records: byte offset = 2 x the entry's 16-byte-record offset
        const unsigned off = (entry[u] & kEntryOffMask) << 1;
        lo[u] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, off, 0, 0);
        hi[u] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, off + 16u, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) body(fused_image{}, (entry[u] >> 23) & 0x1F0u, lo[u], hi[u], true);
    }
    checked_loop(fused_image{});  // tail
  }
  double sx = fx, sy = fy, sz = fz;
#pragma unroll
  for (int o = LPA >> 1; o > 0; o >>= 1) {
    sx += __shfl_xor(sx, o, 64);
    sy += __shfl_xor(sy, o, 64);
    sz += __shfl_xor(sz, o, 64);
  }
  if (active && sub == 0 && forces) {
    const int oi = order[a];
    if (overwrite) {
      forces[3 * oi + 0] = sx;
      forces[3 * oi + 1] = sy;
      forces[3 * oi + 2] = sz;
    } else {
      forces[3 * oi + 0] += sx;
      forces[3 * oi + 1] += sy;
      forces[3 * oi + 2] += sz;
    }
  }
  if (ENERGY) {  // every pair is listed from both atoms: half of the sum
    if (LJ) {
      const double s = wave_sum(e_lj);
      if (lane == 0 && s != 0.0) unsafeAtomicAdd(&energy_row(energies)[TMDHIP_E_LJ], 0.5 * s);
    }
    if (ELEC) {
      const double s = wave_sum(e_el);
      if (lane == 0 && s != 0.0) unsafeAtomicAdd(&energy_row(energies)[TMDHIP_E_ELECTROSTATICS], 0.5 * s);
    }
  }
}

// ---- fused MD-step kernel (integrator.py:61-74 across the step boundary) -----------------------------
// One launch per replica and step: [Langevin kick + second half kick of step s-1] + [first half step of
// step s] + [displacement test that drives the device-side list rebuild].  Values are identical to the
// separate kernels of integrator.hip (same operations on the same registers, no re-association).
template <typename R>
struct MdStepArgs {
  int n;
  const R *pos_in;  // positions before the drift (== pos_out except in the double-buffered bonded variant)
  R *pos_out;
  R *vel;
  const R *f;
  R *f_zero;  // non-null: clear the force after reading it (the all-pairs kernel that follows accumulates)
  const R *mass, *vcoeff;
  R dt, half_dt, gamma;
  uint64_t seed, noise_step, row0;
  ListCheck<R> chk;  // displacement test that drives the rebuilds (CHECK variants)
  typename Vec<R>::T4 *sorted;
  const int *inv;
  const R *qs;
};

// Everything the update of one atom reads, loaded in ONE batch before any arithmetic or store: the kernel is
// a chain of memory round trips per wave (every wave of the launch is resident at once), and stores to the
// position buffers would otherwise order the later loads (inv, ref, qs) behind them.
template <typename R>
struct AtomIn {
  R m, vc, q, h2;
  R v[3], f[3], p[3], r[3];
  int slot;
};

template <typename R, bool SECOND, bool LANGEVIN, bool FIRST, bool CHECK>
__device__ __forceinline__ AtomIn<R> md_load_atom(const MdStepArgs<R> &s, int i, size_t off) {
  AtomIn<R> x;
  const R *vel = s.vel + off, *f = s.f + off, *pos_in = s.pos_in + off;
  x.m = s.mass[i];
  x.vc = (SECOND && LANGEVIN) ? s.vcoeff[i] : R(0);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    x.v[k] = vel[3 * i + k];
    x.f[k] = f[3 * i + k];
    x.p[k] = FIRST ? pos_in[3 * i + k] : R(0);
    x.r[k] = (FIRST && CHECK) ? s.chk.ref[3 * i + k] : R(0);
  }
  x.q = (FIRST && CHECK) ? s.qs[i] : R(0);
  x.h2 = (FIRST && CHECK) ? list_check_limit(s.chk, i) : R(0);
  x.slot = (FIRST && CHECK) ? s.inv[i] : 0;
  return x;
}

// fb = extra force on atom i that is not in `f` (the inline bonded force), added before the division
// by the mass exactly like the separate bonded kernel's `forces[i] += fb`
template <typename R, bool SECOND, bool LANGEVIN, bool FIRST, bool CHECK>
__device__ __forceinline__ void md_step_atom(const MdStepArgs<R> &s, const PairConsts<R> &c, int i, size_t off,
                                             uint64_t row0, const AtomIn<R> &x, const R (&fb)[3], bool add_fb,
                                             const R *noise = nullptr) {  // noise: normal3 of this atom, drawn earlier
#pragma clang fp contract(off)
  R *pos_out = s.pos_out + off, *vel = s.vel + off;
  const R m = x.m;
  R v[3], a[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    v[k] = x.v[k];
    R fk = x.f[k];
    if (add_fb) fk += fb[k];
    a[k] = fk / m;
  }
  if (s.f_zero) {
    R *fz = s.f_zero + off;
#pragma unroll
    for (int k = 0; k < 3; ++k) fz[3 * i + k] = R(0);
  }
  if (SECOND) {
    if (LANGEVIN) {
      const R vc = x.vc;
      R g[3];
      if (noise) {
        g[0] = noise[0], g[1] = noise[1], g[2] = noise[2];
      } else {
        normal3<R>(s.seed, s.noise_step, row0 + (uint64_t)i, g[0], g[1], g[2]);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) v[k] += -s.gamma * v[k] * s.dt + g[k] * vc;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] += s.half_dt * a[k];
  }
  if (FIRST) {
    R p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      p[k] = x.p[k] + (v[k] * s.dt + R(0.5) * a[k] * s.dt * s.dt);
      v[k] = v[k] + s.half_dt * a[k];
      pos_out[3 * i + k] = p[k];
    }
    if (CHECK) {
      // keep the cell-sorted copy the pair kernel reads current (on rebuild steps place_sorted_kernel
      // rewrites it in the new order)
      typename Vec<R>::T4 sv;  // one full 16/32-byte store (partial writes of a record are slower)
      sv.x = p[0];
      sv.y = p[1];
      sv.z = p[2];
      sv.w = x.q;
      s.sorted[x.slot] = sv;
      extent_note<R>(s.chk.ext, p[0], p[1], p[2]);
      list_check_point<R>(s.chk, c, p[0] - x.r[0], p[1] - x.r[1], p[2] - x.r[2], x.h2);
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) vel[3 * i + k] = v[k];
}

// ---- the step in the lean fp32 pair kernel's epilogue (see FusedStep above the kernel) -------------------------
struct FusedStatic {
  MdStepArgs<float> s;  // per-launch fields (pos_in/out, sorted, noise_step, chk.near_host/seq/parity) come from FusedStep
  BondedArgs<float> A;
  int has_bonded;  // 1: light topology, the atoms' bonded records are evaluated here (md_step_bonded_kernel's job);
                   // 2: heavy topology, the bonded force of this launch's positions is in `fbond` (bonded_wave_kernel
                   // ran in front of the launch: it depends on the positions only)
  const float *fbond;  // [3N], original atom order
};

// Step block j of a FUSED pair launch (four waves, 64 atoms): the atoms of the 64 / APB pair blocks that run on the
// same XCD (block ids congruent mod 8) and are neighbours in the cell-sorted order.  Like md_step_bonded_kernel, wave w
// evaluates bonded record slots w, w + 4, ... of all 64 atoms (lane = atom), the partial forces meet in LDS as
// (p0 + p1) + (p2 + p3), and the first wave updates — after it has waited for the pair waves of its atoms.
template <bool LANGEVIN, int APB>
__device__ __forceinline__ void fused_step_blocks(const FusedStatic *__restrict__ fst, const FusedStep &fs,
                                                  const PairConsts<float> &c, int n, const float4 *__restrict__ sorted,
                                                  const int *__restrict__ order, int j, int npair, float *s_lds) {
  constexpr int K = 64 / APB;  // pair blocks per 64 atoms
  float(*s_part)[3][64] = reinterpret_cast<float(*)[3][64]>(s_lds);  // [kQuad][3][64], the pair role's LJ table space
  const int w = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  // With bonded records a step block is 64 atoms (its four waves share their records); without, every wave is a unit
  // of 64 atoms of its own (four waves of which three only met at the barrier doubled the waves of a 10^6-atom LJ launch).
  const bool bonded = fs.bonded == 1;  // (launch-uniform; 2 = the bonded force comes from a buffer: waves are units too)
  const int xcd = j & 7, q = bonded ? (j >> 3) : (j >> 3) * kQuad + w, g8 = npair >> 3;
  const int kc = K * q + lane / APB;  // this lane's pair block within the XCD's eighth
  const int a = (xcd * g8 + kc) * APB + lane % APB;
  const bool exists = kc < g8 && a < n;
  const int o = exists ? order[a] : 0;
  MdStepArgs<float> s = fst->s;
  s.pos_in = fs.pos_in;
  s.pos_out = fs.pos_out;
  s.sorted = fs.sorted_out;
  s.noise_step = fs.noise_step;
  s.f_zero = nullptr;
  s.chk.near_host = fs.near_host;
  s.chk.seq = fs.seq;
  s.chk.parity = fs.parity;
  s.chk.skipped = 0;  // (unknown here: the next launch's first thread looks, kLmViolation)
  const bool integrates = (w == 0 || !bonded) && exists;
  AtomIn<float> x{};
  if (integrates) {  // every load of the update but the force, in flight during the bonded part
    x.m = s.mass[o];
    x.vc = LANGEVIN ? s.vcoeff[o] : 0.f;
    const float4 p = sorted[a];  // x, y, z, scaled charge: exactly what the position buffer holds
    x.p[0] = p.x, x.p[1] = p.y, x.p[2] = p.z;
    x.q = p.w;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      x.v[k] = s.vel[3 * o + k];
      x.r[k] = s.chk.ref[3 * o + k];
    }
    x.h2 = list_check_limit(s.chk, o);
    x.slot = a;
  }
  float fb[3] = {0.f, 0.f, 0.f};
  float g[3] = {0.f, 0.f, 0.f};
  if (bonded) {
    float fx = 0.f, fy = 0.f, fz = 0.f;
    if (exists) {
      const BondedArgs<float> A = fst->A;
      double e[TMDHIP_NENERGY] = {0, 0, 0, 0, 0, 0, 0, 0};  // energies are not wanted on interior steps (dead)
      const AtomRec<float> *rec = A.arec + (size_t)o * A.arec_stride;
      for (int k = w; k < A.arec_stride; k += kQuad) {
        const AtomRec<float> r = rec[k];
        if (r.ent == kNoRec) break;  // records are packed from the front
        eval_rec<float>(A, s.pos_in, o, r, fx, fy, fz, e);
      }
    }
    s_part[w][0][lane] = fx;
    s_part[w][1][lane] = fy;
    s_part[w][2][lane] = fz;
    if (LANGEVIN && integrates) normal3<float>(s.seed, s.noise_step, s.row0 + (uint64_t)o, g[0], g[1], g[2]);
    __syncthreads();
    if (w != 0) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) fb[k] = (s_part[0][k][lane] + s_part[1][k][lane]) + (s_part[2][k][lane] + s_part[3][k][lane]);
  } else {
    if (fs.bonded == 2 && integrates) {
      const float *fbond = fst->fbond;
#pragma unroll
      for (int k = 0; k < 3; ++k) fb[k] = fbond[3 * o + k];
    }
    if (LANGEVIN && integrates) normal3<float>(s.seed, s.noise_step, s.row0 + (uint64_t)o, g[0], g[1], g[2]);
  }
  // Wait for this atom's force record of THIS launch (.w = launch number; a 16-byte access is one request at the L2).
  // Its pair block was dispatched before this block and waits for nothing; the bound only keeps a broken assumption
  // from hanging the GPU.
  const __amdgpu_buffer_rsrc_t frsrc = __builtin_amdgcn_make_buffer_rsrc(fs.fsort, 0, n * 16, 0x00020000);
  v4u f = (v4u){0u, 0u, 0u, fs.gen};
  if (integrates) {
    unsigned spins = 0;
    while (true) {
      f = __builtin_amdgcn_raw_buffer_load_b128(frsrc, a * 16, 0, kAuxDeviceScope);
      if (f.w == fs.gen) break;
      __builtin_amdgcn_s_sleep(TMD_STEP_POLL_SLEEP);
      if (++spins > (1u << 22)) {
        s.chk.flags[F_VIOLATION] = 1;  // the caller rewinds and repeats the batch
        break;
      }
    }
  }
  if (!integrates) return;
  x.f[0] = __uint_as_float(f.x), x.f[1] = __uint_as_float(f.y), x.f[2] = __uint_as_float(f.z);
  md_step_atom<float, true, LANGEVIN, true, true>(s, c, o, 0, s.row0, x, fb, fs.bonded != 0, LANGEVIN ? g : nullptr);
}

template <typename R, bool SECOND, bool LANGEVIN, bool FIRST, bool CHECK>
__global__ void md_step_kernel(MdStepArgs<R> s, PairConsts<R> c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (CHECK && i == 0) list_check_clear(s.chk.flags, s.chk.parity);
  if (i >= s.n) return;
  // replica batch (all-pairs systems, never with CHECK): blockIdx.y = replica
  const size_t off = CHECK ? 0 : (size_t)blockIdx.y * 3 * s.n;
  const uint64_t row0 = s.row0 + (CHECK ? 0 : (uint64_t)blockIdx.y * (uint64_t)s.n);
  const R none[3] = {0, 0, 0};
  const AtomIn<R> x = md_load_atom<R, SECOND, LANGEVIN, FIRST, CHECK>(s, i, off);
  md_step_atom<R, SECOND, LANGEVIN, FIRST, CHECK>(s, c, i, off, row0, x, none, false);
}

// Interior steps of an MD run: the bonded force of the previous step's positions is evaluated HERE
// instead of by a bonded kernel of its own (one launch and one read-modify-write pass over `forces` less
// per step; bit-identical to the separate kernels: the same device functions in the same order, added to
// the stored pair force before the division by the mass).  Partner positions must be the undrifted ones,
// so the step reads pos_in and writes pos_out (two buffers).  Light topologies only (thread per atom,
// per-atom records): for proteins a wave-per-atom variant with lane 0 integrating was measured slower than
// the separate bonded_wave_kernel (alanine dipeptide 47 vs 42.5 us/step: the two phases serialise inside
// each wave).  Without CHECK (all-pairs systems) blockIdx.y is the replica.
template <typename R, bool LANGEVIN, bool CHECK>
__global__ __launch_bounds__(256) void md_step_bonded_kernel(MdStepArgs<R> s, PairConsts<R> c, BondedArgs<R> A,
                                                             const R *__restrict__ boxes) {
  if (CHECK && blockIdx.x == 0 && threadIdx.x == 0) list_check_clear(s.chk.flags, s.chk.parity);
  const int rep = CHECK ? 0 : (int)blockIdx.y;
  const size_t off = (size_t)rep * 3 * s.n;
  const uint64_t row0 = s.row0 + (uint64_t)rep * (uint64_t)s.n;
  if (!CHECK && boxes) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      A.b.box[k] = boxes[6 * rep + k];
      A.b.invbox[k] = boxes[6 * rep + 3 + k];
    }
  }
  const R *pos = s.pos_in + off;
  R fx = 0, fy = 0, fz = 0;
  double e[TMDHIP_NENERGY] = {0, 0, 0, 0, 0, 0, 0, 0};  // energies are not wanted on interior steps (dead)
  // A block of 256 threads = 64 atoms.  Bonded records: wave w evaluates slots w, w + 4, ... of all 64 atoms
  // (lane = atom), so that the lanes of a wave work on the same KIND of record wherever the atoms' record lists
  // look alike — water: waves 0 and 1 evaluate a bond for every atom, wave 2 an angle, wave 3 has nothing to do —
  // instead of four adjacent lanes per atom running the bond and the angle code one after the other (kernel
  // 8.95 -> 8.15 us at C3; the rest is memory round trips).  The per-slot partial forces meet in LDS and are
  // added in the order of eval_atom_quad's butterfly, (p0 + p1) + (p2 + p3): bit-identical to the separate
  // bonded kernel.  The update itself (noise, kicks, drift) runs one atom per lane on the block's first wave,
  // which issues the loads of its 64 atoms before the bonded part so that they are in flight meanwhile.
  __shared__ R s_part[kQuad][3][64];
  const int a0 = blockIdx.x * 64;
  const int w = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  const int mine = a0 + lane;  // this lane's atom: its slots w, w + 4, ... here, its update on the first wave
  const bool integrates = w == 0 && mine < s.n;
  AtomIn<R> x{};
  if (integrates) x = md_load_atom<R, true, LANGEVIN, true, CHECK>(s, mine, off);
  if (mine < s.n) {
    const AtomRec<R> *rec = A.arec + (size_t)mine * A.arec_stride;
    for (int k = w; k < A.arec_stride; k += kQuad) {
      const AtomRec<R> r = rec[k];
      if (r.ent == kNoRec) break;  // records are packed from the front
      eval_rec<R>(A, pos, mine, r, fx, fy, fz, e);
    }
  }
  s_part[w][0][lane] = fx;
  s_part[w][1][lane] = fy;
  s_part[w][2][lane] = fz;
  __syncthreads();
  if (!integrates) return;
  R fb[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) fb[k] = (s_part[0][k][lane] + s_part[1][k][lane]) + (s_part[2][k][lane] + s_part[3][k][lane]);
  md_step_atom<R, true, LANGEVIN, true, CHECK>(s, c, mine, off, row0, x, fb, true);
}

__global__ void halve_count_kernel(unsigned long long *c) { *c >>= 1; }

// tmdhip_md_observe: the per-term energies, the kinetic energies and the list flags of every replica written
// straight into host-mapped memory by one small block, followed by a sequence word the host spins on — instead of
// three device-to-host copy commands and a stream synchronisation (whose wake-up is the slowest part of a short
// call).  flags.p[r] = replica r's int[F_COUNT], or null.
struct ObsFlagPtrs {
  const int *p[16];
};
__global__ void observe_publish_kernel(int nrep, const double *__restrict__ energies, const double *__restrict__ ke,
                                       ObsFlagPtrs flags, double *host_e, double *host_ke, int *host_flags,
                                       unsigned *host_seq, unsigned seq) {
  const int t = threadIdx.x;
  for (int k = t; k < nrep * TMDHIP_NENERGY; k += blockDim.x) host_e[k] = energies ? energies[k] : 0.0;
  for (int k = t; k < nrep; k += blockDim.x) host_ke[k] = ke ? ke[k] : 0.0;
  for (int k = t; k < nrep * F_COUNT; k += blockDim.x) {
    const int r = k / F_COUNT;
    host_flags[k] = flags.p[r] ? flags.p[r][k - r * F_COUNT] : 0;
  }
  __threadfence_system();
  __syncthreads();
  if (t == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// host side of observe_publish_kernel: spin until the device has written `seq` (all results are then in place)
int wait_observed(volatile unsigned *hseq, unsigned seq, hipStream_t st) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 1; *hseq != seq; ++spins) {
    __builtin_ia32_pause();
    if ((spins & 0xFFFFu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
      TMD_HIP(hipStreamSynchronize(st));  // surfaces a device error if there is one
      if (*hseq != seq) return fail("the device did not report the results of the call");
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return 0;
}

// state at the entry of an MD batch (positions, velocities, forces) in one launch; n4 = 16-byte words per array
__global__ void snapshot3_kernel(size_t n4, const uint4 *__restrict__ a, const uint4 *__restrict__ b,
                                 const uint4 *__restrict__ c, uint4 *__restrict__ out, double *__restrict__ zero,
                                 int nzero) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (zero && i < (size_t)nzero) zero[i] = 0.0;  // the call's energy buffer (one fill launch less per call)
  if (i >= n4) return;
  out[i] = a[i];
  out[n4 + i] = b[i];
  out[2 * n4 + i] = c[i];
}

}  // namespace tmd

// =============================================================================================
// host side: context
// =============================================================================================
using namespace tmd;

namespace {

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return 0;
    if (p) {  // a buffer that grows once tends to grow again (atom sets that change at every migration of a
              // domain decomposition, list capacities): 1/8 of slack instead of a hipFree + hipMalloc each time
      TMD_HIP(hipFree(p));
      need += need / 8;
    }
    p = nullptr;
    bytes = 0;
    TMD_HIP(hipMalloc(&p, need));
    bytes = need;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <typename T>
  T *as() const {
    return reinterpret_cast<T *>(p);
  }
};

struct Replica {
  int64_t step = 0;
  int64_t n_compute = 0;
  bool have_list = false;
  double box[3] = {-1, -1, -1};
  Grid grid{};
  int ncell = 0;
  ListGeom lg{1, 64, 0, 0};
  int64_t host_rebuilds = 0;
  DevBuf cell_of, slot, order_tmp, order, inv, count, cell_start, sorted, stype, ref, nlist, nneigh;
  DevBuf sorted_hs;  // per-atom half skins in cell-sorted order (contexts with skin weights)
  DevBuf hs2_dyn;    // (half skin)^2 of the CURRENT list per atom, original order: what the displacement test uses
  const void *skin_vel = nullptr;  // velocities of this replica while tmdhip_md_run is enqueuing (velocity-dependent skins)
  // chain skipping (see ListCheck): host-mapped words {progress, near[2], rebuilds[2]}, sequence number of the last
  // integrator kernel that ran the displacement test, and what the pair kernel of the current step publishes
  unsigned *hostpub = nullptr;
  unsigned seq = 0;
  bool seq_valid = false;
  bool prev_skipped = false;
  unsigned *pub_ptr = nullptr;
  unsigned pub_val = 0;
  int64_t chains_skipped = 0;
  int64_t steps_in_pair_launch = 0;
  DevBuf pos_alt;  // second position buffer of tmdhip_md_run's double-buffered integrator kernel
  // the MD step in the pair kernel's epilogue (FusedStep): the second cell-sorted copy (`sorted` is always the current
  // one: the two are swapped after every fused launch) and the static arguments, on the device and as last uploaded
  DevBuf sorted_alt, fused_dev;
  FusedStatic fused_host;
  bool fused_host_valid = false;
  DevBuf fsort;            // {pair force, launch number} per atom in cell-sorted order (fused launches)
  DevBuf fbond;            // bonded force of a fused launch's positions (heavy topologies), original atom order
  unsigned fused_gen = 0;  // number of the last fused launch
  DevBuf flags;  // int[F_COUNT], see the enum
  DevBuf extent;  // int[6]: keys of the coordinate extent of sorted_xyzq (extent_note)
  DevBuf paircount;  // unsigned long long
  void release() {
    for (DevBuf *b : {&cell_of, &slot, &order_tmp, &order, &inv, &count, &cell_start, &sorted, &stype, &ref, &sorted_hs, &hs2_dyn,
                      &nlist, &nneigh, &flags, &extent, &paircount, &pos_alt, &sorted_alt, &fused_dev, &fsort, &fbond})
      b->release();
  }
};

}  // namespace

struct tmdhip_ctx {
  tmdhip_nonbonded_desc d{};
  int real_size = 4;
  int algorithm = TMDHIP_ALGO_ALLPAIRS;
  double skin = 1.0;        // Verlet skin
  double rlist = 0;         // cutoff + skin
  DevBuf snap;              // pos, vel, forces at the entry of the last tmdhip_md_run (replay)
  size_t snap_bytes = 0;
  DevBuf sync_e;            // tmdhip_compute: per-term energies [R][NENERGY] on the device ...
  void *sync_host = nullptr;  // ... and their pinned host landing zone (+ the list flags of every replica)
  DevBuf obs_ke;              // tmdhip_md_observe: kinetic energies [R] ...
  void *obs_host = nullptr;   // ... and the pinned landing zone of energies, kinetic energies and list flags
  unsigned obs_seq = 0;       // sequence number of the last observe_publish_kernel
  DevBuf types, qs, tab, excl_off, excl_idx;
  // per-atom Verlet skins (tmdhip_set_skin_weights): half_skin[i] = w_i * skin / 2 and its square, original atom
  // order; empty = skin / 2 for every atom
  DevBuf half_skin, half_skin2;
  bool no_chain_skip_once = false;  // the next tmdhip_md_run enqueues every rebuild chain (repetition of a rewound batch)
  // velocity-dependent skins inside tmdhip_md_run (place_sorted_kernel): s_i = min(floor * static_i + time * |v_i|, cap)
  double vskin_floor = 0.8, vskin_time = 0, vskin_cap = 1.2, vskin_cap_len = 0;
  double mean_list_scale = 1;  // mean list length / length of a list at the largest pair radius (per-atom skins)
  DevBuf escratch;  // nreplicas x kEnergySlots x kEnergyStride doubles, all zero between calls (pair_math.h)
  DevBuf boxes;     // nreplicas x {box[3], 1/box[3]} for the replica-batched kernels
  DevBuf pos_alt_all;  // second position buffer [nreplicas][natoms][3] of the batched MD loop
  std::vector<double> boxes_host;  // what `boxes` currently holds
  int max_excl = 0;
  int nactive = 0x7fffffff;  // atoms with original index >= nactive get empty lists (tmdhip_update_atoms)
  int nexcl = 0;             // entries of the exclusion CSR
  std::vector<Replica> rep;
  // bonded part lives in bonded.hip
  void *bonded = nullptr;
  // timing of the dominant kernel
  bool timing = false;
  int timing_stride = 1;    // every n-th launch is timed
  int64_t timing_seen = 0;  // launches since timing was enabled
  int64_t timing_limit = 0; // stop after this many timed launches (0: no limit)
  int64_t timing_taken = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t events_used = 0;
  double timing_ms = 0;
  int64_t timing_launches = 0;
};

namespace tmd {
void bonded_release(tmdhip_ctx *ctx);  // bonded.hip
int bonded_inline_args(tmdhip_ctx *ctx, const double *box, BondedArgs<float> &A);   // bonded.hip
int bonded_inline_args(tmdhip_ctx *ctx, const double *box, BondedArgs<double> &A);  // bonded.hip
void *&ctx_bonded_slot(tmdhip_ctx *ctx) { return ctx->bonded; }
const tmdhip_nonbonded_desc &ctx_desc(const tmdhip_ctx *ctx) { return ctx->d; }
const void *ctx_scaled_charges(const tmdhip_ctx *ctx) { return ctx->qs.p; }
int ctx_nreplicas(const tmdhip_ctx *ctx) { return (int)ctx->rep.size(); }
double *ctx_energy_scratch(const tmdhip_ctx *ctx) { return ctx->escratch.as<double>(); }
int fold_energies(tmdhip_ctx *ctx, double *energies, hipStream_t st, int nrep) {
  hipLaunchKernelGGL(energy_fold_kernel, dim3(nrep), dim3(kEnergySlots), 0, st, ctx->escratch.as<double>(), energies);
  TMD_HIP(hipGetLastError());
  return 0;
}
// device copy of the replicas' boxes ({box[3], 1/box[3]} each, in the context's real type) for the
// replica-batched kernels; uploaded only when a box changes
const void *set_boxes(tmdhip_ctx *ctx, const double *box_host, hipStream_t st) {
  const size_t nrep = ctx->rep.size();
  bool same = ctx->boxes_host.size() == 3 * nrep;
  for (size_t k = 0; same && k < 3 * nrep; ++k) same = ctx->boxes_host[k] == box_host[k];
  if (same) return ctx->boxes.p;
  const bool f32 = ctx->d.dtype == TMDHIP_F32;
  std::vector<float> hf(6 * nrep);
  std::vector<double> hd(6 * nrep);
  for (size_t r = 0; r < nrep; ++r) {
    const double *b = box_host + 3 * r;
    const bool allzero = b[0] == 0 && b[1] == 0 && b[2] == 0;
    for (int k = 0; k < 3; ++k) {
      hf[6 * r + k] = (float)b[k];
      hd[6 * r + k] = b[k];
      hf[6 * r + 3 + k] = (!allzero && hf[6 * r + k] != 0.f) ? 1.0f / hf[6 * r + k] : 0.f;
      hd[6 * r + 3 + k] = (!allzero && b[k] != 0.0) ? 1.0 / b[k] : 0.0;
    }
  }
  const size_t bytes = 6 * nrep * (f32 ? sizeof(float) : sizeof(double));
  if (ctx->boxes.ensure(bytes)) return nullptr;
  // pageable source: the runtime stages it before returning, the vectors may die afterwards
  if (hipMemcpyAsync(ctx->boxes.p, f32 ? (const void *)hf.data() : (const void *)hd.data(), bytes,
                     hipMemcpyHostToDevice, st) != hipSuccess)
    return nullptr;
  if (hipStreamSynchronize(st) != hipSuccess) return nullptr;
  ctx->boxes_host.assign(box_host, box_host + 3 * nrep);
  return ctx->boxes.p;
}
}  // namespace tmd

namespace {

template <typename R>
R cutoff_r2max(double cutoff) {
  if (!(cutoff > 0)) return std::numeric_limits<R>::infinity();
  const R c = (R)cutoff;  // the reference compares against the cutoff cast to the tensor dtype
  R r2 = c * c;
  const R inf = std::numeric_limits<R>::infinity();
  while (std::sqrt(r2) <= c) r2 = std::nextafter(r2, inf);
  while (std::sqrt(r2) > c) r2 = std::nextafter(r2, (R)0);
  return r2;
}

template <typename R>
PairConsts<R> make_consts(const tmdhip_ctx *ctx, const double *box) {
  const auto &d = ctx->d;
  PairConsts<R> c;
  for (int k = 0; k < 3; ++k) {
    c.box[k] = (R)box[k];
    c.invbox[k] = c.box[k] != R(0) ? R(1) / c.box[k] : R(0);
  }
  const bool allzero = box[0] == 0 && box[1] == 0 && box[2] == 0;
  if (allzero)
    for (int k = 0; k < 3; ++k) c.invbox[k] = 0;
  c.r2max = cutoff_r2max<R>(d.cutoff);
  c.terms = d.terms;
  c.switch_on = (d.switch_dist > 0 && d.cutoff > 0) ? 1 : 0;
  c.switch_dist = (R)d.switch_dist;
  c.inv_switch_range = c.switch_on ? (R)(1.0 / (d.cutoff - d.switch_dist)) : R(0);
  c.switch_reference_mode = d.switch_mode == TMDHIP_SWITCH_REFERENCE;
  c.rfa = d.rfa ? 1 : 0;
  if (d.rfa) {
    const double eps = d.solvent_dielectric, den = 2 * eps + 1;
    c.krf = (R)((1.0 / (d.cutoff * d.cutoff * d.cutoff)) * (eps - 1) / den);
    c.crf = (R)((1.0 / d.cutoff) * (3 * eps) / den);
  } else {
    c.krf = c.crf = 0;
  }
  return c;
}

int pick_lpa(int n, int capacity) {
  if (const char *e = std::getenv("TMDHIP_LPA")) {  // tuning override: lanes per atom (power of two, 1..64)
    const int v = std::atoi(e);
    if (v >= 4 && v <= 64 && (v & (v - 1)) == 0) return v;
  }
  // (1) enough waves to hide list/gather latency: >= 8192 waves (32 per CU)
  int lpa = 1;
  while (lpa < 64 && (int64_t)n * lpa < 8192ll * 64) lpa <<= 1;
  // (2) list length: measured optimum LPA = 8 for water (440 entries per atom; 4 and 16 are 10 % slower)
  //     and 4 for liquid argon at 10^6 atoms (90 entries per atom; 1: +25 %, 2: +6 %, 8: +13 %);
  //     capacity = ~1.25 x the expected entries + 32
  const double per_lane = ((capacity - 32) / 1.25) / 44.0;
  int by_len = 4;
  while (by_len < 64 && (double)by_len * 1.4142 < per_lane) by_len <<= 1;
  return std::max(lpa, by_len);
}

// choose grid for the current box; returns false if the cell path cannot be used
bool plan_grid(const tmdhip_ctx *ctx, const double *box, const double *lo, const double *hi, Grid &g) {
  const bool periodic = !(box[0] == 0 && box[1] == 0 && box[2] == 0);
  g.periodic = periodic ? 1 : 0;
  double len[3];
  for (int k = 0; k < 3; ++k) {
    if (periodic) {
      if (!(box[k] > 0)) return false;
      len[k] = box[k];
      g.origin[k] = 0;
    } else {
      len[k] = std::max(hi[k] - lo[k], 1e-3);
      g.origin[k] = lo[k];
    }
  }
  // stencil half-width m: cell edge >= rlist/m.  m=3 (measured at C3: 29^3 cells of ~4 atoms) halves the
  // candidate volume but the build takes 345 us instead of 200: a build wave works on one cell and its
  // fixed costs (stencil set-up, staging the cell's atoms and exclusions, one candidate load per chunk)
  // are then amortised over 4 atoms instead of 14.  The kernel supports it (zreach), the planner stops at 2.
  int mmax = 2;
  if (const char *e = std::getenv("TMDHIP_STENCIL")) {  // tuning override: largest stencil half-width tried
    const int v = std::atoi(e);
    if (v >= 1 && v <= 3) mmax = v;
  }
  for (int m = mmax; m >= 1; --m) {
    bool ok = true;
    int nc[3];
    for (int k = 0; k < 3; ++k) {
      nc[k] = (int)std::floor(len[k] / (ctx->rlist / m));
      if (nc[k] < 1) nc[k] = 1;
      if (periodic && nc[k] < 2 * m + 1) ok = false;
      if (nc[k] > 1024) nc[k] = 1024;
    }
    if (!ok) continue;
    // a build wave works on one cell: at gas/liquid-argon densities half-width 2 leaves ~3 atoms per cell
    // (343k cells for the 10^6-atom LJ box) and the coarser grid is faster overall (179 vs 185 us/step)
    const double per_cell = (double)ctx->d.natoms / ((double)nc[0] * nc[1] * nc[2]);
    if (m == 3 && per_cell < 2.0) continue;
    if (m == 2 && per_cell < 4.0 && !std::getenv("TMDHIP_STENCIL")) {
      bool coarse_ok = true;
      for (int k = 0; k < 3; ++k) coarse_ok = coarse_ok && (!periodic || (int)std::floor(len[k] / ctx->rlist) >= 3);
      if (coarse_ok) continue;
    }
    g.m = m;
    double edge[3];
    for (int k = 0; k < 3; ++k) {
      g.nc[k] = nc[k];
      g.inv_edge[k] = nc[k] / len[k];
      edge[k] = len[k] / nc[k];
    }
    for (int ox = -3; ox <= 3; ++ox)
      for (int oy = -3; oy <= 3; ++oy) {
        int zr = -1;
        if (std::abs(ox) <= m && std::abs(oy) <= m) {
          const double gx = std::max(std::abs(ox) - 1, 0) * edge[0], gy = std::max(std::abs(oy) - 1, 0) * edge[1];
          for (int oz = 0; oz <= m; ++oz) {
            const double gz = std::max(oz - 1, 0) * edge[2];
            if (gx * gx + gy * gy + gz * gz <= ctx->rlist * ctx->rlist) zr = oz;
          }
          g.zreach[ox + m][oy + m] = (signed char)zr;
        }
      }
    return true;
  }
  return false;
}

constexpr size_t kRideMaxAtoms = 2048;  // bonded terms ride on the all-pairs launch up to this many atoms
constexpr int kForcesZeroed = 1 << 17;  // internal: the integrator kernel has already cleared `forces`

template <typename R>
int launch_allpairs(tmdhip_ctx *ctx, const void *pos, const double *box, void *forces, double *energies,
                    int flags, unsigned long long *paircount, hipStream_t st, int nrep = 1,
                    const BondedArgs<R> *bonded = nullptr) {
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
  int nsplit = std::max(1, std::min((n + 15) / 16, 1024 / std::max(nb * nrep, 1)));
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

// displacement test for the step that `rp.step` counts (see ListCheck)
template <typename R>
ListCheck<R> make_check(const tmdhip_ctx *ctx, Replica &rp) {
  ListCheck<R> k;
  k.ref = rp.ref.as<R>();
  k.hard2 = (R)(0.25 * ctx->skin * ctx->skin);
  k.hs2 = ctx->half_skin2.p ? (rp.hs2_dyn.p ? rp.hs2_dyn.as<R>() : ctx->half_skin2.as<R>()) : nullptr;
  k.near_host = nullptr;
  k.seq = 0;
  k.near_frac2 = R(0);
  k.skipped = 0;
  k.flags = rp.flags.as<int>();
  k.parity = (int)(rp.step & 1);
  k.ext = rp.extent.as<int>();
  return k;
}

// Launch with HIP events attached to the dispatch itself (hipExtLaunchKernel: start / stop are recorded by the
// kernel's own packet) when the launch is timed: a hipEventRecord in front of and behind the launch costs two extra
// barrier packets = 6.6 us of stream time per timed launch and puts the dispatch gap into the measurement.
template <typename K, typename... Args>
inline void launch_with_events(K kernel, dim3 grid, dim3 block, unsigned shmem, hipStream_t st, hipEvent_t e0,
                               hipEvent_t e1, Args... args) {
  if (e0 && e1) hipExtLaunchKernelGGL(kernel, grid, block, shmem, st, e0, e1, 0u, args...);
  else hipLaunchKernelGGL(kernel, grid, block, shmem, st, args...);
}

// a FUSED launch of the lean fp32 pair kernel (see FusedStep): device copy of the static part, this launch's part
struct FusedLaunch {
  const FusedStatic *fst;
  FusedStep step;
  bool langevin;
};

template <typename R, bool ENERGY>
int launch_list_pair(tmdhip_ctx *ctx, Replica &rp, const PairConsts<R> &c, R *f, int overwrite, double *energies,
                     unsigned long long *paircount, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr,
                     int lmode = 0, const FusedLaunch *fl = nullptr, bool fold = true) {
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
  if constexpr (std::is_same<R, float>::value) {
    // lean fp32 kernel: the entry's type field holds 32 LJ classes (n > 2^20: every iteration in its checked loop)
    if (fast && !paircount && (f || ENERGY || fl) && ctx->d.ntypes <= kEntryTypes) {
      const unsigned shfast = 0;
      const bool lj = c.terms & TMDHIP_TERM_LJ, el = c.terms & TMDHIP_TERM_ELECTROSTATICS;
#define TMD_LAUNCH_FAST_T(L, A, B, F)       \
  if (c.switch_on && A) {                  \
    TMD_LAUNCH_FAST_S(L, A, B, true, F);   \
  } else {                                 \
    TMD_LAUNCH_FAST_S(L, A, B, false, F);  \
  }
#define TMD_LAUNCH_FAST_S(L, A, B, S, F)                                                                            \
  launch_with_events(list_pair_fast_f32_kernel<L, A, B, ENERGY, S, F>, dim3(npair8 + (F ? fstep.nstep_blocks : 0)), dim3(TMD_FAST_THREADS), shfast, st, e0, e1, n, \
                     rp.sorted.as<R4>(), rp.stype.as<int>(), rp.order.as<int>(), ctx->d.ntypes, ctx->tab.as<R2>(), \
                     rp.nlist.as<unsigned>(), rp.nneigh.as<int>(), rp.lg.maxn, c, f, overwrite,                   \
                     ctx->escratch.as<double>(), rp.pub_ptr, rp.pub_val, rp.extent.as<int>(), rp.flags.as<int>(), \
                     lmode, F ? fl->fst : nullptr, fstep)
#define TMD_LAUNCH_FAST(L, F)               \
  if (lj && el) {                           \
    TMD_LAUNCH_FAST_T(L, true, true, F);    \
  } else if (lj) {                          \
    TMD_LAUNCH_FAST_T(L, true, false, F);   \
  } else {                                  \
    TMD_LAUNCH_FAST_T(L, false, true, F);   \
  }
      constexpr int wpb = TMD_FAST_THREADS / 64;
      const int npair8 = ((waves + wpb - 1) / wpb + 7) / 8 * 8;
      FusedStep fstep{};
      if (fl) {
        // step blocks behind the pair blocks (interior steps of tmdhip_md_run; fused_step_possible() has been asked):
        // one per 64 / (atoms of a pair block) pair blocks of an XCD's eighth
        fstep = fl->step;
        const int k = rp.lg.lpa * 64 / TMD_FAST_THREADS, g8 = npair8 / 8;
        const int units = (g8 + k - 1) / k;  // 64-atom units per XCD's eighth: a block with bonded records, a wave without
        fstep.nstep_blocks = 8 * (fstep.bonded == 1 ? units : (units + 3) / 4);
        if (rp.fsort.bytes < sizeof(R4) * (size_t)n) {
          TMD_TRY(rp.fsort.ensure(sizeof(R4) * (size_t)n));
          TMD_HIP(hipMemsetAsync(rp.fsort.p, 0, rp.fsort.bytes, st));  // launch number 0 = never written
          rp.fused_gen = 0;
        }
        if (rp.fused_gen == 0)  // test knob: start the launch counter just below its wrap-around
          if (const char *e = std::getenv("TMDHIP_DEBUG_FUSED_GEN0")) rp.fused_gen = (unsigned)std::strtoul(e, nullptr, 0);
        if (++rp.fused_gen == 0) rp.fused_gen = 1;  // (0 = "never written" in the records)
        fstep.gen = rp.fused_gen;
        fstep.fsort = rp.fsort.as<float4>();
        if constexpr (!ENERGY) {
#define TMD_LAUNCH_FUSED(L)    \
  if (fl->langevin) {          \
    TMD_LAUNCH_FAST(L, 2);     \
  } else {                     \
    TMD_LAUNCH_FAST(L, 1);     \
  }
          switch (rp.lg.lpa) {
            case 4: TMD_LAUNCH_FUSED(4); break;
            case 8: TMD_LAUNCH_FUSED(8); break;
            case 16: TMD_LAUNCH_FUSED(16); break;
            case 32: TMD_LAUNCH_FUSED(32); break;
            case 64: TMD_LAUNCH_FUSED(64); break;
            default: return fail("fused MD step: unsupported lanes-per-atom");
          }
#undef TMD_LAUNCH_FUSED
        } else {
          return fail("fused MD step with energies");
        }
      } else {
        switch (rp.lg.lpa) {  // (pick_lpa never returns less than 4)
          case 4: TMD_LAUNCH_FAST(4, 0); break;
          case 8: TMD_LAUNCH_FAST(8, 0); break;
          case 16: TMD_LAUNCH_FAST(16, 0); break;
          case 32: TMD_LAUNCH_FAST(32, 0); break;
          default: TMD_LAUNCH_FAST(64, 0); break;
        }
      }
#undef TMD_LAUNCH_FAST
#undef TMD_LAUNCH_FAST_T
#undef TMD_LAUNCH_FAST_S
      TMD_HIP(hipGetLastError());
      if (ENERGY && fold) TMD_TRY(tmd::fold_energies(ctx, energies, st, 1));
      return 0;
    }
  }
  if (fl) return fail("fused MD step: the context does not run the lean fp32 pair kernel");
  if constexpr (std::is_same<R, double>::value) {
    // lean fp64 kernel (same conditions as the fp32 one)
    if (fast && !paircount && (f || ENERGY) && ctx->d.ntypes <= kEntryTypes) {
      const unsigned shfast = 0;
      const bool lj = c.terms & TMDHIP_TERM_LJ, el = c.terms & TMDHIP_TERM_ELECTROSTATICS;
#define TMD_LAUNCH_FAST_T(L, A, B)       \
  if (c.switch_on && A) {               \
    TMD_LAUNCH_FAST_S(L, A, B, true);   \
  } else {                              \
    TMD_LAUNCH_FAST_S(L, A, B, false);  \
  }
#define TMD_LAUNCH_FAST_S(L, A, B, S)                                                                               \
  launch_with_events(list_pair_lean_f64_kernel<L, A, B, ENERGY, S>, dim3((blocks + 7) / 8 * 8), dim3(256), shfast, st, e0, e1, n, \
                     rp.sorted.as<R4>(), rp.stype.as<int>(), rp.order.as<int>(), ctx->d.ntypes, ctx->tab.as<R2>(), \
                     rp.nlist.as<unsigned>(), rp.nneigh.as<int>(), rp.lg.maxn, c, f, overwrite,                   \
                     ctx->escratch.as<double>(), rp.pub_ptr, rp.pub_val, rp.extent.as<int>())
#define TMD_LAUNCH_FAST(L)                  \
  if (lj && el) {                           \
    TMD_LAUNCH_FAST_T(L, true, true);       \
  } else if (lj) {                          \
    TMD_LAUNCH_FAST_T(L, true, false);      \
  } else {                                  \
    TMD_LAUNCH_FAST_T(L, false, true);      \
  }
      switch (rp.lg.lpa) {  // (pick_lpa never returns less than 4)
        case 4: TMD_LAUNCH_FAST(4); break;
        case 8: TMD_LAUNCH_FAST(8); break;
        case 16: TMD_LAUNCH_FAST(16); break;
        case 32: TMD_LAUNCH_FAST(32); break;
        default: TMD_LAUNCH_FAST(64); break;
      }
#undef TMD_LAUNCH_FAST
#undef TMD_LAUNCH_FAST_T
#undef TMD_LAUNCH_FAST_S
      TMD_HIP(hipGetLastError());
      if (ENERGY && fold) TMD_TRY(tmd::fold_energies(ctx, energies, st, 1));
      return 0;
    }
  }
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

template <typename R>
int alloc_replica(tmdhip_ctx *ctx, Replica &rp, int maxn) {
  using R4 = typename Vec<R>::T4;
  const int n = ctx->d.natoms;
  TMD_TRY(rp.cell_of.ensure(sizeof(int) * n));
  TMD_TRY(rp.slot.ensure(sizeof(int) * n));
  TMD_TRY(rp.order_tmp.ensure(sizeof(int) * n));
  TMD_TRY(rp.order.ensure(sizeof(int) * n));
  TMD_TRY(rp.inv.ensure(sizeof(int) * n));
  TMD_TRY(rp.sorted.ensure(sizeof(R4) * n));
  TMD_TRY(rp.sorted_alt.ensure(sizeof(R4) * n));
  TMD_TRY(rp.stype.ensure(sizeof(int) * n));
  if (ctx->half_skin.p) {
    TMD_TRY(rp.sorted_hs.ensure(ctx->real_size * (size_t)n));
    TMD_TRY(rp.hs2_dyn.ensure(ctx->real_size * (size_t)n));
  }
  TMD_TRY(rp.ref.ensure(sizeof(R) * 3 * n));
  TMD_TRY(rp.nneigh.ensure(sizeof(int) * n));
  // lanes per atom from the MEAN list length (capacities are sized for the longest lists); fixed once a list exists
  if (rp.lg.maxn == 0 || !rp.have_list) rp.lg.lpa = pick_lpa(n, (int)((maxn - 32) * ctx->mean_list_scale) + 32);
  rp.lg.apw = 64 / rp.lg.lpa;
  rp.lg.lpa_shift = 0;
  while ((1 << rp.lg.lpa_shift) < rp.lg.lpa) rp.lg.lpa_shift++;
  maxn = (maxn + 4 * rp.lg.lpa - 1) / (4 * rp.lg.lpa) * (4 * rp.lg.lpa);  // whole 16-byte words per lane
  rp.lg.maxn = maxn;
  const size_t groups = (n + rp.lg.apw - 1) / rp.lg.apw;
  if (groups * maxn * rp.lg.apw + 16 * 64 >= (size_t)1 << 30)
    return fail("neighbour list would exceed 2^30 entries per replica (32-bit row offsets)");
  // + 16 wave-rows of padding: the pair kernels prefetch up to three 4-iteration groups past a group's rows
  TMD_TRY(rp.nlist.ensure(sizeof(unsigned) * (groups * maxn * rp.lg.apw + 16 * 64)));
  return 0;
}

// TMDHIP_DEBUG_TIMELINE=1: every block of the list build records its entry / exit cycle counters (4 x u64 per block),
// read back with tmdhip_debug_build_timeline (tools/build_timeline.py).  Null otherwise: the kernel stores nothing.
static DevBuf g_dbg_timeline;
static size_t g_dbg_blocks = 0;
unsigned long long *debug_timeline_buffer(int blocks) {
  static const bool on = std::getenv("TMDHIP_DEBUG_TIMELINE") != nullptr;
  if (!on) return nullptr;
  if (g_dbg_timeline.ensure(sizeof(unsigned long long) * 4 * (size_t)blocks)) return nullptr;
  g_dbg_blocks = (size_t)blocks;
  return g_dbg_timeline.as<unsigned long long>();
}

// Enqueue: displacement check -> conditional rebuild chain -> gather.  `force` forces a rebuild.
// `prechecked`: the fused MD-step kernel already ran the displacement test of this step.
template <typename R>
int enqueue_list_update(tmdhip_ctx *ctx, Replica &rp, const R *pos, const PairConsts<R> &c, int force,
                        hipStream_t st, bool prechecked = false) {
  using R4 = typename Vec<R>::T4;
  const int n = ctx->d.natoms;
  const int parity = (int)(rp.step & 1);
  int *flags = rp.flags.as<int>();
  const int *flag = flags + F_REBUILD0 + parity;
  const int nb = (n + 255) / 256;
  if (!prechecked)
    hipLaunchKernelGGL((check_displacement_kernel<R>), dim3(nb), dim3(256), 0, st, n, pos, make_check<R>(ctx, rp), c,
                       force, rp.inv.as<int>(), ctx->qs.as<R>(), rp.sorted.as<R4>());
  PlaceArgs<R> P;
  P.cell_of = rp.cell_of.as<int>();
  P.cell_start = rp.cell_start.as<int>();
  P.order_tmp = rp.order_tmp.as<int>();
  P.pos = pos;
  P.qs = ctx->qs.as<R>();
  P.types = ctx->types.as<int>();
  P.order = rp.order.as<int>();
  P.inv = rp.inv.as<int>();
  P.sorted = rp.sorted.as<R4>();
  P.stype = rp.stype.as<int>();
  P.ref = rp.ref.as<R>();
  P.half_skin = ctx->half_skin.as<R>();
  P.sorted_hs = rp.sorted_hs.as<R>();
  P.vel = ctx->vskin_time > 0 ? (const R *)rp.skin_vel : nullptr;
  P.vs_floor = (R)ctx->vskin_floor;
  P.vs_time = (R)ctx->vskin_time;
  P.vs_cap = (R)ctx->vskin_cap_len;
  P.hs2_dyn = rp.hs2_dyn.as<R>();
  P.ext = rp.extent.as<int>();
  static const bool prep_small_on = !(std::getenv("TMDHIP_PREP_SMALL") && std::atoi(std::getenv("TMDHIP_PREP_SMALL")) == 0);
  if (prep_small_on && n <= kPrepSmallMaxAtoms && rp.ncell <= kPrepSmallMaxCells) {
    hipLaunchKernelGGL((prep_small_kernel<R>), dim3(1), dim3(1024), 0, st, n, pos, rp.grid, rp.ncell, rp.cell_of.as<int>(),
                       rp.slot.as<int>(), rp.cell_start.as<int>(), rp.order_tmp.as<int>(), P, flag);
  } else {
    hipLaunchKernelGGL((bin_count_kernel<R>), dim3(nb), dim3(256), 0, st, n, pos, rp.grid, rp.cell_of.as<int>(),
                       rp.slot.as<int>(), rp.count.as<int>(), flag);
    hipLaunchKernelGGL(scan_cells_kernel, dim3(1), dim3(1024), 0, st, rp.ncell, rp.count.as<int>(),
                       rp.cell_start.as<int>(), flag);
    hipLaunchKernelGGL(fill_cells_kernel, dim3(nb), dim3(256), 0, st, n, rp.cell_of.as<int>(), rp.slot.as<int>(),
                       rp.cell_start.as<int>(), rp.order_tmp.as<int>(), flag);
    hipLaunchKernelGGL((place_sorted_kernel<R>), dim3(nb), dim3(256), 0, st, n, P, flag);
  }
  const R rl = (R)ctx->rlist;
  constexpr int kMaxBuildBlocks = 16384;
  const bool wskin = ctx->half_skin.p != nullptr;
  // few cells: several blocks per cell (see build_list_kernel), so that ~2 000 waves are in flight
  int split = 1;
  if (const char *e = std::getenv("TMDHIP_BUILD_SPLIT")) split = std::max(1, std::min(std::atoi(e), 8));
  else if (rp.ncell <= 1100) split = 2;  // measured (water boxes of 5 184 / 12 288 / 41 472 atoms = 343 / 729 / 2 197 cells, us per
                                         // MD step at split 1, 2, 4): 29.7 27.7 (28-37) / 37.8 35.5 35.0 / 43.0 44.6 48.3
  if (rp.ncell > kMaxBuildBlocks) split = 1;
  auto launch_build = [&](auto kernel, int blocks) {
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), 0, st, n, rp.sorted.as<R4>(), rp.sorted_hs.as<R>(),
                       rp.stype.as<int>(), rp.order.as<int>(), rp.cell_start.as<int>(), rp.grid, c, rl * rl,
                       (R)ctx->d.cutoff, ctx->excl_off.as<int>(), ctx->excl_idx.as<int>(), rp.lg, rp.nlist.as<unsigned>(),
                       rp.nneigh.as<int>(), flags + F_MAXN, flag, rp.ncell, ctx->nactive, ctx->d.ntypes <= kEntryTypes,
                       debug_timeline_buffer(blocks), split);
  };
  if (rp.ncell <= kMaxBuildBlocks) {
    if (wskin) launch_build(build_list_kernel<R, false, true>, rp.ncell * split);
    else launch_build(build_list_kernel<R, false, false>, rp.ncell * split);
  } else {
    if (wskin) launch_build(build_list_kernel<R, true, true>, kMaxBuildBlocks);
    else launch_build(build_list_kernel<R, true, false>, kMaxBuildBlocks);
  }
  TMD_HIP(hipGetLastError());
  return 0;
}

constexpr int kPrechecked = 1 << 16;  // internal compute flag: displacement test already enqueued
constexpr int kSkipChain = 1 << 18;   // internal compute flag: the host leaves the rebuild chain out for this step
constexpr int kDeferFold = 1 << 20;  // internal compute flag: a bonded evaluation with energies follows and folds the scratch rows
constexpr int kViolationCheck = 1 << 19;  // internal compute flag: ... and the step's displacement test (epilogue of the
                                          // previous pair launch) did not know that: the pair launch looks itself
constexpr int kFallbackAllPairs = 77;  // compute_list: box too small for cells and algorithm = AUTO

template <typename R>
int compute_list(tmdhip_ctx *ctx, Replica &rp, const void *pos_v, const double *box, void *forces,
                 double *energies, int flags, hipStream_t st, const FusedLaunch *fused = nullptr) {
  const int n = ctx->d.natoms;
  const R *pos = (const R *)pos_v;
  const PairConsts<R> c = make_consts<R>(ctx, box);
  const bool box_changed = box[0] != rp.box[0] || box[1] != rp.box[1] || box[2] != rp.box[2];
  int force = 0;
  if (!rp.have_list || box_changed) {
    // (re)plan the grid — host-synchronising path, taken on the first call and when the box changes
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    const bool periodic = !(box[0] == 0 && box[1] == 0 && box[2] == 0);
    double volume;
    if (!periodic) {
      std::vector<R> h(3 * (size_t)n);
      TMD_HIP(hipMemcpyAsync(h.data(), pos, sizeof(R) * 3 * n, hipMemcpyDeviceToHost, st));
      TMD_HIP(hipStreamSynchronize(st));
      for (int k = 0; k < 3; ++k) lo[k] = 1e300, hi[k] = -1e300;
      for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
          lo[k] = std::min(lo[k], (double)h[3 * i + k]);
          hi[k] = std::max(hi[k], (double)h[3 * i + k]);
        }
      for (int k = 0; k < 3; ++k) lo[k] -= 1e-3, hi[k] += 1e-3;
      volume = std::max(hi[0] - lo[0], ctx->rlist) * std::max(hi[1] - lo[1], ctx->rlist) *
               std::max(hi[2] - lo[2], ctx->rlist);
    } else {
      volume = box[0] * box[1] * box[2];
    }
    if (!plan_grid(ctx, box, lo, hi, rp.grid)) {
      if (ctx->d.algorithm == TMDHIP_ALGO_AUTO) return kFallbackAllPairs;  // caller switches the context over
      return fail("cell list cannot be used for this box (fewer than 3 cells of cutoff+skin per edge); use "
                  "TMDHIP_ALGO_ALLPAIRS");
    }
    rp.ncell = rp.grid.nc[0] * rp.grid.nc[1] * rp.grid.nc[2];
    // new list, new extent (the forced rebuild below notes every position again)
    TMD_HIP(hipMemcpyAsync(rp.extent.p, kExtentEmpty, sizeof(kExtentEmpty), hipMemcpyHostToDevice, st));
    TMD_TRY(rp.count.ensure(sizeof(int) * (size_t)rp.ncell));
    TMD_TRY(rp.cell_start.ensure(sizeof(int) * ((size_t)rp.ncell + 1)));
    TMD_HIP(hipMemsetAsync(rp.count.p, 0, sizeof(int) * (size_t)rp.ncell, st));
    if (!rp.have_list) {
      const double dens = n / volume;
      int est = (int)(dens * 4.18879 * ctx->rlist * ctx->rlist * ctx->rlist * 1.3) + 32;
      est = std::min(est, std::max(n - 1, 1));
      TMD_TRY(alloc_replica<R>(ctx, rp, est));
    }
    for (int k = 0; k < 3; ++k) rp.box[k] = box[k];
    force = 1;
  }
  for (int attempt = 0; attempt < 8; ++attempt) {
    if (!force && (flags & kSkipChain)) {  // (the integrator kernel has run this step's test with `skipped` set)
      rp.step++;
      rp.chains_skipped++;
      break;
    }
    TMD_TRY(enqueue_list_update<R>(ctx, rp, pos, c, force, st, !force && (flags & kPrechecked)));
    rp.step++;
    if (!force) break;
    // forced builds are host-visible: size the list from the observed maximum so that later
    // device-side rebuilds have headroom (density fluctuations) without host involvement
    int h[F_COUNT];
    TMD_HIP(hipMemcpyAsync(h, rp.flags.p, sizeof(h), hipMemcpyDeviceToHost, st));
    TMD_HIP(hipStreamSynchronize(st));
    rp.host_rebuilds++;
    int want = (int)(h[F_MAXN] * 1.2) + 8;
    if (const char *e = std::getenv("TMDHIP_DEBUG_LIST_SLACK")) {
      // test knob: size the list for the observed maximum + N entries only, so that a later device-side
      // rebuild overflows and the replay path (tmdhip_md_restore) gets exercised
      want = h[F_MAXN] + std::max(std::atoi(e), 0);
      const int tight = (want + 4 * rp.lg.lpa - 1) / (4 * rp.lg.lpa) * (4 * rp.lg.lpa);
      if (!rp.have_list && h[F_MAXN] <= rp.lg.maxn && rp.lg.maxn > tight) {
        TMD_TRY(alloc_replica<R>(ctx, rp, tight));
        continue;  // rebuild in the tighter geometry
      }
    }
    if (h[F_MAXN] <= rp.lg.maxn && (rp.have_list || want <= rp.lg.maxn)) {
      rp.have_list = true;
      break;
    }
    rp.have_list = true;
    TMD_TRY(alloc_replica<R>(ctx, rp, std::max(want, rp.lg.maxn)));
  }
  R *f = (flags & TMDHIP_WANT_FORCES) ? (R *)forces : nullptr;
  unsigned long long *pc = nullptr;
  if (flags & TMDHIP_COUNT_PAIRS) {
    pc = rp.paircount.as<unsigned long long>();
    TMD_HIP(hipMemsetAsync(pc, 0, sizeof(unsigned long long), st));
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  // every `timing_stride`-th launch is bracketed by events: an event pair costs ~3 us of stream time,
  // so timing every launch would slow down the very loop being measured
  const int64_t seen = ctx->timing_seen++;
  const bool timed = ctx->timing && seen >= 0 && (seen % ctx->timing_stride) == 0 &&
                     (ctx->timing_limit == 0 || ctx->timing_taken < ctx->timing_limit);
  if (timed) ctx->timing_taken++;
  if (timed) {
    if (ctx->events_used >= 4096) TMD_TRY(tmdhip_timing_read(ctx, nullptr, nullptr, 0));
    if (ctx->events_used == ctx->events.size()) {
      hipEvent_t a, b;
      TMD_HIP(hipEventCreate(&a));
      TMD_HIP(hipEventCreate(&b));
      ctx->events.emplace_back(a, b);
    }
    e0 = ctx->events[ctx->events_used].first;
    e1 = ctx->events[ctx->events_used].second;
    ctx->events_used++;
  }
  const int overwrite = (flags & TMDHIP_OVERWRITE_FORCES) ? 1 : 0;
  // list duties of the pair launch's first thread: rp.step counts the NEXT step by now
  const int lmode = ((flags & kViolationCheck) ? kLmViolation : 0) | (((rp.step - 1) & 1) ? kLmParity : 0);
  FusedLaunch fl{};
  if (fused) {
    fl = *fused;
    fl.step.parity = (int)(rp.step & 1);
  }
  if (flags & TMDHIP_WANT_ENERGY)
    TMD_TRY((launch_list_pair<R, true>(ctx, rp, c, f, overwrite, energies, pc, st, e0, e1, lmode, nullptr, !(flags & kDeferFold))));
  else
    TMD_TRY((launch_list_pair<R, false>(ctx, rp, c, f, overwrite, energies, pc, st, e0, e1, lmode, fused ? &fl : nullptr)));
  if (pc) hipLaunchKernelGGL(halve_count_kernel, dim3(1), dim3(1), 0, st, pc);
  return 0;
}

template <typename R>
int upload_params(tmdhip_ctx *ctx) {
  using R2 = typename Vec<R>::T2;
  const auto &d = ctx->d;
  const int n = d.natoms, T = d.ntypes;
  std::vector<R> qs(n);
  const double s = std::sqrt(kElecFactor);
  const R *q = (const R *)d.charges_host;
  for (int i = 0; i < n; ++i) qs[i] = q ? (R)((double)q[i] * s) : R(0);
  TMD_TRY(ctx->qs.ensure(sizeof(R) * std::max(n, 1)));
  TMD_HIP(hipMemcpy(ctx->qs.p, qs.data(), sizeof(R) * n, hipMemcpyHostToDevice));
  std::vector<R2> tab((size_t)T * T);
  const R *A = (const R *)d.lj_A_host, *B = (const R *)d.lj_B_host;
  for (size_t k = 0; k < tab.size(); ++k) {
    tab[k].x = A ? A[k] : R(0);
    tab[k].y = B ? B[k] : R(0);
  }
  TMD_TRY(ctx->tab.ensure(sizeof(R2) * tab.size()));
  TMD_HIP(hipMemcpy(ctx->tab.p, tab.data(), sizeof(R2) * tab.size(), hipMemcpyHostToDevice));
  return 0;
}

}  // namespace

namespace {

template <typename R, bool SECOND, bool LANGEVIN, bool FIRST>
void launch_md_step(const MdStepArgs<R> &a, const PairConsts<R> &c, bool check, hipStream_t st, int nrep = 1) {
  const dim3 grid((a.n + 255) / 256, check ? 1 : nrep), block(256);
  if (check)
    hipLaunchKernelGGL((md_step_kernel<R, SECOND, LANGEVIN, FIRST, true>), grid, block, 0, st, a, c);
  else
    hipLaunchKernelGGL((md_step_kernel<R, SECOND, LANGEVIN, FIRST, false>), grid, block, 0, st, a, c);
}

template <typename R>
void launch_md_step_bonded(const MdStepArgs<R> &a, const PairConsts<R> &c, const BondedArgs<R> &A, bool langevin,
                           bool check, const R *boxes, int nrep, hipStream_t st) {
  const dim3 grid((kQuad * a.n + 255) / 256, check ? 1 : nrep), block(256);
#define TMD_MSB(L, C) hipLaunchKernelGGL((md_step_bonded_kernel<R, L, C>), grid, block, 0, st, a, c, A, boxes)
  if (langevin && check) TMD_MSB(true, true);
  else if (langevin) TMD_MSB(true, false);
  else if (check) TMD_MSB(false, true);
  else TMD_MSB(false, false);
#undef TMD_MSB
}

// ---- chain skipping (ListCheck) --------------------------------------------------------------------
constexpr int64_t kChainSkipMinEntries = 1'000'000;  // list slots from which the host paces itself behind the device.  (Round 2 gated this
                                                    // at 2e7 "because shorter pair kernels cannot hide the host"; measured in round 3 with the gate
                                                    // open, water boxes, us per MD step: 5 184 atoms 23.9 -> 22.6, 12 288 atoms 34.6 -> 28.6,
                                                    // 24 000 atoms 42.4 -> 36.8, bit-identical trajectories.)
constexpr double kChainSkipNear = 0.75;  // "near": beyond this fraction of the displacement limit (0.15 A of room at
                                         // skin 1.2: 2.2 x the largest per-step move seen in the water box, 9.5
                                         // standard deviations of a hydrogen's thermal velocity at 300 K)

// spin until the device has published sequence number `target` (wrap-around safe); false after 0.2 s
bool wait_published(volatile unsigned *hp, unsigned target) {
  if ((int)(hp[0] - target) >= 0) return true;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 1;; ++spins) {
    if ((int)(hp[0] - target) >= 0) return true;
    __builtin_ia32_pause();
    if ((spins & 4095u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) return false;
  }
}

// the static arguments of the fused step travel as a kernel argument (stream-ordered, no pinned staging, no host wait)
__global__ void fused_upload_kernel(FusedStatic v, FusedStatic *dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = v;
}

// can the pair launch of this replica integrate the next step itself?  (lean fp32 kernel, 4 .. 64 lanes per atom: a
// pair block's atoms fit one wave of a step block)
template <typename R>
bool fused_step_possible(const tmdhip_ctx *ctx, const Replica &rp, const PairConsts<R> &c) {
  if (!std::is_same<R, float>::value) return false;
  const char *e = std::getenv("TMDHIP_FUSED_STEP");  // (read per call: tests switch it within a process)
  if (e && std::atoi(e) == 0) return false;
  const bool only_lj_el = c.terms != 0 && (c.terms & ~(TMDHIP_TERM_LJ | TMDHIP_TERM_ELECTROSTATICS)) == 0;
  return only_lj_el && ctx->d.ntypes <= kEntryTypes && rp.lg.lpa >= 4 && rp.lg.lpa <= 64 && TMD_FAST_THREADS / rp.lg.lpa <= 64;
}

template <typename R>
int md_run(tmdhip_ctx *ctx, const tmdhip_md_desc *d, hipStream_t st) {
  using R4 = typename Vec<R>::T4;
  const int n = ctx->d.natoms;
  // TMDHIP_CHAIN_SKIP=0 switches the feature off; the two DEBUG knobs let a test reach the violation + rewind path
  // on a small box (minimum list size, "near" fraction: > 1 = an atom is never reported near its limit)
  const char *e_on = std::getenv("TMDHIP_CHAIN_SKIP"), *e_min = std::getenv("TMDHIP_DEBUG_CHAIN_MIN_ENTRIES"),
             *e_near = std::getenv("TMDHIP_DEBUG_CHAIN_NEAR");
  const bool chain_skip_on = !(e_on && std::atoi(e_on) == 0) && !ctx->no_chain_skip_once;
  const int64_t chain_min_entries = e_min ? std::atoll(e_min) : kChainSkipMinEntries;
  const double chain_near = e_near ? std::atof(e_near) : kChainSkipNear;
  ctx->no_chain_skip_once = false;
  bool pace_timed_out = false;  // the device did not report within wait_published's limit: no more waiting in this call
  const int nrep = (int)ctx->rep.size();
  const bool langevin = d->vcoeff_dev != nullptr;
  const size_t stride = (size_t)n * 3;
  MdStepArgs<R> a{};
  a.n = n;
  a.mass = (const R *)d->mass_dev;
  a.vcoeff = (const R *)d->vcoeff_dev;
  a.dt = (R)d->dt;
  a.half_dt = (R)(0.5 * d->dt);
  a.gamma = (R)d->gamma;
  a.seed = d->seed;
  a.qs = ctx->qs.as<R>();
  // where each replica's positions currently live (caller's tensor, or the context's second buffer while
  // the bonded force is evaluated inside the integrator kernel) and whether the bonded force of the
  // last evaluation is still owed to `forces`
  std::vector<R *> cur(nrep);
  std::vector<char> owed(nrep, 0);
  std::vector<char> stepped(nrep, 0);  // the previous pair launch of the replica has made this iteration's step (FusedStep)
  for (int r = 0; r < nrep; ++r) cur[r] = (R *)d->pos_dev + r * stride;
  // the same for the replica-batched all-pairs mode (all replicas move together)
  R *const home_all = (R *)d->pos_dev;
  R *bcur = home_all;
  bool bowed = false;

  for (int it = 0; it <= d->niter; ++it) {
    const bool first = it < d->niter, second = it > 0;
    a.noise_step = d->step0 + (uint64_t)(it > 0 ? it - 1 : 0);
    if (nrep > 1 && (ctx->algorithm == TMDHIP_ALGO_ALLPAIRS || ctx->d.terms == 0)) {
      // small systems are launch-bound: one launch of every kernel serves all replicas
      for (int r = 0; r < nrep; ++r) {  // leftovers of a cell-list context that fell back to all pairs in this call
        R *home = home_all + r * stride;
        if (owed[r]) {
          TMD_TRY(tmdhip_compute_bonded(ctx, r, cur[r], d->box_host + 3 * r, (R *)d->forces_dev + r * stride, nullptr,
                                        TMDHIP_WANT_FORCES, st));
          owed[r] = 0;
        }
        if (cur[r] != home) {
          TMD_HIP(hipMemcpyAsync(home, cur[r], sizeof(R) * stride, hipMemcpyDeviceToDevice, st));
          cur[r] = home;
        }
      }
      R *f = (R *)d->forces_dev;
      const PairConsts<R> c = make_consts<R>(ctx, d->box_host);
      a.pos_in = a.pos_out = bcur;
      a.vel = (R *)d->vel_dev;
      a.f = f;
      a.f_zero = (first && ctx->d.terms != 0) ? f : nullptr;  // saves the zero-fill launch of the all-pairs path
      a.row0 = 0;
      BondedArgs<R> A;
      if (bowed) {
        // (second && first) the bonded force of step it-1 is evaluated inside the integrator kernel from
        // the undrifted positions in bcur; the drifted ones go to the other buffer
        const bool ok = tmd::bonded_inline_args(ctx, d->box_host, A) == 1;
        const R *boxes = (const R *)tmd::set_boxes(ctx, d->box_host, st);
        if (!ok || !boxes) return fail("tmdhip_md_run: inline bonded state lost");
        R *other = bcur == home_all ? ctx->pos_alt_all.as<R>() : home_all;
        a.pos_out = other;
        launch_md_step_bonded<R>(a, c, A, langevin, false, boxes, nrep, st);
        bcur = other;
        bowed = false;
      } else if (second && first) {
        if (langevin) launch_md_step<R, true, true, true>(a, c, false, st, nrep);
        else launch_md_step<R, true, false, true>(a, c, false, st, nrep);
      } else if (first) {
        launch_md_step<R, false, false, true>(a, c, false, st, nrep);
      } else {
        if (langevin) launch_md_step<R, true, true, false>(a, c, false, st, nrep);
        else launch_md_step<R, true, false, false>(a, c, false, st, nrep);
      }
      TMD_HIP(hipGetLastError());
      if (!first) continue;
      R *pos = bcur;
      int flags_c = TMDHIP_WANT_FORCES;
      double *en = nullptr;
      if (it == d->niter - 1 && d->energies_dev) {
        flags_c |= TMDHIP_WANT_ENERGY;
        en = d->energies_dev;
      }
      const int bmode = tmd::bonded_inline_args(ctx, d->box_host, A);  // 0 none, 1 light, 2 heavy topology
      bool bonded_done = bmode == 0;
      if (ctx->d.terms != 0) {
        for (auto &rp : ctx->rep) rp.n_compute++;
        // heavy topologies, few atoms in total (launch-bound): the bonded terms ride on the all-pairs launch
        // (one wave per atom).  Measured: alanine dipeptide x1 39 -> 32 us/step, but x16 replicas 84 -> 94.
        const bool ride = bmode == 2 && (size_t)n * nrep <= kRideMaxAtoms;
        TMD_TRY(launch_allpairs<R>(ctx, pos, d->box_host, f, en, flags_c | TMDHIP_OVERWRITE_FORCES | kForcesZeroed,
                                   nullptr, st, nrep, ride ? &A : nullptr));
        bonded_done = bonded_done || ride;
      } else {
        TMD_HIP(hipMemsetAsync(f, 0, sizeof(R) * stride * nrep, st));
      }
      if (!bonded_done) {
        if (it + 1 < d->niter && bmode == 1) {
          TMD_TRY(ctx->pos_alt_all.ensure(sizeof(R) * stride * nrep));
          bowed = true;  // the next integrator kernel evaluates this step's bonded force itself
        } else {
          TMD_TRY(tmdhip_compute_bonded(ctx, TMDHIP_ALL_REPLICAS, pos, d->box_host, f, en, flags_c, st));
        }
      }
      continue;
    }
    for (int r = 0; r < nrep; ++r) {
      Replica &rp = ctx->rep[r];
      const double *box = d->box_host + 3 * r;
      R *home = (R *)d->pos_dev + r * stride, *f = (R *)d->forces_dev + r * stride;
      const PairConsts<R> c = make_consts<R>(ctx, box);
      bool list = ctx->algorithm == TMDHIP_ALGO_CELLLIST && ctx->d.terms != 0;
      // the displacement test can ride on the integrator kernel when a list exists for this box
      const bool check = first && list && rp.have_list && box[0] == rp.box[0] && box[1] == rp.box[1] && box[2] == rp.box[2];
      a.vel = (R *)d->vel_dev + r * stride;
      a.f = f;
      a.f_zero = (first && !list && ctx->d.terms != 0) ? f : nullptr;
      const bool zeroed = a.f_zero != nullptr;
      a.row0 = (uint64_t)r * (uint64_t)n;
      rp.skin_vel = a.vel;  // a rebuild in this step sizes the skins from the current velocities
      a.chk = make_check<R>(ctx, rp);
      // Chain skipping (ListCheck): on large lists the host stays one step behind the device — it waits until the
      // pair kernel of the previous step has started (45 us of kernel time are then still ahead of it) — and
      // leaves the rebuild chain out when no atom was near its limit in that step.  On the first step of a call only
      // if the caller says that nothing has moved since the previous one (tmdhip_md_desc::continuation; the report is
      // then the previous call's last), never in the repetition of a rewound batch.
      bool skip_chain = false;
      const bool pace = check && chain_skip_on && (int64_t)n * rp.lg.maxn >= chain_min_entries;
      if (pace) {
        if (!rp.hostpub) {
          TMD_HIP(hipHostMalloc((void **)&rp.hostpub, 8 * sizeof(unsigned), hipHostMallocMapped));
          for (int w = 0; w < 8; ++w) rp.hostpub[w] = 0u;
          rp.seq = 0;
          rp.seq_valid = false;
        }
        volatile unsigned *hp = rp.hostpub;
        // (the first step of a call: only when the caller vouches that nothing has moved since the previous call)
        const bool follows = it > 0 || d->continuation != 0;
        if (rp.seq_valid && follows && !pace_timed_out && !wait_published(hp, rp.seq)) pace_timed_out = true;
        if (rp.seq_valid && follows && !pace_timed_out) {
          // no chain when nobody was near its limit in the previous step — or when that step rebuilt the list
          // (with its chain in place: every displacement is one step old now)
          const bool near = hp[1 + (rp.seq & 1u)] == rp.seq, rebuilt = hp[3 + (rp.seq & 1u)] == rp.seq;
          skip_chain = !near || (rebuilt && !rp.prev_skipped);
        }
        rp.prev_skipped = skip_chain;
        rp.seq += 1;
        if (rp.seq == 0) rp.seq = 1;  // 0 = nothing published yet
        a.chk.near_host = rp.hostpub + 1 + (rp.seq & 1u);
        a.chk.seq = rp.seq;
        a.chk.near_frac2 = (R)(chain_near * chain_near);
        a.chk.skipped = skip_chain ? 1 : 0;
        rp.seq_valid = true;
        rp.pub_ptr = rp.hostpub;
        rp.pub_val = rp.seq;
      } else {
        rp.seq_valid = false;
        rp.pub_ptr = nullptr;
      }
      a.sorted = rp.sorted.as<R4>();
      a.inv = rp.inv.as<int>();
      a.pos_in = a.pos_out = cur[r];
      BondedArgs<R> A;
      std::memset(&A, 0, sizeof(A));
      const bool was_stepped = stepped[r] != 0;
      stepped[r] = 0;
      if (was_stepped) {
        // kicks, drift, displacement test and cell-sorted records of this iteration: done by the previous pair
        // launch's epilogue (cur[r] and rp.sorted already point at its output)
      } else if (owed[r]) {
        // second && first always holds here: the bonded force of step it-1 is evaluated from the
        // undrifted positions in cur[r], the drifted ones go to the other buffer
        if (tmd::bonded_inline_args(ctx, box, A) != 1) return fail("tmdhip_md_run: inline bonded state lost");
        R *other = cur[r] == home ? rp.pos_alt.as<R>() : home;
        a.pos_out = other;
        launch_md_step_bonded<R>(a, c, A, langevin, check, nullptr, 1, st);
        cur[r] = other;
        owed[r] = 0;
      } else if (second && first) {
        if (langevin) launch_md_step<R, true, true, true>(a, c, check, st);
        else launch_md_step<R, true, false, true>(a, c, check, st);
      } else if (first) {
        // Every fused step moves the positions to the other buffer; with an odd number of them ahead (all interior
        // steps of the call, if the first one can be fused) the drift of this first step goes to the second buffer,
        // so that the call ends in the caller's tensor without a copy.
        if (check && list && cur[r] == home && d->niter >= 2 && ((d->niter - 1) & 1) && fused_step_possible<R>(ctx, rp, c)) {
          TMD_TRY(rp.pos_alt.ensure(sizeof(R) * stride));
          a.pos_out = rp.pos_alt.as<R>();
          cur[r] = a.pos_out;
        }
        launch_md_step<R, false, false, true>(a, c, check, st);
      } else {
        if (langevin) launch_md_step<R, true, true, false>(a, c, check, st);
        else launch_md_step<R, true, false, false>(a, c, check, st);
      }
      TMD_HIP(hipGetLastError());
      if (!first) continue;
      R *pos = cur[r];
      // forces of step `it` (forces.py:122-319): nonbonded stores (list path) or accumulates into zeros
      int flags_c = TMDHIP_WANT_FORCES;
      double *en = nullptr;
      if (it == d->niter - 1 && d->energies_dev) {
        flags_c |= TMDHIP_WANT_ENERGY;
        en = d->energies_dev + (size_t)r * TMDHIP_NENERGY;
      }
      if (ctx->d.terms != 0) {
        rp.n_compute++;
        if (list) {
          // interior step on the lean fp32 kernel: the pair launch makes the next step itself (FusedStep)
          FusedLaunch fl{};
          bool fuse = false;
          if constexpr (std::is_same<R, float>::value) {
            const int bm = (check && it + 1 < d->niter && !en && fused_step_possible<R>(ctx, rp, c))
                               ? tmd::bonded_inline_args(ctx, box, A) : -1;
            if (bm >= 0) {
              if (bm == 2) {
                // heavy topology: the bonded force depends on the positions only — it is evaluated in front of the
                // pair launch into a buffer of its own and the step blocks add it (same values, same order as the
                // separate kernels: pair force stored, bonded force added, divided by the mass)
                TMD_TRY(rp.fbond.ensure(sizeof(R) * stride));
                TMD_TRY(tmdhip_compute_bonded(ctx, r, pos, box, rp.fbond.p, nullptr,
                                              TMDHIP_WANT_FORCES | TMDHIP_OVERWRITE_FORCES, st));
              }
              FusedStatic now;
              std::memset(&now, 0, sizeof(now));
              now.s.n = n;
              now.s.vel = a.vel;
              now.s.mass = a.mass;
              now.s.vcoeff = a.vcoeff;
              now.s.dt = a.dt;
              now.s.half_dt = a.half_dt;
              now.s.gamma = a.gamma;
              now.s.seed = a.seed;
              now.s.row0 = a.row0;
              now.s.qs = a.qs;
              now.s.inv = a.inv;
              now.s.chk.ref = a.chk.ref;
              now.s.chk.hard2 = a.chk.hard2;
              now.s.chk.hs2 = a.chk.hs2;
              now.s.chk.flags = a.chk.flags;
              now.s.chk.near_frac2 = (R)(chain_near * chain_near);
              now.s.chk.ext = a.chk.ext;
              if (bm == 1) std::memcpy(&now.A, &A, sizeof(A));
              now.has_bonded = bm;
              now.fbond = bm == 2 ? rp.fbond.as<float>() : nullptr;
              TMD_TRY(rp.fused_dev.ensure(sizeof(FusedStatic)));
              TMD_TRY(rp.pos_alt.ensure(sizeof(R) * stride));
              if (!rp.fused_host_valid || std::memcmp(&rp.fused_host, &now, sizeof(now)) != 0) {
                hipLaunchKernelGGL(fused_upload_kernel, dim3(1), dim3(64), 0, st, now, rp.fused_dev.as<FusedStatic>());
                std::memcpy(&rp.fused_host, &now, sizeof(now));
                rp.fused_host_valid = true;
              }
              fl.fst = rp.fused_dev.as<FusedStatic>();
              fl.langevin = langevin;
              fl.step.pos_in = pos;
              fl.step.pos_out = pos == home ? rp.pos_alt.as<R>() : home;
              fl.step.sorted_out = rp.sorted_alt.as<R4>();
              fl.step.noise_step = d->step0 + (uint64_t)it;
              fl.step.bonded = bm;
              if (pace) {  // the next iteration's sequence number (see the pacing above)
                unsigned nseq = rp.seq + 1;
                if (nseq == 0) nseq = 1;
                fl.step.seq = nseq;
                fl.step.near_host = rp.hostpub + 1 + (nseq & 1u);
              }
              fuse = true;
            }
          }
          const int rc = compute_list<R>(ctx, rp, pos, box, f, en,
                                         flags_c | TMDHIP_OVERWRITE_FORCES | (check ? kPrechecked : 0) |
                                             (skip_chain ? kSkipChain : 0) |
                                             (skip_chain && was_stepped ? kViolationCheck : 0) |
                                             (en && !fuse && rp.have_list && tmd::bonded_inline_args(ctx, box, A) != 0 ? kDeferFold : 0),
                                         st, fuse ? &fl : nullptr);
          rp.pub_ptr = nullptr;
          if (fuse && rc == 0) {
            if constexpr (std::is_same<R, float>::value) cur[r] = fl.step.pos_out;
            std::swap(rp.sorted, rp.sorted_alt);
            stepped[r] = 1;
            rp.steps_in_pair_launch++;
            continue;  // forces of this step never reach `forces`: the last step of the call is never fused
          }
          if (rc == kFallbackAllPairs) {
            ctx->algorithm = TMDHIP_ALGO_ALLPAIRS;
            list = false;
          } else if (rc != 0) {
            return rc;
          }
        }
        if (!list) {
          // heavy topology, small system: bonded terms in the same launch
          const bool ride = tmd::bonded_inline_args(ctx, box, A) == 2 && (size_t)n <= kRideMaxAtoms;
          TMD_TRY(launch_allpairs<R>(ctx, pos, box, f, en,
                                     flags_c | TMDHIP_OVERWRITE_FORCES | (zeroed ? kForcesZeroed : 0), nullptr, st, 1,
                                     ride ? &A : nullptr));
          if (ride) continue;  // forces (and energies) of this step are complete
        }
      } else {
        TMD_HIP(hipMemsetAsync(f, 0, sizeof(R) * stride, st));
      }
      // interior step: the next integrator kernel evaluates this step's bonded force itself
      // (md_step_bonded_kernel); `forces` holds the pair part until then.  (All-pairs contexts with several
      // replicas take the batched branch above from the next iteration on.)
      if (((list && rp.have_list) || (!list && nrep == 1)) && it + 1 < d->niter &&
          tmd::bonded_inline_args(ctx, box, A) == 1) {
        TMD_TRY(rp.pos_alt.ensure(sizeof(R) * stride));
        owed[r] = 1;
      } else {
        TMD_TRY(tmdhip_compute_bonded(ctx, r, pos, box, f, en, flags_c, st));
      }
    }
  }
  if (bcur != home_all)
    TMD_HIP(hipMemcpyAsync(home_all, bcur, sizeof(R) * stride * nrep, hipMemcpyDeviceToDevice, st));
  for (int r = 0; r < nrep; ++r) {
    R *home = (R *)d->pos_dev + r * stride;
    if (cur[r] != home) TMD_HIP(hipMemcpyAsync(home, cur[r], sizeof(R) * stride, hipMemcpyDeviceToDevice, st));
  }
  return 0;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int tmdhip_abi_version(void) { return TMDHIP_ABI_VERSION; }
const char *tmdhip_last_error(void) { return tmd::last_error().c_str(); }

int tmdhip_create(tmdhip_ctx **out, const tmdhip_nonbonded_desc *desc) {
  if (!out || !desc) return fail("tmdhip_create: null argument");
  if (desc->struct_size != (int32_t)sizeof(tmdhip_nonbonded_desc))
    return fail("tmdhip_create: tmdhip_nonbonded_desc size mismatch (ABI)");
  if (desc->natoms <= 0 || desc->ntypes <= 0 || desc->nreplicas <= 0)
    return fail("tmdhip_create: natoms, ntypes and nreplicas must be positive");
  if (desc->dtype != TMDHIP_F32 && desc->dtype != TMDHIP_F64) return fail("tmdhip_create: bad dtype");
  if (!desc->types_host || !desc->excl_offsets_host) return fail("tmdhip_create: types/exclusions missing");
  if ((desc->terms & TMDHIP_TERM_ELECTROSTATICS) && !desc->charges_host)
    return fail("tmdhip_create: electrostatics requested without charges");
  if ((desc->terms & (TMDHIP_TERM_LJ | TMDHIP_TERM_REPULSION)) && !desc->lj_A_host)
    return fail("tmdhip_create: LJ/repulsion requested without the A table");
  if ((desc->terms & (TMDHIP_TERM_LJ | TMDHIP_TERM_REPULSIONCG)) && !desc->lj_B_host)
    return fail("tmdhip_create: LJ/repulsioncg requested without the B table");
  if (desc->rfa && !(desc->cutoff > 0)) return fail("tmdhip_create: reaction field needs a cutoff");
  for (int i = 0; i < desc->natoms; ++i)
    if (desc->types_host[i] < 0 || desc->types_host[i] >= desc->ntypes)
      return fail("tmdhip_create: atom type index out of range");
  int ndev = 0;
  TMD_HIP(hipGetDeviceCount(&ndev));
  if (desc->device < 0 || desc->device >= ndev) return fail("tmdhip_create: no such HIP device");
  TMD_HIP(hipSetDevice(desc->device));

  tmdhip_ctx *ctx = new tmdhip_ctx();
  ctx->d = *desc;
  ctx->real_size = desc->dtype == TMDHIP_F32 ? 4 : 8;
  ctx->skin = desc->skin > 0 ? desc->skin : 1.2;  // measured optimum for the C3 water box (tools/time_kernels.py)
  ctx->rlist = desc->cutoff > 0 ? desc->cutoff + ctx->skin : 0;
  const int n = desc->natoms;
  auto cleanup = [&](int rc) {
    tmdhip_destroy(ctx);
    return rc;
  };
  if (ctx->types.ensure(sizeof(int) * n)) return cleanup(-1);
  if (hipMemcpy(ctx->types.p, desc->types_host, sizeof(int) * n, hipMemcpyHostToDevice) != hipSuccess)
    return cleanup(fail("tmdhip_create: copy of types failed"));
  const int nex = desc->excl_offsets_host[n];
  if (nex > 0 && !desc->excl_index_host) return cleanup(fail("tmdhip_create: exclusion indices missing"));
  for (int i = 0; i < n; ++i) {
    const int b = desc->excl_offsets_host[i], e = desc->excl_offsets_host[i + 1];
    if (e < b) return cleanup(fail("tmdhip_create: exclusion offsets not monotonic"));
    ctx->max_excl = std::max(ctx->max_excl, e - b);
    for (int k = b; k < e; ++k) {
      if (desc->excl_index_host[k] < 0 || desc->excl_index_host[k] >= n)
        return cleanup(fail("tmdhip_create: exclusion index out of range"));
      if (k > b && desc->excl_index_host[k] <= desc->excl_index_host[k - 1])
        return cleanup(fail("tmdhip_create: exclusion rows must be sorted and unique"));
    }
  }
  ctx->nexcl = nex;
  if (ctx->excl_off.ensure(sizeof(int) * (n + 1))) return cleanup(-1);
  if (ctx->excl_idx.ensure(sizeof(int) * std::max(nex, 1))) return cleanup(-1);
  (void)hipMemcpy(ctx->excl_off.p, desc->excl_offsets_host, sizeof(int) * (n + 1), hipMemcpyHostToDevice);
  if (nex) (void)hipMemcpy(ctx->excl_idx.p, desc->excl_index_host, sizeof(int) * nex, hipMemcpyHostToDevice);
  int rc = desc->dtype == TMDHIP_F32 ? upload_params<float>(ctx) : upload_params<double>(ctx);
  if (rc) return cleanup(rc);

  // algorithm choice: the list path needs a cutoff; without one every pair interacts anyway
  int algo = desc->algorithm;
  if (algo == TMDHIP_ALGO_AUTO) algo = (desc->cutoff > 0 && n >= 2048) ? TMDHIP_ALGO_CELLLIST : TMDHIP_ALGO_ALLPAIRS;
  if (algo == TMDHIP_ALGO_CELLLIST) {
    if (!(desc->cutoff > 0)) return cleanup(fail("tmdhip_create: the cell-list path needs a cutoff"));
    if (n >= (1 << 23)) return cleanup(fail("tmdhip_create: cell-list path supports < 2^23 atoms per context (23-bit slot field of a list entry)"));
    if (desc->ntypes > 256) return cleanup(fail("tmdhip_create: cell-list path supports <= 256 atom types"));
    const size_t tabbytes = (size_t)desc->ntypes * desc->ntypes * 2 * ctx->real_size;
    if (tabbytes > 64 * 1024) return cleanup(fail("tmdhip_create: LJ table does not fit in LDS (too many atom types)"));
  }
  ctx->algorithm = algo;
  const size_t esbytes = sizeof(double) * kEnergySlots * kEnergyStride * (size_t)desc->nreplicas;
  if (ctx->escratch.ensure(esbytes)) return cleanup(-1);
  (void)hipMemset(ctx->escratch.p, 0, esbytes);
  ctx->rep.resize(desc->nreplicas);
  for (auto &rp : ctx->rep) {
    if (rp.flags.ensure(sizeof(int) * F_COUNT)) return cleanup(-1);
    (void)hipMemset(rp.flags.p, 0, sizeof(int) * F_COUNT);
    if (rp.paircount.ensure(sizeof(unsigned long long))) return cleanup(-1);
    (void)hipMemset(rp.paircount.p, 0, sizeof(unsigned long long));
    if (rp.extent.ensure(sizeof(kExtentEmpty))) return cleanup(-1);
    (void)hipMemcpy(rp.extent.p, kExtentEmpty, sizeof(kExtentEmpty), hipMemcpyHostToDevice);
  }
  // host arrays are not referenced after create
  ctx->d.types_host = nullptr;
  ctx->d.charges_host = ctx->d.lj_A_host = ctx->d.lj_B_host = nullptr;
  ctx->d.excl_offsets_host = ctx->d.excl_index_host = nullptr;
  *out = ctx;
  return 0;
}

void tmdhip_destroy(tmdhip_ctx *ctx) {
  if (!ctx) return;
  for (auto &rp : ctx->rep) {
    if (rp.hostpub) (void)hipHostFree(rp.hostpub);
    rp.hostpub = nullptr;
    rp.release();
  }
  for (DevBuf *b : {&ctx->types, &ctx->qs, &ctx->tab, &ctx->excl_off, &ctx->excl_idx, &ctx->half_skin, &ctx->half_skin2, &ctx->escratch, &ctx->boxes, &ctx->pos_alt_all, &ctx->snap})
    b->release();
  for (auto &ev : ctx->events) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  if (ctx->sync_host) (void)hipHostFree(ctx->sync_host);
  if (ctx->obs_host) (void)hipHostFree(ctx->obs_host);
  ctx->sync_e.release();
  ctx->obs_ke.release();
  tmd::bonded_release(ctx);
  delete ctx;
}

int tmdhip_compute_nonbonded(tmdhip_ctx *ctx, int replica, const void *pos_dev, const double *box_host,
                             void *forces_dev, double *energies_dev, int flags, void *stream) {
  if (!ctx || !pos_dev || !box_host) return fail("tmdhip_compute_nonbonded: null argument");
  if (replica != TMDHIP_ALL_REPLICAS && (replica < 0 || replica >= (int)ctx->rep.size()))
    return fail("tmdhip_compute_nonbonded: bad replica index");
  if ((flags & TMDHIP_WANT_FORCES) && !forces_dev) return fail("tmdhip_compute_nonbonded: forces requested without a buffer");
  if ((flags & TMDHIP_WANT_ENERGY) && !energies_dev) return fail("tmdhip_compute_nonbonded: energies requested without a buffer");
  if (ctx->d.terms == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (replica == TMDHIP_ALL_REPLICAS) {
    const int nrep = (int)ctx->rep.size();
    const size_t esz = ctx->real_size, stride = (size_t)ctx->d.natoms * 3 * esz;
    if (nrep > 1 && ctx->algorithm == TMDHIP_ALGO_ALLPAIRS && !(flags & TMDHIP_COUNT_PAIRS)) {
      for (auto &rp : ctx->rep) rp.n_compute++;
      return ctx->d.dtype == TMDHIP_F32
                 ? launch_allpairs<float>(ctx, pos_dev, box_host, forces_dev, energies_dev, flags, nullptr, st, nrep)
                 : launch_allpairs<double>(ctx, pos_dev, box_host, forces_dev, energies_dev, flags, nullptr, st, nrep);
    }
    for (int r = 0; r < nrep; ++r)  // (a cell-list context may fall back to all pairs on the way: still correct)
      TMD_TRY(tmdhip_compute_nonbonded(ctx, r, (const char *)pos_dev + r * stride, box_host + 3 * r,
                                       forces_dev ? (char *)forces_dev + r * stride : nullptr,
                                       energies_dev ? energies_dev + (size_t)r * TMDHIP_NENERGY : nullptr, flags,
                                       stream));
    return 0;
  }
  Replica &rp = ctx->rep[replica];
  rp.n_compute++;
  const bool f32 = ctx->d.dtype == TMDHIP_F32;
  if (ctx->algorithm == TMDHIP_ALGO_CELLLIST) {
    const int rc = f32 ? compute_list<float>(ctx, rp, pos_dev, box_host, forces_dev, energies_dev, flags, st)
                       : compute_list<double>(ctx, rp, pos_dev, box_host, forces_dev, energies_dev, flags, st);
    if (rc != kFallbackAllPairs) return rc;
    ctx->algorithm = TMDHIP_ALGO_ALLPAIRS;  // AUTO and the box holds fewer than 3 cells per edge
  }
  unsigned long long *pc = nullptr;
  if (flags & TMDHIP_COUNT_PAIRS) {
    pc = rp.paircount.as<unsigned long long>();
    TMD_HIP(hipMemsetAsync(pc, 0, sizeof(unsigned long long), st));
  }
  return f32 ? launch_allpairs<float>(ctx, pos_dev, box_host, forces_dev, energies_dev, flags, pc, st)
             : launch_allpairs<double>(ctx, pos_dev, box_host, forces_dev, energies_dev, flags, pc, st);
}

int tmdhip_update_atoms(tmdhip_ctx *ctx, int natoms, const int32_t *types_host, const void *charges_host,
                        int nactive) {
  if (!ctx || !types_host) return fail("tmdhip_update_atoms: null argument");
  if (natoms <= 0 || natoms >= (1 << 23)) return fail("tmdhip_update_atoms: natoms out of range");
  if (ctx->nexcl != 0 || ctx->bonded) return fail("tmdhip_update_atoms: only for atomic systems (no exclusions, no bonded terms)");
  if ((ctx->d.terms & TMDHIP_TERM_ELECTROSTATICS) && !charges_host)
    return fail("tmdhip_update_atoms: electrostatics needs charges");
  for (int i = 0; i < natoms; ++i)
    if (types_host[i] < 0 || types_host[i] >= ctx->d.ntypes) return fail("tmdhip_update_atoms: atom type out of range");
  TMD_HIP(hipDeviceSynchronize());  // nothing may still be reading the old per-atom arrays
  const int n = natoms;
  ctx->d.natoms = n;
  ctx->nactive = nactive > 0 ? nactive : 0x7fffffff;
  TMD_TRY(ctx->types.ensure(sizeof(int) * n));
  TMD_HIP(hipMemcpy(ctx->types.p, types_host, sizeof(int) * n, hipMemcpyHostToDevice));
  const double s = std::sqrt(kElecFactor);
  TMD_TRY(ctx->qs.ensure((size_t)ctx->real_size * n));
  if (ctx->d.dtype == TMDHIP_F32) {
    std::vector<float> q(n);
    for (int i = 0; i < n; ++i) q[i] = charges_host ? (float)((double)((const float *)charges_host)[i] * s) : 0.f;
    TMD_HIP(hipMemcpy(ctx->qs.p, q.data(), sizeof(float) * n, hipMemcpyHostToDevice));
  } else {
    std::vector<double> q(n);
    for (int i = 0; i < n; ++i) q[i] = charges_host ? ((const double *)charges_host)[i] * s : 0.0;
    TMD_HIP(hipMemcpy(ctx->qs.p, q.data(), sizeof(double) * n, hipMemcpyHostToDevice));
  }
  TMD_TRY(ctx->excl_off.ensure(sizeof(int) * ((size_t)n + 1)));
  TMD_HIP(hipMemset(ctx->excl_off.p, 0, sizeof(int) * ((size_t)n + 1)));
  ctx->half_skin.release();  // per-atom skins belonged to the old atom set
  ctx->half_skin2.release();
  ctx->rlist = ctx->d.cutoff > 0 ? ctx->d.cutoff + ctx->skin : 0;
  ctx->mean_list_scale = 1;
  for (auto &rp : ctx->rep) {  // the next compute re-plans the grid, re-sizes the buffers and rebuilds
    rp.have_list = false;
    rp.lg.maxn = 0;
  }
  return 0;
}

int tmdhip_set_skin_weights(tmdhip_ctx *ctx, const void *weights_host) {
  if (!ctx) return fail("tmdhip_set_skin_weights: null ctx");
  TMD_HIP(hipDeviceSynchronize());  // nothing may still be reading the old skins
  const int n = ctx->d.natoms;
  for (auto &rp : ctx->rep) rp.have_list = false;  // the next compute re-plans and rebuilds
  if (!weights_host) {
    ctx->half_skin.release();
    ctx->half_skin2.release();
    ctx->rlist = ctx->d.cutoff > 0 ? ctx->d.cutoff + ctx->skin : 0;
    ctx->mean_list_scale = 1;
    return 0;
  }
  if (ctx->algorithm != TMDHIP_ALGO_CELLLIST) return fail("tmdhip_set_skin_weights: only for the cell-list path");
  double wmax = 0, wsum = 0;
  auto fill = [&](auto *w, auto &hs, auto &hs2) {
    for (int i = 0; i < n; ++i) {
      if (!(w[i] > 0) || !(w[i] <= 1)) return fail("tmdhip_set_skin_weights: weights must lie in (0, 1]");
      wmax = std::max(wmax, (double)w[i]);
      wsum += (double)w[i];
      hs[i] = (std::remove_reference_t<decltype(hs[0])>)(0.5 * ctx->skin * (double)w[i]);
      hs2[i] = hs[i] * hs[i];
    }
    return 0;
  };
  TMD_TRY(ctx->half_skin.ensure(ctx->real_size * (size_t)n));
  TMD_TRY(ctx->half_skin2.ensure(ctx->real_size * (size_t)n));
  if (ctx->d.dtype == TMDHIP_F32) {
    std::vector<float> hs(n), hs2(n);
    TMD_TRY(fill((const float *)weights_host, hs, hs2));
    TMD_HIP(hipMemcpy(ctx->half_skin.p, hs.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    TMD_HIP(hipMemcpy(ctx->half_skin2.p, hs2.data(), sizeof(float) * n, hipMemcpyHostToDevice));
  } else {
    std::vector<double> hs(n), hs2(n);
    TMD_TRY(fill((const double *)weights_host, hs, hs2));
    TMD_HIP(hipMemcpy(ctx->half_skin.p, hs.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    TMD_HIP(hipMemcpy(ctx->half_skin2.p, hs2.data(), sizeof(double) * n, hipMemcpyHostToDevice));
  }
  // velocity-dependent skins: TMDHIP_VSKIN = "floor,time_fs,cap" (defaults 0.8, 6, 1.2; "0" switches them off)
  ctx->vskin_floor = 0.8, ctx->vskin_cap = 1.2;
  double time_fs = 6.0;
  if (const char *e = std::getenv("TMDHIP_VSKIN")) {
    double a = 0, b = 0, cc = 0;
    const int got = std::sscanf(e, "%lf,%lf,%lf", &a, &b, &cc);
    if (got == 3 && a > 0 && a <= 1 && b >= 0 && cc >= 1 && cc <= 2) ctx->vskin_floor = a, time_fs = b, ctx->vskin_cap = cc;
    else if (got >= 1 && a == 0) time_fs = 0;
  }
  ctx->vskin_time = time_fs / 48.88821;  // internal time unit (integrator.py:4)
  if (!(ctx->vskin_time > 0)) ctx->vskin_cap = 1.0;
  ctx->vskin_cap_len = 0.5 * ctx->skin * wmax * ctx->vskin_cap;
  ctx->rlist = ctx->d.cutoff + 2.0 * ctx->vskin_cap_len;  // the largest pair radius: sizes the cells and the stencil reach
  ctx->mean_list_scale = std::pow((ctx->d.cutoff + ctx->skin * wsum / n) / ctx->rlist, 3.0);
  return 0;
}

int tmdhip_get_stats(tmdhip_ctx *ctx, int replica, tmdhip_stats *out) {
  if (!ctx || !out) return fail("tmdhip_get_stats: null argument");
  if (replica < 0 || replica >= (int)ctx->rep.size()) return fail("tmdhip_get_stats: bad replica index");
  Replica &rp = ctx->rep[replica];
  std::memset(out, 0, sizeof(*out));
  TMD_HIP(hipDeviceSynchronize());
  int h[F_COUNT] = {0};
  TMD_HIP(hipMemcpy(h, rp.flags.p, sizeof(h), hipMemcpyDeviceToHost));
  unsigned long long pc = 0;
  TMD_HIP(hipMemcpy(&pc, rp.paircount.p, sizeof(pc), hipMemcpyDeviceToHost));
  out->n_compute = rp.n_compute;
  out->n_rebuilds = h[F_NREBUILD];
  out->skin = ctx->skin;
  out->chains_skipped = rp.chains_skipped;
  out->steps_in_pair_launch = rp.steps_in_pair_launch;
  out->pairs_in_cutoff = (int64_t)pc;
  out->algorithm = ctx->algorithm;
  out->max_neighbours = rp.lg.maxn;
  out->overflow = (rp.have_list && h[F_MAXN] > rp.lg.maxn) ? h[F_MAXN] : 0;
  for (int k = 0; k < 3; ++k) out->ncell[k] = rp.grid.nc[k];
  if (rp.have_list) {
    std::vector<int> nn(ctx->d.natoms);
    TMD_HIP(hipMemcpy(nn.data(), rp.nneigh.p, sizeof(int) * nn.size(), hipMemcpyDeviceToHost));
    int64_t s = 0;
    for (int v : nn) s += v;
    out->list_entries = s;
  }
  return 0;
}

// verdict on the list flags of one replica (already on the host): 0 valid, 1 repeat the work, < 0 error
static int judge_flags(tmdhip_ctx *ctx, Replica &rp, const int *h, hipStream_t st) {
  if (h[F_VIOLATION]) {
    // an atom crossed its displacement limit in a step whose rebuild chain had been left out (ListCheck)
    (void)hipMemsetAsync(rp.flags.as<int>() + F_VIOLATION, 0, sizeof(int), st);
    ctx->no_chain_skip_once = true;
    rp.seq_valid = false;
    rp.box[0] = -1;  // re-plan + rebuild
    last_error() = "a neighbour list outlived its skin in a step without a rebuild chain (results since the last check are invalid)";
    if (h[F_MAXN] <= rp.lg.maxn) return 1;
  }
  if (h[F_MAXN] <= rp.lg.maxn) return 0;
  // a device-side rebuild truncated a list: grow the capacity and force a rebuild on the next call
  const int want = (int)(h[F_MAXN] * 1.25) + 16;
  const int rc = ctx->d.dtype == TMDHIP_F32 ? alloc_replica<float>(ctx, rp, want) : alloc_replica<double>(ctx, rp, want);
  if (rc) return rc;
  rp.box[0] = -1;  // forces the re-plan + rebuild path
  last_error() = "neighbour list overflowed (capacity grown, results since the last check are invalid)";
  return 1;
}

int tmdhip_check(tmdhip_ctx *ctx, int replica, void *stream) {
  if (!ctx) return fail("tmdhip_check: null ctx");
  if (replica < 0 || replica >= (int)ctx->rep.size()) return fail("tmdhip_check: bad replica index");
  Replica &rp = ctx->rep[replica];
  if (ctx->algorithm != TMDHIP_ALGO_CELLLIST || !rp.have_list) return 0;
  hipStream_t st = (hipStream_t)stream;
  int h[F_COUNT];
  TMD_HIP(hipMemcpyAsync(h, rp.flags.p, sizeof(h), hipMemcpyDeviceToHost, st));
  TMD_HIP(hipStreamSynchronize(st));
  return judge_flags(ctx, rp, h, st);
}

int tmdhip_compute(tmdhip_ctx *ctx, const void *pos_dev, const double *box_host, void *forces_dev,
                   double *energies_host, void *stream) {
  if (!ctx || !pos_dev || !box_host || !energies_host) return fail("tmdhip_compute: null argument");
  hipStream_t st = (hipStream_t)stream;
  const size_t nrep = ctx->rep.size();
  // landing zone: energies | list flags (padded to 8 bytes: the doubles behind them stay aligned) | kinetic-energy
  // slots | sequence word on a 64-byte line of its own
  const size_t ebytes = sizeof(double) * TMDHIP_NENERGY * nrep, fbytes = (sizeof(int) * F_COUNT * nrep + 7) / 8 * 8;
  const size_t seq_off = (ebytes + fbytes + sizeof(double) * nrep + 63) / 64 * 64;
  TMD_TRY(ctx->sync_e.ensure(ebytes));
  if (!ctx->sync_host) {
    TMD_HIP(hipHostMalloc(&ctx->sync_host, seq_off + 64, hipHostMallocMapped));
    std::memset(ctx->sync_host, 0, seq_off + 64);
  }
  double *he = (double *)ctx->sync_host;
  int *hf = (int *)((char *)ctx->sync_host + ebytes);
  double *e = ctx->sync_e.as<double>();
  TMD_HIP(hipMemsetAsync(e, 0, ebytes, st));
  int flags = TMDHIP_WANT_ENERGY;
  if (forces_dev) {
    flags |= TMDHIP_WANT_FORCES;
    if (ctx->d.terms == 0)  // no nonbonded kernel to store the forces: the bonded kernels add into zeros
      TMD_HIP(hipMemsetAsync(forces_dev, 0, (size_t)ctx->real_size * 3 * ctx->d.natoms * nrep, st));
  }
  TMD_TRY(tmdhip_compute_nonbonded(ctx, TMDHIP_ALL_REPLICAS, pos_dev, box_host, forces_dev, e,
                                   flags | (forces_dev ? TMDHIP_OVERWRITE_FORCES : 0), stream));
  TMD_TRY(tmdhip_compute_bonded(ctx, TMDHIP_ALL_REPLICAS, pos_dev, box_host, forces_dev, e, flags, stream));
  const bool lists = ctx->algorithm == TMDHIP_ALGO_CELLLIST;
  if (nrep <= 16) {  // results through host-mapped memory + a sequence word (see observe_publish_kernel)
    ObsFlagPtrs fp{};
    for (size_t r = 0; r < nrep; ++r) fp.p[r] = lists ? ctx->rep[r].flags.as<int>() : nullptr;
    double *hk = (double *)((char *)ctx->sync_host + ebytes + fbytes);
    volatile unsigned *hseq = (volatile unsigned *)((char *)ctx->sync_host + seq_off);
    if (++ctx->obs_seq == 0) ctx->obs_seq = 1;
    hipLaunchKernelGGL(observe_publish_kernel, dim3(1), dim3(128), 0, st, (int)nrep, e, (const double *)nullptr, fp, he, hk,
                       hf, const_cast<unsigned *>(hseq), ctx->obs_seq);
    TMD_HIP(hipGetLastError());
    TMD_TRY(wait_observed(hseq, ctx->obs_seq, st));  // the one host synchronisation of an energy evaluation
  } else {
    TMD_HIP(hipMemcpyAsync(he, e, ebytes, hipMemcpyDeviceToHost, st));
    if (lists)
      for (size_t r = 0; r < nrep; ++r)
        TMD_HIP(hipMemcpyAsync(hf + r * F_COUNT, ctx->rep[r].flags.p, sizeof(int) * F_COUNT, hipMemcpyDeviceToHost, st));
    TMD_HIP(hipStreamSynchronize(st));  // the one host synchronisation of an energy evaluation
  }
  int verdict = 0;
  if (lists && ctx->algorithm == TMDHIP_ALGO_CELLLIST)
    for (size_t r = 0; r < nrep; ++r)
      if (ctx->rep[r].have_list) {
        const int rc = judge_flags(ctx, ctx->rep[r], hf + r * F_COUNT, st);
        if (rc < 0) return rc;
        verdict |= rc;
      }
  std::memcpy(energies_host, he, ebytes);
  return verdict;
}

int tmdhip_md_run(tmdhip_ctx *ctx, const tmdhip_md_desc *desc, void *stream) {
  if (!ctx || !desc) return fail("tmdhip_md_run: null argument");
  if (desc->struct_size != (int32_t)sizeof(tmdhip_md_desc)) return fail("tmdhip_md_run: tmdhip_md_desc size mismatch (ABI)");
  if (desc->niter < 0) return fail("tmdhip_md_run: niter must be >= 0");
  if (!desc->pos_dev || !desc->vel_dev || !desc->forces_dev || !desc->mass_dev || !desc->box_host)
    return fail("tmdhip_md_run: null buffer");
  if (desc->niter == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int nzero = (int)(TMDHIP_NENERGY * ctx->rep.size());
  bool zeroed = desc->energies_dev == nullptr;
  if (ctx->algorithm == TMDHIP_ALGO_CELLLIST) {
    // state at entry, for tmdhip_md_restore (a truncated list is only detected after the batch)
    const size_t bytes = (size_t)ctx->real_size * 3 * ctx->d.natoms * ctx->rep.size();
    const size_t padded = (bytes + 15) / 16 * 16;
    TMD_TRY(ctx->snap.ensure(3 * padded));
    char *sn = ctx->snap.as<char>();
    const bool aligned = ((uintptr_t)desc->pos_dev | (uintptr_t)desc->vel_dev | (uintptr_t)desc->forces_dev) % 16 == 0 &&
                         bytes % 16 == 0;
    if (aligned) {
      const size_t n4 = bytes / 16;
      const bool fits = (size_t)nzero <= n4;
      hipLaunchKernelGGL(snapshot3_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, n4,
                         (const uint4 *)desc->pos_dev, (const uint4 *)desc->vel_dev, (const uint4 *)desc->forces_dev,
                         (uint4 *)sn, fits ? desc->energies_dev : nullptr, nzero);
      TMD_HIP(hipGetLastError());
      zeroed = zeroed || fits;
    } else {
      TMD_HIP(hipMemcpyAsync(sn, desc->pos_dev, bytes, hipMemcpyDeviceToDevice, st));
      TMD_HIP(hipMemcpyAsync(sn + padded, desc->vel_dev, bytes, hipMemcpyDeviceToDevice, st));
      TMD_HIP(hipMemcpyAsync(sn + 2 * padded, desc->forces_dev, bytes, hipMemcpyDeviceToDevice, st));
    }
    ctx->snap_bytes = bytes;
  }
  if (!zeroed) TMD_HIP(hipMemsetAsync(desc->energies_dev, 0, sizeof(double) * nzero, st));
  const int rc = ctx->d.dtype == TMDHIP_F32 ? md_run<float>(ctx, desc, st) : md_run<double>(ctx, desc, st);
  for (auto &rp : ctx->rep) rp.skin_vel = nullptr;  // rebuilds outside an MD run know no velocities: static skins
  return rc;
}

int tmdhip_md_observe(tmdhip_ctx *ctx, const void *vel_dev, const void *mass_dev, const double *energies_dev,
                      double *out_host, void *stream) {
  if (!ctx || !vel_dev || !mass_dev || !out_host) return fail("tmdhip_md_observe: null argument");
  hipStream_t st = (hipStream_t)stream;
  const size_t nrep = ctx->rep.size();
  const size_t ebytes = sizeof(double) * TMDHIP_NENERGY * nrep, kbytes = sizeof(double) * nrep;
  const size_t fbytes = sizeof(int) * F_COUNT * nrep;
  TMD_TRY(ctx->obs_ke.ensure(kbytes));
  if (!ctx->obs_host) {
    TMD_HIP(hipHostMalloc(&ctx->obs_host, ebytes + kbytes + fbytes + 64, hipHostMallocMapped));
    std::memset(ctx->obs_host, 0, ebytes + kbytes + fbytes + 64);
  }
  double *he = (double *)ctx->obs_host, *hk = he + TMDHIP_NENERGY * nrep;
  int *hf = (int *)((char *)ctx->obs_host + ebytes + kbytes);
  volatile unsigned *hseq = (volatile unsigned *)((char *)ctx->obs_host + ebytes + kbytes + fbytes + 32);
  TMD_TRY(tmdhip_kinetic_energy(ctx->d.dtype, (int64_t)nrep, ctx->d.natoms, vel_dev, mass_dev, ctx->obs_ke.as<double>(), stream));
  const bool lists = ctx->algorithm == TMDHIP_ALGO_CELLLIST;
  if (nrep <= 16) {
    ObsFlagPtrs fp{};
    for (size_t r = 0; r < nrep; ++r) fp.p[r] = lists ? ctx->rep[r].flags.as<int>() : nullptr;
    if (++ctx->obs_seq == 0) ctx->obs_seq = 1;
    hipLaunchKernelGGL(observe_publish_kernel, dim3(1), dim3(128), 0, st, (int)nrep, energies_dev, ctx->obs_ke.as<double>(),
                       fp, he, hk, hf, const_cast<unsigned *>(hseq), ctx->obs_seq);
    TMD_HIP(hipGetLastError());
    TMD_TRY(wait_observed(hseq, ctx->obs_seq, st));
  } else {
    if (energies_dev) TMD_HIP(hipMemcpyAsync(he, energies_dev, ebytes, hipMemcpyDeviceToHost, st));
    else std::memset(he, 0, ebytes);
    TMD_HIP(hipMemcpyAsync(hk, ctx->obs_ke.p, kbytes, hipMemcpyDeviceToHost, st));
    if (lists)
      for (size_t r = 0; r < nrep; ++r)
        TMD_HIP(hipMemcpyAsync(hf + r * F_COUNT, ctx->rep[r].flags.p, sizeof(int) * F_COUNT, hipMemcpyDeviceToHost, st));
    TMD_HIP(hipStreamSynchronize(st));
  }
  int verdict = 0;
  if (lists && ctx->algorithm == TMDHIP_ALGO_CELLLIST)
    for (size_t r = 0; r < nrep; ++r)
      if (ctx->rep[r].have_list) {
        const int rc = judge_flags(ctx, ctx->rep[r], hf + r * F_COUNT, st);
        if (rc < 0) return rc;
        verdict |= rc;
      }
  for (size_t r = 0; r < nrep; ++r) {
    for (int k = 0; k < TMDHIP_NENERGY; ++k) out_host[r * (TMDHIP_NENERGY + 1) + k] = energies_dev ? he[r * TMDHIP_NENERGY + k] : 0.0;
    out_host[r * (TMDHIP_NENERGY + 1) + TMDHIP_NENERGY] = hk[r];
  }
  return verdict;
}

int tmdhip_md_restore(tmdhip_ctx *ctx, const tmdhip_md_desc *desc, void *stream) {
  if (!ctx || !desc) return fail("tmdhip_md_restore: null argument");
  if (!desc->pos_dev || !desc->vel_dev || !desc->forces_dev) return fail("tmdhip_md_restore: null buffer");
  hipStream_t st = (hipStream_t)stream;
  const size_t bytes = (size_t)ctx->real_size * 3 * ctx->d.natoms * ctx->rep.size();
  if (ctx->snap_bytes != bytes || !ctx->snap.p) return fail("tmdhip_md_restore: no saved state of a matching tmdhip_md_run");
  const char *sn = ctx->snap.as<char>();
  const size_t padded = (bytes + 15) / 16 * 16;
  TMD_HIP(hipMemcpyAsync(desc->pos_dev, sn, bytes, hipMemcpyDeviceToDevice, st));
  TMD_HIP(hipMemcpyAsync(desc->vel_dev, sn + padded, bytes, hipMemcpyDeviceToDevice, st));
  TMD_HIP(hipMemcpyAsync(desc->forces_dev, sn + 2 * padded, bytes, hipMemcpyDeviceToDevice, st));
  for (auto &rp : ctx->rep) rp.box[0] = -1;  // re-plan + rebuild from the restored positions
  ctx->no_chain_skip_once = true;            // and no chain is left out while the batch is repeated
  return 0;
}

// debug only (not part of the ABI in include/tmdhip.h): copies the last list build's per-block timeline, returns blocks
int tmdhip_debug_build_timeline(void *out, size_t max_bytes) {
  if (!g_dbg_timeline.p || !out) return 0;
  const size_t bytes = std::min(max_bytes, sizeof(unsigned long long) * 4 * g_dbg_blocks);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpy(out, g_dbg_timeline.p, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int)g_dbg_blocks;
}

int tmdhip_invalidate_list(tmdhip_ctx *ctx, int replica) {
  if (!ctx) return fail("tmdhip_invalidate_list: null ctx");
  if (replica < 0 || replica >= (int)ctx->rep.size()) return fail("tmdhip_invalidate_list: bad replica index");
  ctx->rep[replica].box[0] = -1;  // next compute re-plans the grid and rebuilds (host-synchronising)
  return 0;
}

int tmdhip_timing_enable(tmdhip_ctx *ctx, int on) {
  if (!ctx) return fail("tmdhip_timing_enable: null ctx");
  ctx->timing = on != 0;
  ctx->timing_stride = (on & 0xFFFF) > 1 ? (on & 0xFFFF) : 1;
  ctx->timing_limit = (on >> 16) & 0xFFF;
  ctx->timing_taken = 0;
  ctx->timing_seen = -(int64_t)((on >> 28) & 7);  // the first launches are passed over
  // the events of the first launches are created here, not inside the region being timed (a hipEventCreate
  // costs ~10 us of host time: twenty of them in a 20-step run made the loop enqueue-bound)
  while (ctx->timing && ctx->events.size() < 192) {
    hipEvent_t a, b;
    TMD_HIP(hipEventCreate(&a));
    TMD_HIP(hipEventCreate(&b));
    ctx->events.emplace_back(a, b);
  }
  return 0;
}

int tmdhip_timing_read(tmdhip_ctx *ctx, double *pair_kernel_ms, int64_t *launches, int reset) {
  if (!ctx) return fail("tmdhip_timing_read: null ctx");
  for (size_t k = 0; k < ctx->events_used; ++k) {
    TMD_HIP(hipEventSynchronize(ctx->events[k].second));
    float ms = 0;
    TMD_HIP(hipEventElapsedTime(&ms, ctx->events[k].first, ctx->events[k].second));
    ctx->timing_ms += ms;
    ctx->timing_launches++;
  }
  ctx->events_used = 0;
  if (pair_kernel_ms) *pair_kernel_ms = ctx->timing_ms;
  if (launches) *launches = ctx->timing_launches;
  if (reset) {
    ctx->timing_ms = 0;
    ctx->timing_launches = 0;
  }
  return 0;
}

}  // extern "C"
