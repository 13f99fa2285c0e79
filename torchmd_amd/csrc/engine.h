// Internal header of libtmdhip (gfx950): the types the translation units of the nonbonded engine share — cell grid, list
// geometry and bookkeeping flags, the displacement test, the argument blocks of the MD-step code, the per-replica
// neighbour state and the context — plus the functions they call across each other.  Nothing here is part of the C ABI
// (include/tmdhip.h).
//
//   context.hip        context life cycle, grid planning, list (re)build orchestration, the C entry points
//   list_build.hip     K1 cell binning + K2 Verlet-list build
//   pair_generic.hip   K4 tiled all-pairs kernel, generic list pair kernel, dispatch between the pair kernels
//   pair_fast_f32.hip  K3f lean fp32 list pair kernel (+ the MD step inside the pair launch)
//   pair_lean_f64.hip  K3d lean fp64 list pair kernel
//   md_loop.hip        fused MD-step kernels, tmdhip_md_run / _observe / _restore
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
#pragma once

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
//   F_STEP_TIMEOUT  a step block of a fused pair launch gave up waiting for a force record (pair_fast_f32.hip): its
//                 atoms were NOT integrated; the caller rewinds and repeats the batch with the separate integrator kernel
//   F_RESERVED    unused (was the always-on flag of the look-ahead builds removed in round 5)
//   F_CELLCAP     a cell received more atoms than the member array of the two-launch binning holds (kCellCap): the list of
//                 this build is incomplete; the caller switches the replica to the four-launch binning and repeats
enum { F_REBUILD0 = 0, F_REBUILD1 = 1, F_MAXN = 2, F_NREBUILD = 3, F_VIOLATION = 4, F_STEP_TIMEOUT = 5, F_RESERVED = 6, F_CELLCAP = 7,
       F_COUNT = 8 };
// The list build runs in fp32 in fp64 contexts too (round 6: its fp64 form took 223 us per rebuild at C3 against 146): positions
// folded into the box and half skins are rounded to float (< 1e-5 A at a 100 A box) and every pair radius is widened by this
// margin, so the fp32 build lists a superset of what the fp64 criterion would — completeness is what the displacement test
// guarantees, the pair kernel applies the exact cutoff in the context's precision.
constexpr double kBuildMarginF64 = 2.0e-4;
constexpr int kCellCap = 64;             // members per cell of the two-launch binning
constexpr int kScanPlaceMaxCells = 12288;  // cells whose prefix a block of scan_place_kernel can hold in LDS (48 KB)

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
// The same for a whole wave at once (the placement kernels of a list build): after a re-plan the extent is empty and
// EVERY atom widens it — 98 304 x 6 atomicMin/Max on the same six words took 127-167 us of scan_place_kernel at C3 (round
// 5, profiles/r05_build_experiments.txt) against 13.7 us for the kernel on the rebuilds of an MD run.  Every lane of the wave
// must call (`has`: the lane holds an atom); nothing but six compares and a ballot while no lane is outside the bounds.
template <typename R>
__device__ __forceinline__ void extent_note_wave(int *ext, bool has, R x, R y, R z) {
  if (!ext) return;
  const int k[3] = {extent_key((float)x), extent_key((float)y), extent_key((float)z)};
  bool outside = false;
#pragma unroll
  for (int d = 0; d < 3; ++d) outside = outside || (has && (k[d] < ext[d] || k[d] > ext[3 + d]));
  if (__ballot(outside) == 0ull) return;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    int lo = has ? k[d] : 0x7FFFFFFF, hi = has ? k[d] : (int)0x80000000;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo = min(lo, __shfl_xor(lo, o, 64));
      hi = max(hi, __shfl_xor(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      if (lo < ext[d]) atomicMin(&ext[d], lo);
      if (hi > ext[3 + d]) atomicMax(&ext[3 + d], hi);
    }
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
  // the list build's own view of the atoms (round 5), written here so that the build kernel neither wraps positions nor
  // loads four arrays per candidate: bsorted[slot] = {position folded into [0, box), the atom's half skin of this list
  // (0 without per-atom skins)}, binfo[slot] = original index | LJ class << 27 (class 0 where the entries carry none)
  float4 *bsorted;  // (fp32 in either precision: the list criterion has the skin as slack, engine.h: kBuildMarginF64)
  int *binfo;
  R box[3], invbox[3];
  int type_in_entry;
  // padded rows (fp32): the two dummy records behind the last atom, rewritten with every build (null: none)
  typename Vec<R>::T4 *dummy_a, *dummy_b;  // sorted + n of the target copy, and of the replica's second copy (or null)
  R dummy_pos[2][3];
};

// list entry = type_j << 27 | j << 4 (j = cell-sorted slot, 23 bits): `entry & kEntryOffMask` is the byte offset
// of atom j's float4 record, `entry >> 24` the byte offset of type j in an 8-byte-stride LDS table row (for
// n <= 2^20; larger systems mask it).  Contexts with more than kEntryTypes LJ classes leave the type field 0 (kernels read stype[j]).
constexpr unsigned kEntryOffMask = 0x07FFFFF0u;  // byte offset of atom j's float4 record
constexpr int kEntryTypes = 32;                  // LJ classes that fit the entry's type field
constexpr int kEntryTypeShift = 27;
constexpr int kInfoIndexMask = 0x07FFFFFF;  // Replica::binfo: original index (< 2^23) below the LJ class field
constexpr float kR2Floor = 1.0e-2f;  // (0.1 A)^2: keeps 1/r^14 finite for the self entries that pad a column

// Padded list rows (Replica::pad_rows): the entry that fills the padding slots of atom `a`'s row — one of the two dummy
// records behind the last atom (slots n, n + 1 of the cell-sorted copy), the one further away under the minimum image.
// p = the atom's record (zero for the lanes past the last atom).
__device__ __forceinline__ void pad_dummy_positions(const PairConsts<float> &c, float (&d0)[3], float (&d1)[3]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const bool open = !(c.box[k] > 0.f);
    d0[k] = open ? 1.0e6f : 0.25f * c.box[k];
    d1[k] = open ? 1.0e6f : 0.75f * c.box[k];
  }
}
__device__ __forceinline__ unsigned pad_entry_for(const PairConsts<float> &c, int n, float px, float py, float pz) {
  float d0[3], d1[3];
  pad_dummy_positions(c, d0, d1);
  const float p[3] = {px, py, pz};
  float r0 = 0.f, r1 = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float e0 = min_image(p[k] - d0[k], c.box[k], c.invbox[k]), e1 = min_image(p[k] - d1[k], c.box[k], c.invbox[k]);
    r0 += e0 * e0;
    r1 += e1 * e1;
  }
  return (unsigned)(n + (r1 > r0 ? 1 : 0)) << 4;
}

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
  // the first kernel of a tmdhip_md_run call (first half step only) also saves the state at entry for tmdhip_md_restore —
  // what snapshot3_kernel does as a launch of its own — and clears the call's energy buffer (null: neither)
  R *snap_pos, *snap_vel, *snap_f;
  double *zero;
  int nzero;
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

// ---- the MD step inside the pair launch (FUSED variants of the lean kernels; pair_fast_f32.hip, pair_lean_f64.hip) ----
template <typename R>
struct FusedStepT {   // what does (kernel argument)
  const R *pos_in;  // positions of this launch's forces, original atom order (partners of the bonded terms)
  R *pos_out;       // drifted positions
  typename Vec<R>::T4 *sorted_out;   // their cell-sorted records
  typename Vec<R>::T4 *fsort;        // {pair force, launch number} per atom, cell-sorted order (pair blocks write, step blocks
                                     // watch); fp64: 32 bytes, the launch number in the low word of the fourth double
  unsigned gen;         // number of this launch (never 0): what the pair blocks write beside a force
  unsigned watch_gen;   // what the step blocks wait for: == gen (a test knob makes it differ, so that the wait times out)
  unsigned poll_limit;  // polls of a force record before a step block gives up (F_STEP_TIMEOUT)
  int bonded;           // FusedStatic::has_bonded (0 none, 1 inline records, 2 from FusedStatic::fbond)
  int nstep_blocks;     // step blocks at the end of the grid (a multiple of 8, like the pair blocks)
  uint64_t noise_step;
  unsigned *near_host;  // chain skipping: report words of the NEXT step's displacement test (null: none)
  unsigned seq;
  int parity;           // of the next step
};
using FusedStep = FusedStepT<float>;
constexpr int kAuxDeviceScope = 16;  // sc1 of a gfx942/950 buffer access: coherent across the XCDs' L2s
constexpr int kAuxVolatile = (int)0x80000000;  // bit 31 of a raw-buffer intrinsic's aux operand: a volatile access (the
                                               // compiler must neither hoist it out of a loop nor merge two of them)
// lmode bits (list bookkeeping duties of the launch's first thread)
constexpr int kLmViolation = 1;  // the chain of this step was left out and its displacement test ran in the previous
                                 // launch's epilogue, which could not know that: a rebuild request found now = F_VIOLATION
constexpr int kLmParity = 2;     // parity of this step
constexpr int kLmStream = 8;     // the list does not fit the Infinity Cache: stream it with the non-temporal hint (pair_fast_f32.hip)
constexpr int kLmPadded = 4;     // Replica::pad_rows: the dummy records exist; a wave group whose padgen word equals the rebuild count is padded
constexpr int kFastThreads = 256;  // threads of a block of the lean fp32 pair kernel (step blocks are four waves)

// ---- the step in the lean pair kernels' launch (see FusedStepT above) -------------------------------------------
template <typename R>
struct FusedStaticT {
  MdStepArgs<R> s;  // per-launch fields (pos_in/out, sorted, noise_step, chk.near_host/seq/parity) come from FusedStepT
  BondedArgs<R> A;
  int has_bonded;  // 1: light topology, the atoms' bonded records are evaluated here (md_step_bonded_kernel's job);
                   // 2: heavy topology, the bonded force of this launch's positions is in `fbond` (bonded_wave_kernel
                   // ran in front of the launch: it depends on the positions only)
  const R *fbond;  // [3N], original atom order
  int nactive;     // atoms with original index >= nactive are not integrated (the halo rows of a brick); INT_MAX otherwise
  // Brick of a domain decomposition (tmdhip_dd_run; all null otherwise): the step blocks also keep the migration
  // trigger's displacement maximum (against the positions at the last migration) and write the atom's rows of the
  // outgoing halo messages (per-atom index of the send list), i.e. all of dd_own_kernel's work (domain.hip)
  const R *dd_ref;
  unsigned *dd_disp2;
  const int *dd_csr_off, *dd_csr_row;
  const R *dd_shift;
  R *dd_out;
};
using FusedStatic = FusedStaticT<float>;

// ---- the replicas of a cell-list context in one launch (list_pair_fast_f32_batch_kernel, pair_fast_kernel.h) --------------
// BatchRep: one replica's buffers, device table [nreplicas] uploaded when an entry changes.  The two position buffers and the
// two cell-sorted copies alternate from one fused launch to the next: the table holds both, BatchLaunch::bits says which is
// current.
struct BatchRep {
  float4 *sorted[2];
  float *pos[2];
  const int *stype, *order;
  const unsigned *nlist;
  const int *nneigh;
  float *forces;       // the caller's force array of this replica (FINAL step blocks)
  double *escratch;    // this replica's rows of the energy scratch
  const int *ext;
  int *lflags;
  int *padgen;
  const FusedStatic *fst;
  float4 *fsort;
  unsigned *hostpub;   // host-mapped words of the pacing host: [0] progress, [1 + (seq & 1)] near reports (null: none)
  float box[3], invbox[3];
  int maxn;
  int pad_;
};
constexpr int kBatchMax = 16;  // replicas per batched launch (larger contexts: several launches)
// BatchLaunch::bits of a replica: the low bits are the launch's lmode (kLm*)
constexpr unsigned kBlLmodeMask = 0xFFu;
constexpr unsigned kBlSortedCur = 1u << 8;    // index of the current cell-sorted copy in BatchRep::sorted
constexpr unsigned kBlPosCur = 1u << 9;       // index of the current position buffer in BatchRep::pos
constexpr unsigned kBlNextParity = 1u << 10;  // parity of the NEXT step (FusedStepT::parity)
constexpr unsigned kBlReports = 1u << 11;     // the step blocks report to the pacing host (FusedStepT::near_host)
constexpr unsigned kBlPublish = 1u << 12;     // the replica's first block publishes `pub` as progress
struct BatchLaunch {
  int nrep, pair_blocks, step_blocks, bonded;
  unsigned poll_limit, pad_;
  uint64_t noise_step;
  unsigned gen[kBatchMax], seq[kBatchMax], pub[kBatchMax], bits[kBatchMax];
};

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
    // TMDHIP_DEBUG_POISON=1 (tests): fresh device memory is usually zero on an idle box and arbitrary on a busy one; fill
    // every new buffer with 0xFF bytes (NaN as a real, -1 as an index, the largest launch number) so that a read of
    // something nobody wrote shows on every box
    // (2: zeros instead — the same extra synchronisations without the garbage, to tell the two apart)
    static const int poison = [] { const char *e = std::getenv("TMDHIP_DEBUG_POISON"); return e ? std::atoi(e) : 0; }();
    if (poison) {
      TMD_HIP(hipMemset(p, poison == 2 ? 0x00 : 0xFF, need));
      TMD_HIP(hipStreamSynchronize(nullptr));  // (a memset of device memory need not be complete when the call returns)
    }
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
  int maxn_keep = 0;  // list capacity before the last atom swap (tmdhip_update_atoms / tmdhip_dd_migrate): a floor for the next
                      // estimate, so that a swap between atom sets of the same density does not build its first list twice
  int64_t host_rebuilds = 0;
  DevBuf cell_of, slot, order_tmp, order, inv, count, cell_start, sorted, stype, ref, nlist, nneigh;
  DevBuf padgen;  // int32 per wave group of the list: the rebuild count (flags[F_NREBUILD]) at which the group's rows were padded
  DevBuf members;  // two-launch binning: int32[ncell x kCellCap], the atoms of every cell in arrival order
  bool cell_cap_fallback = false;  // a cell overflowed `members` once: this replica bins with the four launches
  DevBuf sorted_hs;  // per-atom half skins in cell-sorted order (contexts with skin weights)
  DevBuf bsorted, binfo;  // the build kernel's records (PlaceArgs): wrapped position + half skin, original index | LJ class
  DevBuf hs2_dyn;    // (half skin)^2 of the CURRENT list per atom, original order: what the displacement test uses
  const void *skin_vel = nullptr;  // velocities of this replica while tmdhip_md_run is enqueuing (velocity-dependent skins)
  // chain skipping (see ListCheck): host-mapped words {progress, near[2], rebuilds[2]}, sequence number of the last
  // integrator kernel that ran the displacement test, and what the pair kernel of the current step publishes
  unsigned *hostpub = nullptr;
  unsigned seq = 0;
  bool seq_valid = false;
  bool prev_skipped = false;
  // plain evaluations through tmdhip_compute (which reads the flags back before it returns) leave the chain out the same way:
  // the report of the PREVIOUS evaluation says whether anybody was near a limit (enqueue_list_update, `speculate`)
  bool spec_valid = false;
  int spec_backoff = 0;  // evaluations that keep their chain after a wrong guess
  unsigned *pub_ptr = nullptr;
  unsigned pub_val = 0;
  int64_t chains_skipped = 0;
  int64_t steps_in_pair_launch = 0;
  DevBuf pos_alt;  // second position buffer of tmdhip_md_run's double-buffered integrator kernel
  // the MD step in the pair kernel's epilogue (FusedStep): the second cell-sorted copy (`sorted` is always the current
  // one: the two are swapped after every fused launch) and the static arguments, on the device and as last uploaded
  DevBuf sorted_alt, fused_dev;
  alignas(16) unsigned char fused_host[sizeof(FusedStaticT<double>)];  // last upload (a FusedStaticT of the context's precision)
  bool fused_host_valid = false;
  DevBuf fsort;            // {pair force, launch number} per atom in cell-sorted order (fused launches)
  DevBuf fbond;            // bonded force of a fused launch's positions (heavy topologies), original atom order
  unsigned fused_gen = 0;  // number of the last fused launch
  int64_t fused_launches = 0;  // fused launches of this replica (test knob TMDHIP_DEBUG_STEP_TIMEOUT counts them)
  // Padded list rows (pad_entry_for; fp32 contexts of <= 2^20 - 2 atoms whose box keeps the dummy records out of
  // reach): the padding slots of every wave group hold a harmless entry, the lean fp32 pair kernel runs all of a wave's
  // groups in its unchecked loop (kLmPadded).  The pair waves write the padding themselves on their first launch after
  // a list build (`padgen`).  Decided when the grid is planned (a forced rebuild follows, which writes the dummies).
  bool pad_rows = false;
  DevBuf flags;  // int[F_COUNT], see the enum
  DevBuf extent;  // int[6]: keys of the coordinate extent of sorted_xyzq (extent_note)
  DevBuf paircount;  // unsigned long long
  void release() {
    for (DevBuf *b : {&cell_of, &slot, &order_tmp, &order, &inv, &count, &cell_start, &sorted, &stype, &ref, &sorted_hs, &hs2_dyn,
                      &nlist, &nneigh, &padgen, &members, &flags, &extent, &paircount, &pos_alt, &sorted_alt, &fused_dev, &fsort, &fbond,
                      &bsorted, &binfo})
      b->release();
  }
};

}  // namespace tmd

struct tmdhip_ctx {
  tmdhip_nonbonded_desc d{};
  int real_size = 4;
  int algorithm = TMDHIP_ALGO_ALLPAIRS;
  double skin = 1.0;        // Verlet skin
  double rlist = 0;         // cutoff + skin
  tmd::DevBuf snap;              // pos, vel, forces at the entry of the last tmdhip_md_run (replay)
  size_t snap_bytes = 0;
  bool snap_pending = false;     // tmdhip_md_run has left the snapshot (and the zeroing of `snap_zero`) to md_run's first kernel
  double *snap_zero = nullptr;
  int snap_nzero = 0;
  tmd::DevBuf sync_e;            // tmdhip_compute: per-term energies [R][NENERGY] on the device ...
  void *sync_host = nullptr;  // ... and their pinned host landing zone (+ the list flags of every replica)
  tmd::DevBuf obs_ke;              // tmdhip_md_observe: kinetic energies [R] ...
  void *obs_host = nullptr;   // ... and the pinned landing zone of energies, kinetic energies and list flags
  unsigned obs_seq = 0;       // sequence number of the last observe_publish_kernel
  // the last tmdhip_md_run reported its results itself (final_fold_publish_kernel): sequence number it wrote (0: it did not)
  // and the energy buffer the report is of
  unsigned run_published_seq = 0;
  const double *run_published_energies = nullptr;
  tmd::DevBuf types, qs, tab, excl_off, excl_idx;
  // per-atom Verlet skins (tmdhip_set_skin_weights): half_skin[i] = w_i * skin / 2 and its square, original atom
  // order; empty = skin / 2 for every atom
  tmd::DevBuf half_skin, half_skin2;
  bool no_chain_skip_once = false;  // the next tmdhip_md_run enqueues every rebuild chain (repetition of a rewound batch)
  // Fail-safe of the fused pair + step launch (F_STEP_TIMEOUT): the batch that timed out is repeated with the separate
  // integrator kernel (`no_fused_once`, consumed by the next tmdhip_md_run into `fused_off_call`); a context that timed
  // out twice stops fusing for good (`fused_disabled`): the hand-over's in-order-dispatch assumption does not hold here.
  bool no_fused_once = false, fused_off_call = false, fused_disabled = false;
  int64_t fused_step_timeouts = 0;
  // the last step of the last tmdhip_md_run was made by FINAL step blocks (md_step.h): the kinetic energy of the velocities
  // `ke_from_run` points at is in obs_ke already (tmdhip_md_observe then launches no kinetic-energy kernel)
  const void *ke_from_run = nullptr, *ke_from_run_mass = nullptr;
  int64_t final_steps_in_pair_launch = 0;
  // velocity-dependent skins inside tmdhip_md_run (place_sorted_kernel): s_i = min(floor * static_i + time * |v_i|, cap)
  double vskin_floor = 0.8, vskin_time = 0, vskin_cap = 1.2, vskin_cap_len = 0;
  double mean_list_scale = 1;  // mean list length / length of a list at the largest pair radius (per-atom skins)
  tmd::DevBuf escratch;  // nreplicas x kEnergySlots x kEnergyStride doubles, all zero between calls (pair_math.h)
  tmd::DevBuf boxes;     // nreplicas x {box[3], 1/box[3]} for the replica-batched kernels
  tmd::DevBuf pos_alt_all;  // second position buffer [nreplicas][natoms][3] of the batched MD loop
  tmd::DevBuf batch_tab;    // BatchRep[nreplicas]: the replicas' buffers for the batched pair + step launch (cell-list contexts)
  std::vector<tmd::BatchRep> batch_host;  // what batch_tab holds
  int64_t batched_launches = 0;
  tmd::DevBuf chain_tab;    // ChainRepT[nreplicas] (list_build.hip): the replicas' arguments of the batched rebuild chain
  std::vector<unsigned char> chain_host;  // what chain_tab holds
  int64_t batched_chains = 0;
  std::vector<double> boxes_host;  // what `boxes` currently holds
  int max_excl = 0;
  int nactive = 0x7fffffff;  // atoms with original index >= nactive get empty lists (tmdhip_update_atoms)
  // Open-boundary contexts plan their cell grid over the bounding box of the positions, read back from the device —
  // unless the caller knows it (a brick of the domain decomposition: brick + halo, tmdhip_dd_migrate); consumed by
  // the next re-plan.  Atoms outside the bounds are binned into the edge cells (cell_coord clamps), which is safe.
  bool open_bounds_valid = false;
  double open_lo[3] = {0, 0, 0}, open_hi[3] = {0, 0, 0};
  int nexcl = 0;             // entries of the exclusion CSR
  std::vector<tmd::Replica> rep;
  // bonded part lives in bonded.hip
  void *bonded = nullptr;
  // timing of the dominant kernel
  bool timing = false;
  int timing_stride = 1;    // every n-th launch is timed
  int64_t timing_seen = 0;  // launches since timing was enabled
  int64_t timing_limit = 0; // stop after this many timed launches (0: no limit)
  int64_t timing_taken = 0;
  bool timing_interior_only = false;  // launches that also return energies (another kernel variant) are not timed
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t events_used = 0;
  double timing_ms = 0;
  int64_t timing_launches = 0;
};

namespace tmd {

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

// a FUSED launch of a lean pair kernel (see FusedStepT): device copy of the static part, this launch's part
template <typename R>
struct FusedLaunchT {
  const FusedStaticT<R> *fst;
  FusedStepT<R> step;
  bool langevin;
  bool eval_only;  // a plain evaluation with energies: step blocks that only add the bonded force (FUSED = 5)
};
using FusedLaunch = FusedLaunchT<float>;

constexpr size_t kRideMaxAtoms = 2048;  // bonded terms ride on the all-pairs launch up to this many atoms
constexpr int kForcesZeroed = 1 << 17;  // internal: the integrator kernel has already cleared `forces`
constexpr int kPrechecked = 1 << 16;  // internal compute flag: displacement test already enqueued
constexpr int kSkipChain = 1 << 18;   // internal compute flag: the host leaves the rebuild chain out for this step
constexpr int kDeferFold = 1 << 20;  // internal compute flag: a bonded evaluation with energies follows and folds the scratch rows
constexpr int kViolationCheck = 1 << 19;  // internal compute flag: ... and the step's displacement test (epilogue of the
                                          // previous pair launch) did not know that: the pair launch looks itself
constexpr int kListOnly = 1 << 21;   // internal compute flag: list bookkeeping only (displacement test / rebuild chain / chain skipped); the
                                     // caller makes the pair launch itself — the replica-batched launch of md_run — from ListOnlyOut
struct ListOnlyOut {
  int lmode;        // list duties of the launch's first thread (kLm*)
  int next_parity;  // parity of the step the launch's step blocks make
  int chain;        // kDeferChain: this step's rebuild chain is wanted (the host has not left it out) ...
  int chain_parity; // ... behind the flag word of this parity
};
constexpr int kSpecChain = 1 << 23;   // internal compute flag (tmdhip_compute): the chain of a plain evaluation may be left out on the strength of the
                                      // previous evaluation's report; the caller reads the flags back and repeats on F_VIOLATION
constexpr int kDeferChain = 1 << 22;  // internal compute flag (with kListOnly): do not enqueue the rebuild chain, report it (the caller
                                      // enqueues ONE chain for all replicas that want it: enqueue_chain_batch)
constexpr int kFallbackAllPairs = 77;  // compute_list: box too small for cells and algorithm = AUTO

// ---- functions the translation units call across each other ---------------------------------------
// context.hip
const void *set_boxes(tmdhip_ctx *ctx, const double *box_host, hipStream_t st);
int fold_energies(tmdhip_ctx *ctx, double *energies, hipStream_t st, int nrep);
int judge_flags(tmdhip_ctx *ctx, Replica &rp, const int *h, hipStream_t st);
template <typename R>
int alloc_replica(tmdhip_ctx *ctx, Replica &rp, int maxn);
template <typename R>
int compute_list(tmdhip_ctx *ctx, Replica &rp, const void *pos_v, const double *box, void *forces, double *energies,
                 int flags, hipStream_t st, const FusedLaunchT<R> *fused = nullptr, ListOnlyOut *list_only = nullptr);
// bonded.hip
void bonded_release(tmdhip_ctx *ctx);
int bonded_inline_args(tmdhip_ctx *ctx, const double *box, BondedArgs<float> &A);
int bonded_inline_args(tmdhip_ctx *ctx, const double *box, BondedArgs<double> &A);
// list_build.hip: displacement check -> conditional rebuild chain (every kernel returns at once unless the step's flag is set)
template <typename R>
int enqueue_list_update(tmdhip_ctx *ctx, Replica &rp, const R *pos, const PairConsts<R> &c, int force, hipStream_t st,
                        bool prechecked = false, bool speculate = false);
// the rebuild chains of several replicas in one launch per kernel (list_build.hip)
template <typename R>
int enqueue_chain_batch(tmdhip_ctx *ctx, int nsel, const int *reps, const R *const *pos, const int *parity, const double *const *box,
                        hipStream_t st);
// pair_generic.hip
template <typename R>
int launch_allpairs(tmdhip_ctx *ctx, const void *pos, const double *box, void *forces, double *energies, int flags,
                    unsigned long long *paircount, hipStream_t st, int nrep = 1, const BondedArgs<R> *bonded = nullptr);
template <typename R, bool ENERGY>
int launch_list_pair(tmdhip_ctx *ctx, Replica &rp, const PairConsts<R> &c, R *f, int overwrite, double *energies,
                     unsigned long long *paircount, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr,
                     int lmode = 0, const FusedLaunchT<R> *fl = nullptr, bool fold = true);
int halve_pair_count(unsigned long long *count_dev, hipStream_t st);
// pair_fast_f32.hip / pair_lean_f64.hip: the lean kernels behind launch_list_pair (LJ and/or electrostatics, <= 32 LJ classes)
template <bool ENERGY>
int launch_pair_fast_f32(tmdhip_ctx *ctx, Replica &rp, const PairConsts<float> &c, float *f, int overwrite, hipStream_t st,
                         hipEvent_t e0, hipEvent_t e1, int lmode, const FusedLaunch *fl);
template <bool ENERGY>
int launch_pair_lean_f64(tmdhip_ctx *ctx, Replica &rp, const PairConsts<double> &c, double *f, int overwrite,
                         hipStream_t st, hipEvent_t e0, hipEvent_t e1, int lmode);
// pair_fast_f32_batch.hip: the replicas rep0 .. rep0 + bl.nrep - 1 of a cell-list context in one FUSED launch (BatchRep / BatchLaunch)
void fused_grid_shape(const tmdhip_ctx *ctx, const Replica &rp, int bonded, int &pair_blocks, int &step_blocks);
int launch_pair_fast_f32_batch(tmdhip_ctx *ctx, int rep0, const PairConsts<float> &c, const BatchLaunch &bl, int lpa, bool energy,
                               bool langevin, hipStream_t st);
// md_loop.hip
template <typename R>
int md_run(tmdhip_ctx *ctx, const tmdhip_md_desc *d, hipStream_t st);
// device copy of a replica's FusedStatic, re-uploaded (one-thread kernel, stream-ordered) only when a field has changed
template <typename R>
int upload_fused_static(Replica &rp, const FusedStaticT<R> &now, hipStream_t st);
// can the pair launch of this replica integrate the next step itself?  (lean kernels, 4 .. 64 lanes per atom)
template <typename R>
bool fused_step_possible(const tmdhip_ctx *ctx, const Replica &rp, const PairConsts<R> &c);
// tmdhip_compute's two-launch evaluation (md_loop.hip) and the wait for a report's sequence word
int compute_fused_eval(tmdhip_ctx *ctx, const void *pos_dev, const double *box, void *forces_dev, double *e_dev, double *scratch_ke,
                       double *host_e, double *host_ke, int *host_flags, volatile unsigned *host_seq, hipStream_t st);
int wait_observed(volatile unsigned *hseq, unsigned seq, hipStream_t st);
// chain skipping: spin until the device has published sequence number `target` (Replica::hostpub[0]); false after 0.2 s
bool wait_published(volatile unsigned *hp, unsigned target);
// energies, kinetic energies and list flags of every replica through host-mapped memory + a sequence word the host
// spins on (<= 16 replicas); returns when the device has written them
int publish_observables(tmdhip_ctx *ctx, const double *energies_dev, const double *ke_dev, bool lists, double *host_e,
                        double *host_ke, int *host_flags, volatile unsigned *host_seq, hipStream_t st);

}  // namespace tmd
