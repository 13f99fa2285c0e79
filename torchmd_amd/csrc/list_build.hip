// K1 + K2 of the nonbonded engine for gfx950 (MI355X): cell binning and the Verlet-list build.
//
// The reference has no counterpart that runs (torchmd/neighbourlist.py:4-47 is dead code; its 27-neighbour periodic
// convention is what the stencil below generalises): the pair set of torchmd/forces.py:348-357 (all i<j minus
// exclusions) filtered by forces.py:76-81 is produced here by an O(N) cell list + a Verlet list with a skin.  Every
// kernel of the chain starts with `if (!*flag) return`: the rebuild decision is taken on the device by the
// displacement test (engine.h: ListCheck) and needs no host round trip.
#include "engine.h"

namespace tmd {

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

// single block of 1024 threads; cell_start[ncell] = n afterwards; counts are zeroed for the next rebuild.
// Up to 32 cells per thread (grids of <= 32 768 cells; C3: 6 859 cells, 7 per thread) a thread owns a run of consecutive
// cells: its counts in registers, one scan of the 1 024 run sums (wave shuffles + 16 LDS words, two barriers) — the
// chunk-by-chunk loop below makes four barriers per 1 024 cells (10.8 us per rebuild at C3, three times the rest of its work).
constexpr int kScanRun = 32;
__global__ __launch_bounds__(1024) void scan_cells_kernel(int ncell, int *__restrict__ count,
                                                          int *__restrict__ cell_start, const int *flag) {
  if (*flag == 0) return;
  __shared__ int wsum[16];
  __shared__ int carry;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int per = (ncell + 1023) / 1024;
  if (per <= kScanRun) {
    int v[kScanRun], mine = 0;
    const int c0 = t * per;
#pragma unroll
    for (int k = 0; k < kScanRun; ++k) {
      v[k] = (k < per && c0 + k < ncell) ? count[c0 + k] : 0;
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
    for (int k = 0; k < kScanRun; ++k) {
      if (k < per && c0 + k < ncell) {
        cell_start[c0 + k] = run;
        count[c0 + k] = 0;
      }
      run += v[k];
    }
    if (t == 1023) cell_start[ncell] = run;
    return;
  }
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

// atom at position `a` of the unsorted cell order -> its final slot (rank by original index inside the cell) and every
// per-slot copy the pair kernels read
template <typename R>
__device__ __forceinline__ typename Vec<R>::T4 place_atom_at(const PlaceArgs<R> &P, int me, int dst);

// (the caller notes the coordinate extent for its whole wave: extent_note_wave)
template <typename R>
__device__ __forceinline__ typename Vec<R>::T4 place_atom(const PlaceArgs<R> &P, int a) {
  const int me = P.order_tmp[a];
  const int cidx = P.cell_of[me];
  const int s = P.cell_start[cidx], e = P.cell_start[cidx + 1];
  int rank = 0;
  for (int k = s; k < e; ++k) rank += P.order_tmp[k] < me;
  return place_atom_at<R>(P, me, s + rank);
}

// atom `me` at its final slot `dst`: every per-slot copy the pair kernels read
template <typename R>
__device__ __forceinline__ typename Vec<R>::T4 place_atom_at(const PlaceArgs<R> &P, int me, int dst) {
  P.order[dst] = me;
  P.inv[me] = dst;
  typename Vec<R>::T4 v;
  v.x = P.pos[3 * me + 0];
  v.y = P.pos[3 * me + 1];
  v.z = P.pos[3 * me + 2];
  v.w = P.qs[me];
  P.sorted[dst] = v;
  const int ty = P.types[me];
  P.stype[dst] = ty;
  P.binfo[dst] = me | (P.type_in_entry ? ty << kEntryTypeShift : 0);
  R hrec = R(0);
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
    hrec = h;
  }
  float4 b;  // (fp64 contexts: folded in fp64, then rounded — kBuildMarginF64)
  b.x = (float)wrap_into_box(v.x, P.box[0], P.invbox[0]);
  b.y = (float)wrap_into_box(v.y, P.box[1], P.invbox[1]);
  b.z = (float)wrap_into_box(v.z, P.box[2], P.invbox[2]);
  b.w = (float)hrec;
  P.bsorted[dst] = b;
  P.ref[3 * me + 0] = v.x;
  P.ref[3 * me + 1] = v.y;
  P.ref[3 * me + 2] = v.z;
  return v;
}

// dummy record `which` behind the last atom of the cell-sorted copies (padded rows, engine.h: pad_entry_for)
template <typename R>
__device__ __forceinline__ void place_dummy(const PlaceArgs<R> &P, int which) {
  if (!P.dummy_a) return;
  typename Vec<R>::T4 v;
  v.x = P.dummy_pos[which][0];
  v.y = P.dummy_pos[which][1];
  v.z = P.dummy_pos[which][2];
  v.w = R(0);
  P.dummy_a[which] = v;
  if (P.dummy_b) P.dummy_b[which] = v;
}

// ---- two-launch binning (round 4) ---------------------------------------------------------------------------------
// bin_count / scan_cells / fill_cells / place_sorted are four launches — on the steps where the chain is enqueued but
// nothing is rebuilt (~2 of 11) four early-exit launches, and mid-size boxes are host-bound on exactly those steps.
// Two launches do the same work for grids of <= kScanPlaceMaxCells cells:
//   bin_members_kernel   cell of every atom + the cell's member array in arrival order (fixed capacity kCellCap; an
//                        overflow raises F_CELLCAP and the replica falls back to the four launches)
//   scan_place_kernel    every block scans ALL cell counts into LDS (27 KB of reads per block at C3: nothing), block 0
//                        stores cell_start, then thread = atom: final slot = start of its cell + its rank by original
//                        index among the cell's members (deterministic order), and every per-slot copy (place_atom_at)
// The counts are cleared for the next build by the build kernel (one cell per block), not here: other blocks of
// scan_place_kernel may still be reading them.
template <typename R>
__device__ __forceinline__ void bin_members_body(int n, const R *__restrict__ pos, const Grid &g, int *__restrict__ cell_of,
                                                 int *__restrict__ count, int *__restrict__ members, int *flags, const int *flag) {
  if (*flag == 0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int cx = cell_coord(pos[3 * i + 0], g, 0);
  const int cy = cell_coord(pos[3 * i + 1], g, 1);
  const int cz = cell_coord(pos[3 * i + 2], g, 2);
  const int cidx = (cx * g.nc[1] + cy) * g.nc[2] + cz;
  cell_of[i] = cidx;
  const int k = atomicAdd(&count[cidx], 1);
  if (k < kCellCap) members[(size_t)cidx * kCellCap + k] = i;
  else flags[F_CELLCAP] = 1;
}

template <typename R>
__global__ void bin_members_kernel(int n, const R *__restrict__ pos, Grid g, int *__restrict__ cell_of, int *__restrict__ count,
                                   int *__restrict__ members, int *flags, const int *flag) {
  bin_members_body<R>(n, pos, g, cell_of, count, members, flags, flag);
}

template <typename R>
__device__ __forceinline__ void scan_place_body(int n, int ncell, const int *__restrict__ count, const int *__restrict__ members,
                                                int *__restrict__ cell_start_out, const PlaceArgs<R> &P, const int *flag) {
  if (*flag == 0) return;
  extern __shared__ int s_start[];  // [ncell + 1]
  __shared__ int wsum[4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int per = (ncell + 255) / 256, c0 = t * per, c1 = min(c0 + per, ncell);
  int mine = 0;
  for (int k = c0; k < c1; ++k) mine += min(count[k], kCellCap);
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
  for (int k = c0; k < c1; ++k) {
    s_start[k] = run;
    run += min(count[k], kCellCap);
  }
  if (t == 255) s_start[ncell] = run;
  __syncthreads();
  if (blockIdx.x == 0)
    for (int k = t; k <= ncell; k += 256) cell_start_out[k] = s_start[k];
  if (blockIdx.x == 0 && t < 2) place_dummy<R>(P, t);
  const int me = blockIdx.x * blockDim.x + t;
  typename Vec<R>::T4 v{};
  bool placed = false;
  if (me < n) {
    const int cidx = P.cell_of[me];
    const int cnt = min(count[cidx], kCellCap);
    const int *m = members + (size_t)cidx * kCellCap;
    int rank = 0, seen = 0;
    for (int k = 0; k < cnt; ++k) {
      const int o = m[k];
      rank += o < me;
      seen |= o == me;
    }
    // (an atom that did not fit its cell's member array is not placed: F_CELLCAP is set, the build is thrown away)
    if (seen) {
      v = place_atom_at<R>(P, me, s_start[cidx] + rank);
      placed = true;
    }
  }
  extent_note_wave<R>(P.ext, placed, v.x, v.y, v.z);
}

template <typename R>
__global__ __launch_bounds__(256) void scan_place_kernel(int n, int ncell, const int *__restrict__ count,
                                                         const int *__restrict__ members, int *__restrict__ cell_start_out,
                                                         PlaceArgs<R> P, const int *flag) {
  scan_place_body<R>(n, ncell, count, members, cell_start_out, P, flag);
}

template <typename R>
__global__ void place_sorted_kernel(int n, PlaceArgs<R> P, const int *flag) {
  if (*flag == 0) return;
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a < 2) place_dummy<R>(P, a);
  typename Vec<R>::T4 v{};
  if (a < n) v = place_atom<R>(P, a);
  extent_note_wave<R>(P.ext, a < n, v.x, v.y, v.z);
}

// Binning of a small system in ONE launch of one block: count (LDS atomics), scan, fill and place — the work of
// bin_count / scan_cells / fill_cells / place_sorted.  On the ~10 of 11 steps without a rebuild the chain then costs
// two early-exit launches instead of five (~1.7 us each).  A rebuild on one CU is slower than the four parallel
// launches, which sets the size limit — measured, water boxes, us per MD step without / with: 5 184 atoms 27.2 / 23.8,
// 12 288 atoms 34.6 / 36.1, 41 472 atoms 41.8 / 67.1.
constexpr int kPrepSmallMaxCells = 4096;
constexpr int kPrepSmallMaxAtoms = 8192;
template <typename R>
__device__ __forceinline__ void prep_small_body(int n, const R *__restrict__ pos, const Grid &g, int ncell, int *cell_of,
                                                int *__restrict__ slot, int *cell_start, int *order_tmp, const PlaceArgs<R> &P,
                                                const int *flag) {
  // (cell_of, cell_start and order_tmp are read back through P by place_atom below: no __restrict__ on them)
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
  for (int a0 = 0; a0 < n; a0 += 1024) {  // (all lanes stay in the loop: the extent is noted wave by wave)
    const int a = a0 + t;
    typename Vec<R>::T4 v{};
    if (a < n) v = place_atom<R>(P, a);
    extent_note_wave<R>(P.ext, a < n, v.x, v.y, v.z);
  }
  if (t < 2) place_dummy<R>(P, t);
}
template <typename R>
__global__ __launch_bounds__(1024) void prep_small_kernel(int n, const R *__restrict__ pos, Grid g, int ncell,
                                                          int *cell_of, int *__restrict__ slot, int *cell_start,
                                                          int *order_tmp, PlaceArgs<R> P, const int *flag) {
  prep_small_body<R>(n, pos, g, ncell, cell_of, slot, cell_start, order_tmp, P, flag);
}

// ---- K2: Verlet list build ---------------------------------------------------------------------
// One wave per cell.  The candidates (all atoms of the (2m+1)^3 stencil cells, which are contiguous
// runs of the cell-sorted arrays) are streamed through the 64 lanes with coalesced loads; for every
// chunk of 64 candidates the wave loops over the atoms i of its cell (wave-uniform data), tests
// |d|^2 <= rlist^2 and the exclusions, and appends the hits of atom i with a ballot / prefix-popcount
// compaction.  Entry order per atom is fixed by the stencil order -> lists are bit-reproducible.
// WSKIN: per-atom skins — pair (i, j) is listed when |d| <= cutoff + s_i + s_j (s = the atom's half skin: the
// displacement it may reach before a rebuild, see ListCheck), instead of cutoff + skin for every pair.
// LPAS: log2 of the lanes per atom of the list layout as a compile-time constant (3 = the C3 / water layout: the masks and
// shifts of a hit's byte offset become literals — full-rate VALU, no registers), or -1: read from ListGeom.
template <typename R, bool LOOP, bool WSKIN, int LPAS>
__device__ __forceinline__ void build_list_body(
    int n, const typename Vec<R>::T4 *__restrict__ bsorted, const int *__restrict__ binfo,
    const int *__restrict__ cell_start, const Grid &g, const PairConsts<R> &c, R rlist2, R rcut,
    const int *__restrict__ excl_off, const int *__restrict__ excl_idx, const ListGeom &lg,
    unsigned *__restrict__ nlist, int *__restrict__ nneigh, int *__restrict__ status, const int *flag,
    int ncell, int nactive, int type_in_entry, unsigned long long *dbg, int split, int *__restrict__ count_zero) {
  if (*flag == 0) return;
  const unsigned long long dbg_t0 = dbg ? wall_clock64() : 0ull;  // (the device-wide 100 MHz clock: comparable across XCDs)  // TMDHIP_DEBUG_TIMELINE (tools/build_timeline.py)
  using R4 = typename Vec<R>::T4;
  // the whole list as a bounds-checked buffer (< 2^30 entries): an out-of-range store is dropped
  const __amdgpu_buffer_rsrc_t nrsrc = __builtin_amdgcn_make_buffer_rsrc(
      nlist, 0, (int)((((size_t)n + lg.apw - 1) / lg.apw) * (size_t)lg.maxn * lg.apw * 4u), 0x00020000);
  __shared__ int seg_start[128];
  __shared__ int seg_prefix[129];
  // (the periodic image of a segment's cells — 2 bits per axis, 0:-L 1:0 2:+L — rides in bits 24..29 of seg_start: start < 2^23)
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
  if (count_zero && part == 0 && lane == 0) count_zero[cell] = 0;  // (two-launch binning: the cell counts of the next build)
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
  seg_start[lane] = st2[0] | (code2[0] << 24);
  seg_start[lane + 64] = st2[1] | (code2[1] << 24);
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
  __shared__ __align__(16) unsigned s_rowoff[64];  // byte offset of the atom's list row (= s_rec1[].w; four of them are one 16-byte read)
  const unsigned lpas = LPAS >= 0 ? (unsigned)LPAS : (unsigned)lg.lpa_shift;
  const int apw_shift = 6 - (int)lpas;
  const unsigned kmask = (1u << lpas) - 1u;
  // entry k of a row sits at byte ((k / (4 LPA)) << 10) + ((k % LPA) << 4) + (((k / LPA) % 4) << 2)  (list_slot); the
  // masks live in VGPRs (an SGPR operand halves the VALU rate)
  unsigned vmask_hi = ~((4u << lpas) - 1u), vmask_lo = kmask;
  if constexpr (LPAS < 0) {  // run-time layout: the masks live in VGPRs (an SGPR operand halves the VALU rate)
    asm("v_mov_b32 %0, %1" : "=v"(vmask_hi) : "s"(~((4u << lpas) - 1u)));
    asm("v_mov_b32 %0, %1" : "=v"(vmask_lo) : "s"(kmask));
  }
  unsigned vmaxn1 = (unsigned)lg.maxn - 1u;  // (LPAS >= 0: an SGPR operand of one v_min per hit row — the register it would
                                             // take as a VGPR is the one that decides between six and seven waves per SIMD)
  if constexpr (LPAS < 0) asm("v_mov_b32 %0, %1" : "=v"(vmaxn1) : "s"((unsigned)lg.maxn - 1u));
  const unsigned sh_hi = 8u - lpas;  // (k / (4 LPA)) << 10 == (k & ~(4 LPA - 1)) << (10 - 2 - lpa_shift)
  for (int ib = cs; ib < ce; ib += 64) {  // blocks of up to 64 atoms i of this cell (usually one)
    const int iend = min(ib + 64, ce);
    const int ni = iend - ib;
    __syncthreads();
    int long_rows = 0;
    if (lane < ni) {
      const int a = ib + lane;
      R4 p = bsorted[a];  // (already folded into the box; .w = the atom's half skin)
      const unsigned rowoff = (((unsigned)(a >> apw_shift) * (unsigned)lg.maxn) << apw_shift) +
                              ((unsigned)(a & ((1 << apw_shift) - 1)) << (lpas + 2));
      p.w = WSKIN ? rcut + p.w : R(0);
      const int oi = binfo[a] & kInfoIndexMask;
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
      s_rowoff[lane] = rowoff * 4u;
      long_rows = ne > EXS - 1;
    } else {
      // dummy atoms that pad the last batch of four: parked out of reach (never a hit, never a store)
      R4 p;
      p.x = (R)-1e18;
      p.y = p.z = p.w = R(0);
      s_rec0[lane] = p;
      s_rec1[lane] = make_int4(-1, -1, -1, 0);
      s_rowoff[lane] = 0u;
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
    R4 nx_p;  // {wrapped position, half skin}
    int nx_j = cs, nx_code = 0, nx_info = 0;
    bool nx_valid = false;
    // (Round 5 also tried the prefetch straight into LDS — `buffer_load ... lds`, no prefetch registers, no copies at the end
    // of an iteration: 163 us per build against 149 with the registers; profiles/r05_build_experiments.txt.)
    auto fetch = [&](int q0) {
      const int q = q0 + lane;
      nx_valid = q < ncand;
      nx_j = cs;
      if (nx_valid) while (seg_prefix[seg + 1] <= q) ++seg;  // last s with seg_prefix[s] <= q
      const int packed = seg_start[seg];
      nx_code = packed >> 24;
      if (nx_valid) nx_j = (packed & 0x00FFFFFF) + (q - seg_prefix[seg]);
      nx_p = bsorted[nx_j];
      nx_info = binfo[nx_j];
    };
    // ---- candidate prefilter (round 5) ------------------------------------------------------------------------------
    // Only a quarter of the (atom, candidate) tests of a (2m+1)^3 stencil hit, and the build is instruction bound (debug
    // builds, profiles/r05_build_experiments.txt): every 64-candidate chunk costs each batch of four atoms ~130 instructions
    // whether anything is in range or not.  A candidate that lies further from the BOUNDING BOX of this block's atoms than
    // the block's largest reach (+ its own half skin) can hit none of them: 44 % of the stream at C3.  The survivors are
    // compacted ACROSS chunks, in stream order — so every list is the same list, entry for entry — and the batch loop runs
    // on dense chunks: 16 instead of 28 per cell.  The compaction needs no LDS memory: ds_permute_b32 (a forward lane
    // permutation through the LDS crossbar) pushes the kept lanes of a chunk behind the `pending` survivors of the
    // previous ones; what wraps around the 64 lanes starts the next dense chunk.
    R blo[3], bhi[3], breach;
    {
      const R4 p = s_rec0[lane];
      const bool real = lane < ni && p.x > (R)-1e17;  // (passive atoms — halo rows of a brick — are parked out of reach: they list nothing)
      const R big = (R)1e18;
      R v[7] = {real ? p.x : big,  real ? p.y : big,  real ? p.z : big,
                real ? p.x : -big, real ? p.y : -big, real ? p.z : -big, real ? p.w : R(0)};
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = min(v[k], __shfl_xor(v[k], o, 64));
#pragma unroll
        for (int k = 3; k < 7; ++k) v[k] = max(v[k], __shfl_xor(v[k], o, 64));
      }
      // (wave-uniform after the butterfly: into scalar registers — the build runs at seven waves per SIMD on 72 VGPRs)
      auto uniform = [](R x) {
        if constexpr (sizeof(R) == 4) {
          return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
        } else {
          const long long b = __double_as_longlong(x);
          const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
          const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)b >> 32));
          return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        }
      };
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        blo[k] = uniform(v[k]);
        bhi[k] = uniform(v[3 + k]);
      }
      breach = uniform(v[6]);
    }
    // one value of any 4- or 8-byte type to lane `dest4 / 4` (every lane sends, the destinations are a permutation)
    auto push = [&](auto val, int dest4) {
      using T = decltype(val);
      if constexpr (sizeof(T) == 4) {
        int w;
        __builtin_memcpy(&w, &val, 4);
        w = __builtin_amdgcn_ds_permute(dest4, w);
        T out;
        __builtin_memcpy(&out, &w, 4);
        return out;
      } else {
        int w[2];
        __builtin_memcpy(w, &val, 8);
        w[0] = __builtin_amdgcn_ds_permute(dest4, w[0]);
        w[1] = __builtin_amdgcn_ds_permute(dest4, w[1]);
        T out;
        __builtin_memcpy(&out, w, 8);
        return out;
      }
    };
    // survivors waiting for a full chunk: lanes [0, pending)
    R a_x = R(0), a_y = R(0), a_z = R(0);
    R a_s = R(0);
    unsigned a_entry = 0u;
    int pending = 0;  // wave-uniform
    fetch(0);
    for (int q0 = 0; q0 < ncand || pending > 0; q0 += 64) {
      const bool have = q0 < ncand;  // (one more round behind the last chunk flushes what is pending)
      R w_x = R(0), w_y = R(0), w_z = R(0), w_s = R(0);  // this round's push: what wraps around the 64 lanes starts the next chunk
      unsigned w_entry = 0u;
      int total = pending;
      if (have) {
        const R4 rec = nx_p;
        const int info = nx_info;
        const R isj = rec.w;
        R ipx = rec.x, ipy = rec.y, ipz = rec.z;
        const int j = nx_j, code = nx_code;
        const bool valid = nx_valid;
        const unsigned ioj = (unsigned)info & (unsigned)kInfoIndexMask;
        // (bit 0 of an entry is free — the pair kernels mask it: here it carries "somebody in this block excludes this
        // candidate" (s_bm) through the compaction, and is cleared before the entry is stored)
        const unsigned iflag = any_long ? 1u : (s_bm[(ioj & 2047u) >> 5] >> (ioj & 31u)) & 1u;
        const unsigned ientry = ((unsigned)j << 4) | ((unsigned)info & ~(unsigned)kInfoIndexMask) | iflag;
        if (q0 + 64 < ncand) fetch(q0 + 64);
        // candidate position as the periodic image that lies next to this cell: the i loop then needs
        // no minimum-image arithmetic (the list criterion has the skin as slack, so it need not reproduce
        // the reference's rounding; the pair kernel's cutoff test does).
        ipx += (R)((code & 3) - 1) * c.box[0];
        ipy += (R)(((code >> 2) & 3) - 1) * c.box[1];
        ipz += (R)(((code >> 4) & 3) - 1) * c.box[2];
        // distance to the block's bounding box against the block's largest reach (a hair of slack for the rounding of the
        // two different expressions: a candidate this filter drops must fail every atom's own test)
        const R ex = max(max(blo[0] - ipx, ipx - bhi[0]), R(0)), ey = max(max(blo[1] - ipy, ipy - bhi[1]), R(0)),
                ez = max(max(blo[2] - ipz, ipz - bhi[2]), R(0));
        R lim;
        if constexpr (WSKIN) {
          const R rr = breach + isj;
          lim = rr * rr * (R)1.00001;
        } else {
          lim = rlist2 * (R)1.00001;
        }
        const unsigned long long keep = wave_mask_le(ex * ex + ey * ey + ez * ez, valid ? lim : (R)-1);
        const int cnew = (int)__popcll(keep);
        // forward permutation: the kept lanes to [pending, pending + cnew), the others behind them (mod 64)
        const unsigned pre = __builtin_amdgcn_mbcnt_hi((unsigned)(keep >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)keep, 0u));
        const bool kept = __builtin_amdgcn_inverse_ballot_w64(keep);
        const int dest4 = (int)((((unsigned)pending + (kept ? pre : (unsigned)cnew + ((unsigned)lane - pre))) & 63u) << 2);
        w_x = push(ipx, dest4);
        w_y = push(ipy, dest4);
        w_z = push(ipz, dest4);
        if constexpr (WSKIN) w_s = push(isj, dest4);
        w_entry = push(ientry, dest4);
        // the new survivors join the buffer behind the pending ones
        const bool from_buffer = lane < pending;
        a_x = from_buffer ? a_x : w_x;
        a_y = from_buffer ? a_y : w_y;
        a_z = from_buffer ? a_z : w_z;
        if constexpr (WSKIN) a_s = from_buffer ? a_s : w_s;
        a_entry = from_buffer ? a_entry : w_entry;
        total = __builtin_amdgcn_readfirstlane(pending + cnew);  // (wave-uniform: keeps the loop control on the scalar unit)
        if (total < 64) {  // not a full chunk yet
          pending = total;
          continue;
        }
      }
      // a dense chunk in the buffer's lanes [0, min(total, 64)) (the last one of a block may be partial: its lanes past
      // the end are parked far away so that they can never hit)
      R4 pj;
      pj.x = a_x;
      pj.y = a_y;
      pj.z = a_z;
      pj.w = R(0);
      const R sj = a_s;
      const bool parked = lane >= min(total, 64);
      if (parked) pj.x = (R)1e18;
      const unsigned entry = a_entry & ~1u;
      // the original index of a flagged candidate (exclusion compares): gathered only by the chunks that hold one
      const unsigned long long special = __builtin_amdgcn_uicmp(parked ? 0u : (a_entry & 1u), 0u, 33 /* ne */);  // (a wave-wide mask)
      unsigned oj = 0xFFFFFFFFu;
      if (special) oj = (unsigned)binfo[(a_entry >> 4) & 0x7FFFFFu] & (unsigned)kInfoIndexMask;
      // what wrapped around the 64 lanes is the start of the next chunk (w_* stay live through the batch loop)
      struct Carry {
        R x, y, z, s;
        unsigned entry;
        int pending;
      } carry{w_x, w_y, w_z, w_s, w_entry, total >= 64 ? total - 64 : 0};
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
          const unsigned kk = k >> lpas;
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
        for (; t < ni; t += 4, recoff += 64u) {
          const R4 p0 = rec0(recoff), p1 = rec0(recoff + 16u), p2 = rec0(recoff + 32u), p3 = rec0(recoff + 48u);
          unsigned long long m[4] = {in_range(p0), in_range(p1), in_range(p2), in_range(p3)};
          const unsigned long long any = m[0] | m[1] | m[2] | m[3];
          if (!any) continue;
          const int4 base4 = *reinterpret_cast<const int4 *>(&s_cnt[t]);
          const int base[4] = {base4.x, base4.y, base4.z, base4.w};
          const uint4 ro4 = *reinterpret_cast<const uint4 *>(&s_rowoff[t]);
          const unsigned ro[4] = {ro4.x, ro4.y, ro4.z, ro4.w};
          if (any & special) {  // rare: a flagged candidate is in range of one of the four
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int4 ex = *reinterpret_cast<const int4 *>(reinterpret_cast<const char *>(s_rec1) + recoff + 16u * u);
              m[u] &= ~(__builtin_amdgcn_uicmp((unsigned)ex.x, oj, 32 /* eq */) | __builtin_amdgcn_uicmp((unsigned)ex.y, oj, 32) |
                        __builtin_amdgcn_uicmp((unsigned)ex.z, oj, 32));
            }
          }
          int cnt[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            // slot of this lane's hit, clamped to the row's last one: a row that overflows is reported through F_MAXN
            // and its list thrown away (the caller grows the capacity and rebuilds), so what lands there is never used
            // (the counter is the start value of the prefix count: one add less)
            const unsigned k = min(__builtin_amdgcn_mbcnt_hi((unsigned)(m[u] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[u], (unsigned)base[u])),
                                   vmaxn1);
            // byte offset of entry k in the row: iteration kk = k / LPA, lane part k % LPA (list_slot's layout)
            unsigned posb = ro[u] + ((k & vmask_hi) << sh_hi);
            posb += (k & vmask_lo) << 4;
            posb += __builtin_amdgcn_ubfe(k, lpas, 2u) << 2;
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
      a_x = carry.x;
      a_y = carry.y;
      a_z = carry.z;
      a_s = carry.s;
      a_entry = carry.entry;
      pending = __builtin_amdgcn_readfirstlane(carry.pending);
    }
    __syncthreads();
    const int mycnt = s_cnt[lane];
    if (lane < ni) nneigh[ib + lane] = min(mycnt, lg.maxn);
    wmax = max(wmax, lane < ni ? mycnt : 0);
  }
  } while (LOOP && (cell += gridDim.x) < ncell);
  if (dbg && lane == 0) {  // per block: entry / exit time (10 ns ticks of the device-wide clock), XCC id, candidates x atoms of its (last) cell
    unsigned long long *o = dbg + 4 * (size_t)blockIdx.x;
    o[0] = dbg_t0;
    o[1] = wall_clock64();
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
template <typename R, bool LOOP, bool WSKIN, int LPAS>
// (fp32: held to seven waves per SIMD — 72 VGPRs; the allocator is one register over without the hint and spills 16 bytes
// in the prologue with it — because all 6 859 cell blocks of C3 are then resident at once: 7 168 slots)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(sizeof(R) == 4 ? 7 : 1, 8))) void build_list_kernel(
    int n, const typename Vec<R>::T4 *__restrict__ bsorted, const int *__restrict__ binfo,
    const int *__restrict__ cell_start, Grid g, PairConsts<R> c, R rlist2, R rcut,
    const int *__restrict__ excl_off, const int *__restrict__ excl_idx, ListGeom lg,
    unsigned *__restrict__ nlist, int *__restrict__ nneigh, int *__restrict__ status, const int *flag,
    int ncell, int nactive, int type_in_entry, unsigned long long *dbg, int split, int *__restrict__ count_zero) {
  build_list_body<R, LOOP, WSKIN, LPAS>(n, bsorted, binfo, cell_start, g, c, rlist2, rcut, excl_off, excl_idx, lg, nlist, nneigh, status, flag,
                                        ncell, nactive, type_in_entry, dbg, split, count_zero);
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


// ---- the rebuild chains of several replicas in one launch per kernel (round 6) ---------------------------------------------------
// The replicas of a cell-list context rebuild on their own steps, each behind its own flag; enqueued one after the other, R
// chains are 2-3 R launches on every step on which somebody is near a limit, and a small box's rebuild runs on a fraction of
// the chip while the others wait.  Here blockIdx.y picks the replica: its arguments come from a device table (ChainRepT: what
// enqueue_chain would have passed by value, uploaded when an entry changes), what alternates from step to step — positions,
// flag word, target copy — as a kernel argument (ChainSelT).  Same bodies, same per-replica results.
template <typename R>
struct ChainRepT {
  Grid g;
  PlaceArgs<R> P;
  PairConsts<float> c;  // (of the build: fp32 in either precision)
  ListGeom lg;
  int *cell_of, *count, *members, *flags, *slot, *cell_start, *order_tmp;
  const float4 *bsorted;
  const int *binfo;
  unsigned *nlist;
  int *nneigh;
  int *count_zero;
  float rlist2, rcut;
  int ncell, nactive, type_in_entry, split, build_blocks;
  int mode;  // 0: prep_small + build, 1: bin_members + scan_place + build
  int wskin, lpas3;
};
template <typename R>
struct ChainSelT {
  int nsel;
  int together;  // TMDHIP_REPLICA_REBUILDS=together: every replica of the launch rebuilds when ANY of them has asked (chain_any)
  int rep[kBatchMax];
  const R *pos[kBatchMax];
  const int *flag[kBatchMax];
  typename Vec<R>::T4 *sorted[kBatchMax];
};
// Opt-in (TMDHIP_REPLICA_REBUILDS=together): with R replicas somebody rebuilds on most steps, and every build is a ~40-us chain the
// pair launch waits for.  Builds of several replicas in ONE launch cost what one costs, so letting every replica whose chain is in
// the launch rebuild whenever any of them has to turns ~R/11 builds per step into ~1/9.  A list built early is as complete as one
// built on time, but its entries come in another order: a replica's forces are then no longer bit-identical to its run alone
// (same tolerance against the reference) — hence not the default.  The flag words are stable while a chain runs.
template <typename R>
__device__ __forceinline__ int chain_any(const ChainSelT<R> &sel, int y) {
  if (!sel.together) return *sel.flag[y];
  int any = 0;
  for (int k = 0; k < sel.nsel; ++k) any |= *sel.flag[k];
  return any;
}
// the entry's PlaceArgs with this launch's positions and target copy (dummy_a / dummy_b are "both copies": order does not matter)
template <typename R>
__device__ __forceinline__ PlaceArgs<R> chain_place_args(const ChainRepT<R> &A, const ChainSelT<R> &sel, int y) {
  PlaceArgs<R> P = A.P;
  P.pos = sel.pos[y];
  P.sorted = sel.sorted[y];
  return P;
}
template <typename R>
__global__ void bin_members_batch_kernel(int n, const ChainRepT<R> *__restrict__ tab, ChainSelT<R> sel) {
  const int y = blockIdx.y;
  const ChainRepT<R> &A = tab[sel.rep[y]];
  const int go = chain_any(sel, y);
  bin_members_body<R>(n, sel.pos[y], A.g, A.cell_of, A.count, A.members, A.flags, &go);
}
template <typename R>
__global__ __launch_bounds__(256) void scan_place_batch_kernel(int n, const ChainRepT<R> *__restrict__ tab, ChainSelT<R> sel) {
  const int y = blockIdx.y;
  const ChainRepT<R> &A = tab[sel.rep[y]];
  const int go = chain_any(sel, y);
  if (go == 0) return;
  const PlaceArgs<R> P = chain_place_args(A, sel, y);
  scan_place_body<R>(n, A.ncell, A.count, A.members, A.cell_start, P, &go);
}
template <typename R>
__global__ __launch_bounds__(1024) void prep_small_batch_kernel(int n, const ChainRepT<R> *__restrict__ tab, ChainSelT<R> sel) {
  const int y = blockIdx.y;
  const ChainRepT<R> &A = tab[sel.rep[y]];
  const int go = chain_any(sel, y);
  if (go == 0) return;
  const PlaceArgs<R> P = chain_place_args(A, sel, y);
  prep_small_body<R>(n, sel.pos[y], A.g, A.ncell, A.cell_of, A.slot, A.cell_start, A.order_tmp, P, &go);
}
template <typename R, bool WSKIN, int LPAS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(7, 8))) void build_list_batch_kernel(
    int n, const int *__restrict__ excl_off, const int *__restrict__ excl_idx, const ChainRepT<R> *__restrict__ tab, ChainSelT<R> sel) {
  const int y = blockIdx.y;
  const ChainRepT<R> &A = tab[sel.rep[y]];
  if ((int)blockIdx.x >= A.build_blocks) return;
  const int go = chain_any(sel, y);
  build_list_body<float, false, WSKIN, LPAS>(n, A.bsorted, A.binfo, A.cell_start, A.g, A.c, A.rlist2, A.rcut, excl_off, excl_idx, A.lg, A.nlist,
                                         A.nneigh, A.flags + F_MAXN, &go, A.ncell, A.nactive, A.type_in_entry, nullptr, A.split,
                                         A.count_zero);
}

// the buffers a rebuild chain reads and writes (the replica's own)
struct ListTarget {
  DevBuf *cell_of, *slot, *order_tmp, *count, *cell_start, *order, *inv, *stype, *ref, *sorted_hs, *hs2_dyn, *nlist, *nneigh, *sorted, *members;
};

// The rebuild chain: cell binning (one launch for small systems, two or four otherwise) and the list build, all on `st`.
// Every kernel returns at once unless *flag != 0.
// INVARIANT: build_list_kernel is never launched without the placement kernels of the SAME chain in front of it — it reads
// the atoms through bsorted / binfo, which only they write (wrapped positions and half skins of THIS build); a path that
// rebuilt without re-placing (a capacity regrow that reused the cell order, say) would list from stale records.
// The field widths the build relies on: slot and original indices < 2^23 (tmdhip_create refuses more atoms), a segment's
// start in the low 24 bits of seg_start with the image code above it, the original index in kInfoIndexMask of binfo with
// the LJ class above it.
static_assert(kInfoIndexMask == (1 << kEntryTypeShift) - 1, "binfo: original index below the LJ class field");
static_assert((kEntryOffMask >> 4) == (1u << 23) - 1u, "list entries carry 23-bit slots: seg_start packs a start < 2^23 below its image code (bit 24 up)");
// `plan`: fill the replica's entry of the batched chain's table instead of launching (enqueue_chain_batch)
template <typename R>
static int enqueue_chain(tmdhip_ctx *ctx, Replica &rp, const R *pos, const PairConsts<R> &c, const int *flag, hipStream_t st,
                         ChainRepT<R> *plan = nullptr) {
  const ListTarget T = {&rp.cell_of, &rp.slot, &rp.order_tmp, &rp.count, &rp.cell_start, &rp.order, &rp.inv, &rp.stype, &rp.ref,
                        &rp.sorted_hs, &rp.hs2_dyn, &rp.nlist, &rp.nneigh, &rp.sorted, &rp.members};
  const hipStream_t st_build = st;
  using R4 = typename Vec<R>::T4;
  const int n = ctx->d.natoms;
  int *flags = rp.flags.as<int>();
  const int nb = (n + 255) / 256;
  PlaceArgs<R> P;
  std::memset(&P, 0, sizeof(P));  // (padding bytes too: the batched chain compares table entries with memcmp)
  P.cell_of = T.cell_of->as<int>();
  P.cell_start = T.cell_start->as<int>();
  P.order_tmp = T.order_tmp->as<int>();
  P.pos = pos;
  P.qs = ctx->qs.as<R>();
  P.types = ctx->types.as<int>();
  P.order = T.order->as<int>();
  P.inv = T.inv->as<int>();
  P.sorted = T.sorted->as<R4>();
  P.stype = T.stype->as<int>();
  P.ref = T.ref->as<R>();
  P.half_skin = ctx->half_skin.as<R>();
  P.sorted_hs = T.sorted_hs->as<R>();
  P.vel = ctx->vskin_time > 0 ? (const R *)rp.skin_vel : nullptr;
  P.vs_floor = (R)ctx->vskin_floor;
  P.vs_time = (R)ctx->vskin_time;
  P.vs_cap = (R)ctx->vskin_cap_len;
  P.hs2_dyn = T.hs2_dyn->as<R>();
  P.ext = rp.extent.as<int>();
  P.bsorted = rp.bsorted.as<float4>();
  P.binfo = rp.binfo.as<int>();
  // the build's own constants: fp32 in either precision, every pair radius widened by kBuildMarginF64 in fp64 contexts
  const double bmargin = std::is_same<R, double>::value ? kBuildMarginF64 : 0.0;
  PairConsts<float> cb;
  std::memset(&cb, 0, sizeof(cb));
  for (int k = 0; k < 3; ++k) {
    cb.box[k] = (float)c.box[k];
    cb.invbox[k] = (float)c.invbox[k];
  }
  const float rlb = (float)(ctx->rlist + bmargin), rcb = (float)(ctx->d.cutoff + bmargin);
  for (int k = 0; k < 3; ++k) {
    P.box[k] = c.box[k];
    P.invbox[k] = c.invbox[k];
  }
  P.type_in_entry = ctx->d.ntypes <= kEntryTypes;
  P.dummy_a = P.dummy_b = nullptr;
  if (rp.pad_rows) {
    P.dummy_a = T.sorted->as<R4>() + n;
    if (rp.sorted_alt.p) P.dummy_b = rp.sorted_alt.as<R4>() + n;
    for (int k = 0; k < 3; ++k) {  // (pad_dummy_positions' values)
      const bool open = !(c.box[k] > R(0));
      P.dummy_pos[0][k] = open ? R(1.0e6) : R(0.25) * c.box[k];
      P.dummy_pos[1][k] = open ? R(1.0e6) : R(0.75) * c.box[k];
    }
  }
  static const bool prep_small_on = !(std::getenv("TMDHIP_PREP_SMALL") && std::atoi(std::getenv("TMDHIP_PREP_SMALL")) == 0);
  const bool bin2 = !rp.cell_cap_fallback && rp.ncell <= kScanPlaceMaxCells &&
                    T.members->bytes >= sizeof(int) * (size_t)rp.ncell * kCellCap;
  const bool prep_small = prep_small_on && n <= kPrepSmallMaxAtoms && rp.ncell <= kPrepSmallMaxCells;
  if (plan) {
    // (the batched kernels run the one-launch and the two-launch binning; a replica on the four launches, or with more cells
    // than one block per cell covers, keeps a chain of its own: mode -1)
    constexpr int kMaxBlocks = 16384;
    // (blocks per cell as for a lone replica.  Cutting small grids finer because the chip idles behind a batched chain was
    // measured and is slower — 12 288 atoms x 8 / 5 184 atoms x 16, us per step at 2 / 4 / 8 blocks per cell: 92.7 / 97.7 /
    // 105.8 and 98.5 / 99.9 / 110.2, profiles/r06_replica_batch.txt; TMDHIP_BATCH_BUILD_SPLIT overrides)
    int split = 1;
    if (const char *e = std::getenv("TMDHIP_BUILD_SPLIT")) split = std::max(1, std::min(std::atoi(e), 8));
    else if (const char *e2 = std::getenv("TMDHIP_BATCH_BUILD_SPLIT")) split = std::max(1, std::min(std::atoi(e2), 8));
    else if (rp.ncell <= 1100) split = 2;
    if (rp.ncell * split > kMaxBlocks) split = 1;
    std::memset(plan, 0, sizeof(*plan));
    // (the one-block binning of small systems is a saving of launches for a lone replica — it takes 44 us on its one CU at 5 184
    // atoms; a batch shares its launches among the replicas and bins in parallel: two-launch binning wherever it applies)
    static const bool batch_prep_small = std::getenv("TMDHIP_BATCH_PREP_SMALL") && std::atoi(std::getenv("TMDHIP_BATCH_PREP_SMALL")) != 0;
    plan->mode = (rp.ncell > kMaxBlocks) ? -1 : (bin2 && !(prep_small && batch_prep_small)) ? 1 : prep_small ? 0 : -1;
    plan->g = rp.grid;
    plan->P = P;
    plan->P.pos = nullptr;     // (per launch: ChainSelT)
    plan->P.sorted = nullptr;
    if (rp.pad_rows) {         // both copies get the dummy records, whichever is current
      R4 *a = rp.sorted.as<R4>() + n, *b = rp.sorted_alt.p ? rp.sorted_alt.as<R4>() + n : nullptr;
      plan->P.dummy_a = (b && b < a) ? b : a;
      plan->P.dummy_b = (b && b < a) ? a : b;
    }
    plan->c = cb;
    plan->lg = rp.lg;
    plan->cell_of = T.cell_of->as<int>();
    plan->count = T.count->as<int>();
    plan->members = T.members->as<int>();
    plan->flags = flags;
    plan->slot = T.slot->as<int>();
    plan->cell_start = T.cell_start->as<int>();
    plan->order_tmp = T.order_tmp->as<int>();
    plan->bsorted = rp.bsorted.as<float4>();
    plan->binfo = rp.binfo.as<int>();
    plan->nlist = T.nlist->as<unsigned>();
    plan->nneigh = T.nneigh->as<int>();
    plan->count_zero = plan->mode == 1 ? T.count->as<int>() : nullptr;
    plan->rlist2 = rlb * rlb;
    plan->rcut = rcb;
    plan->ncell = rp.ncell;
    plan->nactive = ctx->nactive;
    plan->type_in_entry = ctx->d.ntypes <= kEntryTypes;
    plan->split = split;
    plan->build_blocks = rp.ncell * split;
    plan->wskin = ctx->half_skin.p != nullptr;
    plan->lpas3 = rp.lg.lpa_shift == 3;
    return 0;
  }
  if (prep_small) {
    hipLaunchKernelGGL((prep_small_kernel<R>), dim3(1), dim3(1024), 0, st, n, pos, rp.grid, rp.ncell, T.cell_of->as<int>(),
                       T.slot->as<int>(), T.cell_start->as<int>(), T.order_tmp->as<int>(), P, flag);
  } else if (bin2) {
    hipLaunchKernelGGL((bin_members_kernel<R>), dim3(nb), dim3(256), 0, st, n, pos, rp.grid, T.cell_of->as<int>(), T.count->as<int>(),
                       T.members->as<int>(), flags, flag);
    hipLaunchKernelGGL((scan_place_kernel<R>), dim3(nb), dim3(256), sizeof(int) * ((size_t)rp.ncell + 1), st, n, rp.ncell,
                       T.count->as<int>(), T.members->as<int>(), T.cell_start->as<int>(), P, flag);
  } else {
    hipLaunchKernelGGL((bin_count_kernel<R>), dim3(nb), dim3(256), 0, st, n, pos, rp.grid, T.cell_of->as<int>(),
                       T.slot->as<int>(), T.count->as<int>(), flag);
    hipLaunchKernelGGL(scan_cells_kernel, dim3(1), dim3(1024), 0, st, rp.ncell, T.count->as<int>(),
                       T.cell_start->as<int>(), flag);
    hipLaunchKernelGGL(fill_cells_kernel, dim3(nb), dim3(256), 0, st, n, T.cell_of->as<int>(), T.slot->as<int>(),
                       T.cell_start->as<int>(), T.order_tmp->as<int>(), flag);
    hipLaunchKernelGGL((place_sorted_kernel<R>), dim3(nb), dim3(256), 0, st, n, P, flag);
  }
  constexpr int kMaxBuildBlocks = 16384;
  const bool wskin = ctx->half_skin.p != nullptr;
  // few cells: several blocks per cell (see build_list_kernel), so that ~2 000 waves are in flight
  int split = 1;
  if (const char *e = std::getenv("TMDHIP_BUILD_SPLIT")) split = std::max(1, std::min(std::atoi(e), 8));
  else if (rp.ncell <= 1100) split = 2;  // measured (water boxes of 5 184 / 12 288 / 41 472 atoms = 343 / 729 / 2 197 cells, us per
                                         // MD step at split 1, 2, 4): 29.7 27.7 (28-37) / 37.8 35.5 35.0 / 43.0 44.6 48.3
  if (rp.ncell > kMaxBuildBlocks) split = 1;
  auto launch_build = [&](auto kernel, int blocks) {
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(64), 0, st_build, n, rp.bsorted.as<float4>(), rp.binfo.as<int>(),
                       T.cell_start->as<int>(), rp.grid, cb, rlb * rlb,
                       rcb, ctx->excl_off.as<int>(), ctx->excl_idx.as<int>(), rp.lg, T.nlist->as<unsigned>(),
                       T.nneigh->as<int>(), flags + F_MAXN, flag, rp.ncell, ctx->nactive, ctx->d.ntypes <= kEntryTypes,
                       debug_timeline_buffer(blocks), split,
                       (bin2 && !(prep_small_on && n <= kPrepSmallMaxAtoms && rp.ncell <= kPrepSmallMaxCells)) ? T.count->as<int>() : nullptr);
  };
#define TMD_BUILD(LOOPED, BLOCKS)                                                                   \
  if (rp.lg.lpa_shift == 3) {                                                                      \
    if (wskin) launch_build(build_list_kernel<float, LOOPED, true, 3>, BLOCKS);                    \
    else launch_build(build_list_kernel<float, LOOPED, false, 3>, BLOCKS);                         \
  } else {                                                                                         \
    if (wskin) launch_build(build_list_kernel<float, LOOPED, true, -1>, BLOCKS);                   \
    else launch_build(build_list_kernel<float, LOOPED, false, -1>, BLOCKS);                        \
  }
  if (rp.ncell <= kMaxBuildBlocks) {
    TMD_BUILD(false, rp.ncell * split)
  } else {
    TMD_BUILD(true, kMaxBuildBlocks)
  }
#undef TMD_BUILD
  TMD_HIP(hipGetLastError());
  return 0;
}


template <typename R>
__global__ void chain_upload_kernel(ChainRepT<R> v, ChainRepT<R> *dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = v;
}

// The rebuild chains of the replicas reps[0 .. nsel) in one launch per kernel (see ChainRepT).  pos[k] / parity[k] / box[k]: what
// enqueue_list_update would have been given for replica reps[k] (its displacement test has run already).  Replicas the batched
// kernels do not cover (four-launch binning, more than 16 384 cells) or that differ in kernel variant get chains of their own.
template <typename R>
int enqueue_chain_batch(tmdhip_ctx *ctx, int nsel, const int *reps, const R *const *pos, const int *parity, const double *const *box,
                        hipStream_t st) {
  using R4 = typename Vec<R>::T4;
  const int n = ctx->d.natoms, nrep = (int)ctx->rep.size();
  std::vector<ChainRepT<R>> plan(nsel);
  std::vector<PairConsts<R>> cs(nsel);
  bool uniform = nsel > 1;
  for (int k = 0; k < nsel; ++k) {
    Replica &rp = ctx->rep[reps[k]];
    cs[k] = make_consts<R>(ctx, box[k]);
    TMD_TRY(enqueue_chain<R>(ctx, rp, pos[k], cs[k], rp.flags.as<int>() + F_REBUILD0 + parity[k], st, &plan[k]));
    uniform = uniform && plan[k].mode >= 0 && plan[k].mode == plan[0].mode && plan[k].wskin == plan[0].wskin && plan[k].lpas3 == plan[0].lpas3;
  }
  if (!uniform) {
    for (int k = 0; k < nsel; ++k) {
      Replica &rp = ctx->rep[reps[k]];
      TMD_TRY(enqueue_chain<R>(ctx, rp, pos[k], cs[k], rp.flags.as<int>() + F_REBUILD0 + parity[k], st));
    }
    return 0;
  }
  {
    // Many replicas in one launch (always so with TMDHIP_REPLICA_REBUILDS=together): the chip is as full as under one big box —
    // where ONE block per cell is the fastest cut (T = 54 us x blocks per cell + 111 us at C3's 6 859 cells,
    // profiles/r06_build_experiments.txt); the two blocks per cell of a small grid are for a replica that rebuilds alone.
    // (Which block builds an atom's row does not change the row: the lists are the same entry for entry.)
    long cells = 0;
    for (int k = 0; k < nsel; ++k) cells += plan[k].ncell;
    if (cells >= 3000 && !std::getenv("TMDHIP_BUILD_SPLIT") && !std::getenv("TMDHIP_BATCH_BUILD_SPLIT"))
      for (int k = 0; k < nsel; ++k) {
        plan[k].split = 1;
        plan[k].build_blocks = plan[k].ncell;
      }
  }
  const size_t row = sizeof(ChainRepT<R>);
  TMD_TRY(ctx->chain_tab.ensure(row * (size_t)nrep));
  if (ctx->chain_host.size() != row * (size_t)nrep) ctx->chain_host.assign(row * (size_t)nrep, 0xA5);  // (matches nothing)
  ChainRepT<R> *tab = ctx->chain_tab.as<ChainRepT<R>>();
  for (int k = 0; k < nsel; ++k) {
    unsigned char *have = ctx->chain_host.data() + row * (size_t)reps[k];
    if (std::memcmp(have, &plan[k], row) != 0) {
      hipLaunchKernelGGL((chain_upload_kernel<R>), dim3(1), dim3(64), 0, st, plan[k], tab + reps[k]);
      TMD_HIP(hipGetLastError());
      std::memcpy(have, &plan[k], row);
    }
  }
  for (int g0 = 0; g0 < nsel; g0 += kBatchMax) {
    const int gn = std::min(kBatchMax, nsel - g0);
    ChainSelT<R> sel;
    std::memset(&sel, 0, sizeof(sel));
    sel.nsel = gn;
    {
      const char *e = std::getenv("TMDHIP_REPLICA_REBUILDS");  // ("together": see chain_any; read per call)
      sel.together = (e && std::strcmp(e, "together") == 0) ? 1 : 0;
    }
    int max_cells = 0, max_blocks = 0;
    for (int k = 0; k < gn; ++k) {
      Replica &rp = ctx->rep[reps[g0 + k]];
      sel.rep[k] = reps[g0 + k];
      sel.pos[k] = pos[g0 + k];
      sel.flag[k] = rp.flags.as<int>() + F_REBUILD0 + parity[g0 + k];
      sel.sorted[k] = rp.sorted.as<R4>();
      max_cells = std::max(max_cells, plan[g0 + k].ncell);
      max_blocks = std::max(max_blocks, plan[g0 + k].build_blocks);
    }
    const int nb = (n + 255) / 256;
    if (plan[0].mode == 0) {
      hipLaunchKernelGGL((prep_small_batch_kernel<R>), dim3(1, gn), dim3(1024), 0, st, n, tab, sel);
    } else {
      hipLaunchKernelGGL((bin_members_batch_kernel<R>), dim3(nb, gn), dim3(256), 0, st, n, tab, sel);
      hipLaunchKernelGGL((scan_place_batch_kernel<R>), dim3(nb, gn), dim3(256), sizeof(int) * ((size_t)max_cells + 1), st, n, tab, sel);
    }
    const dim3 bgrid(max_blocks, gn);
#define TMD_BB(W, L) \
  hipLaunchKernelGGL((build_list_batch_kernel<R, W, L>), bgrid, dim3(64), 0, st, n, ctx->excl_off.as<int>(), ctx->excl_idx.as<int>(), tab, sel)
    if (plan[0].lpas3) {
      if (plan[0].wskin) TMD_BB(true, 3);
      else TMD_BB(false, 3);
    } else {
      if (plan[0].wskin) TMD_BB(true, -1);
      else TMD_BB(false, -1);
    }
#undef TMD_BB
    TMD_HIP(hipGetLastError());
    ctx->batched_chains++;
  }
  return 0;
}
template int enqueue_chain_batch<float>(tmdhip_ctx *, int, const int *, const float *const *, const int *, const double *const *, hipStream_t);

// Enqueue: displacement check -> conditional rebuild chain.  `force` forces a rebuild.
// `prechecked`: the fused MD-step kernel already ran the displacement test of this step.
// `speculate` (plain evaluations through tmdhip_compute, round 6): the three launches of the chain return at once on almost every
// evaluation of a minimiser or of a caller that steps the system itself (the External plugin, step(1) loops), and cost ~15 us of
// a ~100-us evaluation.  The displacement test reports to host-mapped words like the MD loop's (ListCheck::near_host); when the
// PREVIOUS evaluation found no atom beyond 75 % of its limit the chain is left out, and an atom that crosses its limit all the
// same raises F_VIOLATION: tmdhip_compute reads the flags back before it returns, reports "repeat", and the repetition re-plans
// and rebuilds (judge_flags); the next 16 evaluations keep their chain.
template <typename R>
int enqueue_list_update(tmdhip_ctx *ctx, Replica &rp, const R *pos, const PairConsts<R> &c, int force,
                        hipStream_t st, bool prechecked, bool speculate) {
  using R4 = typename Vec<R>::T4;
  const int n = ctx->d.natoms;
  const int parity = (int)(rp.step & 1);
  const int *flag = rp.flags.as<int>() + F_REBUILD0 + parity;
  bool skip = false;
  if (!prechecked) {
    ListCheck<R> k = make_check<R>(ctx, rp);
    const char *e_spec = std::getenv("TMDHIP_SPEC_CHAIN");  // (0 switches it off; read per call: tests toggle it)
    const bool spec_on = !(e_spec && std::atoi(e_spec) == 0);
    if (speculate && spec_on && !force) {
      if (!rp.hostpub) {
        TMD_HIP(hipHostMalloc((void **)&rp.hostpub, 8 * sizeof(unsigned), hipHostMallocMapped));
        for (int w = 0; w < 8; ++w) rp.hostpub[w] = 0u;
        rp.seq = 0;
      }
      volatile unsigned *hp = rp.hostpub;
      // (the previous evaluation's report: its call synchronised with the device before it returned)
      if (rp.spec_valid && rp.spec_backoff == 0) skip = hp[1 + (rp.seq & 1u)] != rp.seq;
      if (rp.spec_backoff > 0) rp.spec_backoff--;
      if (++rp.seq == 0) rp.seq = 1;
      k.near_host = rp.hostpub + 1 + (rp.seq & 1u);
      k.seq = rp.seq;
      k.near_frac2 = (R)(0.75 * 0.75);
      k.skipped = skip ? 1 : 0;
      rp.spec_valid = true;
      rp.seq_valid = false;  // (the MD loop's pacing must not read a plain evaluation's report as its own)
    } else {
      rp.spec_valid = false;
    }
    hipLaunchKernelGGL((check_displacement_kernel<R>), dim3((n + 255) / 256), dim3(256), 0, st, n, pos, k, c,
                       force, rp.inv.as<int>(), ctx->qs.as<R>(), rp.sorted.as<R4>());
  } else {
    rp.spec_valid = false;
  }
  if (skip) {
    rp.chains_skipped++;
    return 0;
  }
  return enqueue_chain<R>(ctx, rp, pos, c, flag, st);
}

template int enqueue_list_update<float>(tmdhip_ctx *, Replica &, const float *, const PairConsts<float> &, int, hipStream_t, bool, bool);
template int enqueue_list_update<double>(tmdhip_ctx *, Replica &, const double *, const PairConsts<double> &, int, hipStream_t, bool, bool);

}  // namespace tmd

using namespace tmd;

extern "C" {

// debug aid (TMDHIP_DEBUG_TIMELINE=1, tools/build_timeline.py): copies the last list build's per-block timeline, returns blocks
int tmdhip_debug_build_timeline(void *out, size_t max_bytes) {
  if (!g_dbg_timeline.p || !out) return 0;
  const size_t bytes = std::min(max_bytes, sizeof(unsigned long long) * 4 * g_dbg_blocks);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpy(out, g_dbg_timeline.p, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int)g_dbg_blocks;
}

}  // extern "C"
