// Atom migration of a brick of the domain decomposition, on the device (gfx950).
//
// The reference has no counterpart (single process, single device).  When an atom has moved
// half the halo skin, every brick re-assigns its atoms to the bricks they now lie in, rebuilds its halo plan (which of
// its atoms each of the 26 neighbour directions sees, with which periodic shift) and swaps the atom set of its force
// engine.  torchmd_amd/domain.py does this with torch operations (argsort, nonzero, bincount, host copies of the
// per-atom parameters): 5-7 ms per migration at 125 000 + 50 000 atoms, 43 us per MD step at a migration every ~170
// steps — as much as the step itself.  Here the same migration is ~20 short launches and four small read-backs:
//
//   own phase    classify (brick of every owned atom, counts per destination) -> counts to the host, count exchange ->
//                pack rows by destination -> row exchange (tmdhip_comm: RCCL or in-process) -> sort by global id
//                (hipcub radix sort: the local order never depends on the migration history) -> unpack into the
//                caller's arrays (wrapped position, image offset, velocity, charge, type, mass, migration reference)
//   halo phase   26-bit message mask per atom + per-block counts -> scan per message -> totals to the host, count
//                exchange -> stable fill of the send list (message order, atom order inside a message: reproducible) ->
//                static payload (position + shift, charge, type) -> exchange -> halo rows, engine atom set (scaled
//                charges, LJ classes: tmdhip_update_atoms' arrays written on the device)
//
// Every collective runs before anything of the caller's is overwritten, into library-owned scratch; a capacity that
// turns out too small is reported (return 2, need_* set) and the call resumes where it stopped once the caller has
// grown its arrays.  Arithmetic mirrors domain.py (BrickGrid.owner, HaloPlan) operation by operation, contraction off.
#include <hipcub/hipcub.hpp>

#include "dd_comm.h"

using namespace tmd;

namespace {

constexpr int kW = 12;      // words (of the run precision) per migrating atom: id (8 bytes), global xyz, v, q, type, m
constexpr int kWH = 5;      // words per halo row of the static payload: shifted xyz, charge, type
constexpr int kMsg = 26;    // directed messages of a brick
constexpr int kMaxWorld = 64;

template <typename R>
struct Geo {
  R box[3], edge[3];
  int dims[3];
};

struct IntRow {
  int v[kMaxWorld];
};

template <typename R>
struct MsgTable {
  signed char dir[kMsg][3];  // -1 / 0 / +1 per axis: near the lower face / anywhere / near the upper face
  R shift[kMsg][3];
  R lo_thr[3], hi_thr[3];    // w < lo_thr: near the lower face; w >= hi_thr: near the upper face
  int base[kMsg];            // first row of message m in the send list (fill)
};

// BrickGrid.owner (domain.py): wrapped position and the rank of the brick it lies in
template <typename R>
__device__ __forceinline__ int brick_of(const Geo<R> &g, const R (&x)[3], R (&w)[3]) {
#pragma clang fp contract(off)
  int c[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const R q = x[k] / g.box[k];
    R wk = x[k] - floor(q) * g.box[k];
    if (wk >= g.box[k]) wk = wk - g.box[k];
    w[k] = wk;
    const R f = floor(wk / g.edge[k]);
    long long ci = (long long)f;
    if (ci > g.dims[k] - 1) ci = g.dims[k] - 1;
    if (ci < 0) ci = 0;
    c[k] = (int)ci;
  }
  return (c[0] * g.dims[1] + c[1]) * g.dims[2] + c[2];
}

template <typename R>
__global__ __launch_bounds__(256) void mig_classify_kernel(int64_t nown, const R *__restrict__ pos, Geo<R> g, int world,
                                                           int *__restrict__ dest, int *__restrict__ counts) {
  __shared__ int s_cnt[kMaxWorld];
  if (threadIdx.x < kMaxWorld) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nown) {
    const R x[3] = {pos[3 * i + 0], pos[3 * i + 1], pos[3 * i + 2]};
    R w[3];
    const int d = brick_of<R>(g, x, w);
    dest[i] = d;
    atomicAdd(&s_cnt[d], 1);
  }
  __syncthreads();
  if ((int)threadIdx.x < world && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
}

template <typename R>
__device__ __forceinline__ void put_id(R *o, int64_t id) {
  __builtin_memcpy(o, &id, 8);  // (fp32: two words, fp64: one)
}
template <typename R>
__device__ __forceinline__ int64_t get_id(const R *o) {
  int64_t id;
  __builtin_memcpy(&id, o, 8);
  return id;
}

template <typename R>
__global__ __launch_bounds__(256) void mig_pack_kernel(int64_t nown, const int *__restrict__ dest, const int64_t *__restrict__ ids,
                                                       const R *__restrict__ pos, const R *__restrict__ unwrap,
                                                       const R *__restrict__ vel, const R *__restrict__ charge,
                                                       const int *__restrict__ type, const R *__restrict__ mass, int world,
                                                       IntRow seg_off, int *__restrict__ cursor, R *__restrict__ rows) {
  // rows of one destination are contiguous; their order inside the segment is irrelevant (the receiver sorts by id):
  // a block takes its share of every segment with one atomic per destination it holds
  __shared__ int s_cnt[kMaxWorld], s_base[kMaxWorld];
  if (threadIdx.x < kMaxWorld) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int d = 0, r = 0;
  if (i < nown) {
    d = dest[i];
    r = atomicAdd(&s_cnt[d], 1);
  }
  __syncthreads();
  if ((int)threadIdx.x < world && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], s_cnt[threadIdx.x]);
  __syncthreads();
  if (i >= nown) return;
  R *o = rows + (size_t)(seg_off.v[d] + s_base[d] + r) * kW;
  put_id<R>(o, ids[i]);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    o[2 + k] = pos[3 * i + k] + unwrap[3 * i + k];  // the caller's periodic image (Domain.state_rows)
    o[5 + k] = vel[3 * i + k];
  }
  o[8] = charge[i];
  o[9] = (R)type[i];
  o[10] = mass[i];
  o[11] = R(0);
}

template <typename R>
__global__ void mig_keys_kernel(int64_t n, const R *__restrict__ rows, unsigned long long *__restrict__ keys, int *__restrict__ perm) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  keys[j] = (unsigned long long)get_id<R>(rows + (size_t)j * kW);
  perm[j] = (int)j;
}

template <typename R>
__global__ __launch_bounds__(256) void mig_unpack_kernel(int64_t n, const R *__restrict__ rows, const unsigned long long *__restrict__ keys,
                                                         const int *__restrict__ perm, Geo<R> g, int64_t *__restrict__ ids,
                                                         R *__restrict__ pos, R *__restrict__ unwrap, R *__restrict__ vel,
                                                         R *__restrict__ charge, int *__restrict__ type, R *__restrict__ mass,
                                                         R *__restrict__ ref) {
#pragma clang fp contract(off)
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const R *o = rows + (size_t)perm[j] * kW;
  ids[j] = (int64_t)keys[j];
  const R x[3] = {o[2], o[3], o[4]};
  R w[3];
  (void)brick_of<R>(g, x, w);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    pos[3 * j + k] = w[k];           // own positions live in the wrapped frame (Domain.adopt)
    unwrap[3 * j + k] = x[k] - w[k];
    ref[3 * j + k] = w[k];
    vel[3 * j + k] = o[5 + k];
  }
  charge[j] = o[8];
  type[j] = (int)o[9];
  mass[j] = o[10];
}

// ---- halo plan (HaloPlan in domain.py) -------------------------------------------------------------------------
template <typename R>
__global__ __launch_bounds__(256) void plan_mask_kernel(int64_t nown, const R *__restrict__ pos, MsgTable<R> T,
                                                        unsigned *__restrict__ mask, int nblk, int *__restrict__ blk_cnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned mk = 0;
  if (i < nown) {
    bool lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const R w = pos[3 * i + k];
      lo[k] = w < T.lo_thr[k];
      hi[k] = w >= T.hi_thr[k];
    }
    for (int m = 0; m < kMsg; ++m) {
      bool in = true;
#pragma unroll
      for (int k = 0; k < 3; ++k) in = in && (T.dir[m][k] == 0 || (T.dir[m][k] < 0 ? lo[k] : hi[k]));
      mk |= in ? (1u << m) : 0u;
    }
    mask[i] = mk;
  }
  for (int m = 0; m < kMsg; ++m) {
    const int c = __syncthreads_count((mk >> m) & 1u);
    if (threadIdx.x == 0) blk_cnt[m * nblk + blockIdx.x] = c;
  }
}

// one block per message: exclusive prefix of its per-block counts, in place; total -> msg_tot[m]
__global__ __launch_bounds__(1024) void plan_scan_kernel(int nblk, int *__restrict__ blk_cnt, int *__restrict__ msg_tot) {
  __shared__ int wsum[16];
  int *cnt = blk_cnt + (size_t)blockIdx.x * nblk;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int per = (nblk + 1023) / 1024, c0 = t * per;
  int mine = 0;
  for (int k = c0; k < min(c0 + per, nblk); ++k) mine += cnt[k];
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
  for (int k = c0; k < min(c0 + per, nblk); ++k) {
    const int v = cnt[k];
    cnt[k] = run;
    run += v;
  }
  if (t == 1023) msg_tot[blockIdx.x] = run;
}

// stable fill: rows of message m = its atoms in atom order
template <typename R>
__global__ __launch_bounds__(256) void plan_fill_kernel(int64_t nown, const unsigned *__restrict__ mask, MsgTable<R> T, int nblk,
                                                        const int *__restrict__ blk_off, int *__restrict__ send_index,
                                                        R *__restrict__ send_shift) {
  __shared__ int s_w[kMsg][4];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned mk = i < nown ? mask[i] : 0u;
  int rank[kMsg];
  for (int m = 0; m < kMsg; ++m) {
    const unsigned long long b = __ballot((mk >> m) & 1u);
    rank[m] = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) s_w[m][w] = __popcll(b);
  }
  __syncthreads();
  if (!mk) return;
  for (int m = 0; m < kMsg; ++m) {
    if (!((mk >> m) & 1u)) continue;
    int off = T.base[m] + blk_off[m * nblk + blockIdx.x] + rank[m];
    for (int q = 0; q < w; ++q) off += s_w[m][q];
    send_index[off] = (int)i;
#pragma unroll
    for (int k = 0; k < 3; ++k) send_shift[3 * (size_t)off + k] = T.shift[m][k];
  }
}

template <typename R>
__global__ void halo_static_pack_kernel(int64_t nsend, const int *__restrict__ index, const R *__restrict__ shift,
                                        const R *__restrict__ pos, const R *__restrict__ charge, const int *__restrict__ type,
                                        R *__restrict__ out) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nsend) return;
  const int64_t i = index[k];
#pragma unroll
  for (int x = 0; x < 3; ++x) out[kWH * k + x] = pos[3 * i + x] + shift[3 * k + x];
  out[kWH * k + 3] = charge[i];
  out[kWH * k + 4] = (R)type[i];
}

// halo rows of the position buffer + the engine's atom set (tmdhip_update_atoms' arrays, written on the device)
template <typename R>
__global__ void engine_atoms_kernel(int64_t nown, int64_t nhalo, const R *__restrict__ charge, const int *__restrict__ type,
                                    const R *__restrict__ halo_in, R *__restrict__ pos, const int *__restrict__ type_map,
                                    int ntypes_map, double qscale, R *__restrict__ qs, int *__restrict__ types,
                                    int *__restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nown + nhalo) return;
  R q;
  int t;
  if (i < nown) {
    q = charge[i];
    t = type[i];
  } else {
    const R *h = halo_in + (size_t)(i - nown) * kWH;
#pragma unroll
    for (int k = 0; k < 3; ++k) pos[3 * i + k] = h[k];
    q = h[3];
    t = (int)h[4];
  }
  if (type_map) {
    if (t < 0 || t >= ntypes_map) {
      *bad = 1;
      t = 0;
    } else {
      t = type_map[t];
    }
  }
  qs[i] = (R)((double)q * qscale);
  types[i] = t;
}

inline dim3 grid_for(int64_t n, int t = 256) { return dim3((unsigned)std::max<int64_t>((n + t - 1) / t, 1)); }

// the directed messages of a brick in the order of HaloPlan: by destination rank, then by direction index
struct HostPlan {
  int dir[kMsg][3];
  int dest[kMsg];
  double shift[kMsg][3];
};

void make_host_plan(const tmdhip_dd_brick *b, HostPlan &P) {
  const int px = b->dims[0], py = b->dims[1], pz = b->dims[2];
  const int me[3] = {b->rank / (py * pz), (b->rank / pz) % py, b->rank % pz};
  struct Q {
    int d[3], dest, q;
  };
  std::vector<Q> all;
  int q = 0;
  for (int dx = -1; dx <= 1; ++dx)  // itertools.product((-1, 0, 1), repeat=3) without (0, 0, 0)
    for (int dy = -1; dy <= 1; ++dy)
      for (int dz = -1; dz <= 1; ++dz) {
        if (!dx && !dy && !dz) continue;
        Q e;
        e.d[0] = dx, e.d[1] = dy, e.d[2] = dz;
        const int c[3] = {((me[0] + dx) % px + px) % px, ((me[1] + dy) % py + py) % py, ((me[2] + dz) % pz + pz) % pz};
        e.dest = (c[0] * py + c[1]) * pz + c[2];
        e.q = q++;
        all.push_back(e);
      }
  std::stable_sort(all.begin(), all.end(), [](const Q &a, const Q &b2) { return a.dest != b2.dest ? a.dest < b2.dest : a.q < b2.q; });
  for (int m = 0; m < kMsg; ++m) {
    P.dest[m] = all[m].dest;
    for (int k = 0; k < 3; ++k) {
      P.dir[m][k] = all[m].d[k];
      P.shift[m][k] = 0;
      if (all[m].d[k] == -1 && me[k] == 0) P.shift[m][k] = b->box[k];  // crossing the lower global face: seen at +L
      else if (all[m].d[k] == 1 && me[k] == b->dims[k] - 1) P.shift[m][k] = -b->box[k];
    }
  }
}

// TMDHIP_DEBUG_MIGRATE_TIMES=1: wall time of every stage (with a stream synchronisation behind it) on stderr
struct StageClock {
  bool on = std::getenv("TMDHIP_DEBUG_MIGRATE_TIMES") != nullptr;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void lap(const char *what, int rank, hipStream_t st) {
    if (!on) return;
    (void)hipStreamSynchronize(st);
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[tmdhip_dd_migrate rank %d] %-28s %8.1f us\n", rank, what, std::chrono::duration<double, std::micro>(t1 - t0).count());
    t0 = t1;
  }
};

template <typename R>
int migrate(tmdhip_ctx *ctx, tmdhip_comm *c, tmdhip_dd_brick *b, hipStream_t st) {
  auto &S = c->mig;
  StageClock clk;
  const int world = c->world;
  if (!S.host && hipHostMalloc((void **)&S.host, sizeof(int64_t) * 4 * kMaxWorld, hipHostMallocDefault) != hipSuccess)
    return fail("tmdhip_dd_migrate: pinned allocation failed");
  Geo<R> g;
  for (int k = 0; k < 3; ++k) {
    g.box[k] = (R)b->box[k];
    g.edge[k] = (R)(b->box[k] / (double)b->dims[k]);
    g.dims[k] = b->dims[k];
  }
  R *pos = (R *)b->pos_dev, *unwrap = (R *)b->unwrap_dev, *vel = (R *)b->vel_dev, *charge = (R *)b->charge_dev,
    *mass = (R *)b->mass_dev, *ref = (R *)b->ref_dev;
  const int dtype = b->dtype;

  // ---------------- own phase: collectives into scratch ----------------
  if (c->mig_stage == 0) {
    const int64_t nown = b->nown;
    TMD_TRY(S.dest.ensure(sizeof(int) * (size_t)std::max<int64_t>(nown, 1)));
    TMD_TRY(S.counts.ensure(sizeof(int) * 2 * kMaxWorld));
    TMD_HIP(hipMemsetAsync(S.counts.p, 0, sizeof(int) * 2 * kMaxWorld, st));
    int *counts = S.counts.as<int>(), *cursor = counts + kMaxWorld;
    if (nown > 0)
      hipLaunchKernelGGL((mig_classify_kernel<R>), grid_for(nown), dim3(256), 0, st, nown, pos, g, world, S.dest.as<int>(), counts);
    TMD_HIP(hipGetLastError());
    int *hc = (int *)(S.host + 2 * kMaxWorld);
    TMD_HIP(hipMemcpyAsync(hc, counts, sizeof(int) * world, hipMemcpyDeviceToHost, st));
    TMD_HIP(hipStreamSynchronize(st));
    int64_t *sc = S.host, *rc = S.host + kMaxWorld;
    IntRow seg;
    int run = 0;
    for (int p = 0; p < world; ++p) {
      sc[p] = hc[p];
      seg.v[p] = run;
      run += hc[p];
    }
    if (run != nown) return fail("tmdhip_dd_migrate: classification lost atoms (NaN positions?)");
    TMD_TRY(exchange_counts(c, sc, rc, st));
    int64_t nnew = 0;
    for (int p = 0; p < world; ++p) nnew += rc[p];
    TMD_TRY(S.rows_out.ensure(sizeof(R) * kW * (size_t)std::max<int64_t>(nown, 1)));
    TMD_TRY(S.rows_in.ensure(sizeof(R) * kW * (size_t)std::max<int64_t>(nnew, 1)));
    if (nown > 0)
      hipLaunchKernelGGL((mig_pack_kernel<R>), grid_for(nown), dim3(256), 0, st, nown, S.dest.as<int>(), b->ids_dev, pos, unwrap, vel,
                         charge, b->type_dev, mass, world, seg, cursor, S.rows_out.as<R>());
    TMD_HIP(hipGetLastError());
    TMD_TRY(exchange_rows(c, dtype, S.rows_out.p, sc, S.rows_in.p, rc, kW, st));
    c->mig_nnew = nnew;
    c->mig_stage = 1;
    clk.lap("classify + row exchange", c->rank, st);
  }
  // ---------------- own phase: into the caller's arrays ----------------
  if (c->mig_stage == 1) {
    const int64_t n = c->mig_nnew;
    if (n > b->cap_own || n > b->cap_rows) {
      b->need_own = n;
      b->need_rows = n;
      b->need_send = 0;
      return 2;
    }
    if (n >= ((int64_t)1 << 30)) return fail("tmdhip_dd_migrate: brick too large");
    if (n > 0) {
      TMD_TRY(S.keys_in.ensure(sizeof(unsigned long long) * (size_t)n));
      TMD_TRY(S.keys_out.ensure(sizeof(unsigned long long) * (size_t)n));
      TMD_TRY(S.perm_in.ensure(sizeof(int) * (size_t)n));
      TMD_TRY(S.perm_out.ensure(sizeof(int) * (size_t)n));
      hipLaunchKernelGGL((mig_keys_kernel<R>), grid_for(n), dim3(256), 0, st, n, S.rows_in.as<R>(), S.keys_in.as<unsigned long long>(),
                         S.perm_in.as<int>());
      size_t tmp = 0;
      TMD_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, S.keys_in.as<unsigned long long>(), S.keys_out.as<unsigned long long>(),
                                                  S.perm_in.as<int>(), S.perm_out.as<int>(), (int)n, 0, 64, st));
      TMD_TRY(S.sort_tmp.ensure(std::max<size_t>(tmp, 16)));
      tmp = S.sort_tmp.bytes;
      TMD_HIP(hipcub::DeviceRadixSort::SortPairs(S.sort_tmp.p, tmp, S.keys_in.as<unsigned long long>(), S.keys_out.as<unsigned long long>(),
                                                  S.perm_in.as<int>(), S.perm_out.as<int>(), (int)n, 0, 64, st));
      hipLaunchKernelGGL((mig_unpack_kernel<R>), grid_for(n), dim3(256), 0, st, n, S.rows_in.as<R>(), S.keys_out.as<unsigned long long>(),
                         S.perm_out.as<int>(), g, b->ids_dev, pos, unwrap, vel, charge, b->type_dev, mass, ref);
      TMD_HIP(hipGetLastError());
    }
    TMD_HIP(hipMemsetAsync(b->disp2_dev, 0, sizeof(uint32_t), st));
    b->nown = n;
    c->mig_stage = 2;
    clk.lap("sort by id + unpack", c->rank, st);
  }
  // ---------------- halo phase: plan + collectives into scratch ----------------
  HostPlan P;
  make_host_plan(b, P);
  if (c->mig_stage == 2) {
    const int64_t nown = b->nown;
    for (int k = 0; k < 3; ++k)
      if (b->box[k] / b->dims[k] < b->halo) return fail("tmdhip_dd_migrate: bricks are thinner than the halo: use fewer ranks");
    const int py = b->dims[1], pz = b->dims[2];
    const int me[3] = {b->rank / (py * pz), (b->rank / pz) % py, b->rank % pz};
    MsgTable<R> T;
    for (int k = 0; k < 3; ++k) {
      const double edge = b->box[k] / (double)b->dims[k];
      const R lo = (R)((double)me[k] * edge), hi = (R)((double)me[k] * edge + edge);
      T.lo_thr[k] = lo + (R)b->halo;   // (tensor of the run precision + Python float: rounded in the run precision)
      T.hi_thr[k] = hi - (R)b->halo;
    }
    for (int m = 0; m < kMsg; ++m)
      for (int k = 0; k < 3; ++k) {
        T.dir[m][k] = (signed char)P.dir[m][k];
        T.shift[m][k] = (R)P.shift[m][k];
        T.base[m] = 0;
      }
    const int nblk = (int)std::max<int64_t>((nown + 255) / 256, 1);
    TMD_TRY(S.mask.ensure(sizeof(unsigned) * (size_t)std::max<int64_t>(nown, 1)));
    TMD_TRY(S.blk_cnt.ensure(sizeof(int) * (size_t)kMsg * nblk));
    TMD_TRY(S.msg_tot.ensure(sizeof(int) * kMsg));
    hipLaunchKernelGGL((plan_mask_kernel<R>), dim3(nblk), dim3(256), 0, st, nown, pos, T, S.mask.as<unsigned>(), nblk, S.blk_cnt.as<int>());
    hipLaunchKernelGGL(plan_scan_kernel, dim3(kMsg), dim3(1024), 0, st, nblk, S.blk_cnt.as<int>(), S.msg_tot.as<int>());
    TMD_HIP(hipGetLastError());
    int *ht = (int *)(S.host + 2 * kMaxWorld);
    TMD_HIP(hipMemcpyAsync(ht, S.msg_tot.p, sizeof(int) * kMsg, hipMemcpyDeviceToHost, st));
    TMD_HIP(hipStreamSynchronize(st));
    int64_t *sc = S.host, *rc = S.host + kMaxWorld;
    for (int p = 0; p < world; ++p) sc[p] = 0;
    int64_t nsend = 0;
    for (int m = 0; m < kMsg; ++m) {
      T.base[m] = (int)nsend;
      nsend += ht[m];
      sc[P.dest[m]] += ht[m];
    }
    TMD_TRY(exchange_counts(c, sc, rc, st));
    int64_t nhalo = 0;
    for (int p = 0; p < world; ++p) nhalo += rc[p];
    for (int p = 0; p < world; ++p) {
      c->mig_send_counts[p] = sc[p];
      c->mig_recv_counts[p] = rc[p];
    }
    // the send list in scratch (the caller's arrays may be too small), then the static payload from it
    TMD_TRY(S.perm_in.ensure(sizeof(int) * (size_t)std::max<int64_t>(nsend, 1)));
    TMD_TRY(S.rows_out.ensure(sizeof(R) * (3 + kWH) * (size_t)std::max<int64_t>(nsend, 1)));
    int *sidx = S.perm_in.as<int>();
    R *sshift = S.rows_out.as<R>(), *payload = sshift + 3 * (size_t)std::max<int64_t>(nsend, 1);
    TMD_TRY(S.halo_in.ensure(sizeof(R) * kWH * (size_t)std::max<int64_t>(nhalo, 1)));
    if (nsend > 0) {
      hipLaunchKernelGGL((plan_fill_kernel<R>), dim3(nblk), dim3(256), 0, st, nown, S.mask.as<unsigned>(), T, nblk, S.blk_cnt.as<int>(), sidx,
                         sshift);
      hipLaunchKernelGGL((halo_static_pack_kernel<R>), grid_for(nsend), dim3(256), 0, st, nsend, sidx, sshift, pos, charge, b->type_dev,
                         payload);
      TMD_HIP(hipGetLastError());
    }
    TMD_TRY(exchange_rows(c, dtype, payload, sc, S.halo_in.p, rc, kWH, st));
    c->mig_nsend = nsend;
    c->mig_nhalo = nhalo;
    c->mig_stage = 3;
    clk.lap("halo plan + static exchange", c->rank, st);
  }
  // ---------------- halo phase: into the caller's arrays and the engine ----------------
  {
    const int64_t nown = b->nown, nsend = c->mig_nsend, nhalo = c->mig_nhalo, n = nown + nhalo;
    if (n > b->cap_rows || nsend > b->cap_send) {
      b->need_own = nown;
      b->need_rows = n;
      b->need_send = nsend;
      return 2;
    }
    if (n <= 0 || n >= (1 << 23)) return fail("tmdhip_dd_migrate: atoms of a brick out of range");
    if (nsend > 0) {
      TMD_HIP(hipMemcpyAsync(b->send_index_dev, S.perm_in.p, sizeof(int) * (size_t)nsend, hipMemcpyDeviceToDevice, st));
      TMD_HIP(hipMemcpyAsync(b->send_shift_dev, S.rows_out.p, sizeof(R) * 3 * (size_t)nsend, hipMemcpyDeviceToDevice, st));
    }
    // the engine's atom set (what tmdhip_update_atoms does from host arrays)
    if (ctx->nexcl != 0 || ctx->bonded) return fail("tmdhip_dd_migrate: only for atomic systems (no exclusions, no bonded terms)");
    if (ctx->d.dtype != dtype) return fail("tmdhip_dd_migrate: dtype of the context differs");
    ctx->d.natoms = (int)n;
    ctx->nactive = nown > 0 ? (int)nown : 0x7fffffff;
    TMD_TRY(ctx->types.ensure(sizeof(int) * (size_t)n));
    TMD_TRY(ctx->qs.ensure(sizeof(R) * (size_t)n));
    TMD_TRY(ctx->excl_off.ensure(sizeof(int) * ((size_t)n + 1)));
    TMD_HIP(hipMemsetAsync(ctx->excl_off.p, 0, sizeof(int) * ((size_t)n + 1), st));
    const int *tmap = nullptr;
    if (b->type_map_host && b->ntypes_map > 0) {
      TMD_TRY(S.typemap.ensure(sizeof(int) * ((size_t)b->ntypes_map + 1)));
      TMD_HIP(hipMemcpyAsync(S.typemap.p, b->type_map_host, sizeof(int) * (size_t)b->ntypes_map, hipMemcpyHostToDevice, st));
      tmap = S.typemap.as<int>();
    }
    // an atom type outside the map is reported by the next tmdhip_dd_run (no read-back of its own here)
    TMD_TRY(S.bad.ensure(sizeof(int)));
    int *bad = S.bad.as<int>();
    TMD_HIP(hipMemsetAsync(bad, 0, sizeof(int), st));
    c->mig_bad_pending = true;
    hipLaunchKernelGGL((engine_atoms_kernel<R>), grid_for(n), dim3(256), 0, st, nown, nhalo, charge, b->type_dev, S.halo_in.as<R>(), pos, tmap,
                       b->ntypes_map, std::sqrt(kElecFactor), ctx->qs.as<R>(), ctx->types.as<int>(), bad);
    TMD_HIP(hipGetLastError());
    ctx->half_skin.release();  // per-atom skins belonged to the old atom set
    ctx->half_skin2.release();
    ctx->rlist = ctx->d.cutoff > 0 ? ctx->d.cutoff + ctx->skin : 0;
    ctx->mean_list_scale = 1;
    for (auto &rp : ctx->rep) {  // the next compute re-plans the grid, re-sizes the buffers and rebuilds
      rp.have_list = false;
      if (rp.lg.maxn > 0) rp.maxn_keep = rp.lg.maxn;
      rp.lg.maxn = 0;
    }
    // the bounds of [own | halo] are known: the brick plus its halo (+ the room atoms have until the next migration);
    // the re-plan then needs no read-back of the positions
    {
      const int py = b->dims[1], pz = b->dims[2];
      const int me[3] = {b->rank / (py * pz), (b->rank / pz) % py, b->rank % pz};
      for (int k = 0; k < 3; ++k) {
        const double edge = b->box[k] / (double)b->dims[k];
        ctx->open_lo[k] = me[k] * edge - b->halo;
        ctx->open_hi[k] = me[k] * edge + edge + b->halo;
      }
      ctx->open_bounds_valid = true;
    }
    b->nhalo = nhalo;
    b->nsend = nsend;
    for (int p = 0; p < world; ++p) {
      b->send_counts_host[p] = c->mig_send_counts[p];
      b->recv_counts_host[p] = c->mig_recv_counts[p];
    }
    // the migration trigger starts over; the per-atom index of the send list is of the old list
    c->pending = false;
    c->at = 0;
    c->csr_index = nullptr;
    c->mig_stage = 0;
    clk.lap("engine atom set", c->rank, st);
  }
  return 0;
}

}  // namespace

extern "C" {

int tmdhip_dd_migrate(tmdhip_ctx *ctx, tmdhip_comm *comm, tmdhip_dd_brick *b, void *stream) {
  if (!ctx || !comm || !b) return fail("tmdhip_dd_migrate: null argument");
  if (b->struct_size != (int32_t)sizeof(tmdhip_dd_brick)) return fail("tmdhip_dd_migrate: struct_size mismatch (ABI)");
  if (b->dtype != TMDHIP_F32 && b->dtype != TMDHIP_F64) return fail("tmdhip_dd_migrate: bad dtype");
  if (b->world != comm->world || b->rank != comm->rank) return fail("tmdhip_dd_migrate: rank / world differ from the communicator's");
  if (b->world > kMaxWorld || b->dims[0] * b->dims[1] * b->dims[2] != b->world || b->dims[0] < 1 || b->dims[1] < 1 || b->dims[2] < 1)
    return fail("tmdhip_dd_migrate: bad brick grid");
  if (!(b->box[0] > 0 && b->box[1] > 0 && b->box[2] > 0) || !(b->halo > 0)) return fail("tmdhip_dd_migrate: box and halo must be positive");
  if (!b->ids_dev || !b->pos_dev || !b->unwrap_dev || !b->vel_dev || !b->charge_dev || !b->type_dev || !b->mass_dev || !b->ref_dev ||
      !b->disp2_dev || !b->send_index_dev || !b->send_shift_dev || !b->send_counts_host || !b->recv_counts_host)
    return fail("tmdhip_dd_migrate: null pointer");
  if (b->nown < 0 || b->nown > b->cap_own || b->cap_own > b->cap_rows) return fail("tmdhip_dd_migrate: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  const int rc = b->dtype == TMDHIP_F32 ? migrate<float>(ctx, comm, b, st) : migrate<double>(ctx, comm, b, st);
  // only a return of 2 (capacity too small: grow and call again) is resumable; after an error the next call starts from
  // the beginning on every rank instead of resuming mid-way on stale scratch while the other ranks start at stage 0
  if (rc < 0) comm->mig_stage = 0;
  if (rc < 0 && comm->hub) comm->hub->abort();  // (in-process transport: the other ranks must not wait 30 s for this one)
  return rc;
}

}  // extern "C"
