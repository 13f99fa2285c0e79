// Context of the nonbonded engine for gfx950 (MI355X): life cycle, grid planning, the orchestration of list
// (re)builds and pair launches, and the C entry points of include/tmdhip.h that are not part of the MD loop.
//
// Reference semantics: torchmd/forces.py:27-74 (Forces.__init__, _make_indeces: the pair set as exclusion CSR instead
// of a dense index tensor) and forces.py:260-319 (the nonbonded block of Forces.compute).
#include "engine.h"

namespace tmd {

std::string &last_error() {
  static thread_local std::string s;
  return s;
}
int fail(const std::string &msg) {
  last_error() = msg;
  return -1;
}

// opaque accessors for bonded.hip (it does not include engine.h)
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

int pick_lpa(int64_t n, int capacity) {  // n: atoms whose waves share a launch (the replicas of a batched launch count together)
  if (const char *e = std::getenv("TMDHIP_LPA")) {  // tuning override: lanes per atom (power of two, 1..64)
    const int v = std::atoi(e);
    if (v >= 4 && v <= 64 && (v & (v - 1)) == 0) return v;
  }
  // (1) enough waves to fill the chip: >= 2 560 (10 per CU).  Rounds 1-3 asked for 8 192 "to hide list / gather latency";
  //     measured in round 4 on water boxes (us per MD step at 8 / 16 / 32 / 64 lanes): 12 288 atoms 21.9 / 21.0 / 22.7 /
  //     23.1, 24 000 atoms 27.4 / 29.8 / 32.5 / 35.4, 41 472 atoms 36.3 / 36.5 / 50.5 / 50.4 — fewer, longer waves
  //     amortise a wave's prologue and reduction better than more waves hide latency.  TMDHIP_LPA_WAVES overrides.
  int64_t want_waves = 2560;
  if (const char *e = std::getenv("TMDHIP_LPA_WAVES")) want_waves = std::max(1, std::atoi(e));
  int lpa = 1;
  while (lpa < 64 && n * lpa < want_waves * 64) lpa <<= 1;
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

// Padded list rows (engine.h: pad_entry_for; the pair waves write the padding, pair_fast_f32.hip): fp32 contexts whose entries can address the two dummy records with bits 20..22
// of the slot clear, in a box where one of the dummies is always out of reach — they are half a box diagonal apart, so
// one of them is >= a quarter of the diagonal from any atom at build time, and an atom moves less than one skin before
// the next build.  A box with an open dimension has them 10^6 A out.  TMDHIP_PAD_ROWS=0 switches the padding off.
bool plan_pad_rows(const tmdhip_ctx *ctx, const double *box) {
  const char *e = std::getenv("TMDHIP_PAD_ROWS");  // (read at every re-plan: tests switch it between two evaluations)
  if ((e && std::atoi(e) == 0) || ctx->d.dtype != TMDHIP_F32 || (int64_t)ctx->d.natoms + 2 > (1 << 20)) return false;
  double diag2 = 0;
  for (int k = 0; k < 3; ++k) {
    if (!(box[k] > 0)) return true;
    diag2 += box[k] * box[k];
  }
  return 0.25 * std::sqrt(diag2) > ctx->rlist + 2.0 * ctx->skin;
}

// kLmStream: lists whose rows exceed 384 MB in total (capacity: ~1.4 x the entries) cannot live in the 256 MiB Infinity
// Cache from one launch to the next.  TMDHIP_LIST_STREAM=0 / 1 overrides (A/B).
bool list_streams(const tmdhip_ctx *ctx, const Replica &rp) {
  if (const char *e = std::getenv("TMDHIP_LIST_STREAM")) return std::atoi(e) != 0;
  return (size_t)ctx->d.natoms * (size_t)rp.lg.maxn * 4u > ((size_t)384 << 20);
}

// The events that time a pair launch (tmdhip_timing_enable) carry no system-scope fence: nobody reads device memory on the
// strength of them, and a default event makes its launch release to the system — an L2 write-back in front of the next launch
// of the step loop (round 5, profiles/r05_event_fence_ab.txt: the event-timed launches of the 2 000-step run 46.85 -> 45.0 us,
// the kernel trace's 45.4; 20-step runs 69.4 -> 68.5 us per step).
constexpr unsigned kTimingEventFlags = hipEventDisableSystemFence;

template <typename R>
int alloc_replica(tmdhip_ctx *ctx, Replica &rp, int maxn) {
  using R4 = typename Vec<R>::T4;
  const int n = ctx->d.natoms;
  TMD_TRY(rp.cell_of.ensure(sizeof(int) * n));
  TMD_TRY(rp.slot.ensure(sizeof(int) * n));
  TMD_TRY(rp.order_tmp.ensure(sizeof(int) * n));
  TMD_TRY(rp.order.ensure(sizeof(int) * n));
  TMD_TRY(rp.inv.ensure(sizeof(int) * n));
  // (+ 2: the dummy records the padding entries of a list point at, Replica::pad_rows)
  TMD_TRY(rp.sorted.ensure(sizeof(R4) * ((size_t)n + 2)));
  TMD_TRY(rp.sorted_alt.ensure(sizeof(R4) * ((size_t)n + 2)));
  TMD_TRY(rp.stype.ensure(sizeof(int) * n));
  TMD_TRY(rp.bsorted.ensure(sizeof(float4) * (size_t)n));  // (the build's records are fp32 in either precision)
  TMD_TRY(rp.binfo.ensure(sizeof(int) * (size_t)n));
  if (ctx->half_skin.p) {
    TMD_TRY(rp.sorted_hs.ensure(ctx->real_size * (size_t)n));
    TMD_TRY(rp.hs2_dyn.ensure(ctx->real_size * (size_t)n));
  }
  TMD_TRY(rp.ref.ensure(sizeof(R) * 3 * n));
  TMD_TRY(rp.nneigh.ensure(sizeof(int) * n));
  // lanes per atom from the MEAN list length (capacities are sized for the longest lists); fixed once a list exists
  // (fp32 contexts with several replicas make ONE launch for all of them, md_loop.hip: their waves count together)
  const int64_t launch_atoms = (int64_t)n * (ctx->d.dtype == TMDHIP_F32 ? std::min<int64_t>((int64_t)ctx->rep.size(), kBatchMax) : 1);
  if (rp.lg.maxn == 0 || !rp.have_list) rp.lg.lpa = pick_lpa(launch_atoms, (int)((maxn - 32) * ctx->mean_list_scale) + 32);
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
  // one word per wave group: the rebuild count at which the pair waves padded the group's rows (0: never)
  if (rp.padgen.bytes < sizeof(int) * (groups + 8)) {  // (+ the surplus waves of the last pair block)
    TMD_TRY(rp.padgen.ensure(sizeof(int) * (groups + 8)));
    TMD_HIP(hipMemset(rp.padgen.p, 0, rp.padgen.bytes));
    // the caller's stream need not be ordered behind the null stream (a stream created non-blocking — every side stream
    // of PyTorch — is not), and a memset of device memory may return before it is done: wait for it here (rare path)
    TMD_HIP(hipStreamSynchronize(nullptr));
  }
  return 0;
}

template <typename R>
int compute_list(tmdhip_ctx *ctx, Replica &rp, const void *pos_v, const double *box, void *forces,
                 double *energies, int flags, hipStream_t st, const FusedLaunchT<R> *fused, ListOnlyOut *list_only) {
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
    if (!periodic && ctx->open_bounds_valid) {
      for (int k = 0; k < 3; ++k) lo[k] = ctx->open_lo[k], hi[k] = ctx->open_hi[k];
      ctx->open_bounds_valid = false;
      volume = std::max(hi[0] - lo[0], ctx->rlist) * std::max(hi[1] - lo[1], ctx->rlist) *
               std::max(hi[2] - lo[2], ctx->rlist);
    } else if (!periodic) {
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
    if (const char *e = std::getenv("TMDHIP_BIN2"))  // (A/B, tests: 0 = the four-launch binning; read at every re-plan)
      if (std::atoi(e) == 0) rp.cell_cap_fallback = true;
    if (rp.ncell <= kScanPlaceMaxCells && !rp.cell_cap_fallback)  // two-launch binning: the cells' member arrays
      TMD_TRY(rp.members.ensure(sizeof(int) * (size_t)rp.ncell * kCellCap));
    if (!rp.have_list) {
      const double dens = n / volume;
      int est = (int)(dens * 4.18879 * ctx->rlist * ctx->rlist * ctx->rlist * 1.3) + 32;
      est = std::max(est, rp.maxn_keep);
      est = std::min(est, std::max(n - 1, 1));
      TMD_TRY(alloc_replica<R>(ctx, rp, est));
    }
    for (int k = 0; k < 3; ++k) rp.box[k] = box[k];
    rp.pad_rows = plan_pad_rows(ctx, box);
    force = 1;
  }
  for (int attempt = 0; attempt < 8; ++attempt) {
    if (!force && (flags & kSkipChain)) {  // (the integrator kernel has run this step's test with `skipped` set)
      rp.step++;
      rp.chains_skipped++;
      break;
    }
    if (!force && (flags & kDeferChain) && (flags & kPrechecked) && list_only) {  // (the caller enqueues the chain, with other replicas')
      list_only->chain = 1;
      list_only->chain_parity = (int)(rp.step & 1);
      rp.step++;
      break;
    }
    TMD_TRY(enqueue_list_update<R>(ctx, rp, pos, c, force, st, !force && (flags & kPrechecked), (flags & kSpecChain) != 0));
    rp.step++;
    if (!force) break;
    // forced builds are host-visible: size the list from the observed maximum so that later
    // device-side rebuilds have headroom (density fluctuations) without host involvement
    int h[F_COUNT];
    TMD_HIP(hipMemcpyAsync(h, rp.flags.p, sizeof(h), hipMemcpyDeviceToHost, st));
    TMD_HIP(hipStreamSynchronize(st));
    rp.host_rebuilds++;
    if (h[F_CELLCAP]) {  // a cell overflowed the member array of the two-launch binning: build again with the four launches
      TMD_HIP(hipMemsetAsync(rp.flags.as<int>() + F_CELLCAP, 0, sizeof(int), st));
      rp.cell_cap_fallback = true;
      continue;
    }
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
  if (flags & kListOnly) {  // the caller launches (md_run's replica batch): rp.step counts the NEXT step by now
    if (!list_only) return fail("compute_list: kListOnly without an output block");
    list_only->lmode = ((flags & kViolationCheck) ? kLmViolation : 0) | (((rp.step - 1) & 1) ? kLmParity : 0) |
                       (rp.pad_rows ? kLmPadded : 0) | (list_streams(ctx, rp) ? kLmStream : 0);
    list_only->next_parity = (int)(rp.step & 1);
    return 0;
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
  const bool timeable = ctx->timing && !(ctx->timing_interior_only && (flags & TMDHIP_WANT_ENERGY));
  const int64_t seen = timeable ? ctx->timing_seen++ : -1;
  const bool timed = timeable && seen >= 0 && (seen % ctx->timing_stride) == 0 &&
                     (ctx->timing_limit == 0 || ctx->timing_taken < ctx->timing_limit);
  if (timed) ctx->timing_taken++;
  if (timed) {
    if (ctx->events_used >= 4096) TMD_TRY(tmdhip_timing_read(ctx, nullptr, nullptr, 0));
    if (ctx->events_used == ctx->events.size()) {
      hipEvent_t a, b;
      TMD_HIP(hipEventCreateWithFlags(&a, kTimingEventFlags));
      TMD_HIP(hipEventCreateWithFlags(&b, kTimingEventFlags));
      ctx->events.emplace_back(a, b);
    }
    e0 = ctx->events[ctx->events_used].first;
    e1 = ctx->events[ctx->events_used].second;
    ctx->events_used++;
  }
  const int overwrite = (flags & TMDHIP_OVERWRITE_FORCES) ? 1 : 0;
  // list duties of the pair launch's first thread: rp.step counts the NEXT step by now
  const int lmode = ((flags & kViolationCheck) ? kLmViolation : 0) | (((rp.step - 1) & 1) ? kLmParity : 0) |
                    (rp.pad_rows ? kLmPadded : 0) | (list_streams(ctx, rp) ? kLmStream : 0);
  FusedLaunchT<R> fl{};
  if (fused) {
    fl = *fused;
    fl.step.parity = (int)(rp.step & 1);
  }
  if (flags & TMDHIP_WANT_ENERGY)  // (fused: the final step of an MD call — md_run folds the scratch rows, with the kinetic energy)
    TMD_TRY((launch_list_pair<R, true>(ctx, rp, c, f, overwrite, energies, pc, st, e0, e1, lmode, fused ? &fl : nullptr,
                                       !(flags & kDeferFold) && !fused)));
  else
    TMD_TRY((launch_list_pair<R, false>(ctx, rp, c, f, overwrite, energies, pc, st, e0, e1, lmode, fused ? &fl : nullptr)));
  if (pc) TMD_TRY(halve_pair_count(pc, st));
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

// verdict on the list flags of one replica (already on the host): 0 valid, 1 repeat the work, < 0 error
int judge_flags(tmdhip_ctx *ctx, Replica &rp, const int *h, hipStream_t st) {
  if (h[F_STEP_TIMEOUT]) {
    // a step block of a fused pair launch gave up waiting for its atoms' force records (pair_fast_f32.hip): those atoms
    // were not integrated.  Repeat the batch with the separate integrator kernel; a second time-out of this context
    // switches the fused launch off for good.
    (void)hipMemsetAsync(rp.flags.as<int>() + F_STEP_TIMEOUT, 0, sizeof(int), st);
    ctx->fused_step_timeouts++;
    ctx->no_fused_once = true;
    if (ctx->fused_step_timeouts >= 2) ctx->fused_disabled = true;
    ctx->no_chain_skip_once = true;
    rp.seq_valid = false;
    rp.box[0] = -1;  // re-plan + rebuild
    last_error() = "a step block of the fused pair + step launch timed out waiting for a force record (workgroups not "
                   "dispatched in block order?); the batch is repeated with the separate integrator kernel";
    if (h[F_MAXN] <= rp.lg.maxn) return 1;
  }
  if (h[F_CELLCAP]) {
    // a cell received more atoms than the member array of the two-launch binning holds: the list built from it is
    // incomplete.  This replica bins with the four launches from now on; the work is repeated.
    (void)hipMemsetAsync(rp.flags.as<int>() + F_CELLCAP, 0, sizeof(int), st);
    rp.cell_cap_fallback = true;
    ctx->no_chain_skip_once = true;
    rp.seq_valid = false;
    rp.box[0] = -1;  // re-plan + rebuild
    last_error() = "a cell holds more atoms than the two-launch binning's member array (the replica switches to the "
                   "four-launch binning; results since the last check are invalid)";
    return 1;
  }
  if (h[F_VIOLATION]) {
    // an atom crossed its displacement limit in a step whose rebuild chain had been left out (ListCheck)
    (void)hipMemsetAsync(rp.flags.as<int>() + F_VIOLATION, 0, sizeof(int), st);
    ctx->no_chain_skip_once = true;
    rp.seq_valid = false;
    rp.spec_valid = false;
    rp.spec_backoff = 16;  // (plain evaluations: the next ones keep their chain)
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


template int alloc_replica<float>(tmdhip_ctx *, Replica &, int);
template int alloc_replica<double>(tmdhip_ctx *, Replica &, int);
template int compute_list<float>(tmdhip_ctx *, Replica &, const void *, const double *, void *, double *, int, hipStream_t,
                                 const FusedLaunchT<float> *, ListOnlyOut *);
template int compute_list<double>(tmdhip_ctx *, Replica &, const void *, const double *, void *, double *, int, hipStream_t,
                                  const FusedLaunchT<double> *, ListOnlyOut *);

}  // namespace tmd

using namespace tmd;

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
  // the uploads above went through the null stream; the caller's stream need not be ordered behind it (a stream created
  // non-blocking — every side stream of PyTorch — is not): everything is in place when the call returns
  (void)hipStreamSynchronize(nullptr);
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
  for (DevBuf *b : {&ctx->types, &ctx->qs, &ctx->tab, &ctx->excl_off, &ctx->excl_idx, &ctx->half_skin, &ctx->half_skin2, &ctx->escratch, &ctx->boxes, &ctx->pos_alt_all, &ctx->snap, &ctx->batch_tab, &ctx->chain_tab})
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
  ctx->ke_from_run = nullptr;
  ctx->nactive = nactive > 0 ? nactive : 0x7fffffff;
  ctx->open_bounds_valid = false;
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
    if (rp.lg.maxn > 0) rp.maxn_keep = rp.lg.maxn;
    rp.lg.maxn = 0;
  }
  TMD_HIP(hipStreamSynchronize(nullptr));  // (null-stream uploads complete before a non-blocking stream of the caller uses them)
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
  TMD_HIP(hipStreamSynchronize(nullptr));  // (null-stream uploads complete before a non-blocking stream of the caller uses them)
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
  out->fused_step_timeouts = ctx->fused_step_timeouts;
  out->final_steps_in_pair_launch = ctx->final_steps_in_pair_launch;
  out->batched_launches = ctx->batched_launches;
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
  TMD_TRY(ctx->sync_e.ensure(ebytes + sizeof(double)));  // (+ one word the fused evaluation's fold kernel needs for a sum nobody reads)
  if (!ctx->sync_host) {
    TMD_HIP(hipHostMalloc(&ctx->sync_host, seq_off + 64, hipHostMallocMapped));
    std::memset(ctx->sync_host, 0, seq_off + 64);
  }
  double *he = (double *)ctx->sync_host;
  int *hf = (int *)((char *)ctx->sync_host + ebytes);
  double *e = ctx->sync_e.as<double>();
  double *hk = (double *)((char *)ctx->sync_host + ebytes + fbytes);
  volatile unsigned *hseq = (volatile unsigned *)((char *)ctx->sync_host + seq_off);
  const bool lists = ctx->algorithm == TMDHIP_ALGO_CELLLIST;
  // one replica on the lean fp32 kernel with a light topology: pair + bonded in one launch, fold + report in a second (md_loop.hip)
  const int fused = compute_fused_eval(ctx, pos_dev, box_host, forces_dev, e, e + TMDHIP_NENERGY * nrep, he, hk, hf, hseq, st);
  if (fused < 0) return fused;
  if (fused == 1) {
    TMD_TRY(wait_observed(hseq, ctx->obs_seq, st));
  } else {
  TMD_HIP(hipMemsetAsync(e, 0, ebytes, st));
  int flags = TMDHIP_WANT_ENERGY;
  if (forces_dev) {
    flags |= TMDHIP_WANT_FORCES;
    if (ctx->d.terms == 0)  // no nonbonded kernel to store the forces: the bonded kernels add into zeros
      TMD_HIP(hipMemsetAsync(forces_dev, 0, (size_t)ctx->real_size * 3 * ctx->d.natoms * nrep, st));
  }
  // (kSpecChain: this call reads every replica's list flags back before it returns and reports "repeat" on F_VIOLATION)
  TMD_TRY(tmdhip_compute_nonbonded(ctx, TMDHIP_ALL_REPLICAS, pos_dev, box_host, forces_dev, e,
                                   flags | kSpecChain | (forces_dev ? TMDHIP_OVERWRITE_FORCES : 0), stream));
  TMD_TRY(tmdhip_compute_bonded(ctx, TMDHIP_ALL_REPLICAS, pos_dev, box_host, forces_dev, e, flags, stream));
  if (nrep <= 16) {  // results through host-mapped memory + a sequence word (md_loop.hip: observe_publish_kernel)
    TMD_TRY(publish_observables(ctx, e, nullptr, lists, he, hk, hf, hseq, st));  // the one host synchronisation of an energy evaluation
  } else {
    TMD_HIP(hipMemcpyAsync(he, e, ebytes, hipMemcpyDeviceToHost, st));
    if (lists)
      for (size_t r = 0; r < nrep; ++r)
        TMD_HIP(hipMemcpyAsync(hf + r * F_COUNT, ctx->rep[r].flags.p, sizeof(int) * F_COUNT, hipMemcpyDeviceToHost, st));
    TMD_HIP(hipStreamSynchronize(st));  // the one host synchronisation of an energy evaluation
  }
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
  ctx->timing_interior_only = ((unsigned)on >> 31) != 0;
  // the events of the first launches are created here, not inside the region being timed (a hipEventCreate
  // costs ~10 us of host time: twenty of them in a 20-step run made the loop enqueue-bound)
  while (ctx->timing && ctx->events.size() < 192) {
    hipEvent_t a, b;
    TMD_HIP(hipEventCreateWithFlags(&a, kTimingEventFlags));
    TMD_HIP(hipEventCreateWithFlags(&b, kTimingEventFlags));
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
