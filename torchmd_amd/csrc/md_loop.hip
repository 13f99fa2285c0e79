// The MD loop of the nonbonded engine for gfx950 (MI355X): fused MD-step kernels and tmdhip_md_run / _observe /
// _restore, which enqueue whole batches of steps from C.
//
// Reference semantics: torchmd/integrator.py:61-74 (_first_VV, _second_VV, langevin) in the order of
// Integrator.step (integrator.py:112-125): first_VV(old F) -> compute -> langevin -> second_VV(new F).
#include "engine.h"
#include "md_step.h"

namespace tmd {

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
  if constexpr (FIRST && !SECOND && CHECK) {
    if (s.snap_pos) {  // the state at the entry of the call (tmdhip_md_restore), from the registers just loaded
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        s.snap_pos[3 * i + k] = x.p[k];
        s.snap_vel[3 * i + k] = x.v[k];
        s.snap_f[3 * i + k] = x.f[k];
      }
      if (s.zero && i < s.nzero) s.zero[i] = 0.0;
    }
  }
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

// the launch of the report alone (<= 16 replicas); the caller waits for ctx->obs_seq (wait_observed) when it needs the values
static int launch_publish(tmdhip_ctx *ctx, const double *energies_dev, const double *ke_dev, bool lists, double *host_e,
                          double *host_ke, int *host_flags, volatile unsigned *host_seq, hipStream_t st) {
  const size_t nrep = ctx->rep.size();
  ObsFlagPtrs fp{};
  for (size_t r = 0; r < nrep; ++r) fp.p[r] = lists ? ctx->rep[r].flags.as<int>() : nullptr;
  if (++ctx->obs_seq == 0) ctx->obs_seq = 1;
  hipLaunchKernelGGL(observe_publish_kernel, dim3(1), dim3(128), 0, st, (int)nrep, energies_dev, ke_dev, fp, host_e, host_ke,
                     host_flags, const_cast<unsigned *>(host_seq), ctx->obs_seq);
  TMD_HIP(hipGetLastError());
  return 0;
}

int publish_observables(tmdhip_ctx *ctx, const double *energies_dev, const double *ke_dev, bool lists, double *host_e,
                        double *host_ke, int *host_flags, volatile unsigned *host_seq, hipStream_t st) {
  TMD_TRY(launch_publish(ctx, energies_dev, ke_dev, lists, host_e, host_ke, host_flags, host_seq, st));
  return wait_observed(host_seq, ctx->obs_seq, st);
}

// the host-mapped landing zone of tmdhip_md_observe: energies [R][NENERGY] | kinetic energies [R] | list flags [R][F_COUNT] | sequence word
struct ObsHost {
  double *e, *ke;
  int *flags;
  volatile unsigned *seq;
};
static int obs_host_zone(tmdhip_ctx *ctx, ObsHost &z) {
  const size_t nrep = ctx->rep.size();
  const size_t ebytes = sizeof(double) * TMDHIP_NENERGY * nrep, kbytes = sizeof(double) * nrep, fbytes = sizeof(int) * F_COUNT * nrep;
  if (!ctx->obs_host) {
    TMD_HIP(hipHostMalloc(&ctx->obs_host, ebytes + kbytes + fbytes + 64, hipHostMallocMapped));
    std::memset(ctx->obs_host, 0, ebytes + kbytes + fbytes + 64);
  }
  z.e = (double *)ctx->obs_host;
  z.ke = z.e + TMDHIP_NENERGY * nrep;
  z.flags = (int *)((char *)ctx->obs_host + ebytes + kbytes);
  z.seq = (volatile unsigned *)((char *)ctx->obs_host + ebytes + kbytes + fbytes + 32);
  return 0;
}

// The last kernel of a tmdhip_md_run call whose final step was made by FINAL step blocks (one replica): final_fold_kernel's sums
// AND observe_publish_kernel's report in one launch (round 6) — the energies of the call, the kinetic energy and the list flags go
// to the host-mapped zone with the sequence word behind them, so that a tmdhip_md_observe(TMDHIP_OBSERVE_AFTER_RUN) launches
// nothing and only waits for the word.
__global__ __launch_bounds__(kEnergySlots) void final_fold_publish_kernel(double *__restrict__ scratch, double *__restrict__ out,
                                                                          double *__restrict__ ke, const int *__restrict__ flags,
                                                                          double *host_e, double *host_ke, int *host_flags,
                                                                          unsigned *host_seq, unsigned seq, int accumulate) {
  __shared__ double part[kEnergySlots / 64][TMDHIP_NENERGY + 1];
  double *row = scratch + (size_t)threadIdx.x * kEnergyStride;
#pragma unroll
  for (int k = 0; k <= TMDHIP_NENERGY; ++k) {
    const double v = row[k];
    if (v != 0.0) row[k] = 0.0;
    const double s = wave_sum(v);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][k] = s;
  }
  __syncthreads();
  if (threadIdx.x <= TMDHIP_NENERGY) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kEnergySlots / 64; ++w) s += part[w][threadIdx.x];
    if (threadIdx.x < TMDHIP_NENERGY) {
      // (accumulate: the bonded kernel of a heavy topology has left its energies there already; a plain evaluation overwrites)
      const double e = accumulate ? out[threadIdx.x] + s : s;
      if (s != 0.0 || !accumulate) out[threadIdx.x] = e;
      host_e[threadIdx.x] = e;
    } else {
      ke[0] = s;
      host_ke[0] = s;
    }
  } else if (threadIdx.x >= 64 && threadIdx.x < 64 + F_COUNT) {
    host_flags[threadIdx.x - 64] = flags ? flags[threadIdx.x - 64] : 0;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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
template <typename R>
__global__ void fused_upload_kernel(FusedStaticT<R> v, FusedStaticT<R> *dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = v;
}

template <typename R>
int upload_fused_static(Replica &rp, const FusedStaticT<R> &now, hipStream_t st) {
  static_assert(sizeof(FusedStaticT<R>) <= sizeof(rp.fused_host), "Replica::fused_host holds either precision");
  TMD_TRY(rp.fused_dev.ensure(sizeof(FusedStaticT<double>)));
  if (!rp.fused_host_valid || std::memcmp(rp.fused_host, &now, sizeof(now)) != 0) {
    hipLaunchKernelGGL((fused_upload_kernel<R>), dim3(1), dim3(64), 0, st, now, rp.fused_dev.as<FusedStaticT<R>>());
    TMD_HIP(hipGetLastError());
    std::memcpy(rp.fused_host, &now, sizeof(now));
    rp.fused_host_valid = true;
  }
  return 0;
}
template int upload_fused_static<float>(Replica &, const FusedStaticT<float> &, hipStream_t);
template int upload_fused_static<double>(Replica &, const FusedStaticT<double> &, hipStream_t);

// can the pair launch of this replica integrate the next step itself?  (lean fp32 kernel, 4 .. 64 lanes per atom: a pair
// block's atoms fit one wave of a step block.  fp64: built in round 4, bit-identical and slower — 151 against 124.5 us
// per step at C3, no partial last round of pair blocks for the step blocks to hide in — and removed in round 5.)
template <typename R>
bool fused_step_possible(const tmdhip_ctx *ctx, const Replica &rp, const PairConsts<R> &c) {
  if (std::is_same<R, double>::value) return false;
  const char *e = std::getenv("TMDHIP_FUSED_STEP");  // (read per call: tests switch it within a process)
  if (e && std::atoi(e) == 0) return false;
  if (ctx->fused_off_call || ctx->fused_disabled) return false;  // repetition of a batch whose fused launch timed out
  const bool only_lj_el = c.terms != 0 && (c.terms & ~(TMDHIP_TERM_LJ | TMDHIP_TERM_ELECTROSTATICS)) == 0;
  return only_lj_el && ctx->d.ntypes <= kEntryTypes && rp.lg.lpa >= 4 && rp.lg.lpa <= 64 && kFastThreads / rp.lg.lpa <= 64;
}


// ---- a plain evaluation with energies in TWO launches behind the displacement test (tmdhip_compute, round 6) -------------------------
// Cell-list contexts in fp32 with one replica and a light topology (water, ions): the ENERGY variant of the lean pair launch with
// evaluation-only step blocks (FUSED = 5, md_step.h: FINAL = 2) that add the bonded force of their atoms to the pair force, store
// the sum in the caller's array and leave the bonded energies in the scratch rows; then ONE kernel folds the rows and reports
// energies and list flags to the caller's host-mapped zone.  The bonded kernel, its pass over the force array, the fold launch, the
// report launch and the clearing of the energy buffer go away (5 launches -> 3 with the displacement test).  Returns 1 when the
// evaluation was enqueued this way (the caller waits for ctx->obs_seq), 0 when the context does not qualify, < 0 on errors.
int compute_fused_eval(tmdhip_ctx *ctx, const void *pos_dev, const double *box, void *forces_dev, double *e_dev, double *scratch_ke,
                       double *host_e, double *host_ke, int *host_flags, volatile unsigned *host_seq, hipStream_t st) {
  const char *e_on = std::getenv("TMDHIP_FUSED_EVAL");  // (0: the separate kernels; A/B, tests)
  if (e_on && std::atoi(e_on) == 0) return 0;
  if (ctx->d.dtype != TMDHIP_F32 || ctx->algorithm != TMDHIP_ALGO_CELLLIST || ctx->rep.size() != 1 || !forces_dev || ctx->d.terms == 0 ||
      ctx->no_fused_once)
    return 0;
  Replica &rp = ctx->rep[0];
  const PairConsts<float> c = make_consts<float>(ctx, box);
  if (!rp.have_list || box[0] != rp.box[0] || box[1] != rp.box[1] || box[2] != rp.box[2] || !fused_step_possible<float>(ctx, rp, c)) return 0;
  BondedArgs<float> A;
  std::memset(&A, 0, sizeof(A));
  if (tmd::bonded_inline_args(ctx, box, A) != 1) return 0;
  FusedStaticT<float> now;
  std::memset(&now, 0, sizeof(now));
  now.s.n = ctx->d.natoms;
  now.s.chk.flags = rp.flags.as<int>();
  std::memcpy(&now.A, &A, sizeof(A));
  now.has_bonded = 1;
  now.nactive = ctx->nactive;
  TMD_TRY(upload_fused_static(rp, now, st));
  FusedLaunchT<float> fl{};
  fl.fst = rp.fused_dev.as<FusedStaticT<float>>();
  fl.langevin = false;
  fl.eval_only = true;
  fl.step.pos_in = (const float *)pos_dev;
  fl.step.bonded = 1;
  rp.n_compute++;
  const int rc = compute_list<float>(ctx, rp, pos_dev, box, forces_dev, e_dev,
                                     TMDHIP_WANT_FORCES | TMDHIP_WANT_ENERGY | TMDHIP_OVERWRITE_FORCES | kSpecChain, st, &fl);
  if (rc != 0) return rc < 0 ? rc : fail("tmdhip_compute: the fused evaluation could not be enqueued");
  if (++ctx->obs_seq == 0) ctx->obs_seq = 1;
  hipLaunchKernelGGL(final_fold_publish_kernel, dim3(1), dim3(kEnergySlots), 0, st, ctx->escratch.as<double>(), e_dev, scratch_ke,
                     rp.flags.as<int>(), host_e, host_ke, host_flags, const_cast<unsigned *>(host_seq), ctx->obs_seq, 0);
  TMD_HIP(hipGetLastError());
  return 1;
}

// ---- the replicas of a cell-list context in one pair + step launch (pair_fast_kernel.h: list_pair_fast_f32_batch_kernel) ----
__global__ void batch_upload_kernel(BatchRep v, BatchRep *dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = v;
}

// what md_run collects per replica while it walks the replicas of an iteration that ends in ONE batched launch
struct BatchItem {
  FusedLaunchT<float> fl;
  ListOnlyOut lo;
  float *pos;       // positions of this launch's forces (cur[r])
  float *home, *f;  // the caller's position / force rows of the replica
  unsigned *pub_ptr;
  unsigned pub_val;
  double box[3];
};

// TMDHIP_BATCH_REPLICAS=0: the replica-by-replica loop (A/B, tests; read per call); =2: the batched launch for a single replica
// too (A/B of the batched kernel against the plain one on the same workload)
static bool batch_replicas_on() {
  const char *e = std::getenv("TMDHIP_BATCH_REPLICAS");
  return !(e && std::atoi(e) == 0);
}
static int batch_replicas_min() {
  const char *e = std::getenv("TMDHIP_BATCH_REPLICAS");
  return e && std::atoi(e) == 2 ? 1 : 2;
}

// One launch for the replicas of `items` (all of the context's): table entries that changed are re-uploaded (stream-ordered
// one-thread kernels), the per-launch words travel as the kernel argument.  `energy`: the call's last step (FINAL step blocks).
static int launch_replica_batch(tmdhip_ctx *ctx, std::vector<BatchItem> &items, int bonded, uint64_t noise_step, bool energy,
                                bool langevin, hipStream_t st) {
  const int nrep = (int)items.size(), n = ctx->d.natoms;
  TMD_TRY(ctx->batch_tab.ensure(sizeof(BatchRep) * (size_t)nrep));
  if ((int)ctx->batch_host.size() != nrep) {
    ctx->batch_host.assign(nrep, BatchRep{});
    for (auto &e : ctx->batch_host) std::memset(&e, 0xFF, sizeof(e));  // (matches nothing: every entry is uploaded)
  }
  int pair_blocks = 0, step_blocks = 0;
  fused_grid_shape(ctx, ctx->rep[0], bonded, pair_blocks, step_blocks);
  for (int g0 = 0; g0 < nrep; g0 += kBatchMax) {
    const int gn = std::min(kBatchMax, nrep - g0);
    BatchLaunch bl;
    std::memset(&bl, 0, sizeof(bl));
    bl.nrep = gn;
    bl.pair_blocks = pair_blocks;
    bl.step_blocks = step_blocks;
    bl.bonded = bonded;
    bl.poll_limit = 1u << 22;  // ~4 s of polling
    bl.noise_step = noise_step;
    PairConsts<float> c0 = make_consts<float>(ctx, items[g0].box);
    for (int k = 0; k < gn; ++k) {
      const int r = g0 + k;
      Replica &rp = ctx->rep[r];
      BatchItem &it = items[r];
      if (rp.fsort.bytes < sizeof(float4) * (size_t)n) {
        TMD_TRY(rp.fsort.ensure(sizeof(float4) * (size_t)n));
        TMD_HIP(hipMemsetAsync(rp.fsort.p, 0, rp.fsort.bytes, st));  // launch number 0 = never written
        rp.fused_gen = 0;
      }
      if (++rp.fused_gen == 0) rp.fused_gen = 1;
      rp.fused_launches++;
      const PairConsts<float> c = make_consts<float>(ctx, it.box);
      BatchRep e;
      std::memset(&e, 0, sizeof(e));
      float4 *sa = rp.sorted.as<float4>(), *sb = rp.sorted_alt.as<float4>();
      e.sorted[0] = std::min(sa, sb);
      e.sorted[1] = std::max(sa, sb);
      e.pos[0] = it.home;
      e.pos[1] = rp.pos_alt.as<float>();
      e.stype = rp.stype.as<int>();
      e.order = rp.order.as<int>();
      e.nlist = rp.nlist.as<unsigned>();
      e.nneigh = rp.nneigh.as<int>();
      e.forces = it.f;
      e.escratch = ctx->escratch.as<double>() + (size_t)r * kEnergySlots * kEnergyStride;
      e.ext = rp.extent.as<int>();
      e.lflags = rp.flags.as<int>();
      e.padgen = rp.padgen.as<int>();
      e.fst = it.fl.fst;
      e.fsort = rp.fsort.as<float4>();
      e.hostpub = rp.hostpub;
      for (int x = 0; x < 3; ++x) {
        e.box[x] = c.box[x];
        e.invbox[x] = c.invbox[x];
      }
      e.maxn = rp.lg.maxn;
      if (std::memcmp(&e, &ctx->batch_host[r], sizeof(e)) != 0) {
        hipLaunchKernelGGL(batch_upload_kernel, dim3(1), dim3(64), 0, st, e, ctx->batch_tab.as<BatchRep>() + r);
        TMD_HIP(hipGetLastError());
        ctx->batch_host[r] = e;
      }
      if (it.pos != e.pos[0] && it.pos != e.pos[1]) return fail("batched pair + step launch: positions in neither buffer of the replica");
      unsigned bits = (unsigned)it.lo.lmode & kBlLmodeMask;
      if (sa == e.sorted[1]) bits |= kBlSortedCur;
      if (it.pos == e.pos[1]) bits |= kBlPosCur;
      if (it.lo.next_parity) bits |= kBlNextParity;
      if (it.fl.step.near_host) bits |= kBlReports;
      if (it.pub_ptr) bits |= kBlPublish;
      bl.bits[k] = bits;
      bl.gen[k] = rp.fused_gen;
      bl.seq[k] = it.fl.step.seq;
      bl.pub[k] = it.pub_val;
    }
    TMD_TRY(launch_pair_fast_f32_batch(ctx, g0, c0, bl, ctx->rep[g0].lg.lpa, energy, langevin, st));
  }
  return 0;
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
  ctx->fused_off_call = ctx->no_fused_once;
  ctx->no_fused_once = false;
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
  std::vector<char> finalized(nrep, 0);  // the last pair launch made the call's final kick, bonded force and energies itself (FINAL step blocks)
  ctx->ke_from_run = nullptr;
  ctx->run_published_seq = 0;
  const char *e_final = std::getenv("TMDHIP_FUSED_FINAL");  // (A/B, tests: 0 = the separate kernels behind the last pair launch)
  const bool final_on = !(e_final && std::atoi(e_final) == 0);
  for (int r = 0; r < nrep; ++r) cur[r] = (R *)d->pos_dev + r * stride;
  // the same for the replica-batched all-pairs mode (all replicas move together)
  R *const home_all = (R *)d->pos_dev;
  R *bcur = home_all;
  bool bowed = false;

  if (ctx->snap_pending) {
    // the first kernel of the call takes the snapshot only when it is the plain first half step of a list replica with a valid
    // list (the common case); otherwise the copy kernel runs after all
    const Replica &rp0 = ctx->rep[0];
    const double *box0 = d->box_host;
    const bool takes = nrep == 1 && ctx->algorithm == TMDHIP_ALGO_CELLLIST && ctx->d.terms != 0 && rp0.have_list &&
                       box0[0] == rp0.box[0] && box0[1] == rp0.box[1] && box0[2] == rp0.box[2];
    if (!takes) {
      const size_t bytes = sizeof(R) * stride * nrep, n4 = bytes / 16;
      hipLaunchKernelGGL(snapshot3_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, n4, (const uint4 *)d->pos_dev,
                         (const uint4 *)d->vel_dev, (const uint4 *)d->forces_dev, ctx->snap.as<uint4>(), ctx->snap_zero, ctx->snap_nzero);
      TMD_HIP(hipGetLastError());
      ctx->snap_pending = false;
    }
  }
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
    // The replicas of a cell-list context in ONE pair + step launch (round 6): when every replica of this iteration would make a
    // fused launch of its own — same conditions as below — the loop over the replicas does the per-replica part (pacing, the
    // first half step of a call, the rebuild chain) and the launch follows behind it.
    std::vector<BatchItem> batch_items;
    bool batching = false;
    int batch_bonded = 0;
    if constexpr (std::is_same<R, float>::value) {
      const bool want_e = it == d->niter - 1 && d->energies_dev;
      const bool interior = it + 1 < d->niter && !want_e, final_step = it + 1 == d->niter && want_e && final_on;
      batching = first && nrep >= batch_replicas_min() && ctx->algorithm == TMDHIP_ALGO_CELLLIST && ctx->d.terms != 0 && (interior || final_step) &&
                 batch_replicas_on();
      for (int r = 0; batching && r < nrep; ++r) {
        const Replica &rp = ctx->rep[r];
        const double *box = d->box_host + 3 * r;
        const PairConsts<R> c = make_consts<R>(ctx, box);
        BondedArgs<R> A;
        std::memset(&A, 0, sizeof(A));
        batching = rp.have_list && box[0] == rp.box[0] && box[1] == rp.box[1] && box[2] == rp.box[2] && !owed[r] && !finalized[r] &&
                   fused_step_possible<R>(ctx, rp, c) && rp.lg.lpa == ctx->rep[0].lg.lpa;
        if (batching) {
          const int bm = tmd::bonded_inline_args(ctx, box, A);
          batching = bm >= 0 && (r == 0 || bm == batch_bonded);
          batch_bonded = bm;
        }
      }
      if (batching) batch_items.resize(nrep);
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
        // (the final kick of a call — it == niter, no drift, no test — leaves the report of the call's last step valid:
        // that is what the next call's first step looks at when the caller vouches for a continuation.  Round 4: this
        // branch used to clear it for every iteration without pacing, which made the hint a no-op.)
        if (first) rp.seq_valid = false;
        rp.pub_ptr = nullptr;
      }
      a.sorted = rp.sorted.as<R4>();
      a.inv = rp.inv.as<int>();
      a.pos_in = a.pos_out = cur[r];
      BondedArgs<R> A;
      std::memset(&A, 0, sizeof(A));
      const bool was_stepped = stepped[r] != 0;
      stepped[r] = 0;
      if (finalized[r]) {
        // (it == niter: the final kick was made by the step blocks of the last pair launch)
      } else if (was_stepped) {
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
        if (ctx->snap_pending && check && nrep == 1) {  // (tmdhip_md_run left the snapshot of the entry state to this kernel)
          const size_t padded = (sizeof(R) * stride + 15) / 16 * 16;
          a.snap_pos = ctx->snap.as<R>();
          a.snap_vel = (R *)(ctx->snap.as<char>() + padded);
          a.snap_f = (R *)(ctx->snap.as<char>() + 2 * padded);
          a.zero = ctx->snap_zero;
          a.nzero = ctx->snap_nzero;
          ctx->snap_pending = false;
        }
        launch_md_step<R, false, false, true>(a, c, check, st);
        a.snap_pos = a.snap_vel = a.snap_f = nullptr;
        a.zero = nullptr;
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
          // interior step on a lean kernel: the pair launch makes the next step itself (FusedStepT)
          FusedLaunchT<R> fl{};
          bool fuse = false;
          {
            // interior steps: the launch makes the next step; the last step of a call that wants energies (one replica):
            // the launch makes the final kick, the bonded force + energies and the kinetic energy (FINAL step blocks)
            const bool interior = it + 1 < d->niter && !en;
            const bool final_step = it + 1 == d->niter && en && (nrep == 1 || batching) && final_on && std::is_same<R, float>::value;
            const int bm = (check && (interior || final_step) && fused_step_possible<R>(ctx, rp, c))
                               ? tmd::bonded_inline_args(ctx, box, A) : -1;
            if (bm >= 0) {
              if (bm == 2) {
                // heavy topology: the bonded force depends on the positions only — it is evaluated in front of the
                // pair launch into a buffer of its own and the step blocks add it (same values, same order as the
                // separate kernels: pair force stored, bonded force added, divided by the mass)
                // (the final step: with its energies, which the bonded kernel folds into the call's buffer itself)
                TMD_TRY(rp.fbond.ensure(sizeof(R) * stride));
                TMD_TRY(tmdhip_compute_bonded(ctx, r, pos, box, rp.fbond.p, en,
                                              TMDHIP_WANT_FORCES | TMDHIP_OVERWRITE_FORCES | (en ? TMDHIP_WANT_ENERGY : 0), st));
              }
              FusedStaticT<R> now;
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
              now.fbond = bm == 2 ? rp.fbond.as<R>() : nullptr;
              now.nactive = 0x7fffffff;
              TMD_TRY(rp.pos_alt.ensure(sizeof(R) * stride));
              TMD_TRY(upload_fused_static(rp, now, st));
              fl.fst = rp.fused_dev.as<FusedStaticT<R>>();
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
          if constexpr (std::is_same<R, float>::value) {
            if (batching) {  // list bookkeeping of this replica now, its blocks in the launch behind the loop
              if (!fuse) return fail("tmdhip_md_run: a replica of the batch cannot make a fused launch");
              BatchItem &bi = batch_items[r];
              bi.fl = fl;
              bi.pos = pos;
              bi.home = home;
              bi.f = f;
              for (int k = 0; k < 3; ++k) bi.box[k] = box[k];
              bi.lo.chain = 0;
              TMD_TRY(compute_list<R>(ctx, rp, pos, box, f, en,
                                      flags_c | TMDHIP_OVERWRITE_FORCES | kListOnly | kDeferChain | (check ? kPrechecked : 0) | (skip_chain ? kSkipChain : 0) |
                                          (skip_chain && was_stepped ? kViolationCheck : 0),
                                      st, &fl, &bi.lo));
              bi.pub_ptr = rp.pub_ptr;
              bi.pub_val = rp.pub_val;
              rp.pub_ptr = nullptr;
              continue;
            }
          }
          const int rc = compute_list<R>(ctx, rp, pos, box, f, en,
                                         flags_c | TMDHIP_OVERWRITE_FORCES | (check ? kPrechecked : 0) |
                                             (skip_chain ? kSkipChain : 0) |
                                             (skip_chain && was_stepped ? kViolationCheck : 0) |
                                             (en && !fuse && rp.have_list && tmd::bonded_inline_args(ctx, box, A) != 0 ? kDeferFold : 0),
                                         st, fuse ? &fl : nullptr);
          rp.pub_ptr = nullptr;
          if (fuse && rc == 0 && en) {
            // the call's last step: forces (pair + bonded) are in `forces`, velocities kicked, the energy rows (pair,
            // bonded, kinetic) folded here — into the call's energy buffer and the context's kinetic-energy word
            TMD_TRY(ctx->obs_ke.ensure(sizeof(double) * ctx->rep.size()));
            if (nrep == 1) {  // ... and reported to the host in the same launch (tmdhip_md_observe then only waits for the word)
              ObsHost z;
              TMD_TRY(obs_host_zone(ctx, z));
              if (++ctx->obs_seq == 0) ctx->obs_seq = 1;
              hipLaunchKernelGGL(final_fold_publish_kernel, dim3(1), dim3(kEnergySlots), 0, st, ctx->escratch.as<double>(), en,
                                 ctx->obs_ke.as<double>(), rp.flags.as<int>(), z.e, z.ke, z.flags, const_cast<unsigned *>(z.seq),
                                 ctx->obs_seq, 1);
              ctx->run_published_seq = ctx->obs_seq;
              ctx->run_published_energies = en;
            } else {
              hipLaunchKernelGGL(final_fold_kernel, dim3(1), dim3(kEnergySlots), 0, st, ctx->escratch.as<double>(), en,
                                 ctx->obs_ke.as<double>());
            }
            TMD_HIP(hipGetLastError());
            finalized[r] = 1;
            ctx->ke_from_run = d->vel_dev;
            ctx->ke_from_run_mass = d->mass_dev;
            ctx->final_steps_in_pair_launch++;
            continue;
          }
          if (fuse && rc == 0) {
            cur[r] = fl.step.pos_out;
            std::swap(rp.sorted, rp.sorted_alt);
            stepped[r] = 1;
            rp.steps_in_pair_launch++;
            continue;  // forces of this step never reach `forces`
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
    if constexpr (std::is_same<R, float>::value) {
      if (batching) {
        const bool want_e = it == d->niter - 1 && d->energies_dev;
        {  // the rebuild chains the host has not left out: one launch per kernel for all of them
          std::vector<int> reps, par;
          std::vector<const float *> ps;
          std::vector<const double *> bx;
          // (TMDHIP_REPLICA_REBUILDS=together, list_build.hip: chain_any — every replica is in the launch as soon as one is)
          const char *e_tog = std::getenv("TMDHIP_REPLICA_REBUILDS");
          bool any_chain = false;
          for (int r = 0; r < nrep; ++r) any_chain = any_chain || batch_items[r].lo.chain;
          const bool everybody = any_chain && e_tog && std::strcmp(e_tog, "together") == 0;
          for (int r = 0; r < nrep; ++r)
            if (batch_items[r].lo.chain || everybody) {
              reps.push_back(r);
              // (a replica whose chain the host had left out: compute_list has counted the step already)
              par.push_back(batch_items[r].lo.chain ? batch_items[r].lo.chain_parity : (int)((ctx->rep[r].step - 1) & 1));
              ps.push_back(batch_items[r].pos);
              bx.push_back(batch_items[r].box);
            }
          if (!reps.empty()) TMD_TRY(enqueue_chain_batch<float>(ctx, (int)reps.size(), reps.data(), ps.data(), par.data(), bx.data(), st));
        }
        TMD_TRY(launch_replica_batch(ctx, batch_items, batch_bonded, d->step0 + (uint64_t)it, want_e, langevin, st));
        if (want_e) {
          // the call's last step: forces (pair + bonded) in `forces`, velocities kicked; one fold block per replica adds its
          // scratch rows (pair, bonded, kinetic) into the call's energy buffer and the context's kinetic-energy words
          TMD_TRY(ctx->obs_ke.ensure(sizeof(double) * ctx->rep.size()));
          hipLaunchKernelGGL(final_fold_kernel, dim3(nrep), dim3(kEnergySlots), 0, st, ctx->escratch.as<double>(), d->energies_dev,
                             ctx->obs_ke.as<double>());
          TMD_HIP(hipGetLastError());
          for (int r = 0; r < nrep; ++r) finalized[r] = 1;
          ctx->ke_from_run = d->vel_dev;
          ctx->ke_from_run_mass = d->mass_dev;
          ctx->final_steps_in_pair_launch++;
        } else {
          for (int r = 0; r < nrep; ++r) {
            Replica &rp = ctx->rep[r];
            cur[r] = batch_items[r].fl.step.pos_out;
            std::swap(rp.sorted, rp.sorted_alt);
            stepped[r] = 1;
            rp.steps_in_pair_launch++;
          }
        }
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

template bool fused_step_possible<float>(const tmdhip_ctx *, const Replica &, const PairConsts<float> &);
template bool fused_step_possible<double>(const tmdhip_ctx *, const Replica &, const PairConsts<double> &);
template int md_run<float>(tmdhip_ctx *, const tmdhip_md_desc *, hipStream_t);
template int md_run<double>(tmdhip_ctx *, const tmdhip_md_desc *, hipStream_t);

}  // namespace tmd

using namespace tmd;

extern "C" {

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
      // (md_run's first kernel saves the state itself where it can — one launch less per call —, else runs snapshot3_kernel)
      ctx->snap_pending = true;
      ctx->snap_zero = fits ? desc->energies_dev : nullptr;
      ctx->snap_nzero = nzero;
      zeroed = zeroed || fits;
    } else {
      TMD_HIP(hipMemcpyAsync(sn, desc->pos_dev, bytes, hipMemcpyDeviceToDevice, st));
      TMD_HIP(hipMemcpyAsync(sn + padded, desc->vel_dev, bytes, hipMemcpyDeviceToDevice, st));
      TMD_HIP(hipMemcpyAsync(sn + 2 * padded, desc->forces_dev, bytes, hipMemcpyDeviceToDevice, st));
    }
    ctx->snap_bytes = bytes;
  }
  if (!zeroed) TMD_HIP(hipMemsetAsync(desc->energies_dev, 0, sizeof(double) * nzero, st));
  for (auto &rp : ctx->rep) rp.spec_valid = false;  // (a plain evaluation's report says nothing about positions this call moves)
  const int rc = ctx->d.dtype == TMDHIP_F32 ? md_run<float>(ctx, desc, st) : md_run<double>(ctx, desc, st);
  for (auto &rp : ctx->rep) rp.skin_vel = nullptr;  // rebuilds outside an MD run know no velocities: static skins
  if (rc == 0 && ctx->snap_pending) {
    ctx->snap_pending = false;
    return fail("tmdhip_md_run: the state at entry was not saved (internal error)");
  }
  ctx->snap_pending = false;
  // A call that returns energies is followed by tmdhip_md_observe (what Integrator.step does).  Its two launches — kinetic energy,
  // report to the host — are enqueued HERE, behind the run's last kernel with no host round trip between them (small systems: the
  // device idled ~20 us per call between the two C calls); tmdhip_md_observe(TMDHIP_OBSERVE_AFTER_RUN) then only waits for the
  // sequence word.  (One replica on the lean fp32 kernel: the FINAL launch's fold kernel has reported already.)
  const char *e_rep = std::getenv("TMDHIP_RUN_REPORTS");  // (0: tmdhip_md_observe launches them itself; A/B)
  if (rc == 0 && desc->energies_dev && ctx->run_published_seq == 0 && ctx->rep.size() <= 16 && !(e_rep && std::atoi(e_rep) == 0)) {
    ObsHost z;
    TMD_TRY(obs_host_zone(ctx, z));
    TMD_TRY(ctx->obs_ke.ensure(sizeof(double) * ctx->rep.size()));
    if (ctx->ke_from_run != desc->vel_dev || ctx->ke_from_run_mass != desc->mass_dev) {
      TMD_TRY(tmdhip_kinetic_energy(ctx->d.dtype, (int64_t)ctx->rep.size(), ctx->d.natoms, desc->vel_dev, desc->mass_dev,
                                    ctx->obs_ke.as<double>(), stream));
      ctx->ke_from_run = desc->vel_dev;
      ctx->ke_from_run_mass = desc->mass_dev;
    }
    TMD_TRY(launch_publish(ctx, desc->energies_dev, ctx->obs_ke.as<double>(), ctx->algorithm == TMDHIP_ALGO_CELLLIST, z.e, z.ke, z.flags,
                           z.seq, st));
    ctx->run_published_seq = ctx->obs_seq;
    ctx->run_published_energies = desc->energies_dev;
  }
  return rc;
}

int tmdhip_md_observe(tmdhip_ctx *ctx, const void *vel_dev, const void *mass_dev, const double *energies_dev,
                      double *out_host, int flags, void *stream) {
  if (!ctx || !vel_dev || !mass_dev || !out_host) return fail("tmdhip_md_observe: null argument");
  hipStream_t st = (hipStream_t)stream;
  const size_t nrep = ctx->rep.size();
  const size_t ebytes = sizeof(double) * TMDHIP_NENERGY * nrep, kbytes = sizeof(double) * nrep;
  TMD_TRY(ctx->obs_ke.ensure(kbytes));
  ObsHost z;
  TMD_TRY(obs_host_zone(ctx, z));
  double *he = z.e, *hk = z.ke;
  int *hf = z.flags;
  volatile unsigned *hseq = z.seq;
  const bool after_run = (flags & TMDHIP_OBSERVE_AFTER_RUN) && ctx->ke_from_run == vel_dev && ctx->ke_from_run_mass == mass_dev;
  const bool published = after_run && ctx->run_published_seq != 0 && ctx->run_published_energies == energies_dev;
  const unsigned published_seq = ctx->run_published_seq;
  ctx->run_published_seq = 0;
  if (after_run) {
    // (the FINAL step blocks of the run that just ended have summed the kinetic energy of these velocities — every replica's —
    // and the caller vouches that nothing has written them since)
  } else {
    TMD_TRY(tmdhip_kinetic_energy(ctx->d.dtype, (int64_t)nrep, ctx->d.natoms, vel_dev, mass_dev, ctx->obs_ke.as<double>(), stream));
  }
  ctx->ke_from_run = nullptr;
  const bool lists = ctx->algorithm == TMDHIP_ALGO_CELLLIST;
  if (published) {  // the run's last kernel has reported already: nothing to launch
    TMD_TRY(wait_observed(hseq, published_seq, st));
  } else if (nrep <= 16) {
    TMD_TRY(publish_observables(ctx, energies_dev, ctx->obs_ke.as<double>(), lists, he, hk, hf, hseq, st));
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
  ctx->ke_from_run = nullptr;                // (the velocities are no longer those of the run that ended)
  return 0;
}

}  // extern "C"
