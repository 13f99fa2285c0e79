// The per-atom update of an MD step (integrator.py:61-74 across the step boundary) as device functions shared by the
// fused MD-step kernels (md_loop.hip) and the step blocks of the lean fp32 pair launch (pair_fast_f32.hip): the same
// operations on the same registers in the same order wherever they run, so trajectories are bit-identical.
#pragma once

#include "engine.h"

namespace tmd {

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
                                             const R *noise = nullptr,  // noise: normal3 of this atom, drawn earlier
                                             R *vout = nullptr) {       // the updated velocity (final step blocks: kinetic energy)
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
  if (vout) {
#pragma unroll
    for (int k = 0; k < 3; ++k) vout[k] = v[k];
  }
}

// ---- the MD step inside a pair launch: step blocks (FusedStepT / FusedStaticT in engine.h) -----------------------------
constexpr int kStepPollSleep = 4;  // s_sleep argument between two polls of a force record (x 64 cycles)

// The force record a pair wave leaves for the step block that integrates its atom: {fx, fy, fz, launch number} as ONE
// 16-byte store written through to device scope (sc1) — the number in .w says the force beside it is this launch's.
// (fp32 only.  The fp64 form — 32 bytes as two ordered 16-byte stores — was measured slower than the separate integrator
// kernel in round 4 and removed in round 5: docs/history/round4.md.)
__device__ __forceinline__ void store_force_record(float4 *fsort, int n, int a, float sx, float sy, float sz, unsigned gen) {
  const __amdgpu_buffer_rsrc_t frsrc = __builtin_amdgcn_make_buffer_rsrc(fsort, 0, n * 16, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128((v4u){__float_as_uint(sx), __float_as_uint(sy), __float_as_uint(sz), gen}, frsrc, a * 16, 0,
                                         kAuxDeviceScope);
}
// one look at atom slot a's record: true (and the force in f) when it carries launch number `want`
__device__ __forceinline__ bool load_force_record(const __amdgpu_buffer_rsrc_t &frsrc, float, int a, unsigned want, float (&f)[3]) {
  const v4u r = __builtin_amdgcn_raw_buffer_load_b128(frsrc, a * 16, 0, kAuxDeviceScope | kAuxVolatile);
  if (r.w != want) return false;
  f[0] = __uint_as_float(r.x), f[1] = __uint_as_float(r.y), f[2] = __uint_as_float(r.z);
  return true;
}

// Step block j of a FUSED pair launch (four waves, 64 atoms): the atoms of the 64 / APB pair blocks that run on the
// same XCD (block ids congruent mod 8) and are neighbours in the cell-sorted order.  Like md_step_bonded_kernel, wave w
// evaluates bonded record slots w, w + 4, ... of all 64 atoms (lane = atom), the partial forces meet in LDS as
// (p0 + p1) + (p2 + p3), and the first wave updates — after it has waited for the pair waves of its atoms.
// s_lds: room for kQuad x 3 x 64 values of R (the pair role's LJ table space).
// FINAL (round 5): the step blocks of the LAST pair launch of a tmdhip_md_run call, which also wants the energies: no
// drift follows, so the update is the second half kick (+ thermostat) only; the blocks also leave the complete force
// (pair + bonded) in the caller's force array, the bonded energies of their atoms' records and the kinetic energy after
// the kick in the energy scratch rows (slot kKineticSlot) — the work of the bonded kernel, the final-kick kernel and
// the kinetic-energy kernel that used to follow the last pair launch of every call.  Same device functions in the
// same order: velocities and forces are bit-identical to the separate kernels; the energies are sums of the same
// terms in another order (fp64 atomics).
// FINAL = 2 (round 6): the step blocks of a PLAIN evaluation with energies (tmdhip_compute on a cell-list context with a light
// topology): no velocities, no kick — the blocks evaluate the bonded records of their atoms, wait for the pair force, leave pair +
// bonded force in the caller's array and the bonded energies in the scratch rows: the bonded kernel's launch, and its pass over
// the force array, go away.
constexpr int kKineticSlot = TMDHIP_NENERGY;  // (rows of kEnergyStride = 16 doubles: 8 per-term energies, then this)
// A read-only struct every lane reads at the same address, as scalar loads (constant address space).  For the plain kernel's
// `__restrict__` kernel argument the compiler finds that by itself; the batched launch takes the pointer from its replica table,
// and a pointer that was loaded from memory carries no such promise: the fields arrived by flat loads in VGPRs and the kernel
// needed 124 registers (4 waves per SIMD) where the plain one needs 96 (5 waves).
// (TABLE = false: the plain load — the plain kernels' code stays as it was measured.)
template <bool TABLE, typename T>
__device__ __forceinline__ T load_uniform(const T *p) {
  if (!TABLE) return *p;
  T v;
  __builtin_memcpy(&v, (const __attribute__((address_space(4))) T *)p, sizeof(T));
  return v;
}

template <typename R, bool LANGEVIN, int APB, int FINAL = 0, bool TABLE = false>
__device__ __forceinline__ void fused_step_blocks(const FusedStaticT<R> *__restrict__ fst, const FusedStepT<R> &fs,
                                                  const PairConsts<R> &c, int n, const typename Vec<R>::T4 *__restrict__ sorted,
                                                  const int *__restrict__ order, int j, int npair, R *s_lds,
                                                  R *__restrict__ forces_out = nullptr, double *__restrict__ escratch = nullptr) {
  using R4 = typename Vec<R>::T4;
  constexpr int K = 64 / APB;  // pair blocks per 64 atoms
  R(*s_part)[3][64] = reinterpret_cast<R(*)[3][64]>(s_lds);  // [kQuad][3][64]
  const int w = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  // With bonded records a step block is 64 atoms (its four waves share their records); without, every wave is a unit
  // of 64 atoms of its own (four waves of which three only met at the barrier doubled the waves of a 10^6-atom LJ launch).
  const bool bonded = fs.bonded == 1;  // (launch-uniform; 2 = the bonded force comes from a buffer: waves are units too)
  const int xcd = j & 7, q = bonded ? (j >> 3) : (j >> 3) * kQuad + w, g8 = npair >> 3;
  const int kc = K * q + lane / APB;  // this lane's pair block within the XCD's eighth
  const int a = (xcd * g8 + kc) * APB + lane % APB;
  const bool exists = kc < g8 && a < n;
  const int o = exists ? order[a] : 0;
  MdStepArgs<R> s = load_uniform<TABLE>(&fst->s);
  s.pos_in = fs.pos_in;
  s.pos_out = fs.pos_out;
  s.sorted = fs.sorted_out;
  s.noise_step = fs.noise_step;
  s.f_zero = nullptr;
  s.chk.near_host = fs.near_host;
  s.chk.seq = fs.seq;
  s.chk.parity = fs.parity;
  s.chk.skipped = 0;  // (unknown here: the next launch's first thread looks, kLmViolation)
  // (a brick of a domain decomposition integrates the atoms it owns: the halo rows behind them are passive)
  const bool integrates = (w == 0 || !bonded) && exists && o < load_uniform<TABLE>(&fst->nactive);
  AtomIn<R> x{};
  if (integrates && FINAL != 2) {  // every load of the update but the force, in flight during the bonded part
    x.m = s.mass[o];
    x.vc = LANGEVIN ? s.vcoeff[o] : R(0);
    if (!FINAL) {
      const R4 p = sorted[a];  // x, y, z, scaled charge: exactly what the position buffer holds
      x.p[0] = p.x, x.p[1] = p.y, x.p[2] = p.z;
      x.q = p.w;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      x.v[k] = s.vel[3 * o + k];
      if (!FINAL) x.r[k] = s.chk.ref[3 * o + k];
    }
    if (!FINAL) x.h2 = list_check_limit(s.chk, o);
    x.slot = a;
  }
  R fb[3] = {0, 0, 0};
  R g[3] = {0, 0, 0};
  if (bonded) {
    R fx = 0, fy = 0, fz = 0;
    double ef[TMDHIP_NENERGY] = {0, 0, 0, 0, 0, 0, 0, 0};  // (dead on interior steps: only FINAL reads them)
    if (exists) {
      const BondedArgs<R> A = load_uniform<TABLE>(&fst->A);
      const AtomRec<R> *rec = A.arec + (size_t)o * A.arec_stride;
      for (int k = w; k < A.arec_stride; k += kQuad) {
        const AtomRec<R> r = rec[k];
        if (r.ent == kNoRec) break;  // records are packed from the front
        eval_rec<R>(A, s.pos_in, o, r, fx, fy, fz, ef);
      }
    }
    s_part[w][0][lane] = fx;
    s_part[w][1][lane] = fy;
    s_part[w][2][lane] = fz;
    if (LANGEVIN && integrates) normal3<R>(s.seed, s.noise_step, s.row0 + (uint64_t)o, g[0], g[1], g[2]);
    __syncthreads();
    if (FINAL && escratch) {  // the bonded energies of this wave's records (all four waves evaluate records; every lane takes part)
      double *row = energy_row(escratch);
      wave_energy(exists ? ef[TMDHIP_E_BONDS] : 0.0, row + TMDHIP_E_BONDS);
      wave_energy(exists ? ef[TMDHIP_E_ANGLES] : 0.0, row + TMDHIP_E_ANGLES);
      wave_energy(exists ? ef[TMDHIP_E_DIHEDRALS] : 0.0, row + TMDHIP_E_DIHEDRALS);
      wave_energy(exists ? ef[TMDHIP_E_IMPROPERS] : 0.0, row + TMDHIP_E_IMPROPERS);
      wave_energy(exists ? ef[TMDHIP_E_LJ] : 0.0, row + TMDHIP_E_LJ);
      wave_energy(exists ? ef[TMDHIP_E_ELECTROSTATICS] : 0.0, row + TMDHIP_E_ELECTROSTATICS);
    }
    if (w != 0) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) fb[k] = (s_part[0][k][lane] + s_part[1][k][lane]) + (s_part[2][k][lane] + s_part[3][k][lane]);
  } else {
    if (fs.bonded == 2 && integrates) {
      const R *fbond = load_uniform<TABLE>(&fst->fbond);
#pragma unroll
      for (int k = 0; k < 3; ++k) fb[k] = fbond[3 * o + k];
    }
    if (LANGEVIN && integrates) normal3<R>(s.seed, s.noise_step, s.row0 + (uint64_t)o, g[0], g[1], g[2]);
  }
  // Wait for this atom's force record of THIS launch.  Its pair block has a lower block id: it was dispatched before
  // this block (in-order dispatch of a grid's workgroups — what the hardware does, not something HIP promises) and
  // waits for nothing.  Should that ever not hold, the wait is bounded: the lane gives up, reports F_STEP_TIMEOUT and
  // does NOT integrate its atom; the caller rewinds the batch and repeats it with the separate integrator kernel
  // (judge_flags).  The poll is a volatile device-scope load: nothing may hoist it out of the loop.
  if (!FINAL && !integrates) return;
  const __amdgpu_buffer_rsrc_t frsrc = __builtin_amdgcn_make_buffer_rsrc(fs.fsort, 0, n * (int)sizeof(R4), 0x00020000);
  bool ok = integrates;
  unsigned spins = 0;
  while (ok && !load_force_record(frsrc, R(0), a, fs.watch_gen, x.f)) {
    __builtin_amdgcn_s_sleep(kStepPollSleep);
    if (++spins > fs.poll_limit) {
      s.chk.flags[F_STEP_TIMEOUT] = 1;
      ok = false;  // no update from a stale record
    }
  }
  if constexpr (FINAL == 2) {
    if (ok && forces_out) {  // the complete force of a plain evaluation
#pragma clang fp contract(off)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        R fk = x.f[k];
        if (fs.bonded != 0) fk += fb[k];
        forces_out[3 * o + k] = fk;
      }
    }
    return;
  }
  if constexpr (FINAL == 1) {
    // second half kick (+ thermostat) of the call's last step; the complete force goes to the caller's array (the same
    // sum the kick divides by the mass); kinetic energy of the new velocity, reduced over the wave (every lane is here)
    double ke = 0.0;
    if (ok) {
      R vnew[3];
      md_step_atom<R, true, LANGEVIN, false, false>(s, c, o, 0, s.row0, x, fb, fs.bonded != 0, LANGEVIN ? g : nullptr, vnew);
      if (forces_out) {
#pragma clang fp contract(off)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          R fk = x.f[k];
          if (fs.bonded != 0) fk += fb[k];
          forces_out[3 * o + k] = fk;
        }
      }
      ke = 0.5 * (double)x.m * ((double)vnew[0] * vnew[0] + (double)vnew[1] * vnew[1] + (double)vnew[2] * vnew[2]);  // (kinetic_kernel's expression)
    }
    if (escratch) wave_energy(ke, energy_row(escratch) + kKineticSlot);
    return;
  }
  if (!ok) return;
  md_step_atom<R, true, LANGEVIN, true, true>(s, c, o, 0, s.row0, x, fb, fs.bonded != 0, LANGEVIN ? g : nullptr);
  // (TABLE: the batched launch of a context's replicas — bricks are stepped by tmdhip_dd_run's own launches, domain.hip)
  if (!TABLE && fst->dd_out) {
    // brick of a domain decomposition (dd_own_kernel's extras, same expressions): the running maximum of the squared
    // displacement since the last migration and this atom's rows of the outgoing halo messages
#pragma clang fp contract(off)
    R p[3], dd = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      p[k] = s.pos_out[3 * o + k];  // (this lane's own store of a moment ago)
      const R d = p[k] - fst->dd_ref[3 * o + k];
      dd += d * d;
    }
    const float d2 = sizeof(R) == 4 ? (float)dd : __double2float_ru((double)dd);
    if (__float_as_uint(d2) > *fst->dd_disp2) atomicMax(fst->dd_disp2, __float_as_uint(d2));
    const int s0 = fst->dd_csr_off[o], s1 = fst->dd_csr_off[o + 1];
    for (int q2 = s0; q2 < s1; ++q2) {
      const long long k = fst->dd_csr_row[q2];
#pragma unroll
      for (int xk = 0; xk < 3; ++xk) fst->dd_out[3 * k + xk] = p[xk] + fst->dd_shift[3 * k + xk];
    }
  }
}

}  // namespace tmd
