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

}  // namespace tmd
