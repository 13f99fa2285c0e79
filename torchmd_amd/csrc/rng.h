// Counter-based Gaussian noise for the Langevin thermostat (shared by integrator.hip and the fused
// MD-step kernels in md_loop.hip).
#pragma once

#include "common.h"

namespace tmd {

// ---- Philox4x32-10 (Salmon et al., SC'11) ----------------------------------------------------
struct Philox {
  uint32_t c[4];
};
__device__ __forceinline__ Philox philox4x32_10(uint64_t ctr_lo, uint64_t ctr_hi, uint64_t key) {
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0, c1 = n1, c2 = n2, c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return Philox{{c0, c1, c2, c3}};
}

// three N(0,1) draws for row `row` of step `step` (Box-Muller on 32-bit uniforms in (0,1))
template <typename R>
__device__ __forceinline__ void normal3(uint64_t seed, uint64_t step, uint64_t row, R &g0, R &g1, R &g2) {
  const Philox p = philox4x32_10(row, step, seed);
  // Uniforms (c + 0.5) 2^-32 straight in fp32 (the top 24 bits of c survive, u never reaches 0; 1 is clamped off), and the
  // hardware transcendentals: v_log_f32 (log2, ~1 ulp), v_sqrt_f32, v_sin/v_cos_f32 (argument in revolutions, abs. error
  // ~1e-6 — far below the statistical resolution of a thermostat).  ~40 instructions instead of ~250 with fp64 conversions
  // and the libm routines.  fp64 contexts draw the same fp32 variates and widen them (round 6: the libm path was a third of
  // the fp64 integrator kernel, 17.4 us at C3; the reference draws torch.randn_like from another generator anyway,
  // integrator.py:73 — trajectory parity is statistical in either precision).
  const float inv32 = 2.3283064365386963e-10f;  // 2^-32
  const float u0 = fminf(__builtin_fmaf((float)p.c[0], inv32, 0.5f * inv32), 0.99999994f);
  const float u1 = __builtin_fmaf((float)p.c[1], inv32, 0.5f * inv32);
  const float u2 = fminf(__builtin_fmaf((float)p.c[2], inv32, 0.5f * inv32), 0.99999994f);
  const float u3 = __builtin_fmaf((float)p.c[3], inv32, 0.5f * inv32);
  const float kLn2x2 = -1.3862943611198906f;  // -2 ln 2:  -2 ln u = kLn2x2 * log2 u
  const float r0 = __builtin_amdgcn_sqrtf(kLn2x2 * __builtin_amdgcn_logf(u0));
  const float r1 = __builtin_amdgcn_sqrtf(kLn2x2 * __builtin_amdgcn_logf(u2));
  g0 = (R)(r0 * __builtin_amdgcn_cosf(u1));
  g1 = (R)(r0 * __builtin_amdgcn_sinf(u1));
  g2 = (R)(r1 * __builtin_amdgcn_cosf(u3));
}


}  // namespace tmd
