// Per-pair arithmetic of the nonbonded block (reference torchmd/forces.py:360-491), device side.
//
// Two classes of arithmetic live here:
//  (1) DECISION arithmetic — the minimum image and |d|^2 that decide `dist <= cutoff`
//      (forces.py:76-81, 360-372).  A pair that flips across the cutoff changes a force by up to
//      ~0.05 kcal/mol/A with reaction field (the RF force is non-zero at r_c), so these expressions
//      reproduce the reference's rounding sequence exactly:
//        d  = pos_i - pos_j                       one rounding
//        d -= box * round(d / box)                product and difference rounded separately
//        fp32: |d|^2 = fma(dz,dz, fma(dy,dy, dx*dx))   (what torch.norm's CPU kernel evaluates,
//                                                       verified bitwise on 2e6 vectors)
//        fp64: |d|^2 = (dx*dx + dy*dy) + dz*dz          (no contraction)
//      and the test `sqrt(|d|^2) <= cutoff` is replaced by the equivalent `|d|^2 <= r2max` where
//      r2max is the largest representable value whose correctly-rounded sqrt is <= cutoff
//      (computed on the host, see cutoff_r2max()).  `d / box` is evaluated as `d * (1/box)`: the two
//      can only differ when d/box is within 1 ulp of a half-integer, i.e. |d| ~ box/2 >= cutoff,
//      where the pair is rejected either way.
//  (2) VALUE arithmetic — energies and force magnitudes.  These only need to be accurate to the
//      tolerance (1e-4 fp64 / 1e-2 fp32 kcal/mol/A), so they use rsqrt and contracted FMAs.
#pragma once

#include "common.h"

namespace tmd {

template <typename R>
struct PairConsts {
  R box[3];
  R invbox[3];   // 0 when the box edge is 0 (no wrapping, forces.py:361-362)
  R r2max;       // decision threshold on |d|^2 ; +inf when there is no cutoff
  R switch_dist;
  R inv_switch_range;  // 1/(cutoff - switch_dist)
  R krf, crf;
  uint32_t terms;
  int32_t switch_on;
  int32_t switch_reference_mode;
  int32_t rfa;
};

__device__ __forceinline__ float round_half_even(float x) { return rintf(x); }
__device__ __forceinline__ double round_half_even(double x) { return rint(x); }

template <typename R>
__device__ __forceinline__ R min_image(R d, R box, R invbox) {
#pragma clang fp contract(off)
  R k = round_half_even(d * invbox);
  R p = box * k;
  return d - p;
}

__device__ __forceinline__ float norm2(float dx, float dy, float dz) {
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}
__device__ __forceinline__ double norm2(double dx, double dy, double dz) {
#pragma clang fp contract(off)
  double a = dx * dx;
  double b = dy * dy;
  double c = dz * dz;
  double s = a + b;
  return s + c;
}

__device__ __forceinline__ float fast_rsqrt(float x) { return __frsqrt_rn(x); }
// fp64: the hardware seed (v_rsq_f64, ~2^-26 relative) + one third-order correction, y (1 + h/2 + 3 h^2/8) with h = 1 - x y^2:
// exact to ~2^-76, i.e. to fp64 rounding (round 6: `1.0 / sqrt(x)` — a correctly rounded square root and a division — was most of
// the all-pairs fp64 kernel's arithmetic: thrombin without cutoff, alanine dipeptide in fp64).  VALUE arithmetic only (see above).
__device__ __forceinline__ double fast_rsqrt(double x) {
#ifdef TMD_AB_RSQRT_LIBM
  return 1.0 / sqrt(x);
#else
  const double y = __builtin_amdgcn_rsq(x);
  const double h = __builtin_fma(-(x * y), y, 1.0);
  return __builtin_fma(y, h * __builtin_fma(h, 0.375, 0.5), y);
#endif
}

// Energies e[0..3] = lj, electrostatics, repulsion, repulsioncg (TMDHIP_E_*).
// Returns fscale such that force on i is  -d * fscale  and on j  +d * fscale
// (forces.py:316-319 with unitvec = d/r, force_coeff = dE/dr).
template <typename R, bool ENERGY>
__device__ __forceinline__ R pair_terms(const PairConsts<R> &c, R r2, R qq, R A, R B, R *e) {
  const R rinv = fast_rsqrt(r2);
  const R rinv2 = rinv * rinv;
  const R r = r2 * rinv;
  const R rinv6 = rinv2 * rinv2 * rinv2;
  R dEdr = R(0);
  if (c.terms & TMDHIP_TERM_LJ) {  // forces.py:390-415
    R elj = (A * rinv6 - B) * rinv6;
    R f = (R(-12) * A * rinv6 + R(6) * B) * rinv6 * rinv;
    if (c.switch_on && r > c.switch_dist) {
      const R t = (r - c.switch_dist) * c.inv_switch_range;
      const R sw = R(1) + t * t * t * (R(-10) + t * (R(15) - t * R(6)));
      const R dsw = t * t * (R(-30) + t * (R(60) - t * R(30))) * c.inv_switch_range;
      // upstream's explicit force divides the switching term by r once more (forces.py:410-412)
      f = sw * f + elj * dsw * (c.switch_reference_mode ? rinv : R(1));
      elj *= sw;
    }
    dEdr += f;
    if (ENERGY) e[0] += elj;
  }
  if (c.terms & TMDHIP_TERM_ELECTROSTATICS) {  // forces.py:453-491
    if (c.rfa) {
      dEdr += qq * (R(2) * c.krf * r - rinv2);
      if (ENERGY) e[1] += qq * (rinv + c.krf * r2 - c.crf);
    } else {
      const R eel = qq * rinv;
      dEdr -= eel * rinv;
      if (ENERGY) e[1] += eel;
    }
  }
  if (c.terms & TMDHIP_TERM_REPULSION) {  // forces.py:418-433
    const R er = A * rinv6 * rinv6;
    dEdr += R(-12) * er * rinv;
    if (ENERGY) e[2] += er;
  }
  if (c.terms & TMDHIP_TERM_REPULSIONCG) {  // forces.py:436-450
    const R er = B * rinv6;
    dEdr += R(-6) * er * rinv;
    if (ENERGY) e[3] += er;
  }
  return dEdr * rinv;
}

// wave-wide sum (64 lanes), result valid in every lane
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Energy accumulation.  Thousands of waves adding into the same 8 doubles serialise on one L2 atomic
// unit (~10 ns per atomic: 25 k wave sums = 0.26 ms on the 98k-atom water box, 6x the pair kernel).
// Waves add into one of kEnergySlots scratch rows (128 B apart) instead and energy_fold_kernel adds
// the rows into the caller's 8 doubles and clears them for the next call.
constexpr int kEnergySlots = 256;
constexpr int kEnergyStride = 16;  // doubles per row (8 used)

__device__ __forceinline__ double *energy_row(double *scratch) {
  const unsigned w = (blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  return scratch + (size_t)(w & (kEnergySlots - 1)) * kEnergyStride;
}

// one block of kEnergySlots threads per replica
static __global__ __launch_bounds__(kEnergySlots) void energy_fold_kernel(double *__restrict__ scratch,
                                                                   double *__restrict__ out) {
  __shared__ double part[kEnergySlots / 64][8];
  scratch += (size_t)blockIdx.x * kEnergySlots * kEnergyStride;
  out += (size_t)blockIdx.x * 8;
  double *row = scratch + (size_t)threadIdx.x * kEnergyStride;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double v = row[k];
    if (v != 0.0) row[k] = 0.0;
    const double s = wave_sum(v);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kEnergySlots / 64; ++w) s += part[w][threadIdx.x];
    if (s != 0.0) out[threadIdx.x] += s;
  }
}

// the same for the final step of an MD call made by FINAL step blocks (md_step.h): slot TMDHIP_NENERGY of the rows holds
// the kinetic energy; one block of kEnergySlots threads per replica (blockIdx.x: its scratch rows, its 8 energies, its word of `ke`)
static __global__ __launch_bounds__(kEnergySlots) void final_fold_kernel(double *__restrict__ scratch, double *__restrict__ out,
                                                                  double *__restrict__ ke) {
  __shared__ double part[kEnergySlots / 64][TMDHIP_NENERGY + 1];
  scratch += (size_t)blockIdx.x * kEnergySlots * kEnergyStride;
  out += (size_t)blockIdx.x * TMDHIP_NENERGY;
  ke += blockIdx.x;
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
      if (s != 0.0) out[threadIdx.x] += s;
    } else {
      ke[0] = s;
    }
  }
}

}  // namespace tmd
