// Velocity-Verlet / Langevin kernels for gfx950.
//
// Reference semantics: torchmd/integrator.py:61-74 (_first_VV, _second_VV, langevin) and 8-31
// (kinetic_energy).  The reference issues ~10 elementwise torch ops plus one randn per step over
// [R,N,3]; here each half step is ONE streaming kernel (HBM-bound: first half reads pos,vel,F,m and
// writes pos,vel = 64 B/atom fp32; second half reads vel,F,m,(vcoeff) writes vel = 44 B/atom), and
// the Gaussian noise is generated in registers from a counter-based Philox4x32-10 stream, so no noise
// tensor ever touches HBM.  Operation order inside each expression follows the reference so the
// closed-form tests of tests/test_integrator.py hold to rounding.
#include <hip/hip_runtime.h>

#include "common.h"
#include "pair_math.h"
#include "rng.h"

using namespace tmd;

namespace {

// integrator.py:61-64
template <typename R>
__global__ void first_vv_kernel(int64_t rows, int64_t natoms, R *__restrict__ pos, R *__restrict__ vel,
                                const R *__restrict__ f, const R *__restrict__ mass, R dt, R half_dt) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const R m = mass[i % natoms];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const R a = f[3 * i + k] / m;
    const R v = vel[3 * i + k];
    pos[3 * i + k] += v * dt + R(0.5) * a * dt * dt;
    vel[3 * i + k] = v + half_dt * a;
  }
}

// integrator.py:67-69
template <typename R>
__global__ void second_vv_kernel(int64_t rows, int64_t natoms, R *__restrict__ vel, const R *__restrict__ f,
                                 const R *__restrict__ mass, R half_dt) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const R m = mass[i % natoms];
#pragma unroll
  for (int k = 0; k < 3; ++k) vel[3 * i + k] += half_dt * (f[3 * i + k] / m);
}

// integrator.py:72-74 then 67-69
template <typename R>
__global__ void langevin_second_vv_kernel(int64_t rows, int64_t natoms, R *__restrict__ vel,
                                          const R *__restrict__ f, const R *__restrict__ mass,
                                          const R *__restrict__ vcoeff, R dt, R half_dt, R gamma, uint64_t seed,
                                          uint64_t step) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const R m = mass[i % natoms];
  const R vc = vcoeff[i % natoms];
  R g[3];
  normal3<R>(seed, step, (uint64_t)i, g[0], g[1], g[2]);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    R v = vel[3 * i + k];
    v += -gamma * v * dt + g[k] * vc;
    v += half_dt * (f[3 * i + k] / m);
    vel[3 * i + k] = v;
  }
}

template <typename R>
__global__ void normal_fill_kernel(int64_t n, R *__restrict__ out, uint64_t seed, uint64_t step) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (3 * row >= n) return;
  R g[3];
  normal3<R>(seed, step, (uint64_t)row, g[0], g[1], g[2]);
  for (int k = 0; k < 3; ++k)
    if (3 * row + k < n) out[3 * row + k] = g[k];
}

// integrator.py:8-31: grid = (blocks, replicas)
// DIRECT: one block per replica (small systems) stores its sum — no clear of `ke` in front, no atomics
template <typename R, bool DIRECT>
__global__ __launch_bounds__(256) void kinetic_kernel(int64_t natoms, const R *__restrict__ vel,
                                                      const R *__restrict__ mass, double *__restrict__ ke) {
  const int r = blockIdx.y;
  const R *v = vel + (size_t)r * natoms * 3;
  double acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < natoms; i += (int64_t)gridDim.x * blockDim.x) {
    const R vx = v[3 * i], vy = v[3 * i + 1], vz = v[3 * i + 2];
    acc += 0.5 * (double)mass[i] * ((double)vx * vx + (double)vy * vy + (double)vz * vz);
  }
  acc = wave_sum(acc);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    if (DIRECT) ke[r] = part[0] + part[1] + part[2] + part[3];
    else unsafeAtomicAdd(&ke[r], part[0] + part[1] + part[2] + part[3]);
  }
}

inline dim3 blocks_for(int64_t n, int t) { return dim3((unsigned)((n + t - 1) / t)); }

int check(int dtype, int64_t R_, int64_t N_) {
  if (dtype != TMDHIP_F32 && dtype != TMDHIP_F64) return fail("integrator: bad dtype");
  if (R_ <= 0 || N_ <= 0) return fail("integrator: nreplicas and natoms must be positive");
  return 0;
}

}  // namespace

extern "C" {

int tmdhip_first_vv(int dtype, int64_t nreplicas, int64_t natoms, void *pos, void *vel, const void *forces,
                    const void *mass, double dt, void *stream) {
  TMD_TRY(check(dtype, nreplicas, natoms));
  if (!pos || !vel || !forces || !mass) return fail("tmdhip_first_vv: null pointer");
  const int64_t rows = nreplicas * natoms;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TMDHIP_F32)
    hipLaunchKernelGGL((first_vv_kernel<float>), blocks_for(rows, 256), dim3(256), 0, st, rows, natoms, (float *)pos,
                       (float *)vel, (const float *)forces, (const float *)mass, (float)dt, (float)(0.5 * dt));
  else
    hipLaunchKernelGGL((first_vv_kernel<double>), blocks_for(rows, 256), dim3(256), 0, st, rows, natoms,
                       (double *)pos, (double *)vel, (const double *)forces, (const double *)mass, dt, 0.5 * dt);
  TMD_HIP(hipGetLastError());
  return 0;
}

int tmdhip_second_vv(int dtype, int64_t nreplicas, int64_t natoms, void *vel, const void *forces, const void *mass,
                     double dt, void *stream) {
  TMD_TRY(check(dtype, nreplicas, natoms));
  if (!vel || !forces || !mass) return fail("tmdhip_second_vv: null pointer");
  const int64_t rows = nreplicas * natoms;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TMDHIP_F32)
    hipLaunchKernelGGL((second_vv_kernel<float>), blocks_for(rows, 256), dim3(256), 0, st, rows, natoms,
                       (float *)vel, (const float *)forces, (const float *)mass, (float)(0.5 * dt));
  else
    hipLaunchKernelGGL((second_vv_kernel<double>), blocks_for(rows, 256), dim3(256), 0, st, rows, natoms,
                       (double *)vel, (const double *)forces, (const double *)mass, 0.5 * dt);
  TMD_HIP(hipGetLastError());
  return 0;
}

int tmdhip_langevin_second_vv(int dtype, int64_t nreplicas, int64_t natoms, void *vel, const void *forces,
                              const void *mass, const void *vcoeff, double dt, double gamma, uint64_t seed,
                              uint64_t step, void *stream) {
  TMD_TRY(check(dtype, nreplicas, natoms));
  if (!vel || !forces || !mass || !vcoeff) return fail("tmdhip_langevin_second_vv: null pointer");
  const int64_t rows = nreplicas * natoms;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TMDHIP_F32)
    hipLaunchKernelGGL((langevin_second_vv_kernel<float>), blocks_for(rows, 256), dim3(256), 0, st, rows, natoms,
                       (float *)vel, (const float *)forces, (const float *)mass, (const float *)vcoeff, (float)dt,
                       (float)(0.5 * dt), (float)gamma, seed, step);
  else
    hipLaunchKernelGGL((langevin_second_vv_kernel<double>), blocks_for(rows, 256), dim3(256), 0, st, rows, natoms,
                       (double *)vel, (const double *)forces, (const double *)mass, (const double *)vcoeff, dt,
                       0.5 * dt, gamma, seed, step);
  TMD_HIP(hipGetLastError());
  return 0;
}

int tmdhip_kinetic_energy(int dtype, int64_t nreplicas, int64_t natoms, const void *vel, const void *mass,
                          double *ke, void *stream) {
  TMD_TRY(check(dtype, nreplicas, natoms));
  if (!vel || !mass || !ke) return fail("tmdhip_kinetic_energy: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (natoms <= 8192) {  // small systems (launch-bound): one block per replica, one launch instead of a clear + a launch
    dim3 grid(1, (unsigned)nreplicas);
    if (dtype == TMDHIP_F32)
      hipLaunchKernelGGL((kinetic_kernel<float, true>), grid, dim3(256), 0, st, natoms, (const float *)vel, (const float *)mass, ke);
    else
      hipLaunchKernelGGL((kinetic_kernel<double, true>), grid, dim3(256), 0, st, natoms, (const double *)vel, (const double *)mass, ke);
    TMD_HIP(hipGetLastError());
    return 0;
  }
  TMD_HIP(hipMemsetAsync(ke, 0, sizeof(double) * nreplicas, st));
  const unsigned nb = (unsigned)std::min<int64_t>((natoms + 255) / 256, 256);
  dim3 grid(nb, (unsigned)nreplicas);
  if (dtype == TMDHIP_F32)
    hipLaunchKernelGGL((kinetic_kernel<float, false>), grid, dim3(256), 0, st, natoms, (const float *)vel,
                       (const float *)mass, ke);
  else
    hipLaunchKernelGGL((kinetic_kernel<double, false>), grid, dim3(256), 0, st, natoms, (const double *)vel,
                       (const double *)mass, ke);
  TMD_HIP(hipGetLastError());
  return 0;
}

int tmdhip_normal_fill(int dtype, int64_t n, void *out, uint64_t seed, uint64_t step, void *stream) {
  if (dtype != TMDHIP_F32 && dtype != TMDHIP_F64) return fail("tmdhip_normal_fill: bad dtype");
  if (n <= 0 || !out) return fail("tmdhip_normal_fill: bad arguments");
  const int64_t rows = (n + 2) / 3;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TMDHIP_F32)
    hipLaunchKernelGGL((normal_fill_kernel<float>), blocks_for(rows, 256), dim3(256), 0, st, n, (float *)out, seed, step);
  else
    hipLaunchKernelGGL((normal_fill_kernel<double>), blocks_for(rows, 256), dim3(256), 0, st, n, (double *)out, seed, step);
  TMD_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Molecule wrapping (reference torchmd/wrapper.py:8-30): every bonded group is translated by
// -floor(com/box)*box, com = unweighted mean of its atoms.  The reference loops over the groups in
// Python (4 torch ops per molecule: seconds per call at 32 768 waters); here one thread per group
// computes the offset (members summed in index order -> deterministic) and shifts its atoms.
// Groups larger than 64 atoms are handled by a whole wave.
// ---------------------------------------------------------------------------------------------
namespace {

template <typename R>
__global__ void wrap_groups_kernel(int64_t natoms, int ngroups, const int *__restrict__ goff,
                                   const int *__restrict__ gmem, R *__restrict__ pos, const R *__restrict__ box,
                                   int big_threshold, int big_pass) {
  // grid.y = replica; small groups: one thread each; big groups: one wave each (second launch)
  const int r = blockIdx.y;
  R *p = pos + (size_t)r * natoms * 3;
  const R bx = box[9 * r + 0], by = box[9 * r + 4], bz = box[9 * r + 8];
  if (bx == R(0) && by == R(0) && bz == R(0)) return;
  if (!big_pass) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups) return;
    const int s = goff[g], e = goff[g + 1];
    if (e - s > big_threshold) return;
    R sx = 0, sy = 0, sz = 0;
    for (int k = s; k < e; ++k) {
      const int a = gmem[k];
      sx += p[3 * a], sy += p[3 * a + 1], sz += p[3 * a + 2];
    }
    const R n = (R)(e - s);
    const R ox = floor((sx / n) / bx) * bx, oy = floor((sy / n) / by) * by, oz = floor((sz / n) / bz) * bz;
    for (int k = s; k < e; ++k) {
      const int a = gmem[k];
      p[3 * a] -= ox, p[3 * a + 1] -= oy, p[3 * a + 2] -= oz;
    }
  } else {
    const int g = blockIdx.x;
    const int s = goff[g], e = goff[g + 1];
    if (e - s <= big_threshold) return;
    const int lane = threadIdx.x;
    double sx = 0, sy = 0, sz = 0;
    for (int k = s + lane; k < e; k += 64) {
      const int a = gmem[k];
      sx += (double)p[3 * a], sy += (double)p[3 * a + 1], sz += (double)p[3 * a + 2];
    }
    sx = wave_sum(sx), sy = wave_sum(sy), sz = wave_sum(sz);
    const R n = (R)(e - s);
    const R ox = floor(((R)sx / n) / bx) * bx, oy = floor(((R)sy / n) / by) * by, oz = floor(((R)sz / n) / bz) * bz;
    for (int k = s + lane; k < e; k += 64) {
      const int a = gmem[k];
      p[3 * a] -= ox, p[3 * a + 1] -= oy, p[3 * a + 2] -= oz;
    }
  }
}

}  // namespace

extern "C" int tmdhip_wrap(int dtype, int64_t nreplicas, int64_t natoms, void *pos, const void *box_dev,
                           int32_t ngroups, const int32_t *group_offsets_dev, const int32_t *group_members_dev,
                           int32_t has_big_groups, void *stream) {
  TMD_TRY(check(dtype, nreplicas, natoms));
  if (!pos || !box_dev || !group_offsets_dev || !group_members_dev || ngroups <= 0)
    return fail("tmdhip_wrap: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int BIG = 64;
  dim3 g1((unsigned)((ngroups + 255) / 256), (unsigned)nreplicas), g2((unsigned)ngroups, (unsigned)nreplicas);
  if (dtype == TMDHIP_F32) {
    hipLaunchKernelGGL((wrap_groups_kernel<float>), g1, dim3(256), 0, st, natoms, ngroups, group_offsets_dev,
                       group_members_dev, (float *)pos, (const float *)box_dev, BIG, 0);
    if (has_big_groups)
      hipLaunchKernelGGL((wrap_groups_kernel<float>), g2, dim3(64), 0, st, natoms, ngroups, group_offsets_dev,
                         group_members_dev, (float *)pos, (const float *)box_dev, BIG, 1);
  } else {
    hipLaunchKernelGGL((wrap_groups_kernel<double>), g1, dim3(256), 0, st, natoms, ngroups, group_offsets_dev,
                       group_members_dev, (double *)pos, (const double *)box_dev, BIG, 0);
    if (has_big_groups)
      hipLaunchKernelGGL((wrap_groups_kernel<double>), g2, dim3(64), 0, st, natoms, ngroups, group_offsets_dev,
                         group_members_dev, (double *)pos, (const double *)box_dev, BIG, 1);
  }
  TMD_HIP(hipGetLastError());
  return 0;
}
