// K3d: the lean list pair kernel for fp64 contexts (gfx950).  Reference semantics as pair_fast_f32.hip
// (torchmd/forces.py:260-319, LJ and/or electrostatics).
#include "engine.h"

namespace tmd {

// ---- K3d: the same lean kernel for fp64 contexts ----------------------------------------------------
// 32-byte records (two 16-byte gathers per entry), 16-byte table entries, half-rate arithmetic; 1/r from v_rsq_f64
// and two Newton steps.  Same entry format, list layout and decision arithmetic (min_image_magic's fp64 overload:
// magic number 1.5 * 2^52; norm2's fp64 order).
// (Step blocks behind the pair blocks — the MD step inside the launch, as in the fp32 kernel — were built for fp64 in
// round 4, bit-identical and SLOWER at C3: 151 against 124.5 us per step; the code is in commit daee5f8, the record in
// docs/history/round4.md.  fp64 contexts keep the separate integrator kernel.)
template <int LPA, bool LJ, bool ELEC, bool ENERGY, bool SWITCH>
__global__ __launch_bounds__(256, 2) void list_pair_lean_f64_kernel(
    int n, const double4 *__restrict__ sorted, const int *__restrict__ stype, const int *__restrict__ order,
    int ntypes, const double2 *__restrict__ tab, const unsigned *__restrict__ nlist,
    const int *__restrict__ nneigh, int maxn, PairConsts<double> c, double *__restrict__ forces, int overwrite,
    double *__restrict__ energies, unsigned *publish, unsigned publish_value, const int *__restrict__ ext, int *lflags, int lmode) {
  constexpr int APW = 64 / LPA;
  constexpr int UNROLL = 4;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // tells the host (host-mapped word) that everything enqueued before this launch has completed
    if (publish) __hip_atomic_store(publish, publish_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (lflags) {  // list duties of the launch's first thread (as in the fp32 kernel)
      const int parity = (lmode & kLmParity) ? 1 : 0;
      if ((lmode & kLmViolation) && lflags[F_REBUILD0 + parity] != 0) lflags[F_VIOLATION] = 1;
    }
  }
  __shared__ __align__(16) double2 stab[kEntryTypes * kEntryTypes];  // row of type i: 32 x {-12 A, 6 B}
  const unsigned npair = gridDim.x;
  for (int t = threadIdx.x; t < ntypes * kEntryTypes; t += blockDim.x) {  // rows of existing classes only
    const int ti = t >> 5, tj = t & 31;
    double2 ab = make_double2(0.0, 0.0);
    if (tj < ntypes) ab = tab[ti * ntypes + tj];
    stab[t] = make_double2(-12.0 * ab.x, 6.0 * ab.y);
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  // XCD-aware block order: consecutive block ids go to the 8 XCDs round-robin, so block b works on
  // chunk (b % 8) * npair/8 + b / 8 — every XCD (own L2) gets a contiguous eighth of the cell-sorted
  // atoms and gathers neighbours from that region only.  npair is a multiple of 8; the surplus
  // blocks of the last eighths have nothing to do.
  const int blk = (int)((blockIdx.x & 7u) * (npair >> 3) + (blockIdx.x >> 3));
  if (blk * 4 * APW >= n) return;
  const int wave = blk * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int a = wave * APW + lane / LPA;
  const int sub = lane % LPA;
  const bool active = a < n;
  double4 pi = make_double4(0.0, 0.0, 0.0, 0.0);
  int nn = 0;
  unsigned trow = 0;  // byte offset of this atom's row of the LDS table
  if (active) {
    pi = sorted[a];
    nn = nneigh[a];
    trow = (unsigned)stype[a] << 9;  // rows of 32 x 16 B
  }
  const int myiters = (nn - sub + LPA - 1) / LPA;  // entries kk < myiters are real for this lane
  int itmax = myiters, itmin = myiters;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    itmax = max(itmax, __shfl_xor(itmax, o, 64));
    itmin = min(itmin, __shfl_xor(itmin, o, 64));
  }
  const int nkk = __builtin_amdgcn_readfirstlane(itmax);
  const int nfull = __builtin_amdgcn_readfirstlane(itmin) / UNROLL * UNROLL;  // iterations every lane has entries for
  // a lane's entries of iterations 4G .. 4G+3 are one 16-byte word at row4[G * 64]
  const v4u *row4 = reinterpret_cast<const v4u *>(nlist + (size_t)wave * maxn * APW) + lane;
  // bounds-checked raw buffer over sorted_xyzq: lanes past the end of their list read whatever the
  // (uninitialised) padding entry points at — out-of-range offsets return 0 instead of faulting — and
  // are discarded by `valid`
  const __amdgpu_buffer_rsrc_t srsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<double4 *>(sorted), 0, n * 32, 0x00020000);
  const char *tbase = reinterpret_cast<const char *>(stab);
  const double two_krf = 2.0 * c.krf;
  const double qi2k = pi.w * two_krf;
  const double sw_ir = c.inv_switch_range, sw_t0 = -c.switch_dist * c.inv_switch_range;
  const double bx = c.box[0], by = c.box[1], bz = c.box[2];
  const double ibx = c.invbox[0], iby = c.invbox[1], ibz = c.invbox[2];
  const double r2max = c.r2max;

  double fx = 0.0, fy = 0.0, fz = 0.0;
  double e_lj = 0.0, e_el = 0.0;

  auto body = [&](auto image, unsigned tofs, const v4u &lo, const v4u &hi, bool valid) {  // one list entry
    constexpr bool EXACT = decltype(image)::value;
    const double pjx = __hiloint2double((int)lo.y, (int)lo.x), pjy = __hiloint2double((int)lo.w, (int)lo.z);
    const double pjz = __hiloint2double((int)hi.y, (int)hi.x), pjw = __hiloint2double((int)hi.w, (int)hi.z);
    const double dx = min_image_magic<EXACT>(pi.x - pjx, bx, ibx);
    const double dy = min_image_magic<EXACT>(pi.y - pjy, by, iby);
    const double dz = min_image_magic<EXACT>(pi.z - pjz, bz, ibz);
    const double r2 = norm2(dx, dy, dz);
    const bool hit = valid && (r2 <= r2max);
    // 1/r: v_rsq_f64 (~2^-26 relative) + one third-order correction (round 6; was two Newton steps: three fp64 operations more);
    // rejected entries may produce inf/NaN, discarded below
    double rinv = __builtin_amdgcn_rsq(r2);
#ifdef TMD_AB_NEWTON2
    rinv = rinv * __builtin_fma(-0.5 * r2 * rinv, rinv, 1.5);
    rinv = rinv * __builtin_fma(-0.5 * r2 * rinv, rinv, 1.5);
#else
    {  // one third-order step instead of two Newton steps: h = 1 - r2 y^2 ~ 2^-25, y (1 + h/2 + 3 h^2 / 8) is exact to 2^-76
      const double h = __builtin_fma(-(r2 * rinv), rinv, 1.0);
      rinv = __builtin_fma(rinv, h * __builtin_fma(h, 0.375, 0.5), rinv);
    }
#endif
    const double rinv2 = rinv * rinv;
    const double rinv6 = rinv2 * rinv2 * rinv2;
    double fs;  // (dE/dr) / r
    double2 ab = make_double2(0.0, 0.0);  // (-12 A, 6 B)
    if (LJ) ab = *reinterpret_cast<const double2 *>(tbase + (trow | tofs));
    auto elj_of = [&](double r6) { return __builtin_fma(ab.x * (-1.0 / 12.0), r6, ab.y * (-1.0 / 6.0)) * r6; };
    if (LJ && !SWITCH && ELEC) {
      const double qq = pi.w * pjw;
      const double p = __builtin_fma(ab.x, rinv6, ab.y) * rinv6;
      const double g = __builtin_fma(-qq, rinv, p);
      fs = __builtin_fma(rinv2, g, qi2k * pjw);
      if (ENERGY) e_lj += hit ? elj_of(rinv6) : 0.0;
    } else {
      fs = 0.0;
      double sw = 1.0;
      if (LJ) {
        fs = __builtin_fma(ab.x, rinv6, ab.y) * (rinv6 * rinv2);
        if (SWITCH) {  // same polynomial as the fp32 kernel (forces.py:402-412)
          const double r = r2 * rinv;
          const double t = fmax(__builtin_fma(r, sw_ir, sw_t0), 0.0);
          const double t2 = t * t;
          const double pp = __builtin_fma(t, __builtin_fma(t, -6.0, 15.0), -10.0);
          sw = __builtin_fma(t2 * t, pp, 1.0);
          const double dq = __builtin_fma(t, __builtin_fma(t, -30.0 * sw_ir, 60.0 * sw_ir), -30.0 * sw_ir);
          const double elj = elj_of(rinv6);
          const double x = c.switch_reference_mode ? rinv2 : rinv;
          fs = __builtin_fma(sw, fs, elj * (t2 * dq) * x);
        }
        if (ENERGY) e_lj += hit ? sw * elj_of(rinv6) : 0.0;
      }
      if (ELEC) fs += (pi.w * pjw) * (two_krf - rinv2 * rinv);
    }
    if (ENERGY && ELEC) e_el += hit ? (pi.w * pjw) * (rinv + c.krf * r2 - c.crf) : 0.0;
    fs = hit ? fs : 0.0;
    fx = __builtin_fma(-dx, fs, fx);
    fy = __builtin_fma(-dy, fs, fy);
    fz = __builtin_fma(-dz, fs, fz);
  };

  static_assert(UNROLL == 4, "one dwordx4 of list per lane and group");
  // index words are fetched two groups (8 entries per lane, 2 KB per wave) ahead of their use
  v4u nxa = row4[0], nxb = row4[64];  // rows are padded: always readable
  int kk0 = 0;
  auto checked_loop = [&](auto image) {  // per-lane validity
    for (; kk0 < nkk; kk0 += UNROLL) {
      const v4u cur = nxa;
      nxa = nxb;
      nxb = row4[(size_t)((kk0 >> 2) + 2) * 64];
      const unsigned entry[UNROLL] = {cur.x, cur.y, cur.z, cur.w};
      v4u lo[UNROLL], hi[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const unsigned off = (entry[u] & kEntryOffMask) << 1;
        lo[u] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, off, 0, 0);
        hi[u] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, off + 16u, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) body(image, (entry[u] >> 23) & 0x1F0u, lo[u], hi[u], kk0 + u < myiters);  // padding words are garbage
    }
  };
  if (extent_needs_exact_image(ext, c.box)) {  // wave-uniform, rare: atoms more than 2.4 box edges apart
    checked_loop(exact_image{});
  } else {
    for (; kk0 < nfull; kk0 += UNROLL) {  // every lane has real entries here: no validity test
      const v4u cur = nxa;
      nxa = nxb;
      nxb = row4[(size_t)((kk0 >> 2) + 2) * 64];
      const unsigned entry[UNROLL] = {cur.x, cur.y, cur.z, cur.w};
      v4u lo[UNROLL], hi[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {  // 32-byte records: byte offset = 2 x the entry's 16-byte-record offset
        const unsigned off = (entry[u] & kEntryOffMask) << 1;
        lo[u] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, off, 0, 0);
        hi[u] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, off + 16u, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) body(fused_image{}, (entry[u] >> 23) & 0x1F0u, lo[u], hi[u], true);
    }
    checked_loop(fused_image{});  // tail
  }
  double sx = fx, sy = fy, sz = fz;
#pragma unroll
  for (int o = LPA >> 1; o > 0; o >>= 1) {
    sx += __shfl_xor(sx, o, 64);
    sy += __shfl_xor(sy, o, 64);
    sz += __shfl_xor(sz, o, 64);
  }
  if (active && sub == 0 && forces) {
    const int oi = order[a];
    if (overwrite) {
      forces[3 * oi + 0] = sx;
      forces[3 * oi + 1] = sy;
      forces[3 * oi + 2] = sz;
    } else {
      forces[3 * oi + 0] += sx;
      forces[3 * oi + 1] += sy;
      forces[3 * oi + 2] += sz;
    }
  }
  if (ENERGY) {  // every pair is listed from both atoms: half of the sum
    if (LJ) {
      const double s = wave_sum(e_lj);
      if (lane == 0 && s != 0.0) unsafeAtomicAdd(&energy_row(energies)[TMDHIP_E_LJ], 0.5 * s);
    }
    if (ELEC) {
      const double s = wave_sum(e_el);
      if (lane == 0 && s != 0.0) unsafeAtomicAdd(&energy_row(energies)[TMDHIP_E_ELECTROSTATICS], 0.5 * s);
    }
  }
}

// host side: one launch of the lean fp64 kernel over the replica's list
template <bool ENERGY>
int launch_pair_lean_f64(tmdhip_ctx *ctx, Replica &rp, const PairConsts<double> &c, double *f, int overwrite,
                         hipStream_t st, hipEvent_t e0, hipEvent_t e1, int lmode) {
  const int n = ctx->d.natoms;
  const int apw = rp.lg.apw;
  const int waves = (n + apw - 1) / apw;
  const int blocks = (waves + 3) / 4;
  const int npair8 = (blocks + 7) / 8 * 8;
  const bool lj = c.terms & TMDHIP_TERM_LJ, el = c.terms & TMDHIP_TERM_ELECTROSTATICS;
#define TMD_LAUNCH_FAST_T(L, A, B)       \
  if (c.switch_on && A) {               \
    TMD_LAUNCH_FAST_S(L, A, B, true);   \
  } else {                              \
    TMD_LAUNCH_FAST_S(L, A, B, false);  \
  }
#define TMD_LAUNCH_FAST_S(L, A, B, S)                                                                                       \
  launch_with_events(list_pair_lean_f64_kernel<L, A, B, ENERGY, S>, dim3(npair8), dim3(256), 0u, st, e0, e1, n,              \
                     rp.sorted.as<double4>(), rp.stype.as<int>(), rp.order.as<int>(), ctx->d.ntypes, ctx->tab.as<double2>(), \
                     rp.nlist.as<unsigned>(), rp.nneigh.as<int>(), rp.lg.maxn, c, f, overwrite, ctx->escratch.as<double>(),  \
                     rp.pub_ptr, rp.pub_val, rp.extent.as<int>(), rp.flags.as<int>(), lmode)
#define TMD_LAUNCH_FAST(L)               \
  if (lj && el) {                        \
    TMD_LAUNCH_FAST_T(L, true, true);    \
  } else if (lj) {                       \
    TMD_LAUNCH_FAST_T(L, true, false);   \
  } else {                               \
    TMD_LAUNCH_FAST_T(L, false, true);   \
  }
  switch (rp.lg.lpa) {  // (pick_lpa never returns less than 4)
    case 4: TMD_LAUNCH_FAST(4); break;
    case 8: TMD_LAUNCH_FAST(8); break;
    case 16: TMD_LAUNCH_FAST(16); break;
    case 32: TMD_LAUNCH_FAST(32); break;
    default: TMD_LAUNCH_FAST(64); break;
  }
#undef TMD_LAUNCH_FAST
#undef TMD_LAUNCH_FAST_T
#undef TMD_LAUNCH_FAST_S
  TMD_HIP(hipGetLastError());
  return 0;
}

template int launch_pair_lean_f64<true>(tmdhip_ctx *, Replica &, const PairConsts<double> &, double *, int, hipStream_t,
                                        hipEvent_t, hipEvent_t, int);
template int launch_pair_lean_f64<false>(tmdhip_ctx *, Replica &, const PairConsts<double> &, double *, int, hipStream_t,
                                         hipEvent_t, hipEvent_t, int);

}  // namespace tmd
