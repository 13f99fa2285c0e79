// Launcher of the lean fp32 list pair kernel (pair_fast_kernel.h), one replica per launch.
#include "pair_fast_kernel.h"

namespace tmd {

// host side: one launch of the lean fp32 kernel over the replica's list (fl: with step blocks behind the pair blocks)
template <bool ENERGY>
int launch_pair_fast_f32(tmdhip_ctx *ctx, Replica &rp, const PairConsts<float> &c, float *f, int overwrite, hipStream_t st,
                         hipEvent_t e0, hipEvent_t e1, int lmode, const FusedLaunch *fl) {
  const int n = ctx->d.natoms;
  const int apw = rp.lg.apw;
  const int waves = (n + apw - 1) / apw;
  const bool lj = c.terms & TMDHIP_TERM_LJ, el = c.terms & TMDHIP_TERM_ELECTROSTATICS;
#define TMD_LAUNCH_FAST_T(L, A, B, F)       \
  if (c.switch_on && A) {                  \
    TMD_LAUNCH_FAST_S(L, A, B, true, F);   \
  } else {                                 \
    TMD_LAUNCH_FAST_S(L, A, B, false, F);  \
  }
#define TMD_LAUNCH_FAST_S(L, A, B, S, F)                                                                                  \
  launch_with_events(list_pair_fast_f32_kernel<L, A, B, ENERGY, S, F>, dim3(npair8 + (F ? fstep.nstep_blocks : 0)),        \
                     dim3(kFastThreads), 0u, st, e0, e1, n, rp.sorted.as<float4>(), rp.stype.as<int>(), rp.order.as<int>(), \
                     ctx->d.ntypes, ctx->tab.as<float2>(), rp.nlist.as<unsigned>(), rp.nneigh.as<int>(), rp.lg.maxn, c, f,  \
                     overwrite, ctx->escratch.as<double>(), rp.pub_ptr, rp.pub_val, rp.extent.as<int>(),                  \
                     rp.flags.as<int>(), lmode, F ? fl->fst : nullptr, fstep, rp.padgen.as<int>())
#define TMD_LAUNCH_FAST(L, F)               \
  if (lj && el) {                           \
    TMD_LAUNCH_FAST_T(L, true, true, F);    \
  } else if (lj) {                          \
    TMD_LAUNCH_FAST_T(L, true, false, F);   \
  } else {                                  \
    TMD_LAUNCH_FAST_T(L, false, true, F);   \
  }
  constexpr int wpb = kFastThreads / 64;
  const int npair8 = ((waves + wpb - 1) / wpb + 7) / 8 * 8;
  FusedStep fstep{};
  if (fl) {
    // step blocks behind the pair blocks (interior steps of tmdhip_md_run; fused_step_possible() has been asked):
    // one per 64 / (atoms of a pair block) pair blocks of an XCD's eighth
    fstep = fl->step;
    const int k = rp.lg.lpa * 64 / kFastThreads, g8 = npair8 / 8;
    const int units = (g8 + k - 1) / k;  // 64-atom units per XCD's eighth: a block with bonded records, a wave without
    fstep.nstep_blocks = 8 * (fstep.bonded == 1 ? units : (units + 3) / 4);
    if (rp.fsort.bytes < sizeof(float4) * (size_t)n) {
      TMD_TRY(rp.fsort.ensure(sizeof(float4) * (size_t)n));
      TMD_HIP(hipMemsetAsync(rp.fsort.p, 0, rp.fsort.bytes, st));  // launch number 0 = never written
      rp.fused_gen = 0;
    }
    if (rp.fused_gen == 0)  // test knob: start the launch counter just below its wrap-around
      if (const char *e = std::getenv("TMDHIP_DEBUG_FUSED_GEN0")) rp.fused_gen = (unsigned)std::strtoul(e, nullptr, 0);
    if (++rp.fused_gen == 0) rp.fused_gen = 1;  // (0 = "never written" in the records)
    fstep.gen = fstep.watch_gen = rp.fused_gen;
    fstep.poll_limit = 1u << 22;  // ~4 s of polling
    rp.fused_launches++;
    // test knob: the step blocks of this replica's k-th fused launch wait for a launch number nobody writes
    if (const char *e = std::getenv("TMDHIP_DEBUG_STEP_TIMEOUT"))
      if (rp.fused_launches == std::atoll(e)) fstep.watch_gen ^= 0x80000000u, fstep.poll_limit = 1u << 8;
    fstep.fsort = rp.fsort.as<float4>();
    {
      constexpr int kNve = ENERGY ? 3 : 1, kLangevin = ENERGY ? 4 : 2;  // (with energies: the final step of a call)
      constexpr int kEval = ENERGY ? 5 : 1;  // (a plain evaluation with energies; without ENERGY the branch is dead)
#define TMD_LAUNCH_FUSED(L)            \
  if (ENERGY && fl->eval_only) {       \
    TMD_LAUNCH_FAST(L, kEval);         \
  } else if (fl->langevin) {           \
    TMD_LAUNCH_FAST(L, kLangevin);     \
  } else {                             \
    TMD_LAUNCH_FAST(L, kNve);          \
  }
      switch (rp.lg.lpa) {
        case 8: TMD_LAUNCH_FUSED(8); break;
#ifndef TMD_DEV_LPA8_ONLY  // (developer builds: one lanes-per-atom variant compiles in a fifth of the time)
        case 4: TMD_LAUNCH_FUSED(4); break;
        case 16: TMD_LAUNCH_FUSED(16); break;
        case 32: TMD_LAUNCH_FUSED(32); break;
        case 64: TMD_LAUNCH_FUSED(64); break;
#endif
        default: return fail("fused MD step: unsupported lanes-per-atom");
      }
#undef TMD_LAUNCH_FUSED
    }
  } else {
    switch (rp.lg.lpa) {  // (pick_lpa never returns less than 4)
#ifndef TMD_DEV_LPA8_ONLY
      case 4: TMD_LAUNCH_FAST(4, 0); break;
      case 16: TMD_LAUNCH_FAST(16, 0); break;
      case 32: TMD_LAUNCH_FAST(32, 0); break;
      case 64: TMD_LAUNCH_FAST(64, 0); break;
#endif
      default: TMD_LAUNCH_FAST(8, 0); break;
    }
  }
#undef TMD_LAUNCH_FAST
#undef TMD_LAUNCH_FAST_T
#undef TMD_LAUNCH_FAST_S
  TMD_HIP(hipGetLastError());
  return 0;
}

template int launch_pair_fast_f32<true>(tmdhip_ctx *, Replica &, const PairConsts<float> &, float *, int, hipStream_t,
                                        hipEvent_t, hipEvent_t, int, const FusedLaunch *);
template int launch_pair_fast_f32<false>(tmdhip_ctx *, Replica &, const PairConsts<float> &, float *, int, hipStream_t,
                                         hipEvent_t, hipEvent_t, int, const FusedLaunch *);

}  // namespace tmd

#ifdef TMD_PAIR_TIMELINE
extern "C" int tmdhip_debug_pair_timeline(void *out, size_t bytes) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(tmd::g_pair_timeline), std::min(bytes, sizeof(tmd::g_pair_timeline))) == hipSuccess ? 0 : -1;
}
#endif
