// Launcher of the replica-batched form of the lean fp32 list pair kernel (pair_fast_kernel.h: list_pair_fast_f32_batch_kernel):
// the pair blocks and the step blocks of up to kBatchMax replicas of a cell-list context in ONE launch.  A translation unit of
// its own so that its kernel variants compile beside those of pair_fast_f32.hip, not behind them.
//
// Reference semantics: the replica loop of torchmd/forces.py:105,116 over torchmd/systems.py:6-18's [R, N, 3] tensors.
#include "pair_fast_kernel.h"

namespace tmd {

// grid geometry of one replica's share of a fused launch (the same numbers launch_pair_fast_f32 uses for a launch of its own)
void fused_grid_shape(const tmdhip_ctx *ctx, const Replica &rp, int bonded, int &pair_blocks, int &step_blocks) {
  const int n = ctx->d.natoms, apw = rp.lg.apw, waves = (n + apw - 1) / apw;
  constexpr int wpb = kFastThreads / 64;
  pair_blocks = ((waves + wpb - 1) / wpb + 7) / 8 * 8;
  const int k = rp.lg.lpa * 64 / kFastThreads, g8 = pair_blocks / 8;
  const int units = (g8 + k - 1) / k;
  step_blocks = 8 * (bonded == 1 ? units : (units + 3) / 4);
}

int launch_pair_fast_f32_batch(tmdhip_ctx *ctx, int rep0, const PairConsts<float> &c, const BatchLaunch &bl, int lpa, bool energy,
                               bool langevin, hipStream_t st) {
  const bool lj = c.terms & TMDHIP_TERM_LJ, el = c.terms & TMDHIP_TERM_ELECTROSTATICS;
  const dim3 grid((unsigned)bl.nrep * (unsigned)(bl.pair_blocks + bl.step_blocks)), block(kFastThreads);
  const BatchRep *reps = ctx->batch_tab.as<BatchRep>() + rep0;
#define TMD_B_S(L, A, B, E, S, F)                                                                                          \
  hipLaunchKernelGGL((list_pair_fast_f32_batch_kernel<L, A, B, E, S, F>), grid, block, 0, st, ctx->d.natoms, ctx->d.ntypes, \
                     ctx->tab.as<float2>(), c, reps, bl)
#define TMD_B_T(L, A, B, E, F)        \
  if (c.switch_on && A) {            \
    TMD_B_S(L, A, B, E, true, F);    \
  } else {                           \
    TMD_B_S(L, A, B, E, false, F);   \
  }
#define TMD_B_TERMS(L, E, F)          \
  if (lj && el) {                     \
    TMD_B_T(L, true, true, E, F);     \
  } else if (lj) {                    \
    TMD_B_T(L, true, false, E, F);    \
  } else {                            \
    TMD_B_T(L, false, true, E, F);    \
  }
#define TMD_B_MODE(L)                 \
  if (energy) {                       \
    if (langevin) {                   \
      TMD_B_TERMS(L, true, 4);        \
    } else {                          \
      TMD_B_TERMS(L, true, 3);        \
    }                                 \
  } else {                            \
    if (langevin) {                   \
      TMD_B_TERMS(L, false, 2);       \
    } else {                          \
      TMD_B_TERMS(L, false, 1);       \
    }                                 \
  }
  switch (lpa) {
    case 8: TMD_B_MODE(8); break;
#ifndef TMD_DEV_LPA8_ONLY
    case 4: TMD_B_MODE(4); break;
    case 16: TMD_B_MODE(16); break;
    case 32: TMD_B_MODE(32); break;
    case 64: TMD_B_MODE(64); break;
#endif
    default: return fail("batched pair + step launch: unsupported lanes-per-atom");
  }
#undef TMD_B_MODE
#undef TMD_B_TERMS
#undef TMD_B_T
#undef TMD_B_S
  TMD_HIP(hipGetLastError());
  ctx->batched_launches++;
  return 0;
}

}  // namespace tmd
