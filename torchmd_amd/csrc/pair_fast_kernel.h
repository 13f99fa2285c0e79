// K3f: the lean fp32 list pair kernel of the nonbonded engine for gfx950 (MI355X) — the dominant kernel of the
// headline configuration — and the MD step inside its launch (step blocks).  Device code only: the launchers are
// pair_fast_f32.hip (one replica per launch) and pair_fast_f32_batch.hip (the replicas of a context in one launch).
//
// Reference semantics: torchmd/forces.py:260-319 restricted to LJ (381-415, with or without switching) and/or
// electrostatics (453-491, plain Coulomb or reaction field); the step blocks: torchmd/integrator.py:61-74.
#pragma once

#include "engine.h"
#include "md_step.h"

namespace tmd {

// ---- K3f: lean fp32 specialisation of the list pair kernel ---------------------------------------
// Issue-rate measurements on gfx950 (tools/ubench/valu_rates.hip, 8 waves per SIMD): a plain fp32 / integer
// VALU op (v_fma_f32, v_mul, v_add, v_and, v_mov, v_cndmask) retires in ~2.3 cycles per wave, a PACKED op
// (v_pk_fma/mul/add_f32) in ~4.3 — packing two entries into one instruction buys no ALU throughput on this
// chip — 32-bit shifts and v_mul_u32_u24 run at half rate (4.2), v_cmp costs 5.3, v_rsq/v_rcp 8.2.  The first
// version of this kernel evaluated entries two at a time on float2 vectors: 60 packed ops + 47 v_mov
// (transposes of {pj[u].x, pj[u+1].x} into register pairs) per 4 entries = ~155 cycles per entry.  This
// version is plain scalar code on the natural float4 record: ~32 full-rate ops + 1 v_cmp + 1 v_rsq per
// entry (~85 cycles), no transposes, no shifts:
//   entry = type << 27 | j << 4      -> gather offset = entry & 0x07FFFFF0 (one v_and), LDS table address
//                                       = (type_i << 8) | entry >> 24 (one SDWA v_or; table rows of 32 x 8 B)
//   minimum image by the magic-number trick (3 ops per component, bit-exact, see min_image_magic)
//   force scale factored as  rinv2 * ((a12 rinv6 + b6) rinv6 - qq rinv) + qq 2 krf   (9 ops)
// Same decision arithmetic (bit-exact) as pair_math.h.  Terms: LJ and/or electrostatics (plain Coulomb or
// reaction field), optionally the LJ switching function (SWITCH) and the per-term energies (ENERGY);
// repulsion terms, fp64, more than 32 LJ classes and pair counting take list_pair_kernel.

// What the loop below is shaped by (gfx950, tools/ubench/valu_detail.hip + body_bisect.hip, 6 waves per SIMD; cycles
// per wave-instruction per SIMD at 2.4 GHz):
//   plain fp32 / integer VALU with VGPR, inline-constant or 32-bit-literal operands     2.1 - 2.35
//   ANY SGPR operand (VOP2 src0, VOP3 src0/src2: v_fma/v_mul/v_sub/v_fmac)              4.05   <- half rate
//   v_cmp (VCC or SGPR pair) 4.1, v_cndmask with an SGPR/VCC mask 4.1, SDWA forms 4.1, v_mov_b64 4.1,
//   VOP3-only integer ops (v_perm, v_bfe, v_alignbit, v_and_or, v_lshl_or) 4.1
//   v_rsq_f32: 8.1 back to back, ~10 in bursts of four, ~19 when it stands alone among plain instructions
//   VGPR bank conflicts: none measurable (only three sources in ONE bank cost 4.1)
// The compiler keeps every uniform value (box, 1/box, r2max) in SGPRs — 24 of the 36 v_fma of a 4-entry group read
// one — rotates the prefetched list words with v_mov_b64, advances the list pointer with a 64-bit VALU add and puts
// the list load IN FRONT of the gathers, where every wait for a gather (vmcnt retires in order) also waits for the
// list stream from the Infinity Cache.  Hence: loop constants laundered into VGPRs; the cutoff test as arithmetic
// (v_fma with clamp + v_mul) where no energy is wanted; the four v_rsq of a group issued back to back; list words
// through a raw buffer with a SCALAR running offset, requested behind the gathers issued in the same breath; and the
// unchecked groups software-pipelined over two register sets (gathers of group g+1 in flight while g is evaluated).
// (Measured and not kept, round 4: the gathers of the pipelined loop landing in LDS — `buffer_load_dwordx4 ... lds`, two
// stages of 4 KB per wave, four waves per SIMD — 49.2-50.5 against 45.6-46.9 us, identical checksums; the code is in commit
// daee5f8, the record in profiles/r04_lds_gather_ab.txt.)
#ifndef TMD_FAST_WAVES_BASE  // (A/B builds, with TMD_BASE_ENTRIES_AT_ONCE)
#define TMD_FAST_WAVES_BASE 5
#endif
#ifndef TMD_BASE_ENTRIES_AT_ONCE
#define TMD_BASE_ENTRIES_AT_ONCE 4
#endif
constexpr int kFastWaves = 5;  // waves per SIMD of the pipelined loop (94 VGPRs); measured at 4 / 6 / 7 / 8: docs/history/round3.md
// Waves per SIMD a variant is compiled for.  Round 6: the variants with energies and / or the LJ switching function run the
// pipelined loop too — at five waves with their entries evaluated one at a time (kEntriesAtOnce, below), except the variant
// with the switch AND energies: four waves (120 VGPRs; at five it spills inside the loop).  C3, us per MD step /
// per compute() with energies (profiles/r06_variants_ab.txt): plain loop at five waves (round 5) 75.9 / 110.3 (switched),
// pipelined at five 78.7 / 168.5, pipelined at four 72.6 / 102.4 (unswitched compute(): 110.3 -> 102.4).
// (TMD_FAST_WAVES_ES / _E / _S: A/B builds)
#ifndef TMD_FAST_WAVES_E
#define TMD_FAST_WAVES_E 5  // (entries one at a time, TMD_E_ENTRIES_AT_ONCE)
#endif
#ifndef TMD_FAST_WAVES_S
#define TMD_FAST_WAVES_S 5  // (with its entries evaluated one at a time, TMD_S_ENTRIES_AT_ONCE; all four at once: 4 waves)
#endif
#ifndef TMD_S_ENTRIES_AT_ONCE
#define TMD_S_ENTRIES_AT_ONCE 1
#endif
#ifndef TMD_E_ENTRIES_AT_ONCE
#define TMD_E_ENTRIES_AT_ONCE 1
#endif
#ifndef TMD_FAST_WAVES_ES
#define TMD_FAST_WAVES_ES 4
#endif
#ifndef TMD_LJ_WAVES  // (LJ-only systems: the plain loop; 10^6 argon atoms at 6 / 7 / 8 waves: 139.8 / 136.2-138.2 / 137.7-138.3 us/step)
#define TMD_LJ_WAVES (kFastWaves + 2)
#endif
#ifndef TMD_LJ_ENTRIES_AT_ONCE
#define TMD_LJ_ENTRIES_AT_ONCE 4
#endif
#ifndef TMD_BATCH_WAVES  // (the batched kernel, below; A/B builds: 4 waves for every variant 67.8 us/step at C3, this 61.x)
#define TMD_BATCH_WAVES fast_waves(ELEC, ENERGY, SWITCH)
#endif
constexpr int fast_waves(bool elec, bool energy, bool sw) {
  if (!elec) return TMD_LJ_WAVES;
  return energy && sw ? TMD_FAST_WAVES_ES : energy ? TMD_FAST_WAVES_E : sw ? TMD_FAST_WAVES_S : TMD_FAST_WAVES_BASE;
}
// ---- the MD step inside the pair launch (FUSED variants; tmdhip_md_run, interior steps) -----------------------
// Between two force evaluations an MD step is per-atom work on the force just computed: second half kick of step
// `it` (+ thermostat), first half kick and drift of step it+1, the displacement test, the new record of the
// cell-sorted copy.  As a kernel of its own that is 8.8 us at C3 (22 us at 10^6 atoms): a chain of memory round
// trips (order -> bonded records -> partner positions -> update) with one wave per SIMD and nothing to hide it behind.
// A FUSED launch appends "step blocks" to the grid.  Workgroups are dispatched in order, so a step block starts when
// every pair block has been dispatched — in the slots the launch's last, partial round of pair blocks leaves idle —
// and does everything that does not need the new forces (bonded records of its 64 atoms, noise, loads) while the
// last pair blocks are still gathering; then every lane waits for the force record of ITS atom (the pair wave stores
// {force, launch number} as one 16-byte word) and updates.  Behind the last pair block only one load-update-store
// round remains, and the stored force array, its reload and one launch per step go away.
// Pair blocks never wait for anything, so the wait cannot deadlock; it is bounded all the same.
// Other blocks still read the positions of this launch, so the new ones go to the OTHER position buffer and the
// OTHER cell-sorted copy (the host swaps the two after every fused launch).  Same device functions in the same
// order as md_step_bonded_kernel / md_step_kernel: trajectories are bit-identical to the separate kernels.
#ifdef TMD_PAIR_TIMELINE  // experiment builds only (tools/pair_timeline.py; nothing of it is in the product library): {entry, exit on
                          // the device-wide 100 MHz clock, XCC id, core cycles} of every pair block
static __device__ unsigned long long g_pair_timeline[4 * 65536];
#endif
// The body of the kernel: block `bid` of `nblocks` (pair blocks first, then the step blocks of a FUSED launch) of ONE replica's
// launch.  list_pair_fast_f32_kernel passes blockIdx.x / gridDim.x and its own arguments; the replica-batched kernel
// (list_pair_fast_f32_batch_kernel, round 6) the block's position inside its replica's share of the grid and that replica's
// buffers from a device table.
template <int LPA, bool LJ, bool ELEC, bool ENERGY, bool SWITCH, int FUSED, bool TABLE = false>
__device__ __forceinline__ void pair_fast_body(
    const unsigned bid, const unsigned nblocks,
    int n, const float4 *__restrict__ sorted, const int *__restrict__ stype, const int *__restrict__ order,
    int ntypes, const float2 *__restrict__ tab, const unsigned *__restrict__ nlist,
    const int *__restrict__ nneigh, int maxn, const PairConsts<float> &c, float *__restrict__ forces, int overwrite,
    double *__restrict__ energies, unsigned *publish, unsigned publish_value, const int *__restrict__ ext,
    int *lflags, int lmode, const FusedStatic *__restrict__ fst, const FusedStep &fstep, int *__restrict__ padgen) {
  constexpr int APW = 64 / LPA;
  constexpr int UNROLL = 4;
  constexpr bool kPipelined = ELEC;  // the software-pipelined loop over the unchecked groups (below)
  // Entries of a group (one list word of a lane = 4 entries) evaluated at once.  Four for most variants: distances of all four,
  // four v_rsq back to back, then the four force evaluations.  The SWITCH variant without energies takes them ONE at a time: the
  // compiler then needs 92 VGPRs instead of 110 and the variant runs at five waves per SIMD without scratch (same arithmetic in
  // the same order: results unchanged bit for bit).  C3 with the switch, same box: 71.2-71.6 -> 69.1-69.2 us/step, the launch
  // 52.8-53.5 -> 50.8-51.1 us; two at a time at five waves (8 bytes of scratch outside the loop) 70.3-70.7.  The variants
  // The variant with energies (no switch) likewise: 96 VGPRs, two loop-invariant values parked in scratch around the loop (8-12
  // bytes, none inside it); FINAL launch of a call 54.2 -> 52.6 us, compute() with energies 82.8 -> 81.5.  Switch AND energies
  // keeps four entries and four waves: one at a time at five waves it spills 44 bytes inside the loop;
  // the headline variant one at a time fits six waves with 24 bytes parked around the loop, and is slower there (same box:
  // 65.7-66.0 against 63.5-63.8 us/step): five waves, four at once, as since round 3.
  constexpr int kEntriesAtOnce = (ELEC && SWITCH && !ENERGY) ? TMD_S_ENTRIES_AT_ONCE : (ELEC && ENERGY && !SWITCH) ? TMD_E_ENTRIES_AT_ONCE : (!ELEC && !ENERGY && !SWITCH) ? TMD_LJ_ENTRIES_AT_ONCE : (ELEC && !ENERGY && !SWITCH) ? TMD_BASE_ENTRIES_AT_ONCE : UNROLL;
  static_assert(kEntriesAtOnce == 1 || kEntriesAtOnce == 2 || kEntriesAtOnce == 4, "");
#ifdef TMD_PAIR_TIMELINE
  const unsigned long long tl_t0 = wall_clock64(), tl_c0 = __builtin_readcyclecounter();
#endif
  // FUSED 1 / 2: interior steps (NVE / Langevin step blocks); 3 / 4: the LAST step of a call that wants energies (FINAL
  // step blocks, md_step.h: second half kick + bonded energies + kinetic energy + the complete force); 5: a plain evaluation
  // with energies (tmdhip_compute; step blocks that add the bonded force and leave the bonded energies: no velocities)
  static_assert(FUSED == 0 || (FUSED <= 2 && !ENERGY) || (FUSED >= 3 && ENERGY), "interior steps carry no energies, the final step does");
  static_assert(kFastThreads == 256, "step blocks are four waves");
  if (bid == 0 && threadIdx.x == 0) {
    // tells the host (host-mapped word) that everything enqueued before this launch has completed
    if (publish) __hip_atomic_store(publish, publish_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (lflags) {
      const int parity = (lmode & kLmParity) ? 1 : 0;
      if ((lmode & kLmViolation) && lflags[F_REBUILD0 + parity] != 0) lflags[F_VIOLATION] = 1;
      // the epilogue's test (parity ^ 1) is the next step's: this step's request is history (list_check_clear)
      if (FUSED == 1 || FUSED == 2) lflags[F_REBUILD0 + parity] = 0;
    }
  }
  __shared__ __align__(16) float2 stab[kEntryTypes * kEntryTypes];  // row of type i: 32 x {-12 A, 6 B}
  const int lane = threadIdx.x & 63;
  // pair blocks of the launch (FUSED: step blocks follow them)
  const unsigned npair = FUSED ? nblocks - (unsigned)fstep.nstep_blocks : nblocks;
  if (FUSED && bid >= npair) {
    fused_step_blocks<float, FUSED == 2 || FUSED == 4, kFastThreads / LPA, (FUSED == 5 ? 2 : FUSED >= 3 ? 1 : 0), TABLE>(
        fst, fstep, c, n, sorted, order, (int)(bid - npair), (int)npair, reinterpret_cast<float *>(stab), forces, energies);
    return;
  }
  // XCD-aware block order: consecutive block ids go to the 8 XCDs round-robin, so block b works on
  // chunk (b % 8) * npair/8 + b / 8 — every XCD (own L2) gets a contiguous eighth of the cell-sorted
  // atoms and gathers neighbours from that region only.  npair is a multiple of 8; the surplus
  // blocks of the last eighths have nothing to do.
  const int blk = (int)((bid & 7u) * (npair >> 3) + (bid >> 3));
  if (blk * (int)(blockDim.x >> 6) * APW >= n) return;  // (block-uniform: nobody is left waiting at the barrier below)
  const int wave = __builtin_amdgcn_readfirstlane(blk * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6));
  const int a = wave * APW + lane / LPA;
  const int sub = lane % LPA;
  const bool active = a < n;

  // ---- prologue: every load a wave needs before its first gather is requested HERE, in one batch, and only then
  // is the LJ table staged (a wave's life used to begin with three dependent memory round trips — table, then
  // atom record / list length, then the first list word — 5 500 of its ~40 000 cycles)
  // list words of this wave: group G (iterations 4G .. 4G+3 of all 64 lanes) is the 1 KB at byte G * 1024; rows are
  // padded, and reads past the buffer's end return 0
  const unsigned *wrow = nlist + (size_t)wave * maxn * APW;
  const __amdgpu_buffer_rsrc_t lrsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(wrow), 0, maxn * APW * 4 + 4096, 0x00020000);
  const unsigned lvoff = (unsigned)lane * 16u;
  auto list_word_raw = [&](int g) { return __builtin_amdgcn_raw_buffer_load_b128(lrsrc, lvoff, g * 1024, 0); };
  v4u word = list_word_raw(0);  // list word of the next group to be gathered (in flight)
  float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
  int nn = 0, oi = 0;
  unsigned trow = 0;  // byte offset of this atom's row of the LDS table
  if (active) {
    pi = sorted[a];
    nn = nneigh[a];
    trow = (unsigned)stype[a] << 8;
    oi = order[a];
  }
  // (only the rows of existing classes are ever read: ntypes x 32 entries instead of 32 x 32 — at 10^6 LJ atoms
  // with 64 atoms per block the full table was 15 625 x 8 KB of staging)
  for (int t = threadIdx.x; t < ntypes * kEntryTypes; t += blockDim.x) {
    const int ti = t >> 5, tj = t & 31;
    float2 ab = make_float2(0.f, 0.f);
    if (tj < ntypes) ab = tab[ti * ntypes + tj];
    stab[t] = make_float2(-12.0f * ab.x, 6.0f * ab.y);
  }
  __syncthreads();

  const int myiters = (nn - sub + LPA - 1) / LPA;  // entries kk < myiters are real for this lane
  int itmax = myiters, itmin = myiters;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    itmax = max(itmax, __shfl_xor(itmax, o, 64));
    itmin = min(itmin, __shfl_xor(itmin, o, 64));
  }
  const int nkk = __builtin_amdgcn_readfirstlane(itmax);
  // iterations every lane has entries for.  The unchecked loop takes the table offset as `entry >> 24`, which needs
  // the slot's bits 20..22 to be zero: systems of more than 2^20 atoms run all their iterations in the checked
  // loop, which masks the offset.
  // Padded rows (kLmPadded): the slots between a lane's last entry and the end of the wave's last group hold a harmless
  // entry (a dummy record out of reach, pad_entry_for), so EVERY group is unchecked.  The padding is written by the wave
  // itself on its first launch after a list build: its group's `padgen` word then differs from the rebuild count; that
  // launch still runs its tail checked (the entries it has just stored are for the launches that follow).
  bool padded = false;
  if (lmode & kLmPadded) {
    const int now = __builtin_amdgcn_readfirstlane(lflags[F_NREBUILD]);
    const int have = __builtin_amdgcn_readfirstlane(padgen[wave]);
    padded = have == now;
    if (!padded) {
      const unsigned pad = pad_entry_for(c, n, pi.x, pi.y, pi.z);
      unsigned *row = const_cast<unsigned *>(wrow);
      const int upto = (nkk + UNROLL - 1) / UNROLL * UNROLL;
      for (int kk = myiters; kk < upto; ++kk) row[(((kk >> 2) << 6) + lane) * 4 + (kk & 3)] = pad;
      if (lane == 0) padgen[wave] = now;
    }
  }
  const int nfull = padded ? (nkk + UNROLL - 1) / UNROLL * UNROLL
                           : (n > (1 << 20) ? 0 : __builtin_amdgcn_readfirstlane(itmin) / UNROLL * UNROLL);
  // bounds-checked raw buffer over sorted_xyzq: lanes past the end of their list read whatever the
  // (uninitialised) padding entry points at — out-of-range offsets return 0 instead of faulting — and
  // are discarded by `valid`
  const __amdgpu_buffer_rsrc_t srsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(sorted), 0, (n + 2) * 16, 0x00020000);  // (+ the two dummy records)
  const char *tbase = reinterpret_cast<const char *>(stab);
  const float two_krf = 2.0f * c.krf;
  const float qi2k = pi.w * two_krf;
  auto in_vgpr = [](float sv) {  // a uniform value the compiler can no longer keep in an SGPR
    float v;
    asm("v_mov_b32 %0, %1" : "=v"(v) : "s"(sv));
    return v;
  };
  // LJ switching function (forces.py:399-413) with t = (r - r_s)/(r_c - r_s) clamped to [0, 1]:
  //   S = 1 + t^3 (-10 + t (15 - 6 t)),   S' (r_c - r_s) = t^2 (-30 + 60 t - 30 t^2) = -30 (t (1 - t))^2
  // and with p = (-12 A r^-6 + 6 B) r^-6 (the unswitched force coefficient times r^2) and e12 = p + 6 B r^-6 = -12 E_lj:
  //   (dE/dr)/r = r^-2 [ S p + e12 w c x ],  w = (t (1 - t))^2, c = 2.5 / (r_c - r_s), x = 1 (the reference's explicit
  //   force divides the switching term by r once more, forces.py:410-412) or r (exact: -dE/dr)
  // ~14 plain VALU per entry on top of the unswitched body, no select; constants in VGPRs (see the head comment).
  const float sw_ir = in_vgpr(c.inv_switch_range), sw_t0 = in_vgpr(-c.switch_dist * c.inv_switch_range);
  const float sw_m0 = in_vgpr(c.switch_reference_mode ? 2.5f * c.inv_switch_range : 0.f);
  const float sw_m1 = in_vgpr(c.switch_reference_mode ? 0.f : 2.5f * c.inv_switch_range);
  const float vkrf = in_vgpr(c.krf), vcrf = in_vgpr(c.crf);
  const float vbx = in_vgpr(c.box[0]), vby = in_vgpr(c.box[1]), vbz = in_vgpr(c.box[2]);
  const float vibx = in_vgpr(c.invbox[0]), viby = in_vgpr(c.invbox[1]), vibz = in_vgpr(c.invbox[2]);
  const float vr2max = in_vgpr(c.r2max);
  // cutoff test as arithmetic: step = clamp((r2max' - r2) * 2^100, 0, 1) with r2max' the successor of r2max is exactly
  // 1 for r2 <= r2max and 0 beyond
  const float cut_h = in_vgpr(-1.2676506e30f);  // -2^100
  const float cut_c0 = in_vgpr(__int_as_float(__float_as_int(c.r2max) + 1) * 1.2676506e30f);

  float fx = 0.f, fy = 0.f, fz = 0.f;
  // per-lane fp32 partial sums (~55 pairs), reduced in fp64; e_lj in units of -12 E_lj (e12 above)
  float e_lj = 0.f, e_el = 0.f;

  using checked_t = std::integral_constant<bool, false>;
  using unchecked_t = std::integral_constant<bool, true>;
  // one group = this lane's 4 entries of iterations kk0 .. kk0+3 (one 16-byte list word) and their 4 gathered records;
  // tab[u] = byte offset of entry u's {-12 A, 6 B} in the LDS table (row of type i | 8 x type j)
  auto group = [&](auto image, auto unchecked, const auto &tab, const auto &raw, int kk0) {
    constexpr bool EXACT = decltype(image)::value;
    constexpr bool UNCHECKED = decltype(unchecked)::value;
    // (unchecked groups hold real pairs and dummy records only: every value below is finite and the cutoff can be a factor;
    // the padding words of a checked group are garbage, inf / NaN are discarded by selects)
    constexpr bool ARITH_CUT = UNCHECKED;
    constexpr int NU = (int)std::extent<std::remove_reference_t<decltype(tab)>>::value;
    static_assert(NU == 4, "a stage is one whole list word of a lane");
    float dx[NU], dy[NU], dz[NU], r2[NU], rinv[NU];
    constexpr int HB = kEntriesAtOnce;  // (see kEntriesAtOnce above)
#pragma unroll
    for (int h0 = 0; h0 < NU; h0 += HB) {
#pragma unroll
    for (int u = h0; u < h0 + HB; ++u) {
      dx[u] = min_image_magic<EXACT>(pi.x - __uint_as_float(raw[u].x), vbx, vibx);
      dy[u] = min_image_magic<EXACT>(pi.y - __uint_as_float(raw[u].y), vby, viby);
      dz[u] = min_image_magic<EXACT>(pi.z - __uint_as_float(raw[u].z), vbz, vibz);
      r2[u] = norm2(dx[u], dy[u], dz[u]);
    }
    if constexpr (HB == NU) {
      asm("v_rsq_f32 %0, %4\n\tv_rsq_f32 %1, %5\n\tv_rsq_f32 %2, %6\n\tv_rsq_f32 %3, %7"
          : "=&v"(rinv[0]), "=&v"(rinv[1]), "=&v"(rinv[2]), "=&v"(rinv[3])
          : "v"(r2[0]), "v"(r2[1]), "v"(r2[2]), "v"(r2[3]));
    } else {
      if constexpr (HB == 2) asm("v_rsq_f32 %0, %2\n\tv_rsq_f32 %1, %3" : "=&v"(rinv[h0]), "=&v"(rinv[h0 + 1]) : "v"(r2[h0]), "v"(r2[h0 + 1]));
      else asm("v_rsq_f32 %0, %1" : "=v"(rinv[h0]) : "v"(r2[h0]));
    }
#pragma unroll
    for (int u = h0; u < h0 + HB; ++u) {
      const bool valid = UNCHECKED || (kk0 + u < myiters);  // padding words are garbage
      const float pjw = __uint_as_float(raw[u].w);
      const bool hit = valid && (r2[u] <= vr2max);
      float step = 1.f;
      if (ARITH_CUT) asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(step) : "v"(r2[u]), "v"(cut_h), "v"(cut_c0));
      const float rinv2 = rinv[u] * rinv[u];
      const float rinv6 = rinv2 * rinv2 * rinv2;
      float2 ab = make_float2(0.f, 0.f);  // (-12 A, 6 B)
      if (LJ) ab = *reinterpret_cast<const float2 *>(tbase + tab[u]);
      // LJ: P = r^2 (dE_lj/dr)/r, e12 = -12 E_lj (both switched where SWITCH)
      float P = 0.f, e12 = 0.f;
      if (LJ) {
        P = __builtin_fmaf(ab.x, rinv6, ab.y) * rinv6;  // (a12 rinv6 + b6) rinv6
        if (ENERGY || SWITCH) e12 = __builtin_fmaf(ab.y, rinv6, P);
        if (SWITCH) {
          const float r = r2[u] * rinv[u];
          float t;
          asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(t) : "v"(r), "v"(sw_ir), "v"(sw_t0));
          const float t2 = t * t;
          const float pp = __builtin_fmaf(t, __builtin_fmaf(t, -6.f, 15.f), -10.f);
          const float sw = __builtin_fmaf(t2 * t, pp, 1.f);
          const float tu = __builtin_fmaf(-t, t, t);  // t (1 - t)
          const float w = tu * tu;
          const float xc = __builtin_fmaf(r, sw_m1, sw_m0);
          P = __builtin_fmaf(e12, w * xc, sw * P);
          if (ENERGY) e12 *= sw;
        }
      }
      float fs;  // (dE/dr) / r; rejected entries of a checked group may produce inf/NaN here, the select below discards them
      if (ELEC) {
        const float qq = pi.w * pjw;
        const float g = __builtin_fmaf(-qq, rinv[u], P);
        fs = __builtin_fmaf(rinv2, g, qi2k * pjw);
        if (ENERGY) {  // krf = crf = 0: plain Coulomb
          const float eel = qq * __builtin_fmaf(vkrf, r2[u], rinv[u] - vcrf);
          e_el = ARITH_CUT ? __builtin_fmaf(step, eel, e_el) : e_el + (hit ? eel : 0.f);
        }
      } else {
        fs = P * rinv2;
      }
      if (ENERGY && LJ) e_lj = ARITH_CUT ? __builtin_fmaf(step, e12, e_lj) : e_lj + (hit ? e12 : 0.f);
      if (ARITH_CUT) fs *= step;
      else fs = hit ? fs : 0.f;
      fx = __builtin_fmaf(-dx[u], fs, fx);
      fy = __builtin_fmaf(-dy[u], fs, fy);
      fz = __builtin_fmaf(-dz[u], fs, fz);
    }
    if (HB < NU) __builtin_amdgcn_sched_barrier(0);  // (one after the other: interleaved they need the registers again)
    }
  };

  static_assert(UNROLL == 4, "one dwordx4 of list per lane and group");
  // issue the 4 gathers of the group whose list word is `w` and form its table offsets (unchecked: n <= 2^20, bits
  // 24..27 of an entry are zero; checked: padding words are garbage, the offset is masked)
  auto issue = [&](auto unchecked, const v4u &w, v4u (&raw)[UNROLL], unsigned (&tab)[UNROLL]) {
    const unsigned entry[UNROLL] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) raw[u] = __builtin_amdgcn_raw_buffer_load_b128(srsrc, entry[u] & kEntryOffMask, 0, 0);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) tab[u] = trow | (decltype(unchecked)::value ? entry[u] >> 24 : (entry[u] >> 24) & 0xF8u);
  };
  const int gall = (nkk + UNROLL - 1) / UNROLL;  // groups of this wave
  int g = 0;                                     // next group to evaluate; `word` = its list word
  // (a request past the wave's last group — the look-ahead of its last iterations — would stream 1 KB of padding per
  // wave for nothing: 64 MB of the 560 MB a 10^6-atom LJ launch moves, 12 MB of 222 at C3; scalar branch, wave-uniform)
  // Lists that do not fit the 256 MiB Infinity Cache anyway (kLmStream: 10^6 LJ atoms, large water boxes) are streamed
  // with the non-temporal hint, so that they do not displace the position records from the L2s: 10^6-atom LJ launch
  // 138 -> 127 us.  (At C3, whose 162 MB list lives in the Infinity Cache, the hint costs 35 %: hence a run-time choice;
  // wave-uniform scalar branch.)  Not fetching the words of a group that hold nothing but padding was tried too
  // (offset out of range for those lanes, padding entry formed in the kernel): no gain, removed.
  const bool stream_list = lmode & kLmStream;
  auto list_word = [&](int gg) {
    v4u w = (v4u){0u, 0u, 0u, 0u};
    if (gg < gall) {
      if (stream_list) w = __builtin_amdgcn_raw_buffer_load_b128(lrsrc, lvoff, gg * 1024, 2 /* nt */);
      else w = __builtin_amdgcn_raw_buffer_load_b128(lrsrc, lvoff, gg * 1024, 0);
    }
    return w;
  };
  // A list word is requested AFTER the gathers issued in the same breath (see the head comment); sched_barrier pins
  // that order against the compiler's preference.
  auto checked_loop = [&](auto image) {  // per-lane validity; not pipelined (the tail is short)
    for (; g < gall; ++g) {
      v4u raw[UNROLL];
      unsigned tab[UNROLL];
      issue(checked_t{}, word, raw, tab);
      __builtin_amdgcn_sched_barrier(0);
      word = list_word(g + 1);
      __builtin_amdgcn_sched_barrier(0);
      group(image, checked_t{}, tab, raw, g * UNROLL);
    }
  };
  if (extent_needs_exact_image(ext, c.box)) {  // wave-uniform, rare: atoms more than 2.4 box edges apart
    checked_loop(exact_image{});
  } else {
    const int gfull = nfull / UNROLL;  // groups in which every lane has real entries: no validity test
    // Software pipeline over the unchecked groups: the gathers of group g+1 (and the list word of g+2) are requested
    // before group g is evaluated, into the other register set; a wave then waits for memory once per group, for
    // requests it made a whole group's arithmetic earlier (counters of the unpipelined loop: 44 % of a wave's cycles
    // in s_waitcnt, 27 % issuing — at the ~5 cycles per instruction a wave can issue by itself, six such waves do
    // not fill the VALU pipe).  94 VGPRs: five waves per SIMD.
    if constexpr (!kPipelined) {  // (LJ-only systems: short lists, the plain loop at two waves more per SIMD)
      for (; g < gfull; ++g) {
        v4u raw[UNROLL];
        unsigned tab[UNROLL];
        issue(unchecked_t{}, word, raw, tab);
        __builtin_amdgcn_sched_barrier(0);
        word = list_word(g + 1);
        __builtin_amdgcn_sched_barrier(0);
        group(fused_image{}, unchecked_t{}, tab, raw, g * UNROLL);
      }
    } else if (gfull > 0) {
      v4u ra[UNROLL], rb[UNROLL];
      unsigned ta[UNROLL], tb[UNROLL];
      issue(unchecked_t{}, word, ra, ta);
      __builtin_amdgcn_sched_barrier(0);
      word = list_word(1);
      __builtin_amdgcn_sched_barrier(0);
      while (true) {
        if (g + 1 >= gfull) {
          group(fused_image{}, unchecked_t{}, ta, ra, g * UNROLL);
          g += 1;
          break;
        }
        issue(unchecked_t{}, word, rb, tb);
        __builtin_amdgcn_sched_barrier(0);
        word = list_word(g + 2);
        __builtin_amdgcn_sched_barrier(0);
        group(fused_image{}, unchecked_t{}, ta, ra, g * UNROLL);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 2 >= gfull) {
          group(fused_image{}, unchecked_t{}, tb, rb, (g + 1) * UNROLL);
          g += 2;
          break;
        }
        issue(unchecked_t{}, word, ra, ta);
        __builtin_amdgcn_sched_barrier(0);
        word = list_word(g + 3);
        __builtin_amdgcn_sched_barrier(0);
        group(fused_image{}, unchecked_t{}, tb, rb, (g + 1) * UNROLL);
        __builtin_amdgcn_sched_barrier(0);
        g += 2;
      }
    }
    checked_loop(fused_image{});  // tail
  }
#ifdef TMD_PAIR_TIMELINE
  if (threadIdx.x == 0 && blockIdx.x < 65536u) {
    g_pair_timeline[4 * blockIdx.x + 0] = tl_t0;
    g_pair_timeline[4 * blockIdx.x + 1] = wall_clock64();
    g_pair_timeline[4 * blockIdx.x + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((6 << 11) | 20) | ((unsigned long long)gridDim.x << 32);
    g_pair_timeline[4 * blockIdx.x + 3] = __builtin_readcyclecounter() - tl_c0;
  }
#endif
  float sx = fx, sy = fy, sz = fz;
#pragma unroll
  for (int o = LPA >> 1; o > 0; o >>= 1) {
    sx += __shfl_xor(sx, o, 64);
    sy += __shfl_xor(sy, o, 64);
    sz += __shfl_xor(sz, o, 64);
  }
  if constexpr (FUSED != 0) {
    // The force record {fx, fy, fz, launch number} goes to the cell-sorted array the step blocks watch, as ONE 16-byte
    // store written through to device scope (sc1): the number in .w says the force beside it is this launch's.
    // (A flag per wave behind the stores cost a memory round trip more at the end of the launch; an agent-scope
    // release does it with buffer_wbl2, a write-back of the whole L2 per wave: 365 us per launch.)
    if (active && sub == 0) store_force_record(fstep.fsort, n, a, sx, sy, sz, fstep.gen);
    if constexpr (!ENERGY) return;  // (the final launch of a call also leaves its pair energies, below; its step blocks the force)
  }
  if (FUSED == 0 && active && sub == 0 && forces) {
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
      const double s = wave_sum((double)e_lj);
      if (lane == 0 && s != 0.0) unsafeAtomicAdd(&energy_row(energies)[TMDHIP_E_LJ], (-0.5 / 12.0) * s);  // (e_lj holds -12 E_lj)
    }
    if (ELEC) {
      const double s = wave_sum((double)e_el);
      if (lane == 0 && s != 0.0) unsafeAtomicAdd(&energy_row(energies)[TMDHIP_E_ELECTROSTATICS], 0.5 * s);
    }
  }
}

template <int LPA, bool LJ, bool ELEC, bool ENERGY, bool SWITCH, int FUSED = 0>
// (LJ-only systems — liquid argon, short lists of ~90 entries — run the plain loop at more waves per SIMD: six, 10^6 atoms
// 175.5 -> 168.5 us/step in round 3; seven since round 6, 139.8 -> 136-138; with charges the pipelined loop at 5 waves wins,
// docs/history/round3.md)
__global__ __launch_bounds__(kFastThreads, fast_waves(ELEC, ENERGY, SWITCH)) void list_pair_fast_f32_kernel(
    int n, const float4 *__restrict__ sorted, const int *__restrict__ stype, const int *__restrict__ order,
    int ntypes, const float2 *__restrict__ tab, const unsigned *__restrict__ nlist,
    const int *__restrict__ nneigh, int maxn, PairConsts<float> c, float *__restrict__ forces, int overwrite,
    double *__restrict__ energies, unsigned *publish, unsigned publish_value, const int *__restrict__ ext,
    int *lflags, int lmode, const FusedStatic *__restrict__ fst, FusedStep fstep, int *__restrict__ padgen) {
  pair_fast_body<LPA, LJ, ELEC, ENERGY, SWITCH, FUSED>(blockIdx.x, gridDim.x, n, sorted, stype, order, ntypes, tab, nlist, nneigh, maxn, c,
                                                       forces, overwrite, energies, publish, publish_value, ext, lflags, lmode, fst,
                                                       fstep, padgen);
}

// ---- replicas of a cell-list context in ONE launch (round 6) ----------------------------------------------------------------
// The reference's batch axis is the replica (systems.py:6-18, forces.py:105,116).  Mid-size boxes are latency-bound launches
// (12 288 atoms: 18.7 us per step on a GPU that takes 98 304 atoms in 45), so eight replicas one after the other cost eight
// launches.  Here the grid holds the pair blocks of ALL replicas of the batch, then the step blocks of all of them; a block
// finds its replica from its index and that replica's buffers in a device table (BatchRep, uploaded when a pointer changes);
// what changes from launch to launch travels as a kernel argument (BatchLaunch).  Every replica keeps its own neighbour state,
// rebuild flags and host reports: the block of a replica does exactly what it does in a launch of its own — results are
// bit-identical to the replica-by-replica loop.
// Waves per SIMD as the plain kernel's variants.  (First version: four waves for every variant, because the five-wave build of
// the headline variant spilled 88 bytes per lane — the step blocks read the replica's FusedStatic, a pointer out of the table,
// with flat loads into VGPRs.  Read through the constant address space (md_step.h: load_uniform) the fields are in SGPRs as
// in the plain kernel: 96 VGPRs, no scratch; C3 as ONE replica through this kernel 67.8 -> 61.x us/step, the plain kernel's.)
template <int LPA, bool LJ, bool ELEC, bool ENERGY, bool SWITCH, int FUSED>
__global__ __launch_bounds__(kFastThreads, TMD_BATCH_WAVES) void list_pair_fast_f32_batch_kernel(
    int n, int ntypes, const float2 *__restrict__ tab, PairConsts<float> c, const BatchRep *__restrict__ reps, BatchLaunch bl) {
  static_assert(FUSED != 0, "the batched launch is an MD-step launch");
  const unsigned per_pair = (unsigned)bl.pair_blocks, per_step = (unsigned)bl.step_blocks;
  const unsigned all_pair = per_pair * (unsigned)bl.nrep;
  // (pair blocks of every replica first: the step blocks are dispatched behind ALL of them, into the launch's last round)
  const bool is_pair = blockIdx.x < all_pair;
  const unsigned q = is_pair ? blockIdx.x : blockIdx.x - all_pair, per = is_pair ? per_pair : per_step;
  const unsigned r = q / per, local = q - r * per;
  const BatchRep &S = reps[r];  // (block-uniform: scalar loads)
  const unsigned bits = bl.bits[r];
  const int cs = (bits & kBlSortedCur) ? 1 : 0, cp = (bits & kBlPosCur) ? 1 : 0;
  for (int k = 0; k < 3; ++k) {
    c.box[k] = S.box[k];
    c.invbox[k] = S.invbox[k];
  }
  FusedStep fs{};
  fs.pos_in = S.pos[cp];
  fs.pos_out = S.pos[cp ^ 1];
  fs.sorted_out = S.sorted[cs ^ 1];
  fs.fsort = S.fsort;
  fs.gen = fs.watch_gen = bl.gen[r];
  fs.poll_limit = bl.poll_limit;
  fs.bonded = bl.bonded;
  fs.nstep_blocks = (int)per_step;
  fs.noise_step = bl.noise_step;
  fs.seq = bl.seq[r];
  fs.near_host = (bits & kBlReports) ? S.hostpub + 1 + (fs.seq & 1u) : nullptr;
  fs.parity = (bits & kBlNextParity) ? 1 : 0;
  // (TABLE: the step blocks read the replica's FusedStatic through the constant address space, md_step.h: load_uniform)
  pair_fast_body<LPA, LJ, ELEC, ENERGY, SWITCH, FUSED, true>(
      is_pair ? local : per_pair + local, per_pair + per_step, n, S.sorted[cs], S.stype, S.order, ntypes, tab, S.nlist, S.nneigh, S.maxn, c,
      S.forces, 1, S.escratch, (bits & kBlPublish) ? S.hostpub : nullptr, bl.pub[r], S.ext, S.lflags, (int)(bits & kBlLmodeMask), S.fst, fs,
      S.padgen);
}


}  // namespace tmd
