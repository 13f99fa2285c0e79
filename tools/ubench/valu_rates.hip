// Micro-benchmark: issue cost of the VALU instructions the pair kernel is made of, on gfx950.
// Each kernel runs LOOPS iterations of 64 independent-ish instructions per wave (8 accumulators, so
// dependent-issue latency is covered by ILP 8 and by the other waves of the SIMD).  Reported: ns per
// wave-instruction per SIMD and the equivalent cycles at the measured shader clock (wall_clock64 = 100 MHz
// constant; the shader clock is estimated from the v_mov kernel = known 1 issue / quad... so everything
// is also given RELATIVE to v_fma_f32).
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int LOOPS = 2000;

#define R8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)

// ---- scalar fma -------------------------------------------------------------------------------------
__global__ void k_fma(float *out, float b, float c) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  for (int i = 0; i < LOOPS; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
      asm volatile(
          "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(b), "v"(c));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

#define PK_KERNEL(NAME, INSTR)                                                                                       \
  __global__ void NAME(float *out, float bb, float cc) {                                                             \
    v2f a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,    \
        a6 = a0 + 6.f, a7 = a0 + 7.f;                                                                                \
    v2f b = {bb, bb}, c = {cc, cc};                                                                                  \
    for (int i = 0; i < LOOPS; ++i) {                                                                                \
      _Pragma("unroll") for (int r = 0; r < 8; ++r) asm volatile(                                                    \
          INSTR " %0, %0, %8, %9\n " INSTR " %1, %1, %8, %9\n " INSTR " %2, %2, %8, %9\n " INSTR " %3, %3, %8, %9\n" \
          INSTR " %4, %4, %8, %9\n " INSTR " %5, %5, %8, %9\n " INSTR " %6, %6, %8, %9\n " INSTR " %7, %7, %8, %9\n" \
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                          \
          : "v"(b), "v"(c));                                                                                         \
    }                                                                                                                \
    v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;                                                          \
  }
PK_KERNEL(k_pk_fma, "v_pk_fma_f32")

#define PK2_KERNEL(NAME, INSTR)                                                                                      \
  __global__ void NAME(float *out, float bb, float cc) {                                                             \
    v2f a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,    \
        a6 = a0 + 6.f, a7 = a0 + 7.f;                                                                                \
    v2f b = {bb, bb};                                                                                                \
    for (int i = 0; i < LOOPS; ++i) {                                                                                \
      _Pragma("unroll") for (int r = 0; r < 8; ++r) asm volatile(                                                    \
          INSTR " %0, %0, %8\n " INSTR " %1, %1, %8\n " INSTR " %2, %2, %8\n " INSTR " %3, %3, %8\n"                 \
          INSTR " %4, %4, %8\n " INSTR " %5, %5, %8\n " INSTR " %6, %6, %8\n " INSTR " %7, %7, %8\n"                 \
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                          \
          : "v"(b));                                                                                                 \
    }                                                                                                                \
    v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + cc;                                                     \
  }
PK2_KERNEL(k_pk_mul, "v_pk_mul_f32")
PK2_KERNEL(k_pk_add, "v_pk_add_f32")

// two-operand scalar instructions  d = op(d, b)
#define S2_KERNEL(NAME, INSTR)                                                                                       \
  __global__ void NAME(float *out, float b, float cc) {                                                              \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    for (int i = 0; i < LOOPS; ++i) {                                                                                \
      _Pragma("unroll") for (int r = 0; r < 8; ++r) asm volatile(                                                    \
          INSTR " %0, %0, %8\n " INSTR " %1, %1, %8\n " INSTR " %2, %2, %8\n " INSTR " %3, %3, %8\n"                 \
          INSTR " %4, %4, %8\n " INSTR " %5, %5, %8\n " INSTR " %6, %6, %8\n " INSTR " %7, %7, %8\n"                 \
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                          \
          : "v"(b));                                                                                                 \
    }                                                                                                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + cc;                         \
  }
S2_KERNEL(k_mul, "v_mul_f32")
S2_KERNEL(k_add, "v_add_f32")
S2_KERNEL(k_and, "v_and_b32")
S2_KERNEL(k_mulu24, "v_mul_u32_u24")
S2_KERNEL(k_lshl, "v_lshlrev_b32")

// one-operand  d = op(d)
#define S1_KERNEL(NAME, INSTR)                                                                                       \
  __global__ void NAME(float *out, float b, float cc) {                                                              \
    float a0 = threadIdx.x + 1.f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    for (int i = 0; i < LOOPS; ++i) {                                                                                \
      _Pragma("unroll") for (int r = 0; r < 8; ++r) asm volatile(                                                    \
          INSTR " %0, %0\n " INSTR " %1, %1\n " INSTR " %2, %2\n " INSTR " %3, %3\n"                                 \
          INSTR " %4, %4\n " INSTR " %5, %5\n " INSTR " %6, %6\n " INSTR " %7, %7\n"                                 \
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));                        \
    }                                                                                                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b + cc;                     \
  }
S1_KERNEL(k_mov, "v_mov_b32")
S1_KERNEL(k_rsq, "v_rsq_f32")
S1_KERNEL(k_rcp, "v_rcp_f32")
S1_KERNEL(k_sqrt, "v_sqrt_f32")

// compare into vcc + cndmask (a pair of instructions)
__global__ void k_cmp_cnd(float *out, float b, float cc) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  for (int i = 0; i < LOOPS; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
      asm volatile(
          "v_cmp_le_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_le_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc\n"
          "v_cmp_le_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_cmp_le_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %4, vcc\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
          : "v"(b)
          : "vcc");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + cc;
}

// compare only (into an SGPR pair)
__global__ void k_cmp(float *out, float b, float cc) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  unsigned long long m = 0;
  for (int i = 0; i < LOOPS; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      unsigned long long t0, t1, t2, t3, t4, t5, t6, t7;
      asm volatile(
          "v_cmp_le_f32 %0, %8, %12\n v_cmp_le_f32 %1, %9, %12\n v_cmp_le_f32 %2, %10, %12\n v_cmp_le_f32 %3, %11, %12\n"
          "v_cmp_lt_f32 %4, %8, %12\n v_cmp_lt_f32 %5, %9, %12\n v_cmp_lt_f32 %6, %10, %12\n v_cmp_lt_f32 %7, %11, %12\n"
          : "=s"(t0), "=s"(t1), "=s"(t2), "=s"(t3), "=s"(t4), "=s"(t5), "=s"(t6), "=s"(t7)
          : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b));
      m ^= t0 ^ t1 ^ t2 ^ t3 ^ t4 ^ t5 ^ t6 ^ t7;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(m & 0xff) + cc;
}

// mixes: 4 pk_fma + 4 of something else per 8 instructions
#define MIX_KERNEL(NAME, OTHER)                                                                                      \
  __global__ void NAME(float *out, float bb, float cc) {                                                             \
    v2f a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;                                  \
    float s0 = threadIdx.x + 1.f, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;                                             \
    v2f b = {bb, bb}, c = {cc, cc};                                                                                  \
    for (int i = 0; i < LOOPS; ++i) {                                                                                \
      _Pragma("unroll") for (int r = 0; r < 8; ++r) asm volatile(                                                    \
          "v_pk_fma_f32 %0, %0, %8, %9\n " OTHER " %4, %4\n v_pk_fma_f32 %1, %1, %8, %9\n " OTHER " %5, %5\n"       \
          "v_pk_fma_f32 %2, %2, %8, %9\n " OTHER " %6, %6\n v_pk_fma_f32 %3, %3, %8, %9\n " OTHER " %7, %7\n"       \
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)                          \
          : "v"(b), "v"(c));                                                                                         \
    }                                                                                                                \
    v2f s = a0 + a1 + a2 + a3;                                                                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s0 + s1 + s2 + s3;                                      \
  }
MIX_KERNEL(k_mix_pk_mov, "v_mov_b32")
MIX_KERNEL(k_mix_pk_rsq, "v_rsq_f32")

// pk_fma with op_sel broadcast of the low half of src1 (what a "scalar x pair" product needs)
__global__ void k_pk_fma_opsel(float *out, float bb, float cc) {
  v2f a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f,
      a7 = a0 + 7.f;
  v2f b = {bb, bb + 1.f}, c = {cc, cc};
  for (int i = 0; i < LOOPS; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
      asm volatile(
          "v_pk_fma_f32 %0, %0, %8, %9 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %8, %9 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %2, %2, %8, %9 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %8, %9 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %4, %4, %8, %9 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n v_pk_fma_f32 %5, %5, %8, %9 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
          "v_pk_fma_f32 %6, %6, %8, %9 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n v_pk_fma_f32 %7, %7, %8, %9 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "v"(b), "v"(c));
  }
  v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// pk_fma with SGPR-pair operands (constants such as box / invbox)
__global__ void k_pk_fma_sgpr(float *out, float bb, float cc) {
  v2f a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f,
      a7 = a0 + 7.f;
  v2f b = {bb, bb}, c = {cc, cc};
  for (int i = 0; i < LOOPS; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
      asm volatile(
          "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
          "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
          : "s"(b), "v"(c));
  }
  v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

struct Case {
  const char *name;
  void (*fn)(float *, float, float);
  double instr_per_loop;  // wave-instructions per loop iteration
};

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
  float *out;
  CHECK(hipMalloc(&out, sizeof(float) * cus * 32 * 64));
  std::vector<Case> cases = {
      {"v_fma_f32", k_fma, 64},        {"v_pk_fma_f32", k_pk_fma, 64},   {"v_pk_mul_f32", k_pk_mul, 64},
      {"v_pk_add_f32", k_pk_add, 64},  {"v_mul_f32", k_mul, 64},         {"v_add_f32", k_add, 64},
      {"v_and_b32", k_and, 64},        {"v_mul_u32_u24", k_mulu24, 64},  {"v_lshlrev_b32", k_lshl, 64},
      {"v_mov_b32", k_mov, 64},        {"v_rsq_f32", k_rsq, 64},         {"v_rcp_f32", k_rcp, 64},
      {"v_sqrt_f32", k_sqrt, 64},      {"cmp+cndmask", k_cmp_cnd, 64},   {"v_cmp->sgpr", k_cmp, 64},
      {"pk_fma+mov 1:1", k_mix_pk_mov, 64}, {"pk_fma+rsq 1:1", k_mix_pk_rsq, 64},
      {"pk_fma op_sel", k_pk_fma_opsel, 64}, {"pk_fma sgpr src", k_pk_fma_sgpr, 64},
  };
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int wps : {1, 2, 4, 8}) {  // waves per SIMD
    printf("--- %d wave(s) per SIMD (grid = %d blocks of 256 threads) ---\n", wps, cus * wps);
    for (auto &c : cases) {
      const int blocks = cus * wps;  // one 256-thread block = 4 waves = one per SIMD
      hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
      CHECK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      const double instr_per_simd = (double)LOOPS * c.instr_per_loop * 8 /*r loop*/ / 8 * wps;  // 8 r-iterations x 8 instr = 64
      const double ns = best * 1e6 / instr_per_simd;
      printf("%-18s %8.3f ms   %6.3f ns / wave-instr / SIMD   = %5.2f cycles @2.4GHz\n", c.name, best, ns, ns * 2.4);
    }
  }
  return 0;
}
