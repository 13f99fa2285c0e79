// Micro-benchmark: where do the cycles of the pair-kernel loop body go?  ("the 30 -> 38 us step of
// the builder's ladder is not explained".)  The scalar body of list_pair_fast_f32_kernel (formulation A of
// pair_body.hip, positions and LJ table in LDS so that the texture path is out of the picture) is compiled with
// pieces switched off by a template mask, at the real kernel's shape (12 288 waves x 56 entries per lane, blocks of
// 256) and at 2 / 4 / 6 / 8 resident waves per SIMD (dynamic LDS limits the occupancy):
//   bit 0  j record read from LDS (ds_read_b128)        else: made up from the entry word in registers
//   bit 1  LJ table read from LDS (ds_read_b64)         else: constants
//   bit 2  list words loaded from global memory         else: a register rotated per iteration
//   bit 3  minimum image (9 VALU per entry)
//   bit 4  v_rsq_f32                                    else: one v_mul
//   bit 5  cutoff compare + select (v_cmp -> SGPR pair, v_cndmask)
// Prints us per launch; together with the VALU count of each variant's loop (tools/isa_stats.py on the -S output of
// this file) that gives cycles per VALU instruction per SIMD for every mix.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize body_bisect.hip -o body_bisect && ./body_bisect
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));

struct Consts { float bx, by, bz, ibx, iby, ibz, r2max, two_krf; };

template <bool ON>
__device__ __forceinline__ float mi(float d, float box, float invbox) {
#pragma clang fp contract(off)
  if (!ON) return d;
  const float magic = 12582912.0f;
  const float t = __builtin_fmaf(d, invbox, magic);
  const float k = t - magic;
  return __builtin_fmaf(-k, box, d);
}

template <int MASK>
__global__ __launch_bounds__(256) void body(const float4 *pos, const unsigned *list, const float2 *tab, float *out,
                                            Consts c, int nkk) {
  constexpr bool GATHER = MASK & 1, TABLE = MASK & 2, LIST = MASK & 4, IMAGE = MASK & 8, RSQ = MASK & 16, CUT = MASK & 32;
  __shared__ float2 stab[1024];
  __shared__ float4 spos[128];
  extern __shared__ char occupancy_pad[];
  if (threadIdx.x < 128) spos[threadIdx.x] = pos[threadIdx.x];
  for (int t = threadIdx.x; t < 1024; t += 256) stab[t] = tab[t];
  __syncthreads();
  const char *pbase = (const char *)spos;
  const char *tbase = (const char *)stab;
  const int lane = threadIdx.x & 63;
  const float4 pi = pos[(blockIdx.x * 4 + (threadIdx.x >> 6)) & 127];
  const unsigned trow = (lane & 1) << 8;
  const v4u *row4 = (const v4u *)list + lane;
  const float qi2k = pi.w * c.two_krf;
  float fx = 0, fy = 0, fz = 0;
  v4u nxa = row4[0], nxb = row4[64];
  for (int kk0 = 0; kk0 < nkk; kk0 += 4) {
    const v4u cur = nxa;
    nxa = nxb;
    if (LIST) nxb = row4[(size_t)(((kk0 >> 2) + 2) & 3) * 64];
    else nxb = cur + (v4u){16u, 32u, 48u, 64u};
    const unsigned e[4] = {cur.x, cur.y, cur.z, cur.w};
    v4u raw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (GATHER) raw[u] = *(const v4u *)(pbase + (e[u] & 0x7F0u));
      else raw[u] = (v4u){(e[u] & 0x7F0u) | 0x41000000u, (e[u] & 0x3F0u) | 0x41100000u, (e[u] & 0x5F0u) | 0x41200000u, 0x3ECCCCCDu};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float dx = mi<IMAGE>(pi.x - __uint_as_float(raw[u].x), c.bx, c.ibx);
      const float dy = mi<IMAGE>(pi.y - __uint_as_float(raw[u].y), c.by, c.iby);
      const float dz = mi<IMAGE>(pi.z - __uint_as_float(raw[u].z), c.bz, c.ibz);
      const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
      const float rinv = RSQ ? __frsqrt_rn(r2) : r2 * 0.01f;
      const float rinv2 = rinv * rinv;
      const float rinv6 = rinv2 * rinv2 * rinv2;
      float2 ab = make_float2(-12.f * 5e5f, 6.f * 600.f);
      if (TABLE) ab = *(const float2 *)(tbase + (trow | (e[u] >> 24)));
      const float pjw = __uint_as_float(raw[u].w);
      const float qq = pi.w * pjw;
      const float p = __builtin_fmaf(ab.x, rinv6, ab.y) * rinv6;
      const float g = __builtin_fmaf(-qq, rinv, p);
      float fs = __builtin_fmaf(rinv2, g, qi2k * pjw);
      if (CUT) fs = (r2 <= c.r2max) ? fs : 0.f;
      fx = __builtin_fmaf(-dx, fs, fx);
      fy = __builtin_fmaf(-dy, fs, fy);
      fz = __builtin_fmaf(-dz, fs, fz);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = fx + fy + fz;
}

typedef void (*kern_t)(const float4 *, const unsigned *, const float2 *, float *, Consts, int);
struct K { const char *name; kern_t fn; };

int main() {
  const int natoms = 128, waves = 12288, blocks = waves / 4;
  std::vector<float4> pos(natoms);
  std::vector<float2> tab(1024);
  srand(1);
  for (auto &p : pos) p = make_float4(rand() % 1000 * 0.03f, rand() % 1000 * 0.03f, rand() % 1000 * 0.03f, 0.4f);
  for (auto &t : tab) t = make_float2(-12.f * 5e5f, 6.f * 600.f);
  std::vector<unsigned> list(6 * 64 * 4);
  for (auto &e : list) e = ((unsigned)(rand() % natoms) << 4) | ((unsigned)(rand() % 2) << 27);
  float4 *dpos;
  float2 *dtab;
  unsigned *dlist;
  float *dout;
  CHECK(hipMalloc(&dpos, sizeof(float4) * natoms));
  CHECK(hipMalloc(&dtab, sizeof(float2) * 1024));
  CHECK(hipMalloc(&dlist, sizeof(unsigned) * list.size()));
  CHECK(hipMalloc(&dout, sizeof(float) * blocks * 256));
  CHECK(hipMemcpy(dpos, pos.data(), sizeof(float4) * natoms, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dtab, tab.data(), sizeof(float2) * 1024, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dlist, list.data(), sizeof(unsigned) * list.size(), hipMemcpyHostToDevice));
  Consts c{30.f, 30.f, 30.f, 1 / 30.f, 1 / 30.f, 1 / 30.f, 81.f, 0.001f};
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  K ks[] = {
      {"63 full body", body<63>},
      {"62 - j record from registers", body<62>},
      {"61 - table constants", body<61>},
      {"59 - no list loads", body<59>},
      {"56 - no memory op at all", body<56>},
      {"55 - no minimum image", body<55>},
      {"47 - no rsq", body<47>},
      {"31 - no compare/select", body<31>},
      {"48 - VALU only, no image", body<48>},
      {"40 - VALU only, no rsq", body<40>},
      {"24 - VALU only, no compare", body<24>},
      {" 0 - plain fma/mul chain", body<0>},
  };
  // dynamic LDS per block so that only `wps` blocks (= waves per SIMD) fit a CU's 160 KB (static: 10 KB)
  const int wps_list[] = {8, 6, 4, 2};
  for (int wps : wps_list) {
    const unsigned dyn = wps == 8 ? 0u : (unsigned)(160 * 1024 / wps - 10 * 1024 - 512);
    printf("--- %d waves per SIMD (dynamic LDS %u B) ---\n", wps, dyn);
    for (auto &k : ks) {
      CHECK(hipFuncSetAttribute((const void *)k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 10 * 1024 - 512));
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), dyn, 0, dpos, dlist, dtab, dout, c, 56);
      CHECK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 8; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), dyn, 0, dpos, dlist, dtab, dout, c, 56);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      printf("%-34s %7.2f us\n", k.name, best * 1e3);
    }
  }
  return 0;
}
