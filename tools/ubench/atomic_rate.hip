// Micro-benchmark (round 5, gate (i) of the "evaluate every pair once" plan): what do the scattered fp32 atomic adds of
// a HALF neighbour list cost on gfx950 at the real access pattern of the C3 water box?
//
// The reference scatters +-f of every pair once (torchmd/forces.py:316-319, index_add_); the engine's full list stores
// each atom's force without atomics but evaluates every pair from both sides.  A half list halves the evaluated slots and
// pays three float atomics per slot for the j side.  This program measures that price alone and next to a gather:
//   98 304 atoms in cell order; atom i lists ~205 partners j > "half" of its neighbourhood, arranged like the engine's
//   lists: 25 stencil rows = runs of ~70 consecutive cell-sorted records at offsets of up to +-2 cell layers
//   (+-10 300 records) from i, ~23 % of a run within reach; 8 lanes per atom, entries dealt round-robin to the lanes;
//   blocks mapped to the 8 XCDs like the pair kernel (each XCD a contiguous eighth of the atoms).
// Variants:  none      the loop without atomics (list stream + optional gather): the baseline to subtract
//            agent     unsafeAtomicAdd / __hip_atomic_fetch_add(relaxed, agent) on float[3N]      (3 atomics per slot)
//            agent4    the same on float4 records (16-byte stride: one cache line per atom more often)
//            wg        workgroup scope (the instruction without sc1: executes in the XCD's own L2 — only correct with one
//                      force copy per XCD, measured for the rate)
//            cluster   the cluster-pair pattern of the verdict: per 8x8 tile ONE wave-atomic with 24 active lanes on the 24
//                      consecutive floats of an 8-atom j cluster (1.4e7 adds per step)
//   hipcc --offload-arch=gfx950 -O3 atomic_rate.hip -o atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int LPA = 8, APW = 64 / LPA;

enum { M_NONE = 0, M_AGENT = 1, M_AGENT4 = 2, M_WG = 3 };

template <int MODE, bool GATHER>
__global__ __launch_bounds__(256) void half_list_kernel(int n, int maxk, const int *__restrict__ nlist, const int *__restrict__ nneigh,
                                                        const float4 *__restrict__ pos, float *__restrict__ f, float *__restrict__ sink) {
  const unsigned npair = gridDim.x;
  const int blk = (int)((blockIdx.x & 7u) * (npair >> 3) + (blockIdx.x >> 3));  // XCD-aware order of the pair kernel
  const int wave = blk * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int a = wave * APW + lane / LPA, sub = lane % LPA;
  if (wave * APW >= n) return;
  const float4 pi = a < n ? pos[a] : make_float4(0, 0, 0, 0);
  const int nn = a < n ? nneigh[a] : 0;
  int itmax = (nn - sub + LPA - 1) / LPA;
  for (int o = 32; o > 0; o >>= 1) itmax = max(itmax, __shfl_xor(itmax, o, 64));
  const int *row = nlist + (size_t)wave * maxk * 64;
  float fx = 0, fy = 0, fz = 0;
  for (int kk = 0; kk < itmax; ++kk) {
    const int j = row[kk * 64 + lane];  // -1: padding
    if (j < 0) continue;
    float dx = 1e-3f, dy = 2e-3f, dz = 3e-3f;
    if (GATHER) {
      const float4 pj = pos[j];
      dx = pi.x - pj.x, dy = pi.y - pj.y, dz = pi.z - pj.z;
    }
    const float r2 = dx * dx + dy * dy + dz * dz + 1.0f;
    const float s = 1.0f / r2;
    fx += dx * s, fy += dy * s, fz += dz * s;
    if (MODE == M_AGENT) {
      unsafeAtomicAdd(&f[3 * j + 0], -dx * s);
      unsafeAtomicAdd(&f[3 * j + 1], -dy * s);
      unsafeAtomicAdd(&f[3 * j + 2], -dz * s);
    } else if (MODE == M_AGENT4) {
      unsafeAtomicAdd(&f[4 * j + 0], -dx * s);
      unsafeAtomicAdd(&f[4 * j + 1], -dy * s);
      unsafeAtomicAdd(&f[4 * j + 2], -dz * s);
    } else if (MODE == M_WG) {
      __hip_atomic_fetch_add(&f[3 * j + 0], -dx * s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&f[3 * j + 1], -dy * s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&f[3 * j + 2], -dz * s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  for (int o = LPA >> 1; o > 0; o >>= 1) {
    fx += __shfl_xor(fx, o, 64);
    fy += __shfl_xor(fy, o, 64);
    fz += __shfl_xor(fz, o, 64);
  }
  if (a < n && sub == 0) {
    if (MODE == M_NONE) sink[a] = fx + fy + fz;
    else if (MODE == M_AGENT4) {
      unsafeAtomicAdd(&f[4 * a + 0], fx), unsafeAtomicAdd(&f[4 * a + 1], fy), unsafeAtomicAdd(&f[4 * a + 2], fz);
    } else {
      unsafeAtomicAdd(&f[3 * a + 0], fx), unsafeAtomicAdd(&f[3 * a + 1], fy), unsafeAtomicAdd(&f[3 * a + 2], fz);
    }
  }
}

// cluster pattern: wave = one i cluster (8 atoms), walks its list of j clusters; per tile ONE wave-atomic, 24 active lanes
// on the 24 consecutive floats of the j cluster (after a stand-in for the 8-lane reduction)
template <bool ATOMIC>
__global__ __launch_bounds__(256) void cluster_kernel(int nclusters, int maxt, const int *__restrict__ clist, const int *__restrict__ ntile,
                                                      float *__restrict__ f, float *__restrict__ sink) {
  const unsigned npair = gridDim.x;
  const int blk = (int)((blockIdx.x & 7u) * (npair >> 3) + (blockIdx.x >> 3));
  const int ci = blk * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ci >= nclusters) return;
  const int nt = ntile[ci];
  const int *row = clist + (size_t)ci * maxt;
  float acc = 0.f;
  for (int t = 0; t < nt; ++t) {
    const int cj = __builtin_amdgcn_readfirstlane(row[t]);
    float v = (float)(lane + 1) * 1e-3f + acc * 1e-6f;
    v += __shfl_xor(v, 8, 64);  // stand-in for the reduction over the 8 i lanes (3 steps)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    acc += v;
    if (ATOMIC && lane < 24) unsafeAtomicAdd(&f[24 * cj + lane], v);
  }
  if (lane == 0) sink[ci] = acc;
}

int main(int argc, char **argv) {
  const int n = 98304;
  const int per_atom = argc > 1 ? atoi(argv[1]) : 205;  // half-list entries per atom (C3: 4.06e7 / 98 304 / 2)
  std::mt19937 rng(5);
  // --- the half list: 25 rows (dx layer in -2..2, dy row in -2..2), z-run of 70 records centred on i + offset ---
  const int layer = 19 * 19 * 14, rowlen = 19 * 14;  // records per x layer / per y row of cells (19^3 cells of ~14)
  const int maxk = ((per_atom * 5 / 4 / LPA + 3) / 4 * 4 + 4);  // iterations per lane (capacity)
  const int waves = (n + APW - 1) / APW;
  std::vector<int> nlist((size_t)waves * maxk * 64, -1), nneigh(n, 0);
  long long total = 0;
  std::vector<int> mine;
  for (int i = 0; i < n; ++i) {
    mine.clear();
    for (int ox = -2; ox <= 2; ++ox)
      for (int oy = -2; oy <= 2; ++oy) {
        const long long base = (long long)i + (long long)ox * layer + (long long)oy * rowlen - 35;
        for (int z = 0; z < 70; ++z) {
          if ((rng() & 0xFF) >= 59 * 2 * per_atom / 410) continue;  // ~23 % of a run is within reach (x2: both halves)
          long long j = ((base + z) % n + n) % n;
          if (j == i) continue;
          // "checkerboard" half: keep the pair at the lower index when (i + j) is odd, at the higher when even
          const bool keep = (((i + j) & 1) != 0) == (i < j);
          if (keep) mine.push_back((int)j);
        }
      }
    int cnt = (int)mine.size();
    if (cnt > maxk * LPA) cnt = maxk * LPA;
    nneigh[i] = cnt;
    total += cnt;
    const int w = i / APW, ain = i % APW;
    for (int k = 0; k < cnt; ++k) nlist[((size_t)w * maxk + k / LPA) * 64 + ain * LPA + k % LPA] = mine[k];
  }
  printf("half list: %d atoms, %lld entries (%.1f per atom), capacity %d per lane, %lld atomics per pass\n", n, total,
         (double)total / n, maxk, 3 * total);
  int *d_list, *d_nn;
  float4 *d_pos;
  float *d_f, *d_sink;
  CHECK(hipMalloc(&d_list, nlist.size() * 4));
  CHECK(hipMemcpy(d_list, nlist.data(), nlist.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&d_nn, n * 4));
  CHECK(hipMemcpy(d_nn, nneigh.data(), n * 4, hipMemcpyHostToDevice));
  std::vector<float4> hp(n);
  for (int i = 0; i < n; ++i) hp[i] = make_float4((rng() & 1023) * 0.1f, (rng() & 1023) * 0.1f, (rng() & 1023) * 0.1f, 0.4f);
  CHECK(hipMalloc(&d_pos, n * 16));
  CHECK(hipMemcpy(d_pos, hp.data(), n * 16, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&d_f, (size_t)n * 16 + 4096));
  CHECK(hipMemset(d_f, 0, (size_t)n * 16 + 4096));
  CHECK(hipMalloc(&d_sink, n * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int blocks = ((waves + 3) / 4 + 7) / 8 * 8;
  auto time_it = [&](const char *name, auto launch, double adds) {
    launch();
    CHECK(hipDeviceSynchronize());
    float best = 1e30f, sum = 0;
    const int reps = 20;
    for (int rep = 0; rep < reps; ++rep) {
      CHECK(hipEventRecord(e0, 0));
      launch();
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
      sum += ms;
    }
    printf("%-44s best %8.2f us  mean %8.2f us", name, best * 1e3, sum / reps * 1e3);
    if (adds > 0) printf("   (%.2e adds -> %.1f Gadd/s at best)", adds, adds / (best * 1e-3) * 1e-9);
    printf("\n");
    return best * 1e3f;
  };
#define RUN(MODE, G, label) time_it(label, [&] { hipLaunchKernelGGL((half_list_kernel<MODE, G>), dim3(blocks), dim3(256), 0, 0, n, maxk, d_list, d_nn, d_pos, d_f, d_sink); }, MODE == M_NONE ? 0.0 : 3.0 * total)
  printf("== per-atom half list, 8 lanes per atom, no gather (list stream + atomics only)\n");
  const float b0 = RUN(M_NONE, false, "no atomics");
  const float a0 = RUN(M_AGENT, false, "agent scope, float[3N]");
  const float a1 = RUN(M_AGENT4, false, "agent scope, float4 records");
  const float a2 = RUN(M_WG, false, "workgroup scope, float[3N]");
  printf("== the same with the 16-byte position gather per entry\n");
  const float b1 = RUN(M_NONE, true, "no atomics");
  const float g0 = RUN(M_AGENT, true, "agent scope, float[3N]");
  const float g1 = RUN(M_AGENT4, true, "agent scope, float4 records");
  const float g2 = RUN(M_WG, true, "workgroup scope, float[3N]");
  printf("price of the atomics (us): no gather  agent %.1f  agent4 %.1f  wg %.1f | with gather  agent %.1f  agent4 %.1f  wg %.1f\n", a0 - b0,
         a1 - b0, a2 - b0, g0 - b1, g1 - b1, g2 - b1);

  // --- cluster pattern: 12 288 i clusters x T tiles, one 24-lane wave-atomic per tile -> 1.4e7 adds at T = 48 ---
  const int nc = n / 8;
  for (int T : {48, 96}) {
    std::vector<int> cl((size_t)nc * T), nt(nc, T);
    for (int c = 0; c < nc; ++c)
      for (int t = 0; t < T; ++t) {
        const int ox = (int)(rng() % 5) - 2, oy = (int)(rng() % 5) - 2, oz = (int)(rng() % 9) - 4;
        long long cj = (long long)c + (long long)ox * (layer / 8) + (long long)oy * (rowlen / 8) + oz;
        cl[(size_t)c * T + t] = (int)(((cj % nc) + nc) % nc);
      }
    int *d_cl, *d_nt;
    CHECK(hipMalloc(&d_cl, cl.size() * 4));
    CHECK(hipMemcpy(d_cl, cl.data(), cl.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_nt, nc * 4));
    CHECK(hipMemcpy(d_nt, nt.data(), nc * 4, hipMemcpyHostToDevice));
    const int cblocks = ((nc + 3) / 4 + 7) / 8 * 8;
    char label[96];
    printf("== cluster pattern: %d i clusters x %d tiles, 24 consecutive floats per tile\n", nc, T);
    snprintf(label, sizeof label, "tiles without atomics");
    const float c0 = time_it(label, [&] { hipLaunchKernelGGL(cluster_kernel<false>, dim3(cblocks), dim3(256), 0, 0, nc, T, d_cl, d_nt, d_f, d_sink); }, 0.0);
    snprintf(label, sizeof label, "one 24-lane wave-atomic per tile");
    const float c1 = time_it(label, [&] { hipLaunchKernelGGL(cluster_kernel<true>, dim3(cblocks), dim3(256), 0, 0, nc, T, d_cl, d_nt, d_f, d_sink); }, 24.0 * nc * T);
    printf("price of the atomics: %.1f us for %.2e adds\n", c1 - c0, 24.0 * nc * T);
    CHECK(hipFree(d_cl));
    CHECK(hipFree(d_nt));
  }
  return 0;
}
