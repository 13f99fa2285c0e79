// Micro-benchmark: cost of a vector-memory GATHER per wave instruction on gfx950 (per CU: the four SIMDs
// share one texture-addresser / L1), by access width and by how the 64 lane addresses are arranged.
// The array is 2 KB (L1 resident), so this is the pure address-processing / data-return rate.
//   hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v3u __attribute__((ext_vector_type(3)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
constexpr int ITERS = 512;

template <int W>
__global__ __launch_bounds__(256) void gather(const unsigned *data, const unsigned *offs, unsigned *out, int bytes) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(data), 0, bytes, 0x00020000);
  const int lane = threadIdx.x & 63;
  unsigned acc = 0;
  unsigned off[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) off[u] = offs[u * 64 + lane];
  for (int it = 0; it < ITERS; it += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (W == 1) acc += __builtin_amdgcn_raw_buffer_load_b32(r, off[u], 0, 0);
      if (W == 2) { v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, off[u], 0, 0); acc += v.x ^ v.y; }
      if (W == 3) { v3u v = __builtin_amdgcn_raw_buffer_load_b96(r, off[u], 0, 0); acc += v.x ^ v.y ^ v.z; }
      if (W == 4) { v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, off[u], 0, 0); acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(off[u]));  // opaque: the loads cannot be hoisted
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, blocks = cus * 8;  // 8 waves per SIMD
  const int natoms = 128;
  unsigned *data, *offs, *out;
  CHECK(hipMalloc(&data, 4096));
  CHECK(hipMemset(data, 1, 4096));
  CHECK(hipMalloc(&offs, 8 * 64 * 4));
  CHECK(hipMalloc(&out, blocks * 256 * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  struct Pat { const char *name; int mode; };
  Pat pats[] = {{"random atoms", 0}, {"runs of 2 consecutive atoms", 2}, {"runs of 4", 4}, {"runs of 8", 8}, {"fully consecutive", 64}, {"all lanes same atom", -1}};
  for (int stride : {16, 12}) {
    printf("== record stride %d bytes\n", stride);
    for (auto &p : pats) {
      std::vector<unsigned> h(8 * 64);
      srand(7);
      for (int u = 0; u < 8; ++u)
        for (int l = 0; l < 64; ++l) {
          int a;
          if (p.mode == 0) a = rand() % natoms;
          else if (p.mode == -1) a = 5;
          else { static int base; if (l % p.mode == 0) base = rand() % (natoms - p.mode); a = base + l % p.mode; }
          h[u * 64 + l] = a * stride;
        }
      CHECK(hipMemcpy(offs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
      printf("%-30s", p.name);
      for (int w = 1; w <= 4; ++w) {
        if (stride == 12 && w == 4) continue;
        auto launch = [&]() {
          if (w == 1) hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), 0, 0, data, offs, out, natoms * stride);
          if (w == 2) hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(256), 0, 0, data, offs, out, natoms * stride);
          if (w == 3) hipLaunchKernelGGL(gather<3>, dim3(blocks), dim3(256), 0, 0, data, offs, out, natoms * stride);
          if (w == 4) hipLaunchKernelGGL(gather<4>, dim3(blocks), dim3(256), 0, 0, data, offs, out, natoms * stride);
        };
        launch();
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
          CHECK(hipEventRecord(e0, 0));
          launch();
          CHECK(hipEventRecord(e1, 0));
          CHECK(hipEventSynchronize(e1));
          float ms;
          CHECK(hipEventElapsedTime(&ms, e0, e1));
          best = ms < best ? ms : best;
        }
        const double per_cu = (double)ITERS * 32;  // wave-instructions per CU (32 waves)
        printf("  x%d %6.1f cyc", w, best * 1e6 / per_cu * 2.4);
      }
      printf("   (cycles @2.4 GHz per wave-gather per CU)\n");
    }
  }
  return 0;
}
