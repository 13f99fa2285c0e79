// Library-independent sanity check of a GPU box (round 5, tools/soak.py): tells a bad box from a race in libtmdhip.
// Nothing of the engine is used — plain HIP only.  Known answers:
//   (1) fp32 atomics: every lane of 4 096 waves (all 8 XCDs) adds 1.0f to slot (global id mod 1024) of a float array and
//       0.5f to a slot chosen by a hash; the sums are small integers / half-integers, exact in fp32 in any order.
//   (2) copies: a 64 MiB pattern (word k = hash(k, round)) is written by a kernel, copied device-to-device twice
//       (hipMemcpyAsync) and verified by a kernel; then read back and verified on the host (sampled).
//   (3) ALU: every lane runs a chain of 4 096 dependent fp32 FMAs + integer mixes; the result must equal the host's (fp32
//       FMA is exactly rounded on both sides) for all 2^20 lanes.
//   (4) LDS: a block-wide transpose through shared memory with a known permutation.
// Prints one line per check and "SANITY OK" / "SANITY FAILED"; exit code 0 / 1.
//   hipcc --offload-arch=gfx950 -O3 sanity.hip -o sanity        ./sanity [rounds]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(2); } } while (0)

__host__ __device__ inline unsigned mix(unsigned a, unsigned b) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u);
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  h *= 0x297A2D39u;
  h ^= h >> 15;
  return h;
}

__global__ void atomics_kernel(float *a, float *b) {
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  unsafeAtomicAdd(&a[gid & 1023u], 1.0f);
  unsafeAtomicAdd(&b[mix(gid, 17u) & 1023u], 0.5f);
  atomicAdd(&a[1024 + (gid & 63u)], 2.0f);  // the CAS-free returning form as well
}

__global__ void pattern_write(unsigned *p, size_t n, unsigned round) {
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) p[k] = mix((unsigned)k, round);
}
__global__ void pattern_check(const unsigned *p, size_t n, unsigned round, unsigned long long *bad) {
  unsigned long long mine = 0;
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) mine += p[k] != mix((unsigned)k, round);
  if (mine) atomicAdd(bad, mine);
}

__host__ __device__ inline float alu_chain(unsigned id) {
  float x = (float)(id & 0xFFFFu) * (1.0f / 65536.0f) + 0.25f;
  unsigned u = id;
  for (int k = 0; k < 4096; ++k) {
    x = fmaf(x, 0.99993896484375f, 1.52587890625e-05f * (float)(u & 7u));
    u = u * 1664525u + 1013904223u;
  }
  return x + (float)(u >> 20);
}
__global__ void alu_kernel(float *out) {
  const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
  out[gid] = alu_chain(gid);
}

__global__ __launch_bounds__(256) void lds_kernel(unsigned *out) {
  __shared__ unsigned s[256];
  const unsigned t = threadIdx.x, gid = blockIdx.x * 256 + t;
  s[(t * 37u) & 255u] = mix(gid, 3u);  // 37 is odd: a permutation of 0..255
  __syncthreads();
  out[gid] = s[t];
}

int main(int argc, char **argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 3;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs, %.1f GiB, clock %d MHz\n", prop.name, prop.multiProcessorCount, prop.totalGlobalMem / 1073741824.0, prop.clockRate / 1000);
  int failures = 0;
  for (int r = 0; r < rounds; ++r) {
    // (1) atomics
    {
      float *a, *b;
      CHECK(hipMalloc(&a, 2048 * 4));
      CHECK(hipMalloc(&b, 1024 * 4));
      CHECK(hipMemset(a, 0, 2048 * 4));
      CHECK(hipMemset(b, 0, 1024 * 4));
      const int blocks = 1024, threads = 256;  // 4 096 waves
      hipLaunchKernelGGL(atomics_kernel, dim3(blocks), dim3(threads), 0, 0, a, b);
      CHECK(hipDeviceSynchronize());
      std::vector<float> ha(2048), hb(1024), eb(1024, 0.f);
      CHECK(hipMemcpy(ha.data(), a, 2048 * 4, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(hb.data(), b, 1024 * 4, hipMemcpyDeviceToHost));
      for (unsigned g = 0; g < (unsigned)(blocks * threads); ++g) eb[mix(g, 17u) & 1023u] += 0.5f;
      int bad = 0;
      for (int k = 0; k < 1024; ++k) bad += ha[k] != (float)(blocks * threads / 1024);
      for (int k = 0; k < 64; ++k) bad += ha[1024 + k] != 2.0f * (blocks * threads / 64);
      for (int k = 0; k < 1024; ++k) bad += hb[k] != eb[k];
      printf("round %d  atomics   %s (%d wrong sums)\n", r, bad ? "FAILED" : "ok", bad);
      failures += bad != 0;
      CHECK(hipFree(a));
      CHECK(hipFree(b));
    }
    // (2) copies
    {
      const size_t n = (size_t)16 << 20;  // 64 MiB of words
      unsigned *p, *q, *s;
      unsigned long long *bad;
      CHECK(hipMalloc(&p, n * 4));
      CHECK(hipMalloc(&q, n * 4));
      CHECK(hipMalloc(&s, n * 4));
      CHECK(hipMalloc(&bad, 8));
      CHECK(hipMemset(bad, 0, 8));
      hipStream_t st;
      CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      hipLaunchKernelGGL(pattern_write, dim3(2048), dim3(256), 0, st, p, n, (unsigned)r);
      CHECK(hipMemcpyAsync(q, p, n * 4, hipMemcpyDeviceToDevice, st));
      CHECK(hipMemcpyAsync(s, q, n * 4, hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(pattern_check, dim3(2048), dim3(256), 0, st, s, n, (unsigned)r, bad);
      CHECK(hipStreamSynchronize(st));
      unsigned long long hbad = 0;
      CHECK(hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost));
      std::vector<unsigned> h(n);
      CHECK(hipMemcpy(h.data(), s, n * 4, hipMemcpyDeviceToHost));
      unsigned long long hostbad = 0;
      for (size_t k = 0; k < n; k += 61) hostbad += h[k] != mix((unsigned)k, (unsigned)r);
      printf("round %d  copies    %s (%llu wrong words on the device, %llu in the host sample)\n", r, (hbad || hostbad) ? "FAILED" : "ok", hbad, hostbad);
      failures += (hbad || hostbad) != 0;
      CHECK(hipStreamDestroy(st));
      CHECK(hipFree(p));
      CHECK(hipFree(q));
      CHECK(hipFree(s));
      CHECK(hipFree(bad));
    }
    // (3) ALU
    {
      const unsigned n = 1u << 20;
      float *o;
      CHECK(hipMalloc(&o, n * 4));
      hipLaunchKernelGGL(alu_kernel, dim3(n / 256), dim3(256), 0, 0, o);
      CHECK(hipDeviceSynchronize());
      std::vector<float> h(n);
      CHECK(hipMemcpy(h.data(), o, n * 4, hipMemcpyDeviceToHost));
      int bad = 0;
      for (unsigned g = 0; g < n; g += 17) bad += h[g] != alu_chain(g);  // (the host side of 2^20 chains would take seconds: sampled)
      // all lanes against each other: ids that differ only above bit 16 start from the same x and differ through u only
      printf("round %d  alu       %s (%d of %u sampled lanes differ from the host)\n", r, bad ? "FAILED" : "ok", bad, n / 17);
      failures += bad != 0;
      CHECK(hipFree(o));
    }
    // (4) LDS
    {
      const unsigned blocks = 4096, n = blocks * 256;
      unsigned *o;
      CHECK(hipMalloc(&o, n * 4));
      hipLaunchKernelGGL(lds_kernel, dim3(blocks), dim3(256), 0, 0, o);
      CHECK(hipDeviceSynchronize());
      std::vector<unsigned> h(n);
      CHECK(hipMemcpy(h.data(), o, n * 4, hipMemcpyDeviceToHost));
      int bad = 0;
      // out[b*256 + t] = value written by the thread t' with (37 t') mod 256 == t, i.e. t' = 173 t mod 256 (37 * 173 = 1 mod 256)
      for (unsigned g = 0; g < n; ++g) {
        const unsigned b = g >> 8, t = g & 255u, tp = (t * 173u) & 255u;
        bad += h[g] != mix(b * 256 + tp, 3u);
      }
      printf("round %d  lds       %s (%d wrong words)\n", r, bad ? "FAILED" : "ok", bad);
      failures += bad != 0;
      CHECK(hipFree(o));
    }
  }
  printf(failures ? "SANITY FAILED (%d checks)\n" : "SANITY OK\n", failures);
  return failures ? 1 : 0;
}
