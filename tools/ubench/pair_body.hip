// Micro-benchmark: issue-bound cost of three formulations of the pair-kernel loop body on gfx950, with the
// memory system taken out (a 4 KB list and a 2 KB position array, L1 resident), same wave count / occupancy
// as the real kernel (12288 waves of 64, blocks of 256):
//   A  scalar, natural float4 record per j (one entry = one j)
//   B  packed across two entries with transposes ({pj[u].x, pj[u+1].x} built with v_mov)   [round-1 kernel]
//   C  packed on a pre-transposed j-PAIR record {x0,x1,y0,y1},{z0,z1,q0,q1} (one entry = two consecutive j)
// Prints us per launch and ns per pair slot.   hipcc --offload-arch=gfx950 -O3 pair_body.hip -o pair_body
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

struct Consts { float bx, by, bz, ibx, iby, ibz, r2max, two_krf; };

__device__ __forceinline__ float mi(float d, float box, float invbox) {
#pragma clang fp contract(off)
  const float magic = 12582912.0f;
  const float t = __builtin_fmaf(d, invbox, magic);
  const float k = t - magic;
  return __builtin_fmaf(-k, box, d);
}
__device__ __forceinline__ v2f mi2(v2f d, float box, float invbox) {
#pragma clang fp contract(off)
  const v2f magic = {12582912.0f, 12582912.0f};
  const v2f t = __builtin_elementwise_fma(d, v2f{invbox, invbox}, magic);
  const v2f k = t - magic;
  return __builtin_elementwise_fma(-k, v2f{box, box}, d);
}

// ---- A: scalar ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void body_a(const float4 *pos, const unsigned *list, const float4 *tab, float *out,
                                              Consts c, int nkk) {
  __shared__ float4 stab[256];
  stab[threadIdx.x] = tab[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const float4 pi = pos[(blockIdx.x * 4 + (threadIdx.x >> 6)) & 127];
  const unsigned trow = (lane & 1) << 8;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(pos), 0, 128 * 16, 0x00020000);
  const char *tbase = (const char *)stab;
  const v4u *row4 = (const v4u *)list + lane;
  const float qi2k = pi.w * c.two_krf;
  float fx = 0, fy = 0, fz = 0;
  v4u nxa = row4[0], nxb = row4[64];
  for (int kk0 = 0; kk0 < nkk; kk0 += 4) {
    const v4u cur = nxa;
    nxa = nxb;
    nxb = row4[(size_t)(((kk0 >> 2) + 2) & 3) * 64];
    const unsigned e[4] = {cur.x, cur.y, cur.z, cur.w};
    v4u raw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) raw[u] = __builtin_amdgcn_raw_buffer_load_b128(r, e[u] & 0x7F0u, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float dx = mi(pi.x - __uint_as_float(raw[u].x), c.bx, c.ibx);
      const float dy = mi(pi.y - __uint_as_float(raw[u].y), c.by, c.iby);
      const float dz = mi(pi.z - __uint_as_float(raw[u].z), c.bz, c.ibz);
      const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
      const bool hit = r2 <= c.r2max;
      const float rinv = __frsqrt_rn(r2);
      const float rinv2 = rinv * rinv;
      const float rinv6 = rinv2 * rinv2 * rinv2;
      const float4 ab = *(const float4 *)(tbase + (trow | (e[u] >> 24)));
      const float pjw = __uint_as_float(raw[u].w);
      const float qq = pi.w * pjw;
      const float p = __builtin_fmaf(ab.x, rinv6, ab.y) * rinv6;
      const float g = __builtin_fmaf(-qq, rinv, p);
      float fs = __builtin_fmaf(rinv2, g, qi2k * pjw);
      fs = hit ? fs : 0.f;
      fx = __builtin_fmaf(-dx, fs, fx);
      fy = __builtin_fmaf(-dy, fs, fy);
      fz = __builtin_fmaf(-dz, fs, fz);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = fx + fy + fz;
}

// ---- A: scalar ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void body_a_lds(const float4 *pos, const unsigned *list, const float4 *tab, float *out,
                                              Consts c, int nkk) {
  __shared__ float4 stab[256];
  __shared__ float4 spos[128];
  if (threadIdx.x < 128) spos[threadIdx.x] = pos[threadIdx.x];
  const char *pbase = (const char *)spos;
  stab[threadIdx.x] = tab[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const float4 pi = pos[(blockIdx.x * 4 + (threadIdx.x >> 6)) & 127];
  const unsigned trow = (lane & 1) << 8;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(pos), 0, 128 * 16, 0x00020000);
  const char *tbase = (const char *)stab;
  const v4u *row4 = (const v4u *)list + lane;
  const float qi2k = pi.w * c.two_krf;
  float fx = 0, fy = 0, fz = 0;
  v4u nxa = row4[0], nxb = row4[64];
  for (int kk0 = 0; kk0 < nkk; kk0 += 4) {
    const v4u cur = nxa;
    nxa = nxb;
    nxb = row4[(size_t)(((kk0 >> 2) + 2) & 3) * 64];
    const unsigned e[4] = {cur.x, cur.y, cur.z, cur.w};
    v4u raw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) raw[u] = *(const v4u *)(pbase + (e[u] & 0x7F0u));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float dx = mi(pi.x - __uint_as_float(raw[u].x), c.bx, c.ibx);
      const float dy = mi(pi.y - __uint_as_float(raw[u].y), c.by, c.iby);
      const float dz = mi(pi.z - __uint_as_float(raw[u].z), c.bz, c.ibz);
      const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
      const bool hit = r2 <= c.r2max;
      const float rinv = __frsqrt_rn(r2);
      const float rinv2 = rinv * rinv;
      const float rinv6 = rinv2 * rinv2 * rinv2;
      const float4 ab = *(const float4 *)(tbase + (trow | (e[u] >> 24)));
      const float pjw = __uint_as_float(raw[u].w);
      const float qq = pi.w * pjw;
      const float p = __builtin_fmaf(ab.x, rinv6, ab.y) * rinv6;
      const float g = __builtin_fmaf(-qq, rinv, p);
      float fs = __builtin_fmaf(rinv2, g, qi2k * pjw);
      fs = hit ? fs : 0.f;
      fx = __builtin_fmaf(-dx, fs, fx);
      fy = __builtin_fmaf(-dy, fs, fy);
      fz = __builtin_fmaf(-dz, fs, fz);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = fx + fy + fz;
}

// ---- A: scalar ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void body_a_half(const float4 *pos, const unsigned *list, const float4 *tab, float *out,
                                              Consts c, int nkk) {
  __shared__ float4 stab[256];
  stab[threadIdx.x] = tab[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const float4 pi = pos[(blockIdx.x * 4 + (threadIdx.x >> 6)) & 127];
  const unsigned trow = (lane & 1) << 8;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(pos), 0, 128 * 16, 0x00020000);
  const char *tbase = (const char *)stab;
  const v4u *row4 = (const v4u *)list + lane;
  const float qi2k = pi.w * c.two_krf;
  float fx = 0, fy = 0, fz = 0;
  v4u nxa = row4[0], nxb = row4[64];
  for (int kk0 = 0; kk0 < nkk; kk0 += 4) {
    const v4u cur = nxa;
    nxa = nxb;
    nxb = row4[(size_t)(((kk0 >> 2) + 2) & 3) * 64];
    const unsigned e[4] = {cur.x, cur.y, cur.z, cur.w};
    v4u raw[4];
    raw[0] = __builtin_amdgcn_raw_buffer_load_b128(r, e[0] & 0x7F0u, 0, 0);
    raw[2] = __builtin_amdgcn_raw_buffer_load_b128(r, e[2] & 0x7F0u, 0, 0);
    raw[1] = raw[0] ^ (v4u){e[1] & 0x3000u, e[1] & 0x5000u, e[1] & 0x6000u, 0u};
    raw[3] = raw[2] ^ (v4u){e[3] & 0x3000u, e[3] & 0x5000u, e[3] & 0x6000u, 0u};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float dx = mi(pi.x - __uint_as_float(raw[u].x), c.bx, c.ibx);
      const float dy = mi(pi.y - __uint_as_float(raw[u].y), c.by, c.iby);
      const float dz = mi(pi.z - __uint_as_float(raw[u].z), c.bz, c.ibz);
      const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
      const bool hit = r2 <= c.r2max;
      const float rinv = __frsqrt_rn(r2);
      const float rinv2 = rinv * rinv;
      const float rinv6 = rinv2 * rinv2 * rinv2;
      const float4 ab = *(const float4 *)(tbase + (trow | (e[u] >> 24)));
      const float pjw = __uint_as_float(raw[u].w);
      const float qq = pi.w * pjw;
      const float p = __builtin_fmaf(ab.x, rinv6, ab.y) * rinv6;
      const float g = __builtin_fmaf(-qq, rinv, p);
      float fs = __builtin_fmaf(rinv2, g, qi2k * pjw);
      fs = hit ? fs : 0.f;
      fx = __builtin_fmaf(-dx, fs, fx);
      fy = __builtin_fmaf(-dy, fs, fy);
      fz = __builtin_fmaf(-dz, fs, fz);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = fx + fy + fz;
}

// ---- A: scalar ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void body_a_quarter(const float4 *pos, const unsigned *list, const float4 *tab, float *out,
                                              Consts c, int nkk) {
  __shared__ float4 stab[256];
  stab[threadIdx.x] = tab[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const float4 pi = pos[(blockIdx.x * 4 + (threadIdx.x >> 6)) & 127];
  const unsigned trow = (lane & 1) << 8;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(pos), 0, 128 * 16, 0x00020000);
  const char *tbase = (const char *)stab;
  const v4u *row4 = (const v4u *)list + lane;
  const float qi2k = pi.w * c.two_krf;
  float fx = 0, fy = 0, fz = 0;
  v4u nxa = row4[0], nxb = row4[64];
  for (int kk0 = 0; kk0 < nkk; kk0 += 4) {
    const v4u cur = nxa;
    nxa = nxb;
    nxb = row4[(size_t)(((kk0 >> 2) + 2) & 3) * 64];
    const unsigned e[4] = {cur.x, cur.y, cur.z, cur.w};
    v4u raw[4];
    raw[0] = __builtin_amdgcn_raw_buffer_load_b128(r, e[0] & 0x7F0u, 0, 0);
#pragma unroll
    for (int u = 1; u < 4; ++u) raw[u] = raw[0] ^ (v4u){e[u] & 0x3000u, e[u] & 0x5000u, e[u] & 0x6000u, 0u};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float dx = mi(pi.x - __uint_as_float(raw[u].x), c.bx, c.ibx);
      const float dy = mi(pi.y - __uint_as_float(raw[u].y), c.by, c.iby);
      const float dz = mi(pi.z - __uint_as_float(raw[u].z), c.bz, c.ibz);
      const float r2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
      const bool hit = r2 <= c.r2max;
      const float rinv = __frsqrt_rn(r2);
      const float rinv2 = rinv * rinv;
      const float rinv6 = rinv2 * rinv2 * rinv2;
      const float4 ab = *(const float4 *)(tbase + (trow | (e[u] >> 24)));
      const float pjw = __uint_as_float(raw[u].w);
      const float qq = pi.w * pjw;
      const float p = __builtin_fmaf(ab.x, rinv6, ab.y) * rinv6;
      const float g = __builtin_fmaf(-qq, rinv, p);
      float fs = __builtin_fmaf(rinv2, g, qi2k * pjw);
      fs = hit ? fs : 0.f;
      fx = __builtin_fmaf(-dx, fs, fx);
      fy = __builtin_fmaf(-dy, fs, fy);
      fz = __builtin_fmaf(-dz, fs, fz);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = fx + fy + fz;
}

// ---- B: packed across entries, transposes ---------------------------------------------------------------
__global__ __launch_bounds__(256) void body_b(const float4 *pos, const unsigned *list, const float4 *tab, float *out,
                                              Consts c, int nkk) {
  __shared__ float4 stab[256];
  stab[threadIdx.x] = tab[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const float4 pi = pos[(blockIdx.x * 4 + (threadIdx.x >> 6)) & 127];
  const unsigned trow = (lane & 1) << 8;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(pos), 0, 128 * 16, 0x00020000);
  const char *tbase = (const char *)stab;
  const v4u *row4 = (const v4u *)list + lane;
  const v2f pix = {pi.x, pi.x}, piy = {pi.y, pi.y}, piz = {pi.z, pi.z}, piw = {pi.w, pi.w};
  v2f fx = {0, 0}, fy = {0, 0}, fz = {0, 0};
  v4u nxa = row4[0], nxb = row4[64];
  for (int kk0 = 0; kk0 < nkk; kk0 += 4) {
    const v4u cur = nxa;
    nxa = nxb;
    nxb = row4[(size_t)(((kk0 >> 2) + 2) & 3) * 64];
    const unsigned e[4] = {cur.x, cur.y, cur.z, cur.w};
    v4u raw[4];
    float2 ab[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      raw[u] = __builtin_amdgcn_raw_buffer_load_b128(r, e[u] & 0x7F0u, 0, 0);
      ab[u] = *(const float2 *)(tbase + (trow | (e[u] >> 24)));
    }
#pragma unroll
    for (int u = 0; u < 4; u += 2) {
      const v2f pjx = {__uint_as_float(raw[u].x), __uint_as_float(raw[u + 1].x)};
      const v2f pjy = {__uint_as_float(raw[u].y), __uint_as_float(raw[u + 1].y)};
      const v2f pjz = {__uint_as_float(raw[u].z), __uint_as_float(raw[u + 1].z)};
      const v2f pjw = {__uint_as_float(raw[u].w), __uint_as_float(raw[u + 1].w)};
      const v2f dx = mi2(pix - pjx, c.bx, c.ibx), dy = mi2(piy - pjy, c.by, c.iby), dz = mi2(piz - pjz, c.bz, c.ibz);
      const v2f r2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
      const bool h0 = r2.x <= c.r2max, h1 = r2.y <= c.r2max;
      const v2f rinv = {__frsqrt_rn(r2.x), __frsqrt_rn(r2.y)};
      const v2f rinv2 = rinv * rinv;
      const v2f rinv6 = rinv2 * rinv2 * rinv2;
      const v2f a12 = {ab[u].x, ab[u + 1].x}, b6 = {ab[u].y, ab[u + 1].y};
      v2f fs = __builtin_elementwise_fma(a12, rinv6, b6) * (rinv6 * rinv2);
      fs += (piw * pjw) * (c.two_krf - rinv2 * rinv);
      fs = v2f{h0 ? fs.x : 0.f, h1 ? fs.y : 0.f};
      fx -= dx * fs;
      fy -= dy * fs;
      fz -= dz * fs;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = fx.x + fx.y + fy.x + fy.y + fz.x + fz.y;
}

// ---- C: packed, pre-transposed j-pair records (32 B: {x0,x1,y0,y1},{z0,z1,q0,q1}); one entry = 2 slots -------
__global__ __launch_bounds__(256) void body_c(const float4 *pos, const unsigned *list, const float4 *tab, float *out,
                                              Consts c, int nkk) {
  __shared__ float4 stab[256];  // per (type_i, pair code): {a12_0, a12_1, b6_0, b6_1}
  stab[threadIdx.x] = tab[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const float4 pi = pos[(blockIdx.x * 4 + (threadIdx.x >> 6)) & 127];
  const unsigned trow = (lane & 1) << 8;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(pos), 0, 128 * 16, 0x00020000);
  const char *tbase = (const char *)stab;
  const v4u *row4 = (const v4u *)list + lane;
  const v2f pix = {pi.x, pi.x}, piy = {pi.y, pi.y}, piz = {pi.z, pi.z}, piw = {pi.w, pi.w};
  v2f fx = {0, 0}, fy = {0, 0}, fz = {0, 0};
  v4u nxa = row4[0], nxb = row4[64];
  for (int kk0 = 0; kk0 < nkk; kk0 += 4) {  // 4 entries = 8 slots per iteration
    const v4u cur = nxa;
    nxa = nxb;
    nxb = row4[(size_t)(((kk0 >> 2) + 2) & 3) * 64];
    const unsigned e[4] = {cur.x, cur.y, cur.z, cur.w};
    v4u ra[4], rb[4];
    float4 ab[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned off = e[u] & 0x7E0u;
      ra[u] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
      rb[u] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 16, 0);
      ab[u] = *(const float4 *)(tbase + (trow | (e[u] >> 24)));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const v2f pjx = {__uint_as_float(ra[u].x), __uint_as_float(ra[u].y)};
      const v2f pjy = {__uint_as_float(ra[u].z), __uint_as_float(ra[u].w)};
      const v2f pjz = {__uint_as_float(rb[u].x), __uint_as_float(rb[u].y)};
      const v2f pjw = {__uint_as_float(rb[u].z), __uint_as_float(rb[u].w)};
      const v2f dx = mi2(pix - pjx, c.bx, c.ibx), dy = mi2(piy - pjy, c.by, c.iby), dz = mi2(piz - pjz, c.bz, c.ibz);
      const v2f r2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
      const bool h0 = r2.x <= c.r2max, h1 = r2.y <= c.r2max;
      const v2f rinv = {__frsqrt_rn(r2.x), __frsqrt_rn(r2.y)};
      const v2f rinv2 = rinv * rinv;
      const v2f rinv6 = rinv2 * rinv2 * rinv2;
      const v2f a12 = {ab[u].x, ab[u].y}, b6 = {ab[u].z, ab[u].w};
      v2f fs = __builtin_elementwise_fma(a12, rinv6, b6) * (rinv6 * rinv2);
      fs += (piw * pjw) * (c.two_krf - rinv2 * rinv);
      fs = v2f{h0 ? fs.x : 0.f, h1 ? fs.y : 0.f};
      fx -= dx * fs;
      fy -= dy * fs;
      fz -= dz * fs;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = fx.x + fx.y + fy.x + fy.y + fz.x + fz.y;
}

// ---- B: packed across entries, transposes ---------------------------------------------------------------
__global__ __launch_bounds__(256) void body_b_lds(const float4 *pos, const unsigned *list, const float4 *tab, float *out,
                                              Consts c, int nkk) {
  __shared__ float4 stab[256];
  __shared__ float4 spos[128];
  if (threadIdx.x < 128) spos[threadIdx.x] = pos[threadIdx.x];
  const char *pbase = (const char *)spos;
  stab[threadIdx.x] = tab[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const float4 pi = pos[(blockIdx.x * 4 + (threadIdx.x >> 6)) & 127];
  const unsigned trow = (lane & 1) << 8;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(pos), 0, 128 * 16, 0x00020000);
  const char *tbase = (const char *)stab;
  const v4u *row4 = (const v4u *)list + lane;
  const v2f pix = {pi.x, pi.x}, piy = {pi.y, pi.y}, piz = {pi.z, pi.z}, piw = {pi.w, pi.w};
  v2f fx = {0, 0}, fy = {0, 0}, fz = {0, 0};
  v4u nxa = row4[0], nxb = row4[64];
  for (int kk0 = 0; kk0 < nkk; kk0 += 4) {
    const v4u cur = nxa;
    nxa = nxb;
    nxb = row4[(size_t)(((kk0 >> 2) + 2) & 3) * 64];
    const unsigned e[4] = {cur.x, cur.y, cur.z, cur.w};
    v4u raw[4];
    float2 ab[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      raw[u] = *(const v4u *)(pbase + (e[u] & 0x7F0u));
      ab[u] = *(const float2 *)(tbase + (trow | (e[u] >> 24)));
    }
#pragma unroll
    for (int u = 0; u < 4; u += 2) {
      const v2f pjx = {__uint_as_float(raw[u].x), __uint_as_float(raw[u + 1].x)};
      const v2f pjy = {__uint_as_float(raw[u].y), __uint_as_float(raw[u + 1].y)};
      const v2f pjz = {__uint_as_float(raw[u].z), __uint_as_float(raw[u + 1].z)};
      const v2f pjw = {__uint_as_float(raw[u].w), __uint_as_float(raw[u + 1].w)};
      const v2f dx = mi2(pix - pjx, c.bx, c.ibx), dy = mi2(piy - pjy, c.by, c.iby), dz = mi2(piz - pjz, c.bz, c.ibz);
      const v2f r2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
      const bool h0 = r2.x <= c.r2max, h1 = r2.y <= c.r2max;
      const v2f rinv = {__frsqrt_rn(r2.x), __frsqrt_rn(r2.y)};
      const v2f rinv2 = rinv * rinv;
      const v2f rinv6 = rinv2 * rinv2 * rinv2;
      const v2f a12 = {ab[u].x, ab[u + 1].x}, b6 = {ab[u].y, ab[u + 1].y};
      v2f fs = __builtin_elementwise_fma(a12, rinv6, b6) * (rinv6 * rinv2);
      fs += (piw * pjw) * (c.two_krf - rinv2 * rinv);
      fs = v2f{h0 ? fs.x : 0.f, h1 ? fs.y : 0.f};
      fx -= dx * fs;
      fy -= dy * fs;
      fz -= dz * fs;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = fx.x + fx.y + fy.x + fy.y + fz.x + fz.y;
}

// ---- C: packed, pre-transposed j-pair records (32 B: {x0,x1,y0,y1},{z0,z1,q0,q1}); one entry = 2 slots -------
__global__ __launch_bounds__(256) void body_c_lds(const float4 *pos, const unsigned *list, const float4 *tab, float *out,
                                              Consts c, int nkk) {
  __shared__ float4 stab[256];
  __shared__ float4 spos[128];
  if (threadIdx.x < 128) spos[threadIdx.x] = pos[threadIdx.x];
  const char *pbase = (const char *)spos;
  stab[threadIdx.x] = tab[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const float4 pi = pos[(blockIdx.x * 4 + (threadIdx.x >> 6)) & 127];
  const unsigned trow = (lane & 1) << 8;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(pos), 0, 128 * 16, 0x00020000);
  const char *tbase = (const char *)stab;
  const v4u *row4 = (const v4u *)list + lane;
  const v2f pix = {pi.x, pi.x}, piy = {pi.y, pi.y}, piz = {pi.z, pi.z}, piw = {pi.w, pi.w};
  v2f fx = {0, 0}, fy = {0, 0}, fz = {0, 0};
  v4u nxa = row4[0], nxb = row4[64];
  for (int kk0 = 0; kk0 < nkk; kk0 += 4) {  // 4 entries = 8 slots per iteration
    const v4u cur = nxa;
    nxa = nxb;
    nxb = row4[(size_t)(((kk0 >> 2) + 2) & 3) * 64];
    const unsigned e[4] = {cur.x, cur.y, cur.z, cur.w};
    v4u ra[4], rb[4];
    float4 ab[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned off = e[u] & 0x7E0u;
      ra[u] = *(const v4u *)(pbase + off);
      rb[u] = *(const v4u *)(pbase + off + 16);
      ab[u] = *(const float4 *)(tbase + (trow | (e[u] >> 24)));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const v2f pjx = {__uint_as_float(ra[u].x), __uint_as_float(ra[u].y)};
      const v2f pjy = {__uint_as_float(ra[u].z), __uint_as_float(ra[u].w)};
      const v2f pjz = {__uint_as_float(rb[u].x), __uint_as_float(rb[u].y)};
      const v2f pjw = {__uint_as_float(rb[u].z), __uint_as_float(rb[u].w)};
      const v2f dx = mi2(pix - pjx, c.bx, c.ibx), dy = mi2(piy - pjy, c.by, c.iby), dz = mi2(piz - pjz, c.bz, c.ibz);
      const v2f r2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
      const bool h0 = r2.x <= c.r2max, h1 = r2.y <= c.r2max;
      const v2f rinv = {__frsqrt_rn(r2.x), __frsqrt_rn(r2.y)};
      const v2f rinv2 = rinv * rinv;
      const v2f rinv6 = rinv2 * rinv2 * rinv2;
      const v2f a12 = {ab[u].x, ab[u].y}, b6 = {ab[u].z, ab[u].w};
      v2f fs = __builtin_elementwise_fma(a12, rinv6, b6) * (rinv6 * rinv2);
      fs += (piw * pjw) * (c.two_krf - rinv2 * rinv);
      fs = v2f{h0 ? fs.x : 0.f, h1 ? fs.y : 0.f};
      fx -= dx * fs;
      fy -= dy * fs;
      fz -= dz * fs;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = fx.x + fx.y + fy.x + fy.y + fz.x + fz.y;
}

int main() {
  const int natoms = 128, waves = 12288, blocks = waves / 4;
  std::vector<float4> pos(natoms), tab(256);
  srand(1);
  for (auto &p : pos) p = make_float4(rand() % 1000 * 0.03f, rand() % 1000 * 0.03f, rand() % 1000 * 0.03f, 0.4f);
  for (auto &t : tab) t = make_float4(-12.f * 5e5f, 6.f * 600.f, -12.f * 4e5f, 6.f * 500.f);
  std::vector<unsigned> list(6 * 64 * 4);
  for (auto &e : list) e = ((unsigned)(rand() % natoms) << 4) | ((unsigned)(rand() % 2) << 28);
  float4 *dpos, *dtab;
  unsigned *dlist;
  float *dout;
  CHECK(hipMalloc(&dpos, sizeof(float4) * natoms));
  CHECK(hipMalloc(&dtab, sizeof(float4) * 256));
  CHECK(hipMalloc(&dlist, sizeof(unsigned) * list.size()));
  CHECK(hipMalloc(&dout, sizeof(float) * blocks * 256));
  CHECK(hipMemcpy(dpos, pos.data(), sizeof(float4) * natoms, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dtab, tab.data(), sizeof(float4) * 256, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dlist, list.data(), sizeof(unsigned) * list.size(), hipMemcpyHostToDevice));
  Consts c{30.f, 30.f, 30.f, 1 / 30.f, 1 / 30.f, 1 / 30.f, 81.f, 0.001f};
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  struct K { const char *name; void (*fn)(const float4 *, const unsigned *, const float4 *, float *, Consts, int); int nkk; int slots_per_entry; };
  // real kernel: 442 entries per atom / 8 lanes = 56 iterations per lane (A, B); C: 28 pair entries per lane
  K ks[] = {{"A scalar", body_a, 56, 1}, {"A, gather from LDS", body_a_lds, 56, 1}, {"A, 1 gather / 2 slots", body_a_half, 56, 1},
            {"A, 1 gather / 4 slots", body_a_quarter, 56, 1}, {"B packed+transposes", body_b, 56, 1}, {"C packed j-pairs", body_c, 28, 2}, {"B, gather from LDS", body_b_lds, 56, 1}, {"C, gather from LDS", body_c_lds, 28, 2}};
  for (auto &k : ks) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, dpos, dlist, dtab, dout, c, k.nkk);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 10; ++rep) {
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, dpos, dlist, dtab, dout, c, k.nkk);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    const double slots = (double)waves * 64 * k.nkk * k.slots_per_entry;
    printf("%-22s %7.2f us per launch  (%.3e pair slots, %.2f ps per slot)\n", k.name, best * 1e3, slots, best * 1e9 / slots);
  }
  return 0;
}
