// Spatial domain decomposition support for gfx950: the per-step kernels of one brick.
//
// The reference has no counterpart (single process, single device).  A brick
// integrates the atoms it owns in place inside the position buffer of its force engine ([own | halo] rows)
// and sends the positions of the atoms that lie within `cutoff + skin` of a face to the neighbouring bricks,
// already shifted to the periodic image the receiver sees (torchmd_amd/domain.py).  Per step this file
// contributes two launches:
//   dd_step_kernel    second half kick of step s-1 (+ Langevin) and first half step of step s on the owned
//                     atoms (same expressions, same rounding as integrator.hip's separate kernels:
//                     integrator.py:61-74), plus the running maximum of |x - x_ref|^2 that triggers the next
//                     migration (wave maximum, then at most one atomic per wave);
//   halo_pack_kernel  out[k] = pos[send_index[k]] + send_shift[k] for the rows of all outgoing messages, in
//                     message order: the buffer the all-to-all sends.
// The receiver needs no unpack kernel: the exchange writes straight into the halo rows of its engine's
// position buffer.
//
// tmdhip_dd_run's own loop (round 4) folds the small launches around those two into them — a brick of 125 000 + 50 000
// atoms is launch-bound: twelve 2-5 us launches per step beside a 20-us pair kernel — leaving three kernels and the
// exchange per step:
//   dd_own_kernel     dd_step_kernel's update of the owned atoms + the engine's list displacement test for them + their
//                     records of the cell-sorted copy the pair kernel gathers + the rows of the outgoing messages (an
//                     atom writes its own rows through a per-atom index of the send list built once per migration);
//   (exchange)
//   dd_halo_kernel    displacement test and cell-sorted records of the halo rows that just arrived;
//   pair kernel       with the rebuild chain left out while no atom is near its limit (the host paces itself one step
//                     behind the device exactly like tmdhip_md_run: ListCheck in engine.h).
// Same arithmetic, same rebuild decisions: trajectories are bit-identical to the loop of separate launches
// (TMDHIP_DD_FUSED=0).
//
// Exchange and step loop from C (tmdhip_comm_*, tmdhip_dd_run): RCCL point-to-point between the ranks of the
// brick grid — a brick of a 2 x 2 x 2 grid has 7 distinct neighbour ranks, one per xGMI link — as ONE group of
// ncclSend/ncclRecv per step, enqueued on the same stream as the kernels (no cross-stream events, no host
// round trip).  librccl is opened at run time (the copy the process has already mapped: PyTorch's), so the
// library carries no link-time dependency on it and loads without it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "dd_comm.h"

using namespace tmd;

namespace {

template <typename R, bool LANGEVIN>
__global__ __launch_bounds__(256) void dd_step_kernel(int64_t nown, R *__restrict__ pos, R *__restrict__ vel,
                                                      const R *__restrict__ f, const R *__restrict__ mass,
                                                      const R *__restrict__ vcoeff, R dt, R half_dt, R gamma,
                                                      uint64_t seed, uint64_t step, int phases,
                                                      const R *__restrict__ ref, unsigned *__restrict__ disp2) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float d2 = 0.f;
  if (i < nown) {
    const R m = mass[i];
    R g[3] = {0, 0, 0}, vc = 0;
    if (LANGEVIN && (phases & 1)) {
      vc = vcoeff[i];
      normal3<R>(seed, step, (uint64_t)i, g[0], g[1], g[2]);
    }
    R dd = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const R a = f[3 * i + k] / m;
      R v = vel[3 * i + k];
      if (phases & 1) {  // integrator.py:72-74 (thermostat) then 67-69
        if (LANGEVIN) v += -gamma * v * dt + g[k] * vc;
        v += half_dt * a;
      }
      if (phases & 2) {  // integrator.py:61-64
        const R p = pos[3 * i + k] + (v * dt + R(0.5) * a * dt * dt);
        v = v + half_dt * a;
        pos[3 * i + k] = p;
        if (ref) {
          const R d = p - ref[3 * i + k];
          dd += d * d;
        }
      }
      vel[3 * i + k] = v;
    }
    d2 = sizeof(R) == 4 ? (float)dd : __double2float_ru((double)dd);
  }
  if (disp2 && (phases & 2)) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d2 = fmaxf(d2, __shfl_xor(d2, o, 64));
    // non-negative floats order like their bit patterns
    if ((threadIdx.x & 63) == 0 && __float_as_uint(d2) > *disp2) atomicMax(disp2, __float_as_uint(d2));
  }
}

template <typename R>
__global__ void halo_pack_kernel(int64_t n3, const R *__restrict__ pos, const int32_t *__restrict__ index,
                                 const R *__restrict__ shift, R *__restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n3) return;
  const int64_t row = t / 3;
  const int k = (int)(t - 3 * row);
  out[t] = pos[3 * (int64_t)index[row] + k] + shift[t];
}

// ---- the brick step of tmdhip_dd_run in three launches (see the head comment) -----------------------------------
template <typename R>
struct DdOwnArgs {
  int64_t nown;
  R *pos, *vel;
  const R *f, *mass, *vcoeff;
  R dt, half_dt, gamma;
  uint64_t seed, step;
  int phases;
  const R *ref_mig;  // positions at the last migration (the halo's skin)
  unsigned *disp2;
  ListCheck<R> chk;  // the engine's list: reference positions of its last build, original row order
  typename Vec<R>::T4 *sorted;
  const int *inv;
  const R *qs;
  const int *csr_off, *csr_row;  // send rows of every owned atom (tmdhip_comm: built once per migration)
  const R *shift;
  R *out;
};

template <typename R, bool LANGEVIN>
__global__ __launch_bounds__(256) void dd_own_kernel(DdOwnArgs<R> a, PairConsts<R> c) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool drift = a.phases & 2;
  if (i == 0 && drift) list_check_clear(a.chk.flags, a.chk.parity);
  float d2 = 0.f;
  if (i < a.nown) {
    // every load of the update in one batch (the kernel is a chain of memory round trips per wave)
    const R m = a.mass[i];
    R vc = 0, v[3], fk[3], p[3] = {0, 0, 0}, rm[3] = {0, 0, 0}, rl[3] = {0, 0, 0}, q = 0, h2 = 0;
    int slot = 0, s0 = 0, s1 = 0;
    if (LANGEVIN && (a.phases & 1)) vc = a.vcoeff[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      v[k] = a.vel[3 * i + k];
      fk[k] = a.f[3 * i + k];
      if (drift) {
        p[k] = a.pos[3 * i + k];
        rm[k] = a.ref_mig[3 * i + k];
        rl[k] = a.chk.ref[3 * i + k];
      }
    }
    if (drift) {
      q = a.qs[i];
      h2 = list_check_limit(a.chk, (int)i);
      slot = a.inv[i];
      s0 = a.csr_off[i];
      s1 = a.csr_off[i + 1];
    }
    R g[3] = {0, 0, 0};
    if (LANGEVIN && (a.phases & 1)) normal3<R>(a.seed, a.step, (uint64_t)i, g[0], g[1], g[2]);
    R dd = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // the expressions of dd_step_kernel, in its order
      const R acc = fk[k] / m;
      R vk = v[k];
      if (a.phases & 1) {
        if (LANGEVIN) vk += -a.gamma * vk * a.dt + g[k] * vc;
        vk += a.half_dt * acc;
      }
      if (drift) {
        p[k] = p[k] + (vk * a.dt + R(0.5) * acc * a.dt * a.dt);
        vk = vk + a.half_dt * acc;
        a.pos[3 * i + k] = p[k];
        const R d = p[k] - rm[k];
        dd += d * d;
      }
      a.vel[3 * i + k] = vk;
    }
    d2 = sizeof(R) == 4 ? (float)dd : __double2float_ru((double)dd);
    if (drift) {
      typename Vec<R>::T4 sv;
      sv.x = p[0];
      sv.y = p[1];
      sv.z = p[2];
      sv.w = q;
      a.sorted[slot] = sv;
      extent_note<R>(a.chk.ext, p[0], p[1], p[2]);
      list_check_point<R>(a.chk, c, p[0] - rl[0], p[1] - rl[1], p[2] - rl[2], h2);
      for (int s = s0; s < s1; ++s) {  // this atom's rows of the outgoing messages (halo_pack_kernel's expression)
        const int64_t k = a.csr_row[s];
#pragma unroll
        for (int x = 0; x < 3; ++x) a.out[3 * k + x] = p[x] + a.shift[3 * k + x];
      }
    }
  }
  if (a.disp2 && drift) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d2 = fmaxf(d2, __shfl_xor(d2, o, 64));
    if ((threadIdx.x & 63) == 0 && __float_as_uint(d2) > *a.disp2) atomicMax(a.disp2, __float_as_uint(d2));
  }
}

// halo rows [first, n) of the position buffer, just received: displacement test + cell-sorted records
template <typename R>
__global__ __launch_bounds__(256) void dd_halo_kernel(int first, int n, const R *__restrict__ pos, ListCheck<R> chk,
                                                      PairConsts<R> c, const int *__restrict__ inv, const R *__restrict__ qs,
                                                      typename Vec<R>::T4 *__restrict__ sorted) {
  const int i = first + (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  const R x = pos[3 * (size_t)i + 0], y = pos[3 * (size_t)i + 1], z = pos[3 * (size_t)i + 2];
  typename Vec<R>::T4 rec;
  rec.x = x;
  rec.y = y;
  rec.z = z;
  rec.w = qs[i];
  const int slot = inv[i];
  list_check_atom<R>(chk, c, i, x, y, z);
  extent_note<R>(chk.ext, x, y, z);
  sorted[slot] = rec;
}

// per-atom index of the send list (CSR over the owned atoms): count, scan (hipcub), fill
__global__ void csr_count_kernel(int64_t nsend, const int32_t *__restrict__ index, int *__restrict__ cnt) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nsend) atomicAdd(&cnt[index[k]], 1);
}
__global__ void csr_fill_kernel(int64_t nsend, const int32_t *__restrict__ index, int *__restrict__ cursor,
                                int *__restrict__ row) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nsend) row[atomicAdd(&cursor[index[k]], 1)] = (int)k;
}

inline dim3 blocks_for(int64_t n, int t) { return dim3((unsigned)((n + t - 1) / t)); }

template <typename R>
int launch_dd_step(int64_t nown, void *pos, void *vel, const void *forces, const void *mass, const void *vcoeff,
                   double dt, double gamma, uint64_t seed, uint64_t step, int phases, const void *ref,
                   uint32_t *disp2, hipStream_t st) {
  if (vcoeff)
    hipLaunchKernelGGL((dd_step_kernel<R, true>), blocks_for(nown, 256), dim3(256), 0, st, nown, (R *)pos, (R *)vel,
                       (const R *)forces, (const R *)mass, (const R *)vcoeff, (R)dt, (R)(0.5 * dt), (R)gamma, seed,
                       step, phases, (const R *)ref, disp2);
  else
    hipLaunchKernelGGL((dd_step_kernel<R, false>), blocks_for(nown, 256), dim3(256), 0, st, nown, (R *)pos, (R *)vel,
                       (const R *)forces, (const R *)mass, (const R *)nullptr, (R)dt, (R)(0.5 * dt), (R)0, seed, step,
                       phases, (const R *)ref, disp2);
  TMD_HIP(hipGetLastError());
  return 0;
}


int load_rccl(const char *path, RcclApi &api) {
  const char *names[] = {path, "librccl.so.1", "librccl.so"};
  for (const char *nm : names) {
    if (!nm || !*nm) continue;
    api.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (api.handle) break;
  }
  if (!api.handle) return fail(std::string("librccl could not be opened: ") + (dlerror() ? dlerror() : "?"));
  bool ok = true;
  auto sym = [&](const char *n) {
    void *p = dlsym(api.handle, n);
    ok = ok && p;
    return p;
  };
  api.get_unique_id = (decltype(api.get_unique_id))sym("ncclGetUniqueId");
  api.comm_init_rank = (decltype(api.comm_init_rank))sym("ncclCommInitRank");
  api.comm_destroy = (decltype(api.comm_destroy))sym("ncclCommDestroy");
  api.error_string = (decltype(api.error_string))sym("ncclGetErrorString");
  api.group_start = (decltype(api.group_start))sym("ncclGroupStart");
  api.group_end = (decltype(api.group_end))sym("ncclGroupEnd");
  api.send = (decltype(api.send))sym("ncclSend");
  api.recv = (decltype(api.recv))sym("ncclRecv");
  api.all_reduce = (decltype(api.all_reduce))sym("ncclAllReduce");
  if (!ok) return fail("librccl lacks a required symbol");
  return 0;
}

}  // namespace

struct MaxPtrs {
  const float *p[64];
};
__global__ void local_max_kernel(MaxPtrs src, int world, float *out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float v = src.p[0][0];
    for (int r = 1; r < world; ++r) v = fmaxf(v, src.p[r][0]);
    *out = v;
  }
}


namespace {

int local_exchange_rows(tmdhip_comm *c, size_t esz, const void *send, const int64_t *send_counts, void *recv,
                        const int64_t *recv_counts, int width, hipStream_t st) {
  tmdhip_local_hub *h = c->hub;
  LocalSlot &me = h->slot[c->rank];
  me.send = send;
  me.send_counts = send_counts;
  TMD_HIP(hipEventRecord(me.ready, st));
  if (!h->barrier()) return fail("in-process communicator: a rank did not arrive at the exchange");
  size_t ro = 0;
  for (int p = 0; p < c->world; ++p) {
    const LocalSlot &src = h->slot[p];
    const size_t n = (size_t)src.send_counts[c->rank] * width;  // what p sends to me
    if (n != (size_t)recv_counts[p] * width) {
      h->barrier();
      return fail("in-process communicator: send / receive counts of ranks " + std::to_string(p) + " and " +
                  std::to_string(c->rank) + " disagree");
    }
    if (n) {
      size_t so = 0;
      for (int q = 0; q < c->rank; ++q) so += (size_t)src.send_counts[q] * width;
      if (p != c->rank) TMD_HIP(hipStreamWaitEvent(st, src.ready, 0));
      TMD_HIP(hipMemcpyAsync((char *)recv + ro * esz, (const char *)src.send + so * esz, n * esz, hipMemcpyDeviceToDevice, st));
    }
    ro += n;
  }
  TMD_HIP(hipEventRecord(me.done, st));
  if (!h->barrier()) return fail("in-process communicator: a rank did not arrive behind the exchange");
  for (int p = 0; p < c->world; ++p)  // my send buffer may be rewritten once its readers are through
    if (p != c->rank && send_counts[p] > 0) TMD_HIP(hipStreamWaitEvent(st, h->slot[p].done, 0));
  return 0;
}

int local_allreduce_max(tmdhip_comm *c, float *buf, hipStream_t st) {
  tmdhip_local_hub *h = c->hub;
  LocalSlot &me = h->slot[c->rank];
  me.red = buf;
  TMD_HIP(hipEventRecord(me.ready, st));
  if (!h->barrier()) return fail("in-process communicator: a rank did not arrive at the reduction");
  MaxPtrs src;
  for (int p = 0; p < c->world; ++p) {
    src.p[p] = h->slot[p].red;
    if (p != c->rank) TMD_HIP(hipStreamWaitEvent(st, h->slot[p].ready, 0));
  }
  hipLaunchKernelGGL(local_max_kernel, dim3(1), dim3(64), 0, st, src, c->world, me.red_tmp);
  TMD_HIP(hipGetLastError());
  TMD_HIP(hipEventRecord(me.done, st));
  if (!h->barrier()) return fail("in-process communicator: a rank did not arrive behind the reduction");
  for (int p = 0; p < c->world; ++p)  // everybody has read my operand: now it may receive the result
    if (p != c->rank) TMD_HIP(hipStreamWaitEvent(st, h->slot[p].done, 0));
  TMD_HIP(hipMemcpyAsync(buf, me.red_tmp, sizeof(float), hipMemcpyDeviceToDevice, st));
  return 0;
}

}  // namespace

namespace tmd {

int exchange_rows(tmdhip_comm *c, int dtype, const void *send, const int64_t *send_counts, void *recv,
                  const int64_t *recv_counts, int width, hipStream_t st) {
  const ncclDataType_t dt = dtype == TMDHIP_F32 ? ncclFloat32 : ncclFloat64;
  const size_t esz = dtype == TMDHIP_F32 ? 4 : 8;
  if (c->hub) return local_exchange_rows(c, esz, send, send_counts, recv, recv_counts, width, st);
  TMD_NCCL(c, c->api.group_start());
  size_t so = 0, ro = 0;
  for (int p = 0; p < c->world; ++p) {
    const size_t ns = (size_t)send_counts[p] * width, nr = (size_t)recv_counts[p] * width;
    if (ns) TMD_NCCL(c, c->api.send((const char *)send + so * esz, ns, dt, p, c->comm, st));
    if (nr) TMD_NCCL(c, c->api.recv((char *)recv + ro * esz, nr, dt, p, c->comm, st));
    so += ns;
    ro += nr;
  }
  TMD_NCCL(c, c->api.group_end());
  return 0;
}

// in-place maximum over the ranks of one float on the device
int allreduce_max(tmdhip_comm *c, void *buf, hipStream_t st) {
  if (c->world == 1) return 0;
  if (c->hub) return local_allreduce_max(c, (float *)buf, st);
  TMD_NCCL(c, c->api.all_reduce(buf, buf, 1, ncclFloat32, ncclMax, c->comm, st));
  return 0;
}

// all-to-all of one count per peer between the hosts (migrations: how many rows will arrive from whom)
int exchange_counts(tmdhip_comm *c, const int64_t *send_counts, int64_t *recv_counts, hipStream_t st) {
  if (c->world == 1) {
    recv_counts[0] = send_counts[0];
    return 0;
  }
  if (c->hub) {
    tmdhip_local_hub *h = c->hub;
    h->slot[c->rank].counts = send_counts;
    if (!h->barrier()) return fail("in-process communicator: a rank did not arrive at the count exchange");
    for (int p = 0; p < c->world; ++p) recv_counts[p] = h->slot[p].counts[c->rank];
    if (!h->barrier()) return fail("in-process communicator: a rank did not arrive behind the count exchange");
    return 0;
  }
  // RCCL: the counts travel as 8-byte words through a small device buffer
  TMD_TRY(c->cnt_dev.ensure(sizeof(int64_t) * 2 * (size_t)c->world));
  if (!c->cnt_host && hipHostMalloc((void **)&c->cnt_host, sizeof(int64_t) * 2 * (size_t)c->world, hipHostMallocDefault) != hipSuccess)
    return fail("exchange_counts: pinned allocation failed");
  int64_t *dev = c->cnt_dev.as<int64_t>();
  for (int p = 0; p < c->world; ++p) c->cnt_host[p] = send_counts[p];
  TMD_HIP(hipMemcpyAsync(dev, c->cnt_host, sizeof(int64_t) * c->world, hipMemcpyHostToDevice, st));
  TMD_NCCL(c, c->api.group_start());
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) continue;  // (what a rank keeps needs no message)
    TMD_NCCL(c, c->api.send(dev + p, 1, ncclFloat64, p, c->comm, st));
    TMD_NCCL(c, c->api.recv(dev + c->world + p, 1, ncclFloat64, p, c->comm, st));
  }
  TMD_NCCL(c, c->api.group_end());
  TMD_HIP(hipMemcpyAsync(c->cnt_host + c->world, dev + c->world, sizeof(int64_t) * c->world, hipMemcpyDeviceToHost, st));
  TMD_HIP(hipStreamSynchronize(st));
  for (int p = 0; p < c->world; ++p) recv_counts[p] = p == c->rank ? send_counts[p] : c->cnt_host[c->world + p];
  return 0;
}

}  // namespace tmd

namespace {

// per-atom index of the send list, rebuilt when the list has changed (a few short launches once per migration)
int dd_csr_ready(tmdhip_comm *c, const tmdhip_dd_desc *d, hipStream_t st) {
  if (d->nsend == 0) {  // nothing to send: every atom's row range is empty
    TMD_TRY(c->csr_off.ensure(sizeof(int) * ((size_t)d->nown + 1)));
    if (c->csr_nsend != 0 || c->csr_nown != d->nown) {
      TMD_HIP(hipMemsetAsync(c->csr_off.p, 0, sizeof(int) * ((size_t)d->nown + 1), st));
      c->csr_nsend = 0;
      c->csr_nown = d->nown;
      c->csr_index = nullptr;
    }
    TMD_TRY(c->csr_row.ensure(sizeof(int)));
    return 0;
  }
  if (c->csr_index == d->send_index_dev && c->csr_nsend == d->nsend && c->csr_nown == d->nown) return 0;
  if (d->nown + 1 >= ((int64_t)1 << 31) || d->nsend >= ((int64_t)1 << 31)) return fail("tmdhip_dd_run: brick too large");
  TMD_TRY(c->csr_off.ensure(sizeof(int) * ((size_t)d->nown + 1)));
  TMD_TRY(c->csr_cur.ensure(sizeof(int) * ((size_t)d->nown + 1)));
  TMD_TRY(c->csr_row.ensure(sizeof(int) * (size_t)d->nsend));
  TMD_HIP(hipMemsetAsync(c->csr_cur.p, 0, sizeof(int) * ((size_t)d->nown + 1), st));
  const dim3 gs((unsigned)((d->nsend + 255) / 256));
  hipLaunchKernelGGL(csr_count_kernel, gs, dim3(256), 0, st, d->nsend, d->send_index_dev, c->csr_cur.as<int>());
  // exclusive prefix over the nown + 1 counts (the last one is 0): off[0 .. nown]; the fill cursors start as a copy
  size_t tmp = 0;
  TMD_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, c->csr_cur.as<int>(), c->csr_off.as<int>(), (int)d->nown + 1, st));
  TMD_TRY(c->csr_tmp.ensure(std::max<size_t>(tmp, 16)));
  tmp = c->csr_tmp.bytes;
  TMD_HIP(hipcub::DeviceScan::ExclusiveSum(c->csr_tmp.p, tmp, c->csr_cur.as<int>(), c->csr_off.as<int>(), (int)d->nown + 1, st));
  TMD_HIP(hipMemcpyAsync(c->csr_cur.p, c->csr_off.p, sizeof(int) * ((size_t)d->nown + 1), hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(csr_fill_kernel, gs, dim3(256), 0, st, d->nsend, d->send_index_dev, c->csr_cur.as<int>(), c->csr_row.as<int>());
  TMD_HIP(hipGetLastError());
  c->csr_index = d->send_index_dev;
  c->csr_nsend = d->nsend;
  c->csr_nown = d->nown;
  return 0;
}

constexpr double kDdChainNear = 0.75;  // as tmdhip_md_run's kChainSkipNear (md_loop.hip)

// front half of a fused brick step: the pacing decision (leave the rebuild chain out?) and — unless the previous pair
// launch has made this iteration's update itself (step blocks) — dd_own_kernel
template <typename R>
int dd_fused_front(tmdhip_ctx *ctx, Replica &rp, tmdhip_comm *c, const tmdhip_dd_desc *d, int phases, uint64_t kick_step,
                   bool chain_skip_on, bool stepped, bool &pace_timed_out, bool &skip_chain, hipStream_t st) {
  using R4 = typename Vec<R>::T4;
  const double box0[3] = {0, 0, 0};
  const PairConsts<R> pc = make_consts<R>(ctx, box0);
  DdOwnArgs<R> a{};
  a.nown = d->nown;
  a.pos = (R *)d->pos_dev;
  a.vel = (R *)d->vel_dev;
  a.f = (const R *)d->forces_dev;
  a.mass = (const R *)d->mass_dev;
  a.vcoeff = (const R *)d->vcoeff_dev;
  a.dt = (R)d->dt;
  a.half_dt = (R)(0.5 * d->dt);
  a.gamma = d->vcoeff_dev ? (R)d->gamma : R(0);
  a.seed = d->seed;
  a.step = kick_step;
  a.phases = phases;
  a.ref_mig = (const R *)d->ref_dev;
  a.disp2 = d->disp2_dev;
  a.chk = make_check<R>(ctx, rp);
  skip_chain = false;
  if (chain_skip_on) {  // (tmdhip_md_run's scheme, md_loop.hip: the host stays one step behind the device)
    if (!rp.hostpub) {
      TMD_HIP(hipHostMalloc((void **)&rp.hostpub, 8 * sizeof(unsigned), hipHostMallocMapped));
      for (int w = 0; w < 8; ++w) rp.hostpub[w] = 0u;
      rp.seq = 0;
      rp.seq_valid = false;
    }
    volatile unsigned *hp = rp.hostpub;
    if (rp.seq_valid && !pace_timed_out && !wait_published(hp, rp.seq)) pace_timed_out = true;
    if (rp.seq_valid && !pace_timed_out) {
      const bool near = hp[1 + (rp.seq & 1u)] == rp.seq, rebuilt = hp[3 + (rp.seq & 1u)] == rp.seq;
      skip_chain = !near || (rebuilt && !rp.prev_skipped);
    }
    rp.prev_skipped = skip_chain;
    rp.seq += 1;
    if (rp.seq == 0) rp.seq = 1;
    a.chk.near_host = rp.hostpub + 1 + (rp.seq & 1u);
    a.chk.seq = rp.seq;
    a.chk.near_frac2 = (R)(kDdChainNear * kDdChainNear);
    a.chk.skipped = skip_chain ? 1 : 0;
    rp.seq_valid = true;
    rp.pub_ptr = rp.hostpub;
    rp.pub_val = rp.seq;
  } else {
    rp.seq_valid = false;
    rp.pub_ptr = nullptr;
  }
  // the step's displacement-test state, kept for the halo rows (dd_fused_back)
  c->chk_seq = a.chk.seq;
  c->chk_near = a.chk.near_host;
  c->chk_skipped = a.chk.skipped;
  if (stepped) return 0;  // the step blocks of the previous pair launch have done the rest
  a.sorted = rp.sorted.as<R4>();
  a.inv = rp.inv.as<int>();
  a.qs = ctx->qs.as<R>();
  TMD_TRY(dd_csr_ready(c, d, st));
  a.csr_off = c->csr_off.as<int>();
  a.csr_row = c->csr_row.as<int>();
  a.shift = (const R *)d->send_shift_dev;
  a.out = (R *)d->send_buf_dev;
  const dim3 grid((unsigned)((d->nown + 255) / 256));
  if (d->vcoeff_dev) hipLaunchKernelGGL((dd_own_kernel<R, true>), grid, dim3(256), 0, st, a, pc);
  else hipLaunchKernelGGL((dd_own_kernel<R, false>), grid, dim3(256), 0, st, a, pc);
  TMD_HIP(hipGetLastError());
  return 0;
}

// back half: the halo rows the exchange has delivered (test + cell-sorted records), then the pair launch with the
// displacement test already made and, where the front half said so, without the rebuild chain.  `fuse_next`: the pair
// launch makes the NEXT iteration's update of the owned atoms itself (step blocks behind the pair blocks, FusedStep in
// engine.h: kick of this iteration + drift of the next, the list's displacement test, the cell-sorted records, the
// migration trigger's maximum and the outgoing halo rows) — fp32 bricks on the lean kernel, never the last iteration
// of a call (its forces are wanted in `forces_dev`).  Set on return when it did.
template <typename R>
int dd_fused_back(tmdhip_ctx *ctx, Replica &rp, tmdhip_comm *c, const tmdhip_dd_desc *d, bool skip_chain, bool was_stepped,
                  bool want_fuse_next, uint64_t next_kick_step, bool &fused_next, hipStream_t st) {
  using R4 = typename Vec<R>::T4;
  const double box0[3] = {0, 0, 0};
  const PairConsts<R> pc = make_consts<R>(ctx, box0);
  const int n = ctx->d.natoms;
  ListCheck<R> chk = make_check<R>(ctx, rp);
  chk.near_host = c->chk_near;
  chk.seq = c->chk_seq;
  chk.near_frac2 = (R)(kDdChainNear * kDdChainNear);
  chk.skipped = c->chk_skipped;
  if (d->nhalo > 0) {
    hipLaunchKernelGGL((dd_halo_kernel<R>), dim3((unsigned)((d->nhalo + 255) / 256)), dim3(256), 0, st, (int)d->nown, n,
                       (const R *)d->pos_dev, chk, pc, rp.inv.as<int>(), ctx->qs.as<R>(), rp.sorted.as<R4>());
    TMD_HIP(hipGetLastError());
  }
  rp.n_compute++;
  fused_next = false;
  FusedLaunchT<R> fl{};
  {
    if (want_fuse_next && fused_step_possible<R>(ctx, rp, pc) && ctx->fused_step_timeouts == 0) {
      TMD_TRY(dd_csr_ready(c, d, st));
      FusedStaticT<R> now;
      std::memset(&now, 0, sizeof(now));
      now.s.n = n;
      now.s.vel = (R *)d->vel_dev;
      now.s.mass = (const R *)d->mass_dev;
      now.s.vcoeff = (const R *)d->vcoeff_dev;
      now.s.dt = (R)d->dt;
      now.s.half_dt = (R)(0.5 * d->dt);
      now.s.gamma = d->vcoeff_dev ? (R)d->gamma : R(0);
      now.s.seed = d->seed;
      now.s.row0 = 0;
      now.s.qs = ctx->qs.as<R>();
      now.s.inv = rp.inv.as<int>();
      now.s.chk.ref = chk.ref;
      now.s.chk.hard2 = chk.hard2;
      now.s.chk.hs2 = chk.hs2;
      now.s.chk.flags = chk.flags;
      now.s.chk.near_frac2 = (R)(kDdChainNear * kDdChainNear);
      now.s.chk.ext = chk.ext;
      now.nactive = (int)d->nown;
      now.dd_ref = (const R *)d->ref_dev;
      now.dd_disp2 = d->disp2_dev;
      now.dd_csr_off = c->csr_off.as<int>();
      now.dd_csr_row = c->csr_row.as<int>();
      now.dd_shift = (const R *)d->send_shift_dev;
      now.dd_out = (R *)d->send_buf_dev;
      TMD_TRY(upload_fused_static<R>(rp, now, st));
      fl.fst = rp.fused_dev.as<FusedStaticT<R>>();
      fl.langevin = d->vcoeff_dev != nullptr;
      fl.step.pos_in = (const R *)d->pos_dev;  // (no bonded terms: the update reads the cell-sorted records, so it
      fl.step.pos_out = (R *)d->pos_dev;       // can store the owned rows in place)
      fl.step.sorted_out = rp.sorted_alt.as<R4>();
      fl.step.noise_step = next_kick_step;
      fl.step.bonded = 0;
      if (rp.pub_ptr) {  // pacing on: the next iteration's sequence number
        unsigned nseq = rp.seq + 1;
        if (nseq == 0) nseq = 1;
        fl.step.seq = nseq;
        fl.step.near_host = rp.hostpub + 1 + (nseq & 1u);
      }
      fused_next = true;
    }
  }
  const int rc = compute_list<R>(ctx, rp, d->pos_dev, box0, d->forces_dev, nullptr,
                                 TMDHIP_WANT_FORCES | TMDHIP_OVERWRITE_FORCES | kPrechecked | (skip_chain ? kSkipChain : 0) |
                                     (skip_chain && was_stepped ? kViolationCheck : 0),
                                 st, fused_next ? &fl : nullptr);
  if (rc != 0) {
    fused_next = false;
    return rc;
  }
  if (fused_next) {
    std::swap(rp.sorted, rp.sorted_alt);
    rp.steps_in_pair_launch++;
  }
  return 0;
}

int comm_alloc_trigger(tmdhip_comm *c) {
  if (hipHostMalloc((void **)&c->host_flag, 2 * sizeof(float), hipHostMallocDefault) != hipSuccess)
    return fail("tmdhip_comm_create: pinned allocation failed");
  for (auto &e : c->ev)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail("tmdhip_comm_create: event creation failed");
  return 0;
}

}  // namespace

extern "C" {

int tmdhip_comm_unique_id(const char *librccl_path, void *id_out) {
  if (!id_out) return fail("tmdhip_comm_unique_id: null argument");
  RcclApi api;
  TMD_TRY(load_rccl(librccl_path, api));
  ncclUniqueId id;
  const ncclResult_t r = api.get_unique_id(&id);
  if (r != ncclSuccess) return fail(std::string("ncclGetUniqueId: ") + api.error_string(r));
  static_assert(sizeof(id) == TMDHIP_COMM_ID_BYTES, "unique id size");
  std::memcpy(id_out, &id, sizeof(id));
  return 0;
}

int tmdhip_comm_create(tmdhip_comm **out, const char *librccl_path, const void *id, int rank, int world) {
  if (!out || !id) return fail("tmdhip_comm_create: null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail("tmdhip_comm_create: bad rank / world size");
  tmdhip_comm *c = new tmdhip_comm();
  c->rank = rank;
  c->world = world;
  auto bail = [&](int rc) {
    tmdhip_comm_destroy(c);
    return rc;
  };
  if (load_rccl(librccl_path, c->api)) return bail(-1);
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  const ncclResult_t r = c->api.comm_init_rank(&c->comm, world, uid, rank);
  if (r != ncclSuccess) return bail(fail(std::string("ncclCommInitRank: ") + c->api.error_string(r)));
  if (comm_alloc_trigger(c)) return bail(-1);
  *out = c;
  return 0;
}

int tmdhip_local_hub_create(tmdhip_local_hub **out, int world) {
  if (!out) return fail("tmdhip_local_hub_create: null argument");
  if (world < 1 || world > 64) return fail("tmdhip_local_hub_create: world size must lie in 1..64");
  tmdhip_local_hub *h = new tmdhip_local_hub();
  h->world = world;
  h->slot.resize(world);
  for (auto &s : h->slot) {
    if (hipEventCreateWithFlags(&s.ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void **)&s.red_tmp, sizeof(float)) != hipSuccess) {
      tmdhip_local_hub_destroy(h);
      return fail("tmdhip_local_hub_create: event / buffer creation failed");
    }
  }
  *out = h;
  return 0;
}

void tmdhip_local_hub_destroy(tmdhip_local_hub *h) {
  if (!h) return;
  for (auto &s : h->slot) {
    if (s.ready) (void)hipEventDestroy(s.ready);
    if (s.done) (void)hipEventDestroy(s.done);
    if (s.red_tmp) (void)hipFree(s.red_tmp);
  }
  delete h;
}

int tmdhip_comm_create_local(tmdhip_comm **out, tmdhip_local_hub *hub, int rank) {
  if (!out || !hub) return fail("tmdhip_comm_create_local: null argument");
  if (rank < 0 || rank >= hub->world) return fail("tmdhip_comm_create_local: bad rank");
  tmdhip_comm *c = new tmdhip_comm();
  c->hub = hub;
  c->rank = rank;
  c->world = hub->world;
  if (comm_alloc_trigger(c)) {
    tmdhip_comm_destroy(c);
    return -1;
  }
  *out = c;
  return 0;
}

void tmdhip_comm_destroy(tmdhip_comm *c) {
  if (!c) return;
  if (c->comm && c->api.comm_destroy) c->api.comm_destroy(c->comm);
  for (auto &e : c->ev)
    if (e) (void)hipEventDestroy(e);
  if (c->host_flag) (void)hipHostFree(c->host_flag);
  if (c->cnt_host) (void)hipHostFree(c->cnt_host);
  c->cnt_dev.release();
  c->mig.release();
  c->csr_off.release();
  c->csr_tmp.release();
  c->csr_row.release();
  c->csr_cur.release();
  delete c;
}

int tmdhip_comm_exchange(tmdhip_comm *c, int dtype, const void *send_dev, const int64_t *send_counts_host,
                         void *recv_dev, const int64_t *recv_counts_host, int width, void *stream) {
  if (!c || !send_counts_host || !recv_counts_host) return fail("tmdhip_comm_exchange: null argument");
  if (dtype != TMDHIP_F32 && dtype != TMDHIP_F64) return fail("tmdhip_comm_exchange: bad dtype");
  if (width < 1) return fail("tmdhip_comm_exchange: bad row width");
  return exchange_rows(c, dtype, send_dev, send_counts_host, recv_dev, recv_counts_host, width, (hipStream_t)stream);
}

int tmdhip_dd_reset(tmdhip_comm *c) {
  if (!c) return fail("tmdhip_dd_reset: null argument");
  c->pending = false;
  c->at = 0;
  c->csr_index = nullptr;  // the send list changes with the migration
  c->mig_stage = 0;        // (a native migration that stopped half-way does not resume after a reset)
  return 0;
}

static int dd_run_body(tmdhip_ctx *ctx, tmdhip_comm *c, const tmdhip_dd_desc *d, int32_t *iters_done, void *stream) {
  if (!ctx || !c || !d || !iters_done) return fail("tmdhip_dd_run: null argument");
  if (d->struct_size != (int32_t)sizeof(tmdhip_dd_desc)) return fail("tmdhip_dd_run: struct_size mismatch");
  if (d->dtype != TMDHIP_F32 && d->dtype != TMDHIP_F64) return fail("tmdhip_dd_run: bad dtype");
  if (d->niter < 0 || d->nown < 0 || d->nhalo < 0 || d->nsend < 0 || d->check_every < 1 ||
      (d->first_phases != 2 && d->first_phases != 3))
    return fail("tmdhip_dd_run: bad arguments");
  if (!d->pos_dev || !d->vel_dev || !d->forces_dev || !d->mass_dev || !d->ref_dev || !d->disp2_dev ||
      !d->send_counts_host || !d->recv_counts_host)
    return fail("tmdhip_dd_run: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (c->mig_bad_pending) {  // the last tmdhip_dd_migrate: did an atom carry a type outside the caller's map?
    c->mig_bad_pending = false;
    int bad = 0;
    TMD_HIP(hipMemcpyAsync(&bad, c->mig.bad.p, sizeof(int), hipMemcpyDeviceToHost, st));
    TMD_HIP(hipStreamSynchronize(st));
    if (bad) return fail("tmdhip_dd_migrate: an atom arrived with a type outside type_map_host (the brick's LJ classes are wrong)");
  }
  const double box0[3] = {0, 0, 0};  // images are explicit halo atoms: open boundaries
  const size_t esz = d->dtype == TMDHIP_F32 ? 4 : 8;
  void *halo_rows = (char *)d->pos_dev + (size_t)d->nown * 3 * esz;
  int64_t since = d->since_migration;
  *iters_done = 0;
  // an atom outran the halo: not an error of the call but a verdict on the state (every rank reaches it at the same
  // iteration: the displacement is the maximum over the ranks) — the caller goes back to a state it saved
  auto overrun = [&](const std::string &msg) {
    last_error() = msg;
    *iters_done = 0;
    return TMDHIP_DD_OVERRUN;
  };
  auto kick_drift = [&](int phases, uint64_t kick_step) {
    return tmdhip_dd_step(d->dtype, d->nown, d->pos_dev, d->vel_dev, d->forces_dev, d->mass_dev, d->vcoeff_dev, d->dt,
                          d->gamma, d->seed, kick_step, phases, d->ref_dev, d->disp2_dev, stream);
  };
  // The three-launch step (head comment) needs a cell-list context that holds a list of exactly these rows in an open
  // box; anything else (first call after a migration without a force evaluation, all-pairs bricks) takes the loop of
  // separate launches.  TMDHIP_DD_FUSED=0 forces that loop, 1 the three-launch loop without step blocks (A/B, tests).
  Replica &rp = ctx->rep[0];
  const char *e_fused = std::getenv("TMDHIP_DD_FUSED");
  const int mode = e_fused ? std::max(0, std::min(std::atoi(e_fused), 2)) : 2;  // 0 separate launches, 1 three launches, 2 + step blocks
  const bool fused = mode >= 1 && ctx->algorithm == TMDHIP_ALGO_CELLLIST && ctx->d.terms != 0 && ctx->d.dtype == d->dtype &&
                     d->nown > 0 && (int64_t)ctx->d.natoms == d->nown + d->nhalo && ctx->d.natoms < (1 << 30) && !ctx->half_skin.p;
  bool stepped = false;  // the previous pair launch has made this iteration's update of the owned atoms (step blocks)
  const char *e_skip = std::getenv("TMDHIP_CHAIN_SKIP");
  const bool chain_skip_on = !(e_skip && std::atoi(e_skip) == 0);
  bool pace_timed_out = false;
  rp.seq_valid = false;  // whatever ran between two calls (migration, plain evaluations) published nothing
  for (int it = 0; it < d->niter; ++it) {
    // the kick belongs to the previous iteration (noise counter step0 + it - 1), the drift to this one
    const uint64_t kick_step = d->step0 + (uint64_t)it > 0 ? d->step0 + (uint64_t)it - 1 : 0;
    const int phases = it == 0 ? d->first_phases : 3;
    const bool fuse_now = fused && rp.have_list && rp.box[0] == 0 && rp.box[1] == 0 && rp.box[2] == 0;
    bool skip_chain = false;
    const bool was_stepped = stepped;
    stepped = false;
    if (was_stepped && !fuse_now) return fail("tmdhip_dd_run: the brick's list vanished between two iterations");
    if (fuse_now) {
      TMD_TRY(d->dtype == TMDHIP_F32
                  ? dd_fused_front<float>(ctx, rp, c, d, phases, kick_step, chain_skip_on, was_stepped, pace_timed_out, skip_chain, st)
                  : dd_fused_front<double>(ctx, rp, c, d, phases, kick_step, chain_skip_on, was_stepped, pace_timed_out, skip_chain, st));
    } else {
      TMD_TRY(kick_drift(phases, kick_step));
    }
    ++since;
    if (since % d->check_every == 0) {
      const double limit = 0.5 * d->skin;
      if (c->pending) {
        TMD_HIP(hipEventSynchronize(c->ev[c->cur]));  // recorded check_every steps ago
        const double moved = std::sqrt((double)c->host_flag[c->cur]);
        // the extrapolation below is a prediction; what was MEASURED must never have crossed the limit already
        // (hot atoms, a larger check_every, a changed time step): halo atoms would be missing, forces silently wrong
        if (moved > limit) {
          c->pending = false;
          rp.pub_ptr = nullptr;
          return overrun("tmdhip_dd_run: an atom moved " + std::to_string(moved) + " A since the last migration, beyond the "
                         "halo's half skin of " + std::to_string(limit) + " A, before a migration was requested: the forces of "
                         "the last steps are invalid (use a larger halo skin or a smaller check_every)");
        }
        const double ahead = 1.0 + 2.0 * (double)(since + d->check_every - c->at) / (double)c->at;
        if (moved * ahead > limit) {
          c->pending = false;
          *iters_done = it;
          rp.pub_ptr = nullptr;
          return 1;  // this iteration has drifted; the caller migrates, evaluates the forces and comes back
        }
      }
      const bool first_check = !c->pending;  // first boundary after a migration: nothing measured yet
      c->cur ^= 1;
      TMD_TRY(allreduce_max(c, d->disp2_dev, st));
      TMD_HIP(hipMemcpyAsync(&c->host_flag[c->cur], d->disp2_dev, sizeof(float), hipMemcpyDeviceToHost, st));
      TMD_HIP(hipEventRecord(c->ev[c->cur], st));
      c->pending = true;
      c->at = since;
      if (first_check) {
        // one synchronous look, so that a migration can already be requested now instead of two periods after the
        // last one (the value is the maximum over all ranks: every rank takes the same decision)
        TMD_HIP(hipEventSynchronize(c->ev[c->cur]));
        const double moved = std::sqrt((double)c->host_flag[c->cur]);
        if (moved > limit) {
          c->pending = false;
          rp.pub_ptr = nullptr;
          return overrun("tmdhip_dd_run: an atom moved " + std::to_string(moved) + " A in the first " + std::to_string(since) +
                         " steps after a migration, beyond the halo's half skin of " + std::to_string(limit) + " A");
        }
        const double ahead = 1.0 + 2.0 * (double)d->check_every / (double)since;  // until the next decision
        if (moved * ahead > limit) {
          c->pending = false;
          *iters_done = it;
          rp.pub_ptr = nullptr;
          return 1;
        }
      }
    }
    if (!fuse_now || d->nsend == 0)
      TMD_TRY(tmdhip_halo_pack(d->dtype, d->nsend, d->pos_dev, d->send_index_dev, d->send_shift_dev, d->send_buf_dev, stream));
    TMD_TRY(exchange_rows(c, d->dtype, d->send_buf_dev, d->send_counts_host, halo_rows, d->recv_counts_host, 3, st));
    if (fuse_now) {
      const bool want_next = mode == 2 && it + 1 < d->niter;
      const uint64_t next_kick = d->step0 + (uint64_t)it;  // (= the next iteration's kick_step)
      const int rc = d->dtype == TMDHIP_F32
                         ? dd_fused_back<float>(ctx, rp, c, d, skip_chain, was_stepped, want_next, next_kick, stepped, st)
                         : dd_fused_back<double>(ctx, rp, c, d, skip_chain, was_stepped, want_next, next_kick, stepped, st);
      rp.pub_ptr = nullptr;
      if (rc) return rc < 0 ? rc : fail("tmdhip_dd_run: the brick's box holds too few cells for the list path");
    } else {
      TMD_TRY(tmdhip_compute_nonbonded(ctx, 0, d->pos_dev, box0, d->forces_dev, nullptr,
                                       TMDHIP_WANT_FORCES | TMDHIP_OVERWRITE_FORCES, stream));
    }
    *iters_done = it + 1;
  }
  if (d->niter > 0 || d->first_phases == 3) {
    const uint64_t last = d->step0 + (uint64_t)d->niter;
    TMD_TRY(kick_drift(1, last > 0 ? last - 1 : 0));
  }
  return 0;
}

int tmdhip_dd_run(tmdhip_ctx *ctx, tmdhip_comm *c, const tmdhip_dd_desc *d, int32_t *iters_done, void *stream) {
  const int rc = dd_run_body(ctx, c, d, iters_done, stream);
  // in-process transport: a rank that fails leaves the others waiting at the hub's next rendezvous — wake them up now
  // (their calls fail with "broken hub") instead of after the barrier's 30-s time-out
  if (rc < 0 && c && c->hub) c->hub->abort();
  return rc;
}

int tmdhip_dd_step(int dtype, int64_t nown, void *pos, void *vel, const void *forces, const void *mass,
                   const void *vcoeff, double dt, double gamma, uint64_t seed, uint64_t step, int phases,
                   const void *ref, uint32_t *disp2_dev, void *stream) {
  if (dtype != TMDHIP_F32 && dtype != TMDHIP_F64) return fail("tmdhip_dd_step: bad dtype");
  if (nown < 0 || (phases & ~3) || !(phases & 3)) return fail("tmdhip_dd_step: bad arguments");
  if (nown == 0) return 0;
  if (!pos || !vel || !forces || !mass) return fail("tmdhip_dd_step: null pointer");
  hipStream_t st = (hipStream_t)stream;
  return dtype == TMDHIP_F32
             ? launch_dd_step<float>(nown, pos, vel, forces, mass, vcoeff, dt, gamma, seed, step, phases, ref, disp2_dev, st)
             : launch_dd_step<double>(nown, pos, vel, forces, mass, vcoeff, dt, gamma, seed, step, phases, ref, disp2_dev, st);
}

int tmdhip_halo_pack(int dtype, int64_t count, const void *pos, const int32_t *index_dev, const void *shift_dev,
                     void *out, void *stream) {
  if (dtype != TMDHIP_F32 && dtype != TMDHIP_F64) return fail("tmdhip_halo_pack: bad dtype");
  if (count < 0) return fail("tmdhip_halo_pack: negative count");
  if (count == 0) return 0;
  if (!pos || !index_dev || !shift_dev || !out) return fail("tmdhip_halo_pack: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n3 = 3 * count;
  if (dtype == TMDHIP_F32)
    hipLaunchKernelGGL((halo_pack_kernel<float>), blocks_for(n3, 256), dim3(256), 0, st, n3, (const float *)pos,
                       index_dev, (const float *)shift_dev, (float *)out);
  else
    hipLaunchKernelGGL((halo_pack_kernel<double>), blocks_for(n3, 256), dim3(256), 0, st, n3, (const double *)pos,
                       index_dev, (const double *)shift_dev, (double *)out);
  TMD_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
