// Communicator of the domain decomposition (gfx950 library, internal): the RCCL / in-process transports behind
// tmdhip_comm and the exchange primitives the brick loop (domain.hip) and the migration (dd_migrate.hip) share.
// Nothing here is part of the C ABI (include/tmdhip.h declares the opaque handles).
#pragma once

#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "engine.h"

// The handful of RCCL declarations this file needs, stated locally (values and signatures of the stable NCCL 2 ABI,
// rccl.h): librccl is opened with dlopen at run time, so the library must also BUILD on a machine without the RCCL
// headers — the single-GPU paths do not depend on them at all.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclMax = 2 } ncclRedOp_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId *uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
const char *ncclGetErrorString(ncclResult_t result);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream);
}

struct RcclApi {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
};

// ---- in-process transport: all ranks of the brick grid inside ONE process on ONE device ------------------------
// What RCCL does between processes, between host threads: one thread per rank drives its brick's loop
// (tmdhip_dd_run) on a stream of its own; an exchange is a rendezvous of the threads on the host (a reusable barrier)
// around device-side copies ordered by events:
//   every rank publishes {send buffer, counts} and records `ready` on its stream (its pack kernel is in front of it);
//   barrier;  every rank makes its stream wait for the senders' `ready` events and copies its rows out of their send
//   buffers into its own halo rows, then records `done`;  barrier;  every rank makes its stream wait for the `done` of
//   the ranks that read from it, so that its next pack cannot overwrite rows still being copied.
// The host threads only enqueue; nothing waits for the device.  A rank that does not arrive within 30 s (its loop
// returned with an error) breaks the hub: every later call fails instead of hanging.
// Purpose: the library's own step loop at world 2 / 4 / 8 on a one-GPU box (tests), with the same decisions
// (migration trigger from the max over ranks) as over RCCL.
struct LocalSlot {
  const void *send = nullptr;
  const int64_t *send_counts = nullptr;
  const int64_t *counts = nullptr;  // this rank's operand of a count exchange (host)
  float *red = nullptr;          // this rank's operand of the max reduction (device)
  float *red_tmp = nullptr;      // hub-owned device word the rank reduces into before copying back
  hipEvent_t ready = nullptr, done = nullptr;
};

struct tmdhip_local_hub {
  int world = 1;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t phase = 0;
  bool broken = false;
  int attached = 0;
  std::vector<LocalSlot> slot;
  // false: somebody did not arrive (the hub is broken from then on)
  bool barrier() {
    std::unique_lock<std::mutex> lk(m);
    if (broken) return false;
    const uint64_t my = phase;
    if (++arrived == world) {
      arrived = 0;
      ++phase;
      cv.notify_all();
      return true;
    }
    if (!cv.wait_for(lk, std::chrono::seconds(30), [&] { return phase != my || broken; })) {
      broken = true;
      cv.notify_all();
      return false;
    }
    return !broken;
  }
  // a rank that leaves a hub-backed call early (an error return, a count mismatch) calls this instead of never arriving:
  // the others wake up at once with "broken" rather than after the 30-s time-out
  void abort() {
    std::lock_guard<std::mutex> lk(m);
    broken = true;
    cv.notify_all();
  }
};

// halo-exchange communicator of one rank + the state of the asynchronous migration trigger
struct tmdhip_comm {
  RcclApi api;                       // RCCL transport (hub == nullptr)
  ncclComm_t comm = nullptr;
  tmdhip_local_hub *hub = nullptr;   // in-process transport
  int rank = 0, world = 1;
  // displacement read-back: two pinned slots / events used alternately; `pending` = slot `cur` holds the
  // maximum squared displacement measured `at` steps after the last migration
  float *host_flag = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  int cur = 0;
  bool pending = false;
  int64_t at = 0;
  // per-atom index of the brick's send list (dd_own_kernel writes an atom's outgoing rows itself): valid until the
  // next migration (tmdhip_dd_reset) for the send list it was built from
  tmd::DevBuf csr_off, csr_row, csr_cur, csr_tmp;
  const void *csr_index = nullptr;
  int64_t csr_nsend = -1, csr_nown = -1;
  // count exchanges over RCCL: device staging + pinned landing zone (2 x world words each)
  tmd::DevBuf cnt_dev;
  int64_t *cnt_host = nullptr;
  // scratch of the migration (dd_migrate.hip)
  struct MigScratch {
    tmd::DevBuf dest, counts, rows_out, rows_in, keys_in, keys_out, perm_in, perm_out, sort_tmp, mask, blk_cnt, msg_tot, halo_in,
        typemap, bad;
    int64_t *host = nullptr;  // pinned: counts read back from the device
    void release() {
      for (tmd::DevBuf *b : {&dest, &counts, &rows_out, &rows_in, &keys_in, &keys_out, &perm_in, &perm_out, &sort_tmp, &mask, &blk_cnt,
                             &msg_tot, &halo_in, &typemap, &bad})
        b->release();
      if (host) (void)hipHostFree(host);
      host = nullptr;
    }
  } mig;
  int mig_stage = 0;  // where a tmdhip_dd_migrate that returned 2 resumes (0: start)
  bool mig_bad_pending = false;  // the last migration's "atom type outside the map" flag has not been looked at yet
  int64_t mig_nnew = 0, mig_nsend = 0, mig_nhalo = 0, mig_send_counts[64] = {0}, mig_recv_counts[64] = {0};
  // displacement-test state of the brick step in flight (dd_fused_front -> dd_fused_back)
  unsigned chk_seq = 0;
  unsigned *chk_near = nullptr;
  int chk_skipped = 0;
};

#define TMD_NCCL(c, expr)                                                                                   \
  do {                                                                                                      \
    ncclResult_t _r = (expr);                                                                               \
    if (_r != ncclSuccess) return ::tmd::fail(std::string(#expr) + ": " + (c)->api.error_string(_r));       \
  } while (0)

namespace tmd {
// one grouped exchange of rows of `width` elements (tmdhip_comm_exchange's layout) on `st`
int exchange_rows(tmdhip_comm *c, int dtype, const void *send, const int64_t *send_counts, void *recv,
                  const int64_t *recv_counts, int width, hipStream_t st);
// in-place maximum over the ranks of one float on the device
int allreduce_max(tmdhip_comm *c, void *buf, hipStream_t st);
// recv_counts[p] = what rank p's send_counts holds for this rank (host arrays of `world` entries; synchronises `st`)
int exchange_counts(tmdhip_comm *c, const int64_t *send_counts, int64_t *recv_counts, hipStream_t st);
}  // namespace tmd
