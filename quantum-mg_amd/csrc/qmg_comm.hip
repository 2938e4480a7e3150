// qmg_comm.hip -- the one collective the path needs: a sum all-reduce of a small double vector
// (per-right-hand-side residual norms / inner products) over RCCL on xGMI, so that ranks holding
// different right-hand sides take the same convergence / restart decision.
// Messages are <= a few KiB: latency-bound; callers fuse every reduction of a Krylov step into ONE
// buffer and call this once per step.
//
// librccl is loaded lazily with dlopen so that single-GPU users (and the CPU build check) do not need
// it; the calls return QMG_ERR_UNSUPPORTED if it cannot be loaded.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "qmg_common.h"

namespace qmg {

// the handful of RCCL/NCCL declarations used (ABI-stable C API of rccl.h)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccessV = 0 };
enum { ncclFloat64V = 8 };   // ncclDataType_t: ncclDouble
enum { ncclSumV = 0 };       // ncclRedOp_t

typedef int (*fn_get_unique_id)(ncclUniqueId*);
typedef int (*fn_comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
typedef int (*fn_comm_destroy)(ncclComm_t);
typedef const char* (*fn_get_error_string)(int);

struct Rccl {
  void* handle = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_get_error_string get_error_string = nullptr;
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0;
  bool force = false;
  bool load() {
    if (handle) return true;
    handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!handle) handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!handle) return false;
    get_unique_id = (fn_get_unique_id)dlsym(handle, "ncclGetUniqueId");
    comm_init_rank = (fn_comm_init_rank)dlsym(handle, "ncclCommInitRank");
    all_reduce = (fn_all_reduce)dlsym(handle, "ncclAllReduce");
    comm_destroy = (fn_comm_destroy)dlsym(handle, "ncclCommDestroy");
    get_error_string = (fn_get_error_string)dlsym(handle, "ncclGetErrorString");
    return get_unique_id && comm_init_rank && all_reduce && comm_destroy;
  }
};
static Rccl g_rccl;

}  // namespace qmg

using namespace qmg;

extern "C" {

// Rank 0 calls this and ships the 128 bytes to the other ranks by any host channel (file, env, torch.distributed).
int qmg_comm_get_unique_id(void* id128) {
  if (!id128) return QMG_ERR_INVALID;
  if (!g_rccl.load()) return QMG_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (g_rccl.get_unique_id(&id) != ncclSuccessV) return QMG_ERR_HIP;
  memcpy(id128, &id, sizeof(id));
  return QMG_SUCCESS;
}

// Collective over all ranks; the device must already be selected with qmg_init(local_rank).
int qmg_comm_init(const void* id128, int world, int rank) {
  if (!id128 || world < 1 || rank < 0 || rank >= world) return QMG_ERR_INVALID;
  g_rccl.world = world;
  g_rccl.rank = rank;
  g_rccl.force = getenv("QMG_COMM_FORCE_RCCL") != nullptr;   // test hook: a 1-rank communicator still goes through RCCL
  if (world == 1 && !g_rccl.force) return QMG_SUCCESS;
  if (!g_rccl.load()) return QMG_ERR_UNSUPPORTED;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  if (g_rccl.comm_init_rank(&g_rccl.comm, world, id, rank) != ncclSuccessV) return QMG_ERR_HIP;
  return QMG_SUCCESS;
}

int qmg_comm_world(int* world, int* rank) {
  if (world) *world = g_rccl.world;
  if (rank) *rank = g_rccl.rank;
  return QMG_SUCCESS;
}

// In-place sum over ranks of n doubles in HBM, asynchronous on `stream`. world == 1: no-op.
int qmg_allreduce_sum_f64(double* buf_dev, size_t n, void* stream) {
  if (!buf_dev && n) return QMG_ERR_INVALID;
  if ((g_rccl.world == 1 && !g_rccl.force) || n == 0) return QMG_SUCCESS;
  if (!g_rccl.comm) return QMG_ERR_INVALID;
  if (g_rccl.all_reduce(buf_dev, buf_dev, n, ncclFloat64V, ncclSumV, g_rccl.comm, as_stream(stream)) != ncclSuccessV) return QMG_ERR_HIP;
  return QMG_SUCCESS;
}

int qmg_comm_finalize(void) {
  if (g_rccl.comm) { g_rccl.comm_destroy(g_rccl.comm); g_rccl.comm = nullptr; }
  g_rccl.world = 1; g_rccl.rank = 0; g_rccl.force = false;
  return QMG_SUCCESS;
}

}  // extern "C"
