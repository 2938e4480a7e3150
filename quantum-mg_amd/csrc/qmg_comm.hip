// qmg_comm.hip -- the two exchanges the path has, over RCCL on xGMI:
//   * a sum all-reduce of a small double vector (per-right-hand-side residual norms / inner products), so that ranks
//     holding different right-hand sides take the same convergence / restart decision.  Messages are <= a
//     few KiB: latency-bound; callers fuse every reduction of a Krylov step into ONE buffer and call this once per step;
//   * the halo rows of a y-slab decomposition of ONE lattice (the reference marks the spot:
//     cshift/cshift_2d.h:39-42,72,89 "Becomes MPI"): point-to-point send / recv of the first and last row of a slab to
//     the two neighbouring ranks -- xGMI is point-to-point, so this is its natural pattern (qmg_halo_exchange), plus
//     the "distributed reductions" switch that makes every reduction entry point of the library sum over ranks.
//
// Types and enumerators come from <rccl/rccl.h>; the library itself is loaded lazily with dlopen so that
// single-GPU users (and the CPU build check) do not need librccl at load time; the calls return
// QMG_ERR_UNSUPPORTED if it cannot be loaded.
//
// Rendezvous (qmg_comm_init_env): the 128-byte RCCL id travels the way the launcher already provides --
//   * QMG_COMM_ID_HEX in the environment (a parent that owns a torch.distributed store broadcasts the id and hands
//     it to its child this way: bench.py), or
//   * one TCP exchange on MASTER_ADDR : QMG_COMM_PORT (default MASTER_PORT + 1; MASTER_PORT itself belongs to the
//     launcher's own store): rank 0 listens and sends the id to each of the world-1 ranks that connect.
// Every wait is bounded (QMG_COMM_TIMEOUT_S, default 120 s): a missing peer is an error return, never a hang.
#include <arpa/inet.h>
#include <dlfcn.h>
#include <errno.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <vector>

#include <rccl/rccl.h>

#include "qmg_common.h"

namespace qmg {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclGetErrorString) get_error_string = nullptr;
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0;
  bool force = false;
  std::atomic<bool> dist_reduce{false};   // qmg_comm_set_distributed_reductions (emulated ranks: written by every rank thread)
  bool load() {
    if (handle) return true;
    handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!handle) handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!handle) return false;
    get_unique_id = (decltype(get_unique_id))dlsym(handle, "ncclGetUniqueId");
    comm_init_rank = (decltype(comm_init_rank))dlsym(handle, "ncclCommInitRank");
    all_reduce = (decltype(all_reduce))dlsym(handle, "ncclAllReduce");
    send = (decltype(send))dlsym(handle, "ncclSend");
    recv = (decltype(recv))dlsym(handle, "ncclRecv");
    group_start = (decltype(group_start))dlsym(handle, "ncclGroupStart");
    group_end = (decltype(group_end))dlsym(handle, "ncclGroupEnd");
    comm_destroy = (decltype(comm_destroy))dlsym(handle, "ncclCommDestroy");
    get_error_string = (decltype(get_error_string))dlsym(handle, "ncclGetErrorString");
    return get_unique_id && comm_init_rank && all_reduce && comm_destroy && send && recv && group_start && group_end;
  }
};
static Rccl g_rccl;

// ---------------- ranks emulated by host threads of ONE process on ONE GPU (qmg_comm_emulate_*) ----------------
// The pool's boxes have one GPU and RCCL refuses two ranks on one device, so more than one rank cannot run through RCCL here.
// To still run the SLAB LOGIC -- which row goes to which neighbour, the distributed reductions, the lock-step decisions of a
// solver on slabs -- with R > 1, R host threads can attach as ranks: the transport is then device copies between the
// threads' arrays and host-side sums, fenced by thread barriers (slow, test-only); everything above the transport
// (peer selection, buffer layout, call sequence) is the code the RCCL path runs.
struct ThreadWorld {
  int world = 0;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  long generation = 0;
  bool aborted = false;                     // a rank failed inside a collective, or a barrier timed out: every later wait returns at once
  std::vector<const void*> vec;             // [rank]: the vector whose rows the current exchange sends
  std::vector<std::vector<double>> slot;    // [rank]: reduction operands
  // Bounded (QMG_COMM_TIMEOUT_S): false = the world is aborted -- a rank that failed before a collective never arrives, and the
  // others must come back with an error, not wait for ever.
  bool barrier(int timeout_seconds) {
    std::unique_lock<std::mutex> lk(m);
    if (aborted) return false;
    const long gen = generation;
    if (++arrived == world) { arrived = 0; generation++; cv.notify_all(); return true; }
    const bool released = cv.wait_for(lk, std::chrono::seconds(timeout_seconds), [&] { return generation != gen || aborted; });
    if (!released) { aborted = true; cv.notify_all(); return false; }
    return !aborted || generation != gen;   // released by the last arrival: this barrier completed even if someone aborted right after
  }
  void abort() {
    std::lock_guard<std::mutex> lk(m);
    aborted = true;
    cv.notify_all();
  }
};
static ThreadWorld g_tw;
static int timeout_s() { const char* t = getenv("QMG_COMM_TIMEOUT_S"); const int v = t ? atoi(t) : 120; return v > 0 ? v : 120; }
static int emu_fail(const char* what) { g_tw.abort(); set_hip_error(hipErrorUnknown, what); return QMG_ERR_HIP; }
static thread_local int t_rank = -1;        // >= 0: this host thread is an emulated rank
static inline bool emulated() { return t_rank >= 0 && g_tw.world > 0; }
static inline int my_world() { return emulated() ? g_tw.world : g_rccl.world; }
static inline int my_rank() { return emulated() ? t_rank : g_rccl.rank; }

// in-place sum / max over the emulated ranks of n doubles in HBM.  Never returns between the two barriers of the pair: a local
// error is recorded, the world is aborted (every waiter comes back with an error), and the function leaves at the end.
static int emulated_allreduce(double* buf_dev, int n, bool op_max, hipStream_t st) {
  std::vector<double>& mine = g_tw.slot[t_rank];
  mine.assign((size_t)n, 0.0);
  bool good = hipMemcpyAsync(mine.data(), buf_dev, sizeof(double) * n, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
  if (!good) g_tw.abort();
  if (!g_tw.barrier(timeout_s())) return emu_fail("qmg_comm (emulated ranks): all-reduce aborted or timed out");
  std::vector<double> total((size_t)n, 0.0);
  for (int r = 0; r < g_tw.world; r++)      // rank order: every thread forms the same sum bit for bit
    for (int i = 0; i < n; i++) total[i] = (r == 0) ? g_tw.slot[r][i] : (op_max ? (g_tw.slot[r][i] > total[i] ? g_tw.slot[r][i] : total[i]) : total[i] + g_tw.slot[r][i]);
  if (!g_tw.barrier(timeout_s())) return emu_fail("qmg_comm (emulated ranks): all-reduce aborted or timed out");   // everyone has read every slot
  QMG_HIP_CHECK(hipMemcpyAsync(buf_dev, total.data(), sizeof(double) * n, hipMemcpyHostToDevice, st));
  QMG_HIP_CHECK(hipStreamSynchronize(st));   // `total` leaves scope
  return QMG_SUCCESS;
}

// For the reduction entry points (qmg_blas.hip, qmg_batch.hip): with distributed reductions on and more than one rank
// (or the forced 1-rank communicator of the tests), sum (or max) `n` doubles in HBM over the ranks, in place, on `st`.
bool dist_reductions_on() { return g_rccl.dist_reduce && (emulated() || (g_rccl.comm && (g_rccl.world > 1 || g_rccl.force))); }
int dist_allreduce(double* buf_dev, int n, bool op_max, hipStream_t st) {
  if (!dist_reductions_on() || n == 0) return QMG_SUCCESS;
  if (emulated()) return emulated_allreduce(buf_dev, n, op_max, st);
  if (g_rccl.all_reduce(buf_dev, buf_dev, (size_t)n, ncclDouble, op_max ? ncclMax : ncclSum, g_rccl.comm, st) != ncclSuccess) return QMG_ERR_HIP;
  return QMG_SUCCESS;
}

static_assert(sizeof(ncclUniqueId) == 128, "the C-ABI ships the RCCL id as 128 bytes");

static double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static bool send_all(int fd, const void* buf, size_t n) {
  const char* p = (const char*)buf;
  while (n) { const ssize_t w = send(fd, p, n, MSG_NOSIGNAL); if (w <= 0) { if (errno == EINTR) continue; return false; } p += w; n -= (size_t)w; }
  return true;
}
static bool recv_all(int fd, void* buf, size_t n, double deadline) {
  char* p = (char*)buf;
  while (n) {
    pollfd pf = {fd, POLLIN, 0};
    const double left = deadline - now_s();
    if (left <= 0 || poll(&pf, 1, (int)(left * 1000) + 1) <= 0) return false;
    const ssize_t r = recv(fd, p, n, 0);
    if (r <= 0) { if (r < 0 && errno == EINTR) continue; return false; }
    p += r; n -= (size_t)r;
  }
  return true;
}

// "QMGID" + world + rank: a stray connection (or a rank of another launch) is told apart from a peer
struct Hello { char magic[8]; int world, rank; };

static int tcp_exchange_id(ncclUniqueId* id, int world, int rank) {
  const char* addr = getenv("MASTER_ADDR") ? getenv("MASTER_ADDR") : "127.0.0.1";
  int port = getenv("QMG_COMM_PORT") ? atoi(getenv("QMG_COMM_PORT")) : (getenv("MASTER_PORT") ? atoi(getenv("MASTER_PORT")) + 1 : 29511);
  const double deadline = now_s() + timeout_s();
  char portstr[16];
  snprintf(portstr, sizeof portstr, "%d", port);
  if (rank == 0) {
    const int ls = socket(AF_INET, SOCK_STREAM, 0);
    if (ls < 0) return QMG_ERR_INVALID;
    int one = 1;
    setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    sockaddr_in sa;
    memset(&sa, 0, sizeof sa);
    sa.sin_family = AF_INET; sa.sin_addr.s_addr = htonl(INADDR_ANY); sa.sin_port = htons((uint16_t)port);
    if (bind(ls, (sockaddr*)&sa, sizeof sa) != 0 || listen(ls, world) != 0) { close(ls); set_hip_error(hipErrorUnknown, "qmg_comm: cannot listen on QMG_COMM_PORT"); return QMG_ERR_INVALID; }
    int served = 0;
    while (served < world - 1) {
      pollfd pf = {ls, POLLIN, 0};
      const double left = deadline - now_s();
      if (left <= 0 || poll(&pf, 1, (int)(left * 1000) + 1) <= 0) { close(ls); set_hip_error(hipErrorUnknown, "qmg_comm: timed out waiting for ranks"); return QMG_ERR_INVALID; }
      const int fd = accept(ls, nullptr, nullptr);
      if (fd < 0) continue;
      Hello h;
      if (recv_all(fd, &h, sizeof h, now_s() + 5.0) && !memcmp(h.magic, "QMGID\0\0", 8) && h.world == world && h.rank > 0 && h.rank < world && send_all(fd, id, sizeof *id)) served++;
      close(fd);
    }
    close(ls);
    return QMG_SUCCESS;
  }
  addrinfo hints, *res = nullptr;
  memset(&hints, 0, sizeof hints);
  hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
  if (getaddrinfo(addr, portstr, &hints, &res) != 0 || !res) { set_hip_error(hipErrorUnknown, "qmg_comm: cannot resolve MASTER_ADDR"); return QMG_ERR_INVALID; }
  int rc = QMG_ERR_INVALID;
  while (now_s() < deadline) {   // rank 0 may not be listening yet: retry the connect, bounded
    const int fd = socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) break;
    if (connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
      Hello h;
      memset(&h, 0, sizeof h);
      memcpy(h.magic, "QMGID", 5); h.world = world; h.rank = rank;
      if (send_all(fd, &h, sizeof h) && recv_all(fd, id, sizeof *id, deadline)) { close(fd); rc = QMG_SUCCESS; break; }
    }
    close(fd);
    usleep(50 * 1000);
  }
  freeaddrinfo(res);
  if (rc) set_hip_error(hipErrorUnknown, "qmg_comm: timed out fetching the RCCL id from rank 0");
  return rc;
}

static bool hex_to_id(const char* hex, ncclUniqueId* id) {
  if (strlen(hex) != 2 * sizeof(ncclUniqueId)) return false;
  for (size_t i = 0; i < sizeof(ncclUniqueId); i++) {
    unsigned v;
    if (sscanf(hex + 2 * i, "%2x", &v) != 1) return false;
    id->internal[i] = (char)v;
  }
  return true;
}

}  // namespace qmg

using namespace qmg;

extern "C" {

// Rank 0 calls this and ships the 128 bytes to the other ranks by any host channel (env, TCP, torch.distributed).
int qmg_comm_get_unique_id(void* id128) {
  if (!id128) return QMG_ERR_INVALID;
  if (!g_rccl.load()) return QMG_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (g_rccl.get_unique_id(&id) != ncclSuccess) return QMG_ERR_HIP;
  memcpy(id128, &id, sizeof(id));
  return QMG_SUCCESS;
}

// Collective over all ranks; the device must already be selected with qmg_init(local_rank).
int qmg_comm_init(const void* id128, int world, int rank) {
  if (emulated()) return (world == g_tw.world && rank == t_rank) ? QMG_SUCCESS : QMG_ERR_INVALID;   // thread ranks: nothing to set up
  if (!id128 || world < 1 || rank < 0 || rank >= world) return QMG_ERR_INVALID;
  g_rccl.world = world;
  g_rccl.rank = rank;
  g_rccl.force = getenv("QMG_COMM_FORCE_RCCL") != nullptr;   // test hook: a 1-rank communicator still goes through RCCL
  if (world == 1 && !g_rccl.force) return QMG_SUCCESS;
  if (!g_rccl.load()) return QMG_ERR_UNSUPPORTED;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  if (g_rccl.comm_init_rank(&g_rccl.comm, world, id, rank) != ncclSuccess) return QMG_ERR_HIP;
  return QMG_SUCCESS;
}

// The TCP leg of the rendezvous on its own: rank 0's 128 bytes reach every other rank (bounded waits; no GPU involved,
// so the host logic is testable with plain processes: tests/test_distributed_cpu.py).
int qmg_comm_rendezvous(void* blob128, int world, int rank) {
  if (!blob128 || world < 1 || rank < 0 || rank >= world) return QMG_ERR_INVALID;
  if (world == 1) return QMG_SUCCESS;
  return tcp_exchange_id(reinterpret_cast<ncclUniqueId*>(blob128), world, rank);
}

// qmg_comm_init with the id obtained through the launcher's rendezvous (see the header of this file).
int qmg_comm_init_env(int world, int rank) {
  if (emulated()) return (world == g_tw.world && rank == t_rank) ? QMG_SUCCESS : QMG_ERR_INVALID;
  if (world < 1 || rank < 0 || rank >= world) return QMG_ERR_INVALID;
  const bool force = getenv("QMG_COMM_FORCE_RCCL") != nullptr;
  ncclUniqueId id;
  memset(&id, 0, sizeof id);
  if (world > 1 || force) {
    if (!g_rccl.load()) return QMG_ERR_UNSUPPORTED;
    const char* hex = getenv("QMG_COMM_ID_HEX");
    if (hex) { if (!hex_to_id(hex, &id)) return QMG_ERR_INVALID; }
    else {
      if (rank == 0 && g_rccl.get_unique_id(&id) != ncclSuccess) return QMG_ERR_HIP;
      if (world > 1) { const int rc = tcp_exchange_id(&id, world, rank); if (rc) return rc; }
    }
  }
  return qmg_comm_init(&id, world, rank);
}

int qmg_comm_world(int* world, int* rank) {
  if (world) *world = my_world();
  if (rank) *rank = my_rank();
  return QMG_SUCCESS;
}

// In-place sum over ranks of n doubles in HBM, asynchronous on `stream`. world == 1: no-op.
int qmg_allreduce_sum_f64(double* buf_dev, size_t n, void* stream) {
  if (!buf_dev && n) return QMG_ERR_INVALID;
  if (emulated()) return n ? emulated_allreduce(buf_dev, (int)n, false, as_stream(stream)) : QMG_SUCCESS;
  if ((g_rccl.world == 1 && !g_rccl.force) || n == 0) return QMG_SUCCESS;
  if (!g_rccl.comm) return QMG_ERR_INVALID;
  if (g_rccl.all_reduce(buf_dev, buf_dev, n, ncclDouble, ncclSum, g_rccl.comm, as_stream(stream)) != ncclSuccess) return QMG_ERR_HIP;
  return QMG_SUCCESS;
}

// *all_ok = 1 iff every rank passed ok != 0.  One all-reduce of one double: a rank that failed locally can tell the
// others BEFORE they enter the next data collective, so all ranks leave together instead of one leaving the rest blocked.
int qmg_comm_all_ok(int ok, int* all_ok) {
  if (!all_ok) return QMG_ERR_INVALID;
  *all_ok = ok ? 1 : 0;
  if (emulated()) {
    g_tw.slot[t_rank].assign(1, ok ? 0.0 : 1.0);
    *all_ok = 0;
    if (!g_tw.barrier(timeout_s())) return emu_fail("qmg_comm (emulated ranks): all_ok aborted or timed out");
    double bad_total = 0.0;
    for (int r = 0; r < g_tw.world; r++) bad_total += g_tw.slot[r][0];
    if (!g_tw.barrier(timeout_s())) return emu_fail("qmg_comm (emulated ranks): all_ok aborted or timed out");
    *all_ok = (bad_total == 0.0) ? 1 : 0;
    return QMG_SUCCESS;
  }
  if (g_rccl.world == 1 && !g_rccl.force) return QMG_SUCCESS;
  if (!g_rccl.comm) return QMG_ERR_INVALID;
  static double* flag = nullptr;   // one rank = one device = one buffer
  if (!flag) QMG_HIP_CHECK(hipMalloc((void**)&flag, sizeof(double)));
  const double bad = ok ? 0.0 : 1.0;
  double total = 0.0;
  QMG_HIP_CHECK(hipMemcpy(flag, &bad, sizeof(double), hipMemcpyHostToDevice));
  if (g_rccl.all_reduce(flag, flag, 1, ncclDouble, ncclSum, g_rccl.comm, nullptr) != ncclSuccess) return QMG_ERR_HIP;
  QMG_HIP_CHECK(hipMemcpy(&total, flag, sizeof(double), hipMemcpyDeviceToHost));
  *all_ok = (total == 0.0) ? 1 : 0;
  return QMG_SUCCESS;
}

// on != 0: every host- or device-returning reduction of the library (norm2sq, dot, diffnorm2sq, norminf, multidot and the
// batch reductions) returns the sum (max for norminf) over all ranks -- the vectors are slabs of one lattice.  Off (the
// default): reductions are local, as for ranks that hold different right-hand sides.  Needs an initialised communicator.
int qmg_comm_set_distributed_reductions(int on) {
  if (on && !emulated() && !g_rccl.comm && (g_rccl.world > 1 || g_rccl.force)) return QMG_ERR_INVALID;
  g_rccl.dist_reduce = on != 0;   // (emulated ranks all write the same value)
  return QMG_SUCCESS;
}

// Halo rows of a y-slab: this rank holds rows [y0, y0 + Ly) of the lattice in the usual even-odd layout (Ly even, so the
// colouring of a slab is the global one).  For each of the nrhs vectors, the slab's LAST row (both parities) goes to rank+1,
// which receives it as its halo_lo (its row "-1"), and the FIRST row goes to rank-1 as that rank's halo_hi (its row "Ly").
// Rows are contiguous in the layout, so they are sent straight out of the vector: no pack kernel.
//   halo layout: [system][parity q of the halo row's sites][hr][nc] complex; halo_stride elements between systems.
// Periodic in y over the ranks; with one rank (no communicator) the rows are copied on the device -- the same wrap the
// single-domain kernels do by index.  Asynchronous on `stream`.
int qmg_halo_exchange(int dtype, const void* vec, int Lx, int Ly, int nc, void* halo_lo, void* halo_hi, int nrhs, size_t vec_stride,
                      size_t halo_stride, void* stream) {
  return qmg_halo_exchange_parity(dtype, vec, Lx, Ly, nc, halo_lo, halo_hi, nrhs, vec_stride, halo_stride, 3u, stream);
}

// The same for the rows of ONE parity only (parities: bit 0 = even sites' rows, bit 1 = odd sites' rows).  An even-odd operator piece
// reads one parity of its right-hand side: D_eo the odd rows, D_oe the even rows -- and the Schur systems' vectors are HALF-length
// (only the even half exists), so the other parity's rows must not even be addressed.
int qmg_halo_exchange_parity(int dtype, const void* vec, int Lx, int Ly, int nc, void* halo_lo, void* halo_hi, int nrhs, size_t vec_stride,
                             size_t halo_stride, unsigned parities, void* stream) {
  if (!(parities & 3u)) return QMG_SUCCESS;
  if (!valid_dtype(dtype) || !vec || !halo_lo || !halo_hi || !valid_lattice(Lx, Ly) || nc < 1 || nrhs < 1) return QMG_ERR_INVALID;
  const size_t esz = dtype_size(dtype);
  const size_t row = (size_t)(Lx / 2) * nc;                 // complex elements of one parity's row
  const size_t half = row * (size_t)Ly;
  if (nrhs > 1 && (vec_stride < 2 * half || halo_stride < 2 * row)) return QMG_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  const bool rccl = !emulated() && g_rccl.comm && (g_rccl.world > 1 || g_rccl.force);
  // ONE plan for every transport (RCCL send / recv, thread-emulated ranks, one rank): who the neighbours are and, for system k
  // and parity q, the byte offsets of the row that leaves a vector (its first / last row) and of the slot it lands in.
  const int up = (my_rank() + 1) % my_world(), down = (my_rank() + my_world() - 1) % my_world();
  const size_t row_bytes = row * esz;
  auto first_off = [&](int k, int q) { return ((size_t)k * vec_stride + (size_t)q * half) * esz; };                          // row y = 0 of parity q: becomes `down`'s halo_hi
  auto last_off = [&](int k, int q) { return ((size_t)k * vec_stride + (size_t)q * half + (size_t)(Ly - 1) * row) * esz; };   // row y = Ly - 1: becomes `up`'s halo_lo
  auto halo_off = [&](int k, int q) { return ((size_t)k * halo_stride + (size_t)q * row) * esz; };
  if (emulated()) {   // post my vector, fetch the neighbours' rows with device copies (same layout on every rank)
    bool good = hipStreamSynchronize(st) == hipSuccess;          // my rows are final
    g_tw.vec[t_rank] = vec;
    if (!good) g_tw.abort();
    if (!g_tw.barrier(timeout_s())) return emu_fail("qmg_comm (emulated ranks): halo exchange aborted or timed out");
    for (int k = 0; k < nrhs && good; k++)
      for (int q = 0; q < 2; q++) {
        if (!((parities >> q) & 1u)) continue;
        // down's LAST row is my row -1, up's FIRST row is my row Ly
        if (hipMemcpyAsync((char*)halo_lo + halo_off(k, q), (const char*)g_tw.vec[down] + last_off(k, q), row_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) good = false;
        if (hipMemcpyAsync((char*)halo_hi + halo_off(k, q), (const char*)g_tw.vec[up] + first_off(k, q), row_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) good = false;
      }
    if (hipStreamSynchronize(st) != hipSuccess) good = false;     // my copies are done ...
    if (!good) g_tw.abort();                                      // (recorded; the second barrier of the pair is still passed)
    if (!g_tw.barrier(timeout_s()) || !good) return emu_fail("qmg_comm (emulated ranks): halo exchange failed");   // ... and so are everybody's: the vectors may change again
    return QMG_SUCCESS;
  }
  const ncclDataType_t unit = ncclChar;
  if (rccl && g_rccl.group_start() != ncclSuccess) return QMG_ERR_HIP;
  int rc = QMG_SUCCESS;
  for (int k = 0; k < nrhs && rc == QMG_SUCCESS; k++)
    for (int q = 0; q < 2; q++) {
      if (!((parities >> q) & 1u)) continue;
      const char* first = (const char*)vec + first_off(k, q);
      const char* last = (const char*)vec + last_off(k, q);
      char* lo = (char*)halo_lo + halo_off(k, q);
      char* hi = (char*)halo_hi + halo_off(k, q);
      if (rccl) {
        // order within the group: everything to / from `up` first, then `down`; with two ranks up == down and the
        // matching is by order per peer, so sends and receives are issued in the same (last, first) order on both sides
        if (g_rccl.send(last, row_bytes, unit, up, g_rccl.comm, st) != ncclSuccess) rc = QMG_ERR_HIP;
        if (g_rccl.send(first, row_bytes, unit, down, g_rccl.comm, st) != ncclSuccess) rc = QMG_ERR_HIP;
        if (g_rccl.recv(lo, row_bytes, unit, down, g_rccl.comm, st) != ncclSuccess) rc = QMG_ERR_HIP;
        if (g_rccl.recv(hi, row_bytes, unit, up, g_rccl.comm, st) != ncclSuccess) rc = QMG_ERR_HIP;
      } else {   // one rank: up == down == me, the periodic wrap
        if (hipMemcpyAsync(lo, last, row_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = QMG_ERR_HIP;
        if (hipMemcpyAsync(hi, first, row_bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = QMG_ERR_HIP;
      }
    }
  if (rccl && g_rccl.group_end() != ncclSuccess) return QMG_ERR_HIP;
  return rc;
}

// Ranks emulated by host threads (see ThreadWorld above).  qmg_comm_emulate_begin(world) once, from the thread that then starts
// `world` threads; each of those calls qmg_comm_emulate_attach(rank) first and qmg_init(device) as usual; qmg_comm_emulate_end()
// after they have joined.  While attached, qmg_comm_world / qmg_halo_exchange / qmg_allreduce_sum_f64 / qmg_comm_all_ok and the
// distributed reductions see `world` ranks.
int qmg_comm_emulate_begin(int world) {
  if (world < 1 || g_tw.world != 0) return QMG_ERR_INVALID;
  g_tw.world = world; g_tw.arrived = 0; g_tw.generation = 0; g_tw.aborted = false;
  g_tw.vec.assign((size_t)world, nullptr);
  g_tw.slot.assign((size_t)world, std::vector<double>());
  return QMG_SUCCESS;
}
int qmg_comm_emulate_attach(int rank) {
  if (g_tw.world == 0 || rank < 0 || rank >= g_tw.world) return QMG_ERR_INVALID;
  t_rank = rank;
  return QMG_SUCCESS;
}
int qmg_comm_emulate_end(void) {
  g_tw.world = 0; g_tw.vec.clear(); g_tw.slot.clear();
  t_rank = -1;
  g_rccl.dist_reduce = false;
  return QMG_SUCCESS;
}

int qmg_comm_finalize(void) {
  if (emulated()) return QMG_SUCCESS;   // the emulating process ends with qmg_comm_emulate_end
  if (g_rccl.comm) { g_rccl.comm_destroy(g_rccl.comm); g_rccl.comm = nullptr; }
  g_rccl.world = 1; g_rccl.rank = 0; g_rccl.force = false; g_rccl.dist_reduce = false;
  return QMG_SUCCESS;
}

}  // extern "C"
