// qmg_runtime.hip -- device/runtime plumbing of the C-ABI (no compute).
#include <stdio.h>
#include <string.h>

#include <string>

#include "qmg_common.h"

namespace qmg {
int g_malloc_poison = 0;   // tuning knob "malloc_poison"
static thread_local char g_err[512] = "";
void set_hip_error(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
}
}  // namespace qmg

using namespace qmg;

extern "C" {

const char* qmg_version(void) { return "qmg-hip 0.1 (gfx950)"; }

const char* qmg_last_hip_error(void) { return g_err; }

const char* qmg_status_string(int s) {
  switch (s) {
    case QMG_SUCCESS: return "success";
    case QMG_ERR_INVALID: return "invalid argument";
    case QMG_ERR_HIP: return "HIP runtime error";
    case QMG_ERR_UNSUPPORTED: return "unsupported";
    case QMG_ERR_NO_DEVICE: return "no GPU device";
    default: return "unknown status";
  }
}

int qmg_device_count(int* n) {
  if (!n) return QMG_ERR_INVALID;
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { set_hip_error(e, "hipGetDeviceCount"); *n = 0; return QMG_ERR_NO_DEVICE; }
  *n = c;
  return QMG_SUCCESS;
}

int qmg_init(int device) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) return QMG_ERR_NO_DEVICE;
  if (device < 0 || device >= c) return QMG_ERR_INVALID;
  QMG_HIP_CHECK(hipSetDevice(device));
  // QMG_TUNING="key=value,key=value": qmg_set_tuning for programs that do not call it themselves (A/B runs of the drivers);
  // an unknown key is an error, not a silently ignored typo
  if (const char* t = getenv("QMG_TUNING")) {
    std::string all(t);
    size_t pos = 0;
    while (pos < all.size()) {
      size_t end = all.find(',', pos);
      if (end == std::string::npos) end = all.size();
      const std::string item = all.substr(pos, end - pos);
      const size_t eq = item.find('=');
      if (eq == std::string::npos || qmg_set_tuning(item.substr(0, eq).c_str(), atoi(item.c_str() + eq + 1)) != QMG_SUCCESS) return QMG_ERR_INVALID;
      pos = end + 1;
    }
  }
  return QMG_SUCCESS;
}

int qmg_malloc(void** p, size_t bytes) {
  if (!p) return QMG_ERR_INVALID;
  *p = nullptr;
  if (bytes == 0) return QMG_SUCCESS;
  QMG_HIP_CHECK(hipMalloc(p, bytes));
  // "malloc_poison": every byte 0xFF -- a complex<double> / complex<float> / complex<half> array then reads as NaNs, so a
  // vector that is read before it is written shows up deterministically (in residuals, not as box-dependent behaviour)
  if (g_malloc_poison) QMG_HIP_CHECK(hipMemset(*p, 0xFF, bytes));
  return QMG_SUCCESS;
}

int qmg_free(void* p) {
  if (!p) return QMG_SUCCESS;
  QMG_HIP_CHECK(hipFree(p));
  return QMG_SUCCESS;
}

// Ordered teardown for the calling host thread: wait for the device, then release the library's per-thread workspaces
// (reduction partials, pinned result slots).  Programs call it once per thread that used the library, after their last
// call; the library stays usable (workspaces are re-created on demand).
int qmg_shutdown(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) return QMG_SUCCESS;   // nothing was ever initialised
  QMG_HIP_CHECK(hipDeviceSynchronize());
  release_blas_workspace();
  release_batch_workspace();
  release_stencil_workspace();
  QMG_HIP_CHECK(hipDeviceSynchronize());
  return QMG_SUCCESS;
}

int qmg_mem_info(size_t* free_bytes, size_t* total_bytes) {
  size_t f = 0, t = 0;
  QMG_HIP_CHECK(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return QMG_SUCCESS;
}

static int do_copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, void* stream) {
  if (bytes == 0) return QMG_SUCCESS;
  if (!dst || !src) return QMG_ERR_INVALID;
  if (stream) QMG_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, kind, as_stream(stream)));
  else QMG_HIP_CHECK(hipMemcpy(dst, src, bytes, kind));
  return QMG_SUCCESS;
}
int qmg_memcpy_h2d(void* d, const void* s, size_t n, void* st) { return do_copy(d, s, n, hipMemcpyHostToDevice, st); }
int qmg_memcpy_d2h(void* d, const void* s, size_t n, void* st) { return do_copy(d, s, n, hipMemcpyDeviceToHost, st); }
int qmg_memcpy_d2d(void* d, const void* s, size_t n, void* st) {
  if (n == 0) return QMG_SUCCESS;
  if (!d || !s) return QMG_ERR_INVALID;
  QMG_HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, as_stream(st)));
  return QMG_SUCCESS;
}
int qmg_memset_zero(void* p, size_t n, void* st) {
  if (n == 0) return QMG_SUCCESS;
  if (!p) return QMG_ERR_INVALID;
  QMG_HIP_CHECK(hipMemsetAsync(p, 0, n, as_stream(st)));
  return QMG_SUCCESS;
}

int qmg_stream_create(void** s) {
  if (!s) return QMG_ERR_INVALID;
  hipStream_t st;
  QMG_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *s = (void*)st;
  return QMG_SUCCESS;
}
int qmg_stream_destroy(void* s) {
  if (s) QMG_HIP_CHECK(hipStreamDestroy(as_stream(s)));
  return QMG_SUCCESS;
}
int qmg_stream_sync(void* s) {
  QMG_HIP_CHECK(hipStreamSynchronize(as_stream(s)));
  return QMG_SUCCESS;
}
int qmg_event_create(void** ev) {
  if (!ev) return QMG_ERR_INVALID;
  hipEvent_t e;
  QMG_HIP_CHECK(hipEventCreate(&e));
  *ev = (void*)e;
  return QMG_SUCCESS;
}
int qmg_event_destroy(void* ev) {
  if (ev) QMG_HIP_CHECK(hipEventDestroy((hipEvent_t)ev));
  return QMG_SUCCESS;
}
int qmg_event_record(void* ev, void* stream) {
  if (!ev) return QMG_ERR_INVALID;
  QMG_HIP_CHECK(hipEventRecord((hipEvent_t)ev, as_stream(stream)));
  return QMG_SUCCESS;
}
// make everything submitted to `stream` after this call wait for `ev` (cross-stream dependency without a host sync)
int qmg_stream_wait_event(void* stream, void* ev) {
  if (!ev) return QMG_ERR_INVALID;
  QMG_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), (hipEvent_t)ev, 0));
  return QMG_SUCCESS;
}
int qmg_event_elapsed_ms(void* a, void* b, float* ms) {
  if (!a || !b || !ms) return QMG_ERR_INVALID;
  QMG_HIP_CHECK(hipEventSynchronize((hipEvent_t)b));
  QMG_HIP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
  return QMG_SUCCESS;
}

}  // extern "C"
