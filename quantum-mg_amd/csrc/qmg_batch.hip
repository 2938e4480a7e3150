// qmg_batch.hip -- lock-step batches of independent right-hand sides ("independent right-hand sides"; up to 16
// systems per GPU advance through the same Krylov / K-cycle iteration together).
//
// Why: every coarse-level kernel of the K-cycle streams the SAME matrices (stencil) or the SAME null vectors (transfer)
// for every right-hand side.  Solving k systems one after the other reads them k times; solving them in lock step reads
// them once and turns the coarse apply into an (nc x nc).(nc x k) contraction (kernel C of qmg_stencil.hip, f64 MFMA).
//
// Layout: a batch vector is `nrhs` vectors of `n` complex128 at a common `stride` (complex elements).  Every entry point
// takes a bit mask of ACTIVE systems: a system that has converged inside an inner solve is frozen -- neither read nor
// written -- while the others continue, so each system sees exactly the iteration it would see alone.
//
// Per-system arithmetic is the single-vector kernels' (qmg_blas.hip, qmg_transfer.hip): element-wise ops are identical,
// reductions use the same block partition and the same fixed-order second stage, so a batched inner product is
// bit-identical to the unbatched one.
#include <string.h>
#include <time.h>

#include <initializer_list>

#include "qmg_common.h"

namespace qmg {

constexpr int BRED_BLOCKS = 1024;    // partials per reduction per system (= RED_BLOCKS of qmg_blas.hip)
constexpr int BDOT_MAX = 32;         // vectors per batched multidot / multi_caxpy call

// ---------------- element-wise ----------------
// T = storage scalar (double | float), W = elements per 16-byte access (1 for double; 2 for float when the arrays are
// 16-byte aligned with even n and stride, else 1).  Arithmetic is fp64 in registers for both T.
struct BatchCoef { cplx a[BATCH_MAX], b[BATCH_MAX]; };   // indexed by system id

// read-only operands of a batch whose active systems add up to `blas_nt_mb` MiB or more are streamed non-temporally (as in qmg_blas.hip:
// +8-10 % on vectors the caches cannot hold anyway); the in/out operand never is.  A run-time, launch-uniform choice.
template <typename T, int W>
__device__ __forceinline__ void ldb(const void* p, long i, cplx (&v)[W], bool nt) {
  if (nt) ldc_pack_nt<T, W>(p, i, v);
  else ldc_pack<T, W>(p, i, v);
}
static inline int batch_nt(const BatchIdx& bi, size_t n, int dtype) {
  return g_blas_nt_bytes > 0 && (long)((size_t)bi.n * n * (dtype == QMG_C32 ? 8 : 16)) >= g_blas_nt_bytes;
}

template <int OP, typename T, int W>
__global__ __launch_bounds__(BLOCK) void k_bblas(void* __restrict__ z_, const void* __restrict__ x_, const void* __restrict__ y_, const BatchCoef c,
                                                 const BatchIdx bi, long n, long stride) {
  typedef typename CStore<T>::type ct;
  const int k = bi.id[blockIdx.y];
  ct* z = reinterpret_cast<ct*>(z_) + (long)k * stride;
  const ct* x = reinterpret_cast<const ct*>(x_) + (long)k * stride;
  const ct* y = reinterpret_cast<const ct*>(y_) + (long)k * stride;
  const cplx a = c.a[k], b = c.b[k];
  const long np = n / W;
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < np; i += (long)gridDim.x * BLOCK) {
    cplx r[W], u[W], v[W];
    if (OP == QMG_BOP_CAX || OP == QMG_BOP_CAXPY || OP == QMG_BOP_CXPY) ldc_pack<T, W>(z, i, r);
    if (OP == QMG_BOP_COPY || OP == QMG_BOP_CAXPY || OP == QMG_BOP_CXPY || OP == QMG_BOP_CAXPBYZ) ldb<T, W>(x, i, u, bi.nt);
    if (OP == QMG_BOP_CAXPBYZ) ldb<T, W>(y, i, v, bi.nt && y_ != z_);   // (z = a x + b z is the common aliased use: then y is the in/out operand)
#pragma unroll
    for (int w = 0; w < W; w++) {
      if (OP == QMG_BOP_ZERO) r[w] = cmake(0.0, 0.0);
      else if (OP == QMG_BOP_COPY) r[w] = u[w];
      else if (OP == QMG_BOP_CAX) r[w] = cmul(a, r[w]);
      else if (OP == QMG_BOP_CAXPY) cmac(r[w], a, u[w]);
      else if (OP == QMG_BOP_CXPY) r[w] = cadd(r[w], u[w]);
      else { r[w] = cmul(a, u[w]); cmac(r[w], b, v[w]); }   // CAXPBYZ
    }
    stc_pack<T, W>(z, i, r);
  }
}

// y_k += sum_j a[j][k] x_j,k for up to 8 vector sets per launch (coefficients travel as kernel arguments)
constexpr int BMAXPY_J = 8;
struct BatchMultiAxpy { const void* x[BMAXPY_J]; cplx a[BMAXPY_J][BATCH_MAX]; };
// The NJ vector sets' loads are all requested (in storage form) before the first is widened and used: with a loop over nj and a branch on the
// coefficient around each load, every load waited for the one before it.  A zero coefficient means "this system does not use vector
// set j" (systems of a batch own different numbers of directions): the slot may hold stale pool memory, whose VALUE must not enter the sum
// (0 * inf = nan) -- it is loaded like the others and replaced by zero.
template <typename T, int W, int NJ, bool NT>
__device__ __forceinline__ void bmulti_caxpy_run(typename CStore<T>::type* y, const BatchMultiAxpy& m, int k, long off, long np) {
  typedef typename CStore<T>::type ct;
  typedef typename RawP<T, W>::type raw;
  cplx c[NJ];
  bool use[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++) { c[j] = m.a[j][k]; use[j] = c[j].x != 0.0 || c[j].y != 0.0; }
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < np; i += (long)gridDim.x * BLOCK) {
    raw ur[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const void* xb = reinterpret_cast<const ct*>(m.x[j]) + off;
      ur[j] = NT ? ld_rawp_nt<T, W>(xb, i) : ld_rawp<T, W>(xb, i);
    }
    cplx acc[W];
    ldc_pack<T, W>(y, i, acc);
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      cplx u[W];
      widen_rawp<T, W>(ur[j], u);
#pragma unroll
      for (int w = 0; w < W; w++) cmac(acc[w], c[j], use[j] ? u[w] : cmake(0.0, 0.0));
    }
    stc_pack<T, W>(y, i, acc);
  }
}
// the short-vector form (coarse levels: a launch of a few microseconds, where the unrolled form's longer prologue costs more than its loads gain)
template <typename T, int W>
__global__ __launch_bounds__(BLOCK) void k_bmulti_caxpy_small(void* __restrict__ y_, const BatchMultiAxpy m, int nj, const BatchIdx bi, long n, long stride) {
  typedef typename CStore<T>::type ct;
  const int k = bi.id[blockIdx.y];
  const long off = (long)k * stride;
  ct* y = reinterpret_cast<ct*>(y_) + off;
  const long np = n / W;
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < np; i += (long)gridDim.x * BLOCK) {
    cplx acc[W];
    ldc_pack<T, W>(y, i, acc);
    for (int j = 0; j < nj; j++) {
      const cplx c = m.a[j][k];
      if (c.x != 0.0 || c.y != 0.0) {   // (a zero coefficient: the slot may hold stale pool memory, which must not be read)
        cplx u[W];
        ldb<T, W>(reinterpret_cast<const ct*>(m.x[j]) + off, i, u, bi.nt);
#pragma unroll
        for (int w = 0; w < W; w++) cmac(acc[w], c, u[w]);
      }
    }
    stc_pack<T, W>(y, i, acc);
  }
}
template <typename T, int W>
__global__ __launch_bounds__(BLOCK) void k_bmulti_caxpy(void* __restrict__ y_, const BatchMultiAxpy m, int nj, const BatchIdx bi, long n, long stride) {
  typedef typename CStore<T>::type ct;
  const int k = bi.id[blockIdx.y];
  const long off = (long)k * stride;
  ct* y = reinterpret_cast<ct*>(y_) + off;
  const long np = n / W;
#define QMG_MAXPY_CASE(NJ) case NJ: if (bi.nt) bmulti_caxpy_run<T, W, NJ, true>(y, m, k, off, np); else bmulti_caxpy_run<T, W, NJ, false>(y, m, k, off, np); break;
  switch (nj) {
    QMG_MAXPY_CASE(1) QMG_MAXPY_CASE(2) QMG_MAXPY_CASE(3) QMG_MAXPY_CASE(4) QMG_MAXPY_CASE(5) QMG_MAXPY_CASE(6) QMG_MAXPY_CASE(7)
    default: if (bi.nt) bmulti_caxpy_run<T, W, 8, true>(y, m, k, off, np); else bmulti_caxpy_run<T, W, 8, false>(y, m, k, off, np); break;
  }
#undef QMG_MAXPY_CASE
}

// The flexible GCR's three vector updates of one iteration in ONE pass (bgcr_core): w += sum_j c_j W_j (Gram-Schmidt), r += a w (a = -alpha; the w
// just stored, in its storage precision), z_next = r (optional: the next search direction of an un-preconditioned GCR).  Operation for operation
// k_bmulti_caxpy_small, k_bblas<CAXPY>, k_bblas<COPY> -- the same bits -- with one launch and one read of w and r instead of three launches.
struct BatchCoefA { cplx a[BATCH_MAX]; };
template <typename T, int W>
__global__ __launch_bounds__(BLOCK) void k_bgcr_update(void* __restrict__ w_, const BatchMultiAxpy m, int nj, const BatchCoefA ar, void* __restrict__ r_,
                                                       void* __restrict__ zn_, const BatchIdx bi, long n, long stride) {
  typedef typename CStore<T>::type ct;
  const int k = bi.id[blockIdx.y];
  const long off = (long)k * stride;
  ct* wv = reinterpret_cast<ct*>(w_) + off;
  ct* rv = reinterpret_cast<ct*>(r_) + off;
  ct* zn = zn_ ? reinterpret_cast<ct*>(zn_) + off : nullptr;
  const cplx a = ar.a[k];
  const long np = n / W;
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < np; i += (long)gridDim.x * BLOCK) {
    cplx acc[W], rr[W];
    ldc_pack<T, W>(wv, i, acc);
    ldc_pack<T, W>(rv, i, rr);
    for (int j = 0; j < nj; j++) {
      const cplx c = m.a[j][k];
      if (c.x != 0.0 || c.y != 0.0) {   // (a zero coefficient: the slot may hold stale pool memory, which must not be read)
        cplx u[W];
        ldb<T, W>(reinterpret_cast<const ct*>(m.x[j]) + off, i, u, bi.nt);
#pragma unroll
        for (int q = 0; q < W; q++) cmac(acc[q], c, u[q]);
      }
    }
    if (nj > 0) stc_pack<T, W>(wv, i, acc);
#pragma unroll
    for (int q = 0; q < W; q++) {
      if (sizeof(T) == 4) acc[q] = cmake((double)(float)acc[q].x, (double)(float)acc[q].y);   // what a separate pass would read back
      cmac(rr[q], a, acc[q]);
    }
    stc_pack<T, W>(rv, i, rr);
    if (zn) stc_pack<T, W>(zn, i, rr);
  }
}

// ---------------- reductions (two-stage, deterministic; fp64: partition identical to qmg_blas.hip) ----------------
template <int NV>
__device__ __forceinline__ void bblock_reduce_store(double* v, double* partial_out) {
  __shared__ double sm[NV][BLOCK / WAVE];
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
#pragma unroll
  for (int q = 0; q < NV; q++) {
    const double w = wave_sum(v[q]);
    if (lane == 0) sm[q][wv] = w;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double t = sm[threadIdx.x][0];
#pragma unroll
    for (int w = 1; w < BLOCK / WAVE; w++) t += sm[threadIdx.x][w];
    partial_out[threadIdx.x] = t;
  }
}

// partials layout: [system slot s][block][2*width]
template <int OP, typename T, int W>
__global__ __launch_bounds__(BLOCK) void k_breduce(const void* __restrict__ x_, const void* __restrict__ y_, long n, long stride, const BatchIdx bi,
                                                   double* __restrict__ partials) {
  typedef typename CStore<T>::type ct;
  const long off = (long)bi.id[blockIdx.y] * stride;
  const ct* x = reinterpret_cast<const ct*>(x_) + off;
  const ct* y = reinterpret_cast<const ct*>(y_) + off;
  double v[2] = {0.0, 0.0};
  const long np = n / W;
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < np; i += (long)gridDim.x * BLOCK) {
    cplx av[W], bv[W];
    ldb<T, W>(x, i, av, bi.nt);
    if (OP != QMG_BRED_NORM2) ldb<T, W>(y, i, bv, bi.nt);
#pragma unroll
    for (int w = 0; w < W; w++) {
      const cplx a = av[w];
      if (OP == QMG_BRED_NORM2) { v[0] = fma(a.x, a.x, v[0]); v[0] = fma(a.y, a.y, v[0]); }
      else if (OP == QMG_BRED_DOT) {
        const cplx b = bv[w];
        v[0] = fma(a.x, b.x, v[0]); v[0] = fma(a.y, b.y, v[0]);
        v[1] = fma(a.x, b.y, v[1]); v[1] = fma(-a.y, b.x, v[1]);
      } else {
        const cplx b = bv[w];
        const double dx = a.x - b.x, dy = a.y - b.y;
        v[0] = fma(dx, dx, v[0]); v[0] = fma(dy, dy, v[0]);
      }
    }
  }
  bblock_reduce_store<2>(v, partials + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 2);
}

struct BatchPtrs { const void* x[BDOT_MAX]; };
// KT dots <x_j,k , y_k> per system in one pass over y_k
template <int KT, typename T, int W>
__global__ __launch_bounds__(BLOCK) void k_bmultidot(const BatchPtrs xs, int j0, const void* __restrict__ y_, long n, long stride, const BatchIdx bi,
                                                     double* __restrict__ partials, int jtot) {
  typedef typename CStore<T>::type ct;
  const long off = (long)bi.id[blockIdx.y] * stride;
  const ct* y = reinterpret_cast<const ct*>(y_) + off;
  double v[2 * KT];
#pragma unroll
  for (int q = 0; q < 2 * KT; q++) v[q] = 0.0;
  const long np = n / W;
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < np; i += (long)gridDim.x * BLOCK) {
    cplx bv[W];
    ldb<T, W>(y, i, bv, bi.nt);
#pragma unroll
    for (int q = 0; q < KT; q++) {
      cplx av[W];
      ldb<T, W>(reinterpret_cast<const ct*>(xs.x[j0 + q]) + off, i, av, bi.nt);
#pragma unroll
      for (int w = 0; w < W; w++) {
        const cplx a = av[w], b = bv[w];
        v[2 * q] = fma(a.x, b.x, v[2 * q]); v[2 * q] = fma(a.y, b.y, v[2 * q]);
        v[2 * q + 1] = fma(a.x, b.y, v[2 * q + 1]); v[2 * q + 1] = fma(-a.y, b.x, v[2 * q + 1]);
      }
    }
  }
  bblock_reduce_store<2 * KT>(v, partials + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 2 * jtot + 2 * j0);
}

// stage 2: block (q, s) sums the nparts partials of output q of system slot s in a fixed order
// done / flag / seq (flag != nullptr): the LAST block to finish publishes `seq` in a host-visible word after the results (system-scope
// fences on both sides of the block count), so the host can pick the results up by polling that word instead of waiting for the
// stream's completion signal (wait_results below).
__global__ __launch_bounds__(BLOCK) void k_breduce_final(const double* __restrict__ partials, int nparts, int width, const BatchIdx bi, int out_width,
                                                         double* __restrict__ out, unsigned* done = nullptr, unsigned long long* flag = nullptr,
                                                         unsigned long long seq = 0) {
  __shared__ double sm[BLOCK / WAVE];
  const int q = blockIdx.x, s = blockIdx.y;
  const double* p = partials + (long)s * nparts * width;
  double t = 0.0;
  for (int i = threadIdx.x; i < nparts; i += BLOCK) t += p[(long)i * width + q];
  t = wave_sum(t);
  if ((threadIdx.x & (WAVE - 1)) == 0) sm[threadIdx.x / WAVE] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = sm[0];
#pragma unroll
    for (int w = 1; w < BLOCK / WAVE; w++) r += sm[w];
    out[(long)bi.id[s] * out_width + q] = r;
    if (flag) {
      __threadfence_system();
      if (atomicAdd(done, 1u) == gridDim.x * gridDim.y - 1) {
        atomicExch(done, 0u);
        __threadfence_system();
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// MR(omega) update with the step length formed ON THE DEVICE from the dots the previous launches left in `dots` ([system][4]:
// Re<p,r>, Im<p,r>, <p,p>): alpha = omega conj(<p,r>)... written as krylov.hpp writes it, alpha = omega * <r,p>^* / <p,p> with
// <p,r> = conj(<r,p>); 0 when <p,p> == 0 (the host loop's breakdown `break`: the system is left as it is).
//   x (+)= alpha r_in ;  r_out = r_in - alpha p   (r_out == nullptr: not wanted; r_out may alias r_in)
// XSET: x = alpha r_in (the first step of a smoother that starts from x0 = 0: no zero-fill, no read of x).
template <typename T, int W, bool XSET>
__global__ __launch_bounds__(BLOCK) void k_bmr_update(void* __restrict__ x_, const void* __restrict__ rin_, void* rout_, const void* __restrict__ p_,
                                                      const double* __restrict__ dots, double omega, const BatchIdx bi, long n, long stride) {
  typedef typename CStore<T>::type ct;
  const int k = bi.id[blockIdx.y];
  const long off = (long)k * stride;
  ct* x = reinterpret_cast<ct*>(x_) + off;
  const ct* rin = reinterpret_cast<const ct*>(rin_) + off;
  ct* rout = rout_ ? reinterpret_cast<ct*>(rout_) + off : nullptr;
  const ct* p = reinterpret_cast<const ct*>(p_) + off;
  const double prx = dots[4 * k], pry = dots[4 * k + 1], pp = dots[4 * k + 2];   // <p,r> = (prx, pry)
  // alpha = omega * <p,r> / <p,p> in the host loop's operation order: (omega * pr) / pp
  const cplx alpha = (pp == 0.0) ? cmake(0.0, 0.0) : cmake((omega * prx) / pp, (omega * pry) / pp);
  const cplx malpha = cmake(-alpha.x, -alpha.y);
  const long np = n / W;
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < np; i += (long)gridDim.x * BLOCK) {
    cplx xv[W], rv[W], pv[W];
    ldc_pack<T, W>(rin, i, rv);
    if (!XSET) ldc_pack<T, W>(x, i, xv);
#pragma unroll
    for (int w = 0; w < W; w++) { if (XSET) xv[w] = cmake(0.0, 0.0); cmac(xv[w], alpha, rv[w]); }   // (XSET: the accumulate on an explicit zero, bit for bit)
    stc_pack<T, W>(x, i, xv);
    if (rout) {
      ldb<T, W>(p, i, pv, bi.nt);
#pragma unroll
      for (int w = 0; w < W; w++) cmac(rv[w], malpha, pv[w]);
      stc_pack<T, W>(rout, i, rv);
    }
  }
}

// stage 2 of the MR dots: out[system][4] <- {sum of partial q = 0, 1 (<r,p>: conjugated into <p,r>), q = 3 (<p,p>)}; one block per system
__global__ __launch_bounds__(BLOCK) void k_bmr_final(const double* __restrict__ partials, int nparts, const BatchIdx bi, double* __restrict__ out) {
  __shared__ double sm[3][BLOCK / WAVE];
  const int s = blockIdx.x;
  const double* p = partials + (long)s * nparts * 4;   // multidot of {r, p} against p: [part][4] = Re<r,p>, Im<r,p>, <p,p>, 0
  double t[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < nparts; i += BLOCK) { t[0] += p[(long)i * 4]; t[1] += p[(long)i * 4 + 1]; t[2] += p[(long)i * 4 + 2]; }
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const double w = wave_sum(t[q]);
    if ((threadIdx.x & (WAVE - 1)) == 0) sm[q][threadIdx.x / WAVE] = w;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    double r = sm[threadIdx.x][0];
#pragma unroll
    for (int w = 1; w < BLOCK / WAVE; w++) r += sm[threadIdx.x][w];
    // <p,r> = conj(<r,p>): the imaginary part changes sign (krylov.hpp: pr = conj(d2[0]))
    out[4 * bi.id[s] + threadIdx.x] = (threadIdx.x == 1) ? -r : r;
  }
}

struct BatchWorkspace {
  double* partials = nullptr;   // BATCH_MAX * BRED_BLOCKS * 2 * BDOT_MAX doubles (8 MiB)
  double* pinned = nullptr;     // host-pinned, device-visible results: BATCH_MAX * 2 * BDOT_MAX doubles
  double* result = nullptr;     // the same in HBM: where the results go when they are summed over ranks first
  double* mr = nullptr;         // device-resident scalars of the MR smoother: [system][4] = Re<p,r>, Im<p,r>, <p,p>, - (qmg_batch_mr_*)
  double* epi_part = nullptr;   // partials of an apply's MR epilogue ([system slot][wavefront][4]), grown on demand
  size_t epi_cap = 0;
  unsigned* done = nullptr;              // blocks of the current final stage that have stored their result (device)
  unsigned long long* flag = nullptr;    // host-pinned, coherent: sequence number of the last final stage whose results are in `pinned`
  unsigned long long seq = 0;
  int device = -1;
};
static thread_local BatchWorkspace g_bws;

static int get_bws(BatchWorkspace** out) {
  int dev = 0;
  QMG_HIP_CHECK(hipGetDevice(&dev));
  if (g_bws.device != dev) {
    QMG_HIP_CHECK(hipMalloc((void**)&g_bws.partials, sizeof(double) * BATCH_MAX * BRED_BLOCKS * 2 * BDOT_MAX));
    QMG_HIP_CHECK(hipHostMalloc((void**)&g_bws.pinned, sizeof(double) * BATCH_MAX * 2 * BDOT_MAX, hipHostMallocCoherent));
    QMG_HIP_CHECK(hipHostMalloc((void**)&g_bws.flag, 64, hipHostMallocCoherent));
    *g_bws.flag = 0;
    g_bws.seq = 0;
    QMG_HIP_CHECK(hipMalloc((void**)&g_bws.done, sizeof(unsigned)));
    QMG_HIP_CHECK(hipMemset(g_bws.done, 0, sizeof(unsigned)));
    QMG_HIP_CHECK(hipMalloc((void**)&g_bws.result, sizeof(double) * BATCH_MAX * 2 * BDOT_MAX));
    QMG_HIP_CHECK(hipMemset(g_bws.result, 0, sizeof(double) * BATCH_MAX * 2 * BDOT_MAX));
    QMG_HIP_CHECK(hipMalloc((void**)&g_bws.mr, sizeof(double) * BATCH_MAX * 4));
    QMG_HIP_CHECK(hipMemset(g_bws.mr, 0, sizeof(double) * BATCH_MAX * 4));
    g_bws.device = dev;
  }
  *out = &g_bws;
  return QMG_SUCCESS;
}

void release_batch_workspace() {   // qmg_shutdown (qmg_runtime.hip)
  if (g_bws.partials) hipFree(g_bws.partials);
  if (g_bws.result) hipFree(g_bws.result);
  if (g_bws.pinned) hipHostFree(g_bws.pinned);
  if (g_bws.mr) hipFree(g_bws.mr);
  if (g_bws.epi_part) hipFree(g_bws.epi_part);
  if (g_bws.done) hipFree(g_bws.done);
  if (g_bws.flag) hipHostFree(g_bws.flag);
  g_bws = BatchWorkspace();
}

// How a reduction's results reach the host.  A solver iteration is a chain of short kernels with one or two host decisions in it, and
// hipStreamSynchronize costs ~15 us of completion-signal handling per decision on top of the kernels' own latency (profiles/r03_n13_solve_phase.json:
// 8244 gaps of 22.8 us after k_breduce_final = 16 % of the C3 solve).  With "reduce_spin" (default 1) the final stage's last block publishes a
// sequence number in coherent host memory behind its results and the host polls that word; the stream is NOT synchronised (later launches
// are ordered behind the kernel anyway).  A poll that sees nothing for 2 s falls back to the synchronise, which also surfaces a launch error.
int g_reduce_spin = 1;
static inline bool spin_results(const BatchWorkspace* ws) { return g_reduce_spin != 0 && ws->flag != nullptr; }
static int wait_results(BatchWorkspace* ws, bool flagged, hipStream_t st) {
  if (flagged) {
    const unsigned long long want = ws->seq;
    volatile unsigned long long* f = ws->flag;
    for (long spins = 0;; spins++) {
      if (__atomic_load_n(f, __ATOMIC_ACQUIRE) == want) return QMG_SUCCESS;
      if ((spins & 0xfff) == 0xfff) {
        static thread_local double t0 = 0;
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        const double now = ts.tv_sec + 1e-9 * ts.tv_nsec;
        if (spins == 0xfff) t0 = now;
        else if (now - t0 > 2.0) break;
      }
    }
  }
  QMG_HIP_CHECK(hipStreamSynchronize(st));
  return QMG_SUCCESS;
}

double* mr_epilogue_begin(int nsys, long npart) {
  BatchWorkspace* ws;
  if (get_bws(&ws) != QMG_SUCCESS) return nullptr;
  const size_t need = (size_t)nsys * (size_t)npart * 4;
  if (ws->epi_cap < need) {
    if (ws->epi_part) { if (hipFree(ws->epi_part) != hipSuccess) return nullptr; }   // (synchronises: nothing still writes the old buffer)
    ws->epi_part = nullptr; ws->epi_cap = 0;
    if (hipMalloc((void**)&ws->epi_part, sizeof(double) * need) != hipSuccess) return nullptr;
    ws->epi_cap = need;
  }
  return ws->epi_part;
}

static unsigned bred_grid(long n) {
  long b = (n + BLOCK - 1) / BLOCK;
  if (b > BRED_BLOCKS) b = BRED_BLOCKS;
  if (b < 1) b = 1;
  return (unsigned)b;
}

int mr_epilogue_finish(const unsigned char* ids, int n, long npart, hipStream_t st) {
  BatchWorkspace* ws;
  int rc = get_bws(&ws);
  if (rc) return rc;
  if (n < 1 || n > BATCH_MAX || !ws->epi_part) return QMG_ERR_INVALID;
  BatchIdx bi;
  bi.n = n; bi.nt = 0;
  for (int k = 0; k < BATCH_MAX; k++) bi.id[k] = (k < n) ? ids[k] : (unsigned char)0;
  const bool dist = dist_reductions_on();
  if (dist) QMG_HIP_CHECK(hipMemsetAsync(ws->mr, 0, sizeof(double) * 4 * BATCH_MAX, st));   // inactive slots must not accumulate over the ranks call after call
  k_bmr_final<<<(unsigned)n, BLOCK, 0, st>>>(ws->epi_part, (int)npart, bi, ws->mr);
  QMG_LAUNCH_CHECK();
  if (dist) { rc = dist_allreduce(ws->mr, 4 * BATCH_MAX, false, st); if (rc) return rc; }
  return QMG_SUCCESS;
}

// which access width the arrays of a call allow: 2 complex<float> per 16-byte access needs 16-byte aligned bases, even
// element counts and even strides; complex<double> is always one element per access
static int pack_width(int dtype, size_t n, size_t stride, int nrhs, std::initializer_list<const void*> ptrs) {
  if (dtype != QMG_C32) return 1;
  if ((n & 1) || (nrhs > 1 && (stride & 1))) return 1;
  for (const void* p : ptrs) if (p && !aligned16(p)) return 1;
  return 2;
}

}  // namespace qmg

using namespace qmg;

// dispatch on (dtype, access width): K(T, W) expands to one kernel launch
#define QMG_DISPATCH_TW(dtype, W_, K)                         \
  do {                                                        \
    if ((dtype) == QMG_C32) { if ((W_) == 2) { K(float, 2); } else { K(float, 1); } } \
    else { K(double, 1); }                                    \
  } while (0)

extern "C" {

int qmg_batch_blas_t(int dtype, int op, const double* a, const double* b, const void* x, const void* y, void* z, size_t n, int nrhs, size_t stride,
                     unsigned mask, void* stream) {
  if (!valid_dtype(dtype) || nrhs < 1 || nrhs > BATCH_MAX || (!z && n)) return QMG_ERR_INVALID;
  if (op < QMG_BOP_ZERO || op > QMG_BOP_CAXPBYZ) return QMG_ERR_INVALID;
  if ((op == QMG_BOP_COPY || op == QMG_BOP_CAXPY || op == QMG_BOP_CXPY || op == QMG_BOP_CAXPBYZ) && !x && n) return QMG_ERR_INVALID;
  if (op == QMG_BOP_CAXPBYZ && ((!y && n) || !b)) return QMG_ERR_INVALID;
  if ((op == QMG_BOP_CAX || op == QMG_BOP_CAXPY || op == QMG_BOP_CAXPBYZ) && !a) return QMG_ERR_INVALID;
  BatchIdx bi = expand_mask(mask, nrhs);
  if (bi.n == 0 || n == 0) return QMG_SUCCESS;
  bi.nt = batch_nt(bi, n, dtype);
  BatchCoef c;
  for (int k = 0; k < BATCH_MAX; k++) {
    c.a[k] = (a && k < nrhs) ? make_double2(a[2 * k], a[2 * k + 1]) : make_double2(0.0, 0.0);
    c.b[k] = (b && k < nrhs) ? make_double2(b[2 * k], b[2 * k + 1]) : make_double2(0.0, 0.0);
  }
  const int W = pack_width(dtype, n, stride, nrhs, {x, y, z});
  dim3 grid(grid_1d(n / W), (unsigned)bi.n);
  hipStream_t st = as_stream(stream);
#define QMG_K(T, WW)                                                                                                        \
  switch (op) {                                                                                                             \
    case QMG_BOP_ZERO: k_bblas<QMG_BOP_ZERO, T, WW><<<grid, BLOCK, 0, st>>>(z, nullptr, nullptr, c, bi, (long)n, (long)stride); break;   \
    case QMG_BOP_COPY: k_bblas<QMG_BOP_COPY, T, WW><<<grid, BLOCK, 0, st>>>(z, x, nullptr, c, bi, (long)n, (long)stride); break;        \
    case QMG_BOP_CAX: k_bblas<QMG_BOP_CAX, T, WW><<<grid, BLOCK, 0, st>>>(z, nullptr, nullptr, c, bi, (long)n, (long)stride); break;    \
    case QMG_BOP_CAXPY: k_bblas<QMG_BOP_CAXPY, T, WW><<<grid, BLOCK, 0, st>>>(z, x, nullptr, c, bi, (long)n, (long)stride); break;      \
    case QMG_BOP_CXPY: k_bblas<QMG_BOP_CXPY, T, WW><<<grid, BLOCK, 0, st>>>(z, x, nullptr, c, bi, (long)n, (long)stride); break;        \
    default: k_bblas<QMG_BOP_CAXPBYZ, T, WW><<<grid, BLOCK, 0, st>>>(z, x, y, c, bi, (long)n, (long)stride); break;                   \
  }
  QMG_DISPATCH_TW(dtype, W, QMG_K);
#undef QMG_K
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}
int qmg_batch_blas(int op, const double* a, const double* b, const void* x, const void* y, void* z, size_t n, int nrhs, size_t stride,
                   unsigned mask, void* stream) {
  return qmg_batch_blas_t(QMG_C64, op, a, b, x, y, z, n, nrhs, stride, mask, stream);
}

// coeffs[(j*nrhs + k)*2 + {0,1}]: coefficient of vector set j for system k
int qmg_batch_multi_caxpy_t(int dtype, const double* coeffs, const void* const* xs, int nj, void* y, size_t n, int nrhs, size_t stride, unsigned mask,
                            void* stream) {
  if (!valid_dtype(dtype) || nrhs < 1 || nrhs > BATCH_MAX || nj < 0 || (nj > 0 && (!coeffs || !xs)) || (!y && n)) return QMG_ERR_INVALID;
  BatchIdx bi = expand_mask(mask, nrhs);
  if (bi.n == 0 || n == 0 || nj == 0) return QMG_SUCCESS;
  bi.nt = batch_nt(bi, n, dtype);
  int W = pack_width(dtype, n, stride, nrhs, {y});
  for (int j = 0; j < nj; j++) if (!xs[j]) return QMG_ERR_INVALID; else if (dtype == QMG_C32 && !aligned16(xs[j])) W = 1;
  dim3 grid(grid_1d(n / W), (unsigned)bi.n);
  for (int j0 = 0; j0 < nj; j0 += BMAXPY_J) {
    const int jj = (nj - j0 < BMAXPY_J) ? nj - j0 : BMAXPY_J;
    BatchMultiAxpy m;
    for (int j = 0; j < BMAXPY_J; j++) {
      m.x[j] = (j < jj) ? xs[j0 + j] : nullptr;
      for (int k = 0; k < BATCH_MAX; k++)
        m.a[j][k] = (j < jj && k < nrhs) ? make_double2(coeffs[((size_t)(j0 + j) * nrhs + k) * 2], coeffs[((size_t)(j0 + j) * nrhs + k) * 2 + 1])
                                         : make_double2(0.0, 0.0);
    }
    // vectors of 16 MB and more per system: all loads of a chunk in flight (same-box A/B, C5 shape: the outer flexible GCR's 4096^2 updates 4 % of the solve faster;
    // the coarse levels' few-microsecond launches 1 % slower with it, hence the threshold)
    const bool big = n * (dtype == QMG_C32 ? 8 : 16) >= ((size_t)16 << 20);
#define QMG_K(T, WW) if (big) k_bmulti_caxpy<T, WW><<<grid, BLOCK, 0, as_stream(stream)>>>(y, m, jj, bi, (long)n, (long)stride); \
                     else k_bmulti_caxpy_small<T, WW><<<grid, BLOCK, 0, as_stream(stream)>>>(y, m, jj, bi, (long)n, (long)stride)
    QMG_DISPATCH_TW(dtype, W, QMG_K);
#undef QMG_K
    QMG_LAUNCH_CHECK();
  }
  return QMG_SUCCESS;
}
int qmg_batch_multi_caxpy(const double* coeffs, const void* const* xs, int nj, void* y, size_t n, int nrhs, size_t stride, unsigned mask, void* stream) {
  return qmg_batch_multi_caxpy_t(QMG_C64, coeffs, xs, nj, y, n, nrhs, stride, mask, stream);
}

// w_k += sum_j c[j][k] ws[j]_k ; r_k += a[k] w_k ; z_next_k = r_k (z_next != NULL) for the active systems: see k_bgcr_update
int qmg_batch_gcr_update_t(int dtype, const double* coeffs, const void* const* ws, int nj, void* w, const double* a, void* r, void* z_next, size_t n, int nrhs,
                           size_t stride, unsigned mask, void* stream) {
  if (!valid_dtype(dtype) || nrhs < 1 || nrhs > BATCH_MAX || nj < 0 || (nj > 0 && (!coeffs || !ws)) || !a || ((!w || !r) && n)) return QMG_ERR_INVALID;
  if (w == r || (z_next && (z_next == w || z_next == r))) return QMG_ERR_INVALID;
  BatchIdx bi = expand_mask(mask, nrhs);
  if (bi.n == 0 || n == 0) return QMG_SUCCESS;
  // all but the last chunk of 8 vector sets: the plain multi-axpy.  Vectors of 16 MB and more per system (an outer solve on the fine lattice): every chunk
  // through it -- its long-vector form keeps a chunk's loads in flight, which is worth more there than the saved launch (C5 shape: 2 % of the solve)
  const bool big = n * (dtype == QMG_C32 ? 8 : 16) >= ((size_t)16 << 20);
  const int lead = big ? nj : (nj > BMAXPY_J) ? ((nj - 1) / BMAXPY_J) * BMAXPY_J : 0;
  if (lead > 0) { const int rc = qmg_batch_multi_caxpy_t(dtype, coeffs, ws, lead, w, n, nrhs, stride, mask, stream); if (rc) return rc; }
  bi.nt = batch_nt(bi, n, dtype);
  int W = pack_width(dtype, n, stride, nrhs, {w, r, z_next});
  const int jj = nj - lead;
  BatchMultiAxpy m;
  for (int j = 0; j < BMAXPY_J; j++) {
    m.x[j] = (j < jj) ? ws[lead + j] : nullptr;
    if (j < jj && !ws[lead + j]) return QMG_ERR_INVALID;
    if (j < jj && dtype == QMG_C32 && !aligned16(ws[lead + j])) W = 1;
    for (int k = 0; k < BATCH_MAX; k++)
      m.a[j][k] = (j < jj && k < nrhs) ? make_double2(coeffs[((size_t)(lead + j) * nrhs + k) * 2], coeffs[((size_t)(lead + j) * nrhs + k) * 2 + 1]) : make_double2(0.0, 0.0);
  }
  BatchCoefA ar;
  for (int k = 0; k < BATCH_MAX; k++) ar.a[k] = (k < nrhs) ? make_double2(a[2 * k], a[2 * k + 1]) : make_double2(0.0, 0.0);
  dim3 grid(grid_1d(n / W), (unsigned)bi.n);
#define QMG_K(T, WW) k_bgcr_update<T, WW><<<grid, BLOCK, 0, as_stream(stream)>>>(w, m, jj, ar, r, z_next, bi, (long)n, (long)stride)
  QMG_DISPATCH_TW(dtype, W, QMG_K);
#undef QMG_K
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

// out_host[2*k + {0,1}] for every ACTIVE system k (inactive entries are left untouched); returns when they are on the host (wait_results)
int qmg_batch_reduce_t(int dtype, int op, const void* x, const void* y, size_t n, int nrhs, size_t stride, unsigned mask, double* out_host, void* stream) {
  if (!valid_dtype(dtype) || nrhs < 1 || nrhs > BATCH_MAX || !x || !out_host) return QMG_ERR_INVALID;
  if (op < QMG_BRED_NORM2 || op > QMG_BRED_DIFFNORM2) return QMG_ERR_INVALID;
  if (op != QMG_BRED_NORM2 && !y) return QMG_ERR_INVALID;
  BatchIdx bi = expand_mask(mask, nrhs);
  if (bi.n == 0) return QMG_SUCCESS;
  bi.nt = batch_nt(bi, n, dtype);
  BatchWorkspace* ws;
  int rc = get_bws(&ws);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  const int W = pack_width(dtype, n, stride, nrhs, {x, y});
  const unsigned g = bred_grid((long)(n / W));
  dim3 grid(g, (unsigned)bi.n);
#define QMG_K(T, WW)                                                                                                                      \
  if (op == QMG_BRED_NORM2) k_breduce<QMG_BRED_NORM2, T, WW><<<grid, BLOCK, 0, st>>>(x, nullptr, (long)n, (long)stride, bi, ws->partials); \
  else if (op == QMG_BRED_DOT) k_breduce<QMG_BRED_DOT, T, WW><<<grid, BLOCK, 0, st>>>(x, y, (long)n, (long)stride, bi, ws->partials);      \
  else k_breduce<QMG_BRED_DIFFNORM2, T, WW><<<grid, BLOCK, 0, st>>>(x, y, (long)n, (long)stride, bi, ws->partials)
  QMG_DISPATCH_TW(dtype, W, QMG_K);
#undef QMG_K
  QMG_LAUNCH_CHECK();
  // slabs of one lattice: the per-system results are summed over the ranks before they reach the host (every rank has
  // the same active mask: the lock-step decisions are taken on these very sums)
  const bool dist = dist_reductions_on();
  const bool flagged = !dist && spin_results(ws);
  if (flagged) k_breduce_final<<<dim3(2, (unsigned)bi.n), BLOCK, 0, st>>>(ws->partials, (int)g, 2, bi, 2, ws->pinned, ws->done, ws->flag, ++ws->seq);
  else k_breduce_final<<<dim3(2, (unsigned)bi.n), BLOCK, 0, st>>>(ws->partials, (int)g, 2, bi, 2, dist ? ws->result : ws->pinned);
  QMG_LAUNCH_CHECK();
  if (dist) {
    rc = dist_allreduce(ws->result, 2 * nrhs, false, st);
    if (rc) return rc;
    QMG_HIP_CHECK(hipMemcpyAsync(ws->pinned, ws->result, sizeof(double) * 2 * nrhs, hipMemcpyDeviceToHost, st));
  }
  rc = wait_results(ws, flagged, st);
  if (rc) return rc;
  for (int s = 0; s < bi.n; s++) { out_host[2 * bi.id[s]] = ws->pinned[2 * bi.id[s]]; out_host[2 * bi.id[s] + 1] = ws->pinned[2 * bi.id[s] + 1]; }
  return QMG_SUCCESS;
}
int qmg_batch_reduce(int op, const void* x, const void* y, size_t n, int nrhs, size_t stride, unsigned mask, double* out_host, void* stream) {
  return qmg_batch_reduce_t(QMG_C64, op, x, y, n, nrhs, stride, mask, out_host, stream);
}

// out_host[(k*nj + j)*2 + {0,1}] = <xs[j]_k , y_k> for every active system k; returns when they are on the host (wait_results)
int qmg_batch_multidot_t(int dtype, const void* const* xs, int nj, const void* y, size_t n, int nrhs, size_t stride, unsigned mask, double* out_host,
                         void* stream) {
  if (!valid_dtype(dtype) || nrhs < 1 || nrhs > BATCH_MAX || nj < 1 || nj > BDOT_MAX || !xs || !y || !out_host) return QMG_ERR_INVALID;
  BatchIdx bi = expand_mask(mask, nrhs);
  if (bi.n == 0) return QMG_SUCCESS;
  bi.nt = batch_nt(bi, n, dtype);
  BatchWorkspace* ws;
  int rc = get_bws(&ws);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  BatchPtrs p;
  int W = pack_width(dtype, n, stride, nrhs, {y});
  for (int j = 0; j < BDOT_MAX; j++) {
    p.x[j] = (j < nj) ? xs[j] : nullptr;
    if (j < nj && !xs[j]) return QMG_ERR_INVALID;
    if (j < nj && dtype == QMG_C32 && !aligned16(xs[j])) W = 1;
  }
  const unsigned g = bred_grid((long)(n / W));
  dim3 grid(g, (unsigned)bi.n);
  int j0 = 0;
  while (j0 < nj) {   // same 8/4/2/1 chunking as qmg_multidot
    const int left = nj - j0;
    const int kt = left >= 8 ? 8 : left >= 4 ? 4 : left >= 2 ? 2 : 1;
#define QMG_K(T, WW)                                                                                                  \
    if (kt == 8) k_bmultidot<8, T, WW><<<grid, BLOCK, 0, st>>>(p, j0, y, (long)n, (long)stride, bi, ws->partials, nj);   \
    else if (kt == 4) k_bmultidot<4, T, WW><<<grid, BLOCK, 0, st>>>(p, j0, y, (long)n, (long)stride, bi, ws->partials, nj);   \
    else if (kt == 2) k_bmultidot<2, T, WW><<<grid, BLOCK, 0, st>>>(p, j0, y, (long)n, (long)stride, bi, ws->partials, nj); \
    else k_bmultidot<1, T, WW><<<grid, BLOCK, 0, st>>>(p, j0, y, (long)n, (long)stride, bi, ws->partials, nj)
    QMG_DISPATCH_TW(dtype, W, QMG_K);
#undef QMG_K
    j0 += kt;
    QMG_LAUNCH_CHECK();
  }
  const bool dist = dist_reductions_on();
  const bool flagged = !dist && spin_results(ws);
  if (flagged)
    k_breduce_final<<<dim3(2 * nj, (unsigned)bi.n), BLOCK, 0, st>>>(ws->partials, (int)g, 2 * nj, bi, 2 * nj, ws->pinned, ws->done, ws->flag, ++ws->seq);
  else k_breduce_final<<<dim3(2 * nj, (unsigned)bi.n), BLOCK, 0, st>>>(ws->partials, (int)g, 2 * nj, bi, 2 * nj, dist ? ws->result : ws->pinned);
  QMG_LAUNCH_CHECK();
  if (dist) {
    rc = dist_allreduce(ws->result, 2 * nj * nrhs, false, st);
    if (rc) return rc;
    QMG_HIP_CHECK(hipMemcpyAsync(ws->pinned, ws->result, sizeof(double) * 2 * nj * nrhs, hipMemcpyDeviceToHost, st));
  }
  rc = wait_results(ws, flagged, st);
  if (rc) return rc;
  for (int s = 0; s < bi.n; s++)
    memcpy(out_host + (size_t)bi.id[s] * 2 * nj, ws->pinned + (size_t)bi.id[s] * 2 * nj, sizeof(double) * 2 * nj);
  return QMG_SUCCESS;
}
int qmg_batch_multidot(const void* const* xs, int nj, const void* y, size_t n, int nrhs, size_t stride, unsigned mask, double* out_host, void* stream) {
  return qmg_batch_multidot_t(QMG_C64, xs, nj, y, n, nrhs, stride, mask, out_host, stream);
}

// ---- MR(omega) with every scalar on the device (the smoothers of the K-cycle: fixed iteration counts, no host decision) ----
// qmg_batch_mr_dots_t: <p_k, r_k> and <p_k, p_k> of the active systems into the calling thread's device slot (one pass over p and r,
// the same two-stage deterministic reduction as qmg_batch_multidot of {r, p} against p, summed over ranks under distributed
// reductions).  Nothing comes back to the host; nothing synchronises.
int qmg_batch_mr_dots_t(int dtype, const void* r, const void* p, size_t n, int nrhs, size_t stride, unsigned mask, void* stream) {
  if (!valid_dtype(dtype) || nrhs < 1 || nrhs > BATCH_MAX || !r || !p) return QMG_ERR_INVALID;
  BatchIdx bi = expand_mask(mask, nrhs);
  if (bi.n == 0) return QMG_SUCCESS;
  bi.nt = batch_nt(bi, n, dtype);
  BatchWorkspace* ws;
  int rc = get_bws(&ws);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  BatchPtrs ptr;
  for (int j = 0; j < BDOT_MAX; j++) ptr.x[j] = nullptr;
  ptr.x[0] = r; ptr.x[1] = p;
  int W = pack_width(dtype, n, stride, nrhs, {r, p});
  const unsigned g = bred_grid((long)(n / W));
  dim3 grid(g, (unsigned)bi.n);
#define QMG_K(T, WW) k_bmultidot<2, T, WW><<<grid, BLOCK, 0, st>>>(ptr, 0, p, (long)n, (long)stride, bi, ws->partials, 2)
  QMG_DISPATCH_TW(dtype, W, QMG_K);
#undef QMG_K
  QMG_LAUNCH_CHECK();
  const bool dist = dist_reductions_on();
  if (dist) QMG_HIP_CHECK(hipMemsetAsync(ws->mr, 0, sizeof(double) * 4 * BATCH_MAX, st));   // inactive slots must not accumulate over the ranks call after call
  k_bmr_final<<<(unsigned)bi.n, BLOCK, 0, st>>>(ws->partials, (int)g, bi, ws->mr);
  QMG_LAUNCH_CHECK();
  if (dist) { rc = dist_allreduce(ws->mr, 4 * BATCH_MAX, false, st); if (rc) return rc; }
  return QMG_SUCCESS;
}

// qmg_batch_mr_update_t: with alpha_k = omega <p_k, r_k> / <p_k, p_k> formed on the device from the slot qmg_batch_mr_dots_t (or a stencil
// apply with the MR epilogue) filled on this stream:   x (+)= alpha r_in ;  r_out = r_in - alpha p.
//   x_set != 0: x = alpha r_in (first step from x0 = 0: x is written, never read);  r_out == NULL: the new residual is not wanted;
//   r_out may alias r_in.  <p,p> == 0 leaves the system unchanged (alpha = 0), as the host loop's breakdown exit does.
int qmg_batch_mr_update_t(int dtype, double omega, void* x, const void* r_in, void* r_out, const void* p, int x_set, size_t n, int nrhs, size_t stride,
                          unsigned mask, void* stream) {
  if (!valid_dtype(dtype) || nrhs < 1 || nrhs > BATCH_MAX || !x || !r_in || (r_out && !p)) return QMG_ERR_INVALID;
  BatchIdx bi = expand_mask(mask, nrhs);
  if (bi.n == 0 || n == 0) return QMG_SUCCESS;
  bi.nt = batch_nt(bi, n, dtype);
  BatchWorkspace* ws;
  int rc = get_bws(&ws);
  if (rc) return rc;
  const int W = pack_width(dtype, n, stride, nrhs, {x, r_in, r_out, p});
  dim3 grid(grid_1d(n / W), (unsigned)bi.n);
  hipStream_t st = as_stream(stream);
#define QMG_K(T, WW)                                                                                                               \
  if (x_set) k_bmr_update<T, WW, true><<<grid, BLOCK, 0, st>>>(x, r_in, r_out, p, ws->mr, omega, bi, (long)n, (long)stride);        \
  else k_bmr_update<T, WW, false><<<grid, BLOCK, 0, st>>>(x, r_in, r_out, p, ws->mr, omega, bi, (long)n, (long)stride)
  QMG_DISPATCH_TW(dtype, W, QMG_K);
#undef QMG_K
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

// the slot itself (4 doubles per system: Re<p,r>, Im<p,r>, <p,p>, -) copied to the host: tests and diagnostics; synchronises
int qmg_batch_mr_read_dots(double* out_host, int nrhs, void* stream) {
  if (!out_host || nrhs < 1 || nrhs > BATCH_MAX) return QMG_ERR_INVALID;
  BatchWorkspace* ws;
  int rc = get_bws(&ws);
  if (rc) return rc;
  QMG_HIP_CHECK(hipMemcpyAsync(out_host, ws->mr, sizeof(double) * 4 * nrhs, hipMemcpyDeviceToHost, as_stream(stream)));
  QMG_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
  return QMG_SUCCESS;
}

}  // extern "C"
