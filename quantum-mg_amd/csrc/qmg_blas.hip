// qmg_blas.hip -- device BLAS-1 leaves and global reductions of the multigrid hot path.
//
// The reference takes these from quantum-linalg (absent; semantics inferred from call sites,
// SURVEY 2.2): caxpy/cxpay/caxpbyz/... are single streaming passes; norm2sq/dot/diffnorm2sq are
// the Krylov inner products (stateful_multigrid.h:880,884,904 and every solver iteration).
//
// Reductions are two-stage and DETERMINISTIC: each lane accumulates a grid-strided slice in
// fp64, a DPP/shuffle butterfly sums the wavefront, LDS sums the 4 wavefronts of the block,
// the block writes one partial, and a second tiny launch sums the (fixed number of) partials
// in a fixed order -- no float atomics, so two runs give bit-identical inner products.
#include <string.h>

#include "qmg_common.h"

namespace qmg {

constexpr int RED_BLOCKS = 1024;    // partials per reduction (4 blocks per CU)
constexpr int RED_MAXK = 64;        // multidot width

// vectors of at least this many bytes are streamed with non-temporal loads (tuning key "blas_nt_mb", in MiB; 0 = never)
long g_blas_nt_bytes = 256l << 20;
static inline bool blas_nt(size_t n_complex) { return g_blas_nt_bytes > 0 && (long)(n_complex * sizeof(cplx)) >= g_blas_nt_bytes; }

// ---------------- streaming BLAS-1 ----------------
enum BlasOp { OP_ZERO, OP_COPY, OP_CAX, OP_CAXY, OP_CAXPY, OP_CXPY, OP_CXPAY, OP_CAXPBY, OP_CXPYZ, OP_CAXPBYZ };

// NT: the operands are read with the non-temporal hint.  A five-stream read of 12 GB runs at 6.96 TB/s with it and 6.37 without
// (tools/membw4.hip; sc0 / sc1 make no difference), and a vector larger than the 256 MB Infinity Cache is gone before anyone reads it again
// anyway -- so the launchers ask for it from `g_blas_nt_bytes` per vector upwards and leave smaller vectors to the caches.
template <bool NT> __device__ __forceinline__ cplx ldx(const cplx* p, long i) { return NT ? ldc_nt<double>(p, i) : p[i]; }

template <int OP, bool NT>
__global__ __launch_bounds__(BLOCK) void k_blas(cplx* __restrict__ z, const cplx* __restrict__ x, const cplx* __restrict__ y,
                                                cplx a, cplx b, long n) {
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    cplx r;
    if (OP == OP_ZERO) r = cmake(0.0, 0.0);
    else if (OP == OP_COPY) r = ldx<NT>(x, i);
    else if (OP == OP_CAX) r = cmul(a, z[i]);                      // (z is written back: a non-temporal read of a line about to be stored loses 3 %)
    else if (OP == OP_CAXY) r = cmul(a, ldx<NT>(x, i));
    else if (OP == OP_CAXPY) { r = z[i]; cmac(r, a, ldx<NT>(x, i)); }
    else if (OP == OP_CXPY) r = cadd(z[i], ldx<NT>(x, i));
    else if (OP == OP_CXPAY) { r = ldx<NT>(x, i); cmac(r, a, z[i]); }
    else if (OP == OP_CAXPBY) { r = cmul(b, z[i]); cmac(r, a, ldx<NT>(x, i)); }
    else if (OP == OP_CXPYZ) r = cadd(ldx<NT>(x, i), ldx<NT>(y, i));
    else { r = cmul(a, ldx<NT>(x, i)); cmac(r, b, ldx<NT>(y, i)); }
    if (NT) { __builtin_nontemporal_store(r.x, &z[i].x); __builtin_nontemporal_store(r.y, &z[i].y); }   // (out-of-place results +4 %, in-place neutral)
    else z[i] = r;
  }
}

// y += sum_i a_i x_i for up to 32 vectors in ONE pass (GCR orthogonalisation: 2k separate caxpy launches -> 2)
constexpr int MAXPY_K = 32;
struct MultiAxpy { const cplx* x[MAXPY_K]; cplx a[MAXPY_K]; };
// NJ vectors' elements requested together, then accumulated in order j (the sum's order is that of the plain loop: bit-identical results).  A loop of one
// load and one multiply-add per trip kept a single 16-byte load in flight per lane: 0.58 of the HBM rate on the outer GCR's 4096^2 updates.
template <bool NT, int NJ>
__device__ __forceinline__ void maxpy_chunk(cplx& acc, const MultiAxpy& m, int j0, long i) {
  cplx u[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++) u[j] = ldx<NT>(m.x[j0 + j], i);
#pragma unroll
  for (int j = 0; j < NJ; j++) cmac(acc, m.a[j0 + j], u[j]);
}
template <bool NT>
__global__ __launch_bounds__(BLOCK) void k_multi_caxpy(cplx* __restrict__ y, MultiAxpy m, int k, long n) {
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    cplx acc = y[i];
    int j = 0;
    for (; j + 8 <= k; j += 8) maxpy_chunk<NT, 8>(acc, m, j, i);
    if (k - j >= 4) { maxpy_chunk<NT, 4>(acc, m, j, i); j += 4; }
    if (k - j >= 2) { maxpy_chunk<NT, 2>(acc, m, j, i); j += 2; }
    if (k - j >= 1) maxpy_chunk<NT, 1>(acc, m, j, i);
    y[i] = acc;
  }
}

struct Pattern { double scale[64]; int shuffle[64]; };
__global__ __launch_bounds__(BLOCK) void k_pattern(cplx* __restrict__ y, const cplx* __restrict__ x, long nsite, int nc, Pattern pat) {
  const long n = nsite * nc;
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const long site = i / nc;
    const int c = (int)(i - site * nc);
    const cplx v = x[site * nc + pat.shuffle[c]];
    y[i] = cmake(pat.scale[c] * v.x, pat.scale[c] * v.y);
  }
}

// counter-based Gaussian: splitmix64 -> two uniforms -> Box-Muller; element i depends only on (seed, i)
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ __launch_bounds__(BLOCK) void k_gaussian(cplx* __restrict__ x, long n, unsigned long long seed) {
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const unsigned long long h1 = splitmix64(seed * 0xD1342543DE82EF95ull + 2ull * (unsigned long long)i);
    const unsigned long long h2 = splitmix64(h1 + 2ull * (unsigned long long)i + 1ull);
    const double u1 = ((double)(h1 >> 11) + 1.0) * (1.0 / 9007199254740992.0);   // (0,1]
    const double u2 = (double)(h2 >> 11) * (1.0 / 9007199254740992.0);
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincos(6.283185307179586476925286766559 * u2, &sn, &cs);
    x[i] = cmake(rad * cs, rad * sn);
  }
}

// the same numbers for the rows [y0, y0 + Ly_l) of a (cv) vector over the lattice Lx x Ly_g: element i of the slab takes the value
// the element at its GLOBAL position would get from k_gaussian -- a slab-decomposed run draws the single-domain run's vectors
__global__ __launch_bounds__(BLOCK) void k_gaussian_slab(cplx* __restrict__ x, int hr, int Ly_g, int y0, int Ly_l, int nc, unsigned long long seed) {
  const long row = (long)hr * nc, n = 2 * row * Ly_l;
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const long q = i / (row * Ly_l), rem = i - q * row * Ly_l;          // parity, offset inside the slab's half
    const unsigned long long gi = (unsigned long long)((q * Ly_g + y0) * row + rem);
    const unsigned long long h1 = splitmix64(seed * 0xD1342543DE82EF95ull + 2ull * gi);
    const unsigned long long h2 = splitmix64(h1 + 2ull * gi + 1ull);
    const double u1 = ((double)(h1 >> 11) + 1.0) * (1.0 / 9007199254740992.0);   // (0,1]
    const double u2 = (double)(h2 >> 11) * (1.0 / 9007199254740992.0);
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincos(6.283185307179586476925286766559 * u2, &sn, &cs);
    x[i] = cmake(rad * cs, rad * sn);
  }
}

// ---------------- reductions ----------------
enum RedOp { RED_NORM2, RED_DOT, RED_DIFFNORM2, RED_NORMINF };

template <int NV, bool MAX>
__device__ __forceinline__ void block_reduce_store(double* v, double* partial_out) {
  __shared__ double sm[NV][BLOCK / WAVE];
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
#pragma unroll
  for (int q = 0; q < NV; q++) {
    const double w = MAX ? wave_max(v[q]) : wave_sum(v[q]);
    if (lane == 0) sm[q][wv] = w;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double t = sm[threadIdx.x][0];
#pragma unroll
    for (int w = 1; w < BLOCK / WAVE; w++) t = MAX ? fmax(t, sm[threadIdx.x][w]) : t + sm[threadIdx.x][w];
    partial_out[threadIdx.x] = t;
  }
}

template <int OP, bool NT>
__global__ __launch_bounds__(BLOCK) void k_reduce(const cplx* __restrict__ x, const cplx* __restrict__ y, long n, double* __restrict__ partials) {
  double v[2] = {0.0, 0.0};
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const cplx a = ldx<NT>(x, i);
    if (OP == RED_NORM2) { v[0] = fma(a.x, a.x, v[0]); v[0] = fma(a.y, a.y, v[0]); }
    else if (OP == RED_DOT) {
      const cplx b = ldx<NT>(y, i);
      v[0] = fma(a.x, b.x, v[0]); v[0] = fma(a.y, b.y, v[0]);
      v[1] = fma(a.x, b.y, v[1]); v[1] = fma(-a.y, b.x, v[1]);
    } else if (OP == RED_DIFFNORM2) {
      const cplx b = ldx<NT>(y, i);
      const double dx = a.x - b.x, dy = a.y - b.y;
      v[0] = fma(dx, dx, v[0]); v[0] = fma(dy, dy, v[0]);
    } else {
      v[0] = fmax(v[0], a.x * a.x + a.y * a.y);
    }
  }
  block_reduce_store<2, OP == RED_NORMINF>(v, partials + 2 * blockIdx.x);
}

// stage 2: block q sums the nparts partials of output q in a fixed order (grid = width: a 32-vector multidot has 64
// outputs and they are reduced concurrently, not one after the other)
template <bool MAX, bool SQRT>
__global__ __launch_bounds__(BLOCK) void k_reduce_final(const double* __restrict__ partials, int nparts, int stride, int width, double* __restrict__ out) {
  __shared__ double sm[BLOCK / WAVE];
  const int q = blockIdx.x;
  if (q >= width) return;
  double t = 0.0;
  for (int i = threadIdx.x; i < nparts; i += BLOCK) t = MAX ? fmax(t, partials[(long)i * stride + q]) : t + partials[(long)i * stride + q];
  t = MAX ? wave_max(t) : wave_sum(t);
  if ((threadIdx.x & (WAVE - 1)) == 0) sm[threadIdx.x / WAVE] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = sm[0];
#pragma unroll
    for (int w = 1; w < BLOCK / WAVE; w++) r = MAX ? fmax(r, sm[w]) : r + sm[w];
    out[q] = SQRT ? sqrt(r) : r;
  }
}

// gaussian_wall_source (reductions/reductions.h:90-162): a REAL Gaussian (mean + deviation N(0,1), imaginary part 0) on the elements with
// y == timeslice and component == color, zero everywhere else.  The draw of element i is the real Box-Muller value k_gaussian gives it.
__global__ __launch_bounds__(BLOCK) void k_wall_source(cplx* __restrict__ x, int hr, int Ly, int nc, int timeslice, int color, unsigned long long seed, double deviation, double mean) {
  const long n = 2l * hr * Ly * nc;
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const long site = i / nc;
    const int c = (int)(i - site * nc);
    const int y = (int)((site / hr) % Ly);   // site = (y + p Ly) hr + j
    cplx v = cmake(0.0, 0.0);
    if (c == color && y == timeslice) {
      const unsigned long long h1 = splitmix64(seed * 0xD1342543DE82EF95ull + 2ull * (unsigned long long)i);
      const unsigned long long h2 = splitmix64(h1 + 2ull * (unsigned long long)i + 1ull);
      const double u1 = ((double)(h1 >> 11) + 1.0) * (1.0 / 9007199254740992.0);
      const double u2 = (double)(h2 >> 11) * (1.0 / 9007199254740992.0);
      v.x = mean + deviation * (sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2));
    }
    x[i] = v;
  }
}

struct MultiPtrs { const cplx* x[RED_MAXK]; };
// k dots <x_i, y> in one pass over y: y[i] is loaded once per element and reused for all k vectors.
template <int KT, bool NT>
__global__ __launch_bounds__(BLOCK) void k_multidot(MultiPtrs xs, int k0, const cplx* __restrict__ y, long n, double* __restrict__ partials, int ktot) {
  double v[2 * KT];
#pragma unroll
  for (int q = 0; q < 2 * KT; q++) v[q] = 0.0;
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const cplx b = ldx<NT>(y, i);
#pragma unroll
    for (int q = 0; q < KT; q++) {
      const cplx a = ldx<NT>(xs.x[k0 + q], i);
      v[2 * q] = fma(a.x, b.x, v[2 * q]); v[2 * q] = fma(a.y, b.y, v[2 * q]);
      v[2 * q + 1] = fma(a.x, b.y, v[2 * q + 1]); v[2 * q + 1] = fma(-a.y, b.x, v[2 * q + 1]);
    }
  }
  block_reduce_store<2 * KT, false>(v, partials + (long)blockIdx.x * 2 * ktot + 2 * k0);
}

// per-timeslice reductions (reductions/reductions.h:24-87): sum over x and c for each y.
// In the even-odd layout row y is two contiguous runs (one per parity) of hr*nc elements.
// MODE 0: |a|^2 (norm2sq_cv_timeslice, :24-41); 1: conj(a) b, re and im (dot_cv_timeslice, :69-87); 2: Re conj(a) b (redot_cv_timeslice, :47-66)
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_timeslice(const cplx* __restrict__ a, const cplx* __restrict__ b, int hr, int Ly, int nc, double* __restrict__ out) {
  const int y = blockIdx.x;
  const long half_cv = (long)hr * Ly * nc;
  const long run = (long)hr * nc;
  double v[2] = {0.0, 0.0};
  for (int p = 0; p < 2; p++) {
    const cplx* ap = a + p * half_cv + (long)y * run;
    const cplx* bp = MODE ? b + p * half_cv + (long)y * run : nullptr;
    for (long i = threadIdx.x; i < run; i += BLOCK) {
      const cplx u = ap[i];
      if (MODE) {
        const cplx w = bp[i];
        v[0] = fma(u.x, w.x, v[0]); v[0] = fma(u.y, w.y, v[0]);
        if (MODE == 1) { v[1] = fma(u.x, w.y, v[1]); v[1] = fma(-u.y, w.x, v[1]); }
      } else { v[0] = fma(u.x, u.x, v[0]); v[0] = fma(u.y, u.y, v[0]); }
    }
  }
  double part[2];
  __shared__ double res[2];
  block_reduce_store<2, false>(v, res);
  __syncthreads();
  part[0] = res[0]; part[1] = res[1];
  if (threadIdx.x == 0) {
    if (MODE == 1) { out[2 * y] = part[0]; out[2 * y + 1] = part[1]; }
    else out[y] = part[0];
  }
}

// Per-device workspace for partials and the default result slot.
struct RedWorkspace {
  double* partials = nullptr;   // RED_BLOCKS * 2 * RED_MAXK doubles
  double* result = nullptr;     // 2 * RED_MAXK doubles
  double* pinned = nullptr;     // host-pinned, device-visible: the final kernel writes host results straight here
  int device = -1;
};
static thread_local RedWorkspace g_ws;

static int get_ws(RedWorkspace** out) {
  int dev = 0;
  QMG_HIP_CHECK(hipGetDevice(&dev));
  if (g_ws.device != dev) {
    // (leaks the previous device's few hundred KB if a thread hops devices; one rank = one device here)
    QMG_HIP_CHECK(hipMalloc((void**)&g_ws.partials, sizeof(double) * RED_BLOCKS * 2 * RED_MAXK));
    QMG_HIP_CHECK(hipMalloc((void**)&g_ws.result, sizeof(double) * 2 * RED_MAXK));
    QMG_HIP_CHECK(hipHostMalloc((void**)&g_ws.pinned, sizeof(double) * 2 * RED_MAXK, hipHostMallocDefault));
    g_ws.device = dev;
  }
  *out = &g_ws;
  return QMG_SUCCESS;
}

void release_blas_workspace() {   // qmg_shutdown (qmg_runtime.hip)
  if (g_ws.partials) hipFree(g_ws.partials);
  if (g_ws.result) hipFree(g_ws.result);
  if (g_ws.pinned) hipHostFree(g_ws.pinned);
  g_ws = RedWorkspace();
}

static unsigned red_grid(long n) {
  long b = (n + BLOCK - 1) / BLOCK;
  if (b > RED_BLOCKS) b = RED_BLOCKS;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// Host results: when the caller wants the value on the host only, the final kernel has written it into the
// pinned, device-visible buffer; one stream synchronise makes it readable -- no D2H copy on the critical path
// of a Krylov iteration.
static int finish(double* result_dev, int width, double* out_dev, double* out_host, hipStream_t st) {
  if (out_dev && out_dev != result_dev)
    QMG_HIP_CHECK(hipMemcpyAsync(out_dev, result_dev, sizeof(double) * width, hipMemcpyDeviceToDevice, st));
  if (out_host) {
    if (result_dev == g_ws.pinned) {
      QMG_HIP_CHECK(hipStreamSynchronize(st));
      memcpy(out_host, g_ws.pinned, sizeof(double) * width);
    } else {
      QMG_HIP_CHECK(hipMemcpyAsync(out_host, result_dev, sizeof(double) * width, hipMemcpyDeviceToHost, st));
      QMG_HIP_CHECK(hipStreamSynchronize(st));
    }
  }
  return QMG_SUCCESS;
}

template <int OP>
static int reduce2(const void* x, const void* y, size_t n, int width, double* out_dev, double* out_host, void* stream) {
  if (!x || (!out_dev && !out_host)) return QMG_ERR_INVALID;
  if ((OP == RED_DOT || OP == RED_DIFFNORM2) && !y) return QMG_ERR_INVALID;
  RedWorkspace* ws;
  int rc = get_ws(&ws);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  const unsigned g = red_grid((long)n);
  if (blas_nt(n)) k_reduce<OP, true><<<g, BLOCK, 0, st>>>((const cplx*)x, (const cplx*)y, (long)n, ws->partials);
  else k_reduce<OP, false><<<g, BLOCK, 0, st>>>((const cplx*)x, (const cplx*)y, (long)n, ws->partials);
  QMG_LAUNCH_CHECK();
  const bool dist = dist_reductions_on();   // slabs of one lattice: sum (max) over the ranks before the value leaves HBM
  double* res = out_dev ? out_dev : (dist ? ws->result : ws->pinned);
  if (OP == RED_NORMINF) k_reduce_final<true, true><<<width, BLOCK, 0, st>>>(ws->partials, (int)g, 2, width, res);
  else k_reduce_final<false, false><<<width, BLOCK, 0, st>>>(ws->partials, (int)g, 2, width, res);
  QMG_LAUNCH_CHECK();
  if (dist) { rc = dist_allreduce(res, width, OP == RED_NORMINF, st); if (rc) return rc; }
  return finish(res, width, nullptr, out_host, st);
}

template <int OP>
static int blas_launch(void* z, const void* x, const void* y, cplx a, cplx b, size_t n, void* stream) {
  if (n == 0) return QMG_SUCCESS;
  if (blas_nt(n)) k_blas<OP, true><<<grid_1d(n), BLOCK, 0, as_stream(stream)>>>((cplx*)z, (const cplx*)x, (const cplx*)y, a, b, (long)n);
  else k_blas<OP, false><<<grid_1d(n), BLOCK, 0, as_stream(stream)>>>((cplx*)z, (const cplx*)x, (const cplx*)y, a, b, (long)n);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

}  // namespace qmg

using namespace qmg;

extern "C" {

int qmg_zero_vector(void* x, size_t n, void* s) {
  if (!x && n) return QMG_ERR_INVALID;
  return blas_launch<OP_ZERO>(x, nullptr, nullptr, make_double2(0, 0), make_double2(0, 0), n, s);
}
int qmg_copy_vector(void* dst, const void* src, size_t n, void* s) {
  if ((!dst || !src) && n) return QMG_ERR_INVALID;
  return blas_launch<OP_COPY>(dst, src, nullptr, make_double2(0, 0), make_double2(0, 0), n, s);
}
int qmg_cax(double ar, double ai, void* x, size_t n, void* s) {
  if (!x && n) return QMG_ERR_INVALID;
  return blas_launch<OP_CAX>(x, nullptr, nullptr, make_double2(ar, ai), make_double2(0, 0), n, s);
}
int qmg_caxy(double ar, double ai, const void* x, void* y, size_t n, void* s) {
  if ((!x || !y) && n) return QMG_ERR_INVALID;
  return blas_launch<OP_CAXY>(y, x, nullptr, make_double2(ar, ai), make_double2(0, 0), n, s);
}
int qmg_caxpy(double ar, double ai, const void* x, void* y, size_t n, void* s) {
  if ((!x || !y) && n) return QMG_ERR_INVALID;
  return blas_launch<OP_CAXPY>(y, x, nullptr, make_double2(ar, ai), make_double2(0, 0), n, s);
}
int qmg_cxpy(const void* x, void* y, size_t n, void* s) {
  if ((!x || !y) && n) return QMG_ERR_INVALID;
  return blas_launch<OP_CXPY>(y, x, nullptr, make_double2(0, 0), make_double2(0, 0), n, s);
}
int qmg_cxpay(const void* x, double ar, double ai, void* y, size_t n, void* s) {
  if ((!x || !y) && n) return QMG_ERR_INVALID;
  return blas_launch<OP_CXPAY>(y, x, nullptr, make_double2(ar, ai), make_double2(0, 0), n, s);
}
int qmg_caxpby(double ar, double ai, const void* x, double br, double bi, void* y, size_t n, void* s) {
  if ((!x || !y) && n) return QMG_ERR_INVALID;
  return blas_launch<OP_CAXPBY>(y, x, nullptr, make_double2(ar, ai), make_double2(br, bi), n, s);
}
int qmg_cxpyz(const void* x, const void* y, void* z, size_t n, void* s) {
  if ((!x || !y || !z) && n) return QMG_ERR_INVALID;
  return blas_launch<OP_CXPYZ>(z, x, y, make_double2(0, 0), make_double2(0, 0), n, s);
}
int qmg_caxpbyz(double ar, double ai, const void* x, double br, double bi, const void* y, void* z, size_t n, void* s) {
  if ((!x || !y || !z) && n) return QMG_ERR_INVALID;
  return blas_launch<OP_CAXPBYZ>(z, x, y, make_double2(ar, ai), make_double2(br, bi), n, s);
}

int qmg_multi_caxpy(const double* coeffs, const void* const* xs, int k, void* y, size_t n, void* s) {
  if (k < 0 || (k > 0 && (!coeffs || !xs)) || (!y && n)) return QMG_ERR_INVALID;
  if (n == 0) return QMG_SUCCESS;
  int done = 0;
  while (done < k) {
    const int kk = (k - done > MAXPY_K) ? MAXPY_K : k - done;
    MultiAxpy m;
    for (int j = 0; j < kk; j++) {
      if (!xs[done + j]) return QMG_ERR_INVALID;
      m.x[j] = (const cplx*)xs[done + j];
      m.a[j] = make_double2(coeffs[2 * (done + j)], coeffs[2 * (done + j) + 1]);
    }
    if (blas_nt(n)) k_multi_caxpy<true><<<grid_1d(n), BLOCK, 0, as_stream(s)>>>((cplx*)y, m, kk, (long)n);
    else k_multi_caxpy<false><<<grid_1d(n), BLOCK, 0, as_stream(s)>>>((cplx*)y, m, kk, (long)n);
    QMG_LAUNCH_CHECK();
    done += kk;
  }
  return QMG_SUCCESS;
}

int qmg_caxy_pattern(const double* scale, const int* shuffle, int nc, const void* x, void* y, size_t nsite, void* s) {
  if (!scale || !shuffle || nc < 1 || nc > 64 || !x || !y || x == y) return QMG_ERR_INVALID;
  Pattern pat;
  for (int c = 0; c < nc; c++) {
    if (shuffle[c] < 0 || shuffle[c] >= nc) return QMG_ERR_INVALID;
    pat.scale[c] = scale[c];
    pat.shuffle[c] = shuffle[c];
  }
  if (nsite == 0) return QMG_SUCCESS;
  k_pattern<<<grid_1d(nsite * nc), BLOCK, 0, as_stream(s)>>>((cplx*)y, (const cplx*)x, (long)nsite, nc, pat);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

int qmg_gaussian(void* x, size_t n, unsigned long long seed, void* s) {
  if (!x && n) return QMG_ERR_INVALID;
  if (n == 0) return QMG_SUCCESS;
  k_gaussian<<<grid_1d(n), BLOCK, 0, as_stream(s)>>>((cplx*)x, (long)n, seed);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

int qmg_gaussian_slab(void* x, int Lx, int Ly_global, int y0, int Ly_local, int nc, unsigned long long seed, void* s) {
  if (!x || !valid_lattice(Lx, Ly_global) || !valid_lattice(Lx, Ly_local) || nc < 1 || y0 < 0 || y0 + Ly_local > Ly_global) return QMG_ERR_INVALID;
  k_gaussian_slab<<<grid_1d((size_t)Lx * Ly_local * nc), BLOCK, 0, as_stream(s)>>>((cplx*)x, Lx / 2, Ly_global, y0, Ly_local, nc, seed);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

int qmg_gaussian_wall_source(void* cv, int Lx, int Ly, int nc, int timeslice, int color, unsigned long long seed, double deviation, double mean, void* s) {
  if (!cv || !valid_lattice(Lx, Ly) || nc < 1) return QMG_ERR_INVALID;
  if (timeslice < 0 || timeslice >= Ly || color < 0 || color >= nc) return QMG_ERR_INVALID;   // the facade prints the reference's messages (:94-107)
  k_wall_source<<<grid_1d((size_t)Lx * Ly * nc), BLOCK, 0, as_stream(s)>>>((cplx*)cv, Lx / 2, Ly, nc, timeslice, color, seed, deviation, mean);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

int qmg_norm2sq(const void* x, size_t n, double* od, double* oh, void* s) { return reduce2<RED_NORM2>(x, nullptr, n, 1, od, oh, s); }
int qmg_dot(const void* x, const void* y, size_t n, double* od, double* oh, void* s) { return reduce2<RED_DOT>(x, y, n, 2, od, oh, s); }
int qmg_diffnorm2sq(const void* x, const void* y, size_t n, double* od, double* oh, void* s) { return reduce2<RED_DIFFNORM2>(x, y, n, 1, od, oh, s); }
int qmg_norminf(const void* x, size_t n, double* od, double* oh, void* s) { return reduce2<RED_NORMINF>(x, nullptr, n, 1, od, oh, s); }

int qmg_multidot(const void* const* xs, int k, const void* y, size_t n, double* out_dev, double* out_host, void* stream) {
  if (!xs || !y || k < 1 || k > RED_MAXK || (!out_dev && !out_host)) return QMG_ERR_INVALID;
  RedWorkspace* ws;
  int rc = get_ws(&ws);
  if (rc) return rc;
  MultiPtrs mp;
  for (int i = 0; i < k; i++) { if (!xs[i]) return QMG_ERR_INVALID; mp.x[i] = (const cplx*)xs[i]; }
  hipStream_t st = as_stream(stream);
  const unsigned g = red_grid((long)n);
  int k0 = 0;
  while (k0 < k) {   // chunks of 8 / 4 / 2 / 1 vectors: y is re-read once per chunk (each dot is its own accumulation: the chunking does not change a digit)
    const int rem = k - k0;
    const bool nt = blas_nt(n);
#define QMG_MDOT(KT) { if (nt) k_multidot<KT, true><<<g, BLOCK, 0, st>>>(mp, k0, (const cplx*)y, (long)n, ws->partials, k); \
                       else k_multidot<KT, false><<<g, BLOCK, 0, st>>>(mp, k0, (const cplx*)y, (long)n, ws->partials, k); k0 += KT; }
    if (rem >= 8) QMG_MDOT(8) else if (rem >= 4) QMG_MDOT(4) else if (rem >= 2) QMG_MDOT(2) else QMG_MDOT(1)
#undef QMG_MDOT
    QMG_LAUNCH_CHECK();
  }
  const bool dist = dist_reductions_on();
  double* res = out_dev ? out_dev : (dist ? ws->result : ws->pinned);
  k_reduce_final<false, false><<<2 * k, BLOCK, 0, st>>>(ws->partials, (int)g, 2 * k, 2 * k, res);
  QMG_LAUNCH_CHECK();
  if (dist) { rc = dist_allreduce(res, 2 * k, false, st); if (rc) return rc; }
  return finish(res, 2 * k, nullptr, out_host, st);
}

int qmg_norm2sq_cv_timeslice(const void* cv, int Lx, int Ly, int nc, double* out_dev, double* out_host, void* stream) {
  if (!cv || !valid_lattice(Lx, Ly) || nc < 1 || (!out_dev && !out_host)) return QMG_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  double* res = out_dev;
  double* tmp = nullptr;
  if (!res) { QMG_HIP_CHECK(hipMalloc((void**)&tmp, sizeof(double) * Ly)); res = tmp; }
  k_timeslice<0><<<Ly, BLOCK, 0, st>>>((const cplx*)cv, nullptr, Lx / 2, Ly, nc, res);
  QMG_LAUNCH_CHECK();
  int rc = finish(res, Ly, nullptr, out_host, st);
  if (tmp) { hipStreamSynchronize(st); hipFree(tmp); }
  return rc;
}

int qmg_dot_cv_timeslice(const void* a, const void* b, int Lx, int Ly, int nc, double* out_dev, double* out_host, void* stream) {
  if (!a || !b || !valid_lattice(Lx, Ly) || nc < 1 || (!out_dev && !out_host)) return QMG_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  double* res = out_dev;
  double* tmp = nullptr;
  if (!res) { QMG_HIP_CHECK(hipMalloc((void**)&tmp, sizeof(double) * 2 * Ly)); res = tmp; }
  k_timeslice<1><<<Ly, BLOCK, 0, st>>>((const cplx*)a, (const cplx*)b, Lx / 2, Ly, nc, res);
  QMG_LAUNCH_CHECK();
  int rc = finish(res, 2 * Ly, nullptr, out_host, st);
  if (tmp) { hipStreamSynchronize(st); hipFree(tmp); }
  return rc;
}

// redot_cv_timeslice (reductions.h:47-66): sum[y] = Re sum_{x,c} conj(a) b -- Ly doubles
int qmg_redot_cv_timeslice(const void* a, const void* b, int Lx, int Ly, int nc, double* out_dev, double* out_host, void* stream) {
  if (!a || !b || !valid_lattice(Lx, Ly) || nc < 1 || (!out_dev && !out_host)) return QMG_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  double* res = out_dev;
  double* tmp = nullptr;
  if (!res) { QMG_HIP_CHECK(hipMalloc((void**)&tmp, sizeof(double) * Ly)); res = tmp; }
  k_timeslice<2><<<Ly, BLOCK, 0, st>>>((const cplx*)a, (const cplx*)b, Lx / 2, Ly, nc, res);
  QMG_LAUNCH_CHECK();
  int rc = finish(res, Ly, nullptr, out_host, st);
  if (tmp) { hipStreamSynchronize(st); hipFree(tmp); }
  return rc;
}

}  // extern "C"
