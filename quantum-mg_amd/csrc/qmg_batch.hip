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

#include "qmg_common.h"

namespace qmg {

constexpr int BATCH_MAX = 16;
constexpr int BRED_BLOCKS = 1024;    // partials per reduction per system (= RED_BLOCKS of qmg_blas.hip)
constexpr int BDOT_MAX = 32;         // vectors per batched multidot / multi_caxpy call

struct BatchIdx { int n; unsigned char id[BATCH_MAX]; };

static BatchIdx expand_mask(unsigned mask, int nrhs) {
  BatchIdx b;
  b.n = 0;
  for (int k = 0; k < nrhs && k < BATCH_MAX; k++)
    if ((mask >> k) & 1u) b.id[b.n++] = (unsigned char)k;
  for (int k = b.n; k < BATCH_MAX; k++) b.id[k] = 0;
  return b;
}

// ---------------- element-wise ----------------
struct BatchCoef { cplx a[BATCH_MAX], b[BATCH_MAX]; };   // indexed by system id

template <int OP>
__global__ __launch_bounds__(BLOCK) void k_bblas(cplx* __restrict__ z, const cplx* __restrict__ x, const cplx* __restrict__ y, const BatchCoef c,
                                                 const BatchIdx bi, long n, long stride) {
  const int k = bi.id[blockIdx.y];
  const long off = (long)k * stride;
  const cplx a = c.a[k], b = c.b[k];
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    cplx r;
    if (OP == QMG_BOP_ZERO) r = cmake(0.0, 0.0);
    else if (OP == QMG_BOP_COPY) r = x[off + i];
    else if (OP == QMG_BOP_CAX) r = cmul(a, z[off + i]);
    else if (OP == QMG_BOP_CAXPY) { r = z[off + i]; cmac(r, a, x[off + i]); }
    else if (OP == QMG_BOP_CXPY) r = cadd(z[off + i], x[off + i]);
    else { r = cmul(a, x[off + i]); cmac(r, b, y[off + i]); }   // CAXPBYZ
    z[off + i] = r;
  }
}

// y_k += sum_j a[j][k] x_j,k for up to 8 vector sets per launch (coefficients travel as kernel arguments)
constexpr int BMAXPY_J = 8;
struct BatchMultiAxpy { const cplx* x[BMAXPY_J]; cplx a[BMAXPY_J][BATCH_MAX]; };
__global__ __launch_bounds__(BLOCK) void k_bmulti_caxpy(cplx* __restrict__ y, const BatchMultiAxpy m, int nj, const BatchIdx bi, long n, long stride) {
  const int k = bi.id[blockIdx.y];
  const long off = (long)k * stride;
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    cplx acc = y[off + i];
    for (int j = 0; j < nj; j++) {
      const cplx c = m.a[j][k];
      // a zero coefficient means "this system does not use vector set j" (systems of a batch own different numbers of
      // directions): the slot may hold stale pool memory, which must not be read (0 * inf = nan)
      if (c.x != 0.0 || c.y != 0.0) cmac(acc, c, m.x[j][off + i]);
    }
    y[off + i] = acc;
  }
}

// ---------------- reductions (two-stage, deterministic; partition identical to qmg_blas.hip) ----------------
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
template <int OP>
__global__ __launch_bounds__(BLOCK) void k_breduce(const cplx* __restrict__ x, const cplx* __restrict__ y, long n, long stride, const BatchIdx bi,
                                                   double* __restrict__ partials) {
  const long off = (long)bi.id[blockIdx.y] * stride;
  double v[2] = {0.0, 0.0};
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const cplx a = x[off + i];
    if (OP == QMG_BRED_NORM2) { v[0] = fma(a.x, a.x, v[0]); v[0] = fma(a.y, a.y, v[0]); }
    else if (OP == QMG_BRED_DOT) {
      const cplx b = y[off + i];
      v[0] = fma(a.x, b.x, v[0]); v[0] = fma(a.y, b.y, v[0]);
      v[1] = fma(a.x, b.y, v[1]); v[1] = fma(-a.y, b.x, v[1]);
    } else {
      const cplx b = y[off + i];
      const double dx = a.x - b.x, dy = a.y - b.y;
      v[0] = fma(dx, dx, v[0]); v[0] = fma(dy, dy, v[0]);
    }
  }
  bblock_reduce_store<2>(v, partials + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 2);
}

struct BatchPtrs { const cplx* x[BDOT_MAX]; };
// KT dots <x_j,k , y_k> per system in one pass over y_k
template <int KT>
__global__ __launch_bounds__(BLOCK) void k_bmultidot(const BatchPtrs xs, int j0, const cplx* __restrict__ y, long n, long stride, const BatchIdx bi,
                                                     double* __restrict__ partials, int jtot) {
  const long off = (long)bi.id[blockIdx.y] * stride;
  double v[2 * KT];
#pragma unroll
  for (int q = 0; q < 2 * KT; q++) v[q] = 0.0;
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const cplx b = y[off + i];
#pragma unroll
    for (int q = 0; q < KT; q++) {
      const cplx a = xs.x[j0 + q][off + i];
      v[2 * q] = fma(a.x, b.x, v[2 * q]); v[2 * q] = fma(a.y, b.y, v[2 * q]);
      v[2 * q + 1] = fma(a.x, b.y, v[2 * q + 1]); v[2 * q + 1] = fma(-a.y, b.x, v[2 * q + 1]);
    }
  }
  bblock_reduce_store<2 * KT>(v, partials + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 2 * jtot + 2 * j0);
}

// stage 2: block (q, s) sums the nparts partials of output q of system slot s in a fixed order
__global__ __launch_bounds__(BLOCK) void k_breduce_final(const double* __restrict__ partials, int nparts, int width, const BatchIdx bi, int out_width,
                                                         double* __restrict__ out) {
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
  }
}

struct BatchWorkspace {
  double* partials = nullptr;   // BATCH_MAX * BRED_BLOCKS * 2 * BDOT_MAX doubles (8 MiB)
  double* pinned = nullptr;     // host-pinned, device-visible results: BATCH_MAX * 2 * BDOT_MAX doubles
  int device = -1;
};
static thread_local BatchWorkspace g_bws;

static int get_bws(BatchWorkspace** out) {
  int dev = 0;
  QMG_HIP_CHECK(hipGetDevice(&dev));
  if (g_bws.device != dev) {
    QMG_HIP_CHECK(hipMalloc((void**)&g_bws.partials, sizeof(double) * BATCH_MAX * BRED_BLOCKS * 2 * BDOT_MAX));
    QMG_HIP_CHECK(hipHostMalloc((void**)&g_bws.pinned, sizeof(double) * BATCH_MAX * 2 * BDOT_MAX, hipHostMallocDefault));
    g_bws.device = dev;
  }
  *out = &g_bws;
  return QMG_SUCCESS;
}

static unsigned bred_grid(long n) {
  long b = (n + BLOCK - 1) / BLOCK;
  if (b > BRED_BLOCKS) b = BRED_BLOCKS;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ---------------- transfer: the null vectors are read once per pass of up to KB systems ----------------
struct BXferGeom {
  int fhr, fLy, fnc;
  int chr, cLy, cnc;
  int bx, by;
  long fhalf_vol, chalf_vol;
  long fsize;
};
__device__ __forceinline__ long bcoarse_site_index(const BXferGeom& g, int cx, int cy) {
  const int p = (cx + cy) & 1;
  return (long)(cy + p * g.cLy) * g.chr + (cx >> 1);
}

// fine_k[e] += sum_d null[d][e] * coarse_k[ci(e)*cnc + d]      (transfer.h:455-480 for KB systems at once)
template <int KB>
__global__ __launch_bounds__(BLOCK) void k_bprolong(const cplx* __restrict__ nullv, int nvec, const cplx* __restrict__ coarse, cplx* __restrict__ fine,
                                                    const BXferGeom g, const BatchIdx bi, int s0, long cstride, long fstride) {
  const long row_elems = (long)g.fhr * g.fnc;
  const int nrows = 2 * g.fLy;
  const int ns = (bi.n - s0 < KB) ? bi.n - s0 : KB;
  for (int row = blockIdx.y; row < nrows; row += gridDim.y) {
    const int p = row / g.fLy, y = row - p * g.fLy;
    const int s = (y + p) & 1;
    const int cy = y / g.by;
    for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < row_elems; t += (long)gridDim.x * BLOCK) {
      const int j = (int)(t / g.fnc);
      const int cx = (2 * j + s) / g.bx;
      const long ci = bcoarse_site_index(g, cx, cy);
      const long e = ((long)p * g.fhalf_vol + (long)y * g.fhr) * g.fnc + t;
      // per-slot base pointers outside the d loop; unused slots alias slot 0 (computed, discarded): no branches inside
      cplx acc[KB];
      const cplx* cq[KB];
#pragma unroll
      for (int q = 0; q < KB; q++) {
        const long id = bi.id[s0 + ((q < ns) ? q : 0)];
        cq[q] = coarse + id * cstride + ci * g.cnc;
        acc[q] = fine[id * fstride + e];
      }
      int d = 0;
      for (; d + 4 <= nvec; d += 4) {   // four null-vector loads in flight
        cplx nv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const cplx* src = nullv + (long)(d + u) * g.fsize + e;
          nv[u].x = __builtin_nontemporal_load(&src->x);
          nv[u].y = __builtin_nontemporal_load(&src->y);
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int q = 0; q < KB; q++) cmac(acc[q], nv[u], cq[q][d + u]);
      }
      for (; d < nvec; d++) {
        const cplx nv = nullv[(long)d * g.fsize + e];
#pragma unroll
        for (int q = 0; q < KB; q++) cmac(acc[q], nv, cq[q][d]);
      }
#pragma unroll
      for (int q = 0; q < KB; q++)
        if (q < ns) fine[(long)bi.id[s0 + q] * fstride + e] = acc[q];
    }
  }
}

// coarse_k[ci*cnc + d] += sum_{e in block ci} conj(null[d][e]) fine_k[e]     (transfer.h:487-511)
// A group of TG threads (a power of two, 2..256, about an eighth of the block's element count) owns one coarse site;
// a workgroup carries 256/TG consecutive coarse sites of a coarse row, so neighbouring groups read neighbouring fine
// runs.  Thread l of a group walks elements l, l+TG, ...; DC null vectors x KB systems of partial sums per thread; the
// group sum is a fixed-order shuffle butterfly (inside a wavefront) plus an LDS pass (across the wavefronts of a wide
// group).  One writer per (site, d, system): no atomics, deterministic.
constexpr int BX_DC = 4;
template <int KB>
__global__ __launch_bounds__(BLOCK) void k_brestrict(const cplx* __restrict__ nullv, int nvec, const cplx* __restrict__ fine, cplx* __restrict__ coarse,
                                                     const BXferGeom g, const BatchIdx bi, int s0, long cstride, long fstride, int TG) {
  __shared__ double red[BLOCK / WAVE][BX_DC * KB * 2];
  const int ns = (bi.n - s0 < KB) ? bi.n - s0 : KB;
  const int cLx = 2 * g.chr;
  const long ncs = (long)cLx * g.cLy;
  const int G = (g.bx / 2) * g.fnc;              // contiguous elements the block owns on each fine half-row
  const int nel = 2 * g.by * G;                  // elements in the block
  const int NS = BLOCK / TG;                     // coarse sites per workgroup
  const int grp = threadIdx.x / TG, l = threadIdx.x - grp * TG;
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const int wpg = (TG > WAVE) ? TG / WAVE : 1;   // wavefronts per group
  const long ngroups = (ncs + NS - 1) / NS;
  const cplx* fq[KB];
#pragma unroll
  for (int k = 0; k < KB; k++) fq[k] = fine + (long)bi.id[s0 + ((k < ns) ? k : 0)] * fstride;
  for (long wg = blockIdx.x; wg < ngroups; wg += gridDim.x) {
    const long cs = wg * NS + grp;
    const bool live = cs < ncs;
    const int cy = live ? (int)(cs / cLx) : 0, cx = live ? (int)(cs - (long)cy * cLx) : 0;
    const long ci = bcoarse_site_index(g, cx, cy);
    for (int d0 = 0; d0 < nvec; d0 += BX_DC) {
      const int dn = (nvec - d0 < BX_DC) ? nvec - d0 : BX_DC;
      cplx acc[BX_DC][KB];
#pragma unroll
      for (int q = 0; q < BX_DC; q++)
#pragma unroll
        for (int k = 0; k < KB; k++) acc[q][k] = cmake(0.0, 0.0);
      if (live) {
        for (int t = l; t < nel; t += TG) {
          const int rr = t / G, el = t - rr * G;
          const int p = rr / g.by;
          const int y = cy * g.by + (rr - p * g.by);
          const long e = ((long)p * g.fhalf_vol + (long)y * g.fhr + (long)cx * (g.bx / 2)) * g.fnc + el;
          cplx f[KB];
#pragma unroll
          for (int k = 0; k < KB; k++) f[k] = fq[k][e];   // (unused slots alias slot 0: computed, discarded)
#pragma unroll
          for (int q = 0; q < BX_DC; q++)
            if (q < dn) {
              const cplx* src = nullv + (long)(d0 + q) * g.fsize + e;
              cplx nv;
              nv.x = __builtin_nontemporal_load(&src->x);
              nv.y = __builtin_nontemporal_load(&src->y);
#pragma unroll
              for (int k = 0; k < KB; k++) cmac_conj(acc[q][k], nv, f[k]);
            }
        }
      }
      // butterfly over the group's lanes inside the wavefront
      const int span = (TG < WAVE) ? TG : WAVE;
#pragma unroll
      for (int q = 0; q < BX_DC; q++)
#pragma unroll
        for (int k = 0; k < KB; k++) {
          double sx = acc[q][k].x, sy = acc[q][k].y;
          for (int o = 1; o < span; o <<= 1) { sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); }
          acc[q][k].x = sx; acc[q][k].y = sy;
        }
      if (wpg == 1) {
        if (live && l == 0) {
#pragma unroll
          for (int q = 0; q < BX_DC; q++)
#pragma unroll
            for (int k = 0; k < KB; k++)
              if (q < dn && k < ns) {
                const long o = (long)bi.id[s0 + k] * cstride + ci * g.cnc + d0 + q;
                coarse[o] = cadd(coarse[o], acc[q][k]);
              }
        }
      } else {
        __syncthreads();   // red[] free again
        if (lane == 0) {
#pragma unroll
          for (int q = 0; q < BX_DC; q++)
#pragma unroll
            for (int k = 0; k < KB; k++) { red[wv][(q * KB + k) * 2] = acc[q][k].x; red[wv][(q * KB + k) * 2 + 1] = acc[q][k].y; }
        }
        __syncthreads();
        if (live && l < dn * ns) {
          const int q = l / ns, k = l - q * ns;
          double tx = 0.0, ty = 0.0;
          for (int w = 0; w < wpg; w++) { tx += red[grp * wpg + w][(q * KB + k) * 2]; ty += red[grp * wpg + w][(q * KB + k) * 2 + 1]; }
          const long o = (long)bi.id[s0 + k] * cstride + ci * g.cnc + d0 + q;
          coarse[o] = cadd(coarse[o], cmake(tx, ty));
        }
      }
    }
  }
}

static int make_bgeom(BXferGeom* g, int fLx, int fLy, int fnc, int cLx, int cLy, int cnc) {
  if (!valid_lattice(fLx, fLy) || !valid_lattice(cLx, cLy) || fnc < 1 || cnc < 1) return QMG_ERR_INVALID;
  if (fLx % cLx || fLy % cLy) return QMG_ERR_INVALID;
  g->fhr = fLx / 2; g->fLy = fLy; g->fnc = fnc;
  g->chr = cLx / 2; g->cLy = cLy; g->cnc = cnc;
  g->bx = fLx / cLx; g->by = fLy / cLy;
  g->fhalf_vol = (long)g->fhr * fLy; g->chalf_vol = (long)g->chr * cLy;
  g->fsize = 2 * g->fhalf_vol * fnc;
  if (g->bx % 2) return QMG_ERR_UNSUPPORTED;   // odd block widths: use the single-vector entry points
  return QMG_SUCCESS;
}

}  // namespace qmg

using namespace qmg;

extern "C" {

int qmg_batch_blas(int op, const double* a, const double* b, const void* x, const void* y, void* z, size_t n, int nrhs, size_t stride,
                   unsigned mask, void* stream) {
  if (nrhs < 1 || nrhs > BATCH_MAX || (!z && n)) return QMG_ERR_INVALID;
  if (op < QMG_BOP_ZERO || op > QMG_BOP_CAXPBYZ) return QMG_ERR_INVALID;
  if ((op == QMG_BOP_COPY || op == QMG_BOP_CAXPY || op == QMG_BOP_CXPY || op == QMG_BOP_CAXPBYZ) && !x && n) return QMG_ERR_INVALID;
  if (op == QMG_BOP_CAXPBYZ && ((!y && n) || !b)) return QMG_ERR_INVALID;
  if ((op == QMG_BOP_CAX || op == QMG_BOP_CAXPY || op == QMG_BOP_CAXPBYZ) && !a) return QMG_ERR_INVALID;
  const BatchIdx bi = expand_mask(mask, nrhs);
  if (bi.n == 0 || n == 0) return QMG_SUCCESS;
  BatchCoef c;
  for (int k = 0; k < BATCH_MAX; k++) {
    c.a[k] = (a && k < nrhs) ? make_double2(a[2 * k], a[2 * k + 1]) : make_double2(0.0, 0.0);
    c.b[k] = (b && k < nrhs) ? make_double2(b[2 * k], b[2 * k + 1]) : make_double2(0.0, 0.0);
  }
  dim3 grid(grid_1d(n), (unsigned)bi.n);
  hipStream_t st = as_stream(stream);
  switch (op) {
    case QMG_BOP_ZERO: k_bblas<QMG_BOP_ZERO><<<grid, BLOCK, 0, st>>>((cplx*)z, nullptr, nullptr, c, bi, (long)n, (long)stride); break;
    case QMG_BOP_COPY: k_bblas<QMG_BOP_COPY><<<grid, BLOCK, 0, st>>>((cplx*)z, (const cplx*)x, nullptr, c, bi, (long)n, (long)stride); break;
    case QMG_BOP_CAX: k_bblas<QMG_BOP_CAX><<<grid, BLOCK, 0, st>>>((cplx*)z, nullptr, nullptr, c, bi, (long)n, (long)stride); break;
    case QMG_BOP_CAXPY: k_bblas<QMG_BOP_CAXPY><<<grid, BLOCK, 0, st>>>((cplx*)z, (const cplx*)x, nullptr, c, bi, (long)n, (long)stride); break;
    case QMG_BOP_CXPY: k_bblas<QMG_BOP_CXPY><<<grid, BLOCK, 0, st>>>((cplx*)z, (const cplx*)x, nullptr, c, bi, (long)n, (long)stride); break;
    default: k_bblas<QMG_BOP_CAXPBYZ><<<grid, BLOCK, 0, st>>>((cplx*)z, (const cplx*)x, (const cplx*)y, c, bi, (long)n, (long)stride); break;
  }
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

// coeffs[(j*nrhs + k)*2 + {0,1}]: coefficient of vector set j for system k
int qmg_batch_multi_caxpy(const double* coeffs, const void* const* xs, int nj, void* y, size_t n, int nrhs, size_t stride, unsigned mask,
                          void* stream) {
  if (nrhs < 1 || nrhs > BATCH_MAX || nj < 0 || (nj > 0 && (!coeffs || !xs)) || (!y && n)) return QMG_ERR_INVALID;
  const BatchIdx bi = expand_mask(mask, nrhs);
  if (bi.n == 0 || n == 0 || nj == 0) return QMG_SUCCESS;
  unsigned gx = grid_1d(n);
  dim3 grid(gx, (unsigned)bi.n);
  for (int j0 = 0; j0 < nj; j0 += BMAXPY_J) {
    const int jj = (nj - j0 < BMAXPY_J) ? nj - j0 : BMAXPY_J;
    BatchMultiAxpy m;
    for (int j = 0; j < BMAXPY_J; j++) {
      m.x[j] = (j < jj) ? (const cplx*)xs[j0 + j] : nullptr;
      if (j < jj && !xs[j0 + j]) return QMG_ERR_INVALID;
      for (int k = 0; k < BATCH_MAX; k++)
        m.a[j][k] = (j < jj && k < nrhs) ? make_double2(coeffs[((size_t)(j0 + j) * nrhs + k) * 2], coeffs[((size_t)(j0 + j) * nrhs + k) * 2 + 1])
                                         : make_double2(0.0, 0.0);
    }
    k_bmulti_caxpy<<<grid, BLOCK, 0, as_stream(stream)>>>((cplx*)y, m, jj, bi, (long)n, (long)stride);
    QMG_LAUNCH_CHECK();
  }
  return QMG_SUCCESS;
}

// out_host[2*k + {0,1}] for every ACTIVE system k (inactive entries are left untouched); synchronises the stream
int qmg_batch_reduce(int op, const void* x, const void* y, size_t n, int nrhs, size_t stride, unsigned mask, double* out_host, void* stream) {
  if (nrhs < 1 || nrhs > BATCH_MAX || !x || !out_host) return QMG_ERR_INVALID;
  if (op < QMG_BRED_NORM2 || op > QMG_BRED_DIFFNORM2) return QMG_ERR_INVALID;
  if (op != QMG_BRED_NORM2 && !y) return QMG_ERR_INVALID;
  const BatchIdx bi = expand_mask(mask, nrhs);
  if (bi.n == 0) return QMG_SUCCESS;
  BatchWorkspace* ws;
  int rc = get_bws(&ws);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  const unsigned g = bred_grid((long)n);
  dim3 grid(g, (unsigned)bi.n);
  if (op == QMG_BRED_NORM2) k_breduce<QMG_BRED_NORM2><<<grid, BLOCK, 0, st>>>((const cplx*)x, nullptr, (long)n, (long)stride, bi, ws->partials);
  else if (op == QMG_BRED_DOT) k_breduce<QMG_BRED_DOT><<<grid, BLOCK, 0, st>>>((const cplx*)x, (const cplx*)y, (long)n, (long)stride, bi, ws->partials);
  else k_breduce<QMG_BRED_DIFFNORM2><<<grid, BLOCK, 0, st>>>((const cplx*)x, (const cplx*)y, (long)n, (long)stride, bi, ws->partials);
  QMG_LAUNCH_CHECK();
  k_breduce_final<<<dim3(2, (unsigned)bi.n), BLOCK, 0, st>>>(ws->partials, (int)g, 2, bi, 2, ws->pinned);
  QMG_LAUNCH_CHECK();
  QMG_HIP_CHECK(hipStreamSynchronize(st));
  for (int s = 0; s < bi.n; s++) { out_host[2 * bi.id[s]] = ws->pinned[2 * bi.id[s]]; out_host[2 * bi.id[s] + 1] = ws->pinned[2 * bi.id[s] + 1]; }
  return QMG_SUCCESS;
}

// out_host[(k*nj + j)*2 + {0,1}] = <xs[j]_k , y_k> for every active system k; synchronises the stream
int qmg_batch_multidot(const void* const* xs, int nj, const void* y, size_t n, int nrhs, size_t stride, unsigned mask, double* out_host, void* stream) {
  if (nrhs < 1 || nrhs > BATCH_MAX || nj < 1 || nj > BDOT_MAX || !xs || !y || !out_host) return QMG_ERR_INVALID;
  const BatchIdx bi = expand_mask(mask, nrhs);
  if (bi.n == 0) return QMG_SUCCESS;
  BatchWorkspace* ws;
  int rc = get_bws(&ws);
  if (rc) return rc;
  hipStream_t st = as_stream(stream);
  BatchPtrs p;
  for (int j = 0; j < BDOT_MAX; j++) {
    p.x[j] = (j < nj) ? (const cplx*)xs[j] : nullptr;
    if (j < nj && !xs[j]) return QMG_ERR_INVALID;
  }
  const unsigned g = bred_grid((long)n);
  dim3 grid(g, (unsigned)bi.n);
  int j0 = 0;
  while (j0 < nj) {   // same 4/2/1 chunking as qmg_multidot
    const int left = nj - j0;
    if (left >= 4) { k_bmultidot<4><<<grid, BLOCK, 0, st>>>(p, j0, (const cplx*)y, (long)n, (long)stride, bi, ws->partials, nj); j0 += 4; }
    else if (left >= 2) { k_bmultidot<2><<<grid, BLOCK, 0, st>>>(p, j0, (const cplx*)y, (long)n, (long)stride, bi, ws->partials, nj); j0 += 2; }
    else { k_bmultidot<1><<<grid, BLOCK, 0, st>>>(p, j0, (const cplx*)y, (long)n, (long)stride, bi, ws->partials, nj); j0 += 1; }
    QMG_LAUNCH_CHECK();
  }
  k_breduce_final<<<dim3(2 * nj, (unsigned)bi.n), BLOCK, 0, st>>>(ws->partials, (int)g, 2 * nj, bi, 2 * nj, ws->pinned);
  QMG_LAUNCH_CHECK();
  QMG_HIP_CHECK(hipStreamSynchronize(st));
  for (int s = 0; s < bi.n; s++)
    memcpy(out_host + (size_t)bi.id[s] * 2 * nj, ws->pinned + (size_t)bi.id[s] * 2 * nj, sizeof(double) * 2 * nj);
  return QMG_SUCCESS;
}

int qmg_prolong_batch(const void* nullvecs, int nvec, const void* coarse, void* fine, int fLx, int fLy, int fnc, int cLx, int cLy, int cnc,
                      int nrhs, size_t cstride, size_t fstride, unsigned mask, void* stream) {
  if (!nullvecs || !coarse || !fine || nvec < 1 || nrhs < 1 || nrhs > BATCH_MAX) return QMG_ERR_INVALID;
  BXferGeom g;
  int rc = make_bgeom(&g, fLx, fLy, fnc, cLx, cLy, cnc);
  if (rc) return rc;
  if (nvec != cnc) return QMG_ERR_INVALID;
  const BatchIdx bi = expand_mask(mask, nrhs);
  if (bi.n == 0) return QMG_SUCCESS;
  const long row_elems = (long)g.fhr * g.fnc;
  unsigned gx = (unsigned)((row_elems + BLOCK - 1) / BLOCK);
  unsigned gy = (unsigned)(2 * g.fLy > 65535 ? 65535 : 2 * g.fLy);
  dim3 grid(gx, gy);
  for (int s0 = 0; s0 < bi.n; s0 += 8) {
    if (bi.n - s0 > 4)
      k_bprolong<8><<<grid, BLOCK, 0, as_stream(stream)>>>((const cplx*)nullvecs, nvec, (const cplx*)coarse, (cplx*)fine, g, bi, s0, (long)cstride, (long)fstride);
    else
      k_bprolong<4><<<grid, BLOCK, 0, as_stream(stream)>>>((const cplx*)nullvecs, nvec, (const cplx*)coarse, (cplx*)fine, g, bi, s0, (long)cstride, (long)fstride);
    QMG_LAUNCH_CHECK();
  }
  return QMG_SUCCESS;
}

int qmg_restrict_batch(const void* nullvecs, int nvec, const void* fine, void* coarse, int fLx, int fLy, int fnc, int cLx, int cLy, int cnc,
                       int nrhs, size_t fstride, size_t cstride, unsigned mask, void* stream) {
  if (!nullvecs || !coarse || !fine || nvec < 1 || nrhs < 1 || nrhs > BATCH_MAX) return QMG_ERR_INVALID;
  BXferGeom g;
  int rc = make_bgeom(&g, fLx, fLy, fnc, cLx, cLy, cnc);
  if (rc) return rc;
  if (nvec != cnc) return QMG_ERR_INVALID;
  const BatchIdx bi = expand_mask(mask, nrhs);
  if (bi.n == 0) return QMG_SUCCESS;
  const long ncs = (long)cLx * cLy;
  const int nel = g.bx * g.by * g.fnc;
  // threads per coarse site: ~8 block elements per thread, so that the cross-lane butterfly (2 DC KB log2(TG) shuffles per
  // pass) stays small next to the 8 DC KB complex MACs a thread does per pass (one element per thread made the L0->L1
  // restriction shuffle-bound: 6.3 ms for 8 systems against 0.57 ms per system in the single-vector kernel)
  int TG = 2;
  while (TG * 2 <= nel / 8 && TG < BLOCK) TG <<= 1;
  const long ngroups = (ncs + BLOCK / TG - 1) / (BLOCK / TG);
  unsigned gx = (unsigned)(ngroups > 262144 ? 262144 : ngroups);
  for (int s0 = 0; s0 < bi.n; s0 += 8) {
    if (bi.n - s0 > 4)
      k_brestrict<8><<<gx, BLOCK, 0, as_stream(stream)>>>((const cplx*)nullvecs, nvec, (const cplx*)fine, (cplx*)coarse, g, bi, s0, (long)cstride, (long)fstride, TG);
    else
      k_brestrict<4><<<gx, BLOCK, 0, as_stream(stream)>>>((const cplx*)nullvecs, nvec, (const cplx*)fine, (cplx*)coarse, g, bi, s0, (long)cstride, (long)fstride, TG);
    QMG_LAUNCH_CHECK();
  }
  return QMG_SUCCESS;
}

}  // extern "C"
