// qmg_fill.hip -- cshift, operator construction from U(1) links, and stencil-variant builds.
// Setup-time kernels: correctness and coalesced access first; none of these is on the
// per-iteration path.
#include <string.h>

#include "qmg_common.h"

namespace qmg {

// Neighbour of the parity-p site (y, j) in direction `cdir` (QMG_CSHIFT_FROM_*), as a site index
// in the OPPOSITE half.  The rule is read off cshift_2d.h:60-119,149-210.
__device__ __forceinline__ long neighbour_site(int cdir, int p, int y, int j, int hr, int Ly, long half_vol) {
  const int s = (y + p) & 1;
  int yn = y, jn = j;
  if (cdir == QMG_CSHIFT_FROM_XP1) { jn = j + s; if (jn == hr) jn = 0; }
  else if (cdir == QMG_CSHIFT_FROM_XM1) { jn = j + s - 1; if (jn < 0) jn = hr - 1; }
  else if (cdir == QMG_CSHIFT_FROM_YP1) { yn = (y + 1 == Ly) ? 0 : y + 1; }
  else if (cdir == QMG_CSHIFT_FROM_YM1) { yn = (y == 0) ? Ly - 1 : y - 1; }
  return (long)(1 - p) * half_vol + (long)yn * hr + jn;
}

// lhs(site, k) = rhs(neighbour(site), k) for the output parities selected (par_first/par_count).
__global__ __launch_bounds__(BLOCK) void k_cshift(cplx* __restrict__ lhs, const cplx* __restrict__ rhs, int cdir,
                                                  int dof, int hr, int Ly, int par_first, int par_count) {
  const long half_vol = (long)hr * Ly;
  const long row_elems = (long)hr * dof;
  const int nrows = Ly * par_count;
  for (int row = blockIdx.y; row < nrows; row += gridDim.y) {
    const int p = par_first + row / Ly;
    const int y = row % Ly;
    for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < row_elems; t += (long)gridDim.x * BLOCK) {
      const int j = (int)(t / dof);
      const int k = (int)(t - (long)j * dof);
      const long nb = neighbour_site(cdir, p, y, j, hr, Ly, half_vol);
      const long out = (long)p * half_vol + (long)y * hr + j;
      lhs[out * dof + k] = rhs[nb * dof + k];
    }
  }
}

// ---- Wilson2D::update_links (wilson.h:153-209) ----
__global__ __launch_bounds__(BLOCK) void k_wilson_fill(cplx* __restrict__ clover, cplx* __restrict__ hop,
                                                       const cplx* __restrict__ g, int hr, int Ly, double w) {
  const long half_vol = (long)hr * Ly, vol = 2 * half_vol, cm = vol * 4;
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < vol; i += (long)gridDim.x * BLOCK) {
    const int p = (int)(i / half_vol);
    const long wi = i - (long)p * half_vol;
    const int y = (int)(wi / hr), j = (int)(wi - (long)y * hr);
    const cplx ux = g[i], uy = g[vol + i];
    const cplx uxb = cconj(g[neighbour_site(QMG_CSHIFT_FROM_XM1, p, y, j, hr, Ly, half_vol)]);
    const cplx uyb = cconj(g[vol + neighbour_site(QMG_CSHIFT_FROM_YM1, p, y, j, hr, Ly, half_vol)]);
    const double hw = -0.5 * w;
    cplx* c = clover + 4 * i;
    c[0] = cmake(2.0 * w, 0.0); c[1] = cmake(0.0, 0.0); c[2] = cmake(0.0, 0.0); c[3] = cmake(2.0 * w, 0.0);
    cplx* h = hop + 4 * i;
    // +x : 1/2 [[-w,1],[1,-w]] Ux
    h[0] = cmake(hw * ux.x, hw * ux.y); h[1] = cmake(0.5 * ux.x, 0.5 * ux.y);
    h[2] = h[1];                        h[3] = h[0];
    // +y : 1/2 [[-w,-i],[i,-w]] Uy   (-i u/2 = (u.y/2, -u.x/2);  i u/2 = (-u.y/2, u.x/2))
    h = hop + cm + 4 * i;
    h[0] = cmake(hw * uy.x, hw * uy.y); h[1] = cmake(0.5 * uy.y, -0.5 * uy.x);
    h[2] = cmake(-0.5 * uy.y, 0.5 * uy.x); h[3] = h[0];
    // -x : 1/2 [[-w,-1],[-1,-w]] conj Ux(x - x^)
    h = hop + 2 * cm + 4 * i;
    h[0] = cmake(hw * uxb.x, hw * uxb.y); h[1] = cmake(-0.5 * uxb.x, -0.5 * uxb.y);
    h[2] = h[1];                          h[3] = h[0];
    // -y : 1/2 [[-w,i],[-i,-w]] conj Uy(x - y^)
    h = hop + 3 * cm + 4 * i;
    h[0] = cmake(hw * uyb.x, hw * uyb.y); h[1] = cmake(-0.5 * uyb.y, 0.5 * uyb.x);
    h[2] = cmake(0.5 * uyb.y, -0.5 * uyb.x); h[3] = h[0];
  }
}

// The same for one y-slab (rows y0 .. y0 + Ly_loc - 1 of the global lattice, y0 even): the links are read from the GLOBAL
// gauge field (32 B/site, every rank holds it), so the -y hop of the slab's first row gets the neighbour slab's link.
__global__ __launch_bounds__(BLOCK) void k_wilson_fill_slab(cplx* __restrict__ clover, cplx* __restrict__ hop, const cplx* __restrict__ g,
                                                            int hr, int Ly_g, int y0, int Ly_l, double w) {
  const long hv_l = (long)hr * Ly_l, vol_l = 2 * hv_l, cm = vol_l * 4;
  const long hv_g = (long)hr * Ly_g, vol_g = 2 * hv_g;
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < vol_l; i += (long)gridDim.x * BLOCK) {
    const int p = (int)(i / hv_l);
    const long wi = i - (long)p * hv_l;
    const int yl = (int)(wi / hr), j = (int)(wi - (long)yl * hr);
    const int y = y0 + yl;
    const long gi = (long)p * hv_g + (long)y * hr + j;
    const cplx ux = g[gi], uy = g[vol_g + gi];
    const cplx uxb = cconj(g[neighbour_site(QMG_CSHIFT_FROM_XM1, p, y, j, hr, Ly_g, hv_g)]);
    const cplx uyb = cconj(g[vol_g + neighbour_site(QMG_CSHIFT_FROM_YM1, p, y, j, hr, Ly_g, hv_g)]);
    const double hw = -0.5 * w;
    cplx* c = clover + 4 * i;
    c[0] = cmake(2.0 * w, 0.0); c[1] = cmake(0.0, 0.0); c[2] = cmake(0.0, 0.0); c[3] = cmake(2.0 * w, 0.0);
    cplx* h = hop + 4 * i;
    h[0] = cmake(hw * ux.x, hw * ux.y); h[1] = cmake(0.5 * ux.x, 0.5 * ux.y);
    h[2] = h[1];                        h[3] = h[0];
    h = hop + cm + 4 * i;
    h[0] = cmake(hw * uy.x, hw * uy.y); h[1] = cmake(0.5 * uy.y, -0.5 * uy.x);
    h[2] = cmake(-0.5 * uy.y, 0.5 * uy.x); h[3] = h[0];
    h = hop + 2 * cm + 4 * i;
    h[0] = cmake(hw * uxb.x, hw * uxb.y); h[1] = cmake(-0.5 * uxb.x, -0.5 * uxb.y);
    h[2] = h[1];                          h[3] = h[0];
    h = hop + 3 * cm + 4 * i;
    h[0] = cmake(hw * uyb.x, hw * uyb.y); h[1] = cmake(-0.5 * uyb.y, 0.5 * uyb.x);
    h[2] = cmake(0.5 * uyb.y, -0.5 * uyb.x); h[3] = h[0];
  }
}

// ---- Staggered2D (staggered.h:50-72; eta_y = 1 - 2 (x % 2), :253-259) and GaugedLaplace2D (gaugedlaplace.h:45-68) ----
// mode 0: staggered (hopping only) ; mode 1: gauged Laplace (clover = 4, hopping = -U)
// (y-slab: rows y0 .. y0 + Ly_l - 1 of a lattice of Ly_g rows, the links read from the GLOBAL field as in k_wilson_fill_slab; the whole
//  lattice is y0 = 0, Ly_l = Ly_g)
__global__ __launch_bounds__(BLOCK) void k_nc1_fill(cplx* __restrict__ clover, cplx* __restrict__ hop,
                                                    const cplx* __restrict__ g, int hr, int Ly_g, int y0, int Ly_l, int mode) {
  const long hv_l = (long)hr * Ly_l, vol = 2 * hv_l;
  const long hv_g = (long)hr * Ly_g, vol_g = 2 * hv_g;
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < vol; i += (long)gridDim.x * BLOCK) {
    const int p = (int)(i / hv_l);
    const long wi = i - (long)p * hv_l;
    const int yl = (int)(wi / hr), j = (int)(wi - (long)yl * hr);
    const int y = y0 + yl;
    const long gi = (long)p * hv_g + (long)y * hr + j;
    const cplx ux = g[gi], uy = g[vol_g + gi];
    const cplx uxb = cconj(g[neighbour_site(QMG_CSHIFT_FROM_XM1, p, y, j, hr, Ly_g, hv_g)]);
    const cplx uyb = cconj(g[vol_g + neighbour_site(QMG_CSHIFT_FROM_YM1, p, y, j, hr, Ly_g, hv_g)]);
    if (mode == 0) {
      const double eta = ((y + p) & 1) ? -1.0 : 1.0;   // x = 2j + s, s = (y+p)&1
      hop[i] = cmake(-0.5 * ux.x, -0.5 * ux.y);
      hop[vol + i] = cmake(-0.5 * eta * uy.x, -0.5 * eta * uy.y);
      hop[2 * vol + i] = cmake(0.5 * uxb.x, 0.5 * uxb.y);
      hop[3 * vol + i] = cmake(0.5 * eta * uyb.x, 0.5 * eta * uyb.y);
    } else {
      clover[i] = cmake(4.0, 0.0);
      hop[i] = cmake(-ux.x, -ux.y);
      hop[vol + i] = cmake(-uy.x, -uy.y);
      hop[2 * vol + i] = cmake(-uxb.x, -uxb.y);
      hop[3 * vol + i] = cmake(-uyb.x, -uyb.y);
    }
  }
}

// ---- batched conjugate transpose (cMATcopy_conjtrans_square), optionally gathered from a neighbour ----
// out[site][r][c] = conj(in[src(site)][c][r]);  cdir == 0: src = site.
// (y-slab: the source matrices of rows -1 / Ly come from `halo` = the neighbouring rank's boundary row of the SAME field,
//  [parity][hr][nc^2], filled by qmg_halo_exchange of that field with nc^2 components per site)
__global__ __launch_bounds__(BLOCK) void k_conjtrans(cplx* __restrict__ out, const cplx* __restrict__ in, long nsite,
                                                     int nc, int cdir, int hr, int Ly, const cplx* __restrict__ halo = nullptr) {
  const long nc2 = (long)nc * nc;
  const long total = nsite * nc2;
  const long half_vol = (long)hr * Ly;
  for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (long)gridDim.x * BLOCK) {
    const long site = t / nc2;
    const int e = (int)(t - site * nc2);
    const int r = e / nc, c = e - r * nc;
    long src = site;
    const cplx* from = in;
    if (cdir) {
      const int p = (int)(site / half_vol);
      const long wi = site - (long)p * half_vol;
      const int y = (int)(wi / hr), j = (int)(wi - (long)y * hr);
      src = neighbour_site(cdir, p, y, j, hr, Ly, half_vol);
      if (halo && ((cdir == QMG_CSHIFT_FROM_YP1 && y + 1 == Ly) || (cdir == QMG_CSHIFT_FROM_YM1 && y == 0))) {
        from = halo;
        src = (long)(1 - p) * hr + j;               // the opposite-parity site of the neighbouring rank's boundary row
      }
    }
    out[t] = cconj(from[src * nc2 + (long)c * nc + r]);
  }
}

// ---- right block Jacobi (stencil_2d.h:1452-1601) ----
// One block per site: Gauss-Jordan with partial pivoting on [C | 1] held in LDS.  The reference
// inverts by QR (:1536-1537); any backward-stable inverse agrees to rounding.
__global__ __launch_bounds__(BLOCK) void k_clover_inverse(cplx* __restrict__ cinv, const cplx* __restrict__ clover,
                                                          long vol, long half_vol, int nc, double sr, double si,
                                                          double er, double ei, double dr, double di) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  cplx* m = reinterpret_cast<cplx*>(smem_raw);   // nc x 2nc
  __shared__ int piv_row;
  __shared__ cplx piv_inv;
  const int W = 2 * nc;
  const long nc2 = (long)nc * nc;
  for (long site = blockIdx.x; site < vol; site += gridDim.x) {
    const double sg = (site >= half_vol) ? -1.0 : 1.0;
    for (int t = threadIdx.x; t < nc * W; t += BLOCK) {
      const int r = t / W, c = t - r * W;
      cplx v = cmake(0.0, 0.0);
      if (c < nc) {
        if (clover) v = clover[site * nc2 + (long)r * nc + c];
        if (r == c) {
          const double dg = (nc % 2 == 0) ? (((long)r * (nc + 1) < nc2 / 2) ? 1.0 : -1.0) : 0.0;
          v.x += sr + sg * er + dg * dr;
          v.y += si + sg * ei + dg * di;
        }
      } else if (c - nc == r) v = cmake(1.0, 0.0);
      m[t] = v;
    }
    __syncthreads();
    for (int k = 0; k < nc; k++) {
      if (threadIdx.x == 0) {
        int best = k;
        double bv = m[k * W + k].x * m[k * W + k].x + m[k * W + k].y * m[k * W + k].y;
        for (int r = k + 1; r < nc; r++) {
          const cplx z = m[r * W + k];
          const double a = z.x * z.x + z.y * z.y;
          if (a > bv) { bv = a; best = r; }
        }
        piv_row = best;
        const cplx z = m[best * W + k];
        piv_inv = cmake(z.x / bv, -z.y / bv);
      }
      __syncthreads();
      const int pr = piv_row;
      if (pr != k)
        for (int c = threadIdx.x; c < W; c += BLOCK) { cplx t0 = m[k * W + c]; m[k * W + c] = m[pr * W + c]; m[pr * W + c] = t0; }
      __syncthreads();
      for (int c = threadIdx.x; c < W; c += BLOCK) m[k * W + c] = cmul(m[k * W + c], piv_inv);
      __syncthreads();
      // eliminate column k from every other row; column k itself last (it holds the factors)
      for (int t = threadIdx.x; t < nc * W; t += BLOCK) {
        const int r = t / W, c = t - r * W;
        if (r == k || c == k) continue;
        const cplx f = m[r * W + k];
        const cplx pk = m[k * W + c];
        cplx v = m[t];
        v.x -= f.x * pk.x - f.y * pk.y;
        v.y -= f.x * pk.y + f.y * pk.x;
        m[t] = v;
      }
      __syncthreads();
      for (int r = threadIdx.x; r < nc; r += BLOCK)
        if (r != k) m[r * W + k] = cmake(0.0, 0.0);
      __syncthreads();
    }
    for (int t = threadIdx.x; t < nc * nc; t += BLOCK) {
      const int r = t / nc, c = t - r * nc;
      cinv[site * nc2 + t] = m[r * W + nc + c];
    }
    __syncthreads();
  }
}

// rb_hopping[dir](x) = hopping[dir](x) . cinv(x + dir)   (:1556-1581)
// (y-slab: cinv of rows -1 / Ly from ci_lo / ci_hi, [parity][hr][nc^2], filled by qmg_halo_exchange of cinv as an nc^2-component field)
__global__ __launch_bounds__(BLOCK) void k_rb_hopping(cplx* __restrict__ rb, const cplx* __restrict__ hop,
                                                      const cplx* __restrict__ cinv, int nc, int hr, int Ly,
                                                      const cplx* __restrict__ ci_lo, const cplx* __restrict__ ci_hi) {
  const long half_vol = (long)hr * Ly, vol = 2 * half_vol;
  const long nc2 = (long)nc * nc, cm = vol * nc2;
  const long total = 4 * cm;
  for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (long)gridDim.x * BLOCK) {
    const int dir = (int)(t / cm);
    const long u = t - (long)dir * cm;
    const long site = u / nc2;
    const int e = (int)(u - site * nc2);
    const int r = e / nc, c = e - r * nc;
    const int p = (int)(site / half_vol);
    const long wi = site - (long)p * half_vol;
    const int y = (int)(wi / hr), j = (int)(wi - (long)y * hr);
    const int cdir = (dir == 0) ? QMG_CSHIFT_FROM_XP1 : (dir == 1) ? QMG_CSHIFT_FROM_YP1 : (dir == 2) ? QMG_CSHIFT_FROM_XM1 : QMG_CSHIFT_FROM_YM1;
    const long nb = neighbour_site(cdir, p, y, j, hr, Ly, half_vol);
    const cplx* hrow = hop + (long)dir * cm + site * nc2 + (long)r * nc;
    const cplx* ci = cinv + nb * nc2 + c;
    if (dir == 1 && ci_hi && y + 1 == Ly) ci = ci_hi + ((long)(1 - p) * hr + j) * nc2 + c;   // the +y neighbour lives on the next rank
    if (dir == 3 && ci_lo && y == 0) ci = ci_lo + ((long)(1 - p) * hr + j) * nc2 + c;
    cplx acc = cmake(0.0, 0.0);
    for (int k = 0; k < nc; k++) cmac(acc, hrow[k], ci[(long)k * nc]);
    rb[t] = acc;
  }
}

__global__ __launch_bounds__(BLOCK) void k_identity_cm(cplx* __restrict__ out, long nsite, int nc) {
  const long nc2 = (long)nc * nc, total = nsite * nc2;
  for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < total; t += (long)gridDim.x * BLOCK) {
    const int e = (int)(t % nc2);
    out[t] = cmake((e / nc == e % nc) ? 1.0 : 0.0, 0.0);
  }
}

// complex<double> -> complex<float> (round to nearest): the opt-in fp32 matrix storage of qmg_stencil_apply_mat32
__global__ __launch_bounds__(BLOCK) void k_c64_to_c32(float2* __restrict__ dst, const cplx* __restrict__ src, long n) {
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    const cplx v = src[i];
    dst[i] = make_float2((float)v.x, (float)v.y);
  }
}

// element-wise copy between storage precisions (round to nearest / widen)
template <typename TD, typename TS>
__global__ __launch_bounds__(BLOCK) void k_convert(void* __restrict__ dst, const void* __restrict__ src, long n) {
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) stc<TD>(dst, i, ldc<TS>(src, i));
}

}  // namespace qmg

using namespace qmg;

extern "C" {

int qmg_convert(void* dst, int dst_dtype, const void* src, int src_dtype, size_t n, void* stream) {
  if (!valid_dtype(dst_dtype) || !valid_dtype(src_dtype) || ((!dst || !src) && n)) return QMG_ERR_INVALID;
  if (n == 0) return QMG_SUCCESS;
  hipStream_t st = as_stream(stream);
  const unsigned g = grid_1d(n);
  if (dst_dtype == QMG_C64 && src_dtype == QMG_C64) k_convert<double, double><<<g, BLOCK, 0, st>>>(dst, src, (long)n);
  else if (dst_dtype == QMG_C32 && src_dtype == QMG_C64) k_convert<float, double><<<g, BLOCK, 0, st>>>(dst, src, (long)n);
  else if (dst_dtype == QMG_C64 && src_dtype == QMG_C32) k_convert<double, float><<<g, BLOCK, 0, st>>>(dst, src, (long)n);
  else k_convert<float, float><<<g, BLOCK, 0, st>>>(dst, src, (long)n);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

int qmg_cshift(void* lhs, const void* rhs, int cdir, int eo, int dof, int Lx, int Ly, void* stream) {
  if (!lhs || !rhs || dof < 1 || !valid_lattice(Lx, Ly)) return QMG_ERR_INVALID;
  if (cdir < QMG_CSHIFT_FROM_0 || cdir > QMG_CSHIFT_FROM_YM1) return QMG_ERR_UNSUPPORTED;   // distance-2 (:120-129)
  if (!(eo & QMG_EO_FROM_EVENODD)) return QMG_ERR_INVALID;
  const int hr = Lx / 2;
  const long half_vol = (long)hr * Ly;
  hipStream_t st = as_stream(stream);
  if (cdir == QMG_CSHIFT_FROM_0) {
    // Reference behaviour kept (cshift_2d.h:58,147): half_size ELEMENTS, same half, dof ignored.
    if (eo & QMG_EO_FROM_EVEN) QMG_HIP_CHECK(hipMemcpyAsync(lhs, rhs, sizeof(cplx) * half_vol, hipMemcpyDeviceToDevice, st));
    if (eo & QMG_EO_FROM_ODD)
      QMG_HIP_CHECK(hipMemcpyAsync((cplx*)lhs + half_vol, (const cplx*)rhs + half_vol, sizeof(cplx) * half_vol, hipMemcpyDeviceToDevice, st));
    return QMG_SUCCESS;
  }
  // FROM_EVEN writes the odd half, FROM_ODD writes the even half.
  const bool from_even = eo & QMG_EO_FROM_EVEN, from_odd = eo & QMG_EO_FROM_ODD;
  const int par_first = from_odd ? 0 : 1;
  const int par_count = (from_even && from_odd) ? 2 : 1;
  const long row_elems = (long)hr * dof;
  unsigned gx = (unsigned)((row_elems + BLOCK - 1) / BLOCK);
  if (gx > 1024) gx = 1024;
  const int nrows = Ly * par_count;
  dim3 grid(gx, nrows > 65535 ? 65535 : nrows);
  k_cshift<<<grid, BLOCK, 0, st>>>((cplx*)lhs, (const cplx*)rhs, cdir, dof, hr, Ly, par_first, par_count);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

int qmg_wilson_fill(void* clover, void* hopping, const void* gauge, int Lx, int Ly, double w, void* stream) {
  if (!clover || !hopping || !gauge || !valid_lattice(Lx, Ly)) return QMG_ERR_INVALID;
  k_wilson_fill<<<grid_1d((size_t)Lx * Ly), BLOCK, 0, as_stream(stream)>>>((cplx*)clover, (cplx*)hopping, (const cplx*)gauge, Lx / 2, Ly, w);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

int qmg_wilson_fill_slab(void* clover, void* hopping, const void* gauge_global, int Lx, int Ly_global, int y0, int Ly_local, double w, void* stream) {
  if (!clover || !hopping || !gauge_global || !valid_lattice(Lx, Ly_global) || !valid_lattice(Lx, Ly_local)) return QMG_ERR_INVALID;
  if (y0 < 0 || (y0 & 1) || y0 + Ly_local > Ly_global) return QMG_ERR_INVALID;
  k_wilson_fill_slab<<<grid_1d((size_t)Lx * Ly_local), BLOCK, 0, as_stream(stream)>>>((cplx*)clover, (cplx*)hopping, (const cplx*)gauge_global,
                                                                                    Lx / 2, Ly_global, y0, Ly_local, w);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

int qmg_staggered_fill(void* hopping, const void* gauge, int Lx, int Ly, void* stream) {
  if (!hopping || !gauge || !valid_lattice(Lx, Ly)) return QMG_ERR_INVALID;
  k_nc1_fill<<<grid_1d((size_t)Lx * Ly), BLOCK, 0, as_stream(stream)>>>(nullptr, (cplx*)hopping, (const cplx*)gauge, Lx / 2, Ly, 0, Ly, 0);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

// One y-slab of the staggered / gauged Laplace stencil (rows y0 .. y0 + Ly_local - 1, y0 even) from the GLOBAL gauge field, as qmg_wilson_fill_slab.
int qmg_staggered_fill_slab(void* hopping, const void* gauge_global, int Lx, int Ly_global, int y0, int Ly_local, void* stream) {
  if (!hopping || !gauge_global || !valid_lattice(Lx, Ly_global) || !valid_lattice(Lx, Ly_local)) return QMG_ERR_INVALID;
  if (y0 < 0 || (y0 & 1) || y0 + Ly_local > Ly_global) return QMG_ERR_INVALID;
  k_nc1_fill<<<grid_1d((size_t)Lx * Ly_local), BLOCK, 0, as_stream(stream)>>>(nullptr, (cplx*)hopping, (const cplx*)gauge_global, Lx / 2, Ly_global, y0, Ly_local, 0);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}
int qmg_laplace_fill_slab(void* clover, void* hopping, const void* gauge_global, int Lx, int Ly_global, int y0, int Ly_local, void* stream) {
  if (!clover || !hopping || !gauge_global || !valid_lattice(Lx, Ly_global) || !valid_lattice(Lx, Ly_local)) return QMG_ERR_INVALID;
  if (y0 < 0 || (y0 & 1) || y0 + Ly_local > Ly_global) return QMG_ERR_INVALID;
  k_nc1_fill<<<grid_1d((size_t)Lx * Ly_local), BLOCK, 0, as_stream(stream)>>>((cplx*)clover, (cplx*)hopping, (const cplx*)gauge_global, Lx / 2, Ly_global, y0, Ly_local, 1);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

int qmg_laplace_fill(void* clover, void* hopping, const void* gauge, int Lx, int Ly, void* stream) {
  if (!clover || !hopping || !gauge || !valid_lattice(Lx, Ly)) return QMG_ERR_INVALID;
  k_nc1_fill<<<grid_1d((size_t)Lx * Ly), BLOCK, 0, as_stream(stream)>>>((cplx*)clover, (cplx*)hopping, (const cplx*)gauge, Lx / 2, Ly, 0, Ly, 1);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

int qmg_cmat_conjtrans(void* out, const void* in, size_t nsite, int nc, void* stream) {
  if (!out || !in || nc < 1 || out == in) return QMG_ERR_INVALID;
  if (nsite == 0) return QMG_SUCCESS;
  k_conjtrans<<<grid_1d(nsite * nc * nc), BLOCK, 0, as_stream(stream)>>>((cplx*)out, (const cplx*)in, (long)nsite, nc, 0, 1, 1);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

// build_dagger_stencil (stencil_2d.h:1080-1139):
//   dagger[+x](x) = [hopping[-x](x + x^)]^dag  (:1106-1107), and cyclically for +y, -x, -y.
int qmg_build_dagger(void* dclover, void* dhopping, const void* clover, const void* hopping,
                     int Lx, int Ly, int nc, void* stream) {
  if (!valid_lattice(Lx, Ly) || nc < 1) return QMG_ERR_INVALID;
  const long vol = (long)Lx * Ly, cm = vol * nc * nc;
  hipStream_t st = as_stream(stream);
  if (clover && dclover) {
    k_conjtrans<<<grid_1d((size_t)cm), BLOCK, 0, st>>>((cplx*)dclover, (const cplx*)clover, vol, nc, 0, Lx / 2, Ly);
    QMG_LAUNCH_CHECK();
  }
  if (hopping && dhopping) {
    const int src_dir[4] = {2, 3, 0, 1};   // -x, -y, +x, +y
    const int cdir[4] = {QMG_CSHIFT_FROM_XP1, QMG_CSHIFT_FROM_YP1, QMG_CSHIFT_FROM_XM1, QMG_CSHIFT_FROM_YM1};
    for (int dir = 0; dir < 4; dir++) {
      k_conjtrans<<<grid_1d((size_t)cm), BLOCK, 0, st>>>((cplx*)dhopping + dir * cm, (const cplx*)hopping + src_dir[dir] * cm,
                                                         vol, nc, cdir[dir], Lx / 2, Ly);
      QMG_LAUNCH_CHECK();
    }
  }
  return QMG_SUCCESS;
}

// build_dagger_stencil on a y-slab: dagger[+y](x) at the slab's last row needs hopping[-y] of the next rank's first row, dagger[-y](x)
// at its first row hopping[+y] of the previous rank's last row.  ym_halo_hi = the `hi` buffer of qmg_halo_exchange applied to the -y
// field (hopping + 3 size_cm) with nc^2 components per site, yp_halo_lo = the `lo` buffer of the exchange of the +y field
// (hopping + size_cm).  Everything else is site-local or inside a row.
int qmg_build_dagger_slab(void* dclover, void* dhopping, const void* clover, const void* hopping, int Lx, int Ly, int nc,
                          const void* ym_halo_hi, const void* yp_halo_lo, void* stream) {
  if (!valid_lattice(Lx, Ly) || nc < 1 || (hopping && dhopping && (!ym_halo_hi || !yp_halo_lo))) return QMG_ERR_INVALID;
  const long vol = (long)Lx * Ly, cm = vol * nc * nc;
  hipStream_t st = as_stream(stream);
  if (clover && dclover) {
    k_conjtrans<<<grid_1d((size_t)cm), BLOCK, 0, st>>>((cplx*)dclover, (const cplx*)clover, vol, nc, 0, Lx / 2, Ly);
    QMG_LAUNCH_CHECK();
  }
  if (hopping && dhopping) {
    const int src_dir[4] = {2, 3, 0, 1};
    const int cdir[4] = {QMG_CSHIFT_FROM_XP1, QMG_CSHIFT_FROM_YP1, QMG_CSHIFT_FROM_XM1, QMG_CSHIFT_FROM_YM1};
    const void* halo[4] = {nullptr, ym_halo_hi, nullptr, yp_halo_lo};
    for (int dir = 0; dir < 4; dir++) {
      k_conjtrans<<<grid_1d((size_t)cm), BLOCK, 0, st>>>((cplx*)dhopping + dir * cm, (const cplx*)hopping + src_dir[dir] * cm, vol, nc, cdir[dir], Lx / 2, Ly,
                                                         (const cplx*)halo[dir]);
      QMG_LAUNCH_CHECK();
    }
  }
  return QMG_SUCCESS;
}

int qmg_build_rbjacobi(void* cinv, void* rb_clover, void* rb_hopping, const qmg_stencil_desc* d, void* stream) {
  if (!d || !cinv || !valid_lattice(d->Lx, d->Ly) || d->nc < 1) return QMG_ERR_INVALID;
  if (d->nc > 32) return QMG_ERR_UNSUPPORTED;
  const bool no_shift = d->shift[0] == 0 && d->shift[1] == 0 && d->eo_shift[0] == 0 && d->eo_shift[1] == 0 &&
                        d->dof_shift[0] == 0 && d->dof_shift[1] == 0;
  if (!d->clover && no_shift) return QMG_ERR_INVALID;   // stencil_2d.h:1471-1475
  const int nc = d->nc;
  const long vol = (long)d->Lx * d->Ly;
  hipStream_t st = as_stream(stream);
  const size_t smem = sizeof(cplx) * 2 * nc * nc;
  unsigned gb = vol > 256 * 16 ? 256 * 16 : (unsigned)vol;
  k_clover_inverse<<<gb, BLOCK, smem, st>>>((cplx*)cinv, (const cplx*)d->clover, vol, vol / 2, nc, d->shift[0], d->shift[1],
                                            d->eo_shift[0], d->eo_shift[1], d->dof_shift[0], d->dof_shift[1]);
  QMG_LAUNCH_CHECK();
  if (rb_clover) {
    k_identity_cm<<<grid_1d((size_t)vol * nc * nc), BLOCK, 0, st>>>((cplx*)rb_clover, vol, nc);
    QMG_LAUNCH_CHECK();
  }
  if (d->hopping && rb_hopping) {
    k_rb_hopping<<<grid_1d((size_t)4 * vol * nc * nc), BLOCK, 0, st>>>((cplx*)rb_hopping, (const cplx*)d->hopping, (const cplx*)cinv,
                                                                      nc, d->Lx / 2, d->Ly, nullptr, nullptr);
    QMG_LAUNCH_CHECK();
  }
  return QMG_SUCCESS;
}

// The hopping part of the right-block-Jacobi stencil on a y-slab: rb_hopping[dir](x) = hopping[dir](x) . cinv(x + dir) with cinv of the
// rows -1 / Ly from the halo buffers.  Sequence on a slab: qmg_build_rbjacobi(cinv, rb_clover, NULL, d) -- cinv and the identity clover
// are site-local --, qmg_halo_exchange of cinv as a field with nc^2 components per site, then this.
int qmg_rb_hopping_slab(void* rb_hopping, const qmg_stencil_desc* d, const void* cinv, const void* cinv_halo_lo, const void* cinv_halo_hi, void* stream) {
  if (!rb_hopping || !d || !d->hopping || !cinv || !cinv_halo_lo || !cinv_halo_hi || !valid_lattice(d->Lx, d->Ly) || d->nc < 1) return QMG_ERR_INVALID;
  const long vol = (long)d->Lx * d->Ly;
  k_rb_hopping<<<grid_1d((size_t)4 * vol * d->nc * d->nc), BLOCK, 0, as_stream(stream)>>>((cplx*)rb_hopping, (const cplx*)d->hopping, (const cplx*)cinv, d->nc,
                                                                                            d->Lx / 2, d->Ly, (const cplx*)cinv_halo_lo, (const cplx*)cinv_halo_hi);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

}  // extern "C"

extern "C" int qmg_c64_to_c32(void* dst_f32, const void* src_f64, size_t n, void* stream) {
  if ((!dst_f32 || !src_f64) && n) return QMG_ERR_INVALID;
  if (n == 0) return QMG_SUCCESS;
  qmg::k_c64_to_c32<<<qmg::grid_1d(n), qmg::BLOCK, 0, qmg::as_stream(stream)>>>((float2*)dst_f32, (const qmg::cplx*)src_f64, (long)n);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}
