// qmg_oracle.cpp -- CPU ORACLE (test infrastructure; see qmg_oracle.h header comment).
//
// A plain single-threaded restatement of the reference's algorithm for the
// multigrid hot path, keeping the reference's PASS STRUCTURE (one cshift copy
// pass + one batched mat-vec pass per direction per parity, stencil_2d.h:706-802)
// so that it is also a fair "reference-structured" CPU baseline.  Nothing here
// is copied from /root/reference; each function cites the lines it follows.
//
// Parity status: see qmg_oracle.h ("PINNING STATUS").

#include "qmg_oracle.h"

#include <chrono>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef std::complex<double> cplx;

namespace {

inline cplx* C(double* p) { return reinterpret_cast<cplx*>(p); }
inline const cplx* C(const double* p) { return reinterpret_cast<const cplx*>(p); }

// Explicit complex multiply-add, y += a*b, without the libstdc++ NaN-recovery
// branch (__muldc3); four multiplies + four adds, the textbook 8 flop.
inline void cmac(cplx& y, const cplx& a, const cplx& b) {
  double yr = y.real(), yi = y.imag();
  yr += a.real() * b.real();
  yr -= a.imag() * b.imag();
  yi += a.real() * b.imag();
  yi += a.imag() * b.real();
  y = cplx(yr, yi);
}
inline cplx cmul(const cplx& a, const cplx& b) {
  return cplx(a.real() * b.real() - a.imag() * b.imag(), a.real() * b.imag() + a.imag() * b.real());
}

// ---------------- quantum-linalg leaves (semantics inferred from call sites; parity unpinned) -------------
// cMATxpy(M,x,y,n,nr,nc): y[i,r] += sum_c M[i,r,c] x[i,c]   (row-major per site, README.md:8)
void cMATxpy(const cplx* M, const cplx* x, cplx* y, long nsite, int nc) {
  for (long i = 0; i < nsite; i++) {
    const cplx* m = M + i * nc * nc;
    const cplx* xi = x + i * nc;
    cplx* yi = y + i * nc;
    for (int r = 0; r < nc; r++) {
      cplx acc = yi[r];
      for (int c = 0; c < nc; c++) cmac(acc, m[r * nc + c], xi[c]);
      yi[r] = acc;
    }
  }
}
void caxpy(cplx a, const cplx* x, cplx* y, long n) {
  for (long i = 0; i < n; i++) cmac(y[i], a, x[i]);
}
void conj_transpose_sq(const cplx* in, cplx* out, long nsite, int nc) {  // cMATcopy_conjtrans_square
  for (long i = 0; i < nsite; i++)
    for (int r = 0; r < nc; r++)
      for (int c = 0; c < nc; c++) out[(i * nc + r) * nc + c] = std::conj(in[(i * nc + c) * nc + r]);
}
void matmul_sq(const cplx* X, const cplx* Y, cplx* Z, long nsite, int nc) {  // cMATxtMATyMATz_square: Z = X.Y
  for (long i = 0; i < nsite; i++)
    for (int r = 0; r < nc; r++)
      for (int c = 0; c < nc; c++) {
        cplx acc = 0.0;
        for (int k = 0; k < nc; k++) cmac(acc, X[(i * nc + r) * nc + k], Y[(i * nc + k) * nc + c]);
        Z[(i * nc + r) * nc + c] = acc;
      }
}
// Batched inverse. The reference does QR + back-substitution (stencil_2d.h:1536-1537);
// any backward-stable inverse agrees to rounding, here Gauss-Jordan with partial pivoting.
int inverse_sq(const cplx* A, cplx* Ainv, long nsite, int nc) {
  std::vector<cplx> w(2 * (size_t)nc * nc);
  for (long s = 0; s < nsite; s++) {
    const cplx* a = A + s * nc * nc;
    cplx* m = w.data();           // nc x 2nc augmented
    const int W = 2 * nc;
    for (int r = 0; r < nc; r++)
      for (int c = 0; c < nc; c++) {
        m[r * W + c] = a[r * nc + c];
        m[r * W + nc + c] = (r == c) ? 1.0 : 0.0;
      }
    for (int k = 0; k < nc; k++) {
      int piv = k;
      double best = std::abs(m[k * W + k]);
      for (int r = k + 1; r < nc; r++)
        if (std::abs(m[r * W + k]) > best) { best = std::abs(m[r * W + k]); piv = r; }
      if (best == 0.0) return -1;
      if (piv != k)
        for (int c = 0; c < W; c++) std::swap(m[k * W + c], m[piv * W + c]);
      cplx inv = 1.0 / m[k * W + k];
      for (int c = 0; c < W; c++) m[k * W + c] = cmul(m[k * W + c], inv);
      for (int r = 0; r < nc; r++) {
        if (r == k) continue;
        cplx f = m[r * W + k];
        if (f == 0.0) continue;
        for (int c = 0; c < W; c++) m[r * W + c] -= cmul(f, m[k * W + c]);
      }
    }
    for (int r = 0; r < nc; r++)
      for (int c = 0; c < nc; c++) Ainv[s * nc * nc + r * nc + c] = m[r * W + nc + c];
  }
  return 0;
}

// ---------------- cshift, one source parity (cshift_2d.h:45-222) ----------------
// Output parity p_out = 1 - p_in.  For an output site on row y, s = (y + p_out) & 1 and x = 2j + s
// (lattice.h:204); its neighbours in the opposite half sit at
//   +x: (y, j+s)   -x: (y, j+s-1)   +y: (y+1, j)   -y: (y-1, j)       (all periodic)
// which is what the reference's double-row loops at :60-119 / :149-210 spell out.
template <typename T>
void cshift_half(T* lhs, const T* rhs, int cdir, int p_in, int dof, int Lx, int Ly) {
  const long hr = Lx / 2;
  const long half = hr * Ly;
  const int p_out = 1 - p_in;
  const T* src = rhs + (long)p_in * half * dof;
  T* dst = lhs + (long)p_out * half * dof;
  for (int y = 0; y < Ly; y++) {
    const int s = (y + p_out) & 1;
    long ys = y;
    if (cdir == QO_CSHIFT_FROM_YP1) ys = (y + 1) % Ly;
    if (cdir == QO_CSHIFT_FROM_YM1) ys = (y + Ly - 1) % Ly;
    for (long j = 0; j < hr; j++) {
      long js = j;
      if (cdir == QO_CSHIFT_FROM_XP1) js = (j + s) % hr;
      if (cdir == QO_CSHIFT_FROM_XM1) js = (j + s - 1 + hr) % hr;
      const T* a = src + (ys * hr + js) * dof;
      T* b = dst + ((long)y * hr + j) * dof;
      for (int k = 0; k < dof; k++) b[k] = a[k];
    }
  }
}

template <typename T>
int cshift_T(T* lhs, const T* rhs, int cdir, int eo, int dof, int Lx, int Ly) {
  if (Lx < 2 || Ly < 2 || (Lx & 1) || (Ly & 1)) return -1;   // x-shift loops step two rows (:62,:79)
  if (cdir < QO_CSHIFT_FROM_0 || cdir > QO_CSHIFT_FROM_YM1) return -2;  // distance-2 unsupported (:120-129)
  const long half = (long)(Lx / 2) * Ly;
  if (cdir == QO_CSHIFT_FROM_0) {
    // Reference quirk kept: copies half_size elements, not half_size*dof, and into the SAME half (:58,:147).
    if (eo & QO_EO_FROM_EVEN) for (long i = 0; i < half; i++) lhs[i] = rhs[i];
    if (eo & QO_EO_FROM_ODD) for (long i = 0; i < half; i++) lhs[half + i] = rhs[half + i];
    return 0;
  }
  if (eo & QO_EO_FROM_EVEN) cshift_half(lhs, rhs, cdir, 0, dof, Lx, Ly);
  if (eo & QO_EO_FROM_ODD) cshift_half(lhs, rhs, cdir, 1, dof, Lx, Ly);
  return 0;
}

const int DIR2CSHIFT[4] = {QO_CSHIFT_FROM_XP1, QO_CSHIFT_FROM_YP1, QO_CSHIFT_FROM_XM1, QO_CSHIFT_FROM_YM1};

}  // namespace

extern "C" {

// ---------------- lattice ----------------
int qo_coord_to_index(int Lx, int Ly, int x, int y) {   // lattice.h:75-81
  if (Lx * Ly == 1) return 0;
  int parity = (x + y) % 2;
  int i = (y + parity * Ly) * Lx / 2;
  return i + (x / 2) % (Lx / 2);
}
void qo_index_to_coord(int Lx, int Ly, int i, int* x, int* y) {  // lattice.h:199-205
  if (Lx * Ly == 1) { *x = *y = 0; return; }
  int parity = i / (Lx * Ly / 2);
  *y = i / (Lx / 2) - parity * Ly;
  *x = 2 * (i % (Lx / 2)) + ((*y) % 2 + parity) % 2;
}

int qo_cshift(double* lhs, const double* rhs, int cdir, int eo, int dof, int Lx, int Ly) {
  return cshift_T<cplx>(C(lhs), C(rhs), cdir, eo, dof, Lx, Ly);
}

// ---------------- stencil apply (stencil_2d.h:666-936) ----------------
// Sequence for QO_P_ALL == Stencil2D::apply_M (:912-936):
//   clover sweep (:694-703) ; eo: 4 x {cshift FROM_ODD, cMATxpy on even half} (:718-732) ;
//   oe: 4 x {cshift FROM_EVEN, cMATxpy on odd half} (:787-801) ; shifts (:865-909).
int qo_stencil_apply(const qo_stencil_desc* d, double* lhs_, const double* rhs_, unsigned pieces) {
  const int Lx = d->Lx, Ly = d->Ly, nc = d->nc;
  if (Lx == 1 && Ly == 1 && nc >= 1) {
    // volume == 1 (stencil_2d.h:870-888): every half-volume loop of the clover / hopping passes runs 0 times (their counts are volume / 2), so
    // only apply_M_shift acts, in its corner form: the site counts as even
    cplx* l = C(lhs_);
    const cplx* r = C(rhs_);
    const cplx sh(d->shift[0], d->shift[1]), eo(d->eo_shift[0], d->eo_shift[1]), ds(d->dof_shift[0], d->dof_shift[1]);
    if (pieces & (QO_P_ZERO_E | QO_P_ZERO_O)) for (int c = 0; c < nc; c++) l[c] = 0.0;
    if (pieces & QO_P_SHIFT_E) {
      if (nc % 2 == 0) {
        for (int c = 0; c < nc / 2; c++) { l[c] += (sh + eo + ds) * r[c]; l[c + nc / 2] += (sh + eo - ds) * r[c + nc / 2]; }
      } else {
        for (int c = 0; c < nc; c++) l[c] += (sh + eo) * r[c];
      }
    }
    return 0;
  }
  if (Lx < 2 || Ly < 2 || (Lx & 1) || (Ly & 1) || nc < 1) return -1;
  const long vol = (long)Lx * Ly, half_vol = vol / 2;
  const long size_cv = vol * nc, half_cv = size_cv / 2;
  const long size_cm = size_cv * nc, half_cm = size_cm / 2;
  cplx* lhs = C(lhs_);
  const cplx* rhs = C(rhs_);
  const cplx* clover = C(d->clover);
  const cplx* hopping = C(d->hopping);

  if (pieces & QO_P_ZERO_E) std::memset((void*)lhs, 0, sizeof(cplx) * half_cv);
  if (pieces & QO_P_ZERO_O) std::memset((void*)(lhs + half_cv), 0, sizeof(cplx) * half_cv);

  if (clover) {
    if (pieces & QO_P_CLOVER_E) cMATxpy(clover, rhs, lhs, half_vol, nc);
    if (pieces & QO_P_CLOVER_O) cMATxpy(clover + half_cm, rhs + half_cv, lhs + half_cv, half_vol, nc);
  }
  if (hopping && (pieces & QO_P_HOPPING)) {
    std::vector<cplx> priv((size_t)size_cv);   // priv_cvector (:126)
    for (int dir = 0; dir < 4; dir++)
      if (pieces & (QO_P_EO_XP1 << dir)) {
        cshift_T<cplx>(priv.data(), rhs, DIR2CSHIFT[dir], QO_EO_FROM_ODD, nc, Lx, Ly);
        cMATxpy(hopping + dir * size_cm, priv.data(), lhs, half_vol, nc);
      }
    for (int dir = 0; dir < 4; dir++)
      if (pieces & (QO_P_OE_XP1 << dir)) {
        cshift_T<cplx>(priv.data(), rhs, DIR2CSHIFT[dir], QO_EO_FROM_EVEN, nc, Lx, Ly);
        cMATxpy(hopping + dir * size_cm + half_cm, priv.data() + half_cv, lhs + half_cv, half_vol, nc);
      }
  }
  // apply_M_shift (:865-909)
  const cplx shift(d->shift[0], d->shift[1]), eo(d->eo_shift[0], d->eo_shift[1]), dofs(d->dof_shift[0], d->dof_shift[1]);
  if (pieces & QO_P_SHIFT_E) caxpy(shift + eo, rhs, lhs, half_cv);
  if (pieces & QO_P_SHIFT_O) caxpy(shift - eo, rhs + half_cv, lhs + half_cv, half_cv);
  if (dofs != 0.0 && nc % 2 == 0) {   // strided top/bottom-half shift (:897-908)
    for (int p = 0; p < 2; p++) {
      if (!(pieces & (QO_P_SHIFT_E << p))) continue;
      for (long i = p * half_vol; i < (p + 1) * half_vol; i++)
        for (int c = 0; c < nc; c++) cmac(lhs[i * nc + c], (c < nc / 2) ? dofs : -dofs, rhs[i * nc + c]);
    }
  }
  return 0;
}

double qo_time_apply(const qo_stencil_desc* d, double* lhs, const double* rhs, unsigned pieces, int reps) {
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; r++) qo_stencil_apply(d, lhs, rhs, pieces);
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}

// ---------------- operator fills ----------------
// Wilson2D::update_links (wilson.h:153-209).  gauge: (mu,eo,y,x) nc=1.
//  clover = 2w 1 ; H+x = 1/2 [[-w,1],[1,-w]] Ux(x) ; H+y = 1/2 [[-w,-i],[i,-w]] Uy(x)
//  H-x = 1/2 [[-w,-1],[-1,-w]] conj Ux(x-x^) ; H-y = 1/2 [[-w,i],[-i,-w]] conj Uy(x-y^)
int qo_wilson_fill(double* clover_, double* hopping_, const double* gauge_, int Lx, int Ly, double w) {
  if (Lx < 2 || Ly < 2 || (Lx & 1) || (Ly & 1)) return -1;
  const long vol = (long)Lx * Ly, cm = vol * 4;
  cplx* clover = C(clover_);
  cplx* hop = C(hopping_);
  const cplx* g = C(gauge_);
  const cplx I(0.0, 1.0);
  for (long i = 0; i < vol; i++) {
    clover[4 * i + 0] = 2.0 * w; clover[4 * i + 1] = 0.0; clover[4 * i + 2] = 0.0; clover[4 * i + 3] = 2.0 * w;
  }
  std::vector<cplx> back((size_t)vol);
  for (long i = 0; i < vol; i++) {
    cplx u = g[i];
    hop[4 * i + 0] = -0.5 * w * u; hop[4 * i + 1] = 0.5 * u; hop[4 * i + 2] = 0.5 * u; hop[4 * i + 3] = -0.5 * w * u;
    cplx v = g[vol + i];
    hop[cm + 4 * i + 0] = -0.5 * w * v; hop[cm + 4 * i + 1] = cmul(-0.5 * I, v);
    hop[cm + 4 * i + 2] = cmul(0.5 * I, v); hop[cm + 4 * i + 3] = -0.5 * w * v;
  }
  cshift_T<cplx>(back.data(), g, QO_CSHIFT_FROM_XM1, QO_EO_FROM_EVENODD, 1, Lx, Ly);   // :195
  for (long i = 0; i < vol; i++) {
    cplx u = std::conj(back[i]);
    hop[2 * cm + 4 * i + 0] = -0.5 * w * u; hop[2 * cm + 4 * i + 1] = -0.5 * u;
    hop[2 * cm + 4 * i + 2] = -0.5 * u; hop[2 * cm + 4 * i + 3] = -0.5 * w * u;
  }
  cshift_T<cplx>(back.data(), g + vol, QO_CSHIFT_FROM_YM1, QO_EO_FROM_EVENODD, 1, Lx, Ly);  // :204
  for (long i = 0; i < vol; i++) {
    cplx v = std::conj(back[i]);
    hop[3 * cm + 4 * i + 0] = -0.5 * w * v; hop[3 * cm + 4 * i + 1] = cmul(0.5 * I, v);
    hop[3 * cm + 4 * i + 2] = cmul(-0.5 * I, v); hop[3 * cm + 4 * i + 3] = -0.5 * w * v;
  }
  return 0;
}

// Staggered2D ctor (staggered.h:50-72): no clover; H+x = -1/2 Ux ; H+y = -1/2 eta Uy ;
// H-x = +1/2 conj Ux(x-x^) ; H-y = +1/2 eta conj Uy(x-y^) ; eta = 1 - 2 (x%2) (:253-259).
int qo_staggered_fill(double* hopping_, const double* gauge_, int Lx, int Ly) {
  if (Lx < 2 || Ly < 2 || (Lx & 1) || (Ly & 1)) return -1;
  const long vol = (long)Lx * Ly;
  cplx* hop = C(hopping_);
  const cplx* g = C(gauge_);
  std::vector<cplx> back((size_t)vol);
  for (long i = 0; i < vol; i++) {
    int x, y;
    qo_index_to_coord(Lx, Ly, (int)i, &x, &y);
    double eta = 1.0 - 2.0 * (x % 2);
    hop[i] = -0.5 * g[i];
    hop[vol + i] = (-0.5 * g[vol + i]) * eta;
  }
  cshift_T<cplx>(back.data(), g, QO_CSHIFT_FROM_XM1, QO_EO_FROM_EVENODD, 1, Lx, Ly);
  for (long i = 0; i < vol; i++) hop[2 * vol + i] = 0.5 * std::conj(back[i]);
  cshift_T<cplx>(back.data(), g + vol, QO_CSHIFT_FROM_YM1, QO_EO_FROM_EVENODD, 1, Lx, Ly);
  for (long i = 0; i < vol; i++) {
    int x, y;
    qo_index_to_coord(Lx, Ly, (int)i, &x, &y);
    double eta = 1.0 - 2.0 * (x % 2);
    hop[3 * vol + i] = (0.5 * std::conj(back[i])) * eta;
  }
  return 0;
}

// GaugedLaplace2D ctor (gaugedlaplace.h:45-68): clover 4 ; H+mu = -U_mu ; H-mu = -conj U_mu(x-mu).
int qo_laplace_fill(double* clover_, double* hopping_, const double* gauge_, int Lx, int Ly) {
  if (Lx < 2 || Ly < 2 || (Lx & 1) || (Ly & 1)) return -1;
  const long vol = (long)Lx * Ly;
  cplx* clover = C(clover_);
  cplx* hop = C(hopping_);
  const cplx* g = C(gauge_);
  std::vector<cplx> back((size_t)vol);
  for (long i = 0; i < vol; i++) { clover[i] = 4.0; hop[i] = -g[i]; hop[vol + i] = -g[vol + i]; }
  cshift_T<cplx>(back.data(), g, QO_CSHIFT_FROM_XM1, QO_EO_FROM_EVENODD, 1, Lx, Ly);
  for (long i = 0; i < vol; i++) hop[2 * vol + i] = -std::conj(back[i]);
  cshift_T<cplx>(back.data(), g + vol, QO_CSHIFT_FROM_YM1, QO_EO_FROM_EVENODD, 1, Lx, Ly);
  for (long i = 0; i < vol; i++) hop[3 * vol + i] = -std::conj(back[i]);
  return 0;
}

int qo_free_laplace_fill(double* clover_, double* hopping_, int Lx, int Ly) {  // free_laplace.h:39-41
  const long vol = (long)Lx * Ly;
  cplx* clover = C(clover_);
  cplx* hop = C(hopping_);
  for (long i = 0; i < vol; i++) clover[i] = 4.0;
  for (long i = 0; i < 4 * vol; i++) hop[i] = -1.0;
  return 0;
}

// ---------------- gauge I/O (u1_utils.h:38-67): one phase per line, x outer, y, mu inner ----------------
int qo_phases_to_gauge_u1(double* gauge_, const double* ph, int Lx, int Ly) {
  cplx* g = C(gauge_);
  const long vol = (long)Lx * Ly;
  long k = 0;
  for (int x = 0; x < Lx; x++)
    for (int y = 0; y < Ly; y++)
      for (int mu = 0; mu < 2; mu++) g[mu * vol + qo_coord_to_index(Lx, Ly, x, y)] = std::polar(1.0, ph[k++]);
  return 0;
}
int qo_read_gauge_u1(double* gauge_, int Lx, int Ly, const char* path) {
  FILE* f = std::fopen(path, "r");
  if (!f) return -1;
  std::vector<double> ph((size_t)2 * Lx * Ly);
  for (size_t k = 0; k < ph.size(); k++)
    if (std::fscanf(f, "%lf", &ph[k]) != 1) { std::fclose(f); return -2; }
  std::fclose(f);
  return qo_phases_to_gauge_u1(gauge_, ph.data(), Lx, Ly);
}
int qo_unit_gauge_u1(double* gauge_, int Lx, int Ly) {   // u1_utils.h:172-181
  cplx* g = C(gauge_);
  for (long i = 0; i < 2L * Lx * Ly; i++) g[i] = 1.0;
  return 0;
}

// ---------------- stencil variants ----------------
// build_dagger_stencil (stencil_2d.h:1080-1139): H^dag_{+x}(x) = [H_{-x}(x+x^)]^dag etc.
int qo_build_dagger(double* dclover_, double* dhopping_, const double* clover_, const double* hopping_, int Lx, int Ly, int nc) {
  const long vol = (long)Lx * Ly, cm = vol * nc * nc;
  const int nc2 = nc * nc;
  if (clover_ && dclover_) conj_transpose_sq(C(clover_), C(dclover_), vol, nc);
  if (hopping_ && dhopping_) {
    const cplx* hop = C(hopping_);
    cplx* dh = C(dhopping_);
    std::vector<cplx> priv((size_t)cm);
    const int src_dir[4] = {QO_DIR_XM1, QO_DIR_YM1, QO_DIR_XP1, QO_DIR_YP1};   // :1106,:1111,:1116,:1121
    for (int dir = 0; dir < 4; dir++) {
      if (cshift_T<cplx>(priv.data(), hop + src_dir[dir] * cm, DIR2CSHIFT[dir], QO_EO_FROM_EVENODD, nc2, Lx, Ly)) return -1;
      conj_transpose_sq(priv.data(), dh + dir * cm, vol, nc);
    }
  }
  return 0;
}

// build_rbjacobi_stencil (stencil_2d.h:1452-1601):
//   Cm = clover + diag(shift +- eo_shift +- dof_shift) (:1480-1528) ; cinv = Cm^-1 (:1536-1537)
//   rb clover = identity (:1543-1553) ; H'_mu(x) = H_mu(x) . cinv(x+mu) (:1556-1581)
int qo_build_rbjacobi(double* cinv_, double* rclover_, double* rhopping_, const qo_stencil_desc* d) {
  const int Lx = d->Lx, Ly = d->Ly, nc = d->nc, nc2 = nc * nc;
  const long vol = (long)Lx * Ly, cm = vol * nc2, half_vol = vol / 2;
  const cplx shift(d->shift[0], d->shift[1]), eo(d->eo_shift[0], d->eo_shift[1]), dofs(d->dof_shift[0], d->dof_shift[1]);
  if (!d->clover && shift == 0.0 && eo == 0.0 && dofs == 0.0) return -1;   // :1471-1475
  std::vector<cplx> cmat((size_t)cm, cplx(0.0));
  if (d->clover) std::memcpy((void*)cmat.data(), d->clover, sizeof(cplx) * cm);
  for (long i = 0; i < vol; i++) {
    const bool odd = (i >= half_vol);
    for (int r = 0; r < nc; r++) {
      cplx m = odd ? (shift - eo) : (shift + eo);
      if (nc % 2 == 0) m += (r * (nc + 1) < nc2 / 2) ? dofs : -dofs;   // flat diagonal index < nc2/2 (:1501)
      cmat[i * nc2 + r * nc + r] += m;
    }
  }
  cplx* cinv = C(cinv_);
  if (inverse_sq(cmat.data(), cinv, vol, nc)) return -2;
  if (rclover_) {
    cplx* rc = C(rclover_);
    for (long i = 0; i < cm; i++) rc[i] = 0.0;
    for (long i = 0; i < vol; i++)
      for (int r = 0; r < nc; r++) rc[i * nc2 + r * nc + r] = 1.0;
  }
  if (d->hopping && rhopping_) {
    const cplx* hop = C(d->hopping);
    cplx* rh = C(rhopping_);
    std::vector<cplx> priv((size_t)cm), tmp((size_t)cm);
    const int back_dir[4] = {QO_CSHIFT_FROM_XM1, QO_CSHIFT_FROM_YM1, QO_CSHIFT_FROM_XP1, QO_CSHIFT_FROM_YP1};
    for (int dir = 0; dir < 4; dir++) {
      if (cshift_T<cplx>(priv.data(), hop + dir * cm, back_dir[dir], QO_EO_FROM_EVENODD, nc2, Lx, Ly)) return -3;
      matmul_sq(priv.data(), cinv, tmp.data(), vol, nc);
      cshift_T<cplx>(rh + dir * cm, tmp.data(), DIR2CSHIFT[dir], QO_EO_FROM_EVENODD, nc2, Lx, Ly);
    }
  }
  return 0;
}

// build_rbj_dagger_stencil (stencil_2d.h:1989-2060): dagger of the rbjacobi stencil + cinv^dag.
int qo_build_rbj_dagger(double* dcinv, double* dclover, double* dhopping,
                        const double* cinv, const double* rclover, const double* rhopping, int Lx, int Ly, int nc) {
  if (cinv && dcinv) conj_transpose_sq(C(cinv), C(dcinv), (long)Lx * Ly, nc);
  return qo_build_dagger(dclover, dhopping, rclover, rhopping, Lx, Ly, nc);
}

// ---------------- reductions ----------------
double qo_norm2sq(const double* x, long n) {
  double s = 0.0;
  for (long i = 0; i < 2 * n; i++) s += x[i] * x[i];
  return s;
}
void qo_dot(const double* x_, const double* y_, long n, double out[2]) {
  const cplx* x = C(x_); const cplx* y = C(y_);
  cplx s = 0.0;
  for (long i = 0; i < n; i++) cmac(s, std::conj(x[i]), y[i]);
  out[0] = s.real(); out[1] = s.imag();
}
double qo_diffnorm2sq(const double* x, const double* y, long n) {
  double s = 0.0;
  for (long i = 0; i < 2 * n; i++) { double dlt = x[i] - y[i]; s += dlt * dlt; }
  return s;
}
double qo_norminf(const double* x_, long n) {
  const cplx* x = C(x_);
  double m = 0.0;
  for (long i = 0; i < n; i++) { double a = std::abs(x[i]); if (a > m) m = a; }
  return m;
}
void qo_norm2sq_cv_timeslice(double* sum, const double* cv_, int Lx, int Ly, int nc) {   // reductions.h:24-41
  const cplx* cv = C(cv_);
  for (int t = 0; t < Ly; t++) sum[t] = 0.0;
  const long size_cv = (long)Lx * Ly * nc;
  for (long i = 0; i < size_cv; i++) {
    int x, y;
    qo_index_to_coord(Lx, Ly, (int)(i / nc), &x, &y);
    sum[y] += std::norm(cv[i]);
  }
}
void qo_dot_cv_timeslice(double* sum, const double* a_, const double* b_, int Lx, int Ly, int nc) {   // reductions.h:69-87
  const cplx* a = C(a_); const cplx* b = C(b_);
  cplx* s = C(sum);
  for (int t = 0; t < Ly; t++) s[t] = 0.0;
  const long size_cv = (long)Lx * Ly * nc;
  for (long i = 0; i < size_cv; i++) {
    int x, y;
    qo_index_to_coord(Lx, Ly, (int)(i / nc), &x, &y);
    cmac(s[y], std::conj(a[i]), b[i]);
  }
}

void qo_redot_cv_timeslice(double* sum, const double* a_, const double* b_, int Lx, int Ly, int nc) {   // reductions.h:47-66
  const cplx* a = C(a_); const cplx* b = C(b_);
  for (int t = 0; t < Ly; t++) sum[t] = 0.0;
  const long size_cv = (long)Lx * Ly * nc;
  for (long i = 0; i < size_cv; i++) {
    int x, y;
    qo_index_to_coord(Lx, Ly, (int)(i / nc), &x, &y);
    sum[y] += std::real(std::conj(a[i]) * b[i]);
  }
}

// gaussian_wall_source (reductions.h:90-162): loop over every element, cv_index_to_coord, a real draw where c == color and
// y == timeslice, zero elsewhere.  Draw of element i: splitmix64 counter -> two uniforms -> Box-Muller (real part).
static unsigned long long qo_splitmix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
int qo_gaussian_wall_source(double* cv_, int Lx, int Ly, int nc, int timeslice, int color, unsigned long long seed, double deviation, double mean) {
  if (timeslice >= Ly || timeslice < 0) return -1;   // "[QMG-ERROR]: Cannot create gaussian wall source for t < Nt." (:94-98)
  if (color >= nc || color < 0) return -1;           // (:101-105)
  cplx* cv = C(cv_);
  const long size_cv = (long)Lx * Ly * nc;
  for (long i = 0; i < size_cv; i++) {
    int x, y;
    qo_index_to_coord(Lx, Ly, (int)(i / nc), &x, &y);
    const int c = (int)(i % nc);
    if (c == color && y == timeslice) {
      const unsigned long long h1 = qo_splitmix64(seed * 0xD1342543DE82EF95ull + 2ull * (unsigned long long)i);
      const unsigned long long h2 = qo_splitmix64(h1 + 2ull * (unsigned long long)i + 1ull);
      const double u1 = ((double)(h1 >> 11) + 1.0) * (1.0 / 9007199254740992.0);
      const double u2 = (double)(h2 >> 11) * (1.0 / 9007199254740992.0);
      cv[i] = cplx(mean + deviation * (std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925286766559 * u2)), 0.0);
    } else cv[i] = 0.0;
  }
  return 0;
}

// ---------------- transfer ----------------
// build_mapping (transfer.h:410-448): for coarse site i (even-odd index), the fine cv indices of the
// block [cx*bx,(cx+1)*bx) x [cy*by,(cy+1)*by) x nc_f, sorted ascending (the reference merge-sorts, :440).
int qo_transfer_build_map(int* map, int fLx, int fLy, int fnc, int cLx, int cLy) {
  if (fLx % cLx || fLy % cLy) return -1;
  const int bx = fLx / cLx, by = fLy / cLy;
  const int per = bx * by * fnc;
  const int cvol = cLx * cLy;
  std::vector<int> list((size_t)per);
  for (int i = 0; i < cvol; i++) {
    int cx, cy;
    qo_index_to_coord(cLx, cLy, i, &cx, &cy);
    int n = 0;
    for (int x = cx * bx; x < (cx + 1) * bx; x++)
      for (int y = cy * by; y < (cy + 1) * by; y++)
        for (int c = 0; c < fnc; c++) list[n++] = fnc * qo_coord_to_index(fLx, fLy, x, y) + c;
    // insertion sort (small lists; order is all that matters)
    for (int a = 1; a < per; a++) {
      int v = list[a], b = a - 1;
      while (b >= 0 && list[b] > v) { list[b + 1] = list[b]; b--; }
      list[b + 1] = v;
    }
    std::memcpy(map + (size_t)i * per, list.data(), sizeof(int) * per);
  }
  return per;
}

namespace {
struct MapCache {   // the reference builds coarse_map once per TransferMG; cache the last one here.
  int fLx = 0, fLy = 0, fnc = 0, cLx = 0, cLy = 0, per = 0;
  std::vector<int> map;
  const int* get(int a, int b, int c, int d, int e) {
    if (a != fLx || b != fLy || c != fnc || d != cLx || e != cLy) {
      fLx = a; fLy = b; fnc = c; cLx = d; cLy = e;
      map.assign((size_t)cLx * cLy * (fLx / cLx) * (fLy / cLy) * fnc, 0);
      per = qo_transfer_build_map(map.data(), fLx, fLy, fnc, cLx, cLy);
    }
    return map.data();
  }
};
MapCache g_map;
}  // namespace

// prolong_c2f (transfer.h:455-480). nvec may be < cnc (block-ortho abuse, :552-598): coarse index is cnc*i + d.
int qo_prolong(const double* nullvecs_, int nvec, const double* coarse_, double* fine_,
               int fLx, int fLy, int fnc, int cLx, int cLy, int cnc) {
  if (fLx % cLx || fLy % cLy) return -1;
  const int* map = g_map.get(fLx, fLy, fnc, cLx, cLy);
  const int per = g_map.per;
  const long fsize = (long)fLx * fLy * fnc;
  const cplx* nv = C(nullvecs_); const cplx* coarse = C(coarse_); cplx* fine = C(fine_);
  const int cvol = cLx * cLy;
  for (int i = 0; i < cvol; i++)
    for (int dd = 0; dd < nvec; dd++) {
      const cplx cval = coarse[(long)cnc * i + dd];
      const cplx* v = nv + dd * fsize;
      const int* m = map + (size_t)i * per;
      for (int j = 0; j < per; j++) cmac(fine[m[j]], v[m[j]], cval);
    }
  return 0;
}
// restrict_f2c (transfer.h:487-511)
int qo_restrict(const double* nullvecs_, int nvec, const double* fine_, double* coarse_,
                int fLx, int fLy, int fnc, int cLx, int cLy, int cnc) {
  if (fLx % cLx || fLy % cLy) return -1;
  const int* map = g_map.get(fLx, fLy, fnc, cLx, cLy);
  const int per = g_map.per;
  const long fsize = (long)fLx * fLy * fnc;
  const cplx* nv = C(nullvecs_); const cplx* fine = C(fine_); cplx* coarse = C(coarse_);
  const int cvol = cLx * cLy;
  for (int i = 0; i < cvol; i++)
    for (int dd = 0; dd < nvec; dd++) {
      cplx acc = coarse[(long)cnc * i + dd];
      const cplx* v = nv + dd * fsize;
      const int* m = map + (size_t)i * per;
      for (int j = 0; j < per; j++) cmac(acc, std::conj(v[m[j]]), fine[m[j]]);
      coarse[(long)cnc * i + dd] = acc;
    }
  return 0;
}

// block_orthonormalize (transfer.h:514-607): classical Gram-Schmidt per block written as
// restrict/prolong with a single vector; nvec == coarse nc. One pass (the ctor calls it twice, :160-174).
int qo_block_orthonormalize(double* nullvecs_, int nvec, int fLx, int fLy, int fnc, int cLx, int cLy, double* chol_) {
  const long fsize = (long)fLx * fLy * fnc;
  const int cvol = cLx * cLy, cnc = nvec;
  const long csize = (long)cvol * cnc;
  cplx* nv = C(nullvecs_);
  cplx* chol = chol_ ? C(chol_) : nullptr;
  std::vector<cplx> fine1((size_t)fsize), coarse2((size_t)csize);
  for (int i = 0; i < nvec; i++) {
    for (int j = 0; j < i; j++) {
      std::fill(fine1.begin(), fine1.end(), cplx(0.0));
      std::fill(coarse2.begin(), coarse2.end(), cplx(0.0));
      qo_restrict((double*)(nv + j * fsize), 1, (double*)(nv + i * fsize), (double*)coarse2.data(), fLx, fLy, fnc, cLx, cLy, cnc);
      if (chol)   // copy_vector_blas(block_cholesky + j*cnc + i, cnc*cnc, coarse_cv_2, cnc, cvol) (:560)
        for (int s = 0; s < cvol; s++) chol[(long)s * cnc * cnc + j * cnc + i] = coarse2[(long)s * cnc];
      qo_prolong((double*)(nv + j * fsize), 1, (double*)coarse2.data(), (double*)fine1.data(), fLx, fLy, fnc, cLx, cLy, cnc);
      caxpy(-1.0, fine1.data(), nv + i * fsize, fsize);
    }
    std::fill(fine1.begin(), fine1.end(), cplx(0.0));
    std::fill(coarse2.begin(), coarse2.end(), cplx(0.0));
    qo_restrict((double*)(nv + i * fsize), 1, (double*)(nv + i * fsize), (double*)coarse2.data(), fLx, fLy, fnc, cLx, cLy, cnc);
    for (long k = 0; k < csize; k++) coarse2[k] = cplx(1.0 / std::sqrt(coarse2[k].real()), 0.0);   // inv_real_sqrt over ALL entries (:583)
    if (chol)
      for (int s = 0; s < cvol; s++) chol[(long)s * cnc * cnc + i * (cnc + 1)] = 1.0 / coarse2[(long)s * cnc];   // :588-593
    qo_prolong((double*)(nv + i * fsize), 1, (double*)coarse2.data(), (double*)fine1.data(), fLx, fLy, fnc, cLx, cLy, cnc);
    std::memcpy((void*)(nv + i * fsize), fine1.data(), sizeof(cplx) * fsize);
  }
  return 0;
}

// block_bi_orthonormalize (transfer.h:610-769): two-sided Gram-Schmidt per block, written with single-vector
// restrict / prolong; normalisation splits <r_i,p_i> = |z| e^{i phi} as r_i <- r_i e^{i phi}/sqrt|z| (restrict conjugates
// it), p_i <- p_i / sqrt|z|.  L collects <p_j, r_i> (conjugated at the end, :757), U collects <r_j, p_i>.
int qo_block_bi_orthonormalize(double* pvecs_, double* rvecs_, int nvec, int fLx, int fLy, int fnc, int cLx, int cLy, double* L_, double* U_) {
  const long fsize = (long)fLx * fLy * fnc;
  const int cvol = cLx * cLy, cnc = nvec;
  const long csize = (long)cvol * cnc;
  cplx* P = C(pvecs_);
  cplx* R = C(rvecs_);
  cplx* Lm = L_ ? C(L_) : nullptr;
  cplx* Um = U_ ? C(U_) : nullptr;
  std::vector<cplx> fine1((size_t)fsize), coarse2((size_t)csize);
  auto zero = [&]() { std::fill(fine1.begin(), fine1.end(), cplx(0.0)); std::fill(coarse2.begin(), coarse2.end(), cplx(0.0)); };
  for (int i = 0; i < nvec; i++) {
    for (int j = 0; j < i; j++) {
      zero();
      qo_restrict((double*)(R + j * fsize), 1, (double*)(P + i * fsize), (double*)coarse2.data(), fLx, fLy, fnc, cLx, cLy, cnc);
      if (Um) for (int s = 0; s < cvol; s++) Um[(long)s * cnc * cnc + j * cnc + i] = coarse2[(long)s * cnc];
      qo_prolong((double*)(P + j * fsize), 1, (double*)coarse2.data(), (double*)fine1.data(), fLx, fLy, fnc, cLx, cLy, cnc);
      caxpy(-1.0, fine1.data(), P + i * fsize, fsize);
      zero();
      qo_restrict((double*)(P + j * fsize), 1, (double*)(R + i * fsize), (double*)coarse2.data(), fLx, fLy, fnc, cLx, cLy, cnc);
      if (Lm) for (int s = 0; s < cvol; s++) Lm[(long)s * cnc * cnc + i * cnc + j] = coarse2[(long)s * cnc];
      qo_prolong((double*)(R + j * fsize), 1, (double*)coarse2.data(), (double*)fine1.data(), fLx, fLy, fnc, cLx, cLy, cnc);
      caxpy(-1.0, fine1.data(), R + i * fsize, fsize);
    }
    zero();
    qo_restrict((double*)(R + i * fsize), 1, (double*)(P + i * fsize), (double*)coarse2.data(), fLx, fLy, fnc, cLx, cLy, cnc);
    for (int s = 0; s < cvol; s++) {   // only colour 0 of each coarse site is used below
      const cplx z = coarse2[(long)s * cnc];
      coarse2[(long)s * cnc] = std::polar(1.0 / std::sqrt(std::abs(z)), std::arg(z));
    }
    if (Lm) for (int s = 0; s < cvol; s++) Lm[(long)s * cnc * cnc + i * (cnc + 1)] = 1.0 / coarse2[(long)s * cnc];
    qo_prolong((double*)(R + i * fsize), 1, (double*)coarse2.data(), (double*)fine1.data(), fLx, fLy, fnc, cLx, cLy, cnc);
    std::memcpy((void*)(R + i * fsize), fine1.data(), sizeof(cplx) * fsize);
    std::fill(fine1.begin(), fine1.end(), cplx(0.0));
    for (int s = 0; s < cvol; s++) coarse2[(long)s * cnc] = std::abs(coarse2[(long)s * cnc]);
    if (Um) for (int s = 0; s < cvol; s++) Um[(long)s * cnc * cnc + i * (cnc + 1)] = 1.0 / coarse2[(long)s * cnc];
    qo_prolong((double*)(P + i * fsize), 1, (double*)coarse2.data(), (double*)fine1.data(), fLx, fLy, fnc, cLx, cLy, cnc);
    std::memcpy((void*)(P + i * fsize), fine1.data(), sizeof(cplx) * fsize);
  }
  if (Lm) for (long k = 0; k < (long)cvol * cnc * cnc; k++) Lm[k] = std::conj(Lm[k]);
  return 0;
}

// ---------------- Galerkin coarse operator (coarse.h:90-444) ----------------
// For each coarse colour: one clover probe (all coarse sites) and, per direction and per source
// parity, one hopping probe.  Same-parity results go to the coarse clover, other-parity results to
// hopping[dir]; a coarse dimension of length 1 folds everything into the clover (:226-233).
// Shifts are NOT probed (apply_M_clover / apply_M_hopping exclude them): the caller copies `shift`.
int qo_coarse_build(double* cclover_, double* chopping_, const qo_stencil_desc* f,
                    const double* nullvecs, const double* restrict_vecs, int cLx, int cLy, int cnc) {
  const int fLx = f->Lx, fLy = f->Ly, fnc = f->nc;
  const long fsize = (long)fLx * fLy * fnc;
  const int cvol = cLx * cLy;
  if (cvol == 1) return -9;   // volume-1 corner case (:146-156,:195-205) not restated
  const long csize = (long)cvol * cnc, ccm = csize * cnc;
  const double* rvecs = restrict_vecs ? restrict_vecs : nullvecs;
  cplx* cclover = C(cclover_);
  cplx* chop = C(chopping_);
  for (long i = 0; i < ccm; i++) cclover[i] = 0.0;
  for (long i = 0; i < 4 * ccm; i++) chop[i] = 0.0;
  std::vector<cplx> tc((size_t)csize), tf((size_t)fsize), taf((size_t)fsize);
  auto probe = [&](int color, int lo, int hi, unsigned pieces) {
    std::fill(tc.begin(), tc.end(), cplx(0.0));
    std::fill(tf.begin(), tf.end(), cplx(0.0));
    std::fill(taf.begin(), taf.end(), cplx(0.0));
    for (int i = lo; i < hi; i++) tc[(long)i * cnc + color] = 1.0;
    qo_prolong(nullvecs, cnc, (double*)tc.data(), (double*)tf.data(), fLx, fLy, fnc, cLx, cLy, cnc);
    qo_stencil_apply(f, (double*)taf.data(), (double*)tf.data(), pieces);
    std::fill(tc.begin(), tc.end(), cplx(0.0));
    qo_restrict(rvecs, cnc, (double*)taf.data(), (double*)tc.data(), fLx, fLy, fnc, cLx, cLy, cnc);
  };
  for (int color = 0; color < cnc; color++) {
    probe(color, 0, cvol, QO_P_CLOVER);
    for (int i = 0; i < cvol; i++)
      for (int c = 0; c < cnc; c++) cclover[((long)i * cnc + c) * cnc + color] += tc[(long)i * cnc + c];
    for (int dir = 0; dir < 4; dir++) {
      const unsigned pieces = (QO_P_EO_XP1 << dir) | (QO_P_OE_XP1 << dir);
      const bool fold = ((dir & 1) == 0) ? (cLx == 1) : (cLy == 1);
      for (int par = 0; par < 2; par++) {
        const int lo = par * cvol / 2, hi = lo + cvol / 2;
        probe(color, lo, hi, pieces);
        for (int i = 0; i < cvol; i++) {
          const bool same = (i >= lo && i < hi);
          for (int c = 0; c < cnc; c++) {
            const cplx v = tc[(long)i * cnc + c];
            if (same || fold) cclover[((long)i * cnc + c) * cnc + color] += v;
            else chop[dir * ccm + ((long)i * cnc + c) * cnc + color] += v;
          }
        }
      }
    }
  }
  return 0;
}

}  // extern "C"
