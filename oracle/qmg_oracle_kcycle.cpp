// qmg_oracle_kcycle.cpp -- CPU ORACLE (test infrastructure) for the K-cycle:
// StatefulMultigridMG::mg_preconditioner (multigrid/stateful_multigrid.h:734-1060) driven by a restarted
// flexible GCR, with MR smoothing and a restarted-GCR coarsest solve, exactly as
// tests/n13_wilson_kcycle/wilson_kcycle.cpp:86-122,459-471 configures it.
//
// PARITY UNPINNED for the Krylov drivers: quantum-linalg (minv_vector_gcr_var_precond_restart,
// minv_vector_minres, minv_vector_gcr_restart) is absent and stores no outputs; these are the textbook
// algorithms under the call-site conventions (relative tolerance against ||b||, ops_count = operator
// applications).  What this file pins is the HIP path's K-cycle against an independent CPU statement of
// the same recursion: same hierarchy (the null vectors are INPUTS), same parameters, iteration counts and
// solution compared by tests/test_gpu_kcycle.py.
#include <cmath>
#include <complex>
#include <cstring>
#include <vector>

#include "qmg_oracle.h"

typedef std::complex<double> cplx;
typedef std::vector<cplx> cvec;

namespace {

struct Level {
  int Lx, Ly, nc;
  long size;
  cvec clover, hopping;          // this level's operator
  cplx shift;
  cvec nullv;                    // transfer to the NEXT level: nvec(next nc) x size, block-orthonormalised
  qo_stencil_desc desc() const {
    qo_stencil_desc d;
    d.Lx = Lx; d.Ly = Ly; d.nc = nc;
    d.clover = (const double*)clover.data();
    d.hopping = (const double*)hopping.data();
    d.shift[0] = shift.real(); d.shift[1] = shift.imag();
    d.eo_shift[0] = d.eo_shift[1] = d.dof_shift[0] = d.dof_shift[1] = 0.0;
    return d;
  }
};

struct Params {
  int n_pre, n_post;
  double inner_tol; int inner_max_iter, inner_restart;
  double coarsest_tol; int coarsest_max_iter, coarsest_restart;
  double omega;
};

struct MG {
  std::vector<Level> lv;
  Params p;
  std::vector<long> ops;     // operator applications per level
  std::vector<long> iters;   // Krylov iterations per level (as add_iterations_count)
};

double norm2(const cvec& v, long n) { return qo_norm2sq((const double*)v.data(), n); }
cplx cdot(const cvec& a, const cvec& b, long n) { double o[2]; qo_dot((const double*)a.data(), (const double*)b.data(), n, o); return cplx(o[0], o[1]); }
void axpy(cplx a, const cvec& x, cvec& y, long n) { for (long i = 0; i < n; i++) y[i] += a * x[i]; }

void apply(MG& mg, int l, cvec& out, const cvec& in) {   // out = A_l in   (apply_stencil_2D_M)
  qo_stencil_desc d = mg.lv[l].desc();
  qo_stencil_apply(&d, (double*)out.data(), (const double*)in.data(), QO_P_ALL | QO_P_ZERO);
  mg.ops[l]++;
}

typedef void (*precond_fn)(MG&, int, cvec&, const cvec&);

// minv_vector_minres(x, b, n, iters, tol, omega, op): r = b - A x ; p = A r ; alpha = <p,r>/<p,p> ; x += omega alpha r ; r -= omega alpha p
int minres(MG& mg, int l, cvec& x, const cvec& b, int max_iter, double eps, double omega) {
  const long n = mg.lv[l].size;
  cvec r(n), p(n);
  const double bnorm = std::sqrt(norm2(b, n));
  apply(mg, l, p, x);
  for (long i = 0; i < n; i++) r[i] = b[i] - p[i];
  double rsq = norm2(r, n);
  int k = 0;
  bool conv = (bnorm == 0.0) || (std::sqrt(rsq) < eps * bnorm);
  while (!conv && k < max_iter) {
    apply(mg, l, p, r);
    const cplx pr = cdot(p, r, n);
    const double pp = norm2(p, n);
    if (pp == 0.0) break;
    const cplx alpha = omega * pr / pp;
    axpy(alpha, r, x, n);
    axpy(-alpha, p, r, n);
    rsq = norm2(r, n);
    k++;
    if (std::sqrt(rsq) < eps * bnorm) conv = true;
  }
  return k;
}

// restarted flexible GCR (see quantum-mg_amd/include/qmg/krylov.hpp: same algorithm, CPU vectors)
int gcr(MG& mg, int l, cvec& x, const cvec& b, int max_iter, double eps, int restart, precond_fn prec, double* rsq_out) {
  const long n = mg.lv[l].size;
  const int basis_max = (restart > 0) ? restart : max_iter;
  cvec r(n), tmp(n);
  std::vector<cvec> Z, W;
  std::vector<double> Wn;
  const double bnorm = std::sqrt(norm2(b, n));
  apply(mg, l, tmp, x);
  for (long i = 0; i < n; i++) r[i] = b[i] - tmp[i];
  double rsq = norm2(r, n);
  bool conv = (bnorm == 0.0) || (std::sqrt(rsq) < eps * bnorm);
  int k = 0, kb = 0;
  while (!conv && k < max_iter) {
    if (kb == (int)Z.size()) { Z.push_back(cvec(n)); W.push_back(cvec(n)); Wn.push_back(0.0); }
    cvec& z = Z[kb];
    cvec& w = W[kb];
    if (prec) { std::fill(z.begin(), z.end(), cplx(0.0)); prec(mg, l, z, r); }
    else z = r;
    apply(mg, l, w, z);
    if (kb > 0) {
      std::vector<cplx> c(kb);
      for (int i = 0; i < kb; i++) c[i] = cdot(W[i], w, n);       // all dots first (one fused pass on the device)
      for (int i = 0; i < kb; i++) {
        const cplx beta = c[i] / Wn[i];
        axpy(-beta, W[i], w, n);
        axpy(-beta, Z[i], z, n);
      }
    }
    const double ww = norm2(w, n);
    if (ww == 0.0) break;
    Wn[kb] = ww;
    const cplx alpha = cdot(w, r, n) / ww;
    axpy(alpha, z, x, n);
    axpy(-alpha, w, r, n);
    rsq = norm2(r, n);
    k++; kb++;
    if (std::sqrt(rsq) < eps * bnorm) { conv = true; break; }
    if (kb == basis_max) {
      apply(mg, l, tmp, x);
      for (long i = 0; i < n; i++) r[i] = b[i] - tmp[i];
      rsq = norm2(r, n);
      kb = 0;
      if (std::sqrt(rsq) < eps * bnorm) { conv = true; break; }
    }
  }
  if (rsq_out) *rsq_out = rsq;
  return conv ? k : -k - 1;   // negative = not converged
}

// mg_preconditioner (stateful_multigrid.h:734-1060), QMG_MATVEC_ORIGINAL on every level (n13:426)
void kcycle(MG& mg, int level, cvec& lhs, const cvec& rhs) {
  const int nlev = (int)mg.lv.size();
  Level& F = mg.lv[level];
  const long fn = F.size;
  if (nlev == 1) { lhs = rhs; return; }
  Level& Cc = mg.lv[level + 1];
  const long cn = Cc.size;
  cvec Atmp(fn), z1(fn, cplx(0.0)), r1(fn);
  // 1. pre-smooth (:845-873)
  if (mg.p.n_pre > 0) {
    minres(mg, level, z1, rhs, mg.p.n_pre, 1e-15, mg.p.omega);
    apply(mg, level, Atmp, z1);
    for (long i = 0; i < fn; i++) r1[i] = rhs[i] - Atmp[i];
  } else { r1 = rhs; z1 = rhs; }
  // 2. restrict, coarse solve, (prepare/reconstruct are copies for ORIGINAL) (:875-1002)
  cvec r_coarse(cn, cplx(0.0));
  qo_restrict((const double*)F.nullv.data(), Cc.nc, (const double*)r1.data(), (double*)r_coarse.data(), F.Lx, F.Ly, F.nc, Cc.Lx, Cc.Ly, Cc.nc);
  cvec e_coarse(cn, cplx(0.0));
  int it;
  if (level == nlev - 2) {
    it = gcr(mg, level + 1, e_coarse, r_coarse, mg.p.coarsest_max_iter, mg.p.coarsest_tol, mg.p.coarsest_restart, nullptr, nullptr);
  } else {
    it = gcr(mg, level + 1, e_coarse, r_coarse, mg.p.inner_max_iter, mg.p.inner_tol, mg.p.inner_restart, [](MG& m, int l, cvec& o, const cvec& i) { kcycle(m, l, o, i); }, nullptr);
  }
  mg.iters[level + 1] += (it >= 0) ? it : (-it - 1);
  // 3. prolong and correct (:1013-1021)
  cvec z2(fn, cplx(0.0));
  qo_prolong((const double*)F.nullv.data(), Cc.nc, (const double*)e_coarse.data(), (double*)z2.data(), F.Lx, F.Ly, F.nc, Cc.Lx, Cc.Ly, Cc.nc);
  for (long i = 0; i < fn; i++) lhs[i] = z1[i] + z2[i];
  // 4. post-smooth (:1023-1056)
  if (mg.p.n_post > 0) {
    apply(mg, level, Atmp, lhs);
    cvec r2(fn), z3(fn, cplx(0.0));
    for (long i = 0; i < fn; i++) r2[i] = rhs[i] - Atmp[i];
    minres(mg, level, z3, r2, mg.p.n_post, 1e-15, mg.p.omega);
    for (long i = 0; i < fn; i++) lhs[i] += z3[i];
  }
}

}  // namespace

extern "C" {

// Wilson K-cycle on an L x L lattice with n_refine 4x4 coarsenings to `coarse_dof` colours per level.
//   gauge        nc=1 LatticeGauge, 2 L^2 complex
//   nullvecs[l]  coarse_dof vectors of level-l size (NOT yet block-orthonormalised), l = 0..n_refine-1
//   b            right-hand side; x_out receives the solution
// Setup follows the n13 driver: TransferMG (two block-ortho passes, transfer.h:160-174), CoarseOperator2D
// (Galerkin probes, shift copied: coarse.h:131), outer VPGCR tol/1000/restart 32.
// Returns outer iterations (negative: not converged); fills true_res, ops[level], its[level].
int qo_wilson_kcycle(int L, double mass, int n_refine, int coarse_dof, const double* gauge, const double* const* nullvecs, const double* b_,
                     double tol, int max_iter, int restart, double inner_tol, double coarsest_tol, int n_smooth, double* x_out_,
                     double* true_res, long* ops, long* its) {
  MG mg;
  mg.p.n_pre = mg.p.n_post = n_smooth;
  mg.p.inner_tol = inner_tol; mg.p.inner_max_iter = 1000; mg.p.inner_restart = 32;
  mg.p.coarsest_tol = coarsest_tol; mg.p.coarsest_max_iter = 1000; mg.p.coarsest_restart = 32;
  mg.p.omega = 0.85;
  mg.lv.resize(n_refine + 1);
  Level& f = mg.lv[0];
  f.Lx = f.Ly = L; f.nc = 2; f.size = (long)L * L * 2; f.shift = mass;
  f.clover.resize((size_t)L * L * 4); f.hopping.resize((size_t)L * L * 16);
  if (qo_wilson_fill((double*)f.clover.data(), (double*)f.hopping.data(), gauge, L, L, 1.0)) return -100000;
  int cl = L;
  for (int i = 1; i <= n_refine; i++) {
    cl /= 4;
    Level& F = mg.lv[i - 1];
    Level& Cc = mg.lv[i];
    Cc.Lx = Cc.Ly = cl; Cc.nc = coarse_dof; Cc.size = (long)cl * cl * coarse_dof; Cc.shift = F.shift;
    F.nullv.assign((const cplx*)nullvecs[i - 1], (const cplx*)nullvecs[i - 1] + (size_t)coarse_dof * F.size);
    for (int pass = 0; pass < 2; pass++)
      if (qo_block_orthonormalize((double*)F.nullv.data(), coarse_dof, F.Lx, F.Ly, F.nc, cl, cl, nullptr)) return -100001;
    Cc.clover.resize((size_t)cl * cl * coarse_dof * coarse_dof);
    Cc.hopping.resize(4 * Cc.clover.size());
    qo_stencil_desc fd = F.desc();
    if (qo_coarse_build((double*)Cc.clover.data(), (double*)Cc.hopping.data(), &fd, (const double*)F.nullv.data(), nullptr, cl, cl, coarse_dof)) return -100002;
  }
  mg.ops.assign(n_refine + 1, 0);
  mg.iters.assign(n_refine + 1, 0);
  const long n = f.size;
  cvec b((const cplx*)b_, (const cplx*)b_ + n), x(n, cplx(0.0));
  double rsq = 0.0;
  int it = gcr(mg, 0, x, b, max_iter, tol, restart, [](MG& m, int l, cvec& o, const cvec& i) { kcycle(m, l, o, i); }, &rsq);
  cvec Ax(n);
  apply(mg, 0, Ax, x);
  *true_res = std::sqrt(qo_diffnorm2sq((const double*)b.data(), (const double*)Ax.data(), n) / norm2(b, n));
  std::memcpy(x_out_, x.data(), sizeof(cplx) * n);
  for (int i = 0; i <= n_refine; i++) { ops[i] = mg.ops[i]; its[i] = mg.iters[i]; }
  return it;
}

}  // extern "C"
