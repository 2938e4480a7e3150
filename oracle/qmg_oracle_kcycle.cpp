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
#include <algorithm>
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
  std::vector<double>* outer_hist = nullptr;     // relative residual after every OUTER (level 0) iteration, if requested
  std::vector<long>* coarsest_hist = nullptr;    // iterations of every coarsest solve, in call order (negative: not converged)
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
    if (l == 0 && mg.outer_hist) mg.outer_hist->push_back(std::sqrt(rsq) / bnorm);
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
    if (mg.coarsest_hist) mg.coarsest_hist->push_back(it >= 0 ? it : -(-it - 1));
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
static int kcycle_solve(int L, double mass, int n_refine, int coarse_dof, const double* gauge, const double* const* nullvecs, const double* b_,
                        double tol, int max_iter, int restart, double inner_tol, double coarsest_tol, int n_smooth, double* x_out_,
                        double* true_res, long* ops, long* its, std::vector<double>* outer_hist, std::vector<long>* coarsest_hist) {
  MG mg;
  mg.outer_hist = outer_hist;
  mg.coarsest_hist = coarsest_hist;
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

int qo_wilson_kcycle(int L, double mass, int n_refine, int coarse_dof, const double* gauge, const double* const* nullvecs, const double* b_,
                     double tol, int max_iter, int restart, double inner_tol, double coarsest_tol, int n_smooth, double* x_out_,
                     double* true_res, long* ops, long* its) {
  return kcycle_solve(L, mass, n_refine, coarse_dof, gauge, nullvecs, b_, tol, max_iter, restart, inner_tol, coarsest_tol, n_smooth, x_out_, true_res, ops, its,
                      nullptr, nullptr);
}

// Same solve, also recording the relative residual after each outer iteration (hist[0..nhist)) and the iteration count of
// each coarsest solve in call order (chist[0..nchist); negative = that solve hit its cap).  Returns as qo_wilson_kcycle;
// *nhist_out / *nchist_out receive the number of entries written.
int qo_wilson_kcycle_history(int L, double mass, int n_refine, int coarse_dof, const double* gauge, const double* const* nullvecs, const double* b_,
                             double tol, int max_iter, int restart, double inner_tol, double coarsest_tol, int n_smooth, double* x_out_,
                             double* true_res, long* ops, long* its, double* hist, int nhist, int* nhist_out, long* chist, int nchist, int* nchist_out) {
  std::vector<double> oh;
  std::vector<long> ch;
  const int it = kcycle_solve(L, mass, n_refine, coarse_dof, gauge, nullvecs, b_, tol, max_iter, restart, inner_tol, coarsest_tol, n_smooth, x_out_, true_res, ops,
                              its, &oh, &ch);
  int n = 0;
  for (; n < (int)oh.size() && n < nhist; n++) hist[n] = oh[n];
  if (nhist_out) *nhist_out = n;
  n = 0;
  for (; n < (int)ch.size() && n < nchist; n++) chist[n] = ch[n];
  if (nchist_out) *nchist_out = n;
  return it;
}

// ---------------------------------------------------------------------------------------------------------------------
// CPU twins of the remaining Krylov drivers the path calls (quantum-linalg, ABSENT: semantics from the call sites, SURVEY
// 2.2; PARITY UNPINNED against the reference, pinned against scipy's independent implementations of the same textbook
// algorithms in tests/test_oracle_krylov.py).  Operator: lhs = M rhs for a stencil descriptor; `op` selects
//   0: M   1: M^dagger M (needs the dagger stencil in `dag`)   -- CG wants a Hermitian positive definite operator.
// Conventions of every solver: x holds the initial guess on entry; stop when sqrt(resSq) < tol * |b|; *iters = iterations
// performed; returns 1 if converged, 0 if not.
//   kind 0: CG                (tests/n02_free_laplace_test/free_laplace.cpp:118; stateful_multigrid.h:915-969)
//   kind 1: BiCGStab-L        (tests/n13_wilson_kcycle/wilson_kcycle.cpp:359), L = param_i; iterations count BiCG steps
//   kind 2: Richardson        (tests/n22_wilson_kcycle_adaptive/wilson_kcycle.cpp:289,664), omega = param_d,
//                             residual checked every param_i iterations
//   kind 3: MR / MinRes       (stateful_multigrid.h:851-860,1037-1046), relaxation omega = param_d
//   kind 4: restarted GCR     (stateful_multigrid.h:915-969), restart = param_i (-1: none)
int qo_krylov_solve(int kind, const qo_stencil_desc* d, const qo_stencil_desc* dag, int op, double* x_, const double* b_, int max_iter, double tol,
                    int param_i, double param_d, int* iters, double* res_sq, double* hist, int nhist) {
  const long n = (long)d->Lx * d->Ly * d->nc;
  cvec x((const cplx*)x_, (const cplx*)x_ + n), b((const cplx*)b_, (const cplx*)b_ + n), tmp(n);
  auto A = [&](cvec& out, const cvec& in) {
    if (op == 0) qo_stencil_apply(d, (double*)out.data(), (const double*)in.data(), QO_P_ALL | QO_P_ZERO);
    else { qo_stencil_apply(d, (double*)tmp.data(), (const double*)in.data(), QO_P_ALL | QO_P_ZERO); qo_stencil_apply(dag, (double*)out.data(), (const double*)tmp.data(), QO_P_ALL | QO_P_ZERO); }
  };
  const double bnorm = std::sqrt(norm2(b, n));
  int k = 0;
  double rsq = 0.0;
  bool conv = false;
  auto note = [&](double r2) { if (hist && k - 1 < nhist && k >= 1) hist[k - 1] = std::sqrt(r2) / bnorm; };
  if (kind == 0) {   // CG
    cvec r(n), p(n), Ap(n);
    A(Ap, x);
    for (long i = 0; i < n; i++) r[i] = b[i] - Ap[i];
    p = r;
    rsq = norm2(r, n);
    conv = (bnorm == 0.0) || (std::sqrt(rsq) < tol * bnorm);
    while (!conv && k < max_iter) {
      A(Ap, p);
      const double pAp = cdot(p, Ap, n).real();
      if (pAp == 0.0) break;
      const double alpha = rsq / pAp;
      axpy(alpha, p, x, n);
      axpy(-alpha, Ap, r, n);
      const double rn = norm2(r, n);
      k++;
      note(rn);
      if (std::sqrt(rn) < tol * bnorm) { rsq = rn; conv = true; break; }
      const double beta = rn / rsq;
      rsq = rn;
      for (long i = 0; i < n; i++) p[i] = r[i] + beta * p[i];
    }
  } else if (kind == 1) {   // BiCGStab(L), Sleijpen & Fokkema 1993, Algorithm 3.1
    const int L = param_i;
    std::vector<cvec> r(L + 1, cvec(n)), u(L + 1, cvec(n, cplx(0.0)));
    A(u[0], x);
    for (long i = 0; i < n; i++) r[0][i] = b[i] - u[0][i];
    cvec rt = r[0];
    std::fill(u[0].begin(), u[0].end(), cplx(0.0));
    cplx rho0 = 1.0, alpha = 0.0, omega = 1.0;
    rsq = norm2(r[0], n);
    conv = (bnorm == 0.0) || (std::sqrt(rsq) < tol * bnorm);
    std::vector<cplx> tau((L + 1) * (L + 1)), gam(L + 1), gamp(L + 1), gampp(L + 1);
    std::vector<double> sigma(L + 1);
    bool broke = false;
    while (!conv && k < max_iter && !broke) {
      rho0 = -omega * rho0;
      for (int j = 0; j < L && !broke; j++) {
        const cplx rho1 = cdot(rt, r[j], n);
        if (rho0 == 0.0) { broke = true; break; }
        const cplx beta = alpha * rho1 / rho0;
        rho0 = rho1;
        for (int i = 0; i <= j; i++) for (long q = 0; q < n; q++) u[i][q] = r[i][q] - beta * u[i][q];
        A(u[j + 1], u[j]);
        const cplx g = cdot(rt, u[j + 1], n);
        if (g == 0.0) { broke = true; break; }
        alpha = rho0 / g;
        for (int i = 0; i <= j; i++) axpy(-alpha, u[i + 1], r[i], n);
        A(r[j + 1], r[j]);
        axpy(alpha, u[0], x, n);
        k++;
      }
      if (broke) break;
      for (int j = 1; j <= L && !broke; j++) {
        for (int i = 1; i < j; i++) {
          tau[i * (L + 1) + j] = cdot(r[i], r[j], n) / sigma[i];
          axpy(-tau[i * (L + 1) + j], r[i], r[j], n);
        }
        sigma[j] = norm2(r[j], n);
        if (sigma[j] == 0.0) { broke = true; break; }
        gamp[j] = cdot(r[j], r[0], n) / sigma[j];
      }
      if (broke) break;
      gam[L] = gamp[L];
      omega = gam[L];
      for (int j = L - 1; j >= 1; j--) {
        gam[j] = gamp[j];
        for (int i = j + 1; i <= L; i++) gam[j] -= tau[j * (L + 1) + i] * gam[i];
      }
      for (int j = 1; j < L; j++) {
        gampp[j] = gam[j + 1];
        for (int i = j + 1; i < L; i++) gampp[j] += tau[j * (L + 1) + i] * gam[i + 1];
      }
      axpy(gam[1], r[0], x, n);
      axpy(-gamp[L], r[L], r[0], n);
      axpy(-gam[L], u[L], u[0], n);
      for (int j = 1; j < L; j++) {
        axpy(-gam[j], u[j], u[0], n);
        axpy(gampp[j], r[j], x, n);
        axpy(-gamp[j], r[j], r[0], n);
      }
      rsq = norm2(r[0], n);
      if (hist) for (int q = std::max(0, k - L); q < k && q < nhist; q++) hist[q] = std::sqrt(rsq) / bnorm;
      if (std::sqrt(rsq) < tol * bnorm) conv = true;
    }
  } else if (kind == 2) {   // Richardson: x += omega (b - A x); residual looked at every param_i iterations
    cvec r(n), Ax(n);
    const double omega = param_d;
    const int check = param_i;
    while (k < max_iter) {
      A(Ax, x);
      for (long i = 0; i < n; i++) r[i] = b[i] - Ax[i];
      if (check > 0 && (k % check) == 0) {
        rsq = norm2(r, n);
        if (bnorm == 0.0 || std::sqrt(rsq) < tol * bnorm) { conv = true; break; }
      }
      axpy(omega, r, x, n);
      k++;
    }
    if (!conv) {
      A(Ax, x);
      rsq = qo_diffnorm2sq((const double*)b.data(), (const double*)Ax.data(), n);
      conv = (bnorm == 0.0) || (std::sqrt(rsq) < tol * bnorm);
    }
  } else if (kind == 3) {   // MR(omega)
    cvec r(n), p(n);
    const double omega = param_d;
    A(p, x);
    for (long i = 0; i < n; i++) r[i] = b[i] - p[i];
    rsq = norm2(r, n);
    conv = (bnorm == 0.0) || (std::sqrt(rsq) < tol * bnorm);
    while (!conv && k < max_iter) {
      A(p, r);
      const cplx pr = cdot(p, r, n);
      const double pp = norm2(p, n);
      if (pp == 0.0) break;
      const cplx alpha = omega * pr / pp;
      axpy(alpha, r, x, n);
      axpy(-alpha, p, r, n);
      rsq = norm2(r, n);
      k++;
      note(rsq);
      if (std::sqrt(rsq) < tol * bnorm) conv = true;
    }
  } else if (kind == 4) {   // restarted GCR, unpreconditioned
    const int restart = param_i;
    const int basis_max = (restart > 0) ? restart : max_iter;
    cvec r(n), t(n);
    std::vector<cvec> Z, W;
    std::vector<double> Wn;
    A(t, x);
    for (long i = 0; i < n; i++) r[i] = b[i] - t[i];
    rsq = norm2(r, n);
    conv = (bnorm == 0.0) || (std::sqrt(rsq) < tol * bnorm);
    int kb = 0;
    while (!conv && k < max_iter) {
      if (kb == (int)Z.size()) { Z.push_back(cvec(n)); W.push_back(cvec(n)); Wn.push_back(0.0); }
      Z[kb] = r;
      A(W[kb], Z[kb]);
      for (int i = 0; i < kb; i++) {
        const cplx beta = cdot(W[i], W[kb], n) / Wn[i];
        axpy(-beta, W[i], W[kb], n);
        axpy(-beta, Z[i], Z[kb], n);
      }
      const double ww = norm2(W[kb], n);
      if (ww == 0.0) break;
      Wn[kb] = ww;
      const cplx alpha = cdot(W[kb], r, n) / ww;
      axpy(alpha, Z[kb], x, n);
      axpy(-alpha, W[kb], r, n);
      rsq = norm2(r, n);
      k++; kb++;
      note(rsq);
      if (std::sqrt(rsq) < tol * bnorm) { conv = true; break; }
      if (kb == basis_max) {
        A(t, x);
        for (long i = 0; i < n; i++) r[i] = b[i] - t[i];
        rsq = norm2(r, n);
        kb = 0;
        if (std::sqrt(rsq) < tol * bnorm) { conv = true; break; }
      }
    }
  } else return -1;
  std::memcpy(x_, x.data(), sizeof(cplx) * n);
  if (iters) *iters = k;
  if (res_sq) *res_sq = rsq;
  return conv ? 1 : 0;
}

}  // extern "C"
