// krylov.hpp -- the Krylov drivers the K-cycle calls, on device-resident vectors.
//
// The reference takes these from quantum-linalg (`inverters/generic_*.h`, absent; only the call sites
// are known: stateful_multigrid.h:851-990,1037-1046, tests/n13_wilson_kcycle/wilson_kcycle.cpp:459-466,
// tests/n02_free_laplace_test/free_laplace.cpp:118).  Names, argument order and the inversion_info /
// inversion_verbose_struct conventions follow those call sites; the algorithms are the textbook ones
// (PARITY UNPINNED: there is no stored output of the reference's solvers anywhere).
//   * tolerance is relative:  stop when sqrt(resSq) < tol * ||b||
//   * resSq is the (recursive) residual norm squared at exit; iter counts iterations; ops_count counts
//     operator applications (used by DslashTrackerMG, stateful_multigrid.h:854-865)
// All vectors are device pointers; every reduction is a two-stage device reduction whose result comes back
// to the host because the control flow depends on it.  GCR-type orthogonalisation uses one fused
// multi-dot pass (qmg_multidot) instead of k separate dot kernels.
#ifndef QMG_KRYLOV_HPP
#define QMG_KRYLOV_HPP

#include <cmath>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "qmg_device.hpp"

namespace qmg {

// Scratch vectors for one solve.  Returned to a free list on scope exit instead of hipFree (which synchronises the device): the smoothers run
// thousands of times per solve.  The free list is kept BY CAPACITY: a request for n elements takes the smallest cached block that holds
// n and is not more than twice as large (so the complex<double> scratch of one solve serves the complex<float> batch of the next, and a
// Schur system's half-length vectors fit a full-length block) -- a K-cycle solve that follows another one in the same process then runs
// without a single hipMalloc in its timed region (GB-sized hipMalloc / hipFree calls cost milliseconds each and synchronise).
struct VecPool {
  std::vector<complex<double>*> v;
  size_t n;
  // per host thread (a thread = one stream = one rank when ranks are emulated by threads: qmg_comm_emulate_*)
  struct Shared {
    std::map<size_t, std::vector<complex<double>*>> free_by_cap;   // capacity (elements) -> cached blocks
    std::map<complex<double>*, size_t> cap;                         // every block this pool system has allocated and not yet freed
    size_t cached_elems;
    Shared() : cached_elems(0) {}
  };
  static Shared& shared() { static thread_local Shared s; return s; }
  explicit VecPool(size_t n_) : n(n_) {}
  complex<double>* get() {
    Shared& sh = shared();
    complex<double>* p = 0;
    for (auto it = sh.free_by_cap.lower_bound(n); it != sh.free_by_cap.end() && it->first <= 2 * n; ++it)
      if (!it->second.empty()) { p = it->second.back(); it->second.pop_back(); sh.cached_elems -= it->first; break; }
    if (!p) {
      // nothing cached serves the request.  If the HBM that is left cannot either, the cached blocks of OTHER capacities are given back first (a
      // batch of 3 systems after single-system solves found 36 GB of single-system scratch on the free list that no request of its own could
      // take, and ran out of memory at basis vector 56 of 128: batch_systems_that_fit counts cached scratch as free, so it has to be)
      size_t free_b = 0, total_b = 0;
      const size_t need = n * sizeof(complex<double>);
      if (sh.cached_elems > 0 && qmg_mem_info(&free_b, &total_b) == QMG_SUCCESS && free_b < need + ((size_t)1 << 30)) release_all();
      p = allocate_vector<complex<double>>(n);
      if (p) sh.cap[p] = n;
    }
    v.push_back(p);
    return p;
  }
  ~VecPool() {
    Shared& sh = shared();
    for (auto p : v) {
      if (!p) continue;
      const size_t c = sh.cap[p];
      sh.free_by_cap[c].push_back(p);
      sh.cached_elems += c;
    }
  }
  static size_t cached_bytes() { return shared().cached_elems * sizeof(complex<double>); }
  // make sure the free list holds `count` blocks that serve a request for n elements: a solver's scratch allocated BEFORE its timed region (the device allocator's
  // cost for GB-sized blocks is erratic on this platform: 134 calls took 3 ms in one run and 0.66 s in the next, drivers' `[QMG-TIMING]` lines)
  static bool reserve(size_t n, int count) {
    Shared& sh = shared();
    int have = 0;   // blocks on the free list that a request for n elements would take
    for (auto it = sh.free_by_cap.lower_bound(n); it != sh.free_by_cap.end() && it->first <= 2 * n; ++it) have += (int)it->second.size();
    for (int i = have; i < count; i++) {
      // best effort: leave a quarter of the HBM alone (what the solve cannot find here it allocates on demand, and says so)
      size_t free_b = 0, total_b = 0;
      if (qmg_mem_info(&free_b, &total_b) != QMG_SUCCESS || free_b < total_b / 4 + n * sizeof(complex<double>)) return false;
      void* raw = nullptr;
      const double t0 = wall_now();
      const int rc = qmg_malloc(&raw, n * sizeof(complex<double>));
      alloc_stats().seconds += wall_now() - t0; alloc_stats().mallocs++;
      if (rc != QMG_SUCCESS || !raw) return false;
      complex<double>* p = static_cast<complex<double>*>(raw);
      sh.cap[p] = n;
      sh.free_by_cap[n].push_back(p);
      sh.cached_elems += n;
    }
    return true;
  }
  static void release_all() {
    Shared& sh = shared();
    for (auto& kv : sh.free_by_cap)
      for (auto& p : kv.second) { sh.cap.erase(p); deallocate_vector(&p); }
    sh.free_by_cap.clear();
    sh.cached_elems = 0;
  }
};

inline void report(inversion_verbose_struct* verb, const char* name, int iter, double rel, bool summary_only = false) {
  if (!verb) return;
  if (verb->verbosity == VERB_DETAIL && !summary_only)
    std::cout << verb->verb_prefix << name << " Iter " << iter << " RelTol " << rel << "\n";
}
inline void summary(inversion_verbose_struct* verb, const char* name, bool ok_, int iter, double rel) {
  if (!verb || verb->verbosity == VERB_NONE) return;
  std::cout << verb->verb_prefix << name << (ok_ ? " Success " : " Fail ") << "Iter " << iter << " RelTol " << rel << "\n";
}

inline std::vector<complex<double>> multidot(const std::vector<complex<double>*>& xs, int k, const complex<double>* y, size_t n) {
  std::vector<complex<double>> out(k);
  int done = 0;
  while (done < k) {   // the ABI takes up to 64 vectors per call
    const int kk = (k - done > 64) ? 64 : k - done;
    std::vector<const void*> ptrs(kk);
    for (int i = 0; i < kk; i++) ptrs[i] = xs[done + i];
    std::vector<double> r(2 * kk);
    ok(qmg_multidot(ptrs.data(), kk, y, n, nullptr, r.data(), current_stream()), "qmg_multidot");
    for (int i = 0; i < kk; i++) out[done + i] = complex<double>(r[2 * i], r[2 * i + 1]);
    done += kk;
  }
  return out;
}

// y += sum_i a_i x_i, i < k, in one pass (qmg_multi_caxpy)
inline void multi_caxpy(const std::vector<complex<double>>& a, const std::vector<complex<double>*>& xs, int k, complex<double>* y, size_t n) {
  if (k <= 0) return;
  std::vector<double> cf(2 * k);
  std::vector<const void*> ptrs(k);
  for (int i = 0; i < k; i++) { cf[2 * i] = a[i].real(); cf[2 * i + 1] = a[i].imag(); ptrs[i] = xs[i]; }
  ok(qmg_multi_caxpy(cf.data(), ptrs.data(), k, y, n, current_stream()), "qmg_multi_caxpy");
}

// GCR with raw search directions: y_j such that sum_k alpha_k z'_k = sum_j y_j z_j, z'_k = z_k + sum_{i<k} c[k][i] z'_i
inline std::vector<complex<double>> gcr_direction_weights(const std::vector<complex<double>>& alpha, const std::vector<std::vector<complex<double>>>& c, int K) {
  std::vector<complex<double>> beta(alpha.begin(), alpha.begin() + K);
  for (int k = K - 1; k >= 0; k--)
    for (int i = 0; i < k; i++) beta[i] += beta[k] * c[k][i];
  return beta;
}

}  // namespace qmg

// ---------------------------------------------------------------------------------------------
// MinRes / MR with relaxation omega (minv_vector_minres(x, b, n, iters, tol, omega, op, opdata)).
//   r = b - A x ; repeat: p = A r ; alpha = <p,r>/<p,p> ; x += omega alpha r ; r -= omega alpha p
// ---------------------------------------------------------------------------------------------
// Zero-initial-guess hint.  Every K-cycle smoother and coarse solve starts from a vector the caller has just zeroed
// (stateful_multigrid.h:848, :958, :1010), so r0 = b exactly and the reference's opening A*x0 is an apply of zeros.  A
// caller that KNOWS x0 == 0 constructs a qmg::ZeroGuess right before the solver call; the solver consumes the hint at
// entry (nested preconditioner solves never see it), skips that apply and its two vector passes, and does not count it
// in ops_count.  Results are bit-identical to the unhinted path.
namespace qmg {
inline bool& zero_guess_flag() { static thread_local bool f = false; return f; }
inline bool take_zero_guess() { bool f = zero_guess_flag(); zero_guess_flag() = false; return f; }
struct ZeroGuess { ZeroGuess() { zero_guess_flag() = true; } ~ZeroGuess() { zero_guess_flag() = false; } };
}  // namespace qmg

inline inversion_info minv_vector_minres(complex<double>* phi, complex<double>* phi0, int size, int max_iter, double eps, double omega,
                                         matrix_op_cplx matrix_vector, void* extra_info, inversion_verbose_struct* verb = 0) {
  inversion_info invif;
  invif.name = "MinRes (relaxation parameter " + std::to_string(omega) + ")";
  qmg::VecPool pool(size);
  complex<double>* r = pool.get();
  complex<double>* p = pool.get();
  const double bsq = norm2sq(phi0, size);
  const double bnorm = std::sqrt(bsq);
  int ops = 0;
  double rsq;
  if (qmg::take_zero_guess()) { copy_vector(r, phi0, size); rsq = bsq; }   // x0 == 0: r = b
  else {
    matrix_vector(p, phi, extra_info); ops++;
    caxpbyz(1.0, phi0, -1.0, p, r, size);
    rsq = norm2sq(r, size);
  }
  double rsq_ref = rsq;
  int k = 0;
  bool conv = (bnorm == 0.0) || (std::sqrt(rsq) < eps * bnorm);
  while (!conv && k < max_iter) {
    matrix_vector(p, r, extra_info); ops++;
    // <r,p> and <p,p> in one pass over p; the new residual norm follows analytically:
    // |r - a p|^2 = |r|^2 - (2 omega - omega^2) |<p,r>|^2 / <p,p>   for a = omega <p,r>/<p,p>
    std::vector<complex<double>*> rp = {r, p};
    std::vector<complex<double>> d2 = qmg::multidot(rp, 2, p, size);
    const complex<double> pr = std::conj(d2[0]);
    const double pp = d2[1].real();
    if (pp == 0.0) break;
    const complex<double> alpha = omega * pr / pp;
    caxpy(alpha, r, phi, size);
    // (the subtraction loses absolute accuracy ~1e-16 * rsq_ref: re-anchor with a true norm after every 8 orders of magnitude)
    rsq = rsq - (2.0 * omega - omega * omega) * std::norm(pr) / pp;
    const bool renorm = !(rsq > 1e-8 * rsq_ref) || std::sqrt(rsq) < 4.0 * eps * bnorm;
    // r is only needed by a further iteration or by the re-anchoring: the residual update of the LAST iteration is skipped
    // (callers that want the residual recompute b - A x, as the K-cycle does); x and the returned |r|^2 are unaffected
    if (renorm || k + 1 < max_iter) caxpy(-alpha, p, r, size);
    if (renorm) { rsq = norm2sq(r, size); rsq_ref = rsq; }
    k++;
    qmg::report(verb, "MinRes", k, std::sqrt(rsq) / bnorm);
    if (std::sqrt(rsq) < eps * bnorm) conv = true;
  }
  invif.success = conv;
  invif.iter = k;
  invif.resSq = rsq;
  invif.ops_count = ops;
  qmg::summary(verb, "MinRes", conv, k, bnorm > 0 ? std::sqrt(rsq) / bnorm : 0.0);
  return invif;
}

// ---------------------------------------------------------------------------------------------
// CG (Hermitian positive definite op): minv_vector_cg(x, b, n, max_iter, tol, op, opdata, verb)
// ---------------------------------------------------------------------------------------------
inline inversion_info minv_vector_cg(complex<double>* phi, complex<double>* phi0, int size, int max_iter, double eps, matrix_op_cplx matrix_vector,
                                     void* extra_info, inversion_verbose_struct* verb = 0) {
  inversion_info invif;
  invif.name = "CG";
  qmg::VecPool pool(size);
  complex<double>*r = pool.get(), *p = pool.get(), *Ap = pool.get();
  const double bnorm = std::sqrt(norm2sq(phi0, size));
  int ops = 0;
  if (qmg::take_zero_guess()) copy_vector(r, phi0, size);
  else { matrix_vector(Ap, phi, extra_info); ops++; caxpbyz(1.0, phi0, -1.0, Ap, r, size); }
  copy_vector(p, r, size);
  double rsq = norm2sq(r, size);
  int k = 0;
  bool conv = (bnorm == 0.0) || (std::sqrt(rsq) < eps * bnorm);
  while (!conv && k < max_iter) {
    matrix_vector(Ap, p, extra_info); ops++;
    const double pAp = dot(p, Ap, size).real();
    if (pAp == 0.0) break;
    const double alpha = rsq / pAp;
    caxpy(alpha, p, phi, size);
    caxpy(-alpha, Ap, r, size);
    const double rsq_new = norm2sq(r, size);
    k++;
    qmg::report(verb, "CG", k, std::sqrt(rsq_new) / bnorm);
    if (std::sqrt(rsq_new) < eps * bnorm) { rsq = rsq_new; conv = true; break; }
    const double beta = rsq_new / rsq;
    rsq = rsq_new;
    cxpay(r, beta, p, size);   // p = r + beta p
  }
  invif.success = conv; invif.iter = k; invif.resSq = rsq; invif.ops_count = ops;
  qmg::summary(verb, "CG", conv, k, bnorm > 0 ? std::sqrt(rsq) / bnorm : 0.0);
  return invif;
}

inline inversion_info minv_vector_cg_restart(complex<double>* phi, complex<double>* phi0, int size, int max_iter, double eps, int restart_freq,
                                             matrix_op_cplx matrix_vector, void* extra_info, inversion_verbose_struct* verb = 0) {
  inversion_info total;
  total.name = "Restarted CG(" + std::to_string(restart_freq) + ")";
  const double bnorm = std::sqrt(norm2sq(phi0, size));
  while (total.iter < max_iter) {
    const int chunk = (max_iter - total.iter < restart_freq) ? max_iter - total.iter : restart_freq;
    inversion_info one = minv_vector_cg(phi, phi0, size, chunk, eps, matrix_vector, extra_info, 0);
    total.iter += one.iter; total.ops_count += one.ops_count; total.resSq = one.resSq; total.success = one.success;
    if (one.success || one.iter == 0) break;
  }
  qmg::summary(verb, "CG-restart", total.success, total.iter, bnorm > 0 ? std::sqrt(total.resSq) / bnorm : 0.0);
  return total;
}

// ---------------------------------------------------------------------------------------------
// Flexible (variable-preconditioned) GCR with optional restarts -- the K-cycle's Krylov wrapper.
//   r = b - A x
//   loop:  z_k = M^-1 r (preconditioner; identity when precond == 0)
//          w_k = A z_k ; orthogonalise w_k against w_0..w_{k-1} (and carry z_k along)
//          alpha = <w_k, r>/<w_k,w_k> ; x += alpha z_k ; r -= alpha w_k
//   restart_freq > 0: the basis is dropped every restart_freq directions.
// The z_k are NOT orthogonalised explicitly.  With c_ik the Gram-Schmidt coefficients of w_k, the conjugate
// directions are z'_k = z_k + sum_{i<k} c_ik z'_i and x = x0 + sum_k alpha_k z'_k = x0 + sum_j y_j z_j, where y
// follows from alpha and c by a k x k back-substitution on the host (qmg::gcr_direction_weights).  x is only needed
// at a restart and at exit, so the k-vector pass "z_k -= sum c_ik Z_i" of every iteration (a third of GCR's BLAS-1
// traffic) becomes ONE multi-axpy per restart cycle.  w_k, r and every scalar -- hence every convergence decision --
// are computed exactly as before; x differs by rounding only.
// ---------------------------------------------------------------------------------------------
inline inversion_info qmg_gcr_core(complex<double>* phi, complex<double>* phi0, int size, int max_iter, double eps, int restart_freq,
                                   matrix_op_cplx matrix_vector, void* extra_info, precond_op_cplx precond, void* precond_info,
                                   inversion_verbose_struct* verb, const char* name) {
  inversion_info invif;
  invif.name = name;
  const int basis_max = (restart_freq > 0) ? restart_freq : max_iter;
  qmg::VecPool pool(size);
  complex<double>* r = pool.get();
  complex<double>* tmp = pool.get();
  std::vector<complex<double>*> Z, W;     // raw search directions z_k and orthogonalised images w'_k, allocated on demand
  std::vector<double> Wnorm2;
  std::vector<std::vector<complex<double>>> C;   // C[k][i] = Gram-Schmidt coefficient of w_k against w'_i (i < k), this cycle
  std::vector<complex<double>> alphas;           // alpha_k of this cycle
  auto flush_x = [&](int K) {                    // x += sum_k alpha_k z'_k for the K directions of this cycle
    if (K <= 0) return;
    qmg::multi_caxpy(qmg::gcr_direction_weights(alphas, C, K), Z, K, phi, size);
  };
  const double bsq = norm2sq(phi0, size);
  const double bnorm = std::sqrt(bsq);
  int ops = 0;
  double rsq;
  if (qmg::take_zero_guess()) { copy_vector(r, phi0, size); rsq = bsq; }   // x0 == 0: r = b
  else {
    matrix_vector(tmp, phi, extra_info); ops++;
    caxpbyz(1.0, phi0, -1.0, tmp, r, size);
    rsq = norm2sq(r, size);
  }
  double rsq_ref = rsq;
  bool conv = (bnorm == 0.0) || (std::sqrt(rsq) < eps * bnorm);
  int k = 0, kb = 0;   // total iterations, index within the current basis
  inversion_verbose_struct pverb(verb ? verb->precond_verbosity : VERB_NONE, verb ? verb->precond_verb_prefix : std::string(""));
  if (verb) { pverb.precond_verbosity = verb->precond_verbosity; pverb.precond_verb_prefix = verb->precond_verb_prefix; }
  while (!conv && k < max_iter) {
    if (kb == (int)Z.size()) { Z.push_back(pool.get()); W.push_back(pool.get()); Wnorm2.push_back(0.0); C.push_back(std::vector<complex<double>>()); alphas.push_back(0.0); }
    complex<double>* z = Z[kb];
    complex<double>* w = W[kb];
    if (precond) { zero_vector(z, size); precond(z, r, size, precond_info, &pverb); }
    else copy_vector(z, r, size);
    matrix_vector(w, z, extra_info); ops++;
    C[kb].clear();
    if (kb > 0) {   // Gram-Schmidt of w against the current basis: ONE multi-dot pass, ONE fused multi-axpy pass
      std::vector<complex<double>> c = qmg::multidot(W, kb, w, size);
      for (int i = 0; i < kb; i++) c[i] = -c[i] / Wnorm2[i];
      qmg::multi_caxpy(c, W, kb, w, size);
      C[kb] = c;
    }
    // <r,w> and <w,w> in one pass over w
    std::vector<complex<double>*> rw = {r, w};
    std::vector<complex<double>> d2 = qmg::multidot(rw, 2, w, size);
    const double ww = d2[1].real();
    if (ww == 0.0) break;
    Wnorm2[kb] = ww;
    const complex<double> wr = std::conj(d2[0]);   // <w,r>
    const complex<double> alpha = wr / ww;
    alphas[kb] = alpha;
    caxpy(-alpha, w, r, size);
    // |r - alpha w|^2 = |r|^2 - |<w,r>|^2 / <w,w>: no extra reduction; confirmed by a true norm near convergence
    // (the subtraction loses absolute accuracy ~1e-16 * rsq_ref: re-anchor with a true norm after every 8 orders of magnitude)
    rsq = rsq - std::norm(wr) / ww;
    if (!(rsq > 1e-8 * rsq_ref) || std::sqrt(rsq) < 4.0 * eps * bnorm) { rsq = norm2sq(r, size); rsq_ref = rsq; }
    k++; kb++;
    qmg::report(verb, name, k, std::sqrt(rsq) / bnorm);
    if (std::sqrt(rsq) < eps * bnorm) { conv = true; break; }
    if (kb == basis_max) {   // restart: bring x up to date, recompute the true residual, drop the basis
      flush_x(kb);
      matrix_vector(tmp, phi, extra_info); ops++;
      caxpbyz(1.0, phi0, -1.0, tmp, r, size);
      rsq = norm2sq(r, size);
      rsq_ref = rsq;
      kb = 0;
      if (verb && verb->verbosity >= VERB_RESTART_DETAIL) std::cout << verb->verb_prefix << name << " restart at iter " << k << " RelTol " << std::sqrt(rsq) / bnorm << "\n";
      if (std::sqrt(rsq) < eps * bnorm) { conv = true; break; }
    }
  }
  flush_x(kb);   // (kb == 0 right after a restart: nothing pending)
  invif.success = conv; invif.iter = k; invif.resSq = rsq; invif.ops_count = ops;
  qmg::summary(verb, name, conv, k, bnorm > 0 ? std::sqrt(rsq) / bnorm : 0.0);
  return invif;
}

inline inversion_info minv_vector_gcr(complex<double>* phi, complex<double>* phi0, int size, int max_iter, double eps, matrix_op_cplx op, void* opd,
                                      inversion_verbose_struct* verb = 0) {
  return qmg_gcr_core(phi, phi0, size, max_iter, eps, -1, op, opd, 0, 0, verb, "GCR");
}
inline inversion_info minv_vector_gcr_restart(complex<double>* phi, complex<double>* phi0, int size, int max_iter, double eps, int restart_freq,
                                              matrix_op_cplx op, void* opd, inversion_verbose_struct* verb = 0) {
  return qmg_gcr_core(phi, phi0, size, max_iter, eps, restart_freq, op, opd, 0, 0, verb, "GCR-restart");
}
inline inversion_info minv_vector_gcr_var_precond(complex<double>* phi, complex<double>* phi0, int size, int max_iter, double eps, matrix_op_cplx op,
                                                  void* opd, precond_op_cplx precond, void* precd, inversion_verbose_struct* verb = 0) {
  return qmg_gcr_core(phi, phi0, size, max_iter, eps, -1, op, opd, precond, precd, verb, "VPGCR");
}
inline inversion_info minv_vector_gcr_var_precond_restart(complex<double>* phi, complex<double>* phi0, int size, int max_iter, double eps,
                                                          int restart_freq, matrix_op_cplx op, void* opd, precond_op_cplx precond, void* precd,
                                                          inversion_verbose_struct* verb = 0) {
  return qmg_gcr_core(phi, phi0, size, max_iter, eps, restart_freq, op, opd, precond, precd, verb, "VPGCR-restart");
}

// ---------------------------------------------------------------------------------------------
// BiCGStab(L) (Sleijpen & Fokkema 1993): minv_vector_bicgstab_l(x, b, n, max_iter, tol, L, op, opdata, verb)
// -- the null-vector relaxation of tests/n13_wilson_kcycle/wilson_kcycle.cpp:359.  `iter` counts BiCG steps.
// ---------------------------------------------------------------------------------------------
inline inversion_info minv_vector_bicgstab_l(complex<double>* phi, complex<double>* phi0, int size, int max_iter, double eps, int L,
                                             matrix_op_cplx matrix_vector, void* extra_info, inversion_verbose_struct* verb = 0) {
  inversion_info invif;
  invif.name = "BiCGStab-" + std::to_string(L);
  qmg::VecPool pool(size);
  std::vector<complex<double>*> r(L + 1), u(L + 1);
  for (int i = 0; i <= L; i++) { r[i] = pool.get(); u[i] = pool.get(); }
  complex<double>* rt = pool.get();
  const double bnorm = std::sqrt(norm2sq(phi0, size));
  int ops = 0;
  if (qmg::take_zero_guess()) copy_vector(r[0], phi0, size);
  else { matrix_vector(u[0], phi, extra_info); ops++; caxpbyz(1.0, phi0, -1.0, u[0], r[0], size); }
  copy_vector(rt, r[0], size);
  zero_vector(u[0], size);
  complex<double> rho0 = 1.0, alpha = 0.0, omega = 1.0;
  double rsq = norm2sq(r[0], size);
  bool conv = (bnorm == 0.0) || (std::sqrt(rsq) < eps * bnorm);
  int k = 0;
  std::vector<complex<double>> tau((L + 1) * (L + 1)), gamma(L + 1), gammap(L + 1), gammapp(L + 1);
  std::vector<double> sigma(L + 1);
  bool breakdown = false;
  while (!conv && k < max_iter && !breakdown) {
    rho0 = -omega * rho0;
    for (int j = 0; j < L && !breakdown; j++) {   // BiCG part
      const complex<double> rho1 = dot(rt, r[j], size);
      if (rho0 == 0.0) { breakdown = true; break; }
      const complex<double> beta = alpha * rho1 / rho0;
      rho0 = rho1;
      for (int i = 0; i <= j; i++) cxpay(r[i], -beta, u[i], size);   // u_i = r_i - beta u_i
      matrix_vector(u[j + 1], u[j], extra_info); ops++;
      const complex<double> gam = dot(rt, u[j + 1], size);
      if (gam == 0.0) { breakdown = true; break; }
      alpha = rho0 / gam;
      for (int i = 0; i <= j; i++) caxpy(-alpha, u[i + 1], r[i], size);
      matrix_vector(r[j + 1], r[j], extra_info); ops++;
      caxpy(alpha, u[0], phi, size);
      k++;
    }
    if (breakdown) break;
    for (int j = 1; j <= L; j++) {   // MR part: modified Gram-Schmidt on r_1..r_L
      for (int i = 1; i < j; i++) {
        tau[i * (L + 1) + j] = dot(r[i], r[j], size) / sigma[i];
        caxpy(-tau[i * (L + 1) + j], r[i], r[j], size);
      }
      sigma[j] = norm2sq(r[j], size);
      if (sigma[j] == 0.0) { breakdown = true; break; }
      gammap[j] = dot(r[j], r[0], size) / sigma[j];
    }
    if (breakdown) break;
    gamma[L] = gammap[L];
    omega = gamma[L];
    for (int j = L - 1; j >= 1; j--) {
      gamma[j] = gammap[j];
      for (int i = j + 1; i <= L; i++) gamma[j] -= tau[j * (L + 1) + i] * gamma[i];
    }
    for (int j = 1; j < L; j++) {
      gammapp[j] = gamma[j + 1];
      for (int i = j + 1; i < L; i++) gammapp[j] += tau[j * (L + 1) + i] * gamma[i + 1];
    }
    caxpy(gamma[1], r[0], phi, size);
    caxpy(-gammap[L], r[L], r[0], size);
    caxpy(-gamma[L], u[L], u[0], size);
    for (int j = 1; j < L; j++) {
      caxpy(-gamma[j], u[j], u[0], size);
      caxpy(gammapp[j], r[j], phi, size);
      caxpy(-gammap[j], r[j], r[0], size);
    }
    rsq = norm2sq(r[0], size);
    qmg::report(verb, "BiCGStab-L", k, std::sqrt(rsq) / bnorm);
    if (std::sqrt(rsq) < eps * bnorm) conv = true;
  }
  invif.success = conv; invif.iter = k; invif.resSq = rsq; invif.ops_count = ops;
  qmg::summary(verb, "BiCGStab-L", conv, k, bnorm > 0 ? std::sqrt(rsq) / bnorm : 0.0);
  return invif;
}

// ---------------------------------------------------------------------------------------------
// Richardson relaxation (null-vector generation, tests/n22...:289):
//   minv_vector_richardson(x, b, n, max_iter, tol, omega, check_freq, op, opdata)
//   x += omega (b - A x); the residual norm is only evaluated every check_freq iterations.
// ---------------------------------------------------------------------------------------------
inline inversion_info minv_vector_richardson(complex<double>* phi, complex<double>* phi0, int size, int max_iter, double eps, double omega,
                                             int check_freq, matrix_op_cplx matrix_vector, void* extra_info, inversion_verbose_struct* verb = 0) {
  inversion_info invif;
  invif.name = "Richardson";
  qmg::VecPool pool(size);
  complex<double>*r = pool.get(), *Ax = pool.get();
  const double bnorm = std::sqrt(norm2sq(phi0, size));
  int ops = 0, k = 0;
  double rsq = 0.0;
  bool conv = false;
  qmg::take_zero_guess();   // no saving taken here: the first sweep's A*x0 is folded into the loop
  while (k < max_iter) {
    matrix_vector(Ax, phi, extra_info); ops++;
    caxpbyz(1.0, phi0, -1.0, Ax, r, size);
    if (check_freq > 0 && (k % check_freq) == 0) {
      rsq = norm2sq(r, size);
      if (bnorm == 0.0 || std::sqrt(rsq) < eps * bnorm) { conv = true; break; }
    }
    caxpy(omega, r, phi, size);
    k++;
  }
  if (!conv) {
    matrix_vector(Ax, phi, extra_info); ops++;
    rsq = diffnorm2sq(phi0, Ax, size);
    conv = (bnorm == 0.0) || (std::sqrt(rsq) < eps * bnorm);
  }
  invif.success = conv; invif.iter = k; invif.resSq = rsq; invif.ops_count = ops;
  qmg::summary(verb, "Richardson", conv, k, bnorm > 0 ? std::sqrt(rsq) / bnorm : 0.0);
  return invif;
}

#endif
