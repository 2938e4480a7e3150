// n13_wilson_kcycle_mrhs -- the n13 K-cycle solve for a LOCK-STEP BATCH of independent right-hand sides on one GPU
// (BASELINE configs[3]/[4]: independent right-hand sides, several per GPU; include/qmg/batch.hpp).
//   ./n13_wilson_kcycle_mrhs L mass beta n_refine coarse_dof gauge_file tile nrhs [verify|verify0]
// Setup is n13's (n13_setup.hpp).  Right-hand side k is the gaussian vector of seed (first solve seed)+k, so system 0
// is exactly the system n13_wilson_kcycle solves.  All nrhs <= 16 systems advance through the same VPGCR / K-cycle
// iteration together: every coarse operator and null vector is streamed once per step for the whole batch and the
// coarse applies run on the f64 matrix cores.
// With `verify`, every system is then solved again ALONE by the single-vector path (krylov.hpp / multigrid.hpp) and the
// two solutions, iteration counts and wall times are compared.
#include "n13_setup.hpp"

int main(int argc, char** argv) {
  if (argc < 9) { std::cout << "usage: ./n13_wilson_kcycle_mrhs L mass beta n_refine coarse_dof gauge_file tile nrhs [verify]\n"; return -1; }
  N13 s;
  const int rc = s.build(argc, argv);
  if (rc) return rc;
  const int nrhs = stoi(argv[8]);
  const bool verify0 = (argc > 9) && std::string(argv[9]) == "verify0";   // re-solve system 0 only (= the n13 solve)
  const bool verify = verify0 || ((argc > 9) && std::string(argv[9]) == "verify");
  if (nrhs < 1 || nrhs > qmg::BATCH_MAX) { std::cout << "[QMG-ERROR]: nrhs must be in 1.." << qmg::BATCH_MAX << "\n"; return -1; }
  StatefulMultigridMG* mg = s.mg_object;
  BatchKcycle bk(mg, nrhs);
  if (!bk.supported()) { std::cout << "[QMG-ERROR]: the batched K-cycle implements the ORIGINAL-operator configuration only.\n"; return 4; }
  const size_t n = (size_t)s.lats[0]->get_size_cv_l();
  const unsigned all = qmg::full_mask(nrhs);

  qmg::BatchPool pool(n, nrhs);
  qmg::Batch b = pool.get(), x = pool.get(), Ax = pool.get();
  unsigned long long seed = s.seed;
  for (int k = 0; k < nrhs; k++) gaussian(b.vec(k), n, seed++);
  if (getenv("QMG_MRHS_POINT") && nrhs > 1) {   // test hook: system 1 becomes a point source, which converges on its own schedule
    zero_vector(b.vec(1), n);
    qmg::set_element(b.vec(1), 5, complex<double>(1.0, 0.0));
  }
  const std::vector<double> bsq = qmg::bnorm2sq(b, n, all);
  qmg::bzero(x, n, all);

  inversion_verbose_struct verb = s.verb;
  verb.verbosity = s.quiet ? VERB_SUMMARY : VERB_DETAIL;
  qmg_stream_sync(0);
  auto t0 = std::chrono::steady_clock::now();
  std::vector<inversion_info> inv = bgcr_core(x, b, (int)n, s.max_iter, s.tol, s.restart_freq, apply_stencil_2D_M_batch, (void*)mg->get_stencil(0),
                                              mg_preconditioner_batch, (void*)&bk, all, true, &verb, "VPGCR-restart");
  qmg_stream_sync(0);
  const double solve_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  apply_stencil_2D_M_batch(Ax, x, all, (void*)mg->get_stencil(0));
  const std::vector<double> rsq = qmg::bdiffnorm2sq(b, Ax, n, all);
  bool ok_ = true;
  long total_iters = 0;
  cout << setprecision(12);
  for (int k = 0; k < nrhs; k++) {
    const double true_res = sqrt(rsq[k] / bsq[k]);
    cout << "[QMG-MRHS]: rhs " << k << " " << (inv[k].success ? "converged" : "failed to converge") << " in " << inv[k].iter << " iterations ; alleged tolerance "
         << sqrt(inv[k].resSq / bsq[k]) << " ; check tolerance " << true_res << "\n";
    ok_ = ok_ && inv[k].success && true_res < 10 * s.tol;
    total_iters += inv[k].iter;
  }
  s.print_ops_stats();
  cout << "[QMG-TIMING]: setup " << s.setup_s << " s ; batched solve of " << nrhs << " systems " << solve_s << " s ; aggregate outer iterations/s " << total_iters / solve_s
       << " ; systems/s " << nrhs / solve_s << "\n";

  if (verify) {
    // every system again, alone, through the single-vector path
    double single_s = 0.0, worst = 0.0;
    int max_diff_iter = 0;
    inversion_verbose_struct vq(VERB_NONE, "");
    complex<double>* x1 = mg->check_out(0);
    const int nver = verify0 ? 1 : nrhs;
    for (int k = 0; k < nver; k++) {
      zero_vector(x1, n);
      qmg_stream_sync(0);
      auto t1 = std::chrono::steady_clock::now();
      inversion_info i1 = minv_vector_gcr_var_precond_restart(x1, b.vec(k), (int)n, s.max_iter, s.tol, s.restart_freq, apply_stencil_2D_M, (void*)mg->get_stencil(0),
                                                              StatefulMultigridMG::mg_preconditioner, (void*)mg, &vq);
      qmg_stream_sync(0);
      single_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
      const double diff = sqrt(diffnorm2sq(x1, x.vec(k), n) / norm2sq(x1, n));
      cout << "[QMG-MRHS-VERIFY]: rhs " << k << " single-path iterations " << i1.iter << " (batched " << inv[k].iter << ") ; relative solution difference " << diff
           << " ; single-path solve " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() << " s\n";
      worst = std::max(worst, diff);
      max_diff_iter = std::max(max_diff_iter, std::abs(i1.iter - inv[k].iter));
      ok_ = ok_ && i1.success;
    }
    mg->check_in(x1, 0);
    cout << "[QMG-MRHS-VERIFY]: worst relative solution difference " << worst << " ; largest iteration-count difference " << max_diff_iter << " ; one-at-a-time solves "
         << single_s * nrhs / nver << " s" << (nver < nrhs ? " (extrapolated from system 0)" : "") << " vs batched " << solve_s << " s = " << (single_s * nrhs / nver) / solve_s << "x\n";
    // both solve to 1e-10: solutions agree to cond(A) * 1e-10
    ok_ = ok_ && worst < 1e-6 && max_diff_iter <= 1;
  }
  s.destroy();
  return ok_ ? 0 : 1;
}
