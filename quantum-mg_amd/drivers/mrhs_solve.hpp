// mrhs_solve.hpp -- the batched outer solve shared by n13_wilson_kcycle_mrhs and n22_wilson_kcycle_adaptive (nrhs=K):
// nrhs gaussian right-hand sides (seeds seed, seed+1, ...), solved in lock-step batches (include/qmg/batch.hpp) sized to
// the HBM that is free (qmg::batch_systems_that_fit), per-system report, and optionally the same systems re-solved alone
// by the single-vector path.
#ifndef MRHS_SOLVE_HPP
#define MRHS_SOLVE_HPP
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#include "../include/qmg/qmg.hpp"

// verify_mode: 0 none, 1 every system, 2 system 0 only.  Returns true when every system converged (and verified).
// f32_kcycle (or QMG_F32_KCYCLE in the environment): the K-cycle that preconditions the fp64 outer VPGCR runs entirely in
// complex<float> on the fp32 shadow of the hierarchy (include/qmg/batch.hpp mg_preconditioner_batch_mixed) -- the fp32
// instantiation BASELINE configs[4] names.  The outer solve, its tolerance and the true-residual check stay fp64.
inline bool mrhs_solve_and_report(StatefulMultigridMG* mg, Lattice2D* lat0, int nrhs, unsigned long long seed, double tol, int max_iter, int restart_freq,
                                  bool quiet, int verify_mode, double setup_s, void (*print_ops_stats)(void*), void* stats_arg,
                                  QMGStencilType solve_type = QMG_MATVEC_ORIGINAL, bool f32_kcycle = false) {
  using namespace std;
  // y-slab mode: every rank runs this in lock step on its rows; rank 0 reports
  static std::ostream discard(nullptr);
  std::ostream& cout = (qmg::slab().on && qmg::slab().rank != 0) ? discard : std::cout;
  if (nrhs < 1) { std::cout << "[QMG-ERROR]: nrhs must be positive\n"; return false; }
  const size_t n = (size_t)lat0->get_size_cv_l();
  {
    BatchKcycle probe(mg, 1);
    if (!BatchOp::supported(solve_type)) { std::cout << "[QMG-ERROR]: this driver's outer solve is on the ORIGINAL, RIGHT_JACOBI or RIGHT_SCHUR operator.\n"; return false; }
    if (!probe.supported()) { std::cout << "[QMG-ERROR]: the batched K-cycle does not implement this hierarchy's level / coarsest operator types (or a variant stencil they name is not built).\n"; return false; }
  }
  if (getenv("QMG_F32_KCYCLE")) f32_kcycle = true;
  if (f32_kcycle) {
    BatchKcycle sh(mg, 1);
    const bool half_fine = getenv("QMG_F16_FINE") != 0;   // level-0 matrices of the fp32 K-cycle in 16 bits
    const bool half_coarse = getenv("QMG_F16_COARSE") != 0;   // the Galerkin levels' matrices of the fp32 K-cycle in 16 bits
    if (!sh.enable_f32_hierarchy(half_fine, half_coarse)) { std::cout << "[QMG-ERROR]: could not build the fp32 shadow of the hierarchy\n"; return false; }
    cout << "[QMG-INFO]: K-cycle preconditioner in fp32 (complex<float> vectors, matrices and null vectors; fp64 outer VPGCR)"
         << (half_fine ? "; fine-level matrices of the K-cycle stored in 16 bits" : "") << (half_coarse ? "; coarse-level matrices of the K-cycle stored in 16 bits" : "") << "\n";
  }
  // batch size: what fits (an outer solve rarely needs its whole restart length; 48 directions is a safe expectation for
  // these K-cycles, and bgcr_core stops loudly if a basis vector cannot be allocated)
  const int per_batch = qmg::batch_systems_that_fit(mg, std::min(restart_freq > 0 ? restart_freq : max_iter, 48), std::min(nrhs, qmg::BATCH_MAX));
  cout << "[QMG-MRHS]: " << nrhs << " systems in lock-step batches of " << per_batch << "\n";

  inversion_verbose_struct verb;
  const bool mute = qmg::slab().on && qmg::slab().rank != 0;
  verb.verbosity = mute ? VERB_NONE : quiet ? VERB_SUMMARY : VERB_DETAIL;
  verb.verb_prefix = "Level 0: ";
  verb.precond_verbosity = (quiet || mute) ? VERB_NONE : VERB_SUMMARY;
  verb.precond_verb_prefix = "Prec ";

  bool ok_ = true;
  long total_iters = 0;
  double solve_s = 0.0, single_s = 0.0, worst = 0.0, alloc_in_solve_s = 0.0;
  long allocs_in_solve = 0;
  int max_diff_iter = 0, nver = 0;
  cout << setprecision(12);
  for (int k0 = 0; k0 < nrhs; k0 += per_batch) {
    const int nb = std::min(per_batch, nrhs - k0);
    const unsigned all = qmg::full_mask(nb);
    BatchKcycle bk(mg, nb);
    std::vector<inversion_info> inv;
    std::vector<double> bsq, rsq;
    {
      qmg::BatchPool pool(n, nb);
      qmg::Batch b = pool.get(), x = pool.get(), Ax = pool.get();
      if (b.p == 0 || x.p == 0 || Ax.p == 0) { std::cout << "[QMG-ERROR]: out of device memory for a batch of " << nb << " systems\n"; return false; }
      for (int k = 0; k < nb; k++) gaussian_lattice(b.vec(k), lat0->get_dim_mu(0), lat0->get_dim_mu(1), lat0->get_nc(), seed + (unsigned long long)(k0 + k));
      if (getenv("QMG_MRHS_POINT") && k0 == 0 && nb > 1) {   // test hook: system 1 becomes a point source, which converges on its own schedule
        zero_vector(b.vec(1), n);
        qmg::set_element(b.vec(1), 5, complex<double>(1.0, 0.0));
      }
      bsq = qmg::bnorm2sq(b, n, all);
      // prepare (a copy for ORIGINAL and RIGHT_JACOBI; b_e - D'_eo b_o for the Schur system), solve, reconstruct (RIGHT_JACOBI: x = C^-1 y) -- as the
      // drivers do for one system
      const bool schur = solve_type == QMG_MATVEC_RIGHT_SCHUR, precd = solve_type != QMG_MATVEC_ORIGINAL;
      qmg::Batch b_prep = schur ? pool.get() : b, y = precd ? pool.get() : x;
      if (b_prep.p == 0 || y.p == 0) { std::cout << "[QMG-ERROR]: out of device memory for a batch of " << nb << " systems\n"; return false; }
      BatchOp op0(mg->get_stencil(0), solve_type);
      if (schur) prepare_M_batch(mg->get_stencil(0), solve_type, b_prep, b, all);
      // the solve's scratch, outside its timed region: the batch's outer directions are full-stride vectors
      qmg_reserve_kcycle_scratch(mg, n, std::min(restart_freq > 0 ? restart_freq : max_iter, 64), nb);
      qmg_stream_sync(0);
      const int repeats = getenv("QMG_MRHS_REPEAT") ? atoi(getenv("QMG_MRHS_REPEAT")) : 1;   // diagnostic: time the same solve again
      for (int rep = 0; rep < repeats; rep++) {
        qmg::bzero(y, n, all);
        qmg_stream_sync(0);
        const qmg::AllocStats a0 = qmg::alloc_stats();
        auto t0 = std::chrono::steady_clock::now();
        inv = bgcr_core<double>(y, b_prep, (int)(schur ? n / 2 : n), max_iter, tol, restart_freq, apply_stencil_typed_batch<double>, (void*)&op0,
                                f32_kcycle ? mg_preconditioner_batch_mixed : mg_preconditioner_batch<double>, (void*)&bk, all, true, &verb, "VPGCR-restart");
        qmg_stream_sync(0);
        const double t_rep = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const qmg::AllocStats a1 = qmg::alloc_stats();
        alloc_in_solve_s = a1.seconds - a0.seconds; allocs_in_solve = (a1.mallocs - a0.mallocs) + (a1.frees - a0.frees);
        if (repeats > 1) cout << "[QMG-MRHS]: repeat " << rep << " solve " << t_rep << " s (of which " << alloc_in_solve_s << " s in " << allocs_in_solve << " device allocator calls)\n";
        if (rep == repeats - 1) solve_s += t_rep;
      }
      if (precd) reconstruct_M_batch(mg->get_stencil(0), solve_type, x, y, b, all);
      apply_stencil_2D_M_batch(Ax, x, all, (void*)mg->get_stencil(0));   // true residual against the ORIGINAL operator
      rsq = qmg::bdiffnorm2sq(b, Ax, n, all);
      for (int k = 0; k < nb; k++) {
        const double true_res = sqrt(rsq[k] / bsq[k]);
        cout << "[QMG-MRHS]: rhs " << k0 + k << " " << (inv[k].success ? "converged" : "failed to converge") << " in " << inv[k].iter << " iterations ; alleged tolerance "
             << sqrt(inv[k].resSq / bsq[k]) << " ; check tolerance " << true_res << "\n";
        ok_ = ok_ && inv[k].success && true_res < 20 * tol;
        total_iters += inv[k].iter;
      }
      if (verify_mode != 0) {   // the same systems, alone, through the single-vector path
        inversion_verbose_struct vq(VERB_NONE, "");
        complex<double>* x1 = mg->check_out(0);
        for (int k = 0; k < nb; k++) {
          if (verify_mode == 2 && k0 + k > 0) break;
          zero_vector(x1, n);
          qmg_stream_sync(0);
          auto t1 = std::chrono::steady_clock::now();
          inversion_info i1;
          if (!precd)
            i1 = minv_vector_gcr_var_precond_restart(x1, b.vec(k), (int)n, max_iter, tol, restart_freq, apply_stencil_2D_M, (void*)mg->get_stencil(0),
                                                     StatefulMultigridMG::mg_preconditioner, (void*)mg, &vq);
          else {   // n19's sequence for one system
            complex<double>* bp1 = mg->check_out(0);
            complex<double>* y1 = mg->check_out(0);
            zero_vector(bp1, n); zero_vector(y1, n);
            mg->get_stencil(0)->prepare_M(bp1, b.vec(k), solve_type);
            i1 = minv_vector_gcr_var_precond_restart(y1, bp1, (int)(schur ? n / 2 : n), max_iter, tol, restart_freq, Stencil2D::get_apply_function(solve_type), (void*)mg->get_stencil(0),
                                                     StatefulMultigridMG::mg_preconditioner, (void*)mg, &vq);
            mg->get_stencil(0)->reconstruct_M(x1, y1, b.vec(k), solve_type);
            mg->check_in(y1, 0); mg->check_in(bp1, 0);
          }
          qmg_stream_sync(0);
          const double t_single = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
          single_s += t_single;
          nver++;
          const double diff = sqrt(diffnorm2sq(x1, x.vec(k), n) / norm2sq(x1, n));
          cout << "[QMG-MRHS-VERIFY]: rhs " << k0 + k << " single-path iterations " << i1.iter << " (batched " << inv[k].iter << ") ; relative solution difference " << diff
               << " ; single-path solve " << t_single << " s\n";
          worst = std::max(worst, diff);
          max_diff_iter = std::max(max_diff_iter, std::abs(i1.iter - inv[k].iter));
          ok_ = ok_ && i1.success;
        }
        mg->check_in(x1, 0);
      }
    }
    // (the pool reuses its blocks by capacity: nothing is released between batches)
  }
  if (print_ops_stats) print_ops_stats(stats_arg);
  cout << "[QMG-TIMING]: setup " << setup_s << " s ; batched solve of " << nrhs << " systems " << solve_s << " s ; aggregate outer iterations/s " << total_iters / solve_s
       << " ; systems/s " << nrhs / solve_s << " ; device allocator inside the last solve " << alloc_in_solve_s << " s in " << allocs_in_solve << " calls\n";
  if (nver > 0) {
    cout << "[QMG-MRHS-VERIFY]: worst relative solution difference " << worst << " ; largest iteration-count difference " << max_diff_iter << " ; one-at-a-time solves "
         << single_s * nrhs / nver << " s" << (nver < nrhs ? " (extrapolated)" : "") << " vs batched " << solve_s << " s = " << (single_s * nrhs / nver) / solve_s << "x\n";
    // both solve to 1e-10: solutions agree to cond(A) * 1e-10
    ok_ = ok_ && worst < 1e-6 && max_diff_iter <= 1;
  }
  return ok_;
}

#endif
