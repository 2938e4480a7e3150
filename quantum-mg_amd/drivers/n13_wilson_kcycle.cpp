// n13_wilson_kcycle -- the build's counterpart of tests/n13_wilson_kcycle/wilson_kcycle.cpp on the GPU.
//   ./n13_wilson_kcycle L mass beta n_refine [coarse_dof] [gauge_file] [tile]
// Same constants (n13:86-122): 4x4 blocks, outer VPGCR tol 1e-10 / 1000 its / restart 32; inner and
// coarsest tol 0.2 / 1000 / 32; 2+2 MR smoothing (omega 0.85); null vectors = coarse_dof/2 gaussian
// vectors relaxed on the residual equation by BiCGStab-6 (500 its, 5e-5), orthogonalised, chirally doubled
// (n13:330-372), block-orthonormalised twice by TransferMG.  Same final lines: "Multigrid converged in N
// iterations ..." and "Check tolerance r" (true residual).
// Gauge field: the reference picks a stored config by (L, beta) (n13:148-192) and otherwise runs a heatbath;
// here `gauge_file` (default: the committed l64t64b60 fixture) is read as-is when L = tile and tiled
// periodically for larger L.  Random vectors come from the device generator (seeds 1337, 1338, ...).
#include "n13_setup.hpp"

int main(int argc, char** argv) {
  qmg_driver::Guard guard;
  N13 s;
  const int rc = s.build(argc, argv);
  if (rc) return qmg_driver::leave(rc);
  Lattice2D** lats = s.lats;
  StatefulMultigridMG* mg_object = s.mg_object;
  const double tol = s.tol, setup_s = s.setup_s;
  const int max_iter = s.max_iter, restart_freq = s.restart_freq, n_refine = s.n_refine;
  const char* dump_dir = s.dump_dir;
  unsigned long long seed = s.seed;
  inversion_verbose_struct& verb = s.verb;
  inversion_info invif;
  (void)n_refine;

  complex<double>* b = mg_object->check_out(0);
  gaussian(b, lats[0]->get_size_cv_l(), seed++);
  const double bnorm = sqrt(norm2sq(b, lats[0]->get_size_cv_l()));
  complex<double>* x = mg_object->check_out(0);
  zero_vector(x, lats[0]->get_size_cv_l());
  complex<double>* Ax = mg_object->check_out(0);
  zero_vector(Ax, lats[0]->get_size_cv_l());

  qmg_reserve_kcycle_scratch(mg_object, (size_t)lats[0]->get_size_cv_l(), restart_freq < 24 ? restart_freq : 24);   // the solve's scratch, outside its timed region
  qmg_stream_sync(0);
  qmg_driver::phase("solve");
  const qmg::AllocStats a0 = qmg::alloc_stats();
  auto t0 = std::chrono::steady_clock::now();
  invif = minv_vector_gcr_var_precond_restart(x, b, lats[0]->get_size_cv(), max_iter, tol, restart_freq, apply_stencil_2D_M, (void*)mg_object->get_stencil(0),
                                              StatefulMultigridMG::mg_preconditioner, (void*)mg_object, &verb);
  qmg_stream_sync(0);
  const double solve_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const qmg::AllocStats a1 = qmg::alloc_stats();
  qmg_driver::phase("check");
  cout << "Multigrid " << (invif.success ? "converged" : "failed to converge") << " in " << invif.iter << " iterations with alleged tolerance "
       << sqrt(invif.resSq) / bnorm << ".\n";
  zero_vector(Ax, lats[0]->get_size_cv_l());
  mg_object->apply_stencil(Ax, x, 0);
  const double true_res = sqrt(diffnorm2sq(b, Ax, lats[0]->get_size_cv_l())) / bnorm;
  cout << "Check tolerance " << true_res << "\n";
  if (dump_dir) {
    const size_t n = (size_t)lats[0]->get_size_cv_l();
    std::vector<complex<double>> hb = qmg::to_host(b, n), hx = qmg::to_host(x, n);
    FILE* f = fopen((std::string(dump_dir) + "/b.bin").c_str(), "wb"); fwrite(hb.data(), sizeof(complex<double>), n, f); fclose(f);
    f = fopen((std::string(dump_dir) + "/x.bin").c_str(), "wb"); fwrite(hx.data(), sizeof(complex<double>), n, f); fclose(f);
  }
  s.print_ops_stats();
  cout << "[QMG-TIMING]: setup " << setup_s << " s ; solve " << solve_s << " s ; outer iterations/s " << invif.iter / solve_s
       << " ; device allocator inside the solve " << a1.seconds - a0.seconds << " s in " << (a1.mallocs - a0.mallocs) + (a1.frees - a0.frees) << " calls\n";
  mg_object->check_in(Ax, 0); mg_object->check_in(x, 0); mg_object->check_in(b, 0);

  const bool ok_ = invif.success && true_res < 10 * tol;
  s.destroy();
  return qmg_driver::leave(ok_ ? 0 : 1);
}
