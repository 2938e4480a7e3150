// n13_wilson_kcycle_slab -- the n13 K-cycle (tests/n13_wilson_kcycle/wilson_kcycle.cpp: same constants, same setup, same solve) with
// ONE lattice cut into y-slabs over the ranks: every level of the hierarchy is decomposed.
//   ./n13_wilson_kcycle_slab L mass beta n_refine [coarse_dof] [gauge_file] [tile] [nrhs=K [f32]]
//   nrhs=K: K right-hand sides in lock-step batches (include/qmg/batch.hpp) on the slabs, `f32`: with the K-cycle in complex<float>
//   ranks: one process per GPU under the launcher (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT), or
//          QMG_COMM_EMULATE=R: R host threads of this process on one GPU (the test transport of csrc/qmg_comm.hip)
// Nothing in the multigrid classes knows about ranks.  qmg::slab_begin() switches this thread's facade to slab mode:
//   * every Lattice2D is the rank's rows of the level; Wilson2D fills its rows of the stencil from the global links;
//   * Stencil2D::launch exchanges the halo rows of the right-hand side before every apply (kernel W on the fine level,
//     kernel B with halos on the Galerkin levels); CoarseOperator2D exchanges the null vectors' halo rows for its build;
//   * reductions are summed over the ranks inside the library, so every Krylov layer takes the same decisions everywhere;
//   * gaussian_lattice draws the slab's rows of the single-domain run's vectors, so a decomposed run follows the
//     single-domain run up to the rounding of its reductions.
// Restrict / prolong, block orthonormalisation and all BLAS-1 are slab-local and untouched.  The null-vector relaxations run one
// at a time (the batch kernels have no halo step yet).  Same final lines as n13_wilson_kcycle, printed by rank 0.
#include <thread>

#include "n13_setup.hpp"
#include "mrhs_solve.hpp"

static int run(int rank, int world, int device, int argc, char** argv) {
  if (!qmg::ok(qmg_init(device), "qmg_init")) return 2;
  if (!qmg::ok(qmg_comm_init_env(world, rank), "qmg_comm_init_env")) return 2;
  if (!qmg::slab_begin()) return 2;
  const bool root = rank == 0;
  N13 s;
  int rc = s.build(argc, argv);
  int all = 0;
  qmg_comm_all_ok(rc == 0, &all);
  if (!all) { qmg::slab_end(); qmg_comm_finalize(); return rc ? rc : 5; }
  Lattice2D** lats = s.lats;
  StatefulMultigridMG* mg_object = s.mg_object;
  unsigned long long seed = s.seed;
  const long n = lats[0]->get_size_cv_l();
  int nrhs = 0;
  bool f32 = false;
  bool adjoint = false;
  for (int i = 8; i < argc; i++) {
    if (std::string(argv[i]).rfind("nrhs=", 0) == 0) nrhs = atoi(argv[i] + 5);
    if (std::string(argv[i]) == "f32") f32 = true;
    if (std::string(argv[i]) == "adjoint") adjoint = true;
  }
  if (adjoint) {
    // `adjoint`: build_dagger_stencil on every level's slab (the boundary rows' +-y hops come from the neighbouring ranks: one exchange of
    // the -y and one of the +y hopping field) and check <u, M v> = <M^dag u, v> with the distributed reductions -- a dagger stencil whose
    // boundary rows were wired to the wrong rank breaks the identity at O(1)
    int good = 1;
    for (int level = 0; level < mg_object->get_num_levels(); level++) {
      Stencil2D* st = mg_object->get_stencil(level);
      Lattice2D* lat = lats[level];
      const long nl = lat->get_size_cv_l();
      complex<double>*u = mg_object->check_out(level), *v = mg_object->check_out(level), *w = mg_object->check_out(level);
      gaussian_lattice(u, lat->get_dim_mu(0), lat->get_dim_mu(1), lat->get_nc(), seed++);
      gaussian_lattice(v, lat->get_dim_mu(0), lat->get_dim_mu(1), lat->get_nc(), seed++);
      st->build_dagger_stencil();
      zero_vector(w, nl); st->apply_M(w, v);
      const complex<double> d1 = dot(u, w, nl);
      zero_vector(w, nl); st->apply_M_dagger(w, u);
      const complex<double> d2 = dot(w, v, nl);
      const double rel = std::abs(d1 - d2) / std::abs(d1);
      if (!(rel < 1e-12)) good = 0;
      if (root) cout << setprecision(15) << "[QMG-SLAB]: level " << level << " <u, M v> = " << d1 << " ; <M^dag u, v> = " << d2 << " ; rel diff " << rel << (rel < 1e-12 ? " (ok)" : " (MISMATCH)") << "\n";
      mg_object->check_in(w, level); mg_object->check_in(v, level); mg_object->check_in(u, level);
    }
    qmg_comm_all_ok(good, &all);
    s.destroy();
    qmg::slab_end();
    qmg_comm_finalize();
    return all ? 0 : 1;
  }
  if (nrhs > 0) {   // the lock-step batch engine on slabs: one halo exchange per batch apply, per-system reductions summed over the ranks
    const bool ok_b = mrhs_solve_and_report(mg_object, lats[0], nrhs, seed, s.tol, s.max_iter, s.restart_freq, true, 0, s.setup_s, 0, 0, QMG_MATVEC_ORIGINAL, f32);
    qmg_comm_all_ok(ok_b, &all);
    s.destroy();
    qmg::slab_end();
    qmg_comm_finalize();
    return all ? 0 : 1;
  }

  complex<double>* b = mg_object->check_out(0);
  gaussian_lattice(b, lats[0]->get_dim_mu(0), lats[0]->get_dim_mu(1), lats[0]->get_nc(), seed++);
  const double bnorm = sqrt(norm2sq(b, n));
  complex<double>* x = mg_object->check_out(0);
  zero_vector(x, n);
  complex<double>* Ax = mg_object->check_out(0);
  zero_vector(Ax, n);

  qmg_reserve_kcycle_scratch(s.mg_object, (size_t)lats[0]->get_size_cv_l(), s.restart_freq < 24 ? s.restart_freq : 24);   // the solve's scratch, outside its timed region
  qmg_stream_sync(qmg::current_stream());
  auto t0 = std::chrono::steady_clock::now();
  inversion_info invif = minv_vector_gcr_var_precond_restart(x, b, lats[0]->get_size_cv(), s.max_iter, s.tol, s.restart_freq, apply_stencil_2D_M,
                                                             (void*)mg_object->get_stencil(0), StatefulMultigridMG::mg_preconditioner, (void*)mg_object, &s.verb);
  qmg_stream_sync(qmg::current_stream());
  const double solve_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  zero_vector(Ax, n);
  mg_object->apply_stencil(Ax, x, 0);
  const double true_res = sqrt(diffnorm2sq(b, Ax, n)) / bnorm;
  const double xnorm2 = norm2sq(x, n);
  if (root) {
    cout << "Multigrid " << (invif.success ? "converged" : "failed to converge") << " in " << invif.iter << " iterations with alleged tolerance "
         << sqrt(invif.resSq) / bnorm << ".\n";
    cout << "Check tolerance " << true_res << "\n";
    cout << setprecision(15) << "[QMG-SLAB]: world " << world << " ; |b| " << bnorm << " ; |x|^2 " << xnorm2 << "\n";
    s.print_ops_stats();
    cout << "[QMG-TIMING]: setup " << s.setup_s << " s ; solve " << solve_s << " s ; outer iterations/s " << invif.iter / solve_s << "\n" << std::flush;
  }
  mg_object->check_in(Ax, 0); mg_object->check_in(x, 0); mg_object->check_in(b, 0);
  const int good = invif.success && true_res < 10 * s.tol;
  qmg_comm_all_ok(good, &all);
  s.destroy();
  qmg::slab_end();
  qmg_comm_finalize();
  return all ? 0 : 1;
}

int main(int argc, char** argv) {
  qmg_driver::Guard guard;
  const int emulate = getenv("QMG_COMM_EMULATE") ? atoi(getenv("QMG_COMM_EMULATE")) : 0;
  if (emulate > 0)   // R ranks as host threads on this one GPU (csrc/qmg_comm.hip: ThreadWorld)
    return qmg_driver::leave(qmg_driver::emulate_ranks(emulate, [&](int r) { return run(r, emulate, 0, argc, argv); }, [](void* st) { qmg::current_stream() = st; }));
  const int rank = getenv("RANK") ? atoi(getenv("RANK")) : 0;
  const int world = getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1;
  return qmg_driver::leave(run(rank, world, getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : 0, argc, argv));
}
