// n13_setup.hpp -- argument parsing and multigrid setup shared by n13_wilson_kcycle and n13_wilson_kcycle_mrhs
// (tests/n13_wilson_kcycle/wilson_kcycle.cpp:86-372 restated on the device facade; see n13_wilson_kcycle.cpp).
#ifndef N13_SETUP_HPP
#define N13_SETUP_HPP
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <iomanip>
#include <cstdio>
#include <iostream>
#include <string>
#include <vector>

#include "../include/qmg/qmg.hpp"
#include "driver_common.hpp"

using namespace std;

// Null-vector relaxation for `count` vectors of one level, `batch` at a time in lock step (the reference does
// them one by one, tests/n13_wilson_kcycle/wilson_kcycle.cpp:340-366: guess -> orthogonalise against the finished vectors
// -> BiCGStab-6 on A e = -A guess (tol 5e-5, 500 iterations) -> e + guess -> orthogonalise).  The solves of a batch share
// every read of the operator; within a batch the guesses can only be orthogonalised against the batches before it, the
// finished vectors are orthogonalised in the reference's order afterwards.  T = float runs the relaxation on complex<float>
// copies of the operator and of the vectors (half the bytes of a bandwidth-bound loop that stops at 5e-5 anyway); the
// vectors come back as complex<double> and everything downstream (chiral projection, block orthonormalisation, Galerkin
// build) is fp64.
template <typename T>
inline bool relax_null_vectors_batched(Stencil2D* st, complex<double>** null_vectors, int count, long fsize, int batch, unsigned long long& seed) {
  const bool f32 = sizeof(T) == sizeof(float);
  if (f32 && !st->enable_f32_shadow()) { std::cout << "[QMG-ERROR]: no memory for the fp32 copy of the operator\n"; return false; }
  BatchOp op(st, QMG_MATVEC_ORIGINAL);
  for (int j0 = 0; j0 < count; j0 += batch) {
    const int nb = (count - j0 < batch) ? count - j0 : batch;
    const unsigned mask = qmg::full_mask(nb);
    qmg::BatchPoolT<T> pool((size_t)fsize, nb);
    qmg::BatchT<T> G = pool.get(), B = pool.get(), X = pool.get();
    if (!G.p || !B.p || !X.p) { std::cout << "[QMG-ERROR]: no memory for a batch of " << nb << " null vectors\n"; return false; }
    for (int k = 0; k < nb; k++) {
      complex<double>* g = null_vectors[j0 + k];
      gaussian_lattice(g, st->lat->get_dim_mu(0), st->lat->get_dim_mu(1), st->lat->get_nc(), seed++);
      for (int m = 0; m < j0; m++) orthogonal(g, null_vectors[m], fsize);
      qmg::ok(qmg_convert(G.vec(k), qmg::dtype_of<T>::value, g, QMG_C64, (size_t)fsize, qmg::current_stream()), "qmg_convert");
    }
    apply_stencil_typed_batch<T>(B, G, mask, (void*)&op);
    const qmg::cvec mone(nb, -1.0);
    qmg::bblas<T>(QMG_BOP_CAX, &mone, 0, 0, 0, B, (size_t)fsize, mask);
    qmg::bzero(X, (size_t)fsize, mask);
    bminv_vector_bicgstab_l_zero_guess<T>(X, B, (int)fsize, 500, 5e-5, 6, apply_stencil_typed_batch<T>, (void*)&op, mask);
    qmg::bcxpy(G, X, (size_t)fsize, mask);
    for (int k = 0; k < nb; k++) {
      complex<double>* v = null_vectors[j0 + k];
      qmg::ok(qmg_convert(v, QMG_C64, X.vec(k), qmg::dtype_of<T>::value, (size_t)fsize, qmg::current_stream()), "qmg_convert");
      for (int m = 0; m < j0 + k; m++) orthogonal(v, null_vectors[m], fsize);
    }
  }
  if (f32) st->disable_f32_shadow();
  return true;
}

struct N13 {
  int x_len, y_len, n_refine, coarse_dof;
  double mass, tol;
  int max_iter, restart_freq;
  bool quiet;
  const char* dump_dir;
  unsigned long long seed;
  double setup_s;
  inversion_verbose_struct verb;
  Lattice2D** lats;
  Wilson2D* wilson_op;
  StatefulMultigridMG* mg_object;
  TransferMG** transfer_objs;
  StatefulMultigridMG::LevelSolveMG** level_solve_objs;
  StatefulMultigridMG::CoarsestSolveMG* coarsest_solve_obj;
  complex<double>* gauge_field;

  // returns 0 on success (the reference's exit codes otherwise)
  int build(int argc, char** argv);
  void print_ops_stats() {
    cout << setprecision(6);
    for (int i = 0; i <= n_refine; i++)
      cout << "[QMG-OPS-STATS]: Level " << i << " NullVec " << mg_object->get_tracker_count(QMG_DSLASH_TYPE_NULLVEC, i) << " PreSmooth "
           << mg_object->get_tracker_count(QMG_DSLASH_TYPE_PRESMOOTH, i) << " Krylov " << mg_object->get_tracker_count(QMG_DSLASH_TYPE_KRYLOV, i)
           << " PostSmooth " << mg_object->get_tracker_count(QMG_DSLASH_TYPE_POSTSMOOTH, i) << " Total " << mg_object->get_total_count(i) << "\n";
  }
  void destroy() {
    qmg_driver::phase("teardown: multigrid objects", false);
    delete mg_object;
    for (int i = 0; i < n_refine; i++) { delete transfer_objs[i]; delete level_solve_objs[i]; }
    delete[] transfer_objs; delete[] level_solve_objs; delete coarsest_solve_obj;
    delete wilson_op;
    for (int i = 0; i <= n_refine; i++) delete lats[i];
    delete[] lats;
    deallocate_vector(&gauge_field);
    qmg::VecPool::release_all();
  }
};

inline int N13::build(int argc, char** argv) {
  if (argc < 5) {
    std::cout << "Error: ./wilson_kcycle expects four arguments, L, mass, beta, n_refine. Try mass = -0.075 for beta 6.0.\n";
    return -1;
  }
  cout << setprecision(20);
  // one process per GPU: the launcher's LOCAL_RANK picks the device (torchrun sets it); a single process uses device 0
  // (y-slab mode, n13_wilson_kcycle_slab: the caller has initialised the device and the communicator and called qmg::slab_begin())
  const bool dd = qmg::slab().on;
  const bool root = !dd || qmg::slab().rank == 0;
  if (!dd && !qmg::ok(qmg_init(getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : 0), "qmg_init")) return 2;
  // Galerkin matrices of the preconditioner levels are STORED as complex<float> by default (multigrid.hpp); QMG_COARSE_F32=0 keeps them fp64
  MultigridMG::coarse_storage_from_env();   // QMG_COARSE_F32=0/1, QMG_COARSE_BITS=64/32/16
  x_len = stoi(argv[1]); y_len = stoi(argv[1]);
  mass = stod(argv[2]);
  const double beta = stod(argv[3]);
  (void)beta;
  n_refine = stoi(argv[4]);
  coarse_dof = (argc > 5) ? stoi(argv[5]) : 8;
  const string gauge_file = (argc > 6) ? argv[6] : "../../tests/golden/l64t64b60_heatbath.dat";
  const int tile = (argc > 7) ? stoi(argv[7]) : 64;
  quiet = getenv("QMG_QUIET") != 0 || !root;
  dump_dir = dd ? 0 : getenv("QMG_DUMP_DIR");   // test hook: null vectors, rhs and solution as raw complex128
  const int dof = Wilson2D::get_dof();
  const int x_block = 4, y_block = 4;
  tol = 1e-10; max_iter = 1000; restart_freq = 32;
  if (getenv("QMG_MAX_ITER")) max_iter = atoi(getenv("QMG_MAX_ITER"));   // test hook: cap the outer iterations (residual-history comparisons)
  const double inner_tol = 0.2; const int inner_max_iter = 1000; const int inner_restart_freq = 32;
  const int n_pre_smooth = 2; const double pre_smooth_tol = 1e-15;
  const int n_post_smooth = 2; const double post_smooth_tol = 1e-15;
  const double coarsest_tol = 0.2; const int coarsest_max_iter = 1000; const int coarsest_restart_freq = 32;
  seed = 1337ull;
  // QMG_SMOOTHER=cgne: the K-cycle's smoothers in their CGNE form (LevelSolveMG::pre_cgne / post_cgne, stateful_multigrid.h:847-857: MR on M M^dagger,
  // then M^dagger) on every level; needs the dagger stencil of every smoothed level.  No reference test sets the flags; the default is plain MR.
  const bool cgne = getenv("QMG_SMOOTHER") && std::string(getenv("QMG_SMOOTHER")) == "cgne";
  // null-vector relaxation: QMG_NULL_BATCH systems in lock step (default 8; 1 = one at a time in the reference's order, n13:340-366),
  // on complex<float> copies of the level's operator unless QMG_NULL_F32=0 (the relaxation stops at 5e-5)
  int null_batch = getenv("QMG_NULL_BATCH") ? atoi(getenv("QMG_NULL_BATCH")) : 8;
  if (null_batch < 1) null_batch = 1;
  if (null_batch > qmg::BATCH_MAX) null_batch = qmg::BATCH_MAX;
  const bool null_f32 = getenv("QMG_NULL_F32") ? atoi(getenv("QMG_NULL_F32")) != 0 : true;

  verb.verbosity = !root ? VERB_NONE : quiet ? VERB_SUMMARY : VERB_DETAIL;
  verb.verb_prefix = "Level 0: ";
  verb.precond_verbosity = (quiet || !root) ? VERB_NONE : VERB_SUMMARY;
  verb.precond_verb_prefix = "Prec ";
  inversion_verbose_struct verb_null(VERB_NONE, "");

  // y-slab mode: this rank holds y_len / world rows of every level; the slab must stay a whole, even number of block rows down to the coarsest level
  const int world = dd ? qmg::slab().world : 1;
  int y_loc = y_len / world;
  {
    int rows = y_loc;
    bool fits = (y_len % world == 0) && !(rows & 1);
    for (int i = 0; i < n_refine && fits; i++) { fits = (rows % y_block == 0); rows /= y_block; fits = fits && !(rows & 1) && rows >= 2; }
    if (!fits) { if (root) std::cout << "[QMG-ERROR]: " << y_len << " rows do not split into " << world << " slabs of whole, even block rows on every level.\n"; return 4; }
  }
  qmg_driver::phase("setup: gauge field", root);
  lats = new Lattice2D*[n_refine + 1];
  lats[0] = new Lattice2D(x_len, y_loc, dof);
  Lattice2D* lat_gauge = new Lattice2D(x_len, y_len, 1);
  gauge_field = allocate_vector<complex<double>>(lat_gauge->get_size_gauge());
  bool got = (x_len == tile) ? read_gauge_u1(gauge_field, lat_gauge, gauge_file) : read_gauge_u1_tiled(gauge_field, lat_gauge, gauge_file, tile);
  if (!got) return 3;
  delete lat_gauge;

  qmg_driver::phase("setup: fine operator", root);
  auto t_setup0 = std::chrono::steady_clock::now();
  wilson_op = new Wilson2D(lats[0], mass, gauge_field);
  // QMG_COARSEST_TYPE=mmd | mdm (test hook): the coarsest solve by CG on M M^dagger / M^dagger M (stateful_multigrid.h:930-960); QMG_NORMAL_SHIFT=s adds s to it
  const std::string coarsest_env = getenv("QMG_COARSEST_TYPE") ? getenv("QMG_COARSEST_TYPE") : "";
  const QMGStencilType coarsest_type = (coarsest_env == "mmd") ? QMG_MATVEC_M_MDAGGER : (coarsest_env == "mdm") ? QMG_MATVEC_MDAGGER_M : QMG_MATVEC_ORIGINAL;
  const bool want_dagger = cgne || coarsest_type != QMG_MATVEC_ORIGINAL;
  if (want_dagger) wilson_op->build_dagger_stencil();
  level_solve_objs = new StatefulMultigridMG::LevelSolveMG*[n_refine];
  coarsest_solve_obj = new StatefulMultigridMG::CoarsestSolveMG;
  coarsest_solve_obj->coarsest_stencil_app = coarsest_type;
  if (getenv("QMG_NORMAL_SHIFT")) coarsest_solve_obj->normal_shift = atof(getenv("QMG_NORMAL_SHIFT"));
  coarsest_solve_obj->coarsest_tol = coarsest_tol;
  coarsest_solve_obj->coarsest_iters = coarsest_max_iter;
  coarsest_solve_obj->coarsest_restart_freq = coarsest_restart_freq;
  mg_object = new StatefulMultigridMG(lats[0], wilson_op, coarsest_solve_obj);

  int curr_x_len = x_len, curr_y_len = y_loc;
  transfer_objs = new TransferMG*[n_refine];
  double t_null = 0.0, t_ortho = 0.0, t_galerkin = 0.0;   // setup split: null-vector relaxation / block orthonormalisation / Galerkin build
  auto now = [] { qmg_stream_sync(qmg::current_stream()); return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  for (int i = 1; i <= n_refine; i++) {
    curr_x_len /= x_block; curr_y_len /= y_block;
    lats[i] = new Lattice2D(curr_x_len, curr_y_len, coarse_dof);
    const long fsize = lats[i - 1]->get_size_cv_l();
    complex<double>** null_vectors = new complex<double>*[coarse_dof];
    for (int j = 0; j < coarse_dof; j++) { null_vectors[j] = allocate_vector<complex<double>>(fsize); zero_vector(null_vectors[j], fsize); }
    static const char* const phase_names[3][8] = {
      {"setup: level 1 null vectors", "setup: level 2 null vectors", "setup: level 3 null vectors", "setup: level 4 null vectors", "setup: level 5 null vectors", "setup: level 6 null vectors", "setup: level 7 null vectors", "setup: level 8+ null vectors"},
      {"setup: level 1 block orthonormalisation", "setup: level 2 block orthonormalisation", "setup: level 3 block orthonormalisation", "setup: level 4 block orthonormalisation", "setup: level 5 block orthonormalisation", "setup: level 6 block orthonormalisation", "setup: level 7 block orthonormalisation", "setup: level 8+ block orthonormalisation"},
      {"setup: level 1 Galerkin build", "setup: level 2 Galerkin build", "setup: level 3 Galerkin build", "setup: level 4 Galerkin build", "setup: level 5 Galerkin build", "setup: level 6 Galerkin build", "setup: level 7 Galerkin build", "setup: level 8+ Galerkin build"}};
    const int pl = (i - 1 < 7) ? i - 1 : 7;
    qmg_driver::phase(phase_names[0][pl], root);
    auto t0 = now();
    if (null_batch > 1) {
      bool done = null_f32 ? relax_null_vectors_batched<float>(mg_object->get_stencil(i - 1), null_vectors, coarse_dof / 2, fsize, null_batch, seed)
                           : relax_null_vectors_batched<double>(mg_object->get_stencil(i - 1), null_vectors, coarse_dof / 2, fsize, null_batch, seed);
      if (!done) return 5;
    } else
    for (int j = 0; j < coarse_dof / 2; j++) {
      complex<double>* rand_guess = mg_object->get_storage(i - 1)->check_out();
      gaussian_lattice(rand_guess, lats[i - 1]->get_dim_mu(0), lats[i - 1]->get_dim_mu(1), lats[i - 1]->get_nc(), seed++);
      for (int k = 0; k < j; k++) orthogonal(rand_guess, null_vectors[k], fsize);
      complex<double>* Arand_guess = mg_object->get_storage(i - 1)->check_out();
      zero_vector(Arand_guess, fsize);
      mg_object->get_stencil(i - 1)->apply_M(Arand_guess, rand_guess);
      cax(-1.0, Arand_guess, fsize);
      minv_vector_bicgstab_l(null_vectors[j], Arand_guess, (int)fsize, 500, 5e-5, 6, apply_stencil_2D_M, (void*)mg_object->get_stencil(i - 1), &verb_null);
      cxpy(rand_guess, null_vectors[j], fsize);
      mg_object->get_storage(i - 1)->check_in(rand_guess);
      mg_object->get_storage(i - 1)->check_in(Arand_guess);
      for (int k = 0; k < j; k++) orthogonal(null_vectors[j], null_vectors[k], fsize);
    }
    for (int j = 0; j < coarse_dof / 2; j++) {
      mg_object->get_stencil(i - 1)->chiral_projection_both(null_vectors[j], null_vectors[j + lats[i]->get_nc() / 2]);
      normalize(null_vectors[j], fsize);
      normalize(null_vectors[j + lats[i]->get_nc() / 2], fsize);
    }
    if (dump_dir) {   // raw complex128, coarse_dof vectors back to back (pre block-ortho), for the parity test
      std::string path = std::string(dump_dir) + "/nullvecs_level" + std::to_string(i - 1) + ".bin";
      FILE* f = fopen(path.c_str(), "wb");
      for (int j = 0; j < coarse_dof; j++) {
        std::vector<complex<double>> h = qmg::to_host(null_vectors[j], (size_t)fsize);
        fwrite(h.data(), sizeof(complex<double>), h.size(), f);
      }
      fclose(f);
    }
    auto t1 = now();
    qmg_driver::phase(phase_names[1][pl], root);
    transfer_objs[i - 1] = new TransferMG(lats[i - 1], lats[i], null_vectors, true, false, QMG_DOUBLE_PROJECTION);
    auto t2 = now();
    qmg_driver::phase(phase_names[2][pl], root);
    level_solve_objs[i - 1] = new StatefulMultigridMG::LevelSolveMG;
    level_solve_objs[i - 1]->fine_stencil_app = QMG_MATVEC_ORIGINAL;
    level_solve_objs[i - 1]->intermediate_tol = inner_tol;
    level_solve_objs[i - 1]->intermediate_iters = inner_max_iter;
    level_solve_objs[i - 1]->intermediate_restart_freq = inner_restart_freq;
    level_solve_objs[i - 1]->pre_tol = pre_smooth_tol;
    level_solve_objs[i - 1]->pre_iters = n_pre_smooth;
    level_solve_objs[i - 1]->post_tol = post_smooth_tol;
    level_solve_objs[i - 1]->post_iters = n_post_smooth;
    level_solve_objs[i - 1]->pre_cgne = level_solve_objs[i - 1]->post_cgne = cgne;
    mg_object->push_level(lats[i], transfer_objs[i - 1], level_solve_objs[i - 1], true, true, MultigridMG::QMG_MULTIGRID_PRECOND_ORIGINAL,
                          want_dagger ? CoarseOperator2D::QMG_COARSE_BUILD_DAGGER : CoarseOperator2D::QMG_COARSE_BUILD_ORIGINAL, null_vectors);
    auto t3 = now();
    t_null += secs(t0, t1); t_ortho += secs(t1, t2); t_galerkin += secs(t2, t3);
    for (int j = 0; j < coarse_dof; j++) deallocate_vector(&null_vectors[j]);
    delete[] null_vectors;
    if (root) cout << "[QMG-SETUP]: level " << i << " = " << curr_x_len << "x" << curr_y_len << " nc " << coarse_dof << " built\n";
  }
  qmg_stream_sync(0);
  if (root && cgne) cout << "[QMG-INFO]: CGNE smoothers (MR on M M^dagger, then M^dagger) on every level\n";
  if (root && mg_object->any_coarse_f16()) cout << "[QMG-INFO]: Galerkin matrices of the preconditioner levels are stored as complex<half> (QMG_COARSE_BITS=16; default 32, 64: fp64)\n";
  else if (root && mg_object->any_coarse_f32()) cout << "[QMG-INFO]: Galerkin matrices of the preconditioner levels (and the null vectors of the K-cycle's own transfers) are stored as complex<float> (QMG_COARSE_F32=0: fp64)\n";
  setup_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_setup0).count();
  if (root) cout << setprecision(6) << "[QMG-SETUP-TIMING]: null vectors " << t_null << " s ; block orthonormalisation " << t_ortho << " s ; Galerkin build " << t_galerkin
       << " s ; total " << setup_s << " s\n" << setprecision(20);

  return 0;
}

#endif
