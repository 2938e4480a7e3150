// n22_wilson_kcycle_adaptive -- the build's counterpart of
// tests/n22_wilson_kcycle_adaptive/wilson_kcycle.cpp on the GPU: adaptive multigrid setup, then the n13 solve.
//   ./n22_wilson_kcycle_adaptive L mass beta n_refine n_setup [gauge_file] [tile] [solve_type | nrhs=K]
// Setup (n22:230-426):
//   * initial hierarchy: per level, coarse_dof/2 gaussian vectors relaxed by 10 Richardson iterations
//     (omega 0.33, n22:289, 664), Gram-Schmidt + normalise, chiral doubling, TransferMG (block-ortho x2),
//     Galerkin coarse operator; lower levels by `build_coarse_by_restrict` (n22:628-706), which -- despite its
//     name -- draws fresh gaussian vectors on each level;
//   * n_setup adaptive passes (n22:336-426): on each level the test vector (level 0: the previous test vector;
//     lower levels: the restriction of the finer level's test vector) is improved by 10 iterations of the
//     CURRENT K-cycle (VPGCR, setup-time LevelSolve: 8 inner iterations, tol 1e-10, n22:245-247), the transfer
//     and coarse operator of that level are rebuilt, and all lower levels are rebuilt from fresh vectors;
//   * all setup operator counts are moved to the NullVec tracker (n22:428-431) and the LevelSolve parameters
//     reset to the solve values (inner tol 0.2 / 1000 / 32).
// Solve (n22:494-497): VPGCR tol 1e-10, 1000 its, restart 64, K-cycle preconditioner.  Prints the reference's
// [QMG-OPS-STATS] / [QMG-ITER-STATS] lines (the reference prints the PreSmooth count under "PostSmooth" too,
// n22:511; here the PostSmooth count is printed).
// Several GPUs (SURVEY 8e "setup phase"; not in n22): started one process per GPU with RANK / WORLD_SIZE / LOCAL_RANK in the
// environment (torchrun sets them), every rank builds the same hierarchy, but the adaptive relaxations of a level are
// SHARDED over the ranks (test vector j belongs to rank j mod world) and exchanged with ONE sum all-reduce per level
// (qmg_allreduce_sum_f64, RCCL over xGMI; the vectors a rank does not own are zero in its buffer), after which all ranks
// continue identically.  The 128-byte RCCL id comes through the launcher's rendezvous (qmg_comm_init_env: QMG_COMM_ID_HEX
// or one TCP exchange with rank 0 on MASTER_ADDR:MASTER_PORT+1); a failure on one rank is made known to all ranks
// (qmg_comm_all_ok) before the next collective, so nobody is left blocked.  UNVERIFIED on more than one GPU (the pool
// has one-GPU boxes): exercised with a forced one-rank communicator and, on the host side, by two-process TCP tests.
// In the solve every rank then takes its own right-hand side (seed + rank): the path's "independent right-hand sides".
// solve_type (optional, not in n22): "schur" builds rbjacobi stencils on every level and solves as n19 does --
// the "red-black preconditioned" variant of BASELINE configs[4].
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <string>
#include <thread>
#include <ctime>

#include "../include/qmg/qmg.hpp"
#include "driver_common.hpp"
#include "mrhs_solve.hpp"

#include <vector>

using namespace std;

// per host thread (QMG_COMM_EMULATE: the ranks of a slab run are threads): the seed counter, and whether this rank reports
static thread_local unsigned long long g_seed = 1337ull;
static thread_local bool t_root = true;
static std::ostream& qout() { static std::ostream discard(nullptr); return t_root ? std::cout : discard; }
#define cout qout()
// a Gaussian vector on a level's lattice (slab mode: the slab's rows of the single-domain run's vector)
static void gaussian_on(Lattice2D* lat, complex<double>* v, unsigned long long seed) { gaussian_lattice(v, lat->get_dim_mu(0), lat->get_dim_mu(1), lat->get_nc(), seed); }

struct Setup {
  StatefulMultigridMG* mg;
  complex<double>*** test_vectors;
  int coarse_dof;
  bool schur;
};

static MultigridMG::QMGMultigridPrecondStencil coarsen_from(const Setup& s) {
  return s.schur ? MultigridMG::QMG_MULTIGRID_PRECOND_RIGHT_BLOCK_JACOBI : MultigridMG::QMG_MULTIGRID_PRECOND_ORIGINAL;
}
static CoarseOperator2D::QMGCoarseBuildStencil build_extra(const Setup& s) {
  return s.schur ? CoarseOperator2D::QMG_COARSE_BUILD_RBJACOBI : CoarseOperator2D::QMG_COARSE_BUILD_ORIGINAL;
}

// n22:628-706
static TransferMG* build_coarse_by_restrict(Setup& s, int fine_level, Lattice2D* coarse_lat, StatefulMultigridMG::LevelSolveMG* new_level_solve,
                                            bool fresh_build, inversion_verbose_struct& verb) {
  StatefulMultigridMG* mg = s.mg;
  const int fine_idx = fine_level, coarse_idx = fine_level + 1, coarse_dof = coarse_lat->get_nc();
  const long n = mg->get_lattice(fine_idx)->get_size_cv_l();
  complex<double>** null_vectors = new complex<double>*[coarse_dof];
  for (int j = 0; j < coarse_dof / 2; j++) {
    null_vectors[j] = allocate_vector<complex<double>>(n);
    null_vectors[j + coarse_dof / 2] = allocate_vector<complex<double>>(n);
    zero_vector(null_vectors[j], n);
    zero_vector(null_vectors[j + coarse_dof / 2], n);
    complex<double>* temp_rand = mg->get_storage(fine_idx)->check_out();
    gaussian_on(mg->get_lattice(fine_idx), temp_rand, g_seed++);
    inversion_info invif = minv_vector_richardson(s.test_vectors[fine_idx][j], temp_rand, (int)n, 10, 1e-10, 0.33, 250, apply_stencil_2D_M,
                                                  (void*)mg->get_stencil(fine_idx), &verb);
    mg->add_tracker_count(QMG_DSLASH_TYPE_NULLVEC, invif.ops_count, fine_idx);
    mg->get_storage(fine_idx)->check_in(temp_rand);
    for (int k = 0; k < j; k++) orthogonal(s.test_vectors[fine_idx][j], s.test_vectors[fine_idx][k], n);
    normalize(s.test_vectors[fine_idx][j], n);
    copy_vector(null_vectors[j], s.test_vectors[fine_idx][j], n);
    mg->get_stencil(fine_idx)->chiral_projection_both(null_vectors[j], null_vectors[j + coarse_dof / 2]);
  }
  // n22:682 passes no doubling type here (QMG_DOUBLE_NONE), so the operator built from this transfer has no notion of
  // chirality (coarse.h:104-116) and, one level further down, chiral_projection_both leaves every "down" partner zero:
  // block-orthonormalising those divides by zero, and a hierarchy of four or more levels is all NaN -- in the reference
  // as here (BASELINE configs[4] asks for four levels).  Declaring the doubling changes nothing for <= 3 levels (the
  // flag of the coarsest operator is never read) and makes deeper hierarchies work.
  TransferMG* transfer_obj = new TransferMG(mg->get_lattice(fine_idx), coarse_lat, null_vectors, true, false, QMG_DOUBLE_PROJECTION);
  if (s.schur && !mg->get_stencil(fine_idx)->built_rbjacobi) mg->get_stencil(fine_idx)->build_rbjacobi_stencil();
  if (fresh_build) mg->push_level(coarse_lat, transfer_obj, new_level_solve, true, true, coarsen_from(s), build_extra(s), null_vectors);
  else mg->update_level(coarse_idx, coarse_lat, transfer_obj, new_level_solve, true, true, coarsen_from(s), build_extra(s), null_vectors);
  for (int j = 0; j < coarse_dof; j++) deallocate_vector(&null_vectors[j]);
  delete[] null_vectors;
  return transfer_obj;
}

// slab_mode (QMG_SLAB=1 under a launcher, or QMG_COMM_EMULATE=R threads): ONE lattice cut into y-slabs on every level; every rank
// takes part in every relaxation and in the one solve (facade slab mode).  Otherwise the ranks hold replicas and shard the work.
static int run(int proc_rank, int proc_world, int local_rank, bool slab_mode, int argc, char** argv) {
  g_seed = 1337ull;
  t_root = !slab_mode || proc_rank == 0;
  cout << setprecision(20);
  if (!qmg::ok(qmg_init(local_rank), "qmg_init")) return 2;
  // the rank / world the SHARDING logic below sees: in slab mode nothing is sharded by rank
  const int world = slab_mode ? 1 : proc_world, rank = slab_mode ? 0 : proc_rank;
  const bool use_comm = !slab_mode && (world > 1 || getenv("QMG_COMM_FORCE_RCCL") != 0);
  if (slab_mode) {
    if (!qmg::ok(qmg_comm_init_env(proc_world, proc_rank), "qmg_comm_init_env") || !qmg::slab_begin()) return 2;
  } else if (use_comm) {   // the RCCL id comes through the launcher's rendezvous (QMG_COMM_ID_HEX or one TCP exchange with rank 0; bounded waits)
    if (!qmg::ok(qmg_comm_init_env(world, rank), "qmg_comm_init_env")) return 2;
    cout << "[QMG-INFO]: rank " << rank << " of " << world << " on device " << local_rank << "\n";
  }
  // Galerkin matrices of the preconditioner levels are STORED as complex<float> by default (multigrid.hpp); QMG_COARSE_F32=0 keeps them fp64
  MultigridMG::coarse_storage_from_env();   // QMG_COARSE_F32=0/1, QMG_COARSE_BITS=64/32/16
  const int x_len = stoi(argv[1]), y_len = x_len;
  const double mass = stod(argv[2]);
  const int n_refine = stoi(argv[4]);
  const int n_setup = stoi(argv[5]);
  const string gauge_file = (argc > 6) ? argv[6] : "../../tests/golden/l64t64b60_heatbath.dat";
  const int tile = (argc > 7) ? stoi(argv[7]) : 64;
  // options after `tile`, any order (none of them in n22): "schur" = solve as n19 does; "nrhs=K" = after the reference's single
  // solve, K more gaussian systems in one lock-step batch; "f32" = those batched solves with the K-cycle in fp32
  bool schur = false, f32 = false;
  int nrhs_batched = 0;
  for (int i = 8; i < argc; i++) {
    const string o = argv[i];
    if (o == "schur") schur = true;
    else if (o == "f32") f32 = true;
    else if (o.rfind("nrhs=", 0) == 0) nrhs_batched = stoi(o.substr(5));
    else { cout << "Error: unknown option " << o << "\n"; return -1; }
  }
  if (f32 && nrhs_batched == 0) nrhs_batched = 1;
  const bool quiet = getenv("QMG_QUIET") != 0 || !t_root;
  const int dof = Wilson2D::get_dof();
  const int x_block = 4, y_block = 4, coarse_dof = 8;
  const double tol = 1e-10; const int max_iter = 1000; const int restart_freq = 64;
  const double inner_tol = 0.2; const int inner_max_iter = 1000; const int inner_restart_freq = 32;
  const int n_pre_smooth = 2; const double pre_smooth_tol = 1e-15;
  const int n_post_smooth = 2; const double post_smooth_tol = 1e-15;
  const double coarsest_tol = 0.2; const int coarsest_max_iter = 1000; const int coarsest_restart_freq = 32;
  const QMGStencilType solve_type = schur ? QMG_MATVEC_RIGHT_SCHUR : QMG_MATVEC_ORIGINAL;

  inversion_info invif;
  inversion_verbose_struct verb;
  verb.verbosity = quiet ? VERB_NONE : VERB_SUMMARY;
  verb.verb_prefix = "Level 0: ";
  verb.precond_verbosity = VERB_NONE;
  verb.precond_verb_prefix = "Prec ";

  // slab mode: this rank's rows of every level (whole, even block rows down to the coarsest level)
  const int slabs = slab_mode ? proc_world : 1;
  const int y_loc = y_len / slabs;
  {
    int rows = y_loc;
    bool fits = (y_len % slabs == 0) && !(rows & 1);
    for (int i = 0; i < n_refine && fits; i++) { fits = (rows % y_block == 0); rows /= y_block; fits = fits && !(rows & 1) && rows >= 2; }
    if (!fits) { cout << "[QMG-ERROR]: " << y_len << " rows do not split into " << slabs << " slabs of whole, even block rows on every level.\n"; return 4; }
  }
  Lattice2D** lats = new Lattice2D*[n_refine + 1];
  lats[0] = new Lattice2D(x_len, y_loc, dof);
  Lattice2D* lat_gauge = new Lattice2D(x_len, y_len, 1);
  complex<double>* gauge_field = allocate_vector<complex<double>>(lat_gauge->get_size_gauge());
  bool got = (x_len == tile) ? read_gauge_u1(gauge_field, lat_gauge, gauge_file) : read_gauge_u1_tiled(gauge_field, lat_gauge, gauge_file, tile);
  if (!got) return 3;
  delete lat_gauge;

  auto t_setup0 = std::chrono::steady_clock::now();
  Wilson2D* wilson_op = new Wilson2D(lats[0], mass, gauge_field);
  if (schur) wilson_op->build_rbjacobi_stencil();
  StatefulMultigridMG::CoarsestSolveMG* coarsest_solve_obj = new StatefulMultigridMG::CoarsestSolveMG;
  coarsest_solve_obj->coarsest_stencil_app = solve_type;
  coarsest_solve_obj->coarsest_tol = coarsest_tol;
  coarsest_solve_obj->coarsest_iters = coarsest_max_iter;
  coarsest_solve_obj->coarsest_restart_freq = coarsest_restart_freq;
  StatefulMultigridMG* mg_object = new StatefulMultigridMG(lats[0], wilson_op, coarsest_solve_obj);

  // lattices, test-vector storage, setup-time LevelSolve objects (n22:226-252)
  StatefulMultigridMG::LevelSolveMG** level_solve_objs = new StatefulMultigridMG::LevelSolveMG*[n_refine];
  TransferMG** transfer_objs = new TransferMG*[n_refine];
  complex<double>*** test_vectors = new complex<double>**[n_refine];
  int cx = x_len, cy = y_loc;
  for (int i = 1; i <= n_refine; i++) {
    const int fine_idx = i - 1;
    cx /= x_block; cy /= y_block;
    lats[i] = new Lattice2D(cx, cy, coarse_dof);
    test_vectors[fine_idx] = new complex<double>*[coarse_dof / 2];
    for (int j = 0; j < coarse_dof / 2; j++) {
      test_vectors[fine_idx][j] = allocate_vector<complex<double>>(lats[fine_idx]->get_size_cv_l());
      zero_vector(test_vectors[fine_idx][j], lats[fine_idx]->get_size_cv_l());
    }
    transfer_objs[fine_idx] = 0;
    level_solve_objs[fine_idx] = new StatefulMultigridMG::LevelSolveMG;
    level_solve_objs[fine_idx]->fine_stencil_app = solve_type;
    level_solve_objs[fine_idx]->intermediate_tol = 1e-10;
    level_solve_objs[fine_idx]->intermediate_iters = 8;
    level_solve_objs[fine_idx]->intermediate_restart_freq = 1024;
    level_solve_objs[fine_idx]->pre_tol = pre_smooth_tol;
    level_solve_objs[fine_idx]->pre_iters = n_pre_smooth;
    level_solve_objs[fine_idx]->post_tol = post_smooth_tol;
    level_solve_objs[fine_idx]->post_iters = n_post_smooth;
  }
  Setup S = {mg_object, test_vectors, coarse_dof, schur};

  // initial level 0 -> 1 (n22:270-313): as build_coarse_by_restrict but with chirality-preserving doubling
  {
    const int fine_idx = 0, coarse_idx = 1;
    const long n = lats[0]->get_size_cv_l();
    complex<double>** null_vectors = new complex<double>*[coarse_dof];
    for (int j = 0; j < coarse_dof / 2; j++) {
      null_vectors[j] = allocate_vector<complex<double>>(n);
      null_vectors[j + coarse_dof / 2] = allocate_vector<complex<double>>(n);
      zero_vector(null_vectors[j], n);
      zero_vector(null_vectors[j + coarse_dof / 2], n);
      complex<double>* temp_rand = mg_object->get_storage(fine_idx)->check_out();
      gaussian_on(lats[fine_idx], temp_rand, g_seed++);
      invif = minv_vector_richardson(test_vectors[fine_idx][j], temp_rand, (int)n, 10, 1e-10, 0.33, 250, apply_stencil_2D_M, (void*)mg_object->get_stencil(fine_idx), &verb);
      mg_object->add_tracker_count(QMG_DSLASH_TYPE_NULLVEC, invif.ops_count, fine_idx);
      mg_object->get_storage(fine_idx)->check_in(temp_rand);
      for (int k = 0; k < j; k++) orthogonal(test_vectors[fine_idx][j], test_vectors[fine_idx][k], n);
      normalize(test_vectors[fine_idx][j], n);
      copy_vector(null_vectors[j], test_vectors[fine_idx][j], n);
      mg_object->get_stencil(fine_idx)->chiral_projection_both(null_vectors[j], null_vectors[j + coarse_dof / 2]);
    }
    transfer_objs[fine_idx] = new TransferMG(lats[fine_idx], lats[coarse_idx], null_vectors, true, false, QMG_DOUBLE_PROJECTION);
    mg_object->push_level(lats[coarse_idx], transfer_objs[fine_idx], level_solve_objs[fine_idx], true, true, coarsen_from(S), build_extra(S), null_vectors);
    for (int j = 0; j < coarse_dof; j++) deallocate_vector(&null_vectors[j]);
    delete[] null_vectors;
  }
  for (int i = 1; i < n_refine; i++) transfer_objs[i] = build_coarse_by_restrict(S, i, lats[i + 1], level_solve_objs[i], true, verb);

  // adaptive passes (n22:336-426)
  for (int m = 0; m < n_setup; m++) {
    for (int i = 0; i < n_refine; i++) {
      const int fine_idx = i, coarse_idx = i + 1;
      const long n = lats[fine_idx]->get_size_cv_l();
      complex<double>** null_vectors = new complex<double>*[coarse_dof];
      // The coarse_dof/2 relaxations of one level are independent systems (each reads only its own test vector and the
      // CURRENT hierarchy; the Gram-Schmidt below touches vector j alone), so in the ORIGINAL-operator configuration
      // they advance in lock step through the batched K-cycle (include/qmg/batch.hpp) -- same results, the coarse
      // operators and null vectors streamed once per step for all of them (SURVEY 8e "setup phase").
      const int nb = coarse_dof / 2;
      const bool batched_setup = !schur && nb <= qmg::BATCH_MAX && getenv("QMG_NO_BATCHED_SETUP") == 0;
      if (batched_setup) {
        BatchKcycle bk(mg_object, nb);
        qmg::BatchPool bpool((size_t)n, nb);
        qmg::Batch T = bpool.get(), X = bpool.get();
        const unsigned all = qmg::full_mask(nb);
        for (int j = 0; j < nb; j++) {
          if (i == 0) copy_vector(T.vec(j), test_vectors[fine_idx][j], n);
          else { zero_vector(T.vec(j), n); mg_object->get_transfer(fine_idx - 1)->restrict_f2c(test_vectors[fine_idx - 1][j], T.vec(j)); }
        }
        qmg::bzero(X, n, all);
        inversion_verbose_struct vq(VERB_NONE, "");
        unsigned mine = 0;   // the systems this rank relaxes: j mod world == rank
        for (int j = 0; j < nb; j++) if (j % world == rank) mine |= 1u << j;
        std::vector<inversion_info> binv = bgcr_core(X, T, (int)n, 10, 1e-10, -1, apply_stencil_2D_M_batch, (void*)mg_object->get_stencil(fine_idx),
                                                     mg_preconditioner_batch, (void*)&bk, mine, true, &vq, "VPGCR");
        if (use_comm) {   // one collective per level: everybody gets every relaxed vector (foreign slots of X are still zero here)
          bool local_ok = true;
          for (int j = 0; j < nb; j++) if (qmg::is_active(mine, j) && !(binv[j].resSq == binv[j].resSq)) local_ok = false;   // a NaN relaxation on this rank
          int all_ok = 0;
          if (!qmg::ok(qmg_comm_all_ok(local_ok ? 1 : 0, &all_ok), "qmg_comm_all_ok") || !all_ok) {   // every rank learns of it BEFORE the data collective
            cout << "[QMG-ERROR]: rank " << rank << ": a rank reported a failed relaxation; all ranks stop\n";
            qmg_comm_finalize();
            return 2;
          }
          if (!qmg::ok(qmg_allreduce_sum_f64((double*)X.p, (size_t)2 * X.stride * nb, qmg::current_stream()), "qmg_allreduce_sum_f64")) return 2;
          qmg_stream_sync(qmg::current_stream());
        }
        for (int j = 0; j < nb; j++) {
          copy_vector(test_vectors[fine_idx][j], X.vec(j), n);
          if (!qmg::is_active(mine, j)) continue;   // relaxed (and reported) by another rank
          mg_object->add_tracker_count(QMG_DSLASH_TYPE_NULLVEC, binv[j].ops_count + 1, fine_idx);
          cout << "[QMG-SETUP]: pass " << m << " level " << fine_idx << " test vector " << j << ": " << binv[j].iter << " K-cycle iterations, residual "
               << sqrt(binv[j].resSq) << ", t = " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_setup0).count() << " s (batched)" << std::endl;
        }
      }
      for (int j = 0; j < coarse_dof / 2; j++) {
        null_vectors[j] = allocate_vector<complex<double>>(n);
        null_vectors[j + coarse_dof / 2] = allocate_vector<complex<double>>(n);
        if (!batched_setup) {
          complex<double>* temp_rand = mg_object->get_storage(fine_idx)->check_out();
          if (i == 0) copy_vector(temp_rand, test_vectors[fine_idx][j], n);
          else { zero_vector(temp_rand, n); mg_object->get_transfer(fine_idx - 1)->restrict_f2c(test_vectors[fine_idx - 1][j], temp_rand); }
          zero_vector(test_vectors[fine_idx][j], n);
          // 10 iterations of the current K-cycle on this level (n22:373-376); n22 itself uses the ORIGINAL operator here
          invif = minv_vector_gcr_var_precond(test_vectors[fine_idx][j], temp_rand, (int)n, 10, 1e-10, apply_stencil_2D_M, (void*)mg_object->get_stencil(fine_idx),
                                              schur ? (precond_op_cplx)0 : StatefulMultigridMG::mg_preconditioner, (void*)mg_object, &verb);
          mg_object->get_storage(fine_idx)->check_in(temp_rand);
          mg_object->add_tracker_count(QMG_DSLASH_TYPE_NULLVEC, invif.ops_count + 1, fine_idx);
          cout << "[QMG-SETUP]: pass " << m << " level " << fine_idx << " test vector " << j << ": " << invif.iter << " K-cycle iterations, residual "
               << sqrt(invif.resSq) << ", t = " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_setup0).count() << " s" << std::endl;
        }
        zero_vector(null_vectors[j], n);
        zero_vector(null_vectors[j + coarse_dof / 2], n);
        for (int k = 0; k < j; k++) orthogonal(test_vectors[fine_idx][j], test_vectors[fine_idx][k], n);
        normalize(test_vectors[fine_idx][j], n);
        copy_vector(null_vectors[j], test_vectors[fine_idx][j], n);
        mg_object->get_stencil(fine_idx)->chiral_projection_both(null_vectors[j], null_vectors[j + coarse_dof / 2]);
      }
      delete transfer_objs[fine_idx];
      transfer_objs[fine_idx] = new TransferMG(lats[fine_idx], lats[coarse_idx], null_vectors, true, false, QMG_DOUBLE_PROJECTION);
      mg_object->update_level(coarse_idx, lats[coarse_idx], transfer_objs[fine_idx], level_solve_objs[fine_idx], true, true, coarsen_from(S), build_extra(S), null_vectors);
      for (int j = i + 1; j < n_refine; j++) {
        delete transfer_objs[j];
        transfer_objs[j] = build_coarse_by_restrict(S, j, lats[j + 1], level_solve_objs[j], false, verb);
      }
      for (int j = 0; j < coarse_dof; j++) deallocate_vector(&null_vectors[j]);
      delete[] null_vectors;
      if (i < n_refine - 1) mg_object->go_coarser();
    }
    for (int i = 0; i < n_refine - 1; i++) mg_object->go_finer();
    cout << "[QMG-SETUP]: adaptive pass " << m << " done\n";
  }
  for (int i = 0; i <= n_refine; i++) mg_object->shift_all_to_nullvec(i);
  for (int i = 0; i < n_refine; i++) {   // solve-time parameters (n22:433-448)
    level_solve_objs[i]->fine_stencil_app = solve_type;
    level_solve_objs[i]->intermediate_tol = inner_tol;
    level_solve_objs[i]->intermediate_iters = inner_max_iter;
    level_solve_objs[i]->intermediate_restart_freq = inner_restart_freq;
  }
  qmg_stream_sync(qmg::current_stream());
  const double setup_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_setup0).count();
  if (mg_object->any_coarse_f16()) cout << "[QMG-INFO]: Galerkin matrices (and right-block-Jacobi hops) of the preconditioner levels are stored as complex<half> (QMG_COARSE_BITS=16; default 32, 64: fp64)\n";
  else if (mg_object->any_coarse_f32()) cout << "[QMG-INFO]: Galerkin matrices (and right-block-Jacobi hops) of the preconditioner levels are stored as complex<float> (QMG_COARSE_BITS=64: fp64)\n";

  const long n0 = lats[0]->get_size_cv_l();
  complex<double>* b = mg_object->check_out(0);
  g_seed += (unsigned long long)rank;   // independent right-hand sides: one per rank
  gaussian_on(lats[0], b, g_seed++);
  const double bnorm = sqrt(norm2sq(b, n0));
  complex<double>* x = mg_object->check_out(0);
  zero_vector(x, n0);
  complex<double>* Ax = mg_object->check_out(0);
  complex<double>* b_prep = mg_object->check_out(0);
  zero_vector(b_prep, n0);
  mg_object->get_stencil(0)->prepare_M(b_prep, b, solve_type);
  const int solve_size = schur ? (int)(n0 / 2) : (int)n0;

  verb.verbosity = !t_root ? VERB_NONE : quiet ? VERB_SUMMARY : VERB_DETAIL;   // (thread-emulated ranks share one stdout: only the root rank reports, or its lines are torn)
  verb.verb_prefix = "[QMG-MG-SOLVE-INFO]: Level 0 ";
  qmg_reserve_kcycle_scratch(mg_object, (size_t)solve_size, restart_freq);   // the solve's scratch, outside its timed region
  qmg_stream_sync(qmg::current_stream());
  const qmg::AllocStats a0 = qmg::alloc_stats();
  auto t0 = std::chrono::steady_clock::now();
  invif = minv_vector_gcr_var_precond_restart(x, b_prep, solve_size, max_iter, tol, restart_freq, Stencil2D::get_apply_function(solve_type),
                                              (void*)mg_object->get_stencil(0), StatefulMultigridMG::mg_preconditioner, (void*)mg_object, &verb);
  qmg_stream_sync(qmg::current_stream());
  const double solve_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const qmg::AllocStats a1 = qmg::alloc_stats();
  mg_object->add_tracker_count(QMG_DSLASH_TYPE_KRYLOV, invif.ops_count, 0);
  mg_object->add_iterations_count(invif.iter, 0);
  cout << "Multigrid " << (invif.success ? "converged" : "failed to converge") << " in " << invif.iter << " iterations with alleged tolerance "
       << sqrt(invif.resSq) / bnorm << ".\n";
  for (int i = 0; i < n_refine + 1; i++)
    cout << "[QMG-OPS-STATS]: Level " << i << " NullVec " << mg_object->get_tracker_count(QMG_DSLASH_TYPE_NULLVEC, i) << " PreSmooth "
              << mg_object->get_tracker_count(QMG_DSLASH_TYPE_PRESMOOTH, i) << " Krylov " << mg_object->get_tracker_count(QMG_DSLASH_TYPE_KRYLOV, i)
              << " PostSmooth " << mg_object->get_tracker_count(QMG_DSLASH_TYPE_POSTSMOOTH, i) << " Total " << mg_object->get_total_count(i) << "\n";
  std::vector<double> avg_iter = mg_object->query_average_iterations();
  for (int i = 0; i < n_refine + 1; i++) cout << "[QMG-ITER-STATS]: Level " << i << " AverageIters " << avg_iter[i] << "\n";
  complex<double>* x_rec = mg_object->check_out(0);
  zero_vector(x_rec, n0);
  mg_object->get_stencil(0)->reconstruct_M(x_rec, x, b, solve_type);
  zero_vector(Ax, n0);
  mg_object->apply_stencil(Ax, x_rec, 0);
  const double true_res = sqrt(diffnorm2sq(b, Ax, n0)) / bnorm;
  cout << "Check tolerance " << true_res << "\n";
  if (slab_mode) { const double xn = norm2sq(x_rec, n0); cout << setprecision(15) << "[QMG-SLAB]: world " << proc_world << " ; |b| " << bnorm << " ; |x|^2 " << xn << "\n" << setprecision(20); }
  cout << setprecision(6) << "[QMG-TIMING]: setup " << setup_s << " s ; solve " << solve_s << " s ; outer iterations/s " << invif.iter / solve_s
       << " ; device allocator inside the solve " << a1.seconds - a0.seconds << " s in " << (a1.mallocs - a0.mallocs) + (a1.frees - a0.frees) << " calls\n" << std::flush;   // (a parent that times the batched part out still reads this)
  mg_object->check_in(x_rec, 0); mg_object->check_in(b_prep, 0); mg_object->check_in(Ax, 0); mg_object->check_in(x, 0); mg_object->check_in(b, 0);

  bool ok_ = invif.success && true_res < 20 * tol;
  if (nrhs_batched > 0)
    ok_ = mrhs_solve_and_report(mg_object, lats[0], nrhs_batched, g_seed, tol, max_iter, restart_freq, quiet, getenv("QMG_MRHS_VERIFY") ? 1 : 0, setup_s, 0, 0, solve_type, f32) && ok_;
  delete mg_object;
  for (int i = 0; i < n_refine; i++) {
    delete transfer_objs[i]; delete level_solve_objs[i];
    for (int j = 0; j < coarse_dof / 2; j++) deallocate_vector(&test_vectors[i][j]);
    delete[] test_vectors[i];
  }
  delete[] test_vectors; delete[] transfer_objs; delete[] level_solve_objs; delete coarsest_solve_obj;
  delete wilson_op;
  for (int i = 0; i <= n_refine; i++) delete lats[i];
  delete[] lats;
  deallocate_vector(&gauge_field);
  qmg::VecPool::release_all();
  if (slab_mode) {
    int all = 0;
    qmg_comm_all_ok(ok_, &all);
    ok_ = all != 0;
    qmg::slab_end();
    qmg_comm_finalize();
  } else if (use_comm) qmg_comm_finalize();
  return ok_ ? 0 : 1;
}

int main(int argc, char** argv) {
  qmg_driver::Guard guard;
  if (argc < 6) {
    cout << "Error: ./wilson_kcycle expects five arguments, L, mass, beta, n_refine, n_setup. Try mass = -0.075 for beta 6.0.\n";
    return -1;
  }
  const int emulate = getenv("QMG_COMM_EMULATE") ? atoi(getenv("QMG_COMM_EMULATE")) : 0;
  if (emulate > 0)   // R ranks as host threads on this one GPU (csrc/qmg_comm.hip: ThreadWorld)
    return qmg_driver::leave(qmg_driver::emulate_ranks(emulate, [&](int r) { return run(r, emulate, 0, true, argc, argv); }, [](void* st) { qmg::current_stream() = st; }));
  const bool slab_mode = getenv("QMG_SLAB") != 0;
  return qmg_driver::leave(run(getenv("RANK") ? atoi(getenv("RANK")) : 0, getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1, getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : 0,
             slab_mode, argc, argv));
}
