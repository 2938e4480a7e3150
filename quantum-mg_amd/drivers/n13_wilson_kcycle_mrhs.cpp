// n13_wilson_kcycle_mrhs -- the n13 K-cycle solve for a LOCK-STEP BATCH of independent right-hand sides on one GPU
// (BASELINE configs[3]/[4]: independent right-hand sides, several per GPU; include/qmg/batch.hpp).
//   ./n13_wilson_kcycle_mrhs L mass beta n_refine coarse_dof gauge_file tile nrhs [verify|verify0] [f32]
// `f32` (or QMG_F32_KCYCLE=1): the K-cycle preconditioner runs in complex<float> on the fp32 shadow of the hierarchy (batch.hpp).
// Setup is n13's (n13_setup.hpp).  Right-hand side k is the gaussian vector of seed (first solve seed)+k, so system 0
// is exactly the system n13_wilson_kcycle solves.  All nrhs <= 16 systems advance through the same VPGCR / K-cycle
// iteration together: every coarse operator and null vector is streamed once per step for the whole batch and the
// coarse applies run on the f64 matrix cores.
// Under a one-process-per-GPU launcher (RANK / WORLD_SIZE / LOCAL_RANK) each rank solves its own nrhs systems.
// With `verify`, every system is then solved again ALONE by the single-vector path (krylov.hpp / multigrid.hpp) and the
// two solutions, iteration counts and wall times are compared.
#include "n13_setup.hpp"
#include "mrhs_solve.hpp"

static void n13_print_stats(void* p) { ((N13*)p)->print_ops_stats(); }

int main(int argc, char** argv) {
  qmg_driver::Guard guard;
  if (argc < 9) { std::cout << "usage: ./n13_wilson_kcycle_mrhs L mass beta n_refine coarse_dof gauge_file tile nrhs [verify]\n"; return -1; }
  N13 s;
  const int rc = s.build(argc, argv);
  if (rc) return rc;
  const int nrhs = stoi(argv[8]);
  const int vmode = (argc > 9 && std::string(argv[9]) == "verify") ? 1 : (argc > 9 && std::string(argv[9]) == "verify0") ? 2 : 0;
  // several GPUs: one process per GPU (RANK / WORLD_SIZE / LOCAL_RANK from the launcher); every rank builds the same
  // hierarchy and solves ITS OWN nrhs systems (seeds shifted by rank * nrhs) -- independent right-hand sides sharded over
  // ranks, batched within a rank, no collective in the solve
  const int rank = getenv("RANK") ? atoi(getenv("RANK")) : 0;
  if (getenv("WORLD_SIZE") && atoi(getenv("WORLD_SIZE")) > 1) cout << "[QMG-INFO]: rank " << rank << " of " << getenv("WORLD_SIZE") << " solves systems " << rank * nrhs << " .. " << (rank + 1) * nrhs - 1 << "\n";
  const bool ok_ = mrhs_solve_and_report(s.mg_object, s.lats[0], nrhs, s.seed + (unsigned long long)rank * nrhs, s.tol, s.max_iter, s.restart_freq, s.quiet, vmode, s.setup_s,
                                         n13_print_stats, (void*)&s, QMG_MATVEC_ORIGINAL, std::string(argv[argc - 1]) == "f32");
  s.destroy();
  return qmg_driver::leave(ok_ ? 0 : 1);
}
