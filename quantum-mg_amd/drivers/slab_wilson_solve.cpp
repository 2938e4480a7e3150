// slab_wilson_solve -- ONE Wilson system on ONE lattice, strong-scaled over the ranks by y-slabs (the reference
// is single-process and marks the spot in cshift/cshift_2d.h:39-42,72,89 "Becomes MPI").
//   ./slab_wilson_solve L mass beta [n_therm seed tol verify overlap]        one process per GPU: RANK / WORLD_SIZE / LOCAL_RANK /
//                                                                            MASTER_ADDR / MASTER_PORT (or QMG_COMM_ID_HEX)
// Every rank generates the same quenched U(1) field with the device heatbath (the links are 32 B/site and replicated),
// fills the stencil of ITS rows only (include/qmg/slab.hpp), takes its rows of one global Gaussian source and solves
// M x = b with BiCGStab-6 (the reference's fine-level solver, n13:359) from krylov.hpp UNCHANGED: the operator exchanges
// halo rows with the neighbouring ranks while the interior is applied, and the library's reductions sum over the ranks.
//   verify = 1 (default): before the solve, each rank also builds the single-domain operator and checks that its slab apply
//   (with the real exchange) reproduces its rows of the single-domain apply to 1e-13 -- the exchange's wiring (who sends which
//   row where) cannot be checked by a residual, a wrongly wired operator solves its own system just as well.
// Output (rank 0): "[QMG-SLAB]: ..." lines; the last one carries iterations, seconds, ms per apply and |x|^2 over the whole
// lattice, which must agree between runs with different numbers of ranks.
#include <chrono>
#include <cmath>
#include <iomanip>
#include <iostream>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../include/qmg/qmg.hpp"
#include "driver_common.hpp"
#include "../include/qmg/slab.hpp"

using namespace std;

static double now() { return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count(); }
static std::mutex g_print;
static void say(const std::ostringstream& line) { std::lock_guard<std::mutex> lk(g_print); std::cout << line.str() << std::flush; }

// one rank: a process of the launcher, or -- QMG_COMM_EMULATE=R -- one of R host threads of this process (test transport)
static int run(int rank, int world, int device, int argc, char** argv) {
  if (!qmg::ok(qmg_init(device), "qmg_init")) return 2;
  if (!qmg::ok(qmg_comm_init_env(world, rank), "qmg_comm_init_env")) return 2;
  const int L = stoi(argv[1]);
  const double mass = stod(argv[2]), beta = stod(argv[3]);
  const int n_therm = (argc > 4) ? stoi(argv[4]) : 200;
  const unsigned long long seed = (argc > 5) ? stoull(argv[5]) : 1337ull;
  const double tol = (argc > 6) ? stod(argv[6]) : 1e-10;
  const bool verify = (argc > 7) ? stoi(argv[7]) != 0 : true;
  const bool overlap = (argc > 8) ? stoi(argv[8]) != 0 : true;
  const bool root = rank == 0;

  qmg::SlabGeometry geo(L, L, world, rank);
  int all = 0;
  if (!qmg::ok(qmg_comm_all_ok(geo.valid ? 1 : 0, &all), "qmg_comm_all_ok") || !all) {
    if (root) { std::ostringstream o_; o_ << setprecision(15) << "[QMG-SLAB]: " << L << " rows do not split into " << world << " slabs of an even number of rows\n"; say(o_); }
    qmg_comm_finalize();
    return 3;
  }
  if (root) { std::ostringstream o_; o_ << setprecision(15) << "[QMG-SLAB]: " << L << " x " << L << " Wilson, mass " << mass << ", beta " << beta << ": " << world << " slab(s) of " << geo.Ly_local << " rows\n"; say(o_); }

  // ---- the same gauge field on every rank
  Lattice2D* lat_gauge = new Lattice2D(L, L, 1);
  const size_t ng = (size_t)lat_gauge->get_size_gauge();
  double* phases = allocate_vector<double>(ng);
  complex<double>* gauge = allocate_vector<complex<double>>(ng);
  qmg::ok(qmg_memset_zero(phases, sizeof(double) * ng, qmg::current_stream()), "qmg_memset_zero");
  HeatbathRng generator(seed);
  heatbath_noncompact_update(phases, lat_gauge, beta, n_therm, generator);
  polar_vector(phases, gauge, ng);
  const double plaq = std::real(get_plaquette_u1(gauge, lat_gauge));
  if (root) { std::ostringstream o_; o_ << setprecision(15) << "[QMG-SLAB]: plaquette after " << n_therm << " heatbath sweeps " << plaq << "\n"; say(o_); }

  // ---- this rank's rows of the operator and of the source
  qmg::SlabWilson2D* op = new qmg::SlabWilson2D(geo, mass, gauge);
  op->overlap = overlap;
  const size_t nl = op->size_cv(), n_glob = (size_t)2 * L * L;
  complex<double>* b = allocate_vector<complex<double>>(nl);
  complex<double>* x = allocate_vector<complex<double>>(nl);
  complex<double>* r = allocate_vector<complex<double>>(nl);
  {
    complex<double>* bg = allocate_vector<complex<double>>(n_glob);
    gaussian(bg, n_glob, seed + 17);
    qmg::slab_rows_of(b, bg, geo, 2);
    int ok_here = 1;
    if (verify) {   // the slab apply with the real exchange against this rank's rows of the single-domain apply
      Lattice2D* lat = new Lattice2D(L, L, 2);
      Wilson2D* wilson = new Wilson2D(lat, mass, gauge);
      complex<double>* yg = allocate_vector<complex<double>>(n_glob);
      wilson->apply_M(yg, bg);
      qmg::slab_rows_of(x, yg, geo, 2);                 // x := the rows the slab apply must reproduce
      op->apply_M(r, b);
      const double d2 = diffnorm2sq(r, x, nl), n2 = norm2sq(x, nl);   // local: distributed reductions are still off
      ok_here = (d2 <= 1e-26 * n2) ? 1 : 0;
      { std::ostringstream o_; o_ << setprecision(15) << "[QMG-SLAB]: rank " << rank << " rows [" << geo.y0 << ", " << geo.y0 + geo.Ly_local << "): slab apply vs single-domain apply, rel diff " << sqrt(d2 / n2)
           << (ok_here ? " (ok)" : " (MISMATCH)") << "\n"; say(o_); }
      deallocate_vector(&yg);
      delete wilson; delete lat;
    }
    deallocate_vector(&bg);
    if (!qmg::ok(qmg_comm_all_ok(ok_here, &all), "qmg_comm_all_ok") || !all) { qmg_comm_finalize(); return 4; }
  }

  // ---- apply timing: overlapped and serialised exchange
  qmg::ok(qmg_comm_set_distributed_reductions(1), "qmg_comm_set_distributed_reductions");
  double ms_apply[2] = {0.0, 0.0};
  for (int mode = 0; mode < 2; mode++) {
    op->overlap = mode == 0;
    for (int i = 0; i < 5; i++) op->apply_M(r, b);
    norm2sq(r, nl);                                       // all-reduce: every rank is here
    const double t0 = now();
    const int reps = 50;
    for (int i = 0; i < reps; i++) op->apply_M(r, b);
    norm2sq(r, nl);
    ms_apply[mode] = (now() - t0) / reps * 1e3;
  }
  op->overlap = overlap;
  if (root) { std::ostringstream o_; o_ << setprecision(15) << "[QMG-SLAB]: apply_M on a slab: " << ms_apply[0] << " ms with the exchange overlapped, " << ms_apply[1] << " ms serialised\n"; say(o_); }

  // ---- the solve
  zero_vector(x, nl);
  inversion_verbose_struct verb(VERB_NONE, "[QMG-SLAB-BICGSTAB]: ");
  const double bnorm = sqrt(norm2sq(b, nl));
  const long applies0 = op->applies;
  qmg::VecPool::reserve((size_t)nl, 18);   // the solver's work vectors, outside the timed region (GB-sized hipMallocs: 3 ms or seconds on this pool)
  qmg::ok(qmg_stream_sync(qmg::current_stream()), "qmg_stream_sync");
  const double t0 = now();
  inversion_info info = minv_vector_bicgstab_l(x, b, (int)nl, 100000, tol, 6, qmg::apply_slab_wilson_M, (void*)op, &verb);
  qmg::ok(qmg_stream_sync(qmg::current_stream()), "qmg_stream_sync");
  const double secs = now() - t0;
  op->apply_M(r, x);
  const double relres = sqrt(diffnorm2sq(b, r, nl)) / bnorm;
  const double xnorm2 = norm2sq(x, nl);
  if (root)
    { std::ostringstream o_; o_ << setprecision(15) << "[QMG-SLAB]: BiCGStab-6 " << (info.success ? "converged" : "FAILED") << " in " << info.iter << " iterations, " << secs << " s, " << op->applies - applies0 - 1
         << " applies, true relative residual " << relres << ", |b| " << bnorm << ", |x|^2 " << xnorm2 << ", world " << world << "\n"; say(o_); }
  const int good = info.success && relres < 10 * tol;
  qmg_comm_all_ok(good, &all);

  qmg::ok(qmg_comm_set_distributed_reductions(0), "qmg_comm_set_distributed_reductions");
  deallocate_vector(&b); deallocate_vector(&x); deallocate_vector(&r); deallocate_vector(&phases); deallocate_vector(&gauge);
  delete op; delete lat_gauge;
  qmg::VecPool::release_all();
  qmg_comm_finalize();
  return all ? 0 : 1;
}

int main(int argc, char** argv) {
  qmg_driver::Guard guard;
  if (argc < 4) { std::cout << "usage: ./slab_wilson_solve L mass beta [n_therm seed tol verify overlap]\n"; return -1; }
  const int emulate = getenv("QMG_COMM_EMULATE") ? atoi(getenv("QMG_COMM_EMULATE")) : 0;
  if (emulate > 0)   // R ranks as host threads on this one GPU (csrc/qmg_comm.hip: ThreadWorld)
    return qmg_driver::leave(qmg_driver::emulate_ranks(emulate, [&](int r) { return run(r, emulate, 0, argc, argv); }, [](void* st) { qmg::current_stream() = st; }));
  const int rank = getenv("RANK") ? atoi(getenv("RANK")) : 0;
  const int world = getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1;
  return qmg_driver::leave(run(rank, world, getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : 0, argc, argv));
}
