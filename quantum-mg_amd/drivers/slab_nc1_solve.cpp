// slab_nc1_solve -- the nc = 1 operators of the reference (Staggered2D, operators/staggered.h; GaugedLaplace2D, operators/gaugedlaplace.h)
// with ONE lattice cut into y-slabs over the ranks, through the facade's slab mode: the operator fills its rows of the stencil
// from the global links (qmg_staggered_fill_slab / qmg_laplace_fill_slab), every apply exchanges the halo rows of its right-hand side, the
// reductions are summed over the ranks.  Two solves per operator, as the reference's tests do them:
//   * the full operator with BiCGStab-6 (tests/n20...:105), and
//   * the even-odd preconditioned system with CG + reconstruct (tests/n04_staggered_test, tests/n03_gauge_laplace_test; staggered.h:190-240),
// each checked by the true residual of the FULL system.  gaussian_lattice draws the slab's rows of the single-domain right-hand side, so
// the iteration counts and |x|^2 printed here must agree between 1, 2, 4, ... ranks.
//   ./slab_nc1_solve L mass gauge_file tile          ranks: launcher environment, or QMG_COMM_EMULATE=R host threads on one GPU
#include <thread>

#include "n13_setup.hpp"

static int run(int rank, int world, int device, int argc, char** argv) {
  if (argc < 5) { if (rank == 0) std::cout << "usage: slab_nc1_solve L mass gauge_file tile\n"; return 2; }
  if (!qmg::ok(qmg_init(device), "qmg_init")) return 2;
  if (!qmg::ok(qmg_comm_init_env(world, rank), "qmg_comm_init_env")) return 2;
  if (!qmg::slab_begin()) return 2;
  const bool root = rank == 0;
  const int L = atoi(argv[1]);
  const double mass = atof(argv[2]);
  const char* gauge_file = argv[3];
  const int tile = atoi(argv[4]);
  int all = 0;
  const bool fits = (L % world == 0) && !((L / world) & 1) && L / world >= 2;
  qmg_comm_all_ok(fits, &all);
  if (!all) { if (root) std::cout << "[QMG-ERROR]: " << L << " rows do not split into " << world << " slabs of an even number of rows\n"; qmg::slab_end(); qmg_comm_finalize(); return 4; }
  const int y_loc = L / world;
  Lattice2D* lat_gauge = new Lattice2D(L, L, 1);
  complex<double>* gauge = allocate_vector<complex<double>>(lat_gauge->get_size_gauge());
  const bool got = (L == tile) ? read_gauge_u1(gauge, lat_gauge, gauge_file) : read_gauge_u1_tiled(gauge, lat_gauge, gauge_file, tile);
  qmg_comm_all_ok(got, &all);
  if (!all) { qmg::slab_end(); qmg_comm_finalize(); return 3; }
  Lattice2D* lat = new Lattice2D(L, y_loc, 1);
  const long n = lat->get_size_cv_l();
  complex<double>*b = allocate_vector<complex<double>>(n), *bp = allocate_vector<complex<double>>(n), *x = allocate_vector<complex<double>>(n), *Ax = allocate_vector<complex<double>>(n);
  gaussian_lattice(b, L, y_loc, 1, 1337);
  const double bnorm = sqrt(norm2sq(b, n));
  inversion_verbose_struct quiet(VERB_NONE, "");
  int good = 1;
  auto report = [&](const char* what, Stencil2D* op, const inversion_info& inv) {
    zero_vector(Ax, n);
    op->apply_M(Ax, x);
    const double res = sqrt(diffnorm2sq(b, Ax, n)) / bnorm, xn = norm2sq(x, n);
    if (!(inv.success && res < 1e-8)) good = 0;
    if (root) std::cout << std::setprecision(15) << "[QMG-SLAB]: " << what << " : world " << world << " ; iterations " << inv.iter << " ; true residual " << res << " ; |x|^2 " << xn << "\n";
  };
  {
    Staggered2D stag(lat, mass, gauge);
    zero_vector(x, n);
    inversion_info inv = minv_vector_bicgstab_l(x, b, (int)n, 4000, 1e-10, 6, apply_stencil_2D_M, (void*)&stag, &quiet);
    report("staggered, full operator, BiCGStab-6", &stag, inv);
    zero_vector(bp, n); zero_vector(x, n);
    stag.prepare_b(bp, b);
    inv = minv_vector_cg(x, bp, (int)(n / 2), 4000, 1e-10, apply_eo_staggered_2D_M, (void*)&stag, &quiet);
    stag.reconstruct_x(x, b);
    report("staggered, even-odd preconditioned CG + reconstruct", &stag, inv);
  }
  {
    GaugedLaplace2D lap(lat, mass * mass, gauge);
    zero_vector(bp, n); zero_vector(x, n);
    lap.prepare_b(bp, b);
    inversion_info inv = minv_vector_cg(x, bp, (int)(n / 2), 4000, 1e-10, apply_eo_gauge_laplace_2D_M, (void*)&lap, &quiet);
    lap.reconstruct_x(x, b);
    report("gauged Laplace, even-odd preconditioned CG + reconstruct", &lap, inv);
  }
  qmg_comm_all_ok(good, &all);
  deallocate_vector(&b); deallocate_vector(&bp); deallocate_vector(&x); deallocate_vector(&Ax); deallocate_vector(&gauge);
  delete lat; delete lat_gauge;
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
