// facade_selftest -- device-side checks of the C++ facade against the identities the reference's own
// tests print (they store no numbers): every check recomputes a quantity two ways on the GPU
// through the facade's reference-named methods and compares.  Exit code = number of failed checks.
//   n00  cshift round trip                                   (tests/n00_cshift)
//   n04  staggered even-odd preconditioned solve             (tests/n04_staggered_test, staggered.h:190-240)
//   n03  gauged Laplace even-odd preconditioned solve        (tests/n03_gauge_laplace_test, gaugedlaplace.h:154-204)
//   n08  Galerkin: built coarse op == emulated R A P         (tests/n08_distance1_build_test:117-147, multigrid.h:465-512)
//   n17  <y, M x> = <M^dag y, x>, M^dag M / M M^dag          (tests/n17_dagger_stencil_test:85-96)
//   n18  right-block-Jacobi and Schur solves reconstruct the ORIGINAL system   (tests/n18_rbjacobi_stencil_test:153-231)
//   n21  rbj-dagger normal equations (CGNE / CGNR)           (tests/n21_rbj_dagger_stencil_test:138-209)
//   storage / lattice / multigrid bookkeeping
#include <cmath>
#include <iostream>
#include <string>

#include <unistd.h>

#include "../include/qmg/qmg.hpp"
#include "driver_common.hpp"

using namespace std;

static int failures = 0;
static void check(bool okv, const string& what, double val) {
  cout << (okv ? "[ OK ] " : "[FAIL] ") << what << " : " << val << "\n";
  if (!okv) failures++;
}
static double rel_resid(Stencil2D* op, complex<double>* x, complex<double>* b, long n) {   // ||b - A x|| / ||b|| with the ORIGINAL operator
  complex<double>* Ax = allocate_vector<complex<double>>(n);
  zero_vector(Ax, n);
  op->apply_M(Ax, x);
  const double r = sqrt(diffnorm2sq(b, Ax, n) / norm2sq(b, n));
  deallocate_vector(&Ax);
  return r;
}

int main(int argc, char** argv) {
  qmg_driver::Guard guard;
  const string gauge_file = (argc > 1) ? argv[1] : "../../tests/golden/l32t32b60_heatbath.dat";
  if (!qmg::ok(qmg_init(0), "qmg_init")) return 100;
  const int L = 32;
  Lattice2D lat1(L, L, 1), lat2(L, L, 2);
  complex<double>* gauge = allocate_vector<complex<double>>(lat1.get_size_gauge());
  if (!read_gauge_u1(gauge, &lat1, gauge_file)) return 101;
  const long n1 = lat1.get_size_cv_l(), n2 = lat2.get_size_cv_l();
  inversion_verbose_struct quiet(VERB_NONE, "");

  // ---- lattice / storage bookkeeping
  {
    int bad = 0;
    for (int x = 0; x < L; x++) for (int y = 0; y < L; y++) { int xx, yy; lat2.index_to_coord(lat2.coord_to_index(x, y), xx, yy); if (xx != x || yy != y) bad++; }
    check(bad == 0, "Lattice2D coord<->index round trip", bad);
    ArrayStorageMG<complex<double>> pool(n1, 2);
    complex<double>*a = pool.check_out(), *b = pool.check_out(), *c = pool.check_out();
    bool okp = pool.get_number_allocated() == 3 && pool.get_number_checked() == 3 && a != b && b != c;
    pool.check_in(b); pool.check_in(c);
    okp = okp && pool.get_number_checked() == 1 && pool.check_out() == b;
    pool.check_in(b); pool.check_in(a);
    pool.consolidate(1);
    check(okp && pool.get_number_allocated() >= 1 && pool.get_number_checked() == 0, "ArrayStorageMG check_out/check_in/consolidate", pool.get_number_allocated());
  }

  // ---- n00: cshift there and back is the identity
  {
    complex<double>*v = allocate_vector<complex<double>>(n2), *s = allocate_vector<complex<double>>(n2), *t = allocate_vector<complex<double>>(n2);
    gaussian(v, n2, 5);
    cshift(s, v, QMG_CSHIFT_FROM_XP1, QMG_EO_FROM_EVENODD, 2, &lat2);
    cshift(t, s, QMG_CSHIFT_FROM_XM1, QMG_EO_FROM_EVENODD, 2, &lat2);
    double d = diffnorm2sq(t, v, n2);
    cshift(s, v, QMG_CSHIFT_FROM_YM1, QMG_EO_FROM_EVENODD, 2, &lat2);
    cshift(t, s, QMG_CSHIFT_FROM_YP1, QMG_EO_FROM_EVENODD, 2, &lat2);
    d += diffnorm2sq(t, v, n2);
    check(d == 0.0, "n00 cshift round trips", d);
    deallocate_vector(&v); deallocate_vector(&s); deallocate_vector(&t);
  }

  // ---- n04 / n03: even-odd preconditioned CG for staggered and gauged Laplace
  {
    Staggered2D stag(&lat1, 0.1, gauge);
    complex<double>*b = allocate_vector<complex<double>>(n1), *bp = allocate_vector<complex<double>>(n1), *x = allocate_vector<complex<double>>(n1);
    gaussian(b, n1, 11);
    zero_vector(bp, n1); zero_vector(x, n1);
    stag.prepare_b(bp, b);
    inversion_info inv = minv_vector_cg(x, bp, (int)(n1 / 2), 4000, 1e-10, apply_eo_staggered_2D_M, (void*)&stag, &quiet);
    stag.reconstruct_x(x, b);
    check(inv.success && rel_resid(&stag, x, b, n1) < 1e-8, "n04 staggered eo-preconditioned CG solves the full system", rel_resid(&stag, x, b, n1));
    GaugedLaplace2D lap(&lat1, 0.01, gauge);
    zero_vector(bp, n1); zero_vector(x, n1);
    lap.prepare_b(bp, b);
    inv = minv_vector_cg(x, bp, (int)(n1 / 2), 4000, 1e-10, apply_eo_gauge_laplace_2D_M, (void*)&lap, &quiet);
    lap.reconstruct_x(x, b);
    check(inv.success && rel_resid(&lap, x, b, n1) < 1e-8, "n03 gauged-Laplace eo-preconditioned CG solves the full system", rel_resid(&lap, x, b, n1));
    deallocate_vector(&b); deallocate_vector(&bp); deallocate_vector(&x);
  }

  // ---- Wilson: dagger, normal equations, rbjacobi, Schur, rbj-dagger
  Wilson2D wilson(&lat2, complex<double>(0.05, 0.0), gauge);
  {
    complex<double>*x = allocate_vector<complex<double>>(n2), *y = allocate_vector<complex<double>>(n2), *t = allocate_vector<complex<double>>(n2),
                   *u = allocate_vector<complex<double>>(n2), *b = allocate_vector<complex<double>>(n2), *bp = allocate_vector<complex<double>>(n2);
    gaussian(x, n2, 21); gaussian(y, n2, 22); gaussian(b, n2, 23);
    wilson.build_dagger_stencil();
    zero_vector(t, n2); wilson.apply_M(t, x);
    zero_vector(u, n2); wilson.apply_M_dagger(u, y);
    const complex<double> lhs = dot(y, t, n2), rhs = dot(u, x, n2);
    check(abs(lhs - rhs) / abs(lhs) < 1e-12, "n17 <y, M x> = <M^dag y, x>", abs(lhs - rhs) / abs(lhs));
    // gamma5-hermiticity: M^dag = g5 M g5
    wilson.gamma5(t, y); zero_vector(bp, n2); wilson.apply_M(bp, t); wilson.gamma5(bp);
    check(sqrt(diffnorm2sq(bp, u, n2) / norm2sq(u, n2)) < 1e-13, "Wilson gamma5-hermiticity M^dag = g5 M g5", sqrt(diffnorm2sq(bp, u, n2) / norm2sq(u, n2)));
    // CGNR: M^dag M x = M^dag b through the type dispatch (prepare / apply function / reconstruct)
    zero_vector(bp, n2); wilson.prepare_M(bp, b, QMG_MATVEC_MDAGGER_M);
    zero_vector(t, n2);
    inversion_info inv = minv_vector_cg(t, bp, (int)n2, 4000, 1e-11, Stencil2D::get_apply_function(QMG_MATVEC_MDAGGER_M), (void*)&wilson, &quiet);
    zero_vector(u, n2); wilson.reconstruct_M(u, t, b, QMG_MATVEC_MDAGGER_M);
    check(inv.success && rel_resid(&wilson, u, b, n2) < 1e-8, "n17 CGNR (M^dag M) solves the original system", rel_resid(&wilson, u, b, n2));
    // CGNE: M M^dag y = b, x = M^dag y
    zero_vector(bp, n2); wilson.prepare_M(bp, b, QMG_MATVEC_M_MDAGGER);
    zero_vector(t, n2);
    inv = minv_vector_cg(t, bp, (int)n2, 4000, 1e-11, Stencil2D::get_apply_function(QMG_MATVEC_M_MDAGGER), (void*)&wilson, &quiet);
    zero_vector(u, n2); wilson.reconstruct_M(u, t, b, QMG_MATVEC_M_MDAGGER);
    check(inv.success && rel_resid(&wilson, u, b, n2) < 1e-8, "n17 CGNE (M M^dag) solves the original system", rel_resid(&wilson, u, b, n2));

    // n18: right block Jacobi  (A C^-1) y = b, x = C^-1 y
    wilson.build_rbjacobi_stencil();
    zero_vector(bp, n2); wilson.prepare_M(bp, b, QMG_MATVEC_RIGHT_JACOBI);
    zero_vector(t, n2);
    inv = minv_vector_gcr_restart(t, bp, (int)n2, 4000, 1e-10, 32, Stencil2D::get_apply_function(QMG_MATVEC_RIGHT_JACOBI), (void*)&wilson, &quiet);
    zero_vector(u, n2); wilson.reconstruct_M(u, t, b, QMG_MATVEC_RIGHT_JACOBI);
    check(inv.success && rel_resid(&wilson, u, b, n2) < 1e-8, "n18 right-block-Jacobi GCR reconstructs the original solution", rel_resid(&wilson, u, b, n2));
    // n18/n19: Schur system on the even half
    zero_vector(bp, n2); wilson.prepare_M(bp, b, QMG_MATVEC_RIGHT_SCHUR);
    zero_vector(t, n2);
    inv = minv_vector_gcr_restart(t, bp, (int)(n2 / 2), 4000, 1e-10, 32, Stencil2D::get_apply_function(QMG_MATVEC_RIGHT_SCHUR), (void*)&wilson, &quiet);
    zero_vector(u, n2); wilson.reconstruct_M(u, t, b, QMG_MATVEC_RIGHT_SCHUR);
    check(inv.success && rel_resid(&wilson, u, b, n2) < 1e-8, "n18 Schur (even-odd) GCR reconstructs the original solution", rel_resid(&wilson, u, b, n2));
    const int schur_iters = inv.iter;
    // rbjacobi apply == original apply composed with cinv
    zero_vector(t, n2); wilson.apply_M_rbjacobi_cinv(t, x);
    zero_vector(u, n2); wilson.apply_M(u, t);
    zero_vector(bp, n2); wilson.apply_M_rbjacobi(bp, x);
    check(sqrt(diffnorm2sq(bp, u, n2) / norm2sq(u, n2)) < 1e-12, "rbjacobi apply == M . C^-1", sqrt(diffnorm2sq(bp, u, n2) / norm2sq(u, n2)));

    // n21: rbj-dagger and the two right-Jacobi normal operators
    wilson.build_rbj_dagger_stencil();
    zero_vector(t, n2); wilson.apply_M_rbjacobi(t, x);
    zero_vector(u, n2); wilson.apply_M_rbj_dagger(u, y);
    const complex<double> l2 = dot(y, t, n2), r2 = dot(u, x, n2);
    check(abs(l2 - r2) / abs(l2) < 1e-12, "n21 <y, M_rbj x> = <M_rbj^dag y, x>", abs(l2 - r2) / abs(l2));
    zero_vector(bp, n2); wilson.prepare_M(bp, b, QMG_MATVEC_RBJ_MDAGGER_M);
    zero_vector(t, n2);
    inv = minv_vector_cg(t, bp, (int)n2, 4000, 1e-11, Stencil2D::get_apply_function(QMG_MATVEC_RBJ_MDAGGER_M), (void*)&wilson, &quiet);
    zero_vector(u, n2); wilson.reconstruct_M(u, t, b, QMG_MATVEC_RBJ_MDAGGER_M);
    check(inv.success && rel_resid(&wilson, u, b, n2) < 1e-8, "n21 rbj CGNR (M^dag M) reconstructs the original solution", rel_resid(&wilson, u, b, n2));
    zero_vector(bp, n2); wilson.prepare_M(bp, b, QMG_MATVEC_RBJ_M_MDAGGER);
    zero_vector(t, n2);
    inv = minv_vector_cg(t, bp, (int)n2, 4000, 1e-11, Stencil2D::get_apply_function(QMG_MATVEC_RBJ_M_MDAGGER), (void*)&wilson, &quiet);
    zero_vector(u, n2); wilson.reconstruct_M(u, t, b, QMG_MATVEC_RBJ_M_MDAGGER);
    check(inv.success && rel_resid(&wilson, u, b, n2) < 1e-8, "n21 rbj CGNE (M M^dag) reconstructs the original solution", rel_resid(&wilson, u, b, n2));
    cout << "       (Schur GCR iterations: " << schur_iters << ")\n";
    for (complex<double>** p : {&x, &y, &t, &u, &b, &bp}) deallocate_vector(p);
  }

  // ---- n08: Galerkin through MultigridMG: built coarse stencil vs the emulated level (R A P)
  {
    const int nvec = 4;
    Lattice2D clat(L / 4, L / 4, nvec);
    complex<double>** nv = new complex<double>*[nvec];
    for (int j = 0; j < nvec; j++) { nv[j] = allocate_vector<complex<double>>(n2); gaussian(nv[j], n2, 40 + j); }
    TransferMG transfer(&lat2, &clat, nv, true, false, QMG_DOUBLE_NONE);
    // P^dag P = 1 (n05)
    const long nc_ = clat.get_size_cv_l();
    complex<double>*vc = allocate_vector<complex<double>>(nc_), *vf = allocate_vector<complex<double>>(n2), *vc2 = allocate_vector<complex<double>>(nc_);
    gaussian(vc, nc_, 50); zero_vector(vf, n2); zero_vector(vc2, nc_);
    transfer.prolong_c2f(vc, vf); transfer.restrict_f2c(vf, vc2);
    check(sqrt(diffnorm2sq(vc, vc2, nc_) / norm2sq(vc, nc_)) < 1e-13, "n05 P^dag P = 1 on the coarse space", sqrt(diffnorm2sq(vc, vc2, nc_) / norm2sq(vc, nc_)));
    MultigridMG built(&lat2, &wilson), emulated(&lat2, &wilson);
    built.push_level(&clat, &transfer, true, false, MultigridMG::QMG_MULTIGRID_PRECOND_ORIGINAL, nv);
    emulated.push_level(&clat, &transfer, false, false, MultigridMG::QMG_MULTIGRID_PRECOND_ORIGINAL, (complex<double>**)0);
    complex<double>*o1 = allocate_vector<complex<double>>(nc_), *o2 = allocate_vector<complex<double>>(nc_);
    zero_vector(o1, nc_); zero_vector(o2, nc_);
    built.apply_stencil(o1, vc, 1);
    emulated.apply_stencil(o2, vc, 1);
    check(sqrt(diffnorm2sq(o1, o2, nc_) / norm2sq(o2, nc_)) < 1e-12, "n08 Galerkin: built coarse apply == emulated R A P", sqrt(diffnorm2sq(o1, o2, nc_) / norm2sq(o2, nc_)));
    check(built.get_num_levels() == 2 && built.get_global_null_vectors(0) != 0 && emulated.get_stencil(1) == 0, "MultigridMG level bookkeeping", built.get_num_levels());
    built.pop_level();
    check(built.get_num_levels() == 1, "MultigridMG pop_level", built.get_num_levels());
    for (complex<double>** p : {&vc, &vf, &vc2, &o1, &o2}) deallocate_vector(p);
    for (int j = 0; j < nvec; j++) deallocate_vector(&nv[j]);
    delete[] nv;
  }

  // ---- n01 / n14: U(1) utilities (u1/u1_utils.h): write_gauge_u1 -> read_gauge_u1 round trip in the reference's text format,
  //      plaquette / topology of the stored configuration, a non-compact heatbath step, coarse-shift bookkeeping
  {
    const string tmp_cfg = string(getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp") + "/qmg_selftest_cfg." + to_string((long)getpid()) + ".dat";
    write_gauge_u1(gauge, &lat1, tmp_cfg);
    complex<double>* g2 = allocate_vector<complex<double>>(lat1.get_size_gauge());
    const bool rd = read_gauge_u1(g2, &lat1, tmp_cfg);
    std::remove(tmp_cfg.c_str());
    check(rd && sqrt(diffnorm2sq(gauge, g2, lat1.get_size_gauge()) / norm2sq(gauge, lat1.get_size_gauge())) < 1e-15, "write_gauge_u1 -> read_gauge_u1 round trip",
          sqrt(diffnorm2sq(gauge, g2, lat1.get_size_gauge())));
    const complex<double> plaq = get_plaquette_u1(gauge, &lat1);
    const double topo = get_topo_u1(gauge, &lat1);
    check(plaq.real() > 0.85 && plaq.real() < 0.97 && fabs(topo - std::round(topo)) < 1e-9, "plaquette of the beta = 6.0 fixture in (0.85, 0.97), integer topological charge", plaq.real());
    double* ph = allocate_vector<double>(lat1.get_size_gauge());
    qmg::ok(qmg_u1_gauge_to_phase(ph, gauge, (size_t)lat1.get_size_gauge(), 0), "qmg_u1_gauge_to_phase");
    const double s0 = get_noncompact_action_u1(ph, 6.0, &lat1);
    HeatbathRng rng(99ull);
    heatbath_noncompact_update(ph, &lat1, 6.0, 50, rng);
    polar_vector(ph, g2, (size_t)lat1.get_size_gauge());
    const double p1 = get_plaquette_u1(g2, &lat1).real(), s1 = get_noncompact_action_u1(ph, 6.0, &lat1) / (L * L);
    check(rng.sweeps_done == 50 && p1 > 0.88 && p1 < 0.96 && s1 > 0.4 && s1 < 0.6 && s0 > 0.0, "heatbath at beta = 6.0 keeps the plaquette near exp(-1/12), action per plaquette near 1/2", p1);
    deallocate_vector(&ph); deallocate_vector(&g2);
  }
  // ---- coarse shift after a variant swap (deliberate deviation from coarse.h:131, which leaves shift_backup at 0)
  {
    const int nvec = 8;
    Lattice2D latc(L / 4, L / 4, nvec);
    Wilson2D wm(&lat2, complex<double>(0.13, 0.0), gauge);
    complex<double>** nv = new complex<double>*[nvec];
    for (int j = 0; j < nvec; j++) { nv[j] = allocate_vector<complex<double>>(n2); gaussian(nv[j], n2, 500ull + j); }
    TransferMG tr(&lat2, &latc, nv, true, false, QMG_DOUBLE_NONE);
    CoarseOperator2D co(&latc, &wm, &lat2, &tr, false, false, CoarseOperator2D::QMG_COARSE_BUILD_DAGGER);
    const long ncv = latc.get_size_cv_l();
    complex<double>*v = allocate_vector<complex<double>>(ncv), *a1 = allocate_vector<complex<double>>(ncv), *a2 = allocate_vector<complex<double>>(ncv);
    gaussian(v, ncv, 600ull);
    zero_vector(a1, ncv); co.apply_M(a1, v);
    zero_vector(a2, ncv); co.apply_M_dagger(a2, v);   // swaps the dagger stencil in and out again
    zero_vector(a2, ncv); co.apply_M(a2, v);
    check(co.get_shift() == complex<double>(0.13, 0.0) && diffnorm2sq(a1, a2, ncv) == 0.0, "coarse operator keeps its (non-zero) shift across a dagger swap", co.get_shift().real());
    // the complex<float> copies of enable_f32_matrices mirror the ORIGINAL arrays: while the dagger stencil is swapped in they must not be applied
    zero_vector(a1, ncv); co.apply_M_dagger(a1, v);
    const bool f32on = co.enable_f32_matrices();
    zero_vector(a2, ncv); co.apply_M_dagger(a2, v);
    check(f32on && diffnorm2sq(a1, a2, ncv) == 0.0, "apply_M_dagger is untouched by fp32-stored ORIGINAL matrices", sqrt(diffnorm2sq(a1, a2, ncv) / norm2sq(a1, ncv)));
    zero_vector(a1, ncv); co.apply_M(a1, v);
    co.disable_f32_matrices();
    zero_vector(a2, ncv); co.apply_M(a2, v);
    const double d32 = sqrt(diffnorm2sq(a1, a2, ncv) / norm2sq(a2, ncv));
    check(d32 > 0.0 && d32 < 1e-6, "apply_M with fp32-stored matrices is the fp64 apply to fp32 rounding", d32);
    // complex<half> storage (opt-in): nc = 8 qualifies (a multiple of 4, entries inside half range); the apply is the fp64 apply to half rounding
    const bool f16on = co.enable_f32_matrices(16);
    zero_vector(a1, ncv); co.apply_M(a1, v);
    const double d16 = sqrt(diffnorm2sq(a1, a2, ncv) / norm2sq(a2, ncv));
    check(f16on && co.f32_bits == 16 && d16 > 1e-6 && d16 < 2e-3, "apply_M with 16-bit-stored matrices is the fp64 apply to half rounding", d16);
    co.disable_f32_matrices();
    for (complex<double>** p : {&v, &a1, &a2}) deallocate_vector(p);
    for (int j = 0; j < nvec; j++) deallocate_vector(&nv[j]);
    delete[] nv;
  }

  // ---- the batch engine's operators by type (include/qmg/batch.hpp: apply_stencil_typed_batch / prepare_M_batch / reconstruct_M_batch) against the
  // single-vector type dispatch (Stencil2D::apply_M / prepare_M / reconstruct_M, stencil_2d.h:2418-2527): all nine types, three systems, on the Wilson
  // operator (nc = 2 kernels) and on a Galerkin operator with every variant built (nc = 8 kernels)
  {
    const int nvec = 8, nb = 3;
    Lattice2D latc(L / 4, L / 4, nvec);
    complex<double>** nv = new complex<double>*[nvec];
    for (int j = 0; j < nvec; j++) { nv[j] = allocate_vector<complex<double>>(n2); gaussian(nv[j], n2, 900ull + j); }
    TransferMG tr(&lat2, &latc, nv, true, false, QMG_DOUBLE_NONE);
    CoarseOperator2D co(&latc, &wilson, &lat2, &tr, false, false, CoarseOperator2D::QMG_COARSE_BUILD_ALL);
    co.disable_f32_matrices();
    const QMGStencilType types[9] = {QMG_MATVEC_ORIGINAL, QMG_MATVEC_DAGGER, QMG_MATVEC_RIGHT_JACOBI, QMG_MATVEC_RIGHT_SCHUR, QMG_MATVEC_M_MDAGGER,
                                     QMG_MATVEC_MDAGGER_M, QMG_MATVEC_RBJ_DAGGER, QMG_MATVEC_RBJ_M_MDAGGER, QMG_MATVEC_RBJ_MDAGGER_M};
    Stencil2D* ops[2] = {&wilson, &co};
    for (int o = 0; o < 2; o++) {
      Stencil2D* st = ops[o];
      const size_t n = (size_t)st->get_lattice()->get_size_cv_l();
      qmg::BatchPool pool(n, nb);
      qmg::Batch in = pool.get(), rhs2 = pool.get(), out = pool.get(), ref = pool.get();
      const unsigned all = qmg::full_mask(nb);
      for (int k = 0; k < nb; k++) { gaussian(in.vec(k), (long)n, 950ull + 10 * o + k); gaussian(rhs2.vec(k), (long)n, 980ull + 10 * o + k); }
      double worst_apply = 0.0, worst_prep = 0.0, worst_rec = 0.0;
      for (int t = 0; t < 9; t++) {
        const size_t nsolve = (types[t] == QMG_MATVEC_RIGHT_SCHUR) ? n / 2 : n;
        BatchOp bop(st, types[t]);
        qmg::bzero(out, n, all); qmg::bzero(ref, n, all);
        apply_stencil_typed_batch<double>(out, in, all, (void*)&bop);
        for (int k = 0; k < nb; k++) Stencil2D::get_apply_function(types[t])(ref.vec(k), in.vec(k), (void*)st);
        for (int k = 0; k < nb; k++) worst_apply = std::max(worst_apply, sqrt(diffnorm2sq(out.vec(k), ref.vec(k), (long)nsolve) / norm2sq(ref.vec(k), (long)nsolve)));
        qmg::bzero(out, n, all); qmg::bzero(ref, n, all);
        prepare_M_batch<double>(st, types[t], out, in, all);
        for (int k = 0; k < nb; k++) st->prepare_M(ref.vec(k), in.vec(k), types[t]);
        for (int k = 0; k < nb; k++) worst_prep = std::max(worst_prep, sqrt(diffnorm2sq(out.vec(k), ref.vec(k), (long)n) / norm2sq(ref.vec(k), (long)n)));
        qmg::bzero(out, n, all); qmg::bzero(ref, n, all);
        reconstruct_M_batch<double>(st, types[t], out, in, rhs2, all);
        for (int k = 0; k < nb; k++) st->reconstruct_M(ref.vec(k), in.vec(k), rhs2.vec(k), types[t]);
        for (int k = 0; k < nb; k++) worst_rec = std::max(worst_rec, sqrt(diffnorm2sq(out.vec(k), ref.vec(k), (long)n) / norm2sq(ref.vec(k), (long)n)));
      }
      check(worst_apply < 1e-13, o == 0 ? "batch applies by type == Stencil2D::apply_M by type (Wilson, 9 types x 3 systems)" : "batch applies by type == apply_M by type (Galerkin nc = 8, 9 types x 3 systems)", worst_apply);
      check(worst_prep < 1e-13, o == 0 ? "prepare_M_batch == prepare_M (Wilson)" : "prepare_M_batch == prepare_M (Galerkin nc = 8)", worst_prep);
      check(worst_rec < 1e-13, o == 0 ? "reconstruct_M_batch == reconstruct_M (Wilson)" : "reconstruct_M_batch == reconstruct_M (Galerkin nc = 8)", worst_rec);
      // bcg_core (the coarsest solve of a normal-equation hierarchy) against minv_vector_cg / minv_vector_cg_restart, system by system: same iteration
      // counts, the same solutions (element-wise kernels and reductions are the single-vector ones; an apply of several systems may sum in another order)
      if (o == 1) {
        for (int rf = -1; rf <= 16; rf += 17) {   // one cycle, and restarts every 16 iterations
          BatchOp nop(st, QMG_MATVEC_MDAGGER_M);
          qmg::bzero(out, n, all);
          const std::vector<inversion_info> bi = bcg_core<double>(out, in, (int)n, 400, 1e-9, rf, apply_stencil_typed_batch<double>, (void*)&nop, all, true, (inversion_verbose_struct*)0,
                                                                  rf == -1 ? "CG" : "CG-restart");
          int worst_it = 0;
          double worst_x = 0.0;
          bool all_ok = true;
          for (int k = 0; k < nb; k++) {
            zero_vector(ref.vec(k), (long)n);
            inversion_info si = (rf == -1) ? minv_vector_cg(ref.vec(k), in.vec(k), (int)n, 400, 1e-9, Stencil2D::get_apply_function(QMG_MATVEC_MDAGGER_M), (void*)st, &quiet)
                                           : minv_vector_cg_restart(ref.vec(k), in.vec(k), (int)n, 400, 1e-9, rf, Stencil2D::get_apply_function(QMG_MATVEC_MDAGGER_M), (void*)st, &quiet);
            all_ok = all_ok && si.success && bi[k].success;
            worst_it = std::max(worst_it, std::abs(si.iter - bi[k].iter));
            worst_x = std::max(worst_x, sqrt(diffnorm2sq(out.vec(k), ref.vec(k), (long)n) / norm2sq(ref.vec(k), (long)n)));
          }
          check(all_ok && worst_it <= 1 && worst_x < 1e-7, rf == -1 ? "bcg_core == minv_vector_cg per system (M^dag M, Galerkin nc = 8, 3 systems)" : "bcg_core == minv_vector_cg_restart(16) per system", worst_x);
        }
      }
      // a normal operator with CoarsestSolveMG::normal_shift (shift_function, stateful_multigrid.h:724-729)
      BatchOp sh(st, QMG_MATVEC_RBJ_MDAGGER_M);
      sh.normal_shift = complex<double>(0.37, 0.0); sh.shift_length = n;
      apply_stencil_typed_batch<double>(out, in, all, (void*)&sh);
      Stencil2D::get_apply_function(QMG_MATVEC_RBJ_MDAGGER_M)(ref.vec(1), in.vec(1), (void*)st);
      caxpy(complex<double>(0.37, 0.0), in.vec(1), ref.vec(1), (long)n);
      const double dsh = sqrt(diffnorm2sq(out.vec(1), ref.vec(1), (long)n) / norm2sq(ref.vec(1), (long)n));
      check(dsh < 1e-13, o == 0 ? "shifted normal operator (Wilson)" : "shifted normal operator (Galerkin nc = 8)", dsh);
    }
    for (int j = 0; j < nvec; j++) deallocate_vector(&nv[j]);
    delete[] nv;
  }

  // ---- clear_stencils / prune_stencils on an operator that applies straight from the links: the cached link copy must
  // not outlive the stored arrays.  Reference semantics (stencil_2d.h:339-404): after clear_stencils the matrices are zero, so
  // apply_M adds shift * rhs and nothing else; after pruning the hopping term only clover + shift act.
  {
    const double m = 0.21;
    complex<double>*v = allocate_vector<complex<double>>(n2), *a = allocate_vector<complex<double>>(n2), *w = allocate_vector<complex<double>>(n2);
    gaussian(v, n2, 77);
    {
      Wilson2D wil(&lat2, m, gauge);
      wil.clear_stencils();
      zero_vector(a, n2); wil.apply_M(a, v);
      caxy(m, v, w, n2);
      check(diffnorm2sq(a, w, n2) <= 1e-28 * norm2sq(w, n2), "after clear_stencils apply_M is the shift term alone", sqrt(diffnorm2sq(a, w, n2) / norm2sq(w, n2)));
      zero_vector(a, n2); apply_stencil_2D_M(a, v, (void*)&wil);
      check(diffnorm2sq(a, w, n2) <= 1e-28 * norm2sq(w, n2), "after clear_stencils apply_stencil_2D_M is the shift term alone", sqrt(diffnorm2sq(a, w, n2) / norm2sq(w, n2)));
    }
    {
      Wilson2D wil(&lat2, m, gauge);
      wil.prune_stencils(QMG_PIECE_HOPPING);
      zero_vector(a, n2); wil.apply_M(a, v);
      caxy(2.0 + m, v, w, n2);   // Wilson clover = 2 w 1 (wilson.h:167-170), w = 1
      check(diffnorm2sq(a, w, n2) <= 1e-28 * norm2sq(w, n2), "after pruning the hopping term apply_M is clover + shift", sqrt(diffnorm2sq(a, w, n2) / norm2sq(w, n2)));
    }
    for (complex<double>** p : {&v, &a, &w}) deallocate_vector(p);
  }

  deallocate_vector(&gauge);
  qmg::VecPool::release_all();
  cout << (failures ? "[SELFTEST FAILED] " : "[SELFTEST PASSED] ") << failures << " failure(s)\n";
  return qmg_driver::leave(failures);
}
