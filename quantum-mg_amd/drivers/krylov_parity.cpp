// krylov_parity -- runs each Krylov driver of include/qmg/krylov.hpp (the device restatement of the absent quantum-linalg
// inverters) on ONE fixed system and dumps right-hand sides and solutions, so that
// tests/test_gpu_krylov.py can hold every driver to its CPU twin in the test infrastructure (the twins are pinned to scipy
// in tests/test_oracle_krylov.py): same iteration counts (+-1), same solutions.
//   ./krylov_parity L mass gauge_file dump_dir
// Systems: Wilson (nc = 2) at `mass` on the L x L gauge file for BiCGStab-L (L = 1, 6: n13:359 uses 6), Richardson (n22:289
// parameters), MR(0.85) (the K-cycle smoother), restarted GCR(8) and unrestarted GCR, CG on M^dagger M (the coarsest
// normal-equation solve, stateful_multigrid.h:915-969); gauged Laplace (nc = 1, m^2 = 0.01) for plain CG (n02 / n03).
// The last case drives CG from HOST vectors through apply_stencil_2D_host_thunk, the host-pointer compatibility thunk
// with the reference's exact matrix_op_cplx signature (stencil_2d.h:15-19): an unmodified CPU solver
// calling the GPU operator.
#include <cstdio>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#include "../include/qmg/qmg.hpp"
#include "driver_common.hpp"

using namespace std;

static void dump(const string& dir, const string& name, complex<double>* dev, size_t n) {
  vector<complex<double>> h = qmg::to_host(dev, n);
  FILE* f = fopen((dir + "/" + name + ".bin").c_str(), "wb");
  fwrite(h.data(), sizeof(complex<double>), n, f);
  fclose(f);
}
static void report(const char* name, const inversion_info& i, double bnorm) {
  cout << "[KRYLOV] " << name << " success " << (i.success ? 1 : 0) << " iter " << i.iter << " ops " << i.ops_count << " rel_res " << sqrt(i.resSq) / bnorm << "\n";
}

// a CPU solver that knows nothing about the device: textbook CG on host vectors through a matrix_op_cplx callback
static int host_cg(vector<complex<double>>& x, const vector<complex<double>>& b, int max_iter, double tol, matrix_op_cplx op, void* extra, double* rel_out) {
  const size_t n = b.size();
  vector<complex<double>> r(b), p(b), Ap(n);
  double rsq = 0.0, bsq = 0.0;
  for (size_t i = 0; i < n; i++) { x[i] = 0.0; bsq += norm(b[i]); }
  rsq = bsq;
  int k = 0;
  while (k < max_iter && sqrt(rsq) >= tol * sqrt(bsq)) {
    op(Ap.data(), p.data(), extra);
    complex<double> pAp = 0.0;
    for (size_t i = 0; i < n; i++) pAp += conj(p[i]) * Ap[i];
    const double alpha = rsq / pAp.real();
    double rn = 0.0;
    for (size_t i = 0; i < n; i++) { x[i] += alpha * p[i]; r[i] -= alpha * Ap[i]; rn += norm(r[i]); }
    const double beta = rn / rsq;
    rsq = rn;
    for (size_t i = 0; i < n; i++) p[i] = r[i] + beta * p[i];
    k++;
  }
  *rel_out = sqrt(rsq / bsq);
  return k;
}

int main(int argc, char** argv) {
  qmg_driver::Guard guard;
  if (argc < 5) { cout << "usage: ./krylov_parity L mass gauge_file dump_dir\n"; return -1; }
  cout << setprecision(17);
  if (!qmg::ok(qmg_init(0), "qmg_init")) return 2;
  const int L = stoi(argv[1]);
  const double mass = stod(argv[2]);
  const string gauge_file = argv[3], dir = argv[4];
  Lattice2D lat_g(L, L, 1), lat_w(L, L, 2);
  complex<double>* gauge = allocate_vector<complex<double>>(lat_g.get_size_gauge());
  if (!read_gauge_u1(gauge, &lat_g, gauge_file)) return 3;
  Wilson2D wilson(&lat_w, mass, gauge);
  wilson.build_dagger_stencil();
  GaugedLaplace2D laplace(&lat_g, 0.01, gauge);
  const int nw = (int)lat_w.get_size_cv_l(), nl = (int)lat_g.get_size_cv_l();

  complex<double>* b = allocate_vector<complex<double>>(nw);
  complex<double>* x = allocate_vector<complex<double>>(nw);
  gaussian(b, nw, 4242ull);
  dump(dir, "b_wilson", b, nw);
  const double bn = sqrt(norm2sq(b, nw));
  inversion_info inv;

  zero_vector(x, nw); inv = minv_vector_bicgstab_l(x, b, nw, 500, 1e-9, 1, apply_stencil_2D_M, (void*)&wilson); report("bicgstab1", inv, bn); dump(dir, "x_bicgstab1", x, nw);
  zero_vector(x, nw); inv = minv_vector_bicgstab_l(x, b, nw, 500, 5e-5, 6, apply_stencil_2D_M, (void*)&wilson); report("bicgstab6", inv, bn); dump(dir, "x_bicgstab6", x, nw);
  zero_vector(x, nw); inv = minv_vector_richardson(x, b, nw, 10, 1e-10, 0.33, 250, apply_stencil_2D_M, (void*)&wilson); report("richardson", inv, bn); dump(dir, "x_richardson", x, nw);
  zero_vector(x, nw); inv = minv_vector_minres(x, b, nw, 6, 1e-30, 0.85, apply_stencil_2D_M, (void*)&wilson); report("mr", inv, bn); dump(dir, "x_mr", x, nw);
  zero_vector(x, nw); inv = minv_vector_gcr_restart(x, b, nw, 400, 1e-9, 8, apply_stencil_2D_M, (void*)&wilson); report("gcr8", inv, bn); dump(dir, "x_gcr8", x, nw);
  zero_vector(x, nw); inv = minv_vector_gcr(x, b, nw, 60, 1e-9, apply_stencil_2D_M, (void*)&wilson); report("gcr", inv, bn); dump(dir, "x_gcr", x, nw);
  // CG on the normal operator, right-hand side M^dag b
  complex<double>* bnrm = allocate_vector<complex<double>>(nw);
  zero_vector(bnrm, nw);
  wilson.apply_M_dagger(bnrm, b);
  dump(dir, "b_normal", bnrm, nw);
  zero_vector(x, nw); inv = minv_vector_cg(x, bnrm, nw, 3000, 1e-10, apply_stencil_2D_M_dagger_M, (void*)&wilson); report("cg_normal", inv, sqrt(norm2sq(bnrm, nw))); dump(dir, "x_cg_normal", x, nw);

  // plain CG on the gauged Laplace operator
  complex<double>* bl = allocate_vector<complex<double>>(nl);
  complex<double>* xl = allocate_vector<complex<double>>(nl);
  gaussian(bl, nl, 4343ull);
  dump(dir, "b_laplace", bl, nl);
  zero_vector(xl, nl); inv = minv_vector_cg(xl, bl, nl, 2000, 1e-10, apply_stencil_2D_M, (void*)&laplace); report("cg_laplace", inv, sqrt(norm2sq(bl, nl))); dump(dir, "x_cg_laplace", xl, nl);

  // the same system from HOST vectors through the host-pointer thunk: a CPU CG that only knows matrix_op_cplx
  {
    vector<complex<double>> hb = qmg::to_host(bl, (size_t)nl), hx((size_t)nl);
    complex<double>* dl = allocate_vector<complex<double>>(nl);
    complex<double>* dr = allocate_vector<complex<double>>(nl);
    HostThunkData thunk = {&laplace, apply_stencil_2D_M, dl, dr};
    double rel = 0.0;
    const int it = host_cg(hx, hb, 2000, 1e-10, apply_stencil_2D_host_thunk, (void*)&thunk, &rel);
    cout << "[KRYLOV] host_thunk_cg success " << (rel < 1e-10 ? 1 : 0) << " iter " << it << " ops " << it << " rel_res " << rel << "\n";
    FILE* f = fopen((dir + "/x_host_thunk_cg.bin").c_str(), "wb");
    fwrite(hx.data(), sizeof(complex<double>), hx.size(), f);
    fclose(f);
    deallocate_vector(&dl); deallocate_vector(&dr);
  }
  deallocate_vector(&bl); deallocate_vector(&xl); deallocate_vector(&bnrm); deallocate_vector(&b); deallocate_vector(&x); deallocate_vector(&gauge);
  qmg::VecPool::release_all();
  return qmg_driver::leave(0);
}
