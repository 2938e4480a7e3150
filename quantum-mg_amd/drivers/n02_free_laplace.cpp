// n02_free_laplace -- the build's counterpart of tests/n02_free_laplace_test/free_laplace.cpp on the GPU:
// 32x24 free Laplace, m^2 = 0.01, point sources on an even and an odd site, then a CG inversion.
// Prints the same [QMG-TEST] lines (Self / +x / +y / -x / -y; expected 4.01 and -1; applied twice
// 20.0801, -8.02, 1).  Exit code 1 if a known answer is missed.
#include <cmath>
#include <iomanip>
#include <iostream>

#include "../include/qmg/qmg.hpp"
#include "driver_common.hpp"

using namespace std;

int main(int argc, char** argv) {
  qmg_driver::Guard guard;
  cout << setiosflags(ios::fixed) << setprecision(6);
  if (!qmg::ok(qmg_init(0), "qmg_init")) return 2;
  const int x_len = 32, y_len = 24, dof = 1;
  const double m_sq = 0.1 * 0.1;
  Lattice2D* lat = new Lattice2D(x_len, y_len, dof);
  FreeLaplace2D* lap_stencil = new FreeLaplace2D(lat, m_sq);
  const int cv_size = lat->get_size_cv();
  complex<double>* rhs = allocate_vector<complex<double>>(cv_size);
  complex<double>* lhs = allocate_vector<complex<double>>(cv_size);
  int bad = 0;
  auto check = [&](const char* label, complex<double> got, double want) {
    cout << "[QMG-TEST]: " << label << ": " << got << "\n";
    if (std::abs(got - want) > 1e-12) bad++;
  };
  auto at = [&](complex<double>* v, int x, int y) { return qmg::get_element(v, (size_t)lat->cv_coord_to_index((x + x_len) % x_len, (y + y_len) % y_len, 0)); };

  for (int odd = 0; odd < 2; odd++) {
    const int x0 = x_len / 2, y0 = y_len / 2 + odd;
    zero_vector(rhs, cv_size);
    zero_vector(lhs, cv_size);
    cout << "[QMG-TEST]: Test square laplace on " << (odd ? "odd" : "even") << " point.\n";
    qmg::set_element(rhs, (size_t)lat->cv_coord_to_index(x0, y0, 0), complex<double>(1.0));
    lap_stencil->apply_M(lhs, rhs);
    check("Self", at(lhs, x0, y0), 4.0 + m_sq);
    check("+x", at(lhs, x0 + 1, y0), -1.0);
    check("+y", at(lhs, x0, y0 + 1), -1.0);
    check("-x", at(lhs, x0 - 1, y0), -1.0);
    check("-y", at(lhs, x0, y0 - 1), -1.0);
    if (odd) {   // apply again (free_laplace.cpp:93-100)
      zero_vector(rhs, cv_size);
      lap_stencil->apply_M(rhs, lhs);
      check("Self", at(rhs, x0, y0), 20.0801);
      check("+x", at(rhs, x0 + 1, y0), -8.02);
      check("+2x", at(rhs, x0 + 2, y0), 1.0);
    }
  }

  cout << "[QMG-TEST]: Test a matrix inversion on an even point.\n";
  zero_vector(rhs, cv_size);
  zero_vector(lhs, cv_size);
  qmg::set_element(rhs, (size_t)lat->cv_coord_to_index(x_len / 2, y_len / 2, 0), complex<double>(1.0));
  const double rhs_norm = sqrt(norm2sq(rhs, cv_size));
  inversion_verbose_struct* verb = new inversion_verbose_struct(VERB_SUMMARY, std::string("[QMG-TEST-CG-INFO]: "));
  cout << resetiosflags(ios::fixed) << setiosflags(ios::scientific) << setprecision(6);
  inversion_info invif = minv_vector_cg(lhs, rhs, cv_size, 4000, 1e-7, apply_stencil_2D_M, (void*)lap_stencil, verb);
  cout << "[QMG-TEST]: " << (invif.success ? "Algorithm " : "Potential Error! Algorithm ") << invif.name << " took " << invif.iter
       << " iterations to reach a tolerance of " << sqrt(invif.resSq) / rhs_norm << "\n";
  // true residual
  complex<double>* check_v = allocate_vector<complex<double>>(cv_size);
  apply_stencil_2D_M(check_v, lhs, (void*)lap_stencil);
  const double true_res = sqrt(diffnorm2sq(rhs, check_v, cv_size)) / rhs_norm;
  cout << "[QMG-TEST]: Check tolerance " << true_res << "\n";
  if (!(invif.success && true_res < 2e-7)) bad++;

  deallocate_vector(&rhs); deallocate_vector(&lhs); deallocate_vector(&check_v);
  delete verb; delete lap_stencil; delete lat;
  qmg::VecPool::release_all();
  return qmg_driver::leave(bad ? 1 : 0);
}
