// n20_staggered_goldstone_u1_heatbath -- the build's counterpart of tests/n20_staggered_goldstone_u1_heatbath/staggered_u1.cpp
// on the GPU: quenched non-compact U(1) heatbath, ONE staggered propagator per configuration from a point source at the
// origin (BiCGStab-6, tol 1e-10, n20:45-48,131), Goldstone-pion correlator C(t) = sum_x |S(x,t)|^2 through the per-timeslice
// reduction (reductions/reductions.h:24-50 -> qmg_norm2sq_cv_timeslice), folded and accumulated over configurations.
//   ./n20_staggered_goldstone_u1_heatbath L mass beta n_meas [n_update n_therm seed]
// The reference hard-codes L = 32, mass 0.04, beta 6.0, n_update 100, n_therm 1000 (n20:36-55); its stored results
// (critical_mass.txt: m = 0.1 ... 0.04 at 32^2, beta = 6.0) are what tests/test_gpu_u1.py holds this driver to.
// Same output blocks ([QMG-GAUGE-FINAL], [QMG-BEGIN-PION] ..., [QMG-BEGIN-PION-EFFMASS] ...).  Differences: the
// heatbath is the four-colour parallel one of csrc/qmg_u1.hip (same ensemble, other random stream); the initial guess is the
// zero vector (the reference draws a Gaussian one, n20:86).
#include <cmath>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#include "../include/qmg/qmg.hpp"
#include "driver_common.hpp"

using namespace std;

int main(int argc, char** argv) {
  qmg_driver::Guard guard;
  if (argc < 5) { cout << "usage: ./n20_staggered_goldstone_u1_heatbath L mass beta n_meas [n_update n_therm seed]\n"; return -1; }
  if (!qmg::ok(qmg_init(getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : 0), "qmg_init")) return 2;
  const int x_len = stoi(argv[1]), y_len = x_len;
  const double mass = stod(argv[2]), beta = stod(argv[3]);
  const int n_meas = stoi(argv[4]);
  const int n_update = (argc > 5) ? stoi(argv[5]) : 100;
  const int n_therm = (argc > 6) ? stoi(argv[6]) : 1000;
  HeatbathRng generator((argc > 7) ? stoull(argv[7]) : 1337ull);
  const int dof = Staggered2D::get_dof();
  const int max_iter = 4000, bicgstab_l = 6;
  const double tol = 1e-10;
  const bool quiet = getenv("QMG_QUIET") != 0;

  Lattice2D* lat = new Lattice2D(x_len, y_len, dof);
  const int cv_size = lat->get_size_cv();
  Lattice2D* lat_gauge = new Lattice2D(x_len, y_len, 1);
  complex<double>* gauge_field = allocate_vector<complex<double>>(lat_gauge->get_size_gauge());
  double* phases = allocate_vector<double>(lat_gauge->get_size_gauge());
  qmg::ok(qmg_memset_zero(phases, sizeof(double) * (size_t)lat_gauge->get_size_gauge(), qmg::current_stream()), "qmg_memset_zero");   // unit field
  polar_vector(phases, gauge_field, (size_t)lat_gauge->get_size_gauge());
  Staggered2D* staggered = new Staggered2D(lat, mass, gauge_field);

  complex<double>* src = allocate_vector<complex<double>>(cv_size);
  complex<double>* prop = allocate_vector<complex<double>>(cv_size);
  double plaq = 0.0, plaq_sq = 0.0;
  int count = 0, unconverged = 0;
  vector<double> pion(y_len, 0.0), pion_sq(y_len, 0.0), pion_tmp(y_len);
  inversion_verbose_struct verb(VERB_NONE, "[QMG-STAGGERED-INFO]: ");

  cout << setiosflags(ios::fixed) << setprecision(6);
  int i = 0;
  cout << "[QMG-GAUGE]: " << i << " " << get_plaquette_u1(gauge_field, lat_gauge) << " " << get_topo_u1(gauge_field, lat_gauge) << "\n";
  const int n_max = n_therm + n_update * (n_meas + 1);
  for (i = n_update; i < n_max; i += n_update) {
    heatbath_noncompact_update(phases, lat_gauge, beta, n_update, generator);
    polar_vector(phases, gauge_field, (size_t)lat_gauge->get_size_gauge());
    const double plaq_tmp = std::real(get_plaquette_u1(gauge_field, lat_gauge));
    if (!quiet) cout << i << " " << plaq_tmp << " " << get_topo_u1(gauge_field, lat_gauge) << "\n";
    if (i > n_therm) {
      plaq += plaq_tmp;
      plaq_sq += plaq_tmp * plaq_tmp;
      staggered->update_links(gauge_field);
      zero_vector(src, cv_size);
      qmg::set_element(src, (size_t)lat->cv_coord_to_index(0, 0, 0), complex<double>(1.0, 0.0));
      zero_vector(prop, cv_size);
      inversion_info invif = minv_vector_bicgstab_l(prop, src, cv_size, max_iter, tol, bicgstab_l, apply_stencil_2D_M, (void*)staggered, &verb);
      if (!invif.success) unconverged++;
      norm2sq_cv_timeslice(pion_tmp.data(), prop, lat);   // reductions/reductions.h:24-41
      for (int j = 1; j < y_len / 2; j++) { const double tmp = 0.5 * (pion_tmp[j] + pion_tmp[y_len - j]); pion_tmp[j] = pion_tmp[y_len - j] = tmp; }   // fold
      for (int j = 0; j < y_len; j++) { pion[j] += pion_tmp[j]; pion_sq[j] += pion_tmp[j] * pion_tmp[j]; }
      count++;
    }
  }
  cout << "[QMG-GAUGE-FINAL]: The plaquette is " << plaq / count << " +/- " << sqrt((plaq_sq / count - plaq * plaq / ((double)count * count)) / count) << "\n";
  cout << "[QMG-INFO]: " << count << " measurements, " << unconverged << " unconverged inversions, non-compact action per plaquette "
       << get_noncompact_action_u1(phases, beta, lat_gauge) / ((double)x_len * y_len) << " (equipartition: 0.5)\n";
  cout << setprecision(10);
  cout << "[QMG-BEGIN-PION]\n";
  for (int j = 0; j < y_len; j++)
    cout << j << " " << pion[j] / count << " +/- " << sqrt(fabs(pion_sq[j] / count - pion[j] * pion[j] / ((double)count * count)) / count) << "\n";
  cout << "[QMG-END-PION]\n";
  cout << "[QMG-BEGIN-PION-EFFMASS]\n";
  for (int j = 1; j < y_len - 1; j++) cout << j << " " << std::acosh((pion[j + 1] + pion[j - 1]) / (2.0 * pion[j])) << "\n";
  cout << "[QMG-END-PION-EFFMASS]\n";

  deallocate_vector(&src); deallocate_vector(&prop); deallocate_vector(&phases); deallocate_vector(&gauge_field);
  delete staggered; delete lat_gauge; delete lat;
  qmg::VecPool::release_all();
  return qmg_driver::leave(unconverged == 0 ? 0 : 1);
}
