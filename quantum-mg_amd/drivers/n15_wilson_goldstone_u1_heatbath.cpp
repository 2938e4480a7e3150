// n15_wilson_goldstone_u1_heatbath -- the build's counterpart of tests/n15_wilson_goldstone_u1_heatbath/wilson_u1.cpp on the
// GPU: quenched non-compact U(1) heatbath, Wilson propagators from a point source for both spin components
// (BiCGStab-6, tol 1e-10), pion correlator C(t) = sum_x |S(x,t)|^2 by the per-timeslice reduction
// (reductions/reductions.h:24-50 -> qmg_norm2sq_cv_timeslice), folded and accumulated over configurations.
//   ./n15_wilson_goldstone_u1_heatbath L mass beta n_meas [n_update n_therm seed [out_cfg]]
// The reference hard-codes L = 64, mass -0.07, beta 6.0, n_update 100, n_therm 1000, n_max 100000 (n15:38-58); its stored
// results (critical_mass.txt) are for 32^2, beta = 6.0.  Same output lines: "[QMG-GAUGE]: ...", "<i> <plaq> <topo>",
// "[QMG-GAUGE-FINAL]: The plaquette is ...", [QMG-BEGIN-PION] ... [QMG-END-PION], [QMG-BEGIN-PION-EFFMASS] ...
// Differences: the heatbath is the four-colour parallel one of csrc/qmg_u1.hip (same ensemble, other random
// stream than std::mt19937 + std::normal_distribution); initial guesses are zero vectors.  `out_cfg` writes the last
// configuration with write_gauge_u1 (the reference's text format).
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
  if (argc < 5) { cout << "usage: ./n15_wilson_goldstone_u1_heatbath L mass beta n_meas [n_update n_therm seed [out_cfg]]\n"; return -1; }
  if (!qmg::ok(qmg_init(getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : 0), "qmg_init")) return 2;
  const int x_len = stoi(argv[1]), y_len = x_len;
  const double mass = stod(argv[2]), beta = stod(argv[3]);
  const int n_meas = stoi(argv[4]);
  const int n_update = (argc > 5) ? stoi(argv[5]) : 100;
  const int n_therm = (argc > 6) ? stoi(argv[6]) : 1000;
  HeatbathRng generator((argc > 7) ? stoull(argv[7]) : 1337ull);
  const string out_cfg = (argc > 8) ? argv[8] : "";
  const int dof = Wilson2D::get_dof();
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
  Wilson2D* wilson = new Wilson2D(lat, mass, gauge_field);

  complex<double>* src = allocate_vector<complex<double>>(cv_size);
  complex<double>* prop = allocate_vector<complex<double>>(cv_size);
  double plaq = 0.0, plaq_sq = 0.0;
  int count = 0, unconverged = 0;
  vector<double> pion(y_len, 0.0), pion_sq(y_len, 0.0), pion_up(y_len), pion_down(y_len);
  inversion_verbose_struct verb(VERB_NONE, "[QMG-WILSON-INFO]: ");

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
      wilson->update_links(gauge_field);
      for (int spin = 0; spin < 2; spin++) {   // one inversion per spin component of the point source (n15:134-168)
        zero_vector(src, cv_size);
        qmg::set_element(src, (size_t)lat->cv_coord_to_index(0, 0, spin), complex<double>(1.0, 0.0));
        zero_vector(prop, cv_size);
        inversion_info invif = minv_vector_bicgstab_l(prop, src, cv_size, max_iter, tol, bicgstab_l, apply_stencil_2D_M, (void*)wilson, &verb);
        if (!invif.success) unconverged++;
        vector<double>& p = spin ? pion_down : pion_up;
        norm2sq_cv_timeslice(p.data(), prop, lat);   // reductions/reductions.h:24-41
        for (int j = 1; j < y_len / 2; j++) { const double tmp = 0.5 * (p[j] + p[y_len - j]); p[j] = p[y_len - j] = tmp; }   // fold
      }
      for (int j = 0; j < y_len; j++) {
        pion[j] += pion_up[j] + pion_down[j];
        pion_sq[j] += (pion_up[j] + pion_down[j]) * (pion_up[j] + pion_down[j]);
      }
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
  if (!out_cfg.empty()) write_gauge_u1(gauge_field, lat_gauge, out_cfg);

  deallocate_vector(&src); deallocate_vector(&prop); deallocate_vector(&phases); deallocate_vector(&gauge_field);
  delete wilson; delete lat_gauge; delete lat;
  qmg::VecPool::release_all();
  return qmg_driver::leave(unconverged == 0 ? 0 : 1);
}
