// u1.hpp -- U(1) gauge utilities on device fields (reference: u1/u1_utils.h): text I/O in the reference's format
// (read_gauge_u1 :38-67, write_gauge_u1 :105-168), unit field (:172-181), polar_vector, non-compact heatbath (:607-757; here
// a four-colour parallel heatbath on the device, csrc/qmg_u1.hip), plaquette / topology / non-compact action (:386-508).
// Smearing, gauge transforms and instantons stay out of scope (input preparation no BASELINE config uses).
// `gauge_field` is a DEVICE nc=1 LatticeGauge (mu, eo, y, x) of complex links; `phases` a DEVICE double field in the same order.
#ifndef QMG_U1_HPP
#define QMG_U1_HPP

#include <cstdio>
#include <string>
#include <vector>

#include "lattice2d.hpp"
#include "qmg_device.hpp"

// One phase per line; loop order x outer, y, mu inner (u1_utils.h:53-63).
inline bool read_gauge_u1(complex<double>* gauge_field, Lattice2D* lat, std::string input_file) {
  if (lat->get_nc() != 1) { std::cout << "[QMG-ERROR]: U1 gauge functions require Nc = 1 lattice.\n"; return false; }
  const int x_len = lat->get_dim_mu(0), y_len = lat->get_dim_mu(1);
  std::FILE* f = std::fopen(input_file.c_str(), "r");
  if (!f) { std::cout << "[QMG-ERROR]: cannot open gauge file " << input_file << "\n"; return false; }
  std::vector<complex<double>> host((size_t)lat->get_size_gauge());
  bool good = true;
  for (int x = 0; x < x_len && good; x++)
    for (int y = 0; y < y_len && good; y++)
      for (int mu = 0; mu < 2; mu++) {
        double phase;
        if (std::fscanf(f, "%lf", &phase) != 1) { good = false; break; }
        host[lat->gauge_coord_to_index(x, y, 0, 0, mu)] = std::polar(1.0, phase);
      }
  std::fclose(f);
  if (!good) { std::cout << "[QMG-ERROR]: gauge file " << input_file << " is too short for this lattice.\n"; return false; }
  qmg::upload(gauge_field, host.data(), host.size());
  return true;
}

// Periodic tiling of a small (t_len x t_len) configuration file onto a larger lattice: the same U(1)
// config as an L x L field (valid because the configuration is periodic).  Not in the reference; used by the
// benchmark drivers to reach 2048^2 / 4096^2 from the committed 64^2 fixture.
inline bool read_gauge_u1_tiled(complex<double>* gauge_field, Lattice2D* lat, std::string input_file, int t_len) {
  if (lat->get_nc() != 1) { std::cout << "[QMG-ERROR]: U1 gauge functions require Nc = 1 lattice.\n"; return false; }
  const int x_len = lat->get_dim_mu(0), y_len = lat->get_dim_mu(1);
  if (x_len % t_len || y_len % t_len) { std::cout << "[QMG-ERROR]: lattice is not a multiple of the tile.\n"; return false; }
  std::FILE* f = std::fopen(input_file.c_str(), "r");
  if (!f) { std::cout << "[QMG-ERROR]: cannot open gauge file " << input_file << "\n"; return false; }
  std::vector<double> ph((size_t)2 * t_len * t_len);
  for (size_t k = 0; k < ph.size(); k++)
    if (std::fscanf(f, "%lf", &ph[k]) != 1) { std::fclose(f); std::cout << "[QMG-ERROR]: gauge file too short.\n"; return false; }
  std::fclose(f);
  std::vector<complex<double>> host((size_t)lat->get_size_gauge());
  for (int x = 0; x < x_len; x++)
    for (int y = 0; y < y_len; y++)
      for (int mu = 0; mu < 2; mu++)
        host[lat->gauge_coord_to_index(x, y, 0, 0, mu)] = std::polar(1.0, ph[((size_t)(x % t_len) * t_len + (y % t_len)) * 2 + mu]);
  qmg::upload(gauge_field, host.data(), host.size());
  return true;
}

inline void unit_gauge_u1(complex<double>* gauge_field, Lattice2D* lat) {   // u1_utils.h:172-181
  if (lat->get_nc() != 1) { std::cout << "[QMG-ERROR]: U1 gauge functions require Nc = 1 lattice.\n"; return; }
  std::vector<complex<double>> host((size_t)lat->get_size_gauge(), complex<double>(1.0, 0.0));
  qmg::upload(gauge_field, host.data(), host.size());
}


// write_gauge_u1 (u1_utils.h:105-135): one phase arg(U) per line, fixed notation with 20 digits, loop order x outer, y, mu inner
inline void write_gauge_u1(complex<double>* gauge_field, Lattice2D* lat, std::string output_file) {
  if (lat->get_nc() != 1) { std::cout << "[QMG-ERROR]: U1 gauge functions require Nc = 1 lattice.\n"; return; }
  const int x_len = lat->get_dim_mu(0), y_len = lat->get_dim_mu(1);
  std::vector<complex<double>> host = qmg::to_host(gauge_field, (size_t)lat->get_size_gauge());
  std::FILE* f = std::fopen(output_file.c_str(), "w");
  if (!f) { std::cout << "[QMG-ERROR]: cannot open " << output_file << " for writing\n"; return; }
  for (int x = 0; x < x_len; x++)
    for (int y = 0; y < y_len; y++)
      for (int mu = 0; mu < 2; mu++) std::fprintf(f, "%.20f\n", std::arg(host[lat->gauge_coord_to_index(x, y, 0, 0, mu)]));
  std::fclose(f);
}
// the phase-field overload (:138-168): the non-compact phases themselves, not reduced to (-pi, pi]
inline void write_gauge_u1(double* phase_field, Lattice2D* lat, std::string output_file) {
  if (lat->get_nc() != 1) { std::cout << "[QMG-ERROR]: U1 gauge functions require Nc = 1 lattice.\n"; return; }
  const int x_len = lat->get_dim_mu(0), y_len = lat->get_dim_mu(1);
  std::vector<double> host = qmg::to_host(phase_field, (size_t)lat->get_size_gauge());
  std::FILE* f = std::fopen(output_file.c_str(), "w");
  if (!f) { std::cout << "[QMG-ERROR]: cannot open " << output_file << " for writing\n"; return; }
  for (int x = 0; x < x_len; x++)
    for (int y = 0; y < y_len; y++)
      for (int mu = 0; mu < 2; mu++) std::fprintf(f, "%.20f\n", host[lat->gauge_coord_to_index(x, y, 0, 0, mu)]);
  std::fclose(f);
}

// polar_vector(phases, gauge_field, n): U = exp(i A)
inline void polar_vector(double* phases, complex<double>* gauge_field, size_t n) { qmg::ok(qmg_u1_phase_to_gauge(gauge_field, phases, n, qmg::current_stream()), "qmg_u1_phase_to_gauge"); }

// Non-compact heatbath (u1_utils.h:607-757).  The reference threads a std::mt19937 through; here the generator state is a
// (seed, sweeps done) pair so that successive calls continue one stream.
struct HeatbathRng { unsigned long long seed, sweeps_done; explicit HeatbathRng(unsigned long long s = 1337ull) : seed(s), sweeps_done(0) {} };
inline void heatbath_noncompact_update(double* phase_field, Lattice2D* lat, double beta, int n_update, HeatbathRng& generator) {
  if (lat->get_nc() != 1) { std::cout << "[QMG-ERROR]: U1 gauge functions require Nc = 1 lattice.\n"; return; }
  qmg::ok(qmg_u1_heatbath_noncompact(phase_field, lat->get_dim_mu(0), lat->get_dim_mu(1), beta, n_update, generator.seed, generator.sweeps_done, qmg::current_stream()),
          "qmg_u1_heatbath_noncompact");
  generator.sweeps_done += (unsigned long long)n_update;
}

inline complex<double> get_plaquette_u1(complex<double>* gauge_field, Lattice2D* lat) {   // :424-462
  if (lat->get_nc() != 1) { std::cout << "[QMG-ERROR]: U1 gauge functions require Nc = 1 lattice.\n"; return -50; }
  double o[3] = {0, 0, 0};
  qmg::ok(qmg_u1_plaquette(gauge_field, lat->get_dim_mu(0), lat->get_dim_mu(1), o, qmg::current_stream()), "qmg_u1_plaquette");
  return complex<double>(o[0], o[1]);
}
inline double get_topo_u1(complex<double>* gauge_field, Lattice2D* lat) {   // :465-508
  if (lat->get_nc() != 1) { std::cout << "[QMG-ERROR]: U1 gauge functions require Nc = 1 lattice.\n"; return -50.1; }
  double o[3] = {0, 0, 0};
  qmg::ok(qmg_u1_plaquette(gauge_field, lat->get_dim_mu(0), lat->get_dim_mu(1), o, qmg::current_stream()), "qmg_u1_plaquette");
  return o[2];
}
inline double get_noncompact_action_u1(double* phase_field, double beta, Lattice2D* lat) {   // :386-421
  if (lat->get_nc() != 1) { std::cout << "[QMG-ERROR]: U1 gauge functions require Nc = 1 lattice.\n"; return -50; }
  double o = 0.0;
  qmg::ok(qmg_u1_noncompact_action(phase_field, lat->get_dim_mu(0), lat->get_dim_mu(1), beta, &o, qmg::current_stream()), "qmg_u1_noncompact_action");
  return o;
}

#endif
