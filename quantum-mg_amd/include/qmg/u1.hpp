// u1.hpp -- the two U(1) gauge utilities the path needs as inputs (reference: u1/u1_utils.h:38-67,172-181).
// Gauge generation, smearing, plaquette/topology are input preparation and out of scope.
// The field is read on the host in the reference's text format and uploaded; `gauge_field` is a DEVICE
// nc=1 LatticeGauge (mu, eo, y, x).
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

#endif
