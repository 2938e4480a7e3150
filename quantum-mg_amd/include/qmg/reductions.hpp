// reductions.hpp -- the reference's reductions/reductions.h on device vectors: per-timeslice reductions of a colour vector
// (`norm2sq_cv_timeslice` :24-41, `redot_cv_timeslice` :47-66, `dot_cv_timeslice` :69-87) and `gaussian_wall_source` (:90-162).
// Same names and argument order; `cv` arguments are DEVICE vectors, `sum` is a HOST array of Nt = dim_mu(nd-1) entries (the
// callers print / accumulate them: tests/n15, n16, n20).  One block per timeslice: a row y of the even-odd layout is two
// contiguous runs (csrc/qmg_blas.hip k_timeslice), so there is no per-element index division as in the reference's loop.
// gaussian_wall_source takes a SEED where the reference takes a std::mt19937& (the device generator is counter-based, see
// `gaussian` in qmg_device.hpp): same distribution, a different stream of numbers.
#ifndef QMG_REDUCTIONS_HPP
#define QMG_REDUCTIONS_HPP

#include <vector>

#include "lattice2d.hpp"
#include "qmg_device.hpp"

namespace qmg {
inline bool timeslice_refuses_slabs(const char* fn) {
  if (!slab().on) return false;
  std::cout << "[QMG-ERROR]: " << fn << " is not available in y-slab mode (a timeslice is one row of ONE rank's slab).\n";
  return true;
}
}  // namespace qmg

inline void norm2sq_cv_timeslice(double* sum, complex<double>* cv, Lattice2D* lat) {
  if (qmg::timeslice_refuses_slabs("norm2sq_cv_timeslice")) return;
  qmg::ok(qmg_norm2sq_cv_timeslice(cv, lat->get_dim_mu(0), lat->get_dim_mu(1), lat->get_nc(), nullptr, sum, qmg::current_stream()), "qmg_norm2sq_cv_timeslice");
}

inline void redot_cv_timeslice(double* sum, complex<double>* cv1, complex<double>* cv2, Lattice2D* lat) {
  if (qmg::timeslice_refuses_slabs("redot_cv_timeslice")) return;
  qmg::ok(qmg_redot_cv_timeslice(cv1, cv2, lat->get_dim_mu(0), lat->get_dim_mu(1), lat->get_nc(), nullptr, sum, qmg::current_stream()), "qmg_redot_cv_timeslice");
}

inline void dot_cv_timeslice(complex<double>* sum, complex<double>* cv1, complex<double>* cv2, Lattice2D* lat) {
  if (qmg::timeslice_refuses_slabs("dot_cv_timeslice")) return;
  const int nt = lat->get_dim_mu(1);
  std::vector<double> r(2 * (size_t)nt);
  if (!qmg::ok(qmg_dot_cv_timeslice(cv1, cv2, lat->get_dim_mu(0), nt, lat->get_nc(), nullptr, r.data(), qmg::current_stream()), "qmg_dot_cv_timeslice")) return;
  for (int t = 0; t < nt; t++) sum[t] = complex<double>(r[2 * t], r[2 * t + 1]);
}

// cv = a real Gaussian wall (mean + deviation N(0,1)) on timeslice `timeslice`, component `color`; zero elsewhere (:90-162)
inline void gaussian_wall_source(complex<double>* cv, int timeslice, int color, Lattice2D* lat, unsigned long long seed, double deviation = 1.0, double mean = 0.0) {
  if (qmg::timeslice_refuses_slabs("gaussian_wall_source")) return;
  if (timeslice >= lat->get_dim_mu(1)) { std::cout << "[QMG-ERROR]: Cannot create gaussian wall source for t < Nt.\n"; return; }
  if (color >= lat->get_nc()) { std::cout << "[QMG-ERROR]: Cannot create gaussian wall source for color < Nc.\n"; return; }
  qmg::ok(qmg_gaussian_wall_source(cv, lat->get_dim_mu(0), lat->get_dim_mu(1), lat->get_nc(), timeslice, color, seed, deviation, mean, qmg::current_stream()), "qmg_gaussian_wall_source");
}

#endif
