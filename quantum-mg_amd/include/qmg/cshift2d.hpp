// cshift2d.hpp -- even/odd circular shifts on device arrays (reference: cshift/cshift_2d.h:13-236).
#ifndef QMG_CSHIFT2D_HPP
#define QMG_CSHIFT2D_HPP

#include <iostream>

#include "lattice2d.hpp"
#include "qmg_device.hpp"

// enums qmg_cshift_dir / qmg_eo: defined in include/qmg_hip.h with the reference's names and values.

// lhs(opposite parity half) = rhs(neighbour); complex<double> fields with dof_per_site entries per site.
inline void cshift(complex<double>* lhs, complex<double>* rhs, int cdir, int eo, int dof_per_site, Lattice2D* lat) {
  if (cdir > 5) {
    if (eo & 1) std::cout << "[ERROR-QMG]: cshift_from_even does not support distance two stencils yet.\n";
    if (eo & 2) std::cout << "[ERROR-QMG]: cshift_from_odd does not support distance two stencils yet.\n";
    return;
  }
  if (qmg::slab().on && qmg::slab().world > 1 && (cdir == QMG_CSHIFT_FROM_YP1 || cdir == QMG_CSHIFT_FROM_YM1)) {
    // a y-shift of a slab needs the neighbouring rank's row (the stencil applies take it from qmg_halo_exchange); this host-side utility does not
    std::cout << "[QMG-ERROR]: cshift in the y direction is not decomposed into y-slabs.\n";
    return;
  }
  qmg::ok(qmg_cshift(lhs, rhs, cdir, eo, dof_per_site, lat->get_dim_mu(0), lat->get_dim_mu(1), qmg::current_stream()), "qmg_cshift");
}
inline void cshift_from_even(complex<double>* lhs, complex<double>* rhs, int cdir, int dof, Lattice2D* lat) { cshift(lhs, rhs, cdir, 1, dof, lat); }
inline void cshift_from_odd(complex<double>* lhs, complex<double>* rhs, int cdir, int dof, Lattice2D* lat) { cshift(lhs, rhs, cdir, 2, dof, lat); }

#endif
