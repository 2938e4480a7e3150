// coarse.hpp -- CoarseOperator2D: Galerkin coarse stencil built on the device
// (reference: operators/coarse.h:29-657; the Sigma/LU part of apply_sigma, :661-894, is out of scope).
#ifndef QMG_COARSE_HPP
#define QMG_COARSE_HPP

#include "operators.hpp"
#include "transfer.hpp"

struct CoarseOperator2D : public Stencil2D {
 protected:
  CoarseOperator2D(CoarseOperator2D const&);
  CoarseOperator2D& operator=(CoarseOperator2D const&);
  Lattice2D* fine_lat;
  bool is_chiral;
  bool use_rbjacobi;
  TransferMG* in_transfer;
  QMGDefaultChirality default_chirality;
  complex<double>* scratch;

  void per_site(const std::vector<double>& sc, const std::vector<int>& sh, complex<double>* out, complex<double>* in) {
    const size_t vol = (size_t)lat->get_volume();
    const int nc = lat->get_nc();
    if (out == in) {
      if (!scratch) scratch = allocate_vector<complex<double>>(lat->get_size_cv_l());
      qmg::pattern(sc.data(), sh.data(), nc, in, scratch, vol);
      copy_vector(out, scratch, lat->get_size_cv_l());
    } else {
      qmg::pattern(sc.data(), sh.data(), nc, in, out, vol);
    }
  }
  // scale top half by `top`, bottom half by `bot`; optionally swap halves (sigma1)
  void halves(double top, double bot, bool swap, complex<double>* out, complex<double>* in) {
    const int nc = lat->get_nc();
    std::vector<double> sc(nc);
    std::vector<int> sh(nc);
    for (int c = 0; c < nc; c++) { sc[c] = (c < nc / 2) ? top : bot; sh[c] = swap ? (c + nc / 2) % nc : c; }
    per_site(sc, sh, out, in);
  }

 public:
  enum QMGCoarseBuildStencil {
    QMG_COARSE_BUILD_ORIGINAL = 0, QMG_COARSE_BUILD_DAGGER = 1, QMG_COARSE_BUILD_RBJACOBI = 2, QMG_COARSE_BUILD_DAGGER_RBJACOBI = 3,
    QMG_COARSE_BUILD_RBJDAGGER = 4, QMG_COARSE_BUILD_ALL = 5,
  };

  // bare stencil (coarse.h:75-81)
  CoarseOperator2D(Lattice2D* in_lat, int pieces, bool is_chiral, QMGDefaultChirality def_chiral = QMG_CHIRALITY_NONE, complex<double> in_shift = 0.0,
                   complex<double> in_eo_shift = 0.0, complex<double> in_dof_shift = 0.0)
      : Stencil2D(in_lat, pieces, in_shift, in_eo_shift, in_dof_shift), fine_lat(0), is_chiral(is_chiral), use_rbjacobi(false), in_transfer(0),
        default_chirality(def_chiral), scratch(0) {}

  // Galerkin build (coarse.h:90-471): P^dag A P by 9 probes per coarse colour, all on the device.
  CoarseOperator2D(Lattice2D* in_lat, Stencil2D* fine_stencil, Lattice2D* fine_lattice, TransferMG* transfer, bool is_chiral = false,
                   bool use_rbjacobi = false, QMGCoarseBuildStencil build_extra = QMG_COARSE_BUILD_ORIGINAL)
      : Stencil2D(in_lat, QMG_PIECE_CLOVER_HOPPING, 0.0, 0.0, 0.0), fine_lat(fine_lattice), is_chiral(is_chiral), use_rbjacobi(use_rbjacobi),
        in_transfer(transfer), scratch(0) {
    switch (in_transfer->get_doubling()) {
      case QMG_DOUBLE_PROJECTION: default_chirality = QMG_CHIRALITY_GAMMA_5; break;
      case QMG_DOUBLE_OPERATOR: default_chirality = QMG_CHIRALITY_SIGMA_1; break;
      default: default_chirality = QMG_CHIRALITY_NONE; break;
    }
    if (use_rbjacobi) fine_stencil->perform_swap_rbjacobi();   // :120-123 (zeroes the fine shifts while swapped)
    // :131 transfers only the identity shift -- and sets only `shift`, leaving shift_backup at its constructor value 0, so that
    // in the reference any later swap-back (dagger / rbjacobi applies: perform_swap_* restores the backups) silently resets a
    // Galerkin operator's shift to 0.  DELIBERATE DEVIATION: the backup is set too, so CGNE smoothers / MDM / RBJACOBI builds on
    // an ORIGINAL-built coarse operator keep the mass term (facade_selftest checks apply_M before and after a dagger swap).
    shift = shift_backup = fine_stencil->get_shift();
    qmg_stencil_desc fd = fine_stencil->desc();
    if (qmg::slab().on) {   // the hops that leave the slab need the neighbouring ranks' rows of the prolongator
      const int nv = lat->get_nc();
      const size_t hs = (size_t)fd.Lx * fd.nc;
      complex<double>*lo = allocate_vector<complex<double>>(hs * nv), *hi = allocate_vector<complex<double>>(hs * nv);
      qmg::ok(qmg_halo_exchange(QMG_C64, transfer->device_null_vectors(), fd.Lx, fd.Ly, fd.nc, lo, hi, nv, (size_t)fine_lattice->get_size_cv_l(), hs, qmg::current_stream()),
              "qmg_halo_exchange");
      qmg::ok(qmg_coarse_build_slab(clover, hopping, &fd, transfer->device_null_vectors(), transfer->device_restrict_vectors(), lat->get_dim_mu(0), lat->get_dim_mu(1),
                                    lat->get_nc(), lo, hi, hs, qmg::current_stream()), "qmg_coarse_build_slab");
      qmg::ok(qmg_stream_sync(qmg::current_stream()), "qmg_stream_sync");
      deallocate_vector(&lo); deallocate_vector(&hi);
    } else
    qmg::ok(qmg_coarse_build(clover, hopping, &fd, transfer->device_null_vectors(), transfer->device_restrict_vectors(), lat->get_dim_mu(0),
                             lat->get_dim_mu(1), lat->get_nc(), qmg::current_stream()), "qmg_coarse_build");
    if (use_rbjacobi) fine_stencil->perform_swap_rbjacobi();
    generated = true;
    if (build_extra == QMG_COARSE_BUILD_DAGGER || build_extra == QMG_COARSE_BUILD_DAGGER_RBJACOBI || build_extra == QMG_COARSE_BUILD_ALL) build_dagger_stencil();
    if (build_extra == QMG_COARSE_BUILD_RBJACOBI || build_extra == QMG_COARSE_BUILD_DAGGER_RBJACOBI || build_extra == QMG_COARSE_BUILD_RBJDAGGER ||
        build_extra == QMG_COARSE_BUILD_ALL) build_rbjacobi_stencil();
    if (build_extra == QMG_COARSE_BUILD_RBJDAGGER || build_extra == QMG_COARSE_BUILD_ALL) build_rbj_dagger_stencil();
  }

  ~CoarseOperator2D() { if (scratch) deallocate_vector(&scratch); }

  static int get_dof(int i = 0) { return -1; }
  static chirality_state has_chirality() { return QMG_CHIRAL_UNKNOWN; }

  // coarse gamma5 = diag(+1 top half, -1 bottom half) when chiral, otherwise nothing happens -- not even the
  // copy of the two-argument form (coarse.h:498-521)
  virtual void gamma5(complex<double>* vec) { if (is_chiral) halves(1.0, -1.0, false, vec, vec); }
  virtual void gamma5(complex<double>* g5_vec, complex<double>* vec) { if (is_chiral) halves(1.0, -1.0, false, g5_vec, vec); }
  // sigma1 swaps the two halves of the dof for any even nc (:523-557)
  virtual void sigma1(complex<double>* vec) { if (lat->get_nc() % 2 == 0) halves(1.0, 1.0, true, vec, vec); }
  virtual void sigma1(complex<double>* s1_vec, complex<double>* vec) { if (lat->get_nc() % 2 == 0) halves(1.0, 1.0, true, s1_vec, vec); }
  // projections (:560-640): by halves for gamma5-type chirality, (1 +- sigma1)/2 for sigma1-type
  virtual void chiral_projection(complex<double>* v, bool is_up) {
    if (!is_chiral) return;
    if (default_chirality == QMG_CHIRALITY_GAMMA_5) halves(is_up ? 1.0 : 0.0, is_up ? 0.0 : 1.0, false, v, v);
    else if (default_chirality == QMG_CHIRALITY_SIGMA_1) { sigma1(extra_cvector, v); caxpby(is_up ? 0.5 : -0.5, extra_cvector, 0.5, v, lat->get_size_cv_l()); }
  }
  virtual void chiral_projection_copy(complex<double>* orig, complex<double>* dest, bool is_up) {
    if (!is_chiral) return;
    if (default_chirality == QMG_CHIRALITY_GAMMA_5) halves(is_up ? 1.0 : 0.0, is_up ? 0.0 : 1.0, false, dest, orig);
    else if (default_chirality == QMG_CHIRALITY_SIGMA_1) { sigma1(extra_cvector, orig); caxpbyz(is_up ? 0.5 : -0.5, extra_cvector, 0.5, orig, dest, lat->get_size_cv_l()); }
  }
  virtual void chiral_projection_both(complex<double>* orig_to_up, complex<double>* down) {
    if (!is_chiral) return;
    if (default_chirality == QMG_CHIRALITY_GAMMA_5) {
      halves(0.0, 1.0, false, down, orig_to_up);
      halves(1.0, 0.0, false, orig_to_up, orig_to_up);
    } else if (default_chirality == QMG_CHIRALITY_SIGMA_1) {
      sigma1(extra_cvector, orig_to_up);
      caxpbyz(0.5, orig_to_up, -0.5, extra_cvector, down, lat->get_size_cv_l());
      caxpy(-1.0, down, orig_to_up, lat->get_size_cv_l());
    }
  }
  virtual QMGDefaultChirality get_default_chirality() { return default_chirality; }
};

#endif
