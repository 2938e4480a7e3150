// operators.hpp -- the concrete stencils of the reference on device arrays:
//   Wilson2D (operators/wilson.h), Staggered2D (operators/staggered.h), GaugedLaplace2D
//   (operators/gaugedlaplace.h), FreeLaplace2D (tests/n02_free_laplace_test/free_laplace.h).
// CoarseOperator2D lives in coarse.hpp (it needs TransferMG).
#ifndef QMG_OPERATORS_HPP
#define QMG_OPERATORS_HPP

#include "stencil2d.hpp"

namespace qmg {
inline void pattern(const double* scale, const int* shuffle, int nc, const complex<double>* x, complex<double>* y, size_t nsite) {
  ok(qmg_caxy_pattern(scale, shuffle, nc, x, y, nsite, current_stream()), "qmg_caxy_pattern");
}
}  // namespace qmg

// ---------------- Wilson (nc = 2: two spin components over U(1)) ----------------
struct Wilson2D : public Stencil2D {
 protected:
  Wilson2D(Wilson2D const&);
  Wilson2D& operator=(Wilson2D const&);
  double wilson_coeff;
  complex<double>* scratch;   // for in-place per-site permutations

  void per_site(const double s0, const double s1, int p0, int p1, complex<double>* out, complex<double>* in) {
    const double sc[2] = {s0, s1};
    const int sh[2] = {p0, p1};
    const size_t vol = (size_t)lat->get_volume();
    if (out == in) {
      if (!scratch) scratch = allocate_vector<complex<double>>(lat->get_size_cv_l());
      qmg::pattern(sc, sh, 2, in, scratch, vol);
      copy_vector(out, scratch, lat->get_size_cv_l());
    } else {
      qmg::pattern(sc, sh, 2, in, out, vol);
    }
  }

 public:
  void update_links(complex<double>* gauge_links) {   // wilson.h:153-226; gauge_links: DEVICE nc=1 LatticeGauge (y-slab mode: of the WHOLE lattice)
    if (qmg::slab().on)
      qmg::ok(qmg_wilson_fill_slab(clover, hopping, gauge_links, lat->get_dim_mu(0), lat->get_dim_mu(1) * qmg::slab().world, qmg::slab().rank * lat->get_dim_mu(1),
                                   lat->get_dim_mu(1), wilson_coeff, qmg::current_stream()), "qmg_wilson_fill_slab");
    else
      qmg::ok(qmg_wilson_fill(clover, hopping, gauge_links, lat->get_dim_mu(0), lat->get_dim_mu(1), wilson_coeff, qmg::current_stream()), "qmg_wilson_fill");
    if (built_dagger) { deallocate_vector(&dagger_clover); deallocate_vector(&dagger_hopping); built_dagger = false; }
    if (built_rbjacobi) { deallocate_vector(&rbjacobi_cinv); deallocate_vector(&rbjacobi_clover); deallocate_vector(&rbjacobi_hopping); built_rbjacobi = false; }
    // (the reference leaves a built rbj_dagger stencil dangling here, wilson.h:211-225; it is dropped too)
    if (built_rbj_dagger) { deallocate_vector(&rbj_dagger_cinv); deallocate_vector(&rbj_dagger_clover); deallocate_vector(&rbj_dagger_hopping); built_rbj_dagger = false; }
    set_direct_links(gauge_links, wilson_coeff);   // the ORIGINAL-operator applies go straight from the links (qmg_wilson.hip)
    generated = true;
  }

  Wilson2D(Lattice2D* in_lat, complex<double> mass, complex<double>* gauge_links, double wilson_coeff = 1.0)
      : Stencil2D(in_lat, QMG_PIECE_CLOVER_HOPPING, mass, 0.0, 0.0), wilson_coeff(wilson_coeff), scratch(0) {
    if (lat->get_nc() != 2) { std::cout << "[QMG-ERROR]: Wilson2D only supports Nc = 2.\n"; return; }
    update_links(gauge_links);
  }
  ~Wilson2D() { if (scratch) deallocate_vector(&scratch); }

  static int get_dof(int i = 0) { return 2; }
  static chirality_state has_chirality() { return QMG_CHIRAL_YES; }

  virtual void gamma5(complex<double>* vec) { per_site(1.0, -1.0, 0, 1, vec, vec); }                                   // :74-81
  virtual void gamma5(complex<double>* g5_vec, complex<double>* vec) { per_site(1.0, -1.0, 0, 1, g5_vec, vec); }        // :83-93
  virtual void chiral_projection(complex<double>* v, bool is_up) { is_up ? per_site(1.0, 0.0, 0, 1, v, v) : per_site(0.0, 1.0, 0, 1, v, v); }   // :96-102
  virtual void chiral_projection_copy(complex<double>* orig, complex<double>* dest, bool is_up) {                        // :105-117
    is_up ? per_site(1.0, 0.0, 0, 1, dest, orig) : per_site(0.0, 1.0, 0, 1, dest, orig);
  }
  virtual void chiral_projection_both(complex<double>* orig_to_up, complex<double>* down) {                              // :120-125
    per_site(0.0, 1.0, 0, 1, down, orig_to_up);
    per_site(1.0, 0.0, 0, 1, orig_to_up, orig_to_up);
  }
  virtual void sigma1(complex<double>* vec) { per_site(1.0, 1.0, 1, 0, vec, vec); }                                      // :128-135
  virtual void sigma1(complex<double>* s1_vec, complex<double>* vec) { per_site(1.0, 1.0, 1, 0, s1_vec, vec); }          // :138-143
  virtual QMGDefaultChirality get_default_chirality() { return QMG_CHIRALITY_GAMMA_5; }
};

// ---------------- shared by the two nc = 1 operators: hand-rolled even-odd normal operator ----------------
struct EoPrecNc1 : public Stencil2D {
 protected:
  complex<double>* tmp_eo_space;
  EoPrecNc1(Lattice2D* l, int pieces, complex<double> s) : Stencil2D(l, pieces, s, 0.0, 0.0), tmp_eo_space(0) {}
  ~EoPrecNc1() { if (tmp_eo_space) deallocate_vector(&tmp_eo_space); }
  void drop_variants() {
    if (built_dagger) { if (dagger_clover) deallocate_vector(&dagger_clover); deallocate_vector(&dagger_hopping); built_dagger = false; }
    if (built_rbjacobi) { deallocate_vector(&rbjacobi_cinv); if (rbjacobi_clover) deallocate_vector(&rbjacobi_clover); deallocate_vector(&rbjacobi_hopping); built_rbjacobi = false; }
  }
  // b_new_e = diag b_e - D_eo b_o
  void prepare_b_impl(complex<double>* b_new, complex<double>* b, complex<double> diag) {
    const long half = lat->get_size_cv_l() / 2;
    launch(QMG_P_EO | QMG_P_ZERO_E, b_new, b, 0, hopping, 0.0, 0.0, 0.0);
    caxpby(diag, b, complex<double>(-1.0), b_new, half);
  }
  // lhs_e = diag^2 rhs_e - D_eo D_oe rhs_e
  void apply_eo_prec_impl(complex<double>* lhs, complex<double>* rhs, complex<double> diag) {
    const long cv = lat->get_size_cv_l();
    if (!tmp_eo_space) tmp_eo_space = allocate_vector<complex<double>>(cv);
    launch(QMG_P_OE | QMG_P_ZERO_O, tmp_eo_space, rhs, 0, hopping, 0.0, 0.0, 0.0);
    launch(QMG_P_EO | QMG_P_ZERO_E, tmp_eo_space, tmp_eo_space, 0, hopping, 0.0, 0.0, 0.0);
    caxpbyz(diag * diag, rhs, complex<double>(-1.0), tmp_eo_space, lhs, cv / 2);
  }
  // x_o = (b_o - D_oe x_e) / diag
  void reconstruct_x_impl(complex<double>* x, complex<double>* b, complex<double> diag) {
    const long half = lat->get_size_cv_l() / 2;
    launch(QMG_P_OE | QMG_P_ZERO_O, x, x, 0, hopping, 0.0, 0.0, 0.0);
    caxpby(1.0 / diag, b + half, -1.0 / diag, x + half, half);
  }
};

// ---------------- Staggered (staggered.h) ----------------
struct Staggered2D : public EoPrecNc1 {
  void update_links(complex<double>* gauge_links) {   // :81-123
    if (qmg::slab().on)   // y-slab mode: gauge_links is the gauge field of the WHOLE lattice, this rank fills its rows
      qmg::ok(qmg_staggered_fill_slab(hopping, gauge_links, lat->get_dim_mu(0), lat->get_dim_mu(1) * qmg::slab().world, qmg::slab().rank * lat->get_dim_mu(1),
                                      lat->get_dim_mu(1), qmg::current_stream()), "qmg_staggered_fill_slab");
    else
    qmg::ok(qmg_staggered_fill(hopping, gauge_links, lat->get_dim_mu(0), lat->get_dim_mu(1), qmg::current_stream()), "qmg_staggered_fill");
    drop_variants();
    generated = true;
  }
  Staggered2D(Lattice2D* in_lat, complex<double> mass, complex<double>* gauge_links) : EoPrecNc1(in_lat, QMG_PIECE_HOPPING, mass) {
    if (lat->get_nc() != 1) { std::cout << "[QMG-ERROR]: Staggered2D only supports Nc = 1.\n"; return; }
    update_links(gauge_links);
  }
  static int get_dof(int i = 0) { return 1; }
  static chirality_state has_chirality() { return QMG_CHIRAL_YES; }
  virtual void gamma5(complex<double>* vec) { const long h = lat->get_size_cv_l() / 2; cax(-1.0, vec + h, h); }                      // :140-143
  virtual void gamma5(complex<double>* g5_vec, complex<double>* vec) { const long h = lat->get_size_cv_l() / 2; copy_vector(g5_vec, vec, h); caxy(-1.0, vec + h, g5_vec + h, h); }
  virtual void chiral_projection(complex<double>* v, bool is_up) { const long h = lat->get_size_cv_l() / 2; zero_vector(is_up ? v + h : v, h); }   // :152-158
  virtual void chiral_projection_copy(complex<double>* orig, complex<double>* dest, bool is_up) {
    const long h = lat->get_size_cv_l() / 2;
    if (is_up) { zero_vector(dest + h, h); copy_vector(dest, orig, h); } else { zero_vector(dest, h); copy_vector(dest + h, orig + h, h); }
  }
  virtual void chiral_projection_both(complex<double>* orig_to_up, complex<double>* down) {
    const long h = lat->get_size_cv_l() / 2;
    zero_vector(down, h); copy_vector(down + h, orig_to_up + h, h); zero_vector(orig_to_up + h, h);
  }
  virtual QMGDefaultChirality get_default_chirality() { return QMG_CHIRALITY_GAMMA_5; }
  void prepare_b(complex<double>* b_new, complex<double>* b) { prepare_b_impl(b_new, b, shift); }                 // :190-202
  void apply_eo_prec_M(complex<double>* lhs, complex<double>* rhs) { apply_eo_prec_impl(lhs, rhs, shift); }       // m^2 - D_eo D_oe (:206-224)
  void reconstruct_x(complex<double>* x, complex<double>* b) { reconstruct_x_impl(x, b, shift); }                 // :228-240
};
inline void apply_eo_staggered_2D_M(complex<double>* lhs, complex<double>* rhs, void* extra_data) { ((Staggered2D*)extra_data)->apply_eo_prec_M(lhs, rhs); }

// ---------------- Gauged Laplace (gaugedlaplace.h) ----------------
struct GaugedLaplace2D : public EoPrecNc1 {
  void update_links(complex<double>* gauge_links) {   // :77-115
    if (qmg::slab().on)   // y-slab mode: gauge_links is the gauge field of the WHOLE lattice
      qmg::ok(qmg_laplace_fill_slab(clover, hopping, gauge_links, lat->get_dim_mu(0), lat->get_dim_mu(1) * qmg::slab().world, qmg::slab().rank * lat->get_dim_mu(1),
                                    lat->get_dim_mu(1), qmg::current_stream()), "qmg_laplace_fill_slab");
    else
    qmg::ok(qmg_laplace_fill(clover, hopping, gauge_links, lat->get_dim_mu(0), lat->get_dim_mu(1), qmg::current_stream()), "qmg_laplace_fill");
    drop_variants();
    generated = true;
  }
  GaugedLaplace2D(Lattice2D* in_lat, complex<double> mass_sq, complex<double>* gauge_links) : EoPrecNc1(in_lat, QMG_PIECE_CLOVER_HOPPING, mass_sq) {
    if (lat->get_nc() != 1) { std::cout << "[QMG-ERROR]: GaugedLaplace2D only supports Nc = 1.\n"; return; }
    update_links(gauge_links);
  }
  static int get_dof(int i = 0) { return 1; }
  static chirality_state has_chirality() { return QMG_CHIRAL_NO; }
  virtual void chiral_projection(complex<double>*, bool) { return; }
  virtual void chiral_projection_copy(complex<double>*, complex<double>*, bool) { return; }
  virtual void chiral_projection_both(complex<double>*, complex<double>*) { return; }
  virtual QMGDefaultChirality get_default_chirality() { return QMG_CHIRALITY_NONE; }
  void prepare_b(complex<double>* b_new, complex<double>* b) { prepare_b_impl(b_new, b, 4.0 + shift); }             // :154-166
  void apply_eo_prec_M(complex<double>* lhs, complex<double>* rhs) { apply_eo_prec_impl(lhs, rhs, 4.0 + shift); }   // :170-188
  void reconstruct_x(complex<double>* x, complex<double>* b) { reconstruct_x_impl(x, b, 4.0 + shift); }             // :192-204
};
inline void apply_eo_gauge_laplace_2D_M(complex<double>* lhs, complex<double>* rhs, void* extra_data) { ((GaugedLaplace2D*)extra_data)->apply_eo_prec_M(lhs, rhs); }

// ---------------- Free Laplace (tests/n02_free_laplace_test/free_laplace.h:18-42) ----------------
struct FreeLaplace2D : public Stencil2D {
  FreeLaplace2D(Lattice2D* in_lat, complex<double> mass_sq) : Stencil2D(in_lat, QMG_PIECE_CLOVER_HOPPING, mass_sq, 0.0, 0.0) {
    if (lat->get_nc() != 1) { std::cout << "[QMG-ERROR]: FreeLaplace2D only supports Nc = 1.\n"; return; }
    // 4 on the clover, -1 on the hopping: fill = constant -> zero then shift-by-constant via caxy on a ones vector is overkill; upload.
    std::vector<complex<double>> c((size_t)lat->get_size_cm_l(), 4.0), h((size_t)lat->get_size_hopping_l(), -1.0);
    qmg::upload(clover, c.data(), c.size());
    qmg::upload(hopping, h.data(), h.size());
    generated = true;
  }
  virtual void chiral_projection(complex<double>*, bool) { return; }
  virtual void chiral_projection_copy(complex<double>*, complex<double>*, bool) { return; }
  virtual void chiral_projection_both(complex<double>*, complex<double>*) { return; }
  virtual QMGDefaultChirality get_default_chirality() { return QMG_CHIRALITY_NONE; }
};

#endif
