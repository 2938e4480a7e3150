// transfer.hpp -- TransferMG on device arrays (reference: transfer/transfer.h:42-820).
// The null vectors live in ONE contiguous device allocation, vector-major (d, fine cv index), which is
// exactly the reference's `null_vectors[d][k]` with a fixed stride; `null_vectors[d]` pointers into it
// are exposed for code that indexes them the reference way.  The one-to-many `coarse_map` of the
// reference (:410-448) is not materialised: the kernels derive the block of a fine element from its
// position (regular rectangular blocks).
#ifndef QMG_TRANSFER_HPP
#define QMG_TRANSFER_HPP

#include <iostream>

#include "lattice2d.hpp"
#include "qmg_device.hpp"

enum QMGDoublingType { QMG_DOUBLE_NONE = 0, QMG_DOUBLE_PROJECTION = 1, QMG_DOUBLE_OPERATOR = 2 };

class TransferMG {
 private:
  TransferMG(TransferMG const&);
  TransferMG& operator=(TransferMG const&);

  Lattice2D* fine_lat;
  Lattice2D* coarse_lat;
  int const_num_null_vec;
  int blocksizes[2];
  int fine_sites_per_coarse;
  complex<double>* null_store;       // nvec * fine_size_cv, contiguous
  complex<double>* restrict_store;   // same for an asymmetric restrictor, or 0
  QMGDoublingType doubling;
  bool is_init;
  void* null_store32;                // complex<float> shadow of null_store / restrict_store (enable_f32_shadow), or 0
  void* restrict_store32;
  bool narrow_precond;               // the K-cycle's own transfers may stream the complex<float> copy under complex<double> vectors (enable_narrow_precond)

  bool geometry_ok() {
    fine_sites_per_coarse = fine_lat->get_nc();
    for (int i = 0; i < 2; i++) {
      if (fine_lat->get_dim_mu(i) % coarse_lat->get_dim_mu(i) != 0) {
        std::cout << "[QMG-ERROR]: Fine lattice dimension " << i << "isn't divided evenly by coarse dimension.\n";
        return false;
      }
      blocksizes[i] = fine_lat->get_dim_mu(i) / coarse_lat->get_dim_mu(i);
      fine_sites_per_coarse *= blocksizes[i];
    }
    return true;
  }
  complex<double>* copy_in(complex<double>** vecs) {
    const long fsize = fine_lat->get_size_cv_l();
    complex<double>* store = allocate_vector<complex<double>>((size_t)const_num_null_vec * fsize);
    for (int i = 0; i < const_num_null_vec; i++) copy_vector(store + i * fsize, vecs[i], fsize);
    return store;
  }
  // `passes` Gram-Schmidt passes per block in one launch (tile in LDS; csrc/qmg_setup.hip), the factor saved in the first
  void ortho_passes(complex<double>* chol, int passes) {
    qmg::ok(qmg_block_orthonormalize_n(null_store, const_num_null_vec, fine_lat->get_dim_mu(0), fine_lat->get_dim_mu(1), fine_lat->get_nc(),
                                       coarse_lat->get_dim_mu(0), coarse_lat->get_dim_mu(1), chol, passes, qmg::current_stream()),
            "qmg_block_orthonormalize_n");
  }

 public:
  complex<double>** null_vectors;            // null_vectors[d] -> device vector d (views into one allocation)
  complex<double>** restrict_null_vectors;   // 0 when the restrictor is P^dagger
  complex<double>* block_cholesky;           // device, coarse size_cm, or 0
  complex<double>* block_L;
  complex<double>* block_U;

  // transfer.h:118-179.  in_null_vectors[d]: device vectors; they are copied (the caller keeps ownership).
  TransferMG(Lattice2D* in_fine_lat, Lattice2D* in_coarse_lat, complex<double>** in_null_vectors, bool do_block_ortho = true,
             bool save_decomp = false, QMGDoublingType in_doubling = QMG_DOUBLE_NONE)
      : fine_lat(in_fine_lat), coarse_lat(in_coarse_lat), const_num_null_vec(in_coarse_lat->get_nc()), null_store(0), restrict_store(0),
        doubling(in_doubling), is_init(false), null_store32(0), restrict_store32(0), narrow_precond(false), null_vectors(0), restrict_null_vectors(0), block_cholesky(0), block_L(0), block_U(0) {
    if (!geometry_ok()) return;
    null_store = copy_in(in_null_vectors);
    null_vectors = new complex<double>*[const_num_null_vec];
    for (int i = 0; i < const_num_null_vec; i++) null_vectors[i] = null_store + (long)i * fine_lat->get_size_cv_l();
    if (save_decomp) {
      block_cholesky = allocate_vector<complex<double>>(coarse_lat->get_size_cm_l());
      zero_vector(block_cholesky, coarse_lat->get_size_cm_l());
    }
    if (do_block_ortho) {   // twice; the decomposition is saved on the first pass only (:160-174)
      ortho_passes(block_cholesky, 2);
    }
    is_init = true;
  }

  // Separate prolongator / restrictor (:185-230): block BI-orthonormalisation (:610-769) so that R^dag P = 1 per block;
  // run twice, the LU factors saved on the first pass only (:208-225).
  TransferMG(Lattice2D* in_fine_lat, Lattice2D* in_coarse_lat, complex<double>** in_prolong_null_vectors,
             complex<double>** in_restrict_null_vectors, bool do_block_bi_ortho = true, bool save_decomp = false,
             QMGDoublingType in_doubling = QMG_DOUBLE_NONE)
      : TransferMG(in_fine_lat, in_coarse_lat, in_prolong_null_vectors, false, false, in_doubling) {
    if (!is_init) return;
    restrict_store = copy_in(in_restrict_null_vectors);
    restrict_null_vectors = new complex<double>*[const_num_null_vec];
    for (int i = 0; i < const_num_null_vec; i++) restrict_null_vectors[i] = restrict_store + (long)i * fine_lat->get_size_cv_l();
    if (save_decomp) {
      block_L = allocate_vector<complex<double>>(coarse_lat->get_size_cm_l());
      block_U = allocate_vector<complex<double>>(coarse_lat->get_size_cm_l());
      zero_vector(block_L, coarse_lat->get_size_cm_l());
      zero_vector(block_U, coarse_lat->get_size_cm_l());
    }
    if (do_block_bi_ortho) {
      for (int pass = 0; pass < 2; pass++)
        qmg::ok(qmg_block_bi_orthonormalize(null_store, restrict_store, const_num_null_vec, fine_lat->get_dim_mu(0), fine_lat->get_dim_mu(1),
                                            fine_lat->get_nc(), coarse_lat->get_dim_mu(0), coarse_lat->get_dim_mu(1), pass == 0 ? block_L : 0,
                                            pass == 0 ? block_U : 0, qmg::current_stream()), "qmg_block_bi_orthonormalize");
    }
  }

  ~TransferMG() {
    if (null_vectors) delete[] null_vectors;
    if (restrict_null_vectors) delete[] restrict_null_vectors;
    if (null_store) deallocate_vector(&null_store);
    if (restrict_store) deallocate_vector(&restrict_store);
    if (block_cholesky) deallocate_vector(&block_cholesky);
    if (block_L) deallocate_vector(&block_L);
    if (block_U) deallocate_vector(&block_U);
    disable_f32_shadow();
  }

  // fp32 shadow of the (block-orthonormalised) null vectors for the QMG_C32 K-cycle; the fp64 vectors stay the master copy
  bool enable_f32_shadow() {
    disable_f32_shadow();
    const size_t n = (size_t)const_num_null_vec * (size_t)fine_lat->get_size_cv_l();
    if (qmg_malloc(&null_store32, n * 8) != QMG_SUCCESS) { null_store32 = 0; return false; }
    bool good = qmg::ok(qmg_convert(null_store32, QMG_C32, null_store, QMG_C64, n, qmg::current_stream()), "qmg_convert");
    if (good && restrict_store) {
      if (qmg_malloc(&restrict_store32, n * 8) != QMG_SUCCESS) { restrict_store32 = 0; good = false; }
      else good = qmg::ok(qmg_convert(restrict_store32, QMG_C32, restrict_store, QMG_C64, n, qmg::current_stream()), "qmg_convert");
    }
    if (!good) disable_f32_shadow();
    return good;
  }
  void disable_f32_shadow() {
    if (null_store32) { qmg_free(null_store32); null_store32 = 0; }
    if (restrict_store32) { qmg_free(restrict_store32); restrict_store32 = 0; }
  }   // (narrow_precond stays as it is: the narrow route also needs null_store32, and enable_f32_shadow re-creates the copy through here)
  bool has_f32_shadow() const { return null_store32 != 0; }
  // Opt-in for a hierarchy that only PRECONDITIONS (StatefulMultigridMG under the same policy as Stencil2D::enable_f32_matrices): the K-cycle's own
  // restrict / prolong of ONE system stream the complex<float> copy of the null vectors under complex<double> vectors -- half the bytes of an
  // HBM-bound transfer (qmg_*_batch_nv32; arithmetic fp64).  The Galerkin build, prolong_c2f / restrict_f2c and the batch forms keep the fp64 vectors.
  bool enable_narrow_precond() {
    if (!null_store32 && !enable_f32_shadow()) return false;
    narrow_precond = true;
    return true;
  }
  void disable_narrow_precond() { narrow_precond = false; }

  bool is_initialized() { return is_init; }

  // fine += P coarse (:283-286, 455-480): accumulates, caller zeroes
  void prolong_c2f(complex<double>* coarse_cv, complex<double>* fine_cv) {
    qmg::ok(qmg_prolong(null_store, const_num_null_vec, coarse_cv, fine_cv, fine_lat->get_dim_mu(0), fine_lat->get_dim_mu(1), fine_lat->get_nc(),
                        coarse_lat->get_dim_mu(0), coarse_lat->get_dim_mu(1), coarse_lat->get_nc(), qmg::current_stream()), "qmg_prolong");
  }
  // coarse += R fine (:291-294, 487-511)
  void restrict_f2c(complex<double>* fine_cv, complex<double>* coarse_cv) {
    qmg::ok(qmg_restrict(restrict_store ? restrict_store : null_store, const_num_null_vec, fine_cv, coarse_cv, fine_lat->get_dim_mu(0),
                         fine_lat->get_dim_mu(1), fine_lat->get_nc(), coarse_lat->get_dim_mu(0), coarse_lat->get_dim_mu(1), coarse_lat->get_nc(),
                         qmg::current_stream()), "qmg_restrict");
  }

  // The same two maps for a lock-step batch of right-hand sides (qmg_batch.hip): the null vectors are streamed once per
  // 8 active systems instead of once per system.  Frozen systems (mask bit clear) are not touched.
  void prolong_c2f_batch(complex<double>* coarse, size_t cstride, complex<double>* fine, size_t fstride, int nrhs, unsigned mask) {
    qmg::ok(qmg_prolong_batch(null_store, const_num_null_vec, coarse, fine, fine_lat->get_dim_mu(0), fine_lat->get_dim_mu(1), fine_lat->get_nc(),
                              coarse_lat->get_dim_mu(0), coarse_lat->get_dim_mu(1), coarse_lat->get_nc(), nrhs, cstride, fstride, mask,
                              qmg::current_stream()), "qmg_prolong_batch");
  }
  void restrict_f2c_batch(complex<double>* fine, size_t fstride, complex<double>* coarse, size_t cstride, int nrhs, unsigned mask) {
    qmg::ok(qmg_restrict_batch(restrict_store ? restrict_store : null_store, const_num_null_vec, fine, coarse, fine_lat->get_dim_mu(0),
                               fine_lat->get_dim_mu(1), fine_lat->get_nc(), coarse_lat->get_dim_mu(0), coarse_lat->get_dim_mu(1), coarse_lat->get_nc(),
                               nrhs, fstride, cstride, mask, qmg::current_stream()), "qmg_restrict_batch");
  }

  // either storage precision (T = float: the fp32 shadow of the null vectors)
  template <typename T>
  void prolong_c2f_batch_t(complex<T>* coarse, size_t cstride, complex<T>* fine, size_t fstride, int nrhs, unsigned mask) {
    const bool f = sizeof(T) == sizeof(float);
    if (f && !null_store32) { std::cout << "[QMG-ERROR]: fp32 prolong without an fp32 shadow (TransferMG::enable_f32_shadow).\n"; return; }
    qmg::ok(qmg_prolong_batch_t(f ? QMG_C32 : QMG_C64, f ? (const void*)null_store32 : (const void*)null_store, const_num_null_vec, coarse, fine,
                                fine_lat->get_dim_mu(0), fine_lat->get_dim_mu(1), fine_lat->get_nc(), coarse_lat->get_dim_mu(0), coarse_lat->get_dim_mu(1),
                                coarse_lat->get_nc(), nrhs, cstride, fstride, mask, qmg::current_stream()), "qmg_prolong_batch_t");
  }
  template <typename T>
  void restrict_f2c_batch_t(complex<T>* fine, size_t fstride, complex<T>* coarse, size_t cstride, int nrhs, unsigned mask) {
    const bool f = sizeof(T) == sizeof(float);
    if (f && !null_store32) { std::cout << "[QMG-ERROR]: fp32 restrict without an fp32 shadow (TransferMG::enable_f32_shadow).\n"; return; }
    const void* r64 = restrict_store ? restrict_store : null_store;
    const void* r32 = restrict_store32 ? restrict_store32 : null_store32;
    qmg::ok(qmg_restrict_batch_t(f ? QMG_C32 : QMG_C64, f ? r32 : r64, const_num_null_vec, fine, coarse, fine_lat->get_dim_mu(0), fine_lat->get_dim_mu(1),
                                 fine_lat->get_nc(), coarse_lat->get_dim_mu(0), coarse_lat->get_dim_mu(1), coarse_lat->get_nc(), nrhs, fstride, cstride, mask,
                                 qmg::current_stream()), "qmg_restrict_batch_t");
  }

  // the K-cycle's own transfers (mg_preconditioner_batch): as the _batch_t forms, through the narrow copy where that is enabled and one system is active
  template <typename T>
  void prolong_c2f_precond_t(complex<T>* coarse, size_t cstride, complex<T>* fine, size_t fstride, int nrhs, unsigned mask) {
    if (sizeof(T) == sizeof(double) && narrow_precond && null_store32 && (mask & (mask - 1)) == 0 && mask != 0) {
      const int rc = qmg_prolong_batch_nv32(null_store32, const_num_null_vec, (const void*)coarse, (void*)fine, fine_lat->get_dim_mu(0), fine_lat->get_dim_mu(1), fine_lat->get_nc(),
                                            coarse_lat->get_dim_mu(0), coarse_lat->get_dim_mu(1), coarse_lat->get_nc(), nrhs, cstride, fstride, mask, qmg::current_stream());
      if (rc == QMG_SUCCESS) return;
      if (rc != QMG_ERR_UNSUPPORTED) { qmg::ok(rc, "qmg_prolong_batch_nv32"); return; }
    }
    prolong_c2f_batch_t<T>(coarse, cstride, fine, fstride, nrhs, mask);
  }
  template <typename T>
  void restrict_f2c_precond_t(complex<T>* fine, size_t fstride, complex<T>* coarse, size_t cstride, int nrhs, unsigned mask) {
    if (sizeof(T) == sizeof(double) && narrow_precond && null_store32 && (mask & (mask - 1)) == 0 && mask != 0) {
      const int rc = qmg_restrict_batch_nv32(restrict_store32 ? restrict_store32 : null_store32, const_num_null_vec, (const void*)fine, (void*)coarse, fine_lat->get_dim_mu(0),
                                             fine_lat->get_dim_mu(1), fine_lat->get_nc(), coarse_lat->get_dim_mu(0), coarse_lat->get_dim_mu(1), coarse_lat->get_nc(), nrhs, fstride,
                                             cstride, mask, qmg::current_stream());
      if (rc == QMG_SUCCESS) return;
      if (rc != QMG_ERR_UNSUPPORTED) { qmg::ok(rc, "qmg_restrict_batch_nv32"); return; }
    }
    restrict_f2c_batch_t<T>(fine, fstride, coarse, cstride, nrhs, mask);
  }

  bool is_symmetric() { return restrict_store == 0; }
  bool has_decompositions() { return is_symmetric() ? (block_cholesky != 0) : (block_L != 0 && block_U != 0); }
  void copy_cholesky(complex<double>* save_cholesky) {
    if (block_cholesky == 0) std::cout << "[QMG-WARNING]: In expose_cholesky, block Cholesky has not been computed.\n";
    else copy_vector(save_cholesky, block_cholesky, coarse_lat->get_size_cm_l());
  }
  void copy_LU(complex<double>* save_L, complex<double>* save_U) {
    if (block_L == 0 || block_U == 0) std::cout << "[QMG-WARNING]: In expose_LU, block LU has not been computed.\n";
    else { copy_vector(save_L, block_L, coarse_lat->get_size_cm_l()); copy_vector(save_U, block_U, coarse_lat->get_size_cm_l()); }
  }
  QMGDoublingType get_doubling() { return doubling; }

  // for the coarse-operator build
  const complex<double>* device_null_vectors() const { return null_store; }
  const complex<double>* device_restrict_vectors() const { return restrict_store; }
  Lattice2D* get_fine_lattice() { return fine_lat; }
  Lattice2D* get_coarse_lattice() { return coarse_lat; }
};

#endif
