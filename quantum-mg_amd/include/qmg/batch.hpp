// batch.hpp -- the K-cycle for a LOCK-STEP BATCH of independent right-hand sides (BASELINE configs[3]/[4]:
// "independent right-hand sides", several per GPU).
//
// The reference solves one system at a time (tests/n13_wilson_kcycle/wilson_kcycle.cpp:459-466), so k systems stream
// every coarse operator and every null vector k times.  Here up to 16 systems advance through the SAME iteration of the
// SAME solver together: one launch per step for the whole batch, the matrices / null vectors read once, and the coarse
// applies run as (nc x nc).(nc x k) contractions on the f64 matrix cores (qmg_stencil.hip kernel C).
//
// Semantics: each system sees exactly the iteration it would see alone.  Every scalar (alpha, residual norm, Gram-
// Schmidt coefficient, restart decision, inner tolerance) is per system; a system that converges inside a solve is
// FROZEN (its bit leaves the active mask: no kernel reads or writes it) while the rest continue.  The per-system
// arithmetic is that of krylov.hpp / multigrid.hpp line by line -- element-wise kernels and reductions are
// bit-identical to the single-vector ones, the MFMA apply and the blocked transfer differ in summation order only
// (1e-13), so a batched solve reproduces the single solves to solver tolerance with the same iteration counts (+-1).
//
// Scope: every configuration of stateful_multigrid.h:734-1060 -- fine_stencil_app in {ORIGINAL, RIGHT_JACOBI, RIGHT_SCHUR} with MR or CGNE
// smoothers and flexible-GCR intermediate solves; coarsest_stencil_app one of those (GCR) or one of the four normal-equation operators
// (CG, normal_shift).  A hierarchy whose types name a variant stencil that is not built is rejected loudly (BatchKcycle::supported), not emulated.
//
// Storage precision: every type and function here is a template on the storage scalar T of the batch vectors (double |
// float).  T = double is the engine described above.  T = float is the fp32 instantiation of the path (BASELINE
// configs[4]): the same K-cycle on complex<float> vectors, streaming the complex<float> shadow copies of every level's
// matrices and null vectors (Stencil2D::enable_f32_shadow, TransferMG::enable_f32_shadow) through the QMG_C32 entry
// points of the C-ABI; all scalars, inner products and convergence decisions stay fp64.  It is used as the preconditioner
// of an fp64 flexible outer solve (mg_preconditioner_batch_mixed below), so the solution still reaches its fp64 tolerance.
#ifndef QMG_BATCH_HPP
#define QMG_BATCH_HPP

#include <cmath>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "krylov.hpp"
#include "multigrid.hpp"

namespace qmg {

const int BATCH_MAX = 16;

// The K-cycle's fixed-count smoothers with their scalars on the device (bmr_fixed_zero_guess).  QMG_KCYCLE_DEVICE_SCALARS=0 restores
// the host-scalar loops (bminv_vector_minres_zero_guess: one host round trip per MR iteration, residual recomputed): A/B runs.
inline bool kcycle_device_scalars() {
  static const bool on = !(getenv("QMG_KCYCLE_DEVICE_SCALARS") && atoi(getenv("QMG_KCYCLE_DEVICE_SCALARS")) == 0);
  return on;
}

template <typename T> struct dtype_of;
template <> struct dtype_of<double> { enum { value = QMG_C64 }; };
template <> struct dtype_of<float> { enum { value = QMG_C32 }; };

// nrhs vectors, `stride` complex elements apart
template <typename T>
struct BatchT {
  complex<T>* p;
  size_t stride;
  int nrhs;
  BatchT() : p(0), stride(0), nrhs(0) {}
  BatchT(complex<T>* p_, size_t stride_, int nrhs_) : p(p_), stride(stride_), nrhs(nrhs_) {}
  complex<T>* vec(int k) const { return p + (size_t)k * stride; }
};
typedef BatchT<double> Batch;

inline unsigned full_mask(int nrhs) { return (nrhs >= 32) ? 0xFFFFFFFFu : ((1u << nrhs) - 1u); }
inline bool is_active(unsigned mask, int k) { return (mask >> k) & 1u; }

// batch scratch, recycled like VecPool (whose unit is one complex<double>: an fp32 batch takes half as many units)
template <typename T>
struct BatchPoolT {
  VecPool pool;
  size_t stride;
  int nrhs;
  BatchPoolT(size_t n, int nrhs_) : pool((n * (size_t)nrhs_ * sizeof(complex<T>) + sizeof(complex<double>) - 1) / sizeof(complex<double>)), stride(n), nrhs(nrhs_) {}
  BatchT<T> get() { return BatchT<T>(reinterpret_cast<complex<T>*>(pool.get()), stride, nrhs); }
};
typedef BatchPoolT<double> BatchPool;

typedef std::vector<complex<double>> cvec;

template <typename T>
inline void bblas(int op, const cvec* a, const cvec* b, const BatchT<T>* x, const BatchT<T>* y, BatchT<T> z, size_t n, unsigned mask) {
  std::vector<double> fa, fb;
  if (a) { fa.resize(2 * z.nrhs); for (int k = 0; k < z.nrhs; k++) { fa[2 * k] = (*a)[k].real(); fa[2 * k + 1] = (*a)[k].imag(); } }
  if (b) { fb.resize(2 * z.nrhs); for (int k = 0; k < z.nrhs; k++) { fb[2 * k] = (*b)[k].real(); fb[2 * k + 1] = (*b)[k].imag(); } }
  ok(qmg_batch_blas_t(dtype_of<T>::value, op, a ? fa.data() : 0, b ? fb.data() : 0, x ? x->p : 0, y ? y->p : 0, z.p, n, z.nrhs, z.stride, mask, current_stream()), "qmg_batch_blas");
}
template <typename T> inline void bzero(BatchT<T> z, size_t n, unsigned mask) { bblas<T>(QMG_BOP_ZERO, 0, 0, 0, 0, z, n, mask); }
template <typename T> inline void bcopy(BatchT<T> z, BatchT<T> x, size_t n, unsigned mask) { bblas<T>(QMG_BOP_COPY, 0, 0, &x, 0, z, n, mask); }
template <typename T> inline void bcaxpy(const cvec& a, BatchT<T> x, BatchT<T> y, size_t n, unsigned mask) { bblas<T>(QMG_BOP_CAXPY, &a, 0, &x, 0, y, n, mask); }   // y += a x
template <typename T> inline void bcxpy(BatchT<T> x, BatchT<T> y, size_t n, unsigned mask) { bblas<T>(QMG_BOP_CXPY, 0, 0, &x, 0, y, n, mask); }                   // y += x
template <typename T> inline void bcaxpbyz(const cvec& a, BatchT<T> x, const cvec& b, BatchT<T> y, BatchT<T> z, size_t n, unsigned mask) { bblas<T>(QMG_BOP_CAXPBYZ, &a, &b, &x, &y, z, n, mask); }
template <typename T> inline void bxmyz(BatchT<T> x, BatchT<T> y, BatchT<T> z, size_t n, unsigned mask) {   // z = x - y
  const cvec one(z.nrhs, 1.0), mone(z.nrhs, -1.0);
  bcaxpbyz(one, x, mone, y, z, n, mask);
}
template <typename T> inline void bcxpyz(BatchT<T> x, BatchT<T> y, BatchT<T> z, size_t n, unsigned mask) {   // z = x + y
  const cvec one(z.nrhs, 1.0);
  bcaxpbyz(one, x, one, y, z, n, mask);
}

// per-system |x_k|^2; entries of frozen systems keep `fill`
template <typename T>
inline std::vector<double> bnorm2sq(BatchT<T> x, size_t n, unsigned mask, double fill = 0.0) {
  std::vector<double> raw(2 * x.nrhs, 0.0), out(x.nrhs, fill);
  ok(qmg_batch_reduce_t(dtype_of<T>::value, QMG_BRED_NORM2, x.p, 0, n, x.nrhs, x.stride, mask, raw.data(), current_stream()), "qmg_batch_reduce");
  for (int k = 0; k < x.nrhs; k++) if (is_active(mask, k)) out[k] = raw[2 * k];
  return out;
}
template <typename T>
inline std::vector<double> bdiffnorm2sq(BatchT<T> x, BatchT<T> y, size_t n, unsigned mask) {
  std::vector<double> raw(2 * x.nrhs, 0.0), out(x.nrhs, 0.0);
  ok(qmg_batch_reduce_t(dtype_of<T>::value, QMG_BRED_DIFFNORM2, x.p, y.p, n, x.nrhs, x.stride, mask, raw.data(), current_stream()), "qmg_batch_reduce");
  for (int k = 0; k < x.nrhs; k++) if (is_active(mask, k)) out[k] = raw[2 * k];
  return out;
}
// d[k][j] = <xs[j]_k, y_k>
template <typename T>
inline std::vector<cvec> bmultidot(const std::vector<BatchT<T> >& xs, int nj, BatchT<T> y, size_t n, unsigned mask) {
  std::vector<cvec> out(y.nrhs, cvec(nj, 0.0));
  int done = 0;
  while (done < nj) {   // the ABI takes up to 32 vector sets per call
    const int jj = (nj - done > 32) ? 32 : nj - done;
    std::vector<const void*> ptrs(jj);
    for (int j = 0; j < jj; j++) ptrs[j] = xs[done + j].p;
    std::vector<double> raw((size_t)2 * y.nrhs * jj, 0.0);
    ok(qmg_batch_multidot_t(dtype_of<T>::value, ptrs.data(), jj, y.p, n, y.nrhs, y.stride, mask, raw.data(), current_stream()), "qmg_batch_multidot");
    for (int k = 0; k < y.nrhs; k++)
      if (is_active(mask, k))
        for (int j = 0; j < jj; j++) out[k][done + j] = complex<double>(raw[((size_t)k * jj + j) * 2], raw[((size_t)k * jj + j) * 2 + 1]);
    done += jj;
  }
  return out;
}
// y_k += sum_j c[k][j] xs[j]_k
template <typename T>
inline void bmulti_caxpy(const std::vector<cvec>& c, const std::vector<BatchT<T> >& xs, int nj, BatchT<T> y, size_t n, unsigned mask) {
  if (nj <= 0) return;
  std::vector<double> cf((size_t)2 * nj * y.nrhs, 0.0);
  std::vector<const void*> ptrs(nj);
  for (int j = 0; j < nj; j++) {
    ptrs[j] = xs[j].p;
    for (int k = 0; k < y.nrhs; k++) { cf[((size_t)j * y.nrhs + k) * 2] = c[k][j].real(); cf[((size_t)j * y.nrhs + k) * 2 + 1] = c[k][j].imag(); }
  }
  ok(qmg_batch_multi_caxpy_t(dtype_of<T>::value, cf.data(), ptrs.data(), nj, y.p, n, y.nrhs, y.stride, mask, current_stream()), "qmg_batch_multi_caxpy");
}
// one flexible-GCR iteration's vector updates in one pass: w_k += sum_j c[k][j] Ws[j]_k ; r_k += a[k] w_k ; z_next_k = r_k (z_next.p != 0)
template <typename T>
inline void bgcr_update(const std::vector<cvec>& c, const std::vector<BatchT<T> >& Ws, int nj, BatchT<T> w, const cvec& a, BatchT<T> r, BatchT<T> z_next, size_t n, unsigned mask) {
  std::vector<double> cf((size_t)2 * (nj > 0 ? nj : 1) * w.nrhs, 0.0), af((size_t)2 * w.nrhs, 0.0);
  std::vector<const void*> ptrs(nj > 0 ? nj : 1, (const void*)0);
  for (int j = 0; j < nj; j++) {
    ptrs[j] = Ws[j].p;
    for (int k = 0; k < w.nrhs; k++) { cf[((size_t)j * w.nrhs + k) * 2] = c[k][j].real(); cf[((size_t)j * w.nrhs + k) * 2 + 1] = c[k][j].imag(); }
  }
  for (int k = 0; k < w.nrhs; k++) { af[2 * k] = a[k].real(); af[2 * k + 1] = a[k].imag(); }
  ok(qmg_batch_gcr_update_t(dtype_of<T>::value, nj > 0 ? cf.data() : 0, nj > 0 ? ptrs.data() : 0, nj, w.p, af.data(), r.p, z_next.p, n, w.nrhs, w.stride, mask, current_stream()),
     "qmg_batch_gcr_update");
}
// z_k = x_k across storage precisions (round / widen), active systems only
template <typename TD, typename TS>
inline void bconvert(BatchT<TD> z, BatchT<TS> x, size_t n, unsigned mask) {
  for (int k = 0; k < z.nrhs; k++)
    if (is_active(mask, k)) ok(qmg_convert(z.vec(k), dtype_of<TD>::value, x.vec(k), dtype_of<TS>::value, n, current_stream()), "qmg_convert");
}

}  // namespace qmg

namespace qmg {
// How many systems of a K-cycle solve fit in the HBM that is free right now.  Per system the outer flexible GCR keeps
// 2 (restart or expected iterations) + ~8 vectors of level 0, every intermediate GCR 2 restart + ~12 of its level, and
// the coarsest GCR 2 restart + 4; 15 % head-room.  (4096^2 Wilson, restart 64: ~75 GB per system -- 3 per 288 GB GPU.)
inline int batch_systems_that_fit(StatefulMultigridMG* mg, int outer_basis, int want) {
  size_t free_b = 0, total_b = 0;
  if (qmg_mem_info(&free_b, &total_b) != QMG_SUCCESS) return 1;
  free_b += VecPool::cached_bytes();   // cached scratch is reused by the next solve (by capacity), not returned to the driver and re-requested
  double per_system = 0.0;
  const int nl = mg->get_num_levels();
  for (int i = 0; i < nl; i++) {
    int basis = outer_basis;
    if (i > 0 && i < nl - 1) { const int rf = mg->get_level_solve(i)->intermediate_restart_freq, it = mg->get_level_solve(i)->intermediate_iters; basis = (rf > 0 && rf < it) ? rf : it; }
    if (i == nl - 1 && nl > 1) { const int rf = mg->get_coarsest_solve()->coarsest_restart_freq, it = mg->get_coarsest_solve()->coarsest_iters; basis = (rf > 0 && rf < it) ? rf : it; }
    per_system += (2.0 * basis + 12.0) * (double)mg->get_lattice(i)->get_size_cv_l() * 16.0;
  }
  int fit = (int)(0.85 * (double)free_b / per_system);
  if (fit < 1) fit = 1;
  if (fit > want) fit = want;
  if (fit > BATCH_MAX) fit = BATCH_MAX;
  return fit;
}
}  // namespace qmg

// lhs_k = A rhs_k for the active systems
template <typename T> using batch_matrix_op_t = void (*)(qmg::BatchT<T> lhs, qmg::BatchT<T> rhs, unsigned mask, void* extra_data);
template <typename T> using batch_precond_op_t = void (*)(qmg::BatchT<T> lhs, qmg::BatchT<T> rhs, int size, unsigned mask, void* extra_data, inversion_verbose_struct* verb);
typedef batch_matrix_op_t<double> batch_matrix_op;
typedef batch_precond_op_t<double> batch_precond_op;

inline void apply_stencil_2D_M_batch(qmg::Batch lhs, qmg::Batch rhs, unsigned mask, void* extra_data) {
  ((Stencil2D*)extra_data)->apply_M_overwrite_batch(lhs.p, rhs.p, lhs.nrhs, lhs.stride, mask);
}

// ---- operator variants for a batch (stencil_2d.h:2418-2566): ORIGINAL, the right-block-Jacobi operator and its Schur complement on the K-cycle's levels;
// additionally the four normal-equation forms for the coarsest solve (CG) and the dagger forms the CGNE smoothers end with ----
struct BatchOp {
  Stencil2D* st;
  QMGStencilType type;
  complex<double> normal_shift;   // CoarsestSolveMG::normal_shift: added to a normal operator (shift_function, stateful_multigrid.h:724-729)
  size_t shift_length;
  BatchOp(Stencil2D* st_, QMGStencilType type_) : st(st_), type(type_), normal_shift(0.0), shift_length(0) {}
  // operator of a K-cycle level (smoothed with MR / CGNE, solved with flexible GCR)
  static bool supported(QMGStencilType t) { return t == QMG_MATVEC_ORIGINAL || t == QMG_MATVEC_RIGHT_JACOBI || t == QMG_MATVEC_RIGHT_SCHUR; }
  static bool is_normal(QMGStencilType t) { return t == QMG_MATVEC_M_MDAGGER || t == QMG_MATVEC_MDAGGER_M || t == QMG_MATVEC_RBJ_M_MDAGGER || t == QMG_MATVEC_RBJ_MDAGGER_M; }
  // which variant stencils an operator type needs built (and, for complex<float> vectors, shadowed)
  static bool variants_built(Stencil2D* st, QMGStencilType t) {
    if (!st) return false;
    switch (t) {
      case QMG_MATVEC_ORIGINAL: return true;
      case QMG_MATVEC_RIGHT_JACOBI: case QMG_MATVEC_RIGHT_SCHUR: return st->built_rbjacobi;
      case QMG_MATVEC_DAGGER: case QMG_MATVEC_M_MDAGGER: case QMG_MATVEC_MDAGGER_M: return st->built_dagger;
      case QMG_MATVEC_RBJ_DAGGER: case QMG_MATVEC_RBJ_M_MDAGGER: case QMG_MATVEC_RBJ_MDAGGER_M: return st->built_rbjacobi && st->built_rbj_dagger;
      default: return false;
    }
  }
};
template <typename T> inline qmg::BatchT<T> batch_odd_half(qmg::BatchT<T> v, size_t half) { return qmg::BatchT<T>(v.p + half, v.stride, v.nrhs); }

// lhs_e = rhs_e - D'_eo D'_oe rhs_e (apply_M_rbjacobi_schur, :1886-1908); only the even halves are read / written
template <typename T>
inline void apply_M_rbjacobi_schur_batch(Stencil2D* st, qmg::BatchT<T> lhs, qmg::BatchT<T> rhs, unsigned mask) {
  if (!st->built_rbjacobi) { std::cout << "[QMG-WARNING]: Tried to call apply_M_rbjacobi_schur, but the rbjacobi stencil has not been allocated.\n"; return; }
  const size_t cv = (size_t)st->lat->get_size_cv_l(), half = cv / 2;
  qmg::BatchPoolT<T> pool(lhs.stride, lhs.nrhs);
  qmg::BatchT<T> t = pool.get();
  st->launch_set_batch<T>(QMG_P_OE | QMG_P_ZERO_O, t.p, rhs.p, Stencil2D::QMG_ARR_RBJ_HOPPING, 0.0, 0.0, 0.0, lhs.nrhs, lhs.stride, mask);
  st->launch_set_batch<T>(QMG_P_EO | QMG_P_ZERO_E, t.p, t.p, Stencil2D::QMG_ARR_RBJ_HOPPING, 0.0, 0.0, 0.0, lhs.nrhs, lhs.stride, mask);
  qmg::bxmyz(rhs, t, lhs, half, mask);
}
// lhs_k = M rhs_k (ORIGINAL operator: clover + hopping + shifts), one read of the matrices for the batch
template <typename T>
inline void apply_M_overwrite_batch_t(Stencil2D* st, qmg::BatchT<T> lhs, qmg::BatchT<T> rhs, unsigned mask) {
  st->launch_set_batch<T>(QMG_P_ALL | QMG_P_ZERO, lhs.p, rhs.p, Stencil2D::QMG_ARR_ORIGINAL, st->shift, st->eo_shift, st->dof_shift, lhs.nrhs, lhs.stride, mask);
}
// lhs_k = (1 + H') rhs_k (apply_M_rbjacobi, stencil_2d.h:1818-1844): the identity clover as a unit shift, the identity matrices are never read
template <typename T>
inline bool apply_M_rbjacobi_batch(Stencil2D* st, qmg::BatchT<T> lhs, qmg::BatchT<T> rhs, unsigned mask) {
  if (!st->built_rbjacobi) { std::cout << "[QMG-WARNING]: Tried to call apply_M_rbjacobi (batch), but the rbjacobi stencil has not been allocated.\n"; return false; }
  st->launch_set_batch<T>(QMG_P_HOPPING | QMG_P_SHIFT | QMG_P_ZERO, lhs.p, rhs.p, Stencil2D::QMG_ARR_RBJ_HOPPING, 1.0, 0.0, 0.0, lhs.nrhs, lhs.stride, mask);
  return true;
}
// lhs_k = (1 + H')^dagger rhs_k (apply_M_rbj_dagger, :2265-2278)
template <typename T>
inline bool apply_M_rbj_dagger_batch(Stencil2D* st, qmg::BatchT<T> lhs, qmg::BatchT<T> rhs, unsigned mask) {
  if (!st->built_rbj_dagger) { std::cout << "[QMG-WARNING]: Tried to call apply_M_rbj_dagger (batch), but the right jacobi dagger stencil has not been allocated.\n"; return false; }
  if (sizeof(T) == sizeof(float) && (!st->f32.on || st->f32.rbj_dagger_hopping == 0)) {
    std::cout << "[QMG-ERROR]: fp32 right-block-Jacobi dagger apply without its fp32 shadow (build_rbj_dagger_stencil before enable_f32_shadow).\n";
    return false;
  }
  st->launch_set_batch<T>(QMG_P_HOPPING | QMG_P_SHIFT | QMG_P_ZERO, lhs.p, rhs.p, Stencil2D::QMG_ARR_RBJ_DAGGER, 1.0, 0.0, 0.0, lhs.nrhs, lhs.stride, mask);
  return true;
}
// the second factor of a normal operator: lhs = F2 (F1 rhs) with F1, F2 out of {M, M^dagger, M_rbj, M_rbj^dagger}
template <typename T>
inline bool apply_factor_batch(Stencil2D* st, QMGStencilType factor, qmg::BatchT<T> lhs, qmg::BatchT<T> rhs, unsigned mask) {
  switch (factor) {
    case QMG_MATVEC_ORIGINAL: apply_M_overwrite_batch_t<T>(st, lhs, rhs, mask); return true;
    case QMG_MATVEC_DAGGER: return st->apply_M_dagger_overwrite_batch_t<T>(lhs.p, rhs.p, lhs.nrhs, lhs.stride, mask);
    case QMG_MATVEC_RIGHT_JACOBI: return apply_M_rbjacobi_batch<T>(st, lhs, rhs, mask);
    case QMG_MATVEC_RBJ_DAGGER: return apply_M_rbj_dagger_batch<T>(st, lhs, rhs, mask);
    default: return false;
  }
}
// a normal operator's factors, applied right to left: type = second (first rhs)
inline void normal_factors(QMGStencilType type, QMGStencilType* first, QMGStencilType* second) {
  switch (type) {
    case QMG_MATVEC_M_MDAGGER: *first = QMG_MATVEC_DAGGER; *second = QMG_MATVEC_ORIGINAL; break;             // apply_M_M_dagger (:1424-1435)
    case QMG_MATVEC_MDAGGER_M: *first = QMG_MATVEC_ORIGINAL; *second = QMG_MATVEC_DAGGER; break;             // apply_M_dagger_M (:1400-1411)
    case QMG_MATVEC_RBJ_M_MDAGGER: *first = QMG_MATVEC_RBJ_DAGGER; *second = QMG_MATVEC_RIGHT_JACOBI; break; // apply_M_rbjacobi_MMD (:2354-2371)
    default: *first = QMG_MATVEC_RIGHT_JACOBI; *second = QMG_MATVEC_RBJ_DAGGER; break;                       // apply_M_rbjacobi_MDM (:2282-2299)
  }
}
template <typename T>
inline void apply_normal_batch(BatchOp* op, qmg::BatchT<T> lhs, qmg::BatchT<T> rhs, unsigned mask) {
  QMGStencilType f1, f2;
  normal_factors(op->type, &f1, &f2);
  qmg::BatchPoolT<T> pool(lhs.stride, lhs.nrhs);
  qmg::BatchT<T> t = pool.get();
  if (!t.p || !apply_factor_batch<T>(op->st, f1, t, rhs, mask)) return;
  apply_factor_batch<T>(op->st, f2, lhs, t, mask);
  if (op->normal_shift != 0.0) qmg::bcaxpy(qmg::cvec(lhs.nrhs, op->normal_shift), rhs, lhs, op->shift_length, mask);
}
template <typename T>
inline void apply_stencil_typed_batch(qmg::BatchT<T> lhs, qmg::BatchT<T> rhs, unsigned mask, void* extra_data) {
  BatchOp* op = (BatchOp*)extra_data;
  if (op->type == QMG_MATVEC_RIGHT_SCHUR) apply_M_rbjacobi_schur_batch<T>(op->st, lhs, rhs, mask);
  else if (BatchOp::is_normal(op->type)) apply_normal_batch<T>(op, lhs, rhs, mask);
  else if (op->type == QMG_MATVEC_ORIGINAL) apply_M_overwrite_batch_t<T>(op->st, lhs, rhs, mask);
  else apply_factor_batch<T>(op->st, op->type, lhs, rhs, mask);   // DAGGER, RIGHT_JACOBI, RBJ_DAGGER
}
// ---- the same applies with an EPILOGUE (Stencil2D::launch_set_epi), system by system, when ONE system is active: the BLAS-1 pass that
// would follow the apply (residual, Schur combination, MR dots) happens on the finished site values inside the apply's launch.  A batch
// of several active systems keeps the shared-matrix batch kernels and the separate passes.
namespace qmg {
inline int single_active(unsigned mask, int nrhs) {   // the index of the one active system, or -1
  int k = -1;
  for (int i = 0; i < nrhs; i++) if (is_active(mask, i)) { if (k >= 0) return -1; k = i; }
  return k;
}
}  // namespace qmg
// out = b - A x (dotv == 0), or p = A r with the MR dots <p,r>, <p,p> left in the device slot (b == 0, mr_dots): true if done in fused launches
template <typename T>
inline bool apply_op_fused(BatchOp* op, qmg::BatchT<T> out, qmg::BatchT<T> x, const qmg::BatchT<T>* b, bool mr_dots, unsigned mask) {
  const int k = qmg::single_active(mask, out.nrhs);
  if (k < 0) return false;
  Stencil2D* st = op->st;
  qmg_apply_epilogue e;
  if (op->type == QMG_MATVEC_ORIGINAL) {
    // b - A x: out = 1 b + (-1) acc ; MR: out = acc, dots against x (= r)
    e.other = b ? (const void*)b->p : 0; e.other_scale = 1.0; e.acc_scale = b ? -1.0 : 1.0; e.dotv = mr_dots ? (const void*)x.p : 0;
    return st->launch_set_epi<T>(QMG_P_ALL | QMG_P_ZERO, out.p, x.p, Stencil2D::QMG_ARR_ORIGINAL, st->shift, st->eo_shift, st->dof_shift, out.stride, k, e);
  }
  if (op->type == QMG_MATVEC_RIGHT_JACOBI) {   // (1 + H') x: the same two forms on the right-block-Jacobi hops, the identity clover as a unit shift
    if (!st->built_rbjacobi) return false;
    e.other = b ? (const void*)b->p : 0; e.other_scale = 1.0; e.acc_scale = b ? -1.0 : 1.0; e.dotv = mr_dots ? (const void*)x.p : 0;
    return st->launch_set_epi<T>(QMG_P_HOPPING | QMG_P_SHIFT | QMG_P_ZERO, out.p, x.p, Stencil2D::QMG_ARR_RBJ_HOPPING, 1.0, 0.0, 0.0, out.stride, k, e);
  }
  if (op->type == QMG_MATVEC_M_MDAGGER || op->type == QMG_MATVEC_RBJ_M_MDAGGER) {   // A (A^dagger x): the MR dots of the CGNE smoother (against x) ride on the second apply
    if (b || !mr_dots || op->normal_shift != 0.0) return false;
    const bool rbj = op->type == QMG_MATVEC_RBJ_M_MDAGGER;
    qmg::BatchPoolT<T> pool(out.stride, out.nrhs);
    qmg::BatchT<T> t = pool.get();
    if (!t.p || !apply_factor_batch<T>(st, rbj ? QMG_MATVEC_RBJ_DAGGER : QMG_MATVEC_DAGGER, t, x, mask)) return false;
    e.other = 0; e.other_scale = 0.0; e.acc_scale = 1.0; e.dotv = (const void*)x.p;
    if (rbj ? st->launch_set_epi<T>(QMG_P_HOPPING | QMG_P_SHIFT | QMG_P_ZERO, out.p, t.p, Stencil2D::QMG_ARR_RBJ_HOPPING, 1.0, 0.0, 0.0, out.stride, k, e)
            : st->launch_set_epi<T>(QMG_P_ALL | QMG_P_ZERO, out.p, t.p, Stencil2D::QMG_ARR_ORIGINAL, st->shift, st->eo_shift, st->dof_shift, out.stride, k, e)) return true;
    apply_factor_batch<T>(st, rbj ? QMG_MATVEC_RIGHT_JACOBI : QMG_MATVEC_ORIGINAL, out, t, mask);
    qmg::ok(qmg_batch_mr_dots_t(qmg::dtype_of<T>::value, x.p, out.p, (size_t)st->lat->get_size_cv_l(), out.nrhs, out.stride, mask, qmg::current_stream()), "qmg_batch_mr_dots");
    return true;
  }
  if (op->type != QMG_MATVEC_RIGHT_SCHUR || !st->built_rbjacobi) return false;
  // Schur: A x = x_e - D'_eo D'_oe x_e.  First half plain (t_o = D'_oe x_e), second half with the epilogue on the even sites:
  //   A x      = 1 x_e + (-1) D'_eo t          (MR: dots against x_e)
  //   b - A x  = (b_e - x_e) + D'_eo t  -- two `other` vectors: not one epilogue; the residual form is left to the separate passes
  if (b) return false;
  qmg::BatchPoolT<T> pool(out.stride, out.nrhs);
  qmg::BatchT<T> t = pool.get();
  if (!t.p) return false;
  st->launch_set_batch<T>(QMG_P_OE | QMG_P_ZERO_O, t.p, x.p, Stencil2D::QMG_ARR_RBJ_HOPPING, 0.0, 0.0, 0.0, out.nrhs, out.stride, mask);
  e.other = x.p; e.other_scale = 1.0; e.acc_scale = -1.0; e.dotv = mr_dots ? (const void*)x.p : 0;
  if (st->launch_set_epi<T>(QMG_P_EO | QMG_P_ZERO_E, out.p, t.p, Stencil2D::QMG_ARR_RBJ_HOPPING, 0.0, 0.0, 0.0, out.stride, k, e)) return true;
  // not served: finish the unfused way (t_o is already there)
  st->launch_set_batch<T>(QMG_P_EO | QMG_P_ZERO_E, t.p, t.p, Stencil2D::QMG_ARR_RBJ_HOPPING, 0.0, 0.0, 0.0, out.nrhs, out.stride, mask);
  qmg::bxmyz(x, t, out, (size_t)st->lat->get_size_cv_l() / 2, mask);
  if (mr_dots) qmg::ok(qmg_batch_mr_dots_t(qmg::dtype_of<T>::value, x.p, out.p, (size_t)st->lat->get_size_cv_l() / 2, out.nrhs, out.stride, mask, qmg::current_stream()), "qmg_batch_mr_dots");
  return true;
}
// out = b - A x over the operator's solve size, fused where served
template <typename T>
inline void apply_op_residual(BatchOp* op, qmg::BatchT<T> out, qmg::BatchT<T> x, qmg::BatchT<T> b, qmg::BatchT<T> scratch, size_t size_solve, unsigned mask) {
  if (apply_op_fused<T>(op, out, x, &b, false, mask)) return;
  apply_stencil_typed_batch<T>(scratch, x, mask, (void*)op);
  qmg::bxmyz(b, scratch, out, size_solve, mask);
}

// b_prep = prepare_M(b) (stencil_2d.h:2455-2490), b_prep OVERWRITTEN over the full vector
template <typename T>
inline void prepare_M_batch(Stencil2D* st, QMGStencilType type, qmg::BatchT<T> b_prep, qmg::BatchT<T> b, unsigned mask) {
  const size_t cv = (size_t)st->lat->get_size_cv_l(), half = cv / 2;
  if (type == QMG_MATVEC_RIGHT_SCHUR) {   // b_e - D'_eo b_o on the even half, zero on the odd half (:1912-1928)
    const int k1 = qmg::single_active(mask, b.nrhs);
    qmg_apply_epilogue e;
    e.other = b.p; e.other_scale = 1.0; e.acc_scale = -1.0; e.dotv = 0;
    if (k1 < 0 || !st->launch_set_epi<T>(QMG_P_EO | QMG_P_ZERO_E, b_prep.p, b.p, Stencil2D::QMG_ARR_RBJ_HOPPING, 0.0, 0.0, 0.0, b.stride, k1, e)) {
      st->launch_set_batch<T>(QMG_P_EO | QMG_P_ZERO_E, b_prep.p, b.p, Stencil2D::QMG_ARR_RBJ_HOPPING, 0.0, 0.0, 0.0, b.nrhs, b.stride, mask);
      qmg::bxmyz(b, b_prep, b_prep, half, mask);
    }
    qmg::bzero(batch_odd_half(b_prep, half), cv - half, mask);
  } else if (type == QMG_MATVEC_MDAGGER_M) apply_factor_batch<T>(st, QMG_MATVEC_DAGGER, b_prep, b, mask);            // M^dagger b (prepare_M_dagger_M, :1413-1422)
  else if (type == QMG_MATVEC_RBJ_MDAGGER_M) apply_factor_batch<T>(st, QMG_MATVEC_RBJ_DAGGER, b_prep, b, mask);      // M_rbj^dagger b (prepare_M_rbjacobi_MDM, :2301-2318)
  else qmg::bcopy(b_prep, b, cv, mask);
}
// x = reconstruct_M(y, b) (:2492-2527), x OVERWRITTEN
template <typename T>
inline void reconstruct_M_batch(Stencil2D* st, QMGStencilType type, qmg::BatchT<T> x, qmg::BatchT<T> y, qmg::BatchT<T> b, unsigned mask) {
  const size_t cv = (size_t)st->lat->get_size_cv_l(), half = cv / 2;
  auto cinv = [&](qmg::BatchT<T> out, qmg::BatchT<T> in) {   // out = C^-1 in (apply_M_rbjacobi_cinv, :1848-1866)
    st->launch_set_batch<T>(QMG_P_CLOVER | QMG_P_ZERO, out.p, in.p, Stencil2D::QMG_ARR_RBJ_CINV, 0.0, 0.0, 0.0, x.nrhs, x.stride, mask);
  };
  if (type == QMG_MATVEC_RIGHT_SCHUR) {   // (:1932-1957) t_o = b_o - D'_oe y_e ; t_e = y_e ; x = C^-1 t
    qmg::BatchPoolT<T> pool(x.stride, x.nrhs);
    qmg::BatchT<T> t = pool.get();
    const int k1 = qmg::single_active(mask, x.nrhs);
    qmg_apply_epilogue e;
    e.other = b.p; e.other_scale = 1.0; e.acc_scale = -1.0; e.dotv = 0;   // t_o = b_o - D'_oe y_e on the finished odd sites
    if (k1 < 0 || !st->launch_set_epi<T>(QMG_P_OE | QMG_P_ZERO_O, t.p, y.p, Stencil2D::QMG_ARR_RBJ_HOPPING, 0.0, 0.0, 0.0, x.stride, k1, e)) {
      st->launch_set_batch<T>(QMG_P_OE | QMG_P_ZERO_O, t.p, y.p, Stencil2D::QMG_ARR_RBJ_HOPPING, 0.0, 0.0, 0.0, x.nrhs, x.stride, mask);
      qmg::bxmyz(batch_odd_half(b, half), batch_odd_half(t, half), batch_odd_half(t, half), cv - half, mask);
    }
    qmg::bcopy(t, y, half, mask);
    cinv(x, t);
  } else if (type == QMG_MATVEC_RIGHT_JACOBI || type == QMG_MATVEC_RBJ_MDAGGER_M) cinv(x, y);   // x = C^-1 y (reconstruct_M_rbjacobi :1870-1882, _MDM :2319-2335)
  else if (type == QMG_MATVEC_M_MDAGGER) apply_factor_batch<T>(st, QMG_MATVEC_DAGGER, x, y, mask);   // x = M^dagger y (reconstruct_M_M_dagger, :1437-1446)
  else if (type == QMG_MATVEC_RBJ_M_MDAGGER) {                                                       // x = C^-1 M_rbj^dagger y (reconstruct_M_rbjacobi_MMD, :2373-2392)
    qmg::BatchPoolT<T> pool(x.stride, x.nrhs);
    qmg::BatchT<T> t = pool.get();
    if (t.p && apply_factor_batch<T>(st, QMG_MATVEC_RBJ_DAGGER, t, y, mask)) cinv(x, t);
  } else qmg::bcopy(x, y, cv, mask);
}

// ---------------------------------------------------------------------------------------------
// MR(omega) for a batch: minv_vector_minres of krylov.hpp per system, in lock step.  x0 = 0 is REQUIRED (every use in the
// K-cycle; the caller has zeroed phi): r0 = b.
// ---------------------------------------------------------------------------------------------
template <typename T>
inline std::vector<inversion_info> bminv_vector_minres_zero_guess(qmg::BatchT<T> phi, qmg::BatchT<T> phi0, int size, int max_iter, double eps, double omega,
                                                                   batch_matrix_op_t<T> matrix_vector, void* extra_info, unsigned mask) {
  const int nrhs = phi.nrhs;
  std::vector<inversion_info> inv(nrhs);
  qmg::BatchPoolT<T> pool(phi.stride, nrhs);
  qmg::BatchT<T> r = pool.get(), p = pool.get();
  const std::vector<double> bsq = qmg::bnorm2sq(phi0, size, mask);
  qmg::bcopy(r, phi0, size, mask);
  std::vector<double> rsq = bsq, rsq_ref = bsq, bnorm(nrhs);
  std::vector<int> its(nrhs, 0), ops(nrhs, 0);
  std::vector<bool> conv(nrhs, false);
  unsigned act = 0;
  for (int k = 0; k < nrhs; k++) {
    bnorm[k] = std::sqrt(bsq[k]);
    if (!qmg::is_active(mask, k)) continue;
    conv[k] = (bnorm[k] == 0.0) || (std::sqrt(rsq[k]) < eps * bnorm[k]);
    if (!conv[k] && max_iter > 0) act |= 1u << k;
  }
  std::vector<qmg::BatchT<T> > rp(2);
  rp[0] = r; rp[1] = p;
  while (act) {
    matrix_vector(p, r, act, extra_info);
    const std::vector<qmg::cvec> d2 = qmg::bmultidot(rp, 2, p, size, act);
    qmg::cvec alpha(nrhs, 0.0), malpha(nrhs, 0.0);
    unsigned upd = 0, renorm = 0;
    for (int k = 0; k < nrhs; k++) {
      if (!qmg::is_active(act, k)) continue;
      ops[k]++;
      const complex<double> pr = std::conj(d2[k][0]);
      const double pp = d2[k][1].real();
      if (pp == 0.0) { act &= ~(1u << k); continue; }   // breakdown: this system stops (krylov.hpp `break`)
      alpha[k] = omega * pr / pp; malpha[k] = -alpha[k];
      upd |= 1u << k;
      rsq[k] = rsq[k] - (2.0 * omega - omega * omega) * std::norm(pr) / pp;
      if (!(rsq[k] > 1e-8 * rsq_ref[k]) || std::sqrt(rsq[k]) < 4.0 * eps * bnorm[k]) renorm |= 1u << k;
    }
    qmg::bcaxpy(alpha, r, phi, size, upd);
    // r is only needed by a further iteration or by a true-norm re-anchoring: the residual update of a system's LAST
    // iteration is skipped (the K-cycle recomputes b - A x itself); x and the returned analytic |r|^2 are unaffected
    unsigned need_r = renorm;
    for (int k = 0; k < nrhs; k++) if (qmg::is_active(upd, k) && its[k] + 1 < max_iter) need_r |= 1u << k;
    qmg::bcaxpy(malpha, p, r, size, upd & need_r);
    if (renorm) {
      const std::vector<double> t = qmg::bnorm2sq(r, size, renorm);
      for (int k = 0; k < nrhs; k++) if (qmg::is_active(renorm, k)) { rsq[k] = t[k]; rsq_ref[k] = t[k]; }
    }
    for (int k = 0; k < nrhs; k++) {
      if (!qmg::is_active(upd, k)) continue;
      its[k]++;
      if (std::sqrt(rsq[k]) < eps * bnorm[k]) { conv[k] = true; act &= ~(1u << k); }
      else if (its[k] >= max_iter) act &= ~(1u << k);
    }
  }
  for (int k = 0; k < nrhs; k++) { inv[k].success = conv[k]; inv[k].iter = its[k]; inv[k].resSq = rsq[k]; inv[k].ops_count = ops[k]; inv[k].name = "MinRes (batch)"; }
  return inv;
}

// ---------------------------------------------------------------------------------------------
// MR(omega) with a FIXED iteration count and every scalar on the device (qmg_batch_mr_dots_t / qmg_batch_mr_update_t): the form the
// K-cycle's smoothers take.  Their tolerance (1e-15 in n13 / n19 / n22, 1e-20 in LevelSolveMG's defaults) is below what the
// recursive residual of a few MR steps can reach in fp64, so minv_vector_minres always runs its `max_iter` iterations and NO host
// decision depends on <p,r> / <p,p>: alpha is formed on the device and the per-iteration host round trip disappears.  The
// arithmetic is that of bminv_vector_minres_zero_guess, operation for operation (same reduction order, same alpha = (omega <p,r>) / <p,p>).
//   x0 = 0 is implied: x is WRITTEN by the first step (x = alpha b; no zero fill, no read) -- with iters == 0, x = 0.
//   r_out (optional): the recursive residual b - A x after the last step (the K-cycle's pre-smoother wants it: it IS the residual the
//   reference recomputes with one more apply, stateful_multigrid.h:863-866, up to rounding); without it the last residual update is skipped.
// Returns the number of operator applications per active system.
// ---------------------------------------------------------------------------------------------
namespace qmg {
inline bool mr_tolerance_unreachable(double eps) { return eps <= 1e-14; }
// bgcr_core: the iteration's dots in one pass / one host round trip (see there).  QMG_GCR_FUSED=0 restores the three-pass form.
inline bool gcr_fused_dots() { static const bool on = !(getenv("QMG_GCR_FUSED") && atoi(getenv("QMG_GCR_FUSED")) == 0); return on; }
// ... and its vector updates in one pass (qmg_batch_gcr_update_t).  QMG_GCR_FUSED_UPDATE=0: the separate multi-axpy / axpy / copy passes.
inline bool gcr_fused_update() { static const bool on = !(getenv("QMG_GCR_FUSED_UPDATE") && atoi(getenv("QMG_GCR_FUSED_UPDATE")) == 0); return on; }
}  // namespace qmg
template <typename T>
inline int bmr_fixed_zero_guess(qmg::BatchT<T> x, qmg::BatchT<T> b, qmg::BatchT<T>* r_out, int size, int iters, double omega,
                                batch_matrix_op_t<T> matrix_vector, void* extra_info, unsigned mask, BatchOp* fused_op = 0) {
  if (iters <= 0) { qmg::bzero(x, (size_t)size, mask); if (r_out) qmg::bcopy(*r_out, b, (size_t)size, mask); return 0; }
  qmg::BatchPoolT<T> pool(x.stride, x.nrhs);
  qmg::BatchT<T> p = pool.get();
  qmg::BatchT<T> r = r_out ? *r_out : ((iters > 1) ? pool.get() : qmg::BatchT<T>());
  const int dt = qmg::dtype_of<T>::value;
  for (int it = 0; it < iters; it++) {
    const qmg::BatchT<T>& rin = (it == 0) ? b : r;
    if (!(fused_op && apply_op_fused<T>(fused_op, p, rin, (const qmg::BatchT<T>*)0, true, mask))) {   // p = A r and its dots in one pass, where served
      matrix_vector(p, rin, mask, extra_info);
      qmg::ok(qmg_batch_mr_dots_t(dt, rin.p, p.p, (size_t)size, x.nrhs, x.stride, mask, qmg::current_stream()), "qmg_batch_mr_dots");
    }
    const bool want_r = (it + 1 < iters) || r_out;
    qmg::ok(qmg_batch_mr_update_t(dt, omega, x.p, rin.p, want_r ? r.p : 0, p.p, it == 0, (size_t)size, x.nrhs, x.stride, mask, qmg::current_stream()), "qmg_batch_mr_update");
  }
  return iters;
}

// ---------------------------------------------------------------------------------------------
// BiCGStab(L) for a batch: minv_vector_bicgstab_l of krylov.hpp per system, in lock step (the null-vector relaxation of
// tests/n13_wilson_kcycle/wilson_kcycle.cpp:359, several null vectors at a time).  x0 = 0 is REQUIRED (the caller has
// zeroed phi): r0 = b.  `iter` counts BiCG steps per system; a system that converges, breaks down or reaches max_iter is
// frozen at the end of its L-block.  The closing updates of a block go through one multi-vector pass each.
// ---------------------------------------------------------------------------------------------
template <typename T>
inline std::vector<inversion_info> bminv_vector_bicgstab_l_zero_guess(qmg::BatchT<T> phi, qmg::BatchT<T> phi0, int size, int max_iter, double eps, int L,
                                                                      batch_matrix_op_t<T> matrix_vector, void* extra_info, unsigned mask) {
  const int nrhs = phi.nrhs;
  std::vector<inversion_info> inv(nrhs);
  qmg::BatchPoolT<T> pool(phi.stride, nrhs);
  std::vector<qmg::BatchT<T> > r(L + 1), u(L + 1);
  for (int i = 0; i <= L; i++) { r[i] = pool.get(); u[i] = pool.get(); }
  qmg::BatchT<T> rt = pool.get();
  const std::vector<double> bsq = qmg::bnorm2sq(phi0, size, mask);
  qmg::bcopy(r[0], phi0, size, mask);
  qmg::bcopy(rt, phi0, size, mask);
  qmg::bzero(u[0], size, mask);
  std::vector<double> rsq = bsq, bnorm(nrhs, 0.0);
  std::vector<int> its(nrhs, 0), ops(nrhs, 0);
  std::vector<bool> conv(nrhs, false);
  qmg::cvec rho0(nrhs, 1.0), alpha(nrhs, 0.0), omega(nrhs, 1.0);
  unsigned act = 0;
  for (int k = 0; k < nrhs; k++) {
    bnorm[k] = std::sqrt(bsq[k]);
    if (!qmg::is_active(mask, k)) continue;
    conv[k] = (bnorm[k] == 0.0) || (std::sqrt(rsq[k]) < eps * bnorm[k]);
    if (!conv[k] && max_iter > 0) act |= 1u << k;
  }
  const qmg::cvec one(nrhs, 1.0);
  std::vector<qmg::BatchT<T> > single(1);
  auto bdot1 = [&](qmg::BatchT<T> a, qmg::BatchT<T> b, unsigned m) {   // <a_k, b_k> per system
    single[0] = a;
    const std::vector<qmg::cvec> d = qmg::bmultidot(single, 1, b, size, m);
    qmg::cvec out(nrhs, 0.0);
    for (int k = 0; k < nrhs; k++) out[k] = d[k][0];
    return out;
  };
  std::vector<qmg::cvec> tau(nrhs, qmg::cvec((L + 1) * (L + 1), 0.0)), gamma(nrhs, qmg::cvec(L + 1, 0.0)), gammap(nrhs, qmg::cvec(L + 1, 0.0)),
      gammapp(nrhs, qmg::cvec(L + 1, 0.0));
  std::vector<std::vector<double> > sigma(nrhs, std::vector<double>(L + 1, 0.0));
  while (act) {
    for (int k = 0; k < nrhs; k++) if (qmg::is_active(act, k)) rho0[k] = -omega[k] * rho0[k];
    for (int j = 0; j < L && act; j++) {   // BiCG part
      const qmg::cvec rho1 = bdot1(rt, r[j], act);
      qmg::cvec mbeta(nrhs, 0.0);
      for (int k = 0; k < nrhs; k++) {
        if (!qmg::is_active(act, k)) continue;
        if (rho0[k] == 0.0) { act &= ~(1u << k); continue; }                 // breakdown: this system stops
        mbeta[k] = -(alpha[k] * rho1[k] / rho0[k]);
        rho0[k] = rho1[k];
      }
      if (!act) break;
      for (int i = 0; i <= j; i++) qmg::bcaxpbyz(one, r[i], mbeta, u[i], u[i], size, act);   // u_i = r_i - beta u_i
      matrix_vector(u[j + 1], u[j], act, extra_info);
      const qmg::cvec gam = bdot1(rt, u[j + 1], act);
      qmg::cvec malpha(nrhs, 0.0);
      for (int k = 0; k < nrhs; k++) {
        if (!qmg::is_active(act, k)) continue;
        ops[k]++;
        if (gam[k] == 0.0) { act &= ~(1u << k); continue; }
        alpha[k] = rho0[k] / gam[k];
        malpha[k] = -alpha[k];
      }
      if (!act) break;
      for (int i = 0; i <= j; i++) qmg::bcaxpy(malpha, u[i + 1], r[i], size, act);
      matrix_vector(r[j + 1], r[j], act, extra_info);
      qmg::bcaxpy(alpha, u[0], phi, size, act);
      for (int k = 0; k < nrhs; k++) if (qmg::is_active(act, k)) { ops[k]++; its[k]++; }
    }
    if (!act) break;
    for (int j = 1; j <= L && act; j++) {   // MR part: modified Gram-Schmidt on r_1..r_L
      for (int i = 1; i < j; i++) {
        const qmg::cvec d = bdot1(r[i], r[j], act);
        qmg::cvec mt(nrhs, 0.0);
        for (int k = 0; k < nrhs; k++) if (qmg::is_active(act, k)) { tau[k][i * (L + 1) + j] = d[k] / sigma[k][i]; mt[k] = -tau[k][i * (L + 1) + j]; }
        qmg::bcaxpy(mt, r[i], r[j], size, act);
      }
      std::vector<qmg::BatchT<T> > two(2);
      two[0] = r[j]; two[1] = r[0];
      // <r_j, r_j> and <r_j, r_0> in one pass over r_j: d[k][0] = <r_j, r_j>, d[k][1] = <r_0, r_j> = conj <r_j, r_0>
      const std::vector<qmg::cvec> d = qmg::bmultidot(two, 2, r[j], size, act);
      for (int k = 0; k < nrhs; k++) {
        if (!qmg::is_active(act, k)) continue;
        sigma[k][j] = d[k][0].real();
        if (sigma[k][j] == 0.0) { act &= ~(1u << k); continue; }
        gammap[k][j] = std::conj(d[k][1]) / sigma[k][j];
      }
    }
    if (!act) break;
    std::vector<qmg::cvec> cx(nrhs, qmg::cvec(L, 0.0)), cr(nrhs, qmg::cvec(L, 0.0)), cu(nrhs, qmg::cvec(L, 0.0));
    for (int k = 0; k < nrhs; k++) {
      if (!qmg::is_active(act, k)) continue;
      qmg::cvec &g = gamma[k], &gp = gammap[k], &gpp = gammapp[k], &t = tau[k];
      g[L] = gp[L];
      omega[k] = g[L];
      for (int j = L - 1; j >= 1; j--) {
        g[j] = gp[j];
        for (int i = j + 1; i <= L; i++) g[j] -= t[j * (L + 1) + i] * g[i];
      }
      for (int j = 1; j < L; j++) {
        gpp[j] = g[j + 1];
        for (int i = j + 1; i < L; i++) gpp[j] += t[j * (L + 1) + i] * g[i + 1];
      }
      // x += gamma_1 r_0 + sum_{j<L} gamma''_j r_j ; r_0 -= sum_{j<=L} gamma'_j r_j ; u_0 -= sum_{j<=L} gamma_j u_j
      cx[k][0] = g[1];
      for (int j = 1; j < L; j++) cx[k][j] = gpp[j];
      for (int j = 1; j <= L; j++) { cr[k][j - 1] = -gp[j]; cu[k][j - 1] = -g[j]; }
    }
    std::vector<qmg::BatchT<T> > r0L(r.begin(), r.begin() + L), r1L(r.begin() + 1, r.end()), u1L(u.begin() + 1, u.end());
    qmg::bmulti_caxpy(cx, r0L, L, phi, size, act);     // reads r_0 before it changes
    qmg::bmulti_caxpy(cr, r1L, L, r[0], size, act);
    qmg::bmulti_caxpy(cu, u1L, L, u[0], size, act);
    const std::vector<double> t2 = qmg::bnorm2sq(r[0], size, act);
    for (int k = 0; k < nrhs; k++) {
      if (!qmg::is_active(act, k)) continue;
      rsq[k] = t2[k];
      if (std::sqrt(rsq[k]) < eps * bnorm[k]) { conv[k] = true; act &= ~(1u << k); }
      else if (its[k] >= max_iter) act &= ~(1u << k);
    }
  }
  for (int k = 0; k < nrhs; k++) { inv[k].success = conv[k]; inv[k].iter = its[k]; inv[k].resSq = rsq[k]; inv[k].ops_count = ops[k]; inv[k].name = "BiCGStab-L (batch)"; }
  return inv;
}

// ---------------------------------------------------------------------------------------------
// Flexible GCR with restarts for a batch: qmg_gcr_core of krylov.hpp per system, in lock step.  All systems start
// together, so the basis index kb (and with it the restart points) is common; everything else is per system.
// zero_guess: the caller has zeroed phi, r0 = b (krylov.hpp ZeroGuess).
// ---------------------------------------------------------------------------------------------
template <typename T>
inline std::vector<inversion_info> bgcr_core(qmg::BatchT<T> phi, qmg::BatchT<T> phi0, int size, int max_iter, double eps, int restart_freq,
                                             batch_matrix_op_t<T> matrix_vector, void* extra_info, batch_precond_op_t<T> precond, void* precond_info,
                                             unsigned mask, bool zero_guess, inversion_verbose_struct* verb, const char* name,
                                             const std::vector<double>* eps_per_system = 0) {
  const int nrhs = phi.nrhs;
  std::vector<inversion_info> inv(nrhs);
  std::vector<double> epsv(nrhs, eps);   // relative tolerance per system (the K-cycle's inner tolerance depends on the system)
  if (eps_per_system) epsv = *eps_per_system;
  const int basis_max = (restart_freq > 0) ? restart_freq : max_iter;
  qmg::BatchPoolT<T> pool(phi.stride, nrhs);
  qmg::BatchT<T> r = pool.get(), tmp = pool.get();
  std::vector<qmg::BatchT<T> > Z, W;        // raw search directions and orthogonalised images (krylov.hpp: z is not orthogonalised)
  std::vector<std::vector<double> > Wnorm2;   // [basis index][system]
  std::vector<std::vector<qmg::cvec> > C(nrhs);   // C[system][k][i]: Gram-Schmidt coefficients of this cycle
  std::vector<qmg::cvec> alphas(nrhs);           // alphas[system][k]
  std::vector<int> used(nrhs, 0);                // directions system k has taken in this cycle
  auto flush_x = [&]() {                         // x_k += sum_j y_kj z_j for every system with pending directions
    int K = 0;
    unsigned m = 0;
    for (int k = 0; k < nrhs; k++) if (used[k] > 0) { m |= 1u << k; if (used[k] > K) K = used[k]; }
    if (!m) return;
    std::vector<qmg::cvec> y(nrhs, qmg::cvec(K, 0.0));
    for (int k = 0; k < nrhs; k++) {
      if (used[k] <= 0) continue;
      const qmg::cvec yk = qmg::gcr_direction_weights(alphas[k], C[k], used[k]);
      for (int j = 0; j < used[k]; j++) y[k][j] = yk[j];
      used[k] = 0;
    }
    qmg::bmulti_caxpy(y, Z, K, phi, size, m);
  };
  const std::vector<double> bsq = qmg::bnorm2sq(phi0, size, mask);
  std::vector<double> rsq(nrhs, 0.0), rsq_ref(nrhs, 0.0), bnorm(nrhs, 0.0);
  std::vector<int> its(nrhs, 0), ops(nrhs, 0);
  std::vector<bool> conv(nrhs, false);
  if (zero_guess) { qmg::bcopy(r, phi0, size, mask); rsq = bsq; }
  else {
    matrix_vector(tmp, phi, mask, extra_info);
    for (int k = 0; k < nrhs; k++) if (qmg::is_active(mask, k)) ops[k]++;
    qmg::bxmyz(phi0, tmp, r, size, mask);
    rsq = qmg::bnorm2sq(r, size, mask);
  }
  unsigned act = 0;
  for (int k = 0; k < nrhs; k++) {
    bnorm[k] = std::sqrt(bsq[k]);
    rsq_ref[k] = rsq[k];
    if (!qmg::is_active(mask, k)) continue;
    conv[k] = (bnorm[k] == 0.0) || (std::sqrt(rsq[k]) < epsv[k] * bnorm[k]);
    if (!conv[k] && max_iter > 0) act |= 1u << k;
  }
  int kb = 0;
  bool z_ready = false;
  inversion_verbose_struct pverb(verb ? verb->precond_verbosity : VERB_NONE, verb ? verb->precond_verb_prefix : std::string(""));
  if (verb) { pverb.precond_verbosity = verb->precond_verbosity; pverb.precond_verb_prefix = verb->precond_verb_prefix; }
  std::vector<qmg::BatchT<T> > rw(2);
  while (act) {
    if (kb == (int)Z.size()) { Z.push_back(pool.get()); W.push_back(pool.get()); Wnorm2.push_back(std::vector<double>(nrhs, 0.0)); }
    for (int k = 0; k < nrhs; k++) { if ((int)C[k].size() <= kb) { C[k].push_back(qmg::cvec()); alphas[k].push_back(0.0); } }
    qmg::BatchT<T> z = Z[kb], w = W[kb];
    if (z.p == 0 || w.p == 0 || r.p == 0 || tmp.p == 0) {   // out of HBM: stop, report every active system as not converged
      std::cout << "[QMG-ERROR]: " << name << ": could not allocate basis vector " << kb << " for a batch of " << nrhs << " systems; size the batch with qmg::batch_systems_that_fit.\n";
      break;
    }
    if (precond) { qmg::bzero(z, size, act); precond(z, r, size, act, precond_info, &pverb); }
    else if (!z_ready) qmg::bcopy(z, r, size, act);   // (z_ready: the previous iteration's update pass wrote z = r already)
    z_ready = false;
    matrix_vector(w, z, act, extra_info);
    // ONE reduction pass and one host round trip per iteration (qmg::gcr_fused_dots(); QMG_GCR_FUSED=0: the three-pass form): the Gram-Schmidt
    // coefficients c_i = <W_i, w>, <r, w> and <w, w> come from the same pass over the RAW w; for the orthogonalised w' = w - sum_i (c_i / N_i) W_i
    //   <w', w'> = <w, w> - sum_i |c_i|^2 / N_i          (the W_i are orthogonal)
    //   <r,  w'> = <r, w>                                (r is orthogonal to every W_i of the cycle: each step removed that component)
    // A system whose w' keeps less than 1e-6 of |w|^2 (w almost inside the span: the subtraction has lost its digits) takes the explicit dots.
    std::vector<qmg::cvec> d2(nrhs, qmg::cvec(2, 0.0));
    unsigned explicit_dots = act;
    // With the dots of the fused form alpha is known BEFORE w is orthogonalised, so the Gram-Schmidt update of w, the residual update and (without a
    // preconditioner) the copy z_next = r go through ONE pass (qmg_batch_gcr_update_t: the same bits as the three separate passes).
    bool deferred = false;
    std::vector<qmg::cvec> cdef;
    if (qmg::gcr_fused_dots()) {
      std::vector<qmg::BatchT<T> > basis(W.begin(), W.begin() + kb);
      basis.push_back(r); basis.push_back(w);
      std::vector<qmg::cvec> c = qmg::bmultidot(basis, kb + 2, w, size, act);
      explicit_dots = 0;
      for (int k = 0; k < nrhs; k++) {
        if (!qmg::is_active(act, k)) continue;
        double ww = c[k][kb + 1].real();
        const double ww_raw = ww;
        for (int i = 0; i < kb; i++) { ww -= std::norm(c[k][i]) / Wnorm2[i][k]; c[k][i] = -c[k][i] / Wnorm2[i][k]; }
        d2[k][0] = c[k][kb]; d2[k][1] = ww;
        if (!(ww > 1e-6 * ww_raw)) explicit_dots |= 1u << k;
        c[k].resize(kb);
        C[k][kb] = c[k];
      }
      if (explicit_dots == 0 && qmg::gcr_fused_update()) { deferred = true; cdef = c; }
      else if (kb > 0) qmg::bmulti_caxpy(c, W, kb, w, size, act);
    } else if (kb > 0) {
      std::vector<qmg::cvec> c = qmg::bmultidot(W, kb, w, size, act);
      for (int k = 0; k < nrhs; k++)
        if (qmg::is_active(act, k))
          for (int i = 0; i < kb; i++) c[k][i] = -c[k][i] / Wnorm2[i][k];
      qmg::bmulti_caxpy(c, W, kb, w, size, act);
      for (int k = 0; k < nrhs; k++) if (qmg::is_active(act, k)) C[k][kb] = c[k];
    }
    if (explicit_dots) {
      rw[0] = r; rw[1] = w;
      const std::vector<qmg::cvec> e2 = qmg::bmultidot(rw, 2, w, size, explicit_dots);
      for (int k = 0; k < nrhs; k++) if (qmg::is_active(explicit_dots, k)) d2[k] = e2[k];
    }
    qmg::cvec alpha(nrhs, 0.0), malpha(nrhs, 0.0);
    unsigned upd = 0, renorm = 0;
    // the true norm re-anchors the recurrence when it has lost digits and CONFIRMS a convergence the recurrence announces (fused form; the
    // three-pass form keeps its wider band of 4 eps)
    const double band = qmg::gcr_fused_dots() ? 1.0 : 4.0;
    for (int k = 0; k < nrhs; k++) {
      if (!qmg::is_active(act, k)) continue;
      ops[k]++;
      const double ww = d2[k][1].real();
      if (ww == 0.0) { act &= ~(1u << k); continue; }
      Wnorm2[kb][k] = ww;
      const complex<double> wr = std::conj(d2[k][0]);
      alpha[k] = wr / ww; malpha[k] = -alpha[k];
      alphas[k][kb] = alpha[k];
      used[k] = kb + 1;
      upd |= 1u << k;
      rsq[k] = rsq[k] - std::norm(wr) / ww;
      if (!(rsq[k] > 1e-8 * rsq_ref[k]) || std::sqrt(rsq[k]) < band * epsv[k] * bnorm[k]) renorm |= 1u << k;
    }
    if (deferred) {
      qmg::BatchT<T> z_next;
      if (!precond && kb + 1 < basis_max) {
        if (kb + 1 == (int)Z.size()) { Z.push_back(pool.get()); W.push_back(pool.get()); Wnorm2.push_back(std::vector<double>(nrhs, 0.0)); }
        z_next = Z[kb + 1];
      }
      qmg::bgcr_update(cdef, W, kb, w, malpha, r, z_next, size, upd);
      z_ready = z_next.p != 0;
    } else qmg::bcaxpy(malpha, w, r, size, upd);
    if (renorm) {
      const std::vector<double> t = qmg::bnorm2sq(r, size, renorm);
      for (int k = 0; k < nrhs; k++) if (qmg::is_active(renorm, k)) { rsq[k] = t[k]; rsq_ref[k] = t[k]; }
    }
    kb++;
    for (int k = 0; k < nrhs; k++) {
      if (!qmg::is_active(upd, k)) continue;
      its[k]++;
      if (verb && verb->verbosity == VERB_DETAIL) { std::cout << verb->verb_prefix << name; if (nrhs > 1) std::cout << " rhs " << k; std::cout << " Iter " << its[k] << " RelTol " << std::sqrt(rsq[k]) / bnorm[k] << "\n"; }
      if (std::sqrt(rsq[k]) < epsv[k] * bnorm[k]) { conv[k] = true; act &= ~(1u << k); }
    }
    if (kb == basis_max) flush_x();   // the basis is about to be reused: bring every pending x up to date (frozen systems too)
    if (act && kb == basis_max) {   // restart: true residual, drop the basis (before the iteration cap, as in krylov.hpp)
      matrix_vector(tmp, phi, act, extra_info);
      qmg::bxmyz(phi0, tmp, r, size, act);
      const std::vector<double> t = qmg::bnorm2sq(r, size, act);
      kb = 0;
      z_ready = false;
      for (int k = 0; k < nrhs; k++) {
        if (!qmg::is_active(act, k)) continue;
        ops[k]++;
        rsq[k] = t[k]; rsq_ref[k] = t[k];
        if (std::sqrt(rsq[k]) < epsv[k] * bnorm[k]) { conv[k] = true; act &= ~(1u << k); }
      }
    }
    for (int k = 0; k < nrhs; k++) if (qmg::is_active(act, k) && its[k] >= max_iter) act &= ~(1u << k);
  }
  flush_x();
  for (int k = 0; k < nrhs; k++) {
    inv[k].success = conv[k]; inv[k].iter = its[k]; inv[k].resSq = rsq[k]; inv[k].ops_count = ops[k]; inv[k].name = name;
    if (verb && verb->verbosity != VERB_NONE && qmg::is_active(mask, k)) {   // (one system: krylov.hpp's line, word for word)
      std::cout << verb->verb_prefix << name;
      if (nrhs > 1) std::cout << " rhs " << k;
      std::cout << (conv[k] ? " Success " : " Fail ") << "Iter " << its[k] << " RelTol " << (bnorm[k] > 0 ? std::sqrt(rsq[k]) / bnorm[k] : 0.0) << "\n";
    }
  }
  return inv;
}

// ---------------------------------------------------------------------------------------------
// CG with restarts for a batch: minv_vector_cg / minv_vector_cg_restart of krylov.hpp per system, in lock step (the coarsest solve on a
// normal-equation operator, stateful_multigrid.h:930-960).  Every active system starts each restart cycle together; a system that converges,
// breaks down (<p, A p> == 0) or reaches max_iter is frozen.  restart_freq <= 0: one cycle of max_iter iterations.
// zero_guess: the caller has zeroed phi, the first cycle's r0 = b.
// ---------------------------------------------------------------------------------------------
template <typename T>
inline std::vector<inversion_info> bcg_core(qmg::BatchT<T> phi, qmg::BatchT<T> phi0, int size, int max_iter, double eps, int restart_freq,
                                            batch_matrix_op_t<T> matrix_vector, void* extra_info, unsigned mask, bool zero_guess,
                                            inversion_verbose_struct* verb, const char* name, const std::vector<double>* eps_per_system = 0) {
  const int nrhs = phi.nrhs;
  std::vector<inversion_info> inv(nrhs);
  std::vector<double> epsv(nrhs, eps);
  if (eps_per_system) epsv = *eps_per_system;
  qmg::BatchPoolT<T> pool(phi.stride, nrhs);
  qmg::BatchT<T> r = pool.get(), p = pool.get(), Ap = pool.get();
  const std::vector<double> bsq = qmg::bnorm2sq(phi0, size, mask);
  std::vector<double> rsq(nrhs, 0.0), bnorm(nrhs, 0.0);
  std::vector<int> its(nrhs, 0), ops(nrhs, 0);
  std::vector<bool> conv(nrhs, false);
  for (int k = 0; k < nrhs; k++) bnorm[k] = std::sqrt(bsq[k]);
  const qmg::cvec one(nrhs, 1.0);
  std::vector<qmg::BatchT<T> > pv(1);
  unsigned live = (r.p && p.p && Ap.p) ? mask : 0u;   // systems that may still start a cycle
  if (!live && mask) std::cout << "[QMG-ERROR]: " << name << ": out of device memory for the CG work vectors\n";
  bool first = true;
  while (live) {
    // ---- one cycle (minv_vector_cg with at most `chunk` iterations per system)
    std::vector<int> chunk(nrhs, 0), done_in_cycle(nrhs, 0);
    unsigned act = 0;
    for (int k = 0; k < nrhs; k++) {
      if (!qmg::is_active(live, k)) continue;
      const int left = max_iter - its[k];
      chunk[k] = (restart_freq > 0 && left > restart_freq) ? restart_freq : left;
    }
    if (first && zero_guess) { qmg::bcopy(r, phi0, size, live); rsq = bsq; }
    else {
      matrix_vector(Ap, phi, live, extra_info);
      for (int k = 0; k < nrhs; k++) if (qmg::is_active(live, k)) ops[k]++;
      qmg::bxmyz(phi0, Ap, r, size, live);
      const std::vector<double> t = qmg::bnorm2sq(r, size, live);
      for (int k = 0; k < nrhs; k++) if (qmg::is_active(live, k)) rsq[k] = t[k];
    }
    first = false;
    qmg::bcopy(p, r, size, live);
    for (int k = 0; k < nrhs; k++) {
      if (!qmg::is_active(live, k)) continue;
      conv[k] = (bnorm[k] == 0.0) || (std::sqrt(rsq[k]) < epsv[k] * bnorm[k]);
      if (!conv[k] && chunk[k] > 0) act |= 1u << k;
    }
    while (act) {
      matrix_vector(Ap, p, act, extra_info);
      pv[0] = p;
      const std::vector<qmg::cvec> d = qmg::bmultidot(pv, 1, Ap, size, act);   // <p, A p>
      qmg::cvec alpha(nrhs, 0.0), malpha(nrhs, 0.0);
      unsigned upd = 0;
      for (int k = 0; k < nrhs; k++) {
        if (!qmg::is_active(act, k)) continue;
        ops[k]++;
        const double pAp = d[k][0].real();
        if (pAp == 0.0) { act &= ~(1u << k); continue; }   // breakdown: this system's cycle ends (krylov.hpp `break`)
        alpha[k] = rsq[k] / pAp; malpha[k] = -alpha[k];
        upd |= 1u << k;
      }
      qmg::bcaxpy(alpha, p, phi, size, upd);
      qmg::bcaxpy(malpha, Ap, r, size, upd);
      const std::vector<double> rn = qmg::bnorm2sq(r, size, upd);
      qmg::cvec beta(nrhs, 0.0);
      unsigned go_on = 0;
      for (int k = 0; k < nrhs; k++) {
        if (!qmg::is_active(upd, k)) continue;
        its[k]++; done_in_cycle[k]++;
        if (verb && verb->verbosity == VERB_DETAIL) { std::cout << verb->verb_prefix << "CG"; if (nrhs > 1) std::cout << " rhs " << k; std::cout << " Iter " << its[k] << " RelTol " << std::sqrt(rn[k]) / bnorm[k] << "\n"; }
        if (std::sqrt(rn[k]) < epsv[k] * bnorm[k]) { rsq[k] = rn[k]; conv[k] = true; act &= ~(1u << k); continue; }
        beta[k] = rn[k] / rsq[k];
        rsq[k] = rn[k];
        if (done_in_cycle[k] >= chunk[k]) act &= ~(1u << k);
        else go_on |= 1u << k;
      }
      qmg::bcaxpbyz(one, r, beta, p, p, size, go_on);   // p = r + beta p
    }
    // ---- which systems start another cycle (minv_vector_cg_restart: stop on success, on a cycle without an iteration, at max_iter)
    unsigned next = 0;
    for (int k = 0; k < nrhs; k++) {
      if (!qmg::is_active(live, k)) continue;
      if (restart_freq > 0 && !conv[k] && done_in_cycle[k] > 0 && its[k] < max_iter) next |= 1u << k;
    }
    live = next;
  }
  for (int k = 0; k < nrhs; k++) {
    inv[k].success = conv[k]; inv[k].iter = its[k]; inv[k].resSq = rsq[k]; inv[k].ops_count = ops[k]; inv[k].name = name;
    if (verb && verb->verbosity != VERB_NONE && qmg::is_active(mask, k)) {   // (one system: krylov.hpp's line, word for word)
      std::cout << verb->verb_prefix << name;
      if (nrhs > 1) std::cout << " rhs " << k;
      std::cout << (conv[k] ? " Success " : " Fail ") << "Iter " << its[k] << " RelTol " << (bnorm[k] > 0 ? std::sqrt(rsq[k]) / bnorm[k] : 0.0) << "\n";
    }
  }
  return inv;
}

// ---------------------------------------------------------------------------------------------
// One K-cycle application for the active systems of a batch: StatefulMultigridMG::mg_preconditioner
// (multigrid.hpp; stateful_multigrid.h:734-1060) step for step.  extra_data is a BatchKcycle.
// ---------------------------------------------------------------------------------------------
struct BatchKcycle {
  StatefulMultigridMG* mg;
  int nrhs;
  BatchKcycle(StatefulMultigridMG* mg_, int nrhs_) : mg(mg_), nrhs(nrhs_) {}
  // the configurations the batched cycle implements: ORIGINAL, RIGHT_JACOBI or RIGHT_SCHUR levels with MR or CGNE smoothers and flexible-GCR
  // intermediate solves; the coarsest solve by GCR on one of those operators or by CG on one of the four normal-equation forms -- every
  // combination StatefulMultigridMG::mg_preconditioner takes -- provided the variant stencils they name have been built
  bool supported() {
    const int nl = mg->get_num_levels();
    if (nl < 2) return false;
    for (int i = 0; i < nl - 1; i++) {
      StatefulMultigridMG::LevelSolveMG* ls = mg->get_level_solve(i);
      if (!ls || !BatchOp::supported(ls->fine_stencil_app) || !BatchOp::variants_built(mg->get_stencil(i), ls->fine_stencil_app)) return false;
      // CGNE smoothers (MR on A A^dagger, then A^dagger: stateful_multigrid.h:847-857, 1032-1042) act on the ORIGINAL and RIGHT_JACOBI operators and need
      // the dagger stencil of that operator; on the Schur operator the reference ignores the flag
      if ((ls->pre_cgne || ls->post_cgne) && ls->fine_stencil_app != QMG_MATVEC_RIGHT_SCHUR &&
          !BatchOp::variants_built(mg->get_stencil(i), ls->fine_stencil_app == QMG_MATVEC_ORIGINAL ? QMG_MATVEC_DAGGER : QMG_MATVEC_RBJ_DAGGER)) return false;
    }
    const QMGStencilType ct = mg->get_coarsest_solve()->coarsest_stencil_app;
    return (BatchOp::supported(ct) || BatchOp::is_normal(ct)) && BatchOp::variants_built(mg->get_stencil(nl - 1), ct);
  }
  // complex<float> shadows of every level's matrices and null vectors, for the QMG_C32 K-cycle (the fp64 hierarchy stays
  // the master copy; call again after the hierarchy changes)
  // half_fine: the fine level (nc = 2) additionally keeps its matrices in 16 bits for the K-cycle's own applies (112 B/site)
  // half_coarse: the Galerkin levels keep theirs (and their right-block-Jacobi hops) in 16 bits as well (kernels B32 / C with complex<half> matrices)
  bool enable_f32_hierarchy(bool half_fine = false, bool half_coarse = false) {
    const int nl = mg->get_num_levels();
    for (int i = 0; i < nl; i++) if (!mg->get_stencil(i) || !mg->get_stencil(i)->enable_f32_shadow(i == 0 ? half_fine : half_coarse)) return false;
    for (int i = 0; i < nl - 1; i++) if (!mg->get_transfer(i)->enable_f32_shadow()) return false;
    return true;
  }
};

template <typename T>
inline void mg_preconditioner_batch(qmg::BatchT<T> lhs, qmg::BatchT<T> rhs, int size, unsigned mask, void* extra_data, inversion_verbose_struct* verb) {
  BatchKcycle* bk = (BatchKcycle*)extra_data;
  StatefulMultigridMG* mg = bk->mg;
  const int nrhs = bk->nrhs;
  const int level = mg->get_multigrid_level();
  const int total_num_levels = mg->get_num_levels();
  Stencil2D* fine_stencil = mg->get_stencil(level);
  Stencil2D* coarse_stencil = mg->get_stencil(level + 1);
  TransferMG* transfer = mg->get_transfer(level);
  StatefulMultigridMG::LevelSolveMG* level_solve = mg->get_level_solve();
  const size_t fine_size = (size_t)mg->get_lattice(level)->get_size_cv_l();
  const size_t coarse_size = (size_t)mg->get_lattice(level + 1)->get_size_cv_l();

  inversion_verbose_struct verb2(VERB_SUMMARY, std::string(" "));
  if (verb == 0 || verb->verbosity == VERB_NONE) { verb2.verbosity = VERB_NONE; verb2.precond_verbosity = VERB_NONE; }
  else verb2.precond_verbosity = VERB_SUMMARY;
  verb2.verb_prefix = "  ";
  for (int i = 1; i < level + 1; i++) verb2.verb_prefix += "  ";
  verb2.verb_prefix += "[QMG-MG-SOLVE-INFO]: Level " + std::to_string(level + 1) + " ";

  const QMGStencilType fine_type = level_solve->fine_stencil_app;
  BatchOp fine_op(fine_stencil, fine_type);
  const size_t fine_size_solve = (fine_type == QMG_MATVEC_RIGHT_SCHUR) ? fine_size / 2 : fine_size;

  int coarse_max_iter, coarse_restart;
  double coarse_tol;
  QMGStencilType coarse_type;
  if (level < total_num_levels - 2) {
    StatefulMultigridMG::LevelSolveMG* cs = mg->get_level_solve(level + 1);
    coarse_type = cs->fine_stencil_app; coarse_max_iter = cs->intermediate_iters; coarse_tol = cs->intermediate_tol; coarse_restart = cs->intermediate_restart_freq;
  } else {
    StatefulMultigridMG::CoarsestSolveMG* cs = mg->get_coarsest_solve();
    coarse_type = cs->coarsest_stencil_app; coarse_max_iter = cs->coarsest_iters; coarse_tol = cs->coarsest_tol; coarse_restart = cs->coarsest_restart_freq;
  }
  BatchOp coarse_op(coarse_stencil, coarse_type);
  const size_t coarse_size_solve = (coarse_type == QMG_MATVEC_RIGHT_SCHUR) ? coarse_size / 2 : coarse_size;

  // scratch for this level (recycled through VecPool's per-length free lists)
  qmg::BatchPoolT<T> fpool(fine_size, nrhs), cpool(coarse_size, nrhs);
  qmg::BatchT<T> Atmp = fpool.get(), z1 = fpool.get(), r1 = fpool.get();
  int nact = 0;
  for (int k = 0; k < nrhs; k++) if (qmg::is_active(mask, k)) nact++;

  // One smoother application from x0 = 0: x ~ A^-1 b by `iters` steps of MR(0.85) -- or, `cgne` on the ORIGINAL / RIGHT_JACOBI operator, MR on A A^dagger y = b
  // followed by x = A^dagger y (stateful_multigrid.h:847-857 / 1032-1042; on the Schur operator the reference ignores the flag).  Tolerances no MR step can
  // reach take the fixed-count form with its scalars on the device, which can also hand back its recursive residual b - A x (r_out; in the CGNE
  // form b - M M^dagger y is that same vector).  Returns whether r_out was filled.
  auto smooth = [&](qmg::BatchT<T> x, qmg::BatchT<T> b, qmg::BatchT<T>* r_out, int iters, double tol, bool cgne, QMGDslashType type) -> bool {
    const bool ne = cgne && (fine_type == QMG_MATVEC_ORIGINAL || fine_type == QMG_MATVEC_RIGHT_JACOBI);
    const bool rbj = fine_type == QMG_MATVEC_RIGHT_JACOBI;
    BatchOp ne_op(fine_stencil, rbj ? QMG_MATVEC_RBJ_M_MDAGGER : QMG_MATVEC_M_MDAGGER);
    BatchOp* op = ne ? &ne_op : &fine_op;
    qmg::BatchT<T> y = ne ? fpool.get() : x;
    if (!y.p) { std::cout << "[QMG-ERROR]: out of device memory for the CGNE smoother's iterate\n"; return false; }
    bool have_r = false;
    if (qmg::mr_tolerance_unreachable(tol) && qmg::kcycle_device_scalars()) {
      const int nops = bmr_fixed_zero_guess<T>(y, b, r_out, (int)fine_size_solve, iters, 0.85, apply_stencil_typed_batch<T>, (void*)op, mask, op);
      mg->add_tracker_count(type, (ne ? 2 : 1) * nops * nact, level);
      have_r = r_out != 0;
    } else {
      qmg::bzero(y, fine_size, mask);
      std::vector<inversion_info> inv = bminv_vector_minres_zero_guess<T>(y, b, (int)fine_size_solve, iters, tol, 0.85, apply_stencil_typed_batch<T>, (void*)op, mask);
      for (int k = 0; k < nrhs; k++) if (qmg::is_active(mask, k)) mg->add_tracker_count(type, (ne ? 2 : 1) * inv[k].ops_count, level);
    }
    if (ne) {
      BatchOp dag(fine_stencil, rbj ? QMG_MATVEC_RBJ_DAGGER : QMG_MATVEC_DAGGER);
      apply_stencil_typed_batch<T>(x, y, mask, (void*)&dag);
      mg->add_tracker_count(type, nact, level);
    }
    return have_r;
  };

  // ---- 1. pre-smooth: A z1 ~ rhs, r1 = rhs - A z1
  if (level_solve->pre_iters > 0) {
    // the fixed-count form's recursive residual is r1 (the reference recomputes rhs - A z1 with one more apply, stateful_multigrid.h:863-866:
    // the same vector up to rounding), so that smoother costs pre_iters applies, not pre_iters + 1
    if (!smooth(z1, rhs, &r1, level_solve->pre_iters, level_solve->pre_tol, level_solve->pre_cgne, QMG_DSLASH_TYPE_PRESMOOTH)) {
      apply_op_residual<T>(&fine_op, r1, z1, rhs, Atmp, fine_size_solve, mask);
      mg->add_tracker_count(QMG_DSLASH_TYPE_PRESMOOTH, nact, level);
    }
  } else {
    qmg::bzero(z1, fine_size, mask);
    qmg::bcopy(r1, rhs, fine_size_solve, mask);
    qmg::bcopy(z1, rhs, fine_size_solve, mask);
  }
  // (Schur: the odd half of r1 must not leak stale pool data into the restriction)
  if (fine_type == QMG_MATVEC_RIGHT_SCHUR) qmg::bzero(batch_odd_half(r1, fine_size_solve), fine_size - fine_size_solve, mask);

  // ---- 2. restrict, prepare, coarse solve (recursion = the "K"), reconstruct
  qmg::BatchT<T> r_coarse = cpool.get(), r_coarse_prep = cpool.get(), e_coarse = cpool.get(), e_rec = cpool.get();
  qmg::bzero(r_coarse, coarse_size, mask);
  transfer->restrict_f2c_precond_t<T>(r1.p, r1.stride, r_coarse.p, r_coarse.stride, nrhs, mask);
  std::vector<double> inner_tol(nrhs, coarse_tol);
  if (coarse_type == QMG_MATVEC_ORIGINAL) qmg::bcopy(r_coarse_prep, r_coarse, coarse_size, mask);   // prepare_M is a copy; rnorm_prep == rnorm
  else {
    const std::vector<double> rn = qmg::bnorm2sq(r_coarse, coarse_size, mask);
    prepare_M_batch<T>(coarse_stencil, coarse_type, r_coarse_prep, r_coarse, mask);
    const std::vector<double> rp = qmg::bnorm2sq(r_coarse_prep, coarse_size, mask);
    for (int k = 0; k < nrhs; k++)
      if (qmg::is_active(mask, k) && rp[k] > 0.0) inner_tol[k] = coarse_tol * std::sqrt(rn[k]) / std::sqrt(rp[k]);
  }
  qmg::bzero(e_coarse, coarse_size, mask);
  std::vector<inversion_info> cinv;
  if (level == total_num_levels - 2 && BatchOp::is_normal(coarse_type)) {   // CG on a normal-equation operator, shifted by normal_shift (:930-960)
    coarse_op.normal_shift = mg->get_coarsest_solve()->normal_shift; coarse_op.shift_length = coarse_size_solve;
    cinv = bcg_core<T>(e_coarse, r_coarse_prep, (int)coarse_size_solve, coarse_max_iter, coarse_tol, coarse_restart, apply_stencil_typed_batch<T>, (void*)&coarse_op,
                       mask, true, &verb2, coarse_restart == -1 ? "CG" : "CG-restart", &inner_tol);
  } else if (level == total_num_levels - 2) {
    cinv = bgcr_core<T>(e_coarse, r_coarse_prep, (int)coarse_size_solve, coarse_max_iter, coarse_tol, coarse_restart, apply_stencil_typed_batch<T>, (void*)&coarse_op,
                        (batch_precond_op_t<T>)0, 0, mask, true, &verb2, coarse_restart == -1 ? "GCR" : "GCR-restart", &inner_tol);
  } else {
    mg->go_coarser();
    cinv = bgcr_core<T>(e_coarse, r_coarse_prep, (int)coarse_size_solve, coarse_max_iter, coarse_tol, coarse_restart, apply_stencil_typed_batch<T>, (void*)&coarse_op,
                        mg_preconditioner_batch<T>, (void*)bk, mask, true, &verb2, coarse_restart == -1 ? "VPGCR" : "VPGCR-restart", &inner_tol);
    mg->go_finer();
  }
  for (int k = 0; k < nrhs; k++)
    if (qmg::is_active(mask, k)) { mg->add_tracker_count(QMG_DSLASH_TYPE_KRYLOV, cinv[k].ops_count, level + 1); mg->add_iterations_count(cinv[k].iter, level + 1); }
  reconstruct_M_batch<T>(coarse_stencil, coarse_type, e_rec, e_coarse, r_coarse, mask);

  // ---- 3. prolong and correct: lhs = z1 + P e
  qmg::BatchT<T> z2 = r1;   // r1 is free again
  qmg::bzero(z2, fine_size, mask);
  transfer->prolong_c2f_precond_t<T>(e_rec.p, e_rec.stride, z2.p, z2.stride, nrhs, mask);
  if (coarse_type == QMG_MATVEC_RIGHT_SCHUR) qmg::bzero(batch_odd_half(z2, fine_size / 2), fine_size - fine_size / 2, mask);   // as multigrid.hpp
  qmg::bcxpyz(z1, z2, lhs, fine_size_solve, mask);

  // ---- 4. post-smooth on r2 = rhs - A lhs
  if (level_solve->post_iters > 0) {
    qmg::BatchT<T> r2 = z2, z3 = z1;   // both free again
    apply_op_residual<T>(&fine_op, r2, lhs, rhs, Atmp, fine_size_solve, mask);   // r2 = rhs - A lhs
    smooth(z3, r2, (qmg::BatchT<T>*)0, level_solve->post_iters, level_solve->post_tol, level_solve->post_cgne, QMG_DSLASH_TYPE_POSTSMOOTH);
    qmg::bcxpy(z3, lhs, fine_size_solve, mask);
  }
}

// Scratch of one K-cycle-preconditioned flexible GCR solve, allocated up front (qmg::VecPool::reserve): the outer solve's residual / work
// vectors and 2 x outer_basis search directions of `outer_size` elements, and per level what one visit of mg_preconditioner_batch and the
// level's inner GCR (a handful of iterations at its tolerance of 0.2) check out.  An estimate, not a contract: whatever a solve needs
// beyond it is allocated on demand as before, and the drivers' timing lines say how long the allocator ran inside the solve.
inline bool qmg_reserve_kcycle_scratch(StatefulMultigridMG* mg, size_t outer_size, int outer_basis, int nrhs = 1) {
  bool good = qmg::VecPool::reserve(outer_size * (size_t)nrhs, 2 * outer_basis + 6);
  const int nl = mg->get_num_levels();
  for (int l = 0; l < nl && good; l++) {
    const size_t n = (size_t)mg->get_lattice(l)->get_size_cv_l() * (size_t)nrhs;
    good = qmg::VecPool::reserve(n, l == 0 ? 10 : 40);
  }
  return good;
}

// StatefulMultigridMG::mg_preconditioner for one system through this engine (declared in multigrid.hpp).  QMG_KCYCLE_ENGINE=single
// keeps the single-vector implementation of multigrid.hpp (A/B runs, and the reference-shaped code path for the parity tests).
inline bool qmg_kcycle_via_batch(StatefulMultigridMG* mg, complex<double>* lhs, complex<double>* rhs, int size, inversion_verbose_struct* verb) {
  static const bool on = !(getenv("QMG_KCYCLE_ENGINE") && std::string(getenv("QMG_KCYCLE_ENGINE")) == "single");
  // y-slabs take this engine too: launch_set_batch exchanges the batch's halo rows and every reduction is completed across the ranks (C3 shape on one
  // slab / two thread-emulated slabs: 1.43 / 1.71 s against 1.71 / 2.24 s through the single-vector code).  QMG_KCYCLE_SLAB_ENGINE=single keeps slabs on
  // the single-vector code, whose nc = 2 applies overlap their halo exchange with the interior rows -- the choice to re-measure on real xGMI links.
  static const bool slab_on = !(getenv("QMG_KCYCLE_SLAB_ENGINE") && std::string(getenv("QMG_KCYCLE_SLAB_ENGINE")) == "single");
  if (!on || (qmg::slab().on && !slab_on)) return false;
  BatchKcycle bk(mg, 1);
  if (!bk.supported()) return false;
  const size_t stride = (size_t)mg->get_lattice(mg->get_multigrid_level())->get_size_cv_l();
  mg_preconditioner_batch<double>(qmg::Batch(lhs, stride, 1), qmg::Batch(rhs, stride, 1), size, 1u, (void*)&bk, verb);
  return true;
}

// The fp32 K-cycle as the preconditioner of an fp64 flexible outer solve (BASELINE configs[4] "fp32"): the residual of
// the active systems is rounded to complex<float>, ONE K-cycle runs entirely on the fp32 shadow hierarchy
// (BatchKcycle::enable_f32_hierarchy), and the correction is widened back.  The outer VPGCR orthogonalises and
// measures in fp64, so the solve converges to its fp64 tolerance; only the preconditioner's quality is fp32.
// extra_data: BatchKcycle, as for mg_preconditioner_batch.
inline void mg_preconditioner_batch_mixed(qmg::Batch lhs, qmg::Batch rhs, int size, unsigned mask, void* extra_data, inversion_verbose_struct* verb) {
  BatchKcycle* bk = (BatchKcycle*)extra_data;
  const size_t n = (size_t)bk->mg->get_lattice(bk->mg->get_multigrid_level())->get_size_cv_l();
  qmg::BatchPoolT<float> pool(n, lhs.nrhs);
  qmg::BatchT<float> r32 = pool.get(), z32 = pool.get();
  if (r32.p == 0 || z32.p == 0) { std::cout << "[QMG-ERROR]: out of device memory for the fp32 residual / correction\n"; return; }
  qmg::bzero(r32, n, mask);                       // (Schur: the odd half beyond `size` must be defined)
  qmg::bconvert(r32, rhs, (size_t)size, mask);
  qmg::bzero(z32, n, mask);
  mg_preconditioner_batch<float>(z32, r32, size, mask, extra_data, verb);
  qmg::bconvert(lhs, z32, (size_t)size, mask);
}

#endif
