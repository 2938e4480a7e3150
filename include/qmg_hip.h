/* qmg_hip.h -- C-ABI of libqmg_hip.so: the MI355X (gfx950) drop-in for quantum-mg's
 * multigrid hot path (stencil apply, cshift, restrict/prolong, global reductions).
 *
 * The reference (weinbe2/quantum-mg) is a header-only C++ library with no FFI of its own;
 * its hot path is entered through the `matrix_op_cplx` callback
 *     void (*)(complex<double>* lhs, complex<double>* rhs, void* extra_data)
 * (stencil/stencil_2d.h:15-19) and through public methods of Stencil2D / TransferMG that
 * operate on the PUBLIC arrays `clover`, `hopping`, `null_vectors` (stencil_2d.h:148-210,
 * transfer/transfer.h:79).  This ABI is therefore STATELESS: every entry point takes the
 * raw arrays + lattice extents exactly as the reference method receives them, except that
 * all array pointers are DEVICE (HBM) addresses.  The reference's pointer-swap trick for
 * operator variants (perform_swap_dagger, stencil_2d.h:1142-1178) maps onto passing a
 * different `clover`/`hopping` pair in the descriptor.  The C++ facade in
 * quantum-mg_amd/include/qmg/ rebuilds the reference class surface on top of this file;
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *  - all complex data: interleaved (re,im) double == std::complex<double>
 *  - layouts (reference README.md:4-11): vector (eo,y,x,c); matrix (eo,y,x,c1,c2) row-major;
 *    hopping (mu,eo,y,x,c1,c2), mu in {+x,+y,-x,-y}; site i = (y + p*Ly)*Lx/2 + x/2
 *  - Lx, Ly even and >= 2 (cshift_2d.h:62,79 step two rows at a time)
 *  - extents are 64-bit internally (the reference's `int` size_hopping overflows at
 *    1024^2 x 24^2 x 4, lattice.h:23,40)
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are
 *    asynchronous on that stream unless stated otherwise
 *  - return value: qmg_status (0 = success).  Nothing here falls back to the CPU: a call
 *    that cannot run on the GPU returns an error.
 */
#ifndef QMG_HIP_H
#define QMG_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  QMG_SUCCESS = 0,
  QMG_ERR_INVALID = 1,      /* bad extents / NULL array / unsupported nc             */
  QMG_ERR_HIP = 2,          /* a HIP runtime call failed (qmg_last_hip_error())      */
  QMG_ERR_UNSUPPORTED = 3,  /* valid in the reference but not built yet              */
  QMG_ERR_NO_DEVICE = 4
} qmg_status;

/* Storage precision of device arrays.  The reference is fp64 only (stencil_2d.h:17-18); QMG_C32 is the fp32 instantiation
 * BASELINE configs[4] asks for: complex<float> storage of vectors, matrices and null vectors, half the bytes of every
 * HBM-bound kernel.  Reductions always accumulate and return fp64. */
typedef enum { QMG_C64 = 0, QMG_C32 = 1 } qmg_dtype;

/* cshift directions / parities: names and values identical to cshift/cshift_2d.h:13-36 */
typedef enum {
  QMG_CSHIFT_FROM_0 = 1, QMG_CSHIFT_FROM_XP1 = 2, QMG_CSHIFT_FROM_YP1 = 3, QMG_CSHIFT_FROM_XM1 = 4, QMG_CSHIFT_FROM_YM1 = 5,
  QMG_CSHIFT_FROM_XP2 = 6, QMG_CSHIFT_FROM_YP2 = 7, QMG_CSHIFT_FROM_XM2 = 8, QMG_CSHIFT_FROM_YM2 = 9,
  QMG_CSHIFT_FROM_XP1YP1 = 10, QMG_CSHIFT_FROM_XM1YP1 = 11, QMG_CSHIFT_FROM_XM1YM1 = 12, QMG_CSHIFT_FROM_XP1YM1 = 13
} qmg_cshift_dir;   /* only the distance-1 shifts (1..5) are implemented, as in the reference (:120-129) */
typedef enum { QMG_EO_FROM_EVEN = 1, QMG_EO_FROM_ODD = 2, QMG_EO_FROM_EVENODD = 3 } qmg_eo;
/* hopping direction index mu = 0..3 is {+x,+y,-x,-y} (stencil/stencil_2d.h:25-40) */

/* Which pieces of  lhs (+)= M rhs  one fused launch applies.  Each reference method is one mask:
 *   apply_M            (stencil_2d.h:912-936)  QMG_P_ALL
 *   apply_M_clover     (:694-703)              QMG_P_CLOVER
 *   apply_M_eo / _oe   (:706-802)              QMG_P_EO / QMG_P_OE      (even / odd OUTPUT sites)
 *   ... one direction  (:736-841)              QMG_P_EO_XP1 << dir, QMG_P_OE_XP1 << dir
 *   apply_M_shift      (:865-909)              QMG_P_SHIFT
 *   apply_M_ee / _oo   (:666-692)              QMG_P_CLOVER_E|QMG_P_SHIFT_E with eo/dof shift zeroed
 * QMG_P_ZERO_E/_O overwrite that half instead of accumulating into it (the C wrappers'
 * zero_vector + apply, :2571-2576, in one pass).  A half with no bit set is not touched. */
enum {
  QMG_P_CLOVER_E = 1u << 0,  QMG_P_CLOVER_O = 1u << 1,
  QMG_P_EO_XP1 = 1u << 2, QMG_P_EO_YP1 = 1u << 3, QMG_P_EO_XM1 = 1u << 4, QMG_P_EO_YM1 = 1u << 5,
  QMG_P_OE_XP1 = 1u << 6, QMG_P_OE_YP1 = 1u << 7, QMG_P_OE_XM1 = 1u << 8, QMG_P_OE_YM1 = 1u << 9,
  QMG_P_SHIFT_E = 1u << 10, QMG_P_SHIFT_O = 1u << 11,
  QMG_P_ZERO_E = 1u << 12,  QMG_P_ZERO_O = 1u << 13,
  QMG_P_CLOVER = 3u, QMG_P_EO = 0xFu << 2, QMG_P_OE = 0xFu << 6, QMG_P_HOPPING = 0xFFu << 2,
  QMG_P_SHIFT = 3u << 10, QMG_P_ZERO = 3u << 12,
  QMG_P_ALL = 0xFFFu
};

/* The public data of a Stencil2D (stencil_2d.h:148-177) as the kernels need it. */
typedef struct {
  int Lx, Ly, nc;
  const void* clover;      /* device, Lx*Ly*nc*nc complex, or NULL (stencil has no clover)   */
  const void* hopping;     /* device, 4*Lx*Ly*nc*nc complex, or NULL                         */
  double shift[2];         /* identity shift (mass)                         (:170)           */
  double eo_shift[2];      /* +even / -odd                                  (:173)           */
  double dof_shift[2];     /* +top half / -bottom half of the dof, nc even  (:177)           */
} qmg_stencil_desc;

/* ---------------- runtime ---------------- */
int qmg_init(int device);                       /* hipSetDevice + sanity; QMG_ERR_NO_DEVICE if none */
int qmg_device_count(int* n);
const char* qmg_status_string(int status);
const char* qmg_last_hip_error(void);
const char* qmg_version(void);
int qmg_malloc(void** dev_ptr, size_t bytes);   /* replaces allocate_vector<T> for device arrays */
int qmg_free(void* dev_ptr);
int qmg_shutdown(void);                         /* ordered teardown for the calling host thread: device sync, then the library's per-thread workspaces are released */
int qmg_mem_info(size_t* free_bytes, size_t* total_bytes);   /* HBM free / total on the current device: batch sizing */
int qmg_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream);  /* synchronous when stream == NULL */
int qmg_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);
int qmg_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes, void* stream);
int qmg_memset_zero(void* dev_ptr, size_t bytes, void* stream);
int qmg_stream_create(void** stream);
int qmg_stream_destroy(void* stream);
int qmg_stream_sync(void* stream);
int qmg_event_create(void** ev);
int qmg_event_destroy(void* ev);
int qmg_event_record(void* ev, void* stream);
int qmg_stream_wait_event(void* stream, void* ev);                     /* later work on `stream` waits for ev (no host sync) */
int qmg_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);   /* synchronises on ev_stop */

/* ---------------- cshift (cshift/cshift_2d.h:45-236) ---------------- */
/* lhs(opposite half) = rhs(neighbour); dof complex numbers per site. */
int qmg_cshift(void* lhs, const void* rhs, int cdir, int eo, int dof, int Lx, int Ly, void* stream);

/* ---------------- stencil apply (stencil/stencil_2d.h:666-936, 2418-2453, 2571-2716) ---------------- */
/* lhs (+)= pieces(M) rhs, fused in one launch.  nrhs >= 1 independent right-hand sides stored
 * `vec_stride` complex elements apart share one read of the stencil matrices (vec_stride is
 * ignored for nrhs == 1).  lhs and rhs must not overlap, except that a call with only
 * QMG_P_EO (or only QMG_P_OE) pieces may pass lhs == rhs: it reads one half and writes the
 * other (the reference's in-place use, stencil_2d.h:1904, staggered.h:236). */
int qmg_stencil_apply(const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces,
                      int nrhs, size_t vec_stride, void* stream);
/* The apply with the norms of its results from the same pass: lhs_k (+)= pieces(M) rhs_k and norms[k] = |lhs_k|^2 (the
 * apply followed by quantum-linalg's norm2sq -- call sites stateful_multigrid.h:880,884 -- without re-reading the vector: 40 instead of 56 B/site/rhs for
 * the staggered operator).  fp64, nc = 1 or 2, pieces touching BOTH parities, lhs != rhs, nrhs <= 16; anything else, or a
 * call while distributed reductions are on, is QMG_ERR_UNSUPPORTED.  lhs receives the bytes qmg_stencil_apply writes; the
 * norms are summed in a fixed order (run-to-run reproducible), not in qmg_norm2sq's order (they agree to rounding).
 * norms_dev: nrhs doubles in device memory or NULL; norms_host: nrhs doubles or NULL (then the stream is synchronised). */
int qmg_stencil_apply_norm2(const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces, int nrhs, size_t vec_stride,
                            double* norms_dev, double* norms_host, void* stream);
/* Same, for a lock-step batch of at most 16 systems of which only those with their bit set in `mask` are read or
 * written.  With nc in {8,12,16,24,32} and two or more active systems the apply runs as an (nc x nc).(nc x k)
 * contraction on the f64 matrix cores (v_mfma_f64_16x16x4_f64), the matrices read once for all k. */
int qmg_stencil_apply_batch(const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces,
                            int nrhs, size_t vec_stride, unsigned mask, void* stream);

/* OPT-IN storage format: d->clover / d->hopping point to complex<float> copies of the matrices (qmg_c64_to_c32); vectors,
 * shifts and all arithmetic stay fp64.  Halves the matrix stream of an HBM-bound coarse apply.  Meant for operators that
 * only precondition (the K-cycle inside a flexible fp64 outer solver); parity: equal to 1e-13 to the fp64 apply of the
 * ROUNDED matrices.  nc = 1, 2, 4 return QMG_ERR_UNSUPPORTED.  nrhs <= 16, mask as in qmg_stencil_apply_batch. */
int qmg_stencil_apply_mat32(const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces,
                            int nrhs, size_t vec_stride, unsigned mask, void* stream);
/* The same with the matrices stored as complex<half> (qmg_convert_to_c16) and vectors of either precision; nc a multiple of 4 and > 4
 * (QMG_ERR_UNSUPPORTED otherwise).  qmg_stencil_apply_epi_t takes mat32 = 2 for this storage. */
int qmg_stencil_apply_mat16_t(int vec_dtype, const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces,
                              int nrhs, size_t vec_stride, unsigned mask, void* stream);
int qmg_c64_to_c32(void* dst_f32, const void* src_f64, size_t n, void* stream);

/* ---------------- operator construction from U(1) links (device side) ---------------- */
/* gauge: nc=1 LatticeGauge (mu,eo,y,x), 2*Lx*Ly complex. */
int qmg_wilson_fill(void* clover, void* hopping, const void* gauge, int Lx, int Ly, double wilson_coeff, void* stream); /* wilson.h:153-209 */
int qmg_staggered_fill(void* hopping, const void* gauge, int Lx, int Ly, void* stream);                                /* staggered.h:50-72 */
int qmg_laplace_fill(void* clover, void* hopping, const void* gauge, int Lx, int Ly, void* stream);                    /* gaugedlaplace.h:45-68 */

/* ---------------- U(1) gauge generation and observables (u1/u1_utils.h) ---------------- */
/* Phase field: DEVICE double[2 Lx Ly], phase[mu*V + site] (gauge_coord_to_index); compact links U = exp(i A) in the same order.
 * heatbath_noncompact_update (u1_utils.h:607-757) as a four-colour PARALLEL heatbath (x-links of even / odd rows, y-links of
 * even / odd columns are conditionally independent): same Gibbs measure as the reference's sequential sweep, different
 * update order and random stream.  `seed` and the running sweep number `first_sweep` key a counter-based generator. */
int qmg_u1_heatbath_noncompact(double* phase, int Lx, int Ly, double beta, int n_update, unsigned long long seed,
                               unsigned long long first_sweep, void* stream);
int qmg_u1_phase_to_gauge(void* gauge, const double* phase, size_t n, void* stream);   /* polar_vector: U = exp(i A) */
int qmg_u1_gauge_to_phase(double* phase, const void* gauge, size_t n, void* stream);   /* A = arg U (what write_gauge_u1 stores) */
/* out_host[0..1] = average plaquette (get_plaquette_u1, :424-462), out_host[2] = topological charge (get_topo_u1, :465-508) */
int qmg_u1_plaquette(const void* gauge, int Lx, int Ly, double* out_host, void* stream);
int qmg_u1_noncompact_action(const double* phase, int Lx, int Ly, double beta, double* out_host, void* stream);   /* :386-421 */

/* ---------------- stencil variants (device side) ---------------- */
/* build_dagger_stencil (stencil_2d.h:1080-1139); also serves build_rbj_dagger_stencil (:1989-2060). */
int qmg_build_dagger(void* dagger_clover, void* dagger_hopping, const void* clover, const void* hopping,
                     int Lx, int Ly, int nc, void* stream);
/* build_rbjacobi_stencil (stencil_2d.h:1452-1601): cinv = (clover + shifts)^-1, rb clover = 1,
 * rb hopping_mu(x) = hopping_mu(x) . cinv(x+mu).  nc <= 32. */
int qmg_build_rbjacobi(void* cinv, void* rb_clover, void* rb_hopping, const qmg_stencil_desc* d, void* stream);
/* batched nc x nc conjugate transpose (cMATcopy_conjtrans_square) */
int qmg_cmat_conjtrans(void* out, const void* in, size_t nsite, int nc, void* stream);

/* ---------------- BLAS-1 on device vectors (the quantum-linalg leaves the path calls) ---------------- */
/* n = number of complex elements; scalars are (re,im) pairs passed by value. */
int qmg_zero_vector(void* x, size_t n, void* stream);
int qmg_copy_vector(void* dst, const void* src, size_t n, void* stream);
int qmg_cax(double ar, double ai, void* x, size_t n, void* stream);                                   /* x *= a            */
int qmg_caxy(double ar, double ai, const void* x, void* y, size_t n, void* stream);                   /* y  = a x          */
int qmg_caxpy(double ar, double ai, const void* x, void* y, size_t n, void* stream);                  /* y += a x          */
int qmg_cxpy(const void* x, void* y, size_t n, void* stream);                                         /* y += x            */
int qmg_cxpay(const void* x, double ar, double ai, void* y, size_t n, void* stream);                  /* y  = x + a y      */
int qmg_caxpby(double ar, double ai, const void* x, double br, double bi, void* y, size_t n, void* stream);            /* y = a x + b y */
int qmg_cxpyz(const void* x, const void* y, void* z, size_t n, void* stream);                         /* z  = x + y        */
int qmg_caxpbyz(double ar, double ai, const void* x, double br, double bi, const void* y, void* z, size_t n, void* stream); /* z = a x + b y */
/* y += sum_{i<k} a_i x_i in one pass. coeffs: HOST array of 2k doubles (re,im); xs: HOST array of k device pointers. */
int qmg_multi_caxpy(const double* coeffs, const void* const* xs, int k, void* y, size_t n, void* stream);
/* y[site,c] = scale[c] * x[site,shuffle[c]]  (caxy_shuffle_pattern; gamma5 / sigma1 / chiral projections,
 * wilson.h:74-143, coarse.h:498-657).  nc <= 64; scale/shuffle are HOST arrays of length nc. */
int qmg_caxy_pattern(const double* scale, const int* shuffle, int nc, const void* x, void* y, size_t nsite, void* stream);
/* complex Gaussian fill, unit variance per real component, counter-based (reproducible, layout-independent) */
int qmg_gaussian(void* x, size_t n, unsigned long long seed, void* stream);

/* ---------------- global reductions ---------------- */
/* Two-stage (wavefront DPP + LDS block) deterministic reductions.  `out_dev` is a DEVICE buffer
 * (1 double for real results, 2 for dot) written asynchronously; `out_host`, if non-NULL, receives a
 * copy and makes the call synchronous.  Either may be NULL, not both. */
int qmg_norm2sq(const void* x, size_t n, double* out_dev, double* out_host, void* stream);
int qmg_dot(const void* x, const void* y, size_t n, double* out_dev, double* out_host, void* stream);   /* sum conj(x) y */
int qmg_diffnorm2sq(const void* x, const void* y, size_t n, double* out_dev, double* out_host, void* stream);
int qmg_norminf(const void* x, size_t n, double* out_dev, double* out_host, void* stream);
/* k dot products <x_i, y>, i < k, in one pass over y (GCR orthogonalisation). xs: HOST array of k device pointers.
 * out: 2k doubles. k <= 64. */
int qmg_multidot(const void* const* xs, int k, const void* y, size_t n, double* out_dev, double* out_host, void* stream);
/* reductions/reductions.h:24-87: per-timeslice (per-y) norm2sq / dot. out: Ly (or 2*Ly) doubles. */
int qmg_norm2sq_cv_timeslice(const void* cv, int Lx, int Ly, int nc, double* out_dev, double* out_host, void* stream);
int qmg_dot_cv_timeslice(const void* a, const void* b, int Lx, int Ly, int nc, double* out_dev, double* out_host, void* stream);
/* redot_cv_timeslice (reductions/reductions.h:47-66): sum[y] = Re sum_{x,c} conj(a) b, Ly doubles */
int qmg_redot_cv_timeslice(const void* a, const void* b, int Lx, int Ly, int nc, double* out_dev, double* out_host, void* stream);
/* gaussian_wall_source (reductions/reductions.h:90-162): cv = real Gaussian (mean + deviation N(0,1), imaginary part 0) on the elements with
 * y == timeslice and component == color, zero elsewhere; QMG_ERR_INVALID for timeslice >= Ly or color >= nc (the reference prints and returns).
 * The numbers come from the library's counter-based generator keyed by (seed, element index), not from std::mt19937 (see qmg_gaussian). */
int qmg_gaussian_wall_source(void* cv, int Lx, int Ly, int nc, int timeslice, int color, unsigned long long seed, double deviation, double mean, void* stream);

/* ---------------- transfer (transfer/transfer.h) ---------------- */
/* Null vectors are passed as the reference holds them: nvec fine vectors, vector-major
 * (null_vectors[d][k], transfer.h:79), contiguous with stride fine_size_cv.  Blocks are the
 * regular (fLx/cLx) x (fLy/cLy) rectangles of build_mapping (:410-448). */
int qmg_prolong(const void* nullvecs, int nvec, const void* coarse, void* fine,
                int fLx, int fLy, int fnc, int cLx, int cLy, int cnc, void* stream);     /* fine += P coarse   (:455-480) */
int qmg_restrict(const void* nullvecs, int nvec, const void* fine, void* coarse,
                 int fLx, int fLy, int fnc, int cLx, int cLy, int cnc, void* stream);    /* coarse += R fine   (:487-511) */
/* block_orthonormalize, one pass, in place (:514-607); cholesky (cLx*cLy*nvec*nvec complex) may be NULL. */
int qmg_block_orthonormalize(void* nullvecs, int nvec, int fLx, int fLy, int fnc, int cLx, int cLy,
                             void* cholesky, void* stream);
/* `passes` (1 or 2) passes in ONE launch -- the TransferMG constructor runs two (:160-174), the factor saved in the first:
 * each block's nvec x (bx by nc_f) tile is read once, orthonormalised in LDS, written once.  Asynchronous, no allocation. */
int qmg_block_orthonormalize_n(void* nullvecs, int nvec, int fLx, int fLy, int fnc, int cLx, int cLy,
                               void* cholesky, int passes, void* stream);

/* block_bi_orthonormalize, one pass, in place (:610-769): separate prolongator / restrictor vectors made block
 * bi-orthonormal (R^dag P = 1 per block); block_L / block_U (cLx*cLy*nvec*nvec complex each) may be NULL. */
int qmg_block_bi_orthonormalize(void* prolong_vecs, void* restrict_vecs, int nvec, int fLx, int fLy, int fnc, int cLx, int cLy,
                                void* block_L, void* block_U, void* stream);

/* ---------------- Galerkin coarse operator (operators/coarse.h:90-444) ---------------- */
int qmg_coarse_build(void* coarse_clover, void* coarse_hopping, const qmg_stencil_desc* fine,
                     const void* nullvecs, const void* restrict_vecs /* or NULL */,
                     int cLx, int cLy, int cnc, void* stream);

/* ---------------- lock-step batches of independent right-hand sides ---------------- */
/* A batch vector is nrhs <= 16 vectors of n complex at a common `stride` (complex elements).  `mask` selects the
 * ACTIVE systems; a frozen system is neither read nor written.  Per-system arithmetic (including the reduction
 * order) is that of the single-vector entry points above.  Scalars are per system: a[2k], a[2k+1] = re, im. */
typedef enum { QMG_BOP_ZERO = 0, QMG_BOP_COPY = 1, QMG_BOP_CAX = 2, QMG_BOP_CAXPY = 3, QMG_BOP_CXPY = 4, QMG_BOP_CAXPBYZ = 5 } qmg_batch_op;
/* ZERO: z = 0; COPY: z = x; CAX: z *= a; CAXPY: z += a x; CXPY: z += x; CAXPBYZ: z = a x + b y */
int qmg_batch_blas(int op, const double* a, const double* b, const void* x, const void* y, void* z, size_t n,
                   int nrhs, size_t stride, unsigned mask, void* stream);
/* y_k += sum_j c[j][k] xs[j]_k ; coeffs[(j*nrhs + k)*2 + {0,1}]; xs: HOST array of nj batch base pointers */
int qmg_batch_multi_caxpy(const double* coeffs, const void* const* xs, int nj, void* y, size_t n,
                          int nrhs, size_t stride, unsigned mask, void* stream);
typedef enum { QMG_BRED_NORM2 = 0, QMG_BRED_DOT = 1, QMG_BRED_DIFFNORM2 = 2 } qmg_batch_red;
/* out_host[2k], out_host[2k+1] for each active system k; returns when the results are on the host (tuning key "reduce_spin": by polling a
 * host word behind them, the stream itself is then not synchronised; 0: hipStreamSynchronize) */
int qmg_batch_reduce(int op, const void* x, const void* y, size_t n, int nrhs, size_t stride, unsigned mask,
                     double* out_host, void* stream);
/* out_host[(k*nj + j)*2 + {0,1}] = <xs[j]_k, y_k>, nj <= 32; returns when the results are on the host (as qmg_batch_reduce) */
int qmg_batch_multidot(const void* const* xs, int nj, const void* y, size_t n, int nrhs, size_t stride, unsigned mask,
                       double* out_host, void* stream);
/* transfer.h:455-511 for the batch: the nvec null vectors are read once per 8 active systems */
int qmg_prolong_batch(const void* nullvecs, int nvec, const void* coarse, void* fine, int fLx, int fLy, int fnc,
                      int cLx, int cLy, int cnc, int nrhs, size_t cstride, size_t fstride, unsigned mask, void* stream);
int qmg_restrict_batch(const void* nullvecs, int nvec, const void* fine, void* coarse, int fLx, int fLy, int fnc,
                       int cLx, int cLy, int cnc, int nrhs, size_t fstride, size_t cstride, unsigned mask, void* stream);

/* ---------------- the path in either storage precision (`_t` = typed) ---------------- */
/* Same operations as the entry points above with the storage type of EVERY array argument (vectors, matrices in the
 * descriptor, null vectors) given by `dtype` (qmg_dtype); host-side scalars and reduction results stay double.  With
 * QMG_C64 they ARE the entry points above.  With QMG_C32: the fine kernels (nc = 1, 2, 4) compute in fp32, the coarse
 * kernels widen to fp64 in registers (f64 FMA / f64 MFMA) and round once on store, reductions accumulate in fp64.
 * Parity bar: relative L2 <= 5e-6 per apply against the fp64 oracle on the same (rounded) inputs.
 * fp32 arrays should be 16-byte aligned with even strides (anything qmg_malloc returns is); otherwise the kernels fall
 * back to 8-byte accesses.  nrhs <= 16, `mask` selects the active systems (nrhs = 1, mask = 1 for a single vector). */
int qmg_convert(void* dst, int dst_dtype, const void* src, int src_dtype, size_t n, void* stream);   /* element-wise copy / round / widen */
int qmg_stencil_apply_t(int dtype, const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces,
                        int nrhs, size_t vec_stride, unsigned mask, void* stream);
int qmg_batch_blas_t(int dtype, int op, const double* a, const double* b, const void* x, const void* y, void* z, size_t n,
                     int nrhs, size_t stride, unsigned mask, void* stream);
int qmg_batch_multi_caxpy_t(int dtype, const double* coeffs, const void* const* xs, int nj, void* y, size_t n,
                            int nrhs, size_t stride, unsigned mask, void* stream);
/* The three vector updates of one flexible-GCR iteration (quantum-linalg's minv_vector_gcr_var_precond_restart as called at
 * multigrid/stateful_multigrid.h:977-996 and tests/n13_wilson_kcycle/wilson_kcycle.cpp:459) in ONE pass, for the active systems k:
 *   w_k += sum_j c[j][k] ws[j]_k   (Gram-Schmidt against the cycle's directions; nj >= 0, coeffs as for qmg_batch_multi_caxpy_t)
 *   r_k += a[k] w_k                (a = -alpha; a[2k], a[2k+1] = re, im)
 *   z_next_k = r_k                 (z_next != NULL: the next search direction of an un-preconditioned GCR)
 * bit for bit qmg_batch_multi_caxpy_t, qmg_batch_blas_t(QMG_BOP_CAXPY), qmg_batch_blas_t(QMG_BOP_COPY) in that order.  w, r, z_next distinct. */
int qmg_batch_gcr_update_t(int dtype, const double* coeffs, const void* const* ws, int nj, void* w, const double* a, void* r, void* z_next,
                           size_t n, int nrhs, size_t stride, unsigned mask, void* stream);
int qmg_batch_reduce_t(int dtype, int op, const void* x, const void* y, size_t n, int nrhs, size_t stride, unsigned mask,
                       double* out_host, void* stream);
int qmg_batch_multidot_t(int dtype, const void* const* xs, int nj, const void* y, size_t n, int nrhs, size_t stride, unsigned mask,
                         double* out_host, void* stream);
/* The K-cycle's smoother, MR(omega) of quantum-linalg's minv_vector_minres (call sites multigrid/stateful_multigrid.h:851-860, 1037-1046),
 * with EVERY SCALAR ON THE DEVICE: the smoothers run a fixed number of iterations (their tolerance, 1e-15, is never met), so no host
 * decision depends on <p,r> / <p,p> and the host round trip of each iteration can go.  Per step, after p = A r:
 *   qmg_batch_mr_dots_t     <p_k,r_k>, <p_k,p_k> of the active systems into the calling thread's device slot (one pass over p and r;
 *                           same two-stage deterministic reduction, hence the same bits, as qmg_batch_multidot_t; summed over ranks
 *                           under distributed reductions).  A stencil apply with the MR epilogue (qmg_stencil_apply_epi_t) fills the
 *                           same slot from the apply's own pass instead.
 *   qmg_batch_mr_update_t   alpha_k = omega <p_k,r_k> / <p_k,p_k> formed on the device (0 when <p_k,p_k> == 0);
 *                           x (+)= alpha r_in ;  r_out = r_in - alpha p.  x_set: x = alpha r_in (first step from x0 = 0);
 *                           r_out == NULL: residual not wanted; r_out may alias r_in.
 *   qmg_batch_mr_read_dots  the slot (4 doubles per system: Re<p,r>, Im<p,r>, <p,p>, -) on the host; synchronises (tests). */
/* A stencil apply with an EPILOGUE on every finished site value of the processed parities (kernels B / B32: any nc but 1, 2, 4; one
 * system per launch), instead of separate BLAS-1 passes over the result:
 *   out = other_scale * other + acc_scale * acc   (other == NULL: out = acc_scale * acc)  -- the residual b - A x of stateful_multigrid.h:863-866 /
 *         1023-1029, the Schur complement's r_e - D'_eo t of stencil_2d.h:1894-1907;
 *   dotv != NULL: <out, dotv> and <out, out> go to the calling thread's MR slot (qmg_batch_mr_update_t consumes them) -- minv_vector_minres's
 *         <p,r> and <p,p> with p = out, r = dotv, from the apply's own pass.
 * other / dotv: vectors with lhs's layout, precision and stride (element `system * vec_stride` onwards is used, like lhs and rhs).  Needs
 * overwrite semantics (QMG_P_ZERO on the processed parities); lhs must differ from rhs, other and dotv.  mat32: the matrices are complex<float>
 * (qmg_stencil_apply_mat32's storage) with dtype's vectors.  QMG_ERR_UNSUPPORTED: this operator is not served with an epilogue (nc = 1, 2, 4:
 * see qmg_wilson_apply_direct_epi) -- run the separate passes. */
typedef struct { const void* other; double other_scale, acc_scale; const void* dotv; } qmg_apply_epilogue;
int qmg_stencil_apply_epi_t(int dtype, int mat32, const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces, size_t vec_stride, int system,
                            const qmg_apply_epilogue* epi, void* stream);
/* qmg_wilson_apply_direct / qmg_wilson_hops_direct with the same epilogue (kernel W, nc = 2, both storage precisions): ONE system per launch
 * (exactly one bit of `mask`), whole lattice or slab with all its rows (rows = 0). */
int qmg_wilson_apply_direct_epi(int dtype, const qmg_stencil_desc* d, const void* gauge, int gauge_Ly, int y0, double wilson_coeff, void* lhs, const void* rhs,
                                const void* halo_lo, const void* halo_hi, unsigned pieces, int nrhs, size_t vec_stride, size_t halo_stride, unsigned mask,
                                const qmg_apply_epilogue* epi, void* stream);
int qmg_wilson_hops_direct_epi(int dtype, const qmg_stencil_desc* d, const void* gauge, int gauge_Ly, int y0, double wilson_coeff, double hop_scale, void* lhs,
                               const void* rhs, const void* halo_lo, const void* halo_hi, unsigned pieces, int nrhs, size_t vec_stride, size_t halo_stride,
                               unsigned mask, const qmg_apply_epilogue* epi, void* stream);
int qmg_batch_mr_dots_t(int dtype, const void* r, const void* p, size_t n, int nrhs, size_t stride, unsigned mask, void* stream);
int qmg_batch_mr_update_t(int dtype, double omega, void* x, const void* r_in, void* r_out, const void* p, int x_set, size_t n, int nrhs,
                          size_t stride, unsigned mask, void* stream);
int qmg_batch_mr_read_dots(double* out_host, int nrhs, void* stream);
int qmg_prolong_batch_t(int dtype, const void* nullvecs, int nvec, const void* coarse, void* fine, int fLx, int fLy, int fnc,
                        int cLx, int cLy, int cnc, int nrhs, size_t cstride, size_t fstride, unsigned mask, void* stream);
int qmg_restrict_batch_t(int dtype, const void* nullvecs, int nvec, const void* fine, void* coarse, int fLx, int fLy, int fnc,
                         int cLx, int cLy, int cnc, int nrhs, size_t fstride, size_t cstride, unsigned mask, void* stream);
/* The same two operations (transfer/transfer.h:455-511) on complex<double> vectors with the null vectors STORED as complex<float> -- the narrow copy a
 * preconditioner level keeps of its prolongator (half of the transfer's bytes; arithmetic fp64: the result is the fp64 operation with the rounded
 * null vectors).  System by system through the single-vector kernels: meant for ONE active system (several share one read of the fp64 null vectors in
 * qmg_*_batch_t instead).  Even fnc and block width, null vectors 16-byte aligned; QMG_ERR_UNSUPPORTED otherwise. */
int qmg_prolong_batch_nv32(const void* nullvecs_c32, int nvec, const void* coarse, void* fine, int fLx, int fLy, int fnc,
                           int cLx, int cLy, int cnc, int nrhs, size_t cstride, size_t fstride, unsigned mask, void* stream);
int qmg_restrict_batch_nv32(const void* nullvecs_c32, int nvec, const void* fine, void* coarse, int fLx, int fLy, int fnc,
                            int cLx, int cLy, int cnc, int nrhs, size_t fstride, size_t cstride, unsigned mask, void* stream);

/* 16-bit storage of the fine operator (SURVEY 8f-4 "16-bit-storage smoother"; nc = 2 only): d->clover / d->hopping point to
 * complex<half> copies (qmg_convert_to_c16), vectors are complex<float>, arithmetic fp32: 112 B/site instead of 192.  The
 * rounding (2^-11) perturbs the OPERATOR, so this is for applies inside a preconditioner only. */
int qmg_convert_to_c16(void* dst_c16, const void* src, int src_dtype, size_t n, void* stream);
int qmg_convert_from_c16(void* dst, int dst_dtype, const void* src_c16, size_t n, void* stream);   /* complex<half> -> QMG_C64 / QMG_C32, exact */
int qmg_stencil_apply_h16(const qmg_stencil_desc* d, void* lhs, const void* rhs, unsigned pieces,
                          int nrhs, size_t vec_stride, unsigned mask, void* stream);

/* ---------------- multi-GPU: independent right-hand sides per rank ---------------- */
/* The path shards over right-hand sides; every rank holds a replica of the stencil/transfer data and
 * there is no halo exchange.  The one collective is a sum all-reduce (RCCL over xGMI) of a small
 * vector of per-RHS reduction results, once per Krylov reduction step, so all ranks take the same
 * convergence decision.  Rank 0 creates the 128-byte id and ships it by any host channel. */
int qmg_comm_get_unique_id(void* id128);
int qmg_comm_init(const void* id128, int world, int rank);      /* collective; after qmg_init(local_rank) */
/* qmg_comm_init with the id fetched through the launcher's rendezvous: QMG_COMM_ID_HEX (256 hex digits) if set, else one
 * TCP exchange with rank 0 on MASTER_ADDR : QMG_COMM_PORT (default MASTER_PORT + 1).  Bounded waits (QMG_COMM_TIMEOUT_S,
 * default 120 s): a missing rank is an error return, never a hang. */
int qmg_comm_init_env(int world, int rank);
int qmg_comm_rendezvous(void* blob128, int world, int rank);   /* the TCP leg alone: rank 0's 128 bytes reach every rank (host only) */
/* *all_ok = 1 iff every rank passed ok != 0 (one tiny all-reduce): lets an error on one rank stop all ranks together
 * instead of leaving the others blocked in the next collective. */
int qmg_comm_all_ok(int ok, int* all_ok);
int qmg_comm_world(int* world, int* rank);
int qmg_allreduce_sum_f64(double* buf_dev, size_t n, void* stream);   /* in place, async; no-op when world == 1 */
int qmg_comm_finalize(void);

/* ---------------- y-slab domain decomposition of ONE lattice ----------------
 * The reference is single-process and marks where this goes: cshift/cshift_2d.h:39-42,72,89 ("Becomes MPI").  Rank r of R
 * holds rows [r Ly/R, (r+1) Ly/R) of the global lattice as an ordinary even-odd lattice of Ly/R rows (Ly/R even, so the
 * colouring of a slab is the global one); every array keeps the layout of section 3.  Three pieces:
 *   qmg_halo_exchange         first / last row of a vector -> the neighbouring ranks' halo buffers (RCCL send/recv on xGMI;
 *                             one rank: device copies = the periodic wrap).  Halo buffer: [system][parity][Lx/2][nc] complex.
 *   qmg_stencil_apply_slab    the apply with rows -1 / Ly read from the halo buffers; `rows` splits it into the interior
 *                             (no halo needed: overlaps the exchange) and the two boundary rows.  nc = 2 in this round.
 *   qmg_comm_set_distributed_reductions   reductions of slab vectors are summed over the ranks inside the library.
 * Stencil arrays of a slab: qmg_wilson_fill_slab from the global gauge field (32 B/site, replicated). */
#define QMG_SLAB_H16 0x100   /* or-ed into the storage argument: matrices stored as complex<half> (vectors QMG_C32) */
#define QMG_SLAB_M32 0x200   /* nc != 2: matrices stored as complex<float> whatever the vectors' type (a preconditioner level's narrow Galerkin copy) */
#define QMG_SLAB_M16 0x400   /* nc != 2, a multiple of 4: matrices stored as complex<half> whatever the vectors' type */
int qmg_halo_exchange(int dtype, const void* vec, int Lx, int Ly_local, int nc, void* halo_lo, void* halo_hi, int nrhs, size_t vec_stride,
                      size_t halo_stride, void* stream);
/* rows of one parity only (parities: bit 0 even sites' rows, bit 1 odd sites' rows): what a D_eo / D_oe piece reads; the vectors of the
 * even-odd Schur systems are half-length, the other parity's rows do not exist */
int qmg_halo_exchange_parity(int dtype, const void* vec, int Lx, int Ly_local, int nc, void* halo_lo, void* halo_hi, int nrhs, size_t vec_stride,
                             size_t halo_stride, unsigned parities, void* stream);
int qmg_stencil_apply_slab(int storage, const qmg_stencil_desc* d, void* lhs, const void* rhs, const void* halo_lo, const void* halo_hi,
                           unsigned pieces, int nrhs, size_t vec_stride, size_t halo_stride, unsigned mask, int rows, void* stream);
int qmg_wilson_fill_slab(void* clover, void* hopping, const void* gauge_global, int Lx, int Ly_global, int y0, int Ly_local, double wilson_coeff,
                         void* stream);
int qmg_comm_set_distributed_reductions(int on);
/* setup on slabs: the Galerkin build with the null vectors' halo rows, and a Gaussian vector that equals the slab's rows of the
 * single-domain qmg_gaussian vector with the same seed (so a decomposed run draws the single-domain run's vectors) */
int qmg_coarse_build_slab(void* cclover, void* chopping, const qmg_stencil_desc* fine, const void* nullvecs, const void* restrict_vecs, int cLx, int cLy, int cnc,
                          const void* P_halo_lo, const void* P_halo_hi, size_t halo_stride, void* stream);
int qmg_gaussian_slab(void* x, int Lx, int Ly_global, int y0, int Ly_local, int nc, unsigned long long seed, void* stream);
/* right-block-Jacobi hopping on a slab: after qmg_build_rbjacobi(cinv, rb_clover, NULL, d) and a halo exchange of cinv (nc^2 components) */
int qmg_rb_hopping_slab(void* rb_hopping, const qmg_stencil_desc* d, const void* cinv, const void* cinv_halo_lo, const void* cinv_halo_hi, void* stream);
/* build_dagger_stencil (stencil_2d.h:1080-1139) on a y-slab.  ym_halo_hi: the `hi` buffer of qmg_halo_exchange applied to the -y hopping
 * field (hopping + 3 size_cm, nc^2 components per site); yp_halo_lo: the `lo` buffer of the exchange of the +y field (hopping + size_cm). */
int qmg_build_dagger_slab(void* dclover, void* dhopping, const void* clover, const void* hopping, int Lx, int Ly, int nc,
                          const void* ym_halo_hi, const void* yp_halo_lo, void* stream);
/* The nc = 1 operator fills on a y-slab (rows y0 .. y0 + Ly_local - 1, y0 even) from the GLOBAL gauge field, as qmg_wilson_fill_slab:
 * Staggered2D (staggered.h:50-72) and GaugedLaplace2D (gaugedlaplace.h:45-68). */
int qmg_staggered_fill_slab(void* hopping, const void* gauge_global, int Lx, int Ly_global, int y0, int Ly_local, void* stream);
int qmg_laplace_fill_slab(void* clover, void* hopping, const void* gauge_global, int Lx, int Ly_global, int y0, int Ly_local, void* stream);
/* Test transport: `world` host threads of one process act as ranks on one GPU (device copies + host sums behind thread barriers),
 * because one-GPU boxes cannot run two RCCL ranks.  Everything above the transport is the code the RCCL path runs. */
int qmg_comm_emulate_begin(int world);
int qmg_comm_emulate_attach(int rank);
int qmg_comm_emulate_end(void);

/* ---------------- the Wilson operator straight from the gauge links (csrc/qmg_wilson.hip, kernel W) ----------------
 * Same result as qmg_wilson_fill + qmg_stencil_apply (operators/wilson.h:153-209 + stencil_2d.h:912-936), without the stored
 * matrices: 96 B/site instead of 384 in fp64 (48 instead of 192 in fp32).  d: the vectors' lattice (nc = 2) and the shifts;
 * gauge: links in `dtype` on the lattice Lx x gauge_Ly, the vectors covering its rows y0 .. y0 + d->Ly - 1.  Whole lattice:
 * gauge_Ly = d->Ly, y0 = 0, halos NULL, rows 0.  y-slab: halos from qmg_halo_exchange, rows as in qmg_stencil_apply_slab.
 * Piece sets served: clover + every hop of the processed parities (shift pieces optional), or every hop alone;
 * QMG_ERR_UNSUPPORTED otherwise -- the stored stencil serves the rest. */
int qmg_wilson_apply_direct(int dtype, const qmg_stencil_desc* d, const void* gauge, int gauge_Ly, int y0, double wilson_coeff, void* lhs, const void* rhs,
                            const void* halo_lo, const void* halo_hi, unsigned pieces, int nrhs, size_t vec_stride, size_t halo_stride, unsigned mask, int rows,
                            void* stream);
/* The hops of the RIGHT-BLOCK-JACOBI Wilson stencil (stencil_2d.h:1556-1581, H'_mu(x) = H_mu(x) . cinv(x + mu)) from the links, for
 * a cinv that is `hop_scale` times the identity at every site (real mass, no eo / dof shift; hop_scale = the [0][0] entry
 * qmg_build_rbjacobi left in cinv): 128 instead of 320 B per written site for the D'_eo / D'_oe of a Schur-complement apply.
 * pieces: every hop of the processed parities and nothing else (QMG_ERR_UNSUPPORTED otherwise).  fp64: bit for bit the stored
 * right-block-Jacobi hopping through the site kernel. */
int qmg_wilson_hops_direct(int dtype, const qmg_stencil_desc* d, const void* gauge, int gauge_Ly, int y0, double wilson_coeff, double hop_scale, void* lhs,
                           const void* rhs, const void* halo_lo, const void* halo_hi, unsigned pieces, int nrhs, size_t vec_stride, size_t halo_stride,
                           unsigned mask, int rows, void* stream);

/* ---------------- tuning hooks (not part of the reference surface) ---------------- */
/* Dispatch / codegen knobs, all with the defaults the measurements in profiles/ chose; results never depend on them
 * beyond summation order.  Unknown keys return QMG_ERR_INVALID.
 *   "stencil_nt"    bit 0: non-temporal loads of the stencil matrices, bit 1: non-temporal stores, nc <= 4 kernels (3)
 *   "stencil_pair"  0: one site per lane group (kernel A); 1 / 2: both parities of a column on 1 / 2 rows per lane group (2)
 *   "pair_prefetch" 1: fp64 batches with nc = 1 request system k+1's right-hand side ahead of system k's arithmetic (1)
 *   "blas_nt_mb"    BLAS-1 kernels read their read-only operands non-temporally from vectors of this many MiB upwards, 0 = never (256)
 *   "stencil_rows"  cap on gridDim.y of the stencil kernels, 0 = one block row per lattice row (0)
 *   "stencil_mfma"  1: multi-rhs coarse applies (nc in 8,12,16,24,32; >= 4-5 systems) on the f64 matrix cores, 2-MFMA
 *                   packing for <= 8 systems; 2: plain 4-MFMA products; 0: vector-FMA kernel B only (1)
 *   "gen_sites"     cap on sites per block of kernel B, 0 = register-limited maximum (0)
 *   "gen32"         fp32-stored matrices, even nc: 1 = fp32 tile end to end (kernel B32), 2 = same with 2-site tiles,
 *                   0 = kernel B with widening loads (1)
 *   "stencil_site"  nc = 2 through the site kernel (csrc/qmg_site.hip): bit 0 fp64 where it is the faster one (hops-only,
 *                   one system), bit 1 fp32, bit 2 fp64 always (A/B measurements) (3)
 *   "site_block"    threads per block of the site kernel: 64, 128 or 256 (256); "site_gy": cap on its grid.y, 0 = rows (0);
 *                   "site_generic": 1 = its run-time-flag variant instead of the compile-time piece shapes (0)
 *   "mfma_vl"       kernel C: right-hand sides through an LDS slice (coalesced loads / stores) (1)
 *   "mfma_pair8"    kernel C at nc = 8 with 5-8 systems: a wavefront owns two adjacent sites (block-diagonal 16 x 16 tile) (1)
 *   "wilson_pair"   the full Wilson operator from the links: 0 = one site per lane group (kernel W), 1 = both parities of a column
 *                   (kernel W2), 2 = W2 on two rows per lane group for one system on an even run of rows, else as 1 (2)
 *   "xfer_pack"     1: complex<float> single-system restrict / prolong move two elements per lane (16-byte accesses) (1)
 *   "xfer_tile"     1: batched restrict / prolong as LDS-tiled kernels; 0: the one-system kernels, system by system (1)
 *   "xfer_mfma"     batched restrict / prolong (2-16 systems) as contractions on the f64 matrix cores (even block width, <= 32 null vectors, a
 *                   chunk of the tile within 60 KB of LDS): 1 = where that is the faster kernel (the complex<float> restrict), 2 = every served
 *                   shape, 0 = never (1)
 *   "setup_fused"   1: block-local setup kernels (block orthonormalisation in LDS, Galerkin build as per-block products);
 *                   0: the full-lattice restrict / prolong / probe passes of the reference's formulation (1)
 *   "malloc_poison" 1: qmg_malloc fills every allocation with 0xFF bytes (NaNs in every storage precision): a buffer read before
 *                   it is written then shows deterministically (0)
 *   "reduce_spin"   1: the host picks the results of the batch reductions (qmg_batch_reduce*, qmg_batch_multidot*) up by polling a sequence
 *                   number the final stage's last block publishes in coherent host memory behind them; the stream is NOT synchronised by
 *                   these calls then (later launches are ordered behind the kernel anyway).  0: hipStreamSynchronize, as before (1)
 * (The ablation switch of tools/variants.py exists only in the tools build, `make DIAG=1`; this library has no such key.) */
/* (QMG_TUNING="key=value,key=value" in the environment applies the same settings inside qmg_init -- for A/B runs of programs that do not call this.) */
int qmg_set_tuning(const char* key, int value);

#ifdef __cplusplus
}
#endif
#endif /* QMG_HIP_H */
