// multigrid.hpp -- ArrayStorageMG, MultigridMG and StatefulMultigridMG (with the K-cycle
// `mg_preconditioner`) on device vectors.  Reference: storage/array_storage.h:23-155,
// multigrid/multigrid.h:54-600, multigrid/stateful_multigrid.h:43-1062.  ARPACK deflation
// (stateful_multigrid.h:611-712) is out of scope (no ARPACK; the reference guards it with NO_ARPACK).
#ifndef QMG_MULTIGRID_HPP
#define QMG_MULTIGRID_HPP

#include <map>
#include <string>
#include <vector>

#include "coarse.hpp"
#include "krylov.hpp"

using std::vector;
using std::to_string;

// ---------------- device-vector pool (array_storage.h) ----------------
template <typename T>
class ArrayStorageMG {
 private:
  ArrayStorageMG(ArrayStorageMG const&);
  ArrayStorageMG& operator=(ArrayStorageMG const&);
  const long array_length;
  int allocated_arrays;
  vector<bool> is_checked_out;
  int n_checked;
  vector<T*> arrays;

 public:
  ArrayStorageMG(const long length, const int n_prealloc = 1) : array_length(length), allocated_arrays(n_prealloc), n_checked(0) {
    if (n_prealloc < 1) std::cout << "[QMG-ERROR]: ArrayStorageMG cannot preallocate less than one vector.\n";
    for (int i = 0; i < n_prealloc; i++) { arrays.push_back(allocate_vector<T>(array_length)); is_checked_out.push_back(false); }
  }
  ~ArrayStorageMG() { for (int i = 0; i < allocated_arrays; i++) deallocate_vector(&arrays[i]); }
  T* check_out() {
    for (int i = 0; i < allocated_arrays; i++)
      if (!is_checked_out[i]) { is_checked_out[i] = true; n_checked++; return arrays[i]; }
    arrays.push_back(allocate_vector<T>(array_length));
    is_checked_out.push_back(true);
    n_checked++;
    return arrays[allocated_arrays++];
  }
  void check_in(T* arr) {
    for (int i = 0; i < allocated_arrays; i++)
      if (arrays[i] == arr) {
        if (is_checked_out[i]) { is_checked_out[i] = false; n_checked--; }
        else cout << "[QMG_WARNING]: Returned array that wasn't checked out.\n";
        return;
      }
    cout << "[QMG_WARNING]: Returned array that doesn't live in library.\n";
  }
  int get_number_allocated() { return allocated_arrays; }
  int get_number_checked() { return n_checked; }
  void consolidate(int minimum = 1) {   // :128-154 (the reference also decrements n_checked for a free array; not repeated)
    for (int i = allocated_arrays - 1; i > 0; i--) {
      if (allocated_arrays <= minimum) break;
      if (!is_checked_out[i]) {
        deallocate_vector(&arrays[i]);
        arrays.erase(arrays.begin() + i);
        is_checked_out.erase(is_checked_out.begin() + i);
        allocated_arrays--;
      }
    }
  }
};

// ---------------- level bookkeeping (multigrid.h) ----------------
class MultigridMG {
 protected:
  MultigridMG(MultigridMG const&);
  MultigridMG& operator=(MultigridMG const&);
  int num_levels;
  vector<Lattice2D*> lattice_list;
  vector<TransferMG*> transfer_list;
  vector<Stencil2D*> stencil_list;
  vector<bool> is_stencil_managed;
  vector<ArrayStorageMG<complex<double>>*> storage_list;
  vector<complex<double>**> global_null_vectors;

  complex<double>** copy_global(complex<double>** nvecs, int n, Lattice2D* fine) {
    if (nvecs == 0) return 0;
    complex<double>** out = new complex<double>*[n];
    for (int j = 0; j < n; j++) {
      out[j] = 0;
      if (nvecs[j] != 0) { out[j] = allocate_vector<complex<double>>(fine->get_size_cv_l()); copy_vector(out[j], nvecs[j], fine->get_size_cv_l()); }
    }
    return out;
  }
  void free_global(int level_fine) {
    if (global_null_vectors[level_fine] == 0) return;
    const int n = lattice_list[level_fine + 1]->get_nc();
    for (int j = 0; j < n; j++) if (global_null_vectors[level_fine][j] != 0) deallocate_vector(&global_null_vectors[level_fine][j]);
    delete[] global_null_vectors[level_fine];
    global_null_vectors[level_fine] = 0;
  }

 public:
  // Storage precision of the Galerkin coarse operators (not in the reference, which is fp64 throughout): with it ON every coarse
  // operator built from now on keeps a complex<float> copy of its matrices and streams that in ORIGINAL-operator applies
  // (Stencil2D::enable_f32_matrices) -- half the bytes of an HBM-bound apply.  Vectors, shifts, arithmetic and the Galerkin
  // builds stay fp64; an apply then equals the fp64 apply of the ROUNDED matrices (relative 6e-8 per entry).
  //   -1 (default): ON for a StatefulMultigridMG -- a hierarchy that only PRECONDITIONS a flexible fp64 outer solve, whose true
  //      residual still reaches its fp64 tolerance (C3: the same 13 outer iterations, 1e-10) -- and OFF for a plain MultigridMG,
  //      whose coarse operators are used as operators (Galerkin identity R A P to 1e-15: n08);
  //    0 / 1: off / on for both (drivers: QMG_COARSE_F32=0 / 1);
  //    2: ON with complex<half> copies (Stencil2D::enable_f32_matrices(16): a quarter of the fp64 stream; levels whose nc or value range
  //       does not allow it keep complex<float>) -- opt-in (drivers: QMG_COARSE_BITS=16).
  static int& coarse_f32_storage() { static int f = -1; return f; }
  virtual bool coarse_f32_default() const { return false; }
  // (nc = 1, 2, 4 have no fp32-stored kernel; y-slabs stream the narrow copies like a whole lattice: qmg_stencil_apply_slab with QMG_SLAB_M32 / _M16)
  bool coarse_f32_wanted(int nc) const {
    if (nc == 1 || nc == 2 || nc == 4) return false;
    const int f = coarse_f32_storage();
    return f < 0 ? coarse_f32_default() : f != 0;
  }
  // true if any level of this hierarchy streams complex<float> Galerkin matrices (drivers print it)
  bool any_coarse_f32() { for (int i = 1; i < num_levels; i++) if (stencil_list[i] && stencil_list[i]->f32_matrices) return true; return false; }
  bool any_coarse_f16() { for (int i = 1; i < num_levels; i++) if (stencil_list[i] && stencil_list[i]->f32_matrices && stencil_list[i]->f32_bits == 16) return true; return false; }
  // drivers: QMG_COARSE_F32=0 / 1 and QMG_COARSE_BITS=64 / 32 / 16 (the latter wins)
  static void coarse_storage_from_env() {
    if (getenv("QMG_COARSE_F32")) coarse_f32_storage() = atoi(getenv("QMG_COARSE_F32")) != 0 ? 1 : 0;
    if (getenv("QMG_COARSE_BITS")) { const int b = atoi(getenv("QMG_COARSE_BITS")); coarse_f32_storage() = (b == 16) ? 2 : (b == 32) ? 1 : 0; }
  }

  enum QMGMultigridPrecondStencil { QMG_MULTIGRID_PRECOND_ORIGINAL = 0, QMG_MULTIGRID_PRECOND_RIGHT_BLOCK_JACOBI = 1 };

  MultigridMG(Lattice2D* in_lat, Stencil2D* in_stencil) {
    num_levels = 1;
    lattice_list.push_back(in_lat);
    storage_list.push_back(new ArrayStorageMG<complex<double>>(in_lat->get_size_cv_l(), 6));
    stencil_list.push_back(in_stencil);
    is_stencil_managed.push_back(false);
  }
  virtual ~MultigridMG() {
    for (int i = 0; i < num_levels; i++) {
      if (storage_list[i] != 0) delete storage_list[i];
      if (is_stencil_managed[i] && stencil_list[i] != 0) delete stencil_list[i];
    }
    for (int i = 0; i < num_levels - 1; i++) free_global(i);
  }

  int get_num_levels() { return num_levels; }
  Lattice2D* get_lattice(int i) { if (i >= 0 && i < num_levels) return lattice_list[i]; cout << "[QMG-ERROR]: Out of range: Lattice2D level " << i << " does not exist in MultigridMG object.\n"; return 0; }
  TransferMG* get_transfer(int i) { if (i >= 0 && i < num_levels - 1) return transfer_list[i]; cout << "[QMG-ERROR]: Out of range: TransferMG level " << i << " does not exist in MultigridMG object.\n"; return 0; }
  Stencil2D* get_stencil(int i) { if (i >= 0 && i < num_levels) return stencil_list[i]; cout << "[QMG-ERROR]: Out of range: Stencil2D level " << i << " does not exist in MultigridMG object.\n"; return 0; }
  ArrayStorageMG<complex<double>>* get_storage(int i) { if (i >= 0 && i < num_levels) return storage_list[i]; cout << "[QMG-ERROR]: Out of range: ArrayStorageMG level " << i << " does not exist in MultigridMG object.\n"; return 0; }
  complex<double>** get_global_null_vectors(int i) { if (i >= 0 && i < num_levels - 1) return global_null_vectors[i]; cout << "[QMG-ERROR]: Out of range: global null vectors level " << i << " does not exist in MultigridMG object.\n"; return 0; }

  // multigrid.h:257-302
  void push_level(Lattice2D* new_lat, TransferMG* new_transfer, bool build_stencil, bool is_chiral, QMGMultigridPrecondStencil build_stencil_from,
                  CoarseOperator2D::QMGCoarseBuildStencil build_extra, complex<double>** nvecs = 0) {
    num_levels++;
    lattice_list.push_back(new_lat);
    transfer_list.push_back(new_transfer);
    storage_list.push_back(new ArrayStorageMG<complex<double>>(new_lat->get_size_cv_l(), 6));
    if (build_stencil) {
      stencil_list.push_back(new CoarseOperator2D(new_lat, stencil_list[num_levels - 2], lattice_list[num_levels - 2], new_transfer, is_chiral,
                                                  build_stencil_from != QMG_MULTIGRID_PRECOND_ORIGINAL, build_extra));
      if (coarse_f32_wanted(new_lat->get_nc())) { stencil_list.back()->enable_f32_matrices(coarse_f32_storage() == 2 ? 16 : 32); new_transfer->enable_narrow_precond(); }
      is_stencil_managed.push_back(true);
    } else {
      stencil_list.push_back(0);
      is_stencil_managed.push_back(false);
    }
    global_null_vectors.push_back(copy_global(nvecs, new_lat->get_nc(), lattice_list[num_levels - 2]));
  }
  void push_level(Lattice2D* new_lat, TransferMG* new_transfer, bool build_stencil = false, bool is_chiral = false,
                  QMGMultigridPrecondStencil build_stencil_from = QMG_MULTIGRID_PRECOND_ORIGINAL, complex<double>** nvecs = 0) {
    push_level(new_lat, new_transfer, build_stencil, is_chiral, build_stencil_from, CoarseOperator2D::QMG_COARSE_BUILD_ORIGINAL, nvecs);
  }
  void push_level(Lattice2D* new_lat, TransferMG* new_transfer, complex<double>** nvecs) {
    push_level(new_lat, new_transfer, false, false, QMG_MULTIGRID_PRECOND_ORIGINAL, CoarseOperator2D::QMG_COARSE_BUILD_ORIGINAL, nvecs);
  }

  void pop_level() {   // :324-372
    if (num_levels == 1) { std::cout << "[QMG-ERROR]: In MultigridMG::pop_level, cannot pop when there is only one level.\n"; return; }
    const int i = num_levels - 1;
    if (storage_list[i] != 0) delete storage_list[i];
    storage_list.pop_back();
    if (is_stencil_managed[i] && stencil_list[i] != 0) delete stencil_list[i];
    stencil_list.pop_back();
    is_stencil_managed.pop_back();
    free_global(i - 1);
    global_null_vectors.pop_back();
    transfer_list.pop_back();
    lattice_list.pop_back();
    num_levels--;
  }

  void update_level(int level, Lattice2D* new_lat, TransferMG* new_transfer, bool build_stencil, bool is_chiral,
                    QMGMultigridPrecondStencil build_stencil_from, CoarseOperator2D::QMGCoarseBuildStencil build_extra, complex<double>** nvecs = 0) {   // :375-450
    if (level < 1 || level >= num_levels) {
      std::cout << "[QMG-ERROR]: In MultigridMG::update_level, cannot update level " << level << " as it does not exist yet anyway.\n";
      return;
    }
    if (storage_list[level] != 0) delete storage_list[level];
    if (is_stencil_managed[level] && stencil_list[level] != 0) delete stencil_list[level];
    free_global(level - 1);
    lattice_list[level] = new_lat;
    transfer_list[level - 1] = new_transfer;
    storage_list[level] = new ArrayStorageMG<complex<double>>(new_lat->get_size_cv_l(), 6);
    if (build_stencil) {
      stencil_list[level] = new CoarseOperator2D(new_lat, stencil_list[level - 1], lattice_list[level - 1], new_transfer, is_chiral,
                                                 build_stencil_from != QMG_MULTIGRID_PRECOND_ORIGINAL, build_extra);
      if (coarse_f32_wanted(new_lat->get_nc())) { stencil_list[level]->enable_f32_matrices(coarse_f32_storage() == 2 ? 16 : 32); new_transfer->enable_narrow_precond(); }
      is_stencil_managed[level] = true;
    } else {
      stencil_list[level] = 0;
      is_stencil_managed[level] = false;
    }
    global_null_vectors[level - 1] = copy_global(nvecs, new_lat->get_nc(), lattice_list[level - 1]);
  }
  void update_level(int level, Lattice2D* new_lat, TransferMG* new_transfer, bool build_stencil = false, bool is_chiral = false,
                    QMGMultigridPrecondStencil build_stencil_from = QMG_MULTIGRID_PRECOND_ORIGINAL, complex<double>** nvecs = 0) {
    update_level(level, new_lat, new_transfer, build_stencil, is_chiral, build_stencil_from, CoarseOperator2D::QMG_COARSE_BUILD_ORIGINAL, nvecs);
  }

  // lhs += A_i rhs; a level whose stencil was not built is emulated as R A_{i-1} P (:465-512)
  void apply_stencil(complex<double>* lhs, complex<double>* rhs, int i, QMGStencilType app_type = QMG_MATVEC_ORIGINAL) {
    if (i < 0 || i >= num_levels) { cout << "[QMG-ERROR]: Out of range: Cannot apply stencil at level " << i << "\n"; return; }
    if (stencil_list[i] != 0) { stencil_list[i]->apply_M(lhs, rhs, app_type); return; }
    if (app_type != QMG_MATVEC_ORIGINAL) { std::cout << "[QMG-ERROR]: In MultigridMG::apply_stencil, the emulated operator must be QMG_MATVEC_ORIGINAL.\n"; return; }
    complex<double>* pro_rhs = storage_list[i - 1]->check_out();
    complex<double>* Apro_rhs = storage_list[i - 1]->check_out();
    zero_vector(pro_rhs, lattice_list[i - 1]->get_size_cv_l());
    zero_vector(Apro_rhs, lattice_list[i - 1]->get_size_cv_l());
    transfer_list[i - 1]->prolong_c2f(rhs, pro_rhs);
    apply_stencil(Apro_rhs, pro_rhs, i - 1);
    transfer_list[i - 1]->restrict_f2c(Apro_rhs, lhs);
    storage_list[i - 1]->check_in(pro_rhs);
    storage_list[i - 1]->check_in(Apro_rhs);
  }
  void prolong_c2f(complex<double>* coarse_cv, complex<double>* fine_cv, int i) {
    if (i >= 0 && i < num_levels - 1) transfer_list[i]->prolong_c2f(coarse_cv, fine_cv);
    else cout << "[QMG-ERROR]: Out of range: Cannot apply prolong at level " << i << "\n";
  }
  void restrict_f2c(complex<double>* fine_cv, complex<double>* coarse_cv, int i) {
    if (i >= 0 && i < num_levels - 1) transfer_list[i]->restrict_f2c(fine_cv, coarse_cv);
    else cout << "[QMG-ERROR]: Out of range: Cannot apply prolong at level " << i << "\n";
  }
  complex<double>* check_out(int i) { if (i >= 0 && i < num_levels) return storage_list[i]->check_out(); cout << "[QMG-ERROR]: Out of range: Cannot check out vector at level " << i << ".\n"; return 0; }
  void check_in(complex<double>* vec, int i) { if (i >= 0 && i < num_levels) storage_list[i]->check_in(vec); else cout << "[QMG-ERROR]: Out of range: Cannot check in vector at level " << i << ".\n"; }
  int get_storage_number_allocated(int i) { if (i >= 0 && i < num_levels) return storage_list[i]->get_number_allocated(); cout << "[QMG-ERROR]: Out of range: Cannot query number of allocated arrays at level " << i << ".\n"; return -1; }
  int get_storage_number_checked(int i) { if (i >= 0 && i < num_levels) return storage_list[i]->get_number_checked(); cout << "[QMG-ERROR]: Out of range: Cannot query number of checked out arrays at level " << i << ".\n"; return -1; }
};

// ---------------- solve state + K-cycle (stateful_multigrid.h) ----------------
class StatefulMultigridMG;
// batch.hpp: the K-cycle engine every supported configuration runs on (one system = a batch of one); false = not served there
inline bool qmg_kcycle_via_batch(StatefulMultigridMG* mg, complex<double>* lhs, complex<double>* rhs, int size, inversion_verbose_struct* verb);
enum QMGDslashType { QMG_DSLASH_TYPE_NULLVEC = 0, QMG_DSLASH_TYPE_KRYLOV = 1, QMG_DSLASH_TYPE_PRESMOOTH = 2, QMG_DSLASH_TYPE_POSTSMOOTH = 3 };

class StatefulMultigridMG : public MultigridMG {
 private:
  StatefulMultigridMG(StatefulMultigridMG const&);
  StatefulMultigridMG& operator=(StatefulMultigridMG const&);
  int current_level;

 public:
  struct LevelSolveMG {   // :62-114
    QMGStencilType fine_stencil_app;
    double intermediate_tol; int intermediate_iters; int intermediate_restart_freq;
    double pre_tol; int pre_iters; bool pre_cgne;
    double post_tol; int post_iters; bool post_cgne;
    LevelSolveMG() : fine_stencil_app(QMG_MATVEC_ORIGINAL), intermediate_tol(1e-20), intermediate_iters(10000000), intermediate_restart_freq(32),
                     pre_tol(1e-20), pre_iters(1000000), pre_cgne(false), post_tol(1e-20), post_iters(1000000), post_cgne(false) {}
  };

  class DslashTrackerMG {   // :118-200
    std::map<QMGDslashType, int> tracker;
    int iterations, total;
   public:
    DslashTrackerMG() { reset_tracker(); }
    void add_tracker_count(QMGDslashType type, int accum) { tracker[type] += accum; total += accum; }
    void add_iterations_count(int accum) { iterations += accum; }
    void shift_all_to_nullvec() {
      tracker[QMG_DSLASH_TYPE_NULLVEC] += tracker[QMG_DSLASH_TYPE_KRYLOV] + tracker[QMG_DSLASH_TYPE_PRESMOOTH] + tracker[QMG_DSLASH_TYPE_POSTSMOOTH];
      tracker[QMG_DSLASH_TYPE_KRYLOV] = tracker[QMG_DSLASH_TYPE_PRESMOOTH] = tracker[QMG_DSLASH_TYPE_POSTSMOOTH] = 0;
      iterations = 0;
    }
    int get_tracker_count(QMGDslashType type) { return tracker[type]; }
    int get_total_count() { return total; }
    int get_iterations_count() { return iterations; }
    void reset_tracker() {
      tracker[QMG_DSLASH_TYPE_NULLVEC] = tracker[QMG_DSLASH_TYPE_KRYLOV] = tracker[QMG_DSLASH_TYPE_PRESMOOTH] = tracker[QMG_DSLASH_TYPE_POSTSMOOTH] = 0;
      total = 0; iterations = 0;
    }
  };

  struct CoarsestSolveMG {   // :204-241 (the `deflate` member exists only with ARPACK)
    QMGStencilType coarsest_stencil_app;
    double coarsest_tol; int coarsest_iters; int coarsest_restart_freq;
    double normal_shift;
    CoarsestSolveMG() : coarsest_stencil_app(QMG_MATVEC_ORIGINAL), coarsest_tol(1e-20), coarsest_iters(100000000), coarsest_restart_freq(32), normal_shift(0.0) {}
  };

 protected:
  vector<LevelSolveMG*> level_solve_list;
  vector<DslashTrackerMG*> dslash_tracker_list;
  CoarsestSolveMG* coarsest_solve;

  static bool valid_fine_app(LevelSolveMG* s) {
    return s->fine_stencil_app == QMG_MATVEC_ORIGINAL || s->fine_stencil_app == QMG_MATVEC_RIGHT_JACOBI || s->fine_stencil_app == QMG_MATVEC_RIGHT_SCHUR;
  }

 public:
  StatefulMultigridMG(Lattice2D* in_lat, Stencil2D* in_stencil, CoarsestSolveMG* in_coarsest_solve)
      : MultigridMG(in_lat, in_stencil), current_level(0), coarsest_solve(in_coarsest_solve) {
    dslash_tracker_list.push_back(new DslashTrackerMG());
  }
  ~StatefulMultigridMG() { for (size_t i = 0; i < dslash_tracker_list.size(); i++) delete dslash_tracker_list[i]; }
  virtual bool coarse_f32_default() const { return true; }   // this hierarchy is a preconditioner (mg_preconditioner)

  void set_multigrid_level(int level) {
    if (level >= 0 && level < get_num_levels()) current_level = level;
    else cout << "[QMG-ERROR]: Out of range: StatefulMultigridMG->current_level " << level << " is outside of [0,max_level-1].\n";
  }
  void go_finer() { if (current_level > 0) current_level--; else cout << "[QMG-ERROR]: Out of range: Cannot go finer than the top level in StatefulMultigridMG.\n"; }
  void go_coarser() { if (current_level < get_num_levels() - 2) current_level++; else cout << "[QMG-ERROR]: Out of range: Cannot go coarser than the second-coarsest level in StatefulMultigridMG.\n"; }
  int get_multigrid_level() { return current_level; }
  LevelSolveMG* get_level_solve(int i) {
    if (i >= 0 && i < get_num_levels() - 1 && level_solve_list[i] != 0) return level_solve_list[i];
    cout << "[QMG-ERROR]: Out of range: LevelSolveMG level " << i << " does not exist in StatefulMultigridMG object.\n";
    return 0;
  }
  LevelSolveMG* get_level_solve() { return get_level_solve(current_level); }
  CoarsestSolveMG* get_coarsest_solve() { return coarsest_solve; }

  // the six push_level flavours (:374-445)
  void push_level(Lattice2D* l, TransferMG* t, bool build_stencil, bool is_chiral, QMGMultigridPrecondStencil from, CoarseOperator2D::QMGCoarseBuildStencil extra, complex<double>** nvecs = 0) {
    MultigridMG::push_level(l, t, build_stencil, is_chiral, from, extra, nvecs);
    level_solve_list.push_back(0);
    dslash_tracker_list.push_back(new DslashTrackerMG());
  }
  void push_level(Lattice2D* l, TransferMG* t, bool build_stencil = false, bool is_chiral = false, QMGMultigridPrecondStencil from = QMG_MULTIGRID_PRECOND_ORIGINAL, complex<double>** nvecs = 0) {
    push_level(l, t, build_stencil, is_chiral, from, CoarseOperator2D::QMG_COARSE_BUILD_ORIGINAL, nvecs);
  }
  void push_level(Lattice2D* l, TransferMG* t, complex<double>** nvecs) { push_level(l, t, false, false, QMG_MULTIGRID_PRECOND_ORIGINAL, CoarseOperator2D::QMG_COARSE_BUILD_ORIGINAL, nvecs); }
  void push_level(Lattice2D* l, TransferMG* t, LevelSolveMG* in_solve, bool build_stencil, bool is_chiral, QMGMultigridPrecondStencil from, CoarseOperator2D::QMGCoarseBuildStencil extra, complex<double>** nvecs = 0) {
    MultigridMG::push_level(l, t, build_stencil, is_chiral, from, extra, nvecs);
    if (!valid_fine_app(in_solve)) std::cout << "[QMG-ERROR]: In StatefulMultigridMG:;push_level, LevelSolveMG::fine_stencil_app should only be original, right jacobi, or schur.\n";
    level_solve_list.push_back(in_solve);
    dslash_tracker_list.push_back(new DslashTrackerMG());
  }
  void push_level(Lattice2D* l, TransferMG* t, LevelSolveMG* in_solve, bool build_stencil = false, bool is_chiral = false, QMGMultigridPrecondStencil from = QMG_MULTIGRID_PRECOND_ORIGINAL, complex<double>** nvecs = 0) {
    push_level(l, t, in_solve, build_stencil, is_chiral, from, CoarseOperator2D::QMG_COARSE_BUILD_ORIGINAL, nvecs);
  }
  void push_level(Lattice2D* l, TransferMG* t, LevelSolveMG* in_solve, complex<double>** nvecs) {
    push_level(l, t, in_solve, false, false, QMG_MULTIGRID_PRECOND_ORIGINAL, CoarseOperator2D::QMG_COARSE_BUILD_ORIGINAL, nvecs);
  }
  void pop_level() {
    level_solve_list.pop_back();
    delete dslash_tracker_list.back();
    dslash_tracker_list.pop_back();
    MultigridMG::pop_level();
  }
  void update_level(int level, Lattice2D* l, TransferMG* t, LevelSolveMG* in_solve, bool build_stencil, bool is_chiral, QMGMultigridPrecondStencil from, CoarseOperator2D::QMGCoarseBuildStencil extra, complex<double>** nvecs = 0) {
    if (!valid_fine_app(in_solve)) { std::cout << "[QMG-ERROR]: In StatefulMultigridMG:;update_level, LevelSolveMG::fine_stencil_app should only be original, right jacobi, or schur.\n"; return; }
    MultigridMG::update_level(level, l, t, build_stencil, is_chiral, from, extra, nvecs);
    level_solve_list[level - 1] = in_solve;
  }
  void update_level(int level, Lattice2D* l, TransferMG* t, LevelSolveMG* in_solve, bool build_stencil = false, bool is_chiral = false, QMGMultigridPrecondStencil from = QMG_MULTIGRID_PRECOND_ORIGINAL, complex<double>** nvecs = 0) {
    update_level(level, l, t, in_solve, build_stencil, is_chiral, from, CoarseOperator2D::QMG_COARSE_BUILD_ORIGINAL, nvecs);
  }

  // trackers (:500-609)
  bool in_range(int i, const char* what) { if (i >= 0 && i < num_levels) return true; cout << "[QMG-ERROR]: Out of range: Cannot " << what << " at level " << i << ".\n"; return false; }
  void add_tracker_count(QMGDslashType type, int accum, int i) { if (in_range(i, "update tracker")) dslash_tracker_list[i]->add_tracker_count(type, accum); }
  void add_iterations_count(int accum, int i) { if (in_range(i, "update tracker")) dslash_tracker_list[i]->add_iterations_count(accum); }
  void shift_all_to_nullvec(int i) { if (in_range(i, "shift to null vectors")) dslash_tracker_list[i]->shift_all_to_nullvec(); }
  int get_tracker_count(QMGDslashType type, int i) { return in_range(i, "query tracker") ? dslash_tracker_list[i]->get_tracker_count(type) : -1; }
  int get_total_count(int i) { return in_range(i, "query tracker") ? dslash_tracker_list[i]->get_total_count() : -1; }
  int get_iterations_count(int i) { return in_range(i, "query tracker") ? dslash_tracker_list[i]->get_iterations_count() : -1; }
  std::vector<double> query_average_iterations() {
    std::vector<double> avg(num_levels);
    avg[0] = dslash_tracker_list[0]->get_iterations_count();
    for (int i = 1; i < num_levels; i++) avg[i] = ((double)dslash_tracker_list[i]->get_iterations_count()) / ((double)dslash_tracker_list[i - 1]->get_iterations_count());
    return avg;
  }
  void reset_tracker(int i = -1) {
    if (i == -1) { for (int j = 0; j < num_levels; j++) dslash_tracker_list[j]->reset_tracker(); }
    else if (in_range(i, "reset tracker")) dslash_tracker_list[i]->reset_tracker();
  }

 protected:
  struct ShiftedFunctionStruct { matrix_op_cplx function; void* extra_data; complex<double> extra_shift; int length; };
  static void shift_function(complex<double>* out, complex<double>* in, void* data) {   // :724-729
    ShiftedFunctionStruct* s = (ShiftedFunctionStruct*)data;
    s->function(out, in, s->extra_data);
    caxpy(s->extra_shift, in, out, s->length);
  }

 public:
  // One K-cycle application, lhs ~= A^-1 rhs (stateful_multigrid.h:734-1060), every step a device kernel.
  static void mg_preconditioner(complex<double>* lhs, complex<double>* rhs, int size, void* extra_data, inversion_verbose_struct* verb) {
    StatefulMultigridMG* mg = (StatefulMultigridMG*)extra_data;
    const int level = mg->get_multigrid_level();
    Stencil2D* fine_stencil = mg->get_stencil(level);
    const int total_num_levels = mg->get_num_levels();

    LevelSolveMG* level_solve = (total_num_levels > 1) ? mg->get_level_solve() : 0;
    if (total_num_levels > 1 && level_solve == 0) { std::cout << "[QMG-MG-SOLVE-ERROR]: Level solve for level " << level << " does not exist.\n"; return; }
    const long fine_size = mg->get_lattice(level)->get_size_cv_l();
    if (total_num_levels == 1) { copy_vector(lhs, rhs, fine_size); return; }   // :803-807
    // ONE engine: every configuration runs in the lock-step batch engine as a batch of one system (batch.hpp: same algorithm step for
    // step, the fixed-count smoothers' scalars on the device, apply epilogues).  What continues below is the reference-shaped
    // single-vector code: QMG_KCYCLE_ENGINE=single (A/B runs, the digit-for-digit slab comparisons of the tests), QMG_KCYCLE_SLAB_ENGINE=single,
    // and a hierarchy whose level types name a variant stencil that was not built (it warns exactly as the reference does).
    if (qmg_kcycle_via_batch(mg, lhs, rhs, size, verb)) return;

    Stencil2D* coarse_stencil = mg->get_stencil(level + 1);
    TransferMG* transfer = mg->get_transfer(level);
    ArrayStorageMG<complex<double>>* fine_storage = mg->get_storage(level);
    ArrayStorageMG<complex<double>>* coarse_storage = mg->get_storage(level + 1);
    const long coarse_size = mg->get_lattice(level + 1)->get_size_cv_l();

    inversion_info invif;
    inversion_verbose_struct verb2(VERB_SUMMARY, std::string(" "));
    if (verb == 0 || verb->verbosity == VERB_NONE) { verb2.verbosity = VERB_NONE; verb2.precond_verbosity = VERB_NONE; }
    else verb2.precond_verbosity = VERB_SUMMARY;
    verb2.verb_prefix = "  ";
    for (int i = 1; i < level + 1; i++) verb2.verb_prefix += "  ";
    verb2.verb_prefix += "[QMG-MG-SOLVE-INFO]: Level " + to_string(level + 1) + " ";

    const int n_pre_smooth = level_solve->pre_iters, n_post_smooth = level_solve->post_iters;
    const double pre_smooth_tol = level_solve->pre_tol, post_smooth_tol = level_solve->post_tol;
    const bool pre_cgne = level_solve->pre_cgne, post_cgne = level_solve->post_cgne;
    const QMGStencilType fine_type = level_solve->fine_stencil_app;
    matrix_op_cplx apply_fine_M = Stencil2D::get_apply_function(fine_type);
    long fine_size_solve = fine_size;
    if (fine_type == QMG_MATVEC_RIGHT_SCHUR) fine_size_solve /= 2;

    int coarse_max_iter, coarse_restart;
    double coarse_tol;
    QMGStencilType coarse_type;
    if (level < total_num_levels - 2) {
      LevelSolveMG* cs = mg->get_level_solve(level + 1);
      coarse_type = cs->fine_stencil_app; coarse_max_iter = cs->intermediate_iters; coarse_tol = cs->intermediate_tol; coarse_restart = cs->intermediate_restart_freq;
    } else {
      CoarsestSolveMG* cs = mg->get_coarsest_solve();
      coarse_type = cs->coarsest_stencil_app; coarse_max_iter = cs->coarsest_iters; coarse_tol = cs->coarsest_tol; coarse_restart = cs->coarsest_restart_freq;
    }
    matrix_op_cplx apply_coarse_M = Stencil2D::get_apply_function(coarse_type);
    long coarse_size_solve = coarse_size;
    if (coarse_type == QMG_MATVEC_RIGHT_SCHUR) coarse_size_solve /= 2;

    complex<double>* Atmp = fine_storage->check_out();
    complex<double>* z1 = fine_storage->check_out();
    zero_vector(z1, fine_size);
    complex<double>* r1 = fine_storage->check_out();

    // ---- 1. pre-smooth: A z1 ~ rhs, r1 = rhs - A z1 (:845-873)
    if (n_pre_smooth > 0) {
      if (pre_cgne && (fine_type == QMG_MATVEC_ORIGINAL || fine_type == QMG_MATVEC_RIGHT_JACOBI)) {
        complex<double>* z1_prec = fine_storage->check_out();
        zero_vector(z1_prec, fine_size);
        qmg::zero_guess_flag() = true;   // the iterate was zeroed just above: r0 = rhs, no A*0 (krylov.hpp)
        invif = minv_vector_minres(z1_prec, rhs, (int)fine_size_solve, n_pre_smooth, pre_smooth_tol, 0.85,
                                   Stencil2D::get_apply_function(fine_type == QMG_MATVEC_ORIGINAL ? QMG_MATVEC_M_MDAGGER : QMG_MATVEC_RBJ_M_MDAGGER), (void*)fine_stencil);
        fine_stencil->apply_M(z1, z1_prec, fine_type == QMG_MATVEC_ORIGINAL ? QMG_MATVEC_DAGGER : QMG_MATVEC_RBJ_DAGGER);
        mg->add_tracker_count(QMG_DSLASH_TYPE_PRESMOOTH, 2 * invif.ops_count + 1, level);
        fine_storage->check_in(z1_prec);
      } else {
        qmg::zero_guess_flag() = true;   // the iterate was zeroed just above: r0 = rhs, no A*0 (krylov.hpp)
        invif = minv_vector_minres(z1, rhs, (int)fine_size_solve, n_pre_smooth, pre_smooth_tol, 0.85, apply_fine_M, (void*)fine_stencil);
        mg->add_tracker_count(QMG_DSLASH_TYPE_PRESMOOTH, invif.ops_count, level);
      }
      apply_fine_M(Atmp, z1, (void*)fine_stencil);   // = zero_vector + apply_M(.., fine_type), one launch for ORIGINAL
      mg->add_tracker_count(QMG_DSLASH_TYPE_PRESMOOTH, 1, level);
      caxpbyz(1.0, rhs, -1.0, Atmp, r1, fine_size_solve);
    } else {
      zero_vector(Atmp, fine_size_solve);
      copy_vector(r1, rhs, fine_size_solve);
      copy_vector(z1, rhs, fine_size_solve);
    }
    // (Schur: the odd half of r1 must not leak stale pool data into the restriction)
    if (fine_type == QMG_MATVEC_RIGHT_SCHUR) zero_vector(r1 + fine_size_solve, fine_size - fine_size_solve);

    // ---- 2. restrict, prepare, coarse solve (recursion = the "K"), reconstruct (:875-1002)
    complex<double>* r_coarse = coarse_storage->check_out();
    zero_vector(r_coarse, coarse_size);
    transfer->restrict_f2c(r1, r_coarse);
    fine_storage->check_in(r1);
    const double rnorm = std::sqrt(norm2sq(r_coarse, coarse_size));
    complex<double>* r_coarse_prep = coarse_storage->check_out();
    zero_vector(r_coarse_prep, coarse_size);
    coarse_stencil->prepare_M(r_coarse_prep, r_coarse, coarse_type);
    const double rnorm_prep = std::sqrt(norm2sq(r_coarse_prep, coarse_size));
    complex<double>* e_coarse = coarse_storage->check_out();
    zero_vector(e_coarse, coarse_size);
    const double inner_tol = (rnorm_prep > 0.0) ? coarse_tol * rnorm / rnorm_prep : coarse_tol;
    qmg::zero_guess_flag() = true;   // e_coarse was zeroed just above; consumed by whichever solver runs next
    if (level == total_num_levels - 2) {
      const bool coarsest_normal = (coarse_type == QMG_MATVEC_M_MDAGGER || coarse_type == QMG_MATVEC_MDAGGER_M ||
                                    coarse_type == QMG_MATVEC_RBJ_M_MDAGGER || coarse_type == QMG_MATVEC_RBJ_MDAGGER_M);
      ShiftedFunctionStruct shift_struct;
      matrix_op_cplx op = apply_coarse_M;
      void* opdata = (void*)coarse_stencil;
      if (coarsest_normal && mg->get_coarsest_solve()->normal_shift != 0.0) {
        shift_struct.function = apply_coarse_M; shift_struct.extra_data = (void*)coarse_stencil;
        shift_struct.extra_shift = mg->get_coarsest_solve()->normal_shift; shift_struct.length = (int)coarse_size_solve;
        op = shift_function; opdata = (void*)&shift_struct;
      }
      if (coarse_restart == -1) {
        if (!coarsest_normal) invif = minv_vector_gcr(e_coarse, r_coarse_prep, (int)coarse_size_solve, coarse_max_iter, inner_tol, op, opdata, &verb2);
        else invif = minv_vector_cg(e_coarse, r_coarse_prep, (int)coarse_size_solve, coarse_max_iter, inner_tol, op, opdata, &verb2);
      } else {
        if (!coarsest_normal) invif = minv_vector_gcr_restart(e_coarse, r_coarse_prep, (int)coarse_size_solve, coarse_max_iter, inner_tol, coarse_restart, op, opdata, &verb2);
        else invif = minv_vector_cg_restart(e_coarse, r_coarse_prep, (int)coarse_size_solve, coarse_max_iter, inner_tol, coarse_restart, op, opdata, &verb2);
      }
    } else {
      mg->go_coarser();
      if (coarse_restart == -1)
        invif = minv_vector_gcr_var_precond(e_coarse, r_coarse_prep, (int)coarse_size_solve, coarse_max_iter, inner_tol, apply_coarse_M, (void*)coarse_stencil,
                                            mg_preconditioner, (void*)mg, &verb2);
      else
        invif = minv_vector_gcr_var_precond_restart(e_coarse, r_coarse_prep, (int)coarse_size_solve, coarse_max_iter, inner_tol, coarse_restart, apply_coarse_M,
                                                    (void*)coarse_stencil, mg_preconditioner, (void*)mg, &verb2);
      mg->go_finer();
    }
    mg->add_tracker_count(QMG_DSLASH_TYPE_KRYLOV, invif.ops_count, level + 1);
    mg->add_iterations_count(invif.iter, level + 1);
    coarse_storage->check_in(r_coarse_prep);
    complex<double>* e_coarse_reconstruct = coarse_storage->check_out();
    zero_vector(e_coarse_reconstruct, coarse_size);
    coarse_stencil->reconstruct_M(e_coarse_reconstruct, e_coarse, r_coarse, coarse_type);
    coarse_storage->check_in(r_coarse);
    coarse_storage->check_in(e_coarse);

    // ---- 3. prolong and correct (:1013-1021)
    complex<double>* z2 = fine_storage->check_out();
    zero_vector(z2, fine_size);
    transfer->prolong_c2f(e_coarse_reconstruct, z2);
    if (coarse_type == QMG_MATVEC_RIGHT_SCHUR) zero_vector(z2 + fine_size / 2, fine_size / 2);
    coarse_storage->check_in(e_coarse_reconstruct);
    cxpyz(z1, z2, lhs, fine_size_solve);
    fine_storage->check_in(z1);
    fine_storage->check_in(z2);

    // ---- 4. post-smooth on r2 = rhs - A lhs (:1023-1056)
    if (n_post_smooth > 0) {
      apply_fine_M(Atmp, lhs, (void*)fine_stencil);
      complex<double>* r2 = fine_storage->check_out();
      caxpbyz(1.0, rhs, -1.0, Atmp, r2, fine_size_solve);
      complex<double>* z3 = fine_storage->check_out();
      zero_vector(z3, fine_size);
      if (post_cgne && (fine_type == QMG_MATVEC_ORIGINAL || fine_type == QMG_MATVEC_RIGHT_JACOBI)) {
        complex<double>* z3_prec = fine_storage->check_out();
        zero_vector(z3_prec, fine_size);
        qmg::zero_guess_flag() = true;   // the iterate was zeroed just above: r0 = rhs, no A*0 (krylov.hpp)
        invif = minv_vector_minres(z3_prec, r2, (int)fine_size_solve, n_post_smooth, post_smooth_tol, 0.85,
                                   Stencil2D::get_apply_function(fine_type == QMG_MATVEC_ORIGINAL ? QMG_MATVEC_M_MDAGGER : QMG_MATVEC_RBJ_M_MDAGGER), (void*)fine_stencil);
        fine_stencil->apply_M(z3, z3_prec, fine_type == QMG_MATVEC_ORIGINAL ? QMG_MATVEC_DAGGER : QMG_MATVEC_RBJ_DAGGER);
        mg->add_tracker_count(QMG_DSLASH_TYPE_POSTSMOOTH, 2 * invif.ops_count + 1, level);
        fine_storage->check_in(z3_prec);
      } else {
        qmg::zero_guess_flag() = true;   // the iterate was zeroed just above: r0 = rhs, no A*0 (krylov.hpp)
        invif = minv_vector_minres(z3, r2, (int)fine_size_solve, n_post_smooth, post_smooth_tol, 0.85, apply_fine_M, (void*)fine_stencil);
        mg->add_tracker_count(QMG_DSLASH_TYPE_POSTSMOOTH, invif.ops_count, level);
      }
      cxpy(z3, lhs, fine_size_solve);
      fine_storage->check_in(r2);
      fine_storage->check_in(z3);
    }
    fine_storage->check_in(Atmp);
  }
};

#endif
