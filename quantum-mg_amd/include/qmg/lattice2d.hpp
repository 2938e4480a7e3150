// lattice2d.hpp -- Lattice2D: even-odd index algebra, same public surface as the reference's
// lattice/lattice.h:12-396 (method names, argument order, return conventions) so host code written
// against it compiles unchanged.  Sizes are kept in 64 bits internally (the reference's `int`
// size_hopping overflows for a 1024^2, nc=24 level, lattice.h:23,40); the getters the reference
// exposes as `int` stay `int` and a `_l` variant returns the full value.
#ifndef QMG_LATTICE2D_HPP
#define QMG_LATTICE2D_HPP

class Lattice2D {
 private:
  static const int nd = 2;
  int dims[2];
  int nc;
  long volume, size_cv, size_cm, size_gauge, size_hopping, size_corner;

  void resize() {
    volume = (long)dims[0] * dims[1];
    size_cv = volume * nc;
    size_cm = size_cv * nc;
    size_gauge = size_cm * nd;
    size_hopping = size_gauge * 2;
    size_corner = size_gauge * 2;
  }

 public:
  Lattice2D(int xlen, int ylen, int my_nc) : nc(my_nc) { dims[0] = xlen; dims[1] = ylen; resize(); }
  Lattice2D(const Lattice2D& o) : nc(o.nc) { dims[0] = o.dims[0]; dims[1] = o.dims[1]; resize(); }
  ~Lattice2D() {}

  void update_nc(int my_nc) { nc = my_nc; resize(); }

  // ---- coordinates -> indices (lattice.h:75-182) ----
  inline int coord_to_index(int x, int y) const {
    if (volume == 1) return 0;
    const int parity = (x + y) & 1;
    return (y + parity * dims[1]) * (dims[0] / 2) + (x / 2) % (dims[0] / 2);
  }
  inline int coord_to_index(int* c) const { return coord_to_index(c[0], c[1]); }
  inline int dof_coord_to_index(int total_dof, int x, int y, int dof) const { return total_dof * coord_to_index(x, y) + dof; }
  inline int dof_coord_to_index(int total_dof, int* c, int dof) const { return dof_coord_to_index(total_dof, c[0], c[1], dof); }
  inline int dof_coord_to_index(int total_dof, int i, int dof) const { return total_dof * i + dof; }
  inline int cv_coord_to_index(int x, int y, int c) const { return nc * coord_to_index(x, y) + c; }
  inline int cv_coord_to_index(int* co, int c) const { return cv_coord_to_index(co[0], co[1], c); }
  inline int cv_coord_to_index(int i, int c) const { return nc * i + c; }
  inline int cm_coord_to_index(int x, int y, int c1, int c2) const { return nc * cv_coord_to_index(x, y, c1) + c2; }
  inline int cm_coord_to_index(int* co, int c1, int c2) const { return cm_coord_to_index(co[0], co[1], c1, c2); }
  inline int cm_coord_to_index(int i, int c1, int c2) const { return nc * cv_coord_to_index(i, c1) + c2; }
  inline long gauge_coord_to_index(int x, int y, int c1, int c2, int mu) const { return mu * size_cm + cm_coord_to_index(x, y, c1, c2); }
  inline long gauge_coord_to_index(int* co, int c1, int c2, int mu) const { return gauge_coord_to_index(co[0], co[1], c1, c2, mu); }
  inline long gauge_coord_to_index(int i, int c1, int c2, int mu) const { return mu * size_cm + cm_coord_to_index(i, c1, c2); }
  inline long hopping_coord_to_index(int x, int y, int c1, int c2, int mu) const { return mu * size_cm + cm_coord_to_index(x, y, c1, c2); }
  inline long hopping_coord_to_index(int* co, int c1, int c2, int mu) const { return hopping_coord_to_index(co[0], co[1], c1, c2, mu); }
  inline long hopping_coord_to_index(int i, int c1, int c2, int mu) const { return mu * size_cm + cm_coord_to_index(i, c1, c2); }
  inline long corner_coord_to_index(int x, int y, int c1, int c2, int munu) const { return munu * size_cm + cm_coord_to_index(x, y, c1, c2); }
  inline long corner_coord_to_index(int* co, int c1, int c2, int munu) const { return corner_coord_to_index(co[0], co[1], c1, c2, munu); }
  inline long corner_coord_to_index(int i, int c1, int c2, int munu) const { return munu * size_cm + cm_coord_to_index(i, c1, c2); }
  inline int vol_index_dof_to_cv_index(int vol_index, int c) const { return nc * vol_index + c; }

  // ---- indices -> coordinates (lattice.h:199-282) ----
  inline void index_to_coord(int i, int& x, int& y) const {
    if (volume == 1) { x = y = 0; return; }
    const int parity = i / (int)(volume / 2);
    y = i / (dims[0] / 2) - parity * dims[1];
    x = 2 * (i % (dims[0] / 2)) + ((y & 1) + parity) % 2;
  }
  inline void index_to_coord(int i, int* c) const { index_to_coord(i, c[0], c[1]); }
  inline void dof_index_to_coord(int i, int total_dof, int& x, int& y, int& dof) const { index_to_coord(i / total_dof, x, y); dof = i % total_dof; }
  inline void dof_index_to_coord(int i, int total_dof, int* c, int& dof) const { dof_index_to_coord(i, total_dof, c[0], c[1], dof); }
  inline void cv_index_to_coord(int i, int& x, int& y, int& c) const { index_to_coord(i / nc, x, y); c = i % nc; }
  inline void cv_index_to_coord(int i, int* co, int& c) const { cv_index_to_coord(i, co[0], co[1], c); }
  inline void cm_index_to_coord(int i, int& x, int& y, int& c1, int& c2) const { cv_index_to_coord(i / nc, x, y, c1); c2 = i % nc; }
  inline void cm_index_to_coord(int i, int* co, int& c1, int& c2) const { cm_index_to_coord(i, co[0], co[1], c1, c2); }
  inline void gauge_index_to_coord(long i, int& x, int& y, int& c1, int& c2, int& mu) const { mu = (int)(i / size_cm); cm_index_to_coord((int)(i - mu * size_cm), x, y, c1, c2); }
  inline void hopping_index_to_coord(long i, int& x, int& y, int& c1, int& c2, int& mu) const { gauge_index_to_coord(i, x, y, c1, c2, mu); }
  inline void corner_index_to_coord(long i, int& x, int& y, int& c1, int& c2, int& munu) const { gauge_index_to_coord(i, x, y, c1, c2, munu); }

  // ---- parity queries.  NB the reference's *_is_even family returns `i > size/2`, i.e. true for
  // the ODD half (lattice.h:288-316); kept as is because callers may rely on it. ----
  inline bool index_is_even(int i) const { return i > (volume / 2); }
  inline bool cv_index_is_even(int i) const { return i > (size_cv / 2); }
  inline bool cm_index_is_even(int i) const { return i > (size_cm / 2); }
  inline bool gauge_index_is_even(long i) const { return i > (size_gauge / 2); }
  inline bool hopping_index_is_even(long i) const { return i > (size_hopping / 2); }
  inline bool corner_index_is_even(long i) const { return i > (size_corner / 2); }
  inline bool coord_is_even(int x, int y) const { return (x + y) % 2 == 0; }

  // ---- info ----
  inline void get_dim(int* d) const { d[0] = dims[0]; d[1] = dims[1]; }
  inline int get_dim_mu(int mu) const { return (mu >= 0 && mu < nd) ? dims[mu] : -1; }
  inline int get_nd() const { return nd; }
  inline int get_nc() const { return nc; }
  inline int get_nc_nc() const { return nc * nc; }
  inline int get_volume() const { return (int)volume; }
  inline int get_size_dof(int total_dof) const { return (int)(volume * total_dof); }
  inline int get_size_cv() const { return (int)size_cv; }
  inline int get_size_cm() const { return (int)size_cm; }
  inline int get_size_gauge() const { return (int)size_gauge; }
  inline int get_size_hopping() const { return (int)size_hopping; }
  inline int get_size_corner() const { return (int)size_corner; }
  // 64-bit variants (not in the reference)
  inline long get_size_cv_l() const { return size_cv; }
  inline long get_size_cm_l() const { return size_cm; }
  inline long get_size_hopping_l() const { return size_hopping; }
};

#endif
