// slab.hpp -- y-slab domain decomposition of ONE lattice over the ranks of the communicator.
//
// The reference is single-process; its shift routines mark where a distributed version would exchange data
// (cshift/cshift_2d.h:39-42,72,89: "Becomes MPI").  Here rank r of R holds rows [r Ly/R, (r+1) Ly/R) of the global lattice
// as an ordinary Lattice2D of Ly/R rows, so every vector routine of the facade works on a slab unchanged; what changes is
//   * the operator: SlabWilson2D::apply_M sends the slab's first and last row of the right-hand side to the neighbouring
//     ranks (qmg_halo_exchange: RCCL send / recv on a second stream) WHILE the interior rows are applied, then applies the two
//     boundary rows from the received halos (qmg_stencil_apply_slab);
//   * the reductions: qmg_comm_set_distributed_reductions(1) makes norm2sq / dot / multidot ... of the library return the
//     sum over ranks, so the Krylov solvers of krylov.hpp run unchanged and take identical decisions on every rank.
// The operator is applied straight from the gauge links (qmg_wilson_apply_direct: the links are 32 B/site and replicated on
// every rank, a slab stores no stencil at all); QMG_WILSON_DIRECT=0 switches to the slab's rows of the stored stencil.
// One rank (no communicator): the exchange degenerates to device copies and the slab is the whole lattice.
// This round covers the fine Wilson operator (nc = 2) in fp64: one strong-scaled Krylov solve.  The K-cycle's coarse levels
// need the same halo step in the generic-nc kernels and a halo of the null vectors in the block-local Galerkin build.
#ifndef QMG_SLAB_HPP
#define QMG_SLAB_HPP

#include "qmg_device.hpp"
#include "lattice2d.hpp"

namespace qmg {

struct SlabGeometry {
  int Lx = 0, Ly_global = 0, world = 1, rank = 0, y0 = 0, Ly_local = 0;
  bool valid = false;
  // rows per rank must be even (the colouring of a slab is then the global one) and at least 2
  SlabGeometry(int x_len, int y_len, int world_, int rank_) : Lx(x_len), Ly_global(y_len), world(world_), rank(rank_) {
    if (world < 1 || rank < 0 || rank >= world || y_len % world) return;
    Ly_local = y_len / world;
    y0 = rank * Ly_local;
    valid = Ly_local >= 2 && !(Ly_local & 1) && !(x_len & 1) && x_len >= 2;
  }
};

// rows of a slab out of a vector over the global lattice (both on the device): two contiguous copies, one per parity
inline void slab_rows_of(complex<double>* slab, const complex<double>* global, const SlabGeometry& g, int nc) {
  const size_t row = (size_t)(g.Lx / 2) * nc;
  for (int q = 0; q < 2; q++)
    ok(qmg_memcpy_d2d(slab + (size_t)q * g.Ly_local * row, global + ((size_t)q * g.Ly_global + g.y0) * row, sizeof(complex<double>) * g.Ly_local * row,
                      current_stream()), "qmg_memcpy_d2d");
}

class SlabWilson2D {
 public:
  SlabGeometry geo;
  Lattice2D* lat = nullptr;          // the slab as a lattice: Lx x Ly_local, nc = 2
  complex<double>*clover = nullptr, *hopping = nullptr;
  complex<double>*halo_lo = nullptr, *halo_hi = nullptr;
  qmg_stencil_desc desc;
  void *comm_stream = nullptr, *ev_rhs = nullptr, *ev_halo = nullptr;
  bool overlap = true;               // false: exchange, then one launch over all rows
  bool from_links = true;            // apply straight from the (replicated, global) links: qmg_wilson_apply_direct; false: the slab's stored stencil
  const complex<double>* gauge = nullptr;
  double w = 1.0;
  long applies = 0;

  // gauge_global: the U(1) links of the WHOLE lattice on the device (32 B/site; every rank holds them)
  SlabWilson2D(const SlabGeometry& g, double mass, const complex<double>* gauge_global, double wilson_coeff = 1.0) : geo(g) {
    lat = new Lattice2D(g.Lx, g.Ly_local, 2);
    const size_t vol = (size_t)g.Lx * g.Ly_local;
    gauge = gauge_global; w = wilson_coeff;
    from_links = !(getenv("QMG_WILSON_DIRECT") && atoi(getenv("QMG_WILSON_DIRECT")) == 0);
    halo_lo = allocate_vector<complex<double>>(2 * (size_t)g.Lx);   // [parity][Lx/2][2]
    halo_hi = allocate_vector<complex<double>>(2 * (size_t)g.Lx);
    if (!from_links) {   // the slab's rows of the stored stencil (384 B/site) instead of the replicated links (the caller keeps gauge_global alive either way)
      clover = allocate_vector<complex<double>>(4 * vol);
      hopping = allocate_vector<complex<double>>(16 * vol);
      ok(qmg_wilson_fill_slab(clover, hopping, gauge_global, g.Lx, g.Ly_global, g.y0, g.Ly_local, wilson_coeff, current_stream()), "qmg_wilson_fill_slab");
    }
    desc.Lx = g.Lx; desc.Ly = g.Ly_local; desc.nc = 2;
    desc.clover = clover; desc.hopping = hopping;
    desc.shift[0] = mass; desc.shift[1] = 0.0;
    desc.eo_shift[0] = desc.eo_shift[1] = desc.dof_shift[0] = desc.dof_shift[1] = 0.0;
    ok(qmg_stream_create(&comm_stream), "qmg_stream_create");
    ok(qmg_event_create(&ev_rhs), "qmg_event_create");
    ok(qmg_event_create(&ev_halo), "qmg_event_create");
  }
  ~SlabWilson2D() {
    qmg_stream_sync(current_stream());
    qmg_stream_sync(comm_stream);
    if (clover) deallocate_vector(&clover);
    if (hopping) deallocate_vector(&hopping);
    deallocate_vector(&halo_lo); deallocate_vector(&halo_hi);
    qmg_event_destroy(ev_rhs); qmg_event_destroy(ev_halo); qmg_stream_destroy(comm_stream);
    delete lat;
  }
  size_t size_cv() const { return (size_t)lat->get_size_cv(); }

  // one launch over the rows `rows` selects (0 all, 1 interior, 2 boundary)
  void launch_rows(complex<double>* lhs, complex<double>* rhs, unsigned pieces, int rows, void* st) {
    const size_t hs = 2 * (size_t)geo.Lx;
    if (from_links)
      ok(qmg_wilson_apply_direct(QMG_C64, &desc, gauge, geo.Ly_global, geo.y0, w, lhs, rhs, halo_lo, halo_hi, pieces, 1, 0, hs, 1u, rows, st), "qmg_wilson_apply_direct");
    else
      ok(qmg_stencil_apply_slab(QMG_C64, &desc, lhs, rhs, halo_lo, halo_hi, pieces, 1, 0, hs, 1u, rows, st), "qmg_stencil_apply_slab");
  }
  // lhs = pieces(M) rhs on the slab; rhs must not alias lhs
  void apply(complex<double>* lhs, complex<double>* rhs, unsigned pieces) {
    void* st = current_stream();
    const size_t hs = 2 * (size_t)geo.Lx;
    applies++;
    if (!overlap) {
      ok(qmg_halo_exchange(QMG_C64, rhs, geo.Lx, geo.Ly_local, 2, halo_lo, halo_hi, 1, 0, hs, st), "qmg_halo_exchange");
      launch_rows(lhs, rhs, pieces, 0, st);
      return;
    }
    // the exchange on its own stream, behind the producer of rhs; the interior rows meanwhile; the boundary rows after both
    ok(qmg_event_record(ev_rhs, st), "qmg_event_record");
    ok(qmg_stream_wait_event(comm_stream, ev_rhs), "qmg_stream_wait_event");
    ok(qmg_halo_exchange(QMG_C64, rhs, geo.Lx, geo.Ly_local, 2, halo_lo, halo_hi, 1, 0, hs, comm_stream), "qmg_halo_exchange");
    ok(qmg_event_record(ev_halo, comm_stream), "qmg_event_record");
    launch_rows(lhs, rhs, pieces, 1, st);
    ok(qmg_stream_wait_event(st, ev_halo), "qmg_stream_wait_event");
    launch_rows(lhs, rhs, pieces, 2, st);
    // the next exchange overwrites the halos: it is ordered behind this boundary launch through ev_rhs of the next apply
  }
  void apply_M(complex<double>* lhs, complex<double>* rhs) { apply(lhs, rhs, QMG_P_ALL | QMG_P_ZERO); }
};

// krylov.hpp's matrix_op_cplx signature (the reference's apply_stencil_2D_M, stencil_2d.h:1897)
inline void apply_slab_wilson_M(complex<double>* lhs, complex<double>* rhs, void* extra) { static_cast<SlabWilson2D*>(extra)->apply_M(lhs, rhs); }

}  // namespace qmg
#endif
