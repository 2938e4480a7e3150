// qmg_setup.hip -- multigrid SETUP as block-local kernels.
//
// The reference phrases both setup steps as full-lattice passes:
//   * block_orthonormalize (transfer/transfer.h:514-607): modified Gram-Schmidt per block written as nvec(nvec+1)/2
//     single-vector restrict -> prolong -> axpy triples over the WHOLE fine lattice, and the whole thing twice (:160-174);
//   * CoarseOperator2D (operators/coarse.h:140-444): 9 probes per coarse colour, each a unit vector -> prolong -> partial
//     fine apply -> restrict over the whole fine lattice.
// The first round restated those passes as kernels (kept in qmg_transfer.hip as the fallback).  Both are block-LOCAL:
//   * a coarse site's block of the null vectors is an nvec x nel tile (nel = bx by nc_f fine elements) that fits LDS;
//     Gram-Schmidt on it touches nothing else.  k_block_ortho: one thread group per block, the tile read ONCE, both passes
//     (and the Cholesky factor of the first) done in LDS, written ONCE: 2 nvec size_cv_f complex of traffic instead of
//     ~3 nvec^2 size_cv_f per pass.
//   * the Galerkin operator is  C^(X) = sum_{x,y in B_X} R(x)^dag A(x,y) P(y),  H^_mu(X) = sum_{x in B_X, x+mu not in B_X}
//     R(x)^dag H_mu(x) P(x+mu): per fine site and piece two small products, T = M_piece(x) . P(nb) (nc_f x nc_c) and
//     acc += R(x)^dag . T (nc_c x nc_c).  k_galerkin: one workgroup per coarse site walks its 16 fine sites x 5 pieces and
//     writes the five nc_c x nc_c blocks once: ONE pass over the fine stencil and ~1.5 over the null vectors instead of
//     9 nc_c prolong/apply/restrict triples.
// Same arithmetic as the reference up to summation order (parity 1e-12 against the oracle's restatement of the passes).
#include <string.h>

#include "qmg_common.h"

namespace qmg {

int g_setup_fused = 1;   // tuning knob "setup_fused": 1 = the block-local kernels below, 0 = the first round's full-lattice passes

// fallbacks (qmg_transfer.hip)
int block_orthonormalize_passes(void* nullvecs, int nvec, int fLx, int fLy, int fnc, int cLx, int cLy, void* cholesky, void* stream);
int coarse_build_probes(void* cclover, void* chopping, const qmg_stencil_desc* fine, const void* nullvecs, const void* restrict_vecs, int cLx, int cLy, int cnc,
                        void* stream);

struct SetupGeom {
  int fhr, fLy, fnc;
  int chr, cLy, cnc;
  int bx, by;
  long fhalf_vol, fsize;
};
__device__ __forceinline__ long sg_coarse_index(const SetupGeom& g, int cx, int cy) {
  const int p = (cx + cy) & 1;
  return (long)(cy + p * g.cLy) * g.chr + (cx >> 1);
}
// fine element e (0 <= e < nel, row-major over the block's 2*by half-row runs of G elements) of block (cx, cy)
__device__ __forceinline__ long sg_block_elem(const SetupGeom& g, int cx, int cy, int e, int G) {
  const int rr = e / G, el = e - rr * G;
  const int p = rr / g.by, y = cy * g.by + (rr - p * g.by);
  return ((long)p * g.fhalf_vol + (long)y * g.fhr + (long)cx * (g.bx / 2)) * g.fnc + el;
}

// ---------------------------------------------------------------------------------------------------------------------
// Block orthonormalisation.  TPB threads (32, 64, 128 or 256: the power of two >= nel) own one block; thread t owns the
// tile columns e = t, t + TPB, ... of EVERY vector, so between the reductions no thread reads what another one wrote and
// the only synchronisation is the reduction itself (shuffle butterfly inside a wavefront, plus an LDS hop and a barrier
// when the block spans several wavefronts).  Order of operations = the reference's (modified Gram-Schmidt, :548-566):
// for i: for j < i: c = <v_j, v_i> with the CURRENT v_i; chol[j][i] = c; v_i -= c v_j;  then n = <v_i, v_i>,
// chol[i][i] = sqrt(n), v_i /= sqrt(n).  `passes` = 2 runs it twice (transfer.h:160-174), factor saved in the first.
template <int TPB>
__global__ __launch_bounds__(BLOCK) void k_block_ortho(cplx* __restrict__ nullv, int nvec, cplx* __restrict__ chol, const SetupGeom g, int passes, int groups_per_wg) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int G = (g.bx / 2) * g.fnc;
  const int nel = 2 * g.by * G;
  const int wg_threads = TPB * groups_per_wg;
  const int grp = threadIdx.x / TPB, t = threadIdx.x - grp * TPB;
  cplx* tile = reinterpret_cast<cplx*>(smem_raw) + (size_t)grp * nvec * nel;                       // [nvec][nel]
  double* red = reinterpret_cast<double*>(reinterpret_cast<cplx*>(smem_raw) + (size_t)groups_per_wg * nvec * nel);   // [wavefronts of the workgroup][2]
  const int cLx = 2 * g.chr;
  const long ncs = (long)cLx * g.cLy;
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  constexpr int WPG = (TPB > WAVE) ? TPB / WAVE : 1;   // wavefronts per group

  // sum over the TPB threads of a group; every thread of the group gets the result.  All threads of the workgroup call it
  // the same number of times (the loops below have uniform trip counts), so the barriers line up.
  auto group_sum2 = [&](double& a, double& b) {
#pragma unroll
    for (int m = 1; m < ((TPB < WAVE) ? TPB : WAVE); m <<= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
    if (WPG > 1) {
      __syncthreads();   // red[] free again
      if (lane == 0) { red[2 * wv] = a; red[2 * wv + 1] = b; }
      __syncthreads();
      a = 0.0; b = 0.0;
      for (int w = 0; w < WPG; w++) { a += red[2 * (grp * WPG + w)]; b += red[2 * (grp * WPG + w) + 1]; }
    }
  };

  for (long cs0 = (long)blockIdx.x * groups_per_wg; cs0 < ncs; cs0 += (long)gridDim.x * groups_per_wg) {
    const long cs = cs0 + grp;
    const bool live = cs < ncs;
    const int cy = live ? (int)(cs / cLx) : 0, cx = live ? (int)(cs - (long)cy * cLx) : 0;
    const long ci = sg_coarse_index(g, cx, cy);
    if (live)
      for (int d = 0; d < nvec; d++)
        for (int e = t; e < nel; e += TPB) tile[d * nel + e] = nullv[(long)d * g.fsize + sg_block_elem(g, cx, cy, e, G)];
    for (int pass = 0; pass < passes; pass++) {
      cplx* ch = (pass == 0 && chol && live) ? chol + ci * nvec * nvec : nullptr;
      for (int i = 0; i < nvec; i++) {
        cplx* vi = tile + i * nel;
        for (int j = 0; j < i; j++) {
          const cplx* vj = tile + j * nel;
          double cr = 0.0, cim = 0.0;
          if (live)
            for (int e = t; e < nel; e += TPB) {   // conj(v_j) v_i
              const cplx a = vj[e], b = vi[e];
              cr = fma(a.x, b.x, cr); cr = fma(a.y, b.y, cr);
              cim = fma(a.x, b.y, cim); cim = fma(-a.y, b.x, cim);
            }
          group_sum2(cr, cim);
          if (ch && t == 0) ch[j * nvec + i] = cmake(cr, cim);
          if (live)
            for (int e = t; e < nel; e += TPB) {   // v_i -= c v_j
              const cplx a = vj[e];
              cplx b = vi[e];
              b.x = fma(-cr, a.x, b.x); b.x = fma(cim, a.y, b.x);
              b.y = fma(-cr, a.y, b.y); b.y = fma(-cim, a.x, b.y);
              vi[e] = b;
            }
        }
        double n2 = 0.0, dummy = 0.0;
        if (live)
          for (int e = t; e < nel; e += TPB) { const cplx b = vi[e]; n2 = fma(b.x, b.x, n2); n2 = fma(b.y, b.y, n2); }
        group_sum2(n2, dummy);
        const double inv = 1.0 / sqrt(n2);
        if (ch && t == 0) ch[i * (nvec + 1)] = cmake(1.0 / inv, 0.0);   // = 1 / (1/sqrt(n2)) as the reference stores it (:588-593)
        if (live)
          for (int e = t; e < nel; e += TPB) { cplx b = vi[e]; b.x *= inv; b.y *= inv; vi[e] = b; }
      }
    }
    if (live)
      for (int d = 0; d < nvec; d++)
        for (int e = t; e < nel; e += TPB) nullv[(long)d * g.fsize + sg_block_elem(g, cx, cy, e, G)] = tile[d * nel + e];
    if (wg_threads > WAVE) __syncthreads();   // the next block's loads must not overtake a slower group's last reads of red[]
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Galerkin coarse operator: one workgroup per coarse site X.  For each fine site x of the block and each piece p (clover,
// +x, +y, -x, -y) with neighbour nb:  T = M_p(x) . P(nb)  (nc_f x nc_c, through LDS), then  acc += R(x)^dag . T  (nc_c x
// nc_c, in registers: thread k owns the output entries k, k + 256, ...).  A hop whose neighbour lies in the same block
// goes to the coarse clover (coarse.h:222-224, 250-252), one that leaves it to the coarse hopping of that direction,
// stored at the OUTPUT site X as in the fine layout.  The identity shift is NOT part of the build (coarse.h:131).
constexpr int GAL_MAXOUT = 4;   // output entries per thread: nc_c^2 <= 4 * 256, i.e. nc_c <= 32
// y-slab of a larger lattice: the prolongator's rows -1 / fLy (needed by the -y / +y hops that leave the slab) come from
// P_lo / P_hi ([null vector][parity][fhr][nf], halo_stride elements between vectors; qmg_halo_exchange of the null vectors).
__global__ __launch_bounds__(BLOCK) void k_galerkin(cplx* __restrict__ cclover, cplx* __restrict__ chopping, const cplx* __restrict__ fclover,
                                                    const cplx* __restrict__ fhopping, const cplx* __restrict__ P, const cplx* __restrict__ R, const SetupGeom g,
                                                    const cplx* __restrict__ P_lo, const cplx* __restrict__ P_hi, long halo_stride) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int nf = g.fnc, nc = g.cnc;
  cplx* Vx = reinterpret_cast<cplx*>(smem_raw);   // [nf][nc]   R(x)[r][a]
  cplx* Vn = Vx + nf * nc;                        // [nf][nc]   P(nb)[c][b]
  cplx* Mp = Vn + nf * nc;                        // [nf][nf]
  cplx* T = Mp + nf * nf;                         // [nf][nc]
  const int cLx = 2 * g.chr, fLx = 2 * g.fhr;
  const long fvol_cm = 2 * g.fhalf_vol * (long)nf * nf;   // elements per fine matrix field
  const long ccm = (long)cLx * g.cLy * nc * nc;
  const int tid = threadIdx.x;
  for (long cs = blockIdx.x; cs < (long)cLx * g.cLy; cs += gridDim.x) {
    const int cy = (int)(cs / cLx), cx = (int)(cs - (long)cy * cLx);
    const long ci = sg_coarse_index(g, cx, cy);
    cplx acc[5][GAL_MAXOUT];   // [0] clover, [1 + dir] hopping
#pragma unroll
    for (int p = 0; p < 5; p++)
#pragma unroll
      for (int k = 0; k < GAL_MAXOUT; k++) acc[p][k] = cmake(0.0, 0.0);
    for (int ly = 0; ly < g.by; ly++)
      for (int lx = 0; lx < g.bx; lx++) {
        const int x = cx * g.bx + lx, y = cy * g.by + ly;
        const int par = (x + y) & 1;
        const long site = (long)(y + par * g.fLy) * g.fhr + (x >> 1);
        __syncthreads();   // previous site's accumulation has read Vx
        for (int k = tid; k < nf * nc; k += BLOCK) { const int a = k / nf, r = k - a * nf; Vx[r * nc + a] = R[(long)a * g.fsize + site * nf + r]; }
#pragma unroll
        for (int p = 0; p < 5; p++) {   // 0: clover (nb = x); 1..4: +x, +y, -x, -y
          if ((p == 0 && !fclover) || (p > 0 && !fhopping)) continue;
          int nx = x, ny = y;
          if (p == 1) nx = (x + 1 == fLx) ? 0 : x + 1;
          if (p == 2) ny = (y + 1 == g.fLy) ? 0 : y + 1;
          if (p == 3) nx = (x == 0) ? fLx - 1 : x - 1;
          if (p == 4) ny = (y == 0) ? g.fLy - 1 : y - 1;
          // a hop that leaves the slab reads the prolongator from the halo rows (and is a coarse HOP, never clover)
          const cplx* Pn = P;
          long pstride = g.fsize;
          bool off_slab = false;
          if (p == 2 && P_hi && y + 1 == g.fLy) { Pn = P_hi; off_slab = true; }
          if (p == 4 && P_lo && y == 0) { Pn = P_lo; off_slab = true; }
          const bool inside = !off_slab && (nx / g.bx == cx) && (ny / g.by == cy);
          const int npar = (nx + (off_slab ? (p == 2 ? y + 1 : y - 1) : ny)) & 1;   // colour of the neighbour on the GLOBAL lattice
          long nsite = (long)(ny + npar * g.fLy) * g.fhr + (nx >> 1);
          if (off_slab) { nsite = (long)npar * g.fhr + (nx >> 1); pstride = halo_stride; }
          const cplx* M = (p == 0) ? fclover + site * nf * nf : fhopping + (long)(p - 1) * fvol_cm + site * nf * nf;
          __syncthreads();   // previous piece's products have read Vn, Mp, T
          for (int k = tid; k < nf * nc; k += BLOCK) { const int b = k / nf, c = k - b * nf; Vn[c * nc + b] = Pn[(long)b * pstride + nsite * nf + c]; }
          for (int k = tid; k < nf * nf; k += BLOCK) Mp[k] = M[k];
          __syncthreads();
          for (int k = tid; k < nf * nc; k += BLOCK) {   // T[r][b] = sum_c M[r][c] P(nb)[c][b]
            const int r = k / nc, b = k - r * nc;
            cplx s = cmake(0.0, 0.0);
            for (int c = 0; c < nf; c++) cmac(s, Mp[r * nf + c], Vn[c * nc + b]);
            T[k] = s;
          }
          __syncthreads();
#pragma unroll
          for (int q = 0; q < GAL_MAXOUT; q++) {   // acc[a][b] += sum_r conj(R(x)[r][a]) T[r][b]
            const int k = tid + q * BLOCK;
            if (k < nc * nc) {
              const int a = k / nc, b = k - a * nc;
              cplx s = cmake(0.0, 0.0);
              for (int r = 0; r < nf; r++) cmac_conj(s, Vx[r * nc + a], T[r * nc + b]);
              if (p == 0 || inside) acc[0][q] = cadd(acc[0][q], s);
              else acc[p][q] = cadd(acc[p][q], s);
            }
          }
        }
      }
#pragma unroll
    for (int q = 0; q < GAL_MAXOUT; q++) {
      const int k = tid + q * BLOCK;
      if (k < nc * nc) {
        cclover[ci * nc * nc + k] = acc[0][q];
#pragma unroll
        for (int d = 0; d < 4; d++) chopping[(long)d * ccm + ci * nc * nc + k] = acc[1 + d][q];
      }
    }
  }
}

static int make_sgeom(SetupGeom* g, int fLx, int fLy, int fnc, int cLx, int cLy, int cnc) {
  if (!valid_lattice(fLx, fLy) || !valid_lattice(cLx, cLy) || fnc < 1 || cnc < 1) return QMG_ERR_INVALID;
  if (fLx % cLx || fLy % cLy) return QMG_ERR_INVALID;
  g->fhr = fLx / 2; g->fLy = fLy; g->fnc = fnc;
  g->chr = cLx / 2; g->cLy = cLy; g->cnc = cnc;
  g->bx = fLx / cLx; g->by = fLy / cLy;
  g->fhalf_vol = (long)g->fhr * fLy;
  g->fsize = 2 * g->fhalf_vol * fnc;
  return QMG_SUCCESS;
}

}  // namespace qmg

using namespace qmg;

extern "C" {

// block_orthonormalize (transfer.h:514-607), `passes` in {1, 2} passes in ONE launch (the TransferMG constructor runs two,
// the decomposition saved in the first: :160-174).  Falls back to the full-lattice passes for odd block widths or tiles
// beyond LDS.  No allocation and no synchronisation here: asynchronous on `stream`.
int qmg_block_orthonormalize_n(void* nullvecs, int nvec, int fLx, int fLy, int fnc, int cLx, int cLy, void* cholesky, int passes, void* stream) {
  if (!nullvecs || nvec < 1 || passes < 1 || passes > 2) return QMG_ERR_INVALID;
  SetupGeom g;
  int rc = make_sgeom(&g, fLx, fLy, fnc, cLx, cLy, nvec);
  if (rc) return rc;
  const int nel = g.bx * g.by * fnc;
  int TPB = 32;
  while (TPB < nel && TPB < BLOCK) TPB <<= 1;
  int groups = (TPB >= 64) ? 1 : 2;   // a workgroup is at least one wavefront
  const size_t tile_bytes = sizeof(cplx) * (size_t)nvec * nel;
  size_t smem = tile_bytes * groups + sizeof(double) * 2 * (BLOCK / WAVE);
  if (!g_setup_fused || (g.bx & 1) || smem > 150 * 1024) {
    for (int p = 0; p < passes && !rc; p++) rc = block_orthonormalize_passes(nullvecs, nvec, fLx, fLy, fnc, cLx, cLy, p == 0 ? cholesky : nullptr, stream);
    return rc;
  }
  while (groups * 2 * TPB <= BLOCK && tile_bytes * groups * 2 <= 24 * 1024) groups *= 2;   // small tiles: several blocks per workgroup
  smem = tile_bytes * groups + sizeof(double) * 2 * (BLOCK / WAVE);
  const long ncs = (long)cLx * cLy;
  long nwg = (ncs + groups - 1) / groups;
  if (nwg > 262144) nwg = 262144;
  hipStream_t st = as_stream(stream);
#define QMG_BO(TPBV)                                                                                                                        \
  {                                                                                                                                         \
    if (smem > 64 * 1024) QMG_HIP_CHECK(hipFuncSetAttribute((const void*)k_block_ortho<TPBV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
    k_block_ortho<TPBV><<<(unsigned)nwg, TPBV * groups, smem, st>>>((cplx*)nullvecs, nvec, (cplx*)cholesky, g, passes, groups);             \
  }
  if (TPB == 32) QMG_BO(32) else if (TPB == 64) QMG_BO(64) else if (TPB == 128) QMG_BO(128) else QMG_BO(256)
#undef QMG_BO
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}
int qmg_block_orthonormalize(void* nullvecs, int nvec, int fLx, int fLy, int fnc, int cLx, int cLy, void* cholesky, void* stream) {
  return qmg_block_orthonormalize_n(nullvecs, nvec, fLx, fLy, fnc, cLx, cLy, cholesky, 1, stream);
}

// CoarseOperator2D constructor, steps 1-2 (coarse.h:137-444) as ONE launch.  Falls back to the probe loop for coarse
// colour counts beyond 32 or fine blocks beyond LDS.  Asynchronous on `stream`, no allocation.
int qmg_coarse_build(void* cclover, void* chopping, const qmg_stencil_desc* fine, const void* nullvecs, const void* restrict_vecs, int cLx, int cLy, int cnc,
                     void* stream) {
  return qmg_coarse_build_slab(cclover, chopping, fine, nullvecs, restrict_vecs, cLx, cLy, cnc, nullptr, nullptr, 0, stream);
}

// The same on a y-slab: P_halo_lo / P_halo_hi = the null vectors' rows -1 / fLy from the neighbouring ranks (qmg_halo_exchange with
// nrhs = cnc, vec_stride = the fine size_cv), halo_stride elements between vectors; NULL = periodic in y (the whole lattice).
int qmg_coarse_build_slab(void* cclover, void* chopping, const qmg_stencil_desc* fine, const void* nullvecs, const void* restrict_vecs, int cLx, int cLy, int cnc,
                          const void* P_halo_lo, const void* P_halo_hi, size_t halo_stride, void* stream) {
  if (!cclover || !chopping || !fine || !nullvecs) return QMG_ERR_INVALID;
  if ((P_halo_lo == nullptr) != (P_halo_hi == nullptr)) return QMG_ERR_INVALID;
  SetupGeom g;
  int rc = make_sgeom(&g, fine->Lx, fine->Ly, fine->nc, cLx, cLy, cnc);
  if (rc) return rc;
  const int nf = fine->nc;
  const size_t smem = sizeof(cplx) * ((size_t)3 * nf * cnc + (size_t)nf * nf);
  if (!g_setup_fused || cnc * cnc > GAL_MAXOUT * BLOCK || smem > 150 * 1024) {
    if (P_halo_lo) return QMG_ERR_UNSUPPORTED;   // the full-lattice probe passes have no halo step
    return coarse_build_probes(cclover, chopping, fine, nullvecs, restrict_vecs, cLx, cLy, cnc, stream);
  }
  long nwg = (long)cLx * cLy;
  if (nwg > 262144) nwg = 262144;
  if (smem > 64 * 1024) QMG_HIP_CHECK(hipFuncSetAttribute((const void*)k_galerkin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  k_galerkin<<<(unsigned)nwg, BLOCK, smem, as_stream(stream)>>>((cplx*)cclover, (cplx*)chopping, (const cplx*)fine->clover, (const cplx*)fine->hopping,
                                                               (const cplx*)nullvecs, (const cplx*)(restrict_vecs ? restrict_vecs : nullvecs), g,
                                                               (const cplx*)P_halo_lo, (const cplx*)P_halo_hi, (long)halo_stride);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

}  // extern "C"
