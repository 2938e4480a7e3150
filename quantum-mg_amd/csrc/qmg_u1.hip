// qmg_u1.hip -- U(1) gauge-field generation and observables on the device (reference: u1/u1_utils.h).
//
// The reference generates its quenched U(1) fields with a SEQUENTIAL non-compact heatbath on the host
// (heatbath_noncompact_update, u1_utils.h:607-757; its own comment: "This algorithm can't be parallelized as is... We
// would need subsets").  The subsets exist: with the non-compact action S = beta/2 sum_p theta_p^2, theta_p = A_x(x) +
// A_y(x+xhat) - A_x(x+yhat) - A_y(x), the conditional distribution of a link given the rest is Gaussian,
//     A_mu(x) ~ N(-staple/2, 1/(2 beta)),
// and the staple of an x-link at (x, y) (u1_utils.h:644-650) contains x-links only from rows y+1 and y-1, the staple of a
// y-link at (x, y) (:658-664) y-links only from columns x+1 and x-1.  So all x-links of the EVEN rows are conditionally
// independent given everything else, likewise the odd rows, and y-links by even / odd columns: one sweep is four
// launches, each a perfectly parallel exact heatbath step of a quarter of the links.  Same stationary distribution as
// the sequential sweep (each step samples a conditional of the same Gibbs measure), different update order and random
// stream -- configurations are not reproduced link by link, the ENSEMBLE is (tests pin plaquette and m_pi).
//
// Layout: phase field = two real nc=1 lattice fields (mu = 0, 1), phase[mu*V + site], site = even-odd index
// (lattice.h:75-81), as `gauge_coord_to_index` gives it; compact links U = exp(i A) in the same order (complex).
#include <string.h>

#include "qmg_common.h"

namespace qmg {

__device__ __forceinline__ long eo_index(int x, int y, int Lx, int Ly) {
  const int p = (x + y) & 1;
  return (long)(y + p * Ly) * (Lx >> 1) + (x >> 1);
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// one N(0,1) draw keyed by (seed, counter): counter-based, so the field does not depend on the launch geometry
__device__ __forceinline__ double gaussian_draw(unsigned long long seed, unsigned long long counter) {
  const unsigned long long h1 = mix64(seed * 0xD1342543DE82EF95ull + 2ull * counter);
  const unsigned long long h2 = mix64(h1 + 2ull * counter + 1ull);
  const double u1 = ((double)(h1 >> 11) + 1.0) * (1.0 / 9007199254740992.0);   // (0,1]
  const double u2 = (double)(h2 >> 11) * (1.0 / 9007199254740992.0);
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

// mu = 0: x-links of the rows with y & 1 == line ; mu = 1: y-links of the columns with x & 1 == line
__global__ __launch_bounds__(BLOCK) void k_heatbath_noncompact(double* __restrict__ phase, int Lx, int Ly, int mu, int line, double width,
                                                               unsigned long long seed, unsigned long long sweep) {
  const long V = (long)Lx * Ly;
  const long nlinks = V / 2;
  double* Ax = phase;
  double* Ay = phase + V;
  for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < nlinks; t += (long)gridDim.x * BLOCK) {
    int x, y;
    if (mu == 0) { x = (int)(t % Lx); y = 2 * (int)(t / Lx) + line; }
    else { y = (int)(t % Ly); x = 2 * (int)(t / Ly) + line; }
    const int xp = (x + 1 == Lx) ? 0 : x + 1, xm = (x == 0) ? Lx - 1 : x - 1;
    const int yp = (y + 1 == Ly) ? 0 : y + 1, ym = (y == 0) ? Ly - 1 : y - 1;
    double staple;
    if (mu == 0) {   // u1_utils.h:644-650
      staple = Ay[eo_index(xp, y, Lx, Ly)] - Ax[eo_index(x, yp, Lx, Ly)] - Ay[eo_index(x, y, Lx, Ly)]
             - Ay[eo_index(xp, ym, Lx, Ly)] - Ax[eo_index(x, ym, Lx, Ly)] + Ay[eo_index(x, ym, Lx, Ly)];
    } else {         // :658-664
      staple = Ax[eo_index(x, yp, Lx, Ly)] - Ay[eo_index(xp, y, Lx, Ly)] - Ax[eo_index(x, y, Lx, Ly)]
             - Ax[eo_index(xm, yp, Lx, Ly)] - Ay[eo_index(xm, y, Lx, Ly)] + Ax[eo_index(xm, y, Lx, Ly)];
    }
    const long site = eo_index(x, y, Lx, Ly);
    const double g = gaussian_draw(seed, (sweep * 2ull + (unsigned long long)mu) * (unsigned long long)V + (unsigned long long)site);
    (mu == 0 ? Ax : Ay)[site] = width * g - 0.5 * staple;
  }
}

// U = exp(i A) (polar_vector)
__global__ __launch_bounds__(BLOCK) void k_phase_to_gauge(cplx* __restrict__ gauge, const double* __restrict__ phase, long n) {
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) {
    double s, c;
    sincos(phase[i], &s, &c);
    gauge[i] = cmake(c, s);
  }
}
// A = arg U
__global__ __launch_bounds__(BLOCK) void k_gauge_to_phase(double* __restrict__ phase, const cplx* __restrict__ gauge, long n) {
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (long)gridDim.x * BLOCK) phase[i] = atan2(gauge[i].y, gauge[i].x);
}

// Per-block partial sums of: plaquette (re, im), topological charge density arg(P)/2pi, and -- from a phase field --
// the non-compact plaquette angle squared.  MODE 0: compact links; MODE 1: phases.
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_plaquette(const cplx* __restrict__ gauge, const double* __restrict__ phase, int Lx, int Ly, double* __restrict__ partials) {
  const long V = (long)Lx * Ly;
  double v[3] = {0.0, 0.0, 0.0};
  for (long t = (long)blockIdx.x * BLOCK + threadIdx.x; t < V; t += (long)gridDim.x * BLOCK) {
    const int x = (int)(t % Lx), y = (int)(t / Lx);
    const int xp = (x + 1 == Lx) ? 0 : x + 1, yp = (y + 1 == Ly) ? 0 : y + 1;
    const long s = eo_index(x, y, Lx, Ly), sx = eo_index(xp, y, Lx, Ly), sy = eo_index(x, yp, Lx, Ly);
    if (MODE == 0) {   // U_x(x) U_y(x+xhat) U_x^*(x+yhat) U_y^*(x)   (u1_utils.h:424-462)
      cplx p = cmul(gauge[s], gauge[V + sx]);
      p = cmul(p, cconj(gauge[sy]));
      p = cmul(p, cconj(gauge[V + s]));
      v[0] += p.x; v[1] += p.y;
      v[2] += atan2(p.y, p.x);   // get_topo_u1 (:465-508): sum arg / 2 pi
    } else {           // theta_p = A_x(x) + A_y(x+xhat) - A_x(x+yhat) - A_y(x)   (get_noncompact_action_u1, :386-421)
      const double th = phase[s] + phase[V + sx] - phase[sy] - phase[V + s];
      v[0] += th * th;
    }
  }
  __shared__ double sm[3][BLOCK / WAVE];
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const double w = wave_sum(v[q]);
    if (lane == 0) sm[q][wv] = w;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = 0.0;
    for (int w = 0; w < BLOCK / WAVE; w++) t += sm[threadIdx.x][w];
    partials[(long)blockIdx.x * 3 + threadIdx.x] = t;
  }
}
__global__ void k_sum3(const double* __restrict__ partials, int nparts, double* __restrict__ out) {
  if (threadIdx.x < 3) {
    double t = 0.0;
    for (int i = 0; i < nparts; i++) t += partials[(long)i * 3 + threadIdx.x];   // fixed order: deterministic
    out[threadIdx.x] = t;
  }
}

static int plaquette_sums(const void* gauge, const double* phase, int Lx, int Ly, double out[3], void* stream, int mode) {
  if (!valid_lattice(Lx, Ly)) return QMG_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  const long V = (long)Lx * Ly;
  long nb = (V + BLOCK - 1) / BLOCK;
  if (nb > 1024) nb = 1024;
  double* buf = nullptr;
  QMG_HIP_CHECK(hipMalloc((void**)&buf, sizeof(double) * (3 * nb + 3)));
  if (mode == 0) k_plaquette<0><<<(unsigned)nb, BLOCK, 0, st>>>((const cplx*)gauge, nullptr, Lx, Ly, buf);
  else k_plaquette<1><<<(unsigned)nb, BLOCK, 0, st>>>(nullptr, phase, Lx, Ly, buf);
  k_sum3<<<1, 64, 0, st>>>(buf, (int)nb, buf + 3 * nb);
  hipError_t e = hipMemcpyAsync(out, buf + 3 * nb, sizeof(double) * 3, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  hipFree(buf);
  if (e != hipSuccess) { set_hip_error(e, "plaquette_sums"); return QMG_ERR_HIP; }
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

}  // namespace qmg

using namespace qmg;

extern "C" {

// heatbath_noncompact_update (u1_utils.h:607-757) as a four-colour parallel heatbath: n_update sweeps of {x-links even rows,
// x-links odd rows, y-links even columns, y-links odd columns}.  phase: DEVICE double[2 Lx Ly].  `seed` + the running sweep
// index `first_sweep` key the counter-based generator (pass the number of sweeps already done to continue a stream).
int qmg_u1_heatbath_noncompact(double* phase, int Lx, int Ly, double beta, int n_update, unsigned long long seed, unsigned long long first_sweep, void* stream) {
  if (!phase || !valid_lattice(Lx, Ly) || !(beta > 0.0) || n_update < 0) return QMG_ERR_INVALID;
  const double width = sqrt(0.5 / beta);
  const long nlinks = (long)Lx * Ly / 2;
  const unsigned g = grid_1d((size_t)nlinks);
  hipStream_t st = as_stream(stream);
  for (int i = 0; i < n_update; i++) {
    const unsigned long long sweep = first_sweep + (unsigned long long)i;
    k_heatbath_noncompact<<<g, BLOCK, 0, st>>>(phase, Lx, Ly, 0, 0, width, seed, sweep);
    k_heatbath_noncompact<<<g, BLOCK, 0, st>>>(phase, Lx, Ly, 0, 1, width, seed, sweep);
    k_heatbath_noncompact<<<g, BLOCK, 0, st>>>(phase, Lx, Ly, 1, 0, width, seed, sweep);
    k_heatbath_noncompact<<<g, BLOCK, 0, st>>>(phase, Lx, Ly, 1, 1, width, seed, sweep);
  }
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

// polar_vector(phases, gauge_field, size_gauge): U = exp(i A); and its inverse A = arg U (the phases write_gauge_u1 stores)
int qmg_u1_phase_to_gauge(void* gauge, const double* phase, size_t n, void* stream) {
  if ((!gauge || !phase) && n) return QMG_ERR_INVALID;
  if (n == 0) return QMG_SUCCESS;
  k_phase_to_gauge<<<grid_1d(n), BLOCK, 0, as_stream(stream)>>>((cplx*)gauge, phase, (long)n);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}
int qmg_u1_gauge_to_phase(double* phase, const void* gauge, size_t n, void* stream) {
  if ((!gauge || !phase) && n) return QMG_ERR_INVALID;
  if (n == 0) return QMG_SUCCESS;
  k_gauge_to_phase<<<grid_1d(n), BLOCK, 0, as_stream(stream)>>>(phase, (const cplx*)gauge, (long)n);
  QMG_LAUNCH_CHECK();
  return QMG_SUCCESS;
}

// get_plaquette_u1 (u1_utils.h:424-462): volume average of the plaquette (re, im) -> out_host[0..1];
// get_topo_u1 (:465-508): sum_p arg(P) / 2 pi -> out_host[2].  Synchronous.
int qmg_u1_plaquette(const void* gauge, int Lx, int Ly, double* out_host, void* stream) {
  if (!gauge || !out_host) return QMG_ERR_INVALID;
  double s[3];
  const int rc = plaquette_sums(gauge, nullptr, Lx, Ly, s, stream, 0);
  if (rc) return rc;
  const double V = (double)Lx * Ly;
  out_host[0] = s[0] / V; out_host[1] = s[1] / V; out_host[2] = s[2] * 0.5 / 3.14159265358979323846;
  return QMG_SUCCESS;
}
// get_noncompact_action_u1 (:386-421): beta/2 sum_p theta_p^2
int qmg_u1_noncompact_action(const double* phase, int Lx, int Ly, double beta, double* out_host, void* stream) {
  if (!phase || !out_host) return QMG_ERR_INVALID;
  double s[3];
  const int rc = plaquette_sums(nullptr, phase, Lx, Ly, s, stream, 1);
  if (rc) return rc;
  *out_host = 0.5 * beta * s[0];
  return QMG_SUCCESS;
}

}  // extern "C"
