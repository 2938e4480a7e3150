// Host-only checks of the facade's solver bookkeeping (no GPU call is made; links libqmg_hip.so for the symbols only).
//   1. qmg::gcr_direction_weights: GCR with RAW search directions + back-substitution == GCR with explicitly
//      orthogonalised directions (krylov.hpp), on a small dense host problem.
//   2. qmg::ZeroGuess / take_zero_guess: the hint is consumed exactly once and cleared on scope exit.
//   3. batch masks and views (batch.hpp).
#include <complex>
#include <cstdio>
#include <random>
#include <vector>

#include "../../quantum-mg_amd/include/qmg/qmg.hpp"

typedef std::complex<double> cd;
typedef std::vector<cd> vec;

static cd dotc(const vec& a, const vec& b) { cd s = 0; for (size_t i = 0; i < a.size(); i++) s += std::conj(a[i]) * b[i]; return s; }
static void axpy(cd a, const vec& x, vec& y) { for (size_t i = 0; i < x.size(); i++) y[i] += a * x[i]; }
static vec matvec(const std::vector<vec>& A, const vec& x) { vec y(x.size(), 0.0); for (size_t i = 0; i < x.size(); i++) for (size_t j = 0; j < x.size(); j++) y[i] += A[i][j] * x[j]; return y; }

int main() {
  int fails = 0;
  // ---- 1. raw-z GCR == explicit GCR
  const int n = 24, K = 9;
  std::mt19937 gen(7);
  std::normal_distribution<double> g(0.0, 1.0);
  std::vector<vec> A(n, vec(n));
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) A[i][j] = cd(g(gen), g(gen)) * 0.2 + (i == j ? cd(3.0, 0.5) : cd(0.0));
  vec b(n);
  for (int i = 0; i < n; i++) b[i] = cd(g(gen), g(gen));
  // explicit: z and w both orthogonalised, x updated every step
  vec x1(n, 0.0), r = b;
  std::vector<vec> Z, W;
  std::vector<double> W2;
  // raw: keep z raw, record c and alpha
  vec x2(n, 0.0);
  std::vector<vec> Zraw;
  std::vector<std::vector<cd>> C(K);
  std::vector<cd> alphas(K);
  for (int k = 0; k < K; k++) {
    vec z = r;                       // identity preconditioner
    Zraw.push_back(z);
    vec w = matvec(A, z);
    std::vector<cd> c(k);
    for (int i = 0; i < k; i++) c[i] = -dotc(W[i], w) / W2[i];
    for (int i = 0; i < k; i++) { axpy(c[i], W[i], w); axpy(c[i], Z[i], z); }
    C[k] = c;
    const double ww = dotc(w, w).real();
    const cd alpha = dotc(w, r) / ww;
    alphas[k] = alpha;
    axpy(alpha, z, x1);
    axpy(-alpha, w, r);
    Z.push_back(z); W.push_back(w); W2.push_back(ww);
  }
  const std::vector<cd> y = qmg::gcr_direction_weights(alphas, C, K);
  for (int k = 0; k < K; k++) axpy(y[k], Zraw[k], x2);
  double diff = 0, nrm = 0;
  for (int i = 0; i < n; i++) { diff += std::norm(x1[i] - x2[i]); nrm += std::norm(x1[i]); }
  if (!(diff <= 1e-26 * nrm)) { printf("FAIL gcr_direction_weights: rel diff %.3e\n", std::sqrt(diff / nrm)); fails++; }
  // and x really solves A x ~ b progressively
  vec res = matvec(A, x2);
  double rr = 0, bb = 0;
  for (int i = 0; i < n; i++) { rr += std::norm(b[i] - res[i]); bb += std::norm(b[i]); }
  if (!(rr < 1e-6 * bb)) { printf("FAIL gcr residual %.3e\n", std::sqrt(rr / bb)); fails++; }

  // ---- 2. zero-guess hint
  if (qmg::take_zero_guess()) { printf("FAIL hint set by default\n"); fails++; }
  { qmg::ZeroGuess zg; if (!qmg::take_zero_guess()) { printf("FAIL hint not delivered\n"); fails++; } if (qmg::take_zero_guess()) { printf("FAIL hint delivered twice\n"); fails++; } }
  { qmg::ZeroGuess zg; }
  if (qmg::take_zero_guess()) { printf("FAIL hint leaked out of scope\n"); fails++; }

  // ---- 3. batch views and masks
  if (qmg::full_mask(5) != 0x1Fu || qmg::full_mask(16) != 0xFFFFu) { printf("FAIL full_mask\n"); fails++; }
  if (!qmg::is_active(0b0100u, 2) || qmg::is_active(0b0100u, 1)) { printf("FAIL is_active\n"); fails++; }
  cd* base = reinterpret_cast<cd*>(0x1000);
  qmg::Batch bt(base, 100, 4);
  if (bt.vec(3) != base + 300 || batch_odd_half(bt, 50).vec(1) != base + 150) { printf("FAIL batch views\n"); fails++; }
  if (!BatchOp::supported(QMG_MATVEC_ORIGINAL) || !BatchOp::supported(QMG_MATVEC_RIGHT_SCHUR) || BatchOp::supported(QMG_MATVEC_M_MDAGGER)) { printf("FAIL BatchOp::supported\n"); fails++; }

  printf("%s (%d failures)\n", fails ? "HOST LOGIC FAILED" : "host logic ok", fails);
  return fails ? 1 : 0;
}
