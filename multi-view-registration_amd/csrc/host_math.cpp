// csrc/host_math.cpp -- the O(1) host-side solves of the hot path.
//
// north_star keeps "the 3x3 SVD on host": the GPU reduces a scan pair to a
// handful of f64 moments, the host turns them into poses.
//   * Umeyama / Kabsch      : TransformationEstimationSVD inside icp.align
//                             (mvr/src/registrator.cpp:569)
//   * LUM edge + graph solve: pcl::registration::LUM::computeEdge / compute
//                             (mvr/src/registrator.cpp:627-663; App. A.6), here
//                             evaluated from raw second moments so that all 16
//                             LUM iterations of one outer pass need no further
//                             pass over the points.
//   * turntable prior       : PointCloud::initRotation (point_cloud.cpp:400-413),
//                             Registrator::getRotationMatrix (registrator.cpp:331-342)
// Pure host C++, double precision; results are cast to float where the
// reference's types are float (Eigen::Matrix4f).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include <immintrin.h>

#include "mvr_internal.h"

namespace mvr {

namespace {

struct M3 {
  double a[3][3];
  double &operator()(int r, int c) { return a[r][c]; }
  double operator()(int r, int c) const { return a[r][c]; }
};

M3 zero3() { M3 m; std::memset(&m, 0, sizeof m); return m; }
M3 mul(const M3 &A, const M3 &B)
{
  M3 C = zero3();
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) C(r, c) += A(r, k) * B(k, c);
  return C;
}
M3 transpose(const M3 &A) { M3 T; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) T(r, c) = A(c, r); return T; }
M3 add(const M3 &A, const M3 &B, double sb = 1.0) { M3 C; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C(r, c) = A(r, c) + sb * B(r, c); return C; }
M3 outer(const double u[3], const double v[3]) { M3 C; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C(r, c) = u[r] * v[c]; return C; }
void mulv(const M3 &A, const double v[3], double out[3]) { for (int r = 0; r < 3; ++r) out[r] = A(r, 0) * v[0] + A(r, 1) * v[1] + A(r, 2) * v[2]; }
M3 sym6(const double s[6]) { M3 m; m(0, 0) = s[0]; m(0, 1) = m(1, 0) = s[1]; m(0, 2) = m(2, 0) = s[2]; m(1, 1) = s[3]; m(1, 2) = m(2, 1) = s[4]; m(2, 2) = s[5]; return m; }
double det(const M3 &m)
{
  return m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) - m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
         m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
}

}  // namespace

// Hestenes one-sided Jacobi; singular values descending (Eigen::JacobiSVD order).
void svd3(const double A[9], double U[9], double S[3], double V[9])
{
  double col[3][3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};   // col[j] = j-th column of A*V ; v[j] = j-th column of V
  for (int j = 0; j < 3; ++j) for (int r = 0; r < 3; ++r) col[j][r] = A[3 * r + j];
  for (int sweep = 0; sweep < 64; ++sweep) {
    bool any = false;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double app = 0, aqq = 0, apq = 0;
        for (int r = 0; r < 3; ++r) { app += col[p][r] * col[p][r]; aqq += col[q][r] * col[q][r]; apq += col[p][r] * col[q][r]; }
        // columns orthogonal to working precision (|cos| <= 2^-53): rotating further only churns the last bits.
        // (a tau whose square overflows gives t = 0, the right limit)
        if (apq * apq <= 1.2e-32 * (app * aqq)) continue;
        any = true;
        const double tau = (aqq - app) / (2.0 * apq);
        const double t = std::copysign(1.0, tau) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
        const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
        for (int r = 0; r < 3; ++r) {
          const double x = col[p][r], y = col[q][r];
          col[p][r] = cs * x - sn * y; col[q][r] = sn * x + cs * y;
          const double vx = v[p][r], vy = v[q][r];
          v[p][r] = cs * vx - sn * vy; v[q][r] = sn * vx + cs * vy;
        }
      }
    if (!any) break;
  }
  double nrm[3]; int ord[3] = {0, 1, 2};
  for (int j = 0; j < 3; ++j) nrm[j] = std::sqrt(col[j][0] * col[j][0] + col[j][1] * col[j][1] + col[j][2] * col[j][2]);
  std::sort(ord, ord + 3, [&](int x, int y) { return nrm[x] > nrm[y]; });
  double u[3][3];
  for (int j = 0; j < 3; ++j) {
    const int o = ord[j];
    S[j] = nrm[o];
    for (int r = 0; r < 3; ++r) { V[3 * r + j] = v[o][r]; u[j][r] = nrm[o] > 0 ? col[o][r] / nrm[o] : 0.0; }
  }
  if (S[0] <= DBL_MIN) { u[0][0] = 1; u[0][1] = 0; u[0][2] = 0; u[1][0] = 0; u[1][1] = 1; u[1][2] = 0; S[1] = S[2] = 0; }
  if (S[1] <= S[0] * 1e-15) {   // rank <= 1: any unit vector orthogonal to u0
    int k = 0;
    for (int r = 1; r < 3; ++r) if (std::fabs(u[0][r]) < std::fabs(u[0][k])) k = r;
    double w[3] = {-u[0][k] * u[0][0], -u[0][k] * u[0][1], -u[0][k] * u[0][2]};
    w[k] += 1.0;
    const double n = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    for (int r = 0; r < 3; ++r) u[1][r] = w[r] / n;
  }
  if (S[2] <= S[0] * 1e-15) {   // rank <= 2: u2 = u0 x u1
    u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
    u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
    u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
  }
  for (int j = 0; j < 3; ++j) for (int r = 0; r < 3; ++r) U[3 * r + j] = u[j][r];
}

// Eigen::umeyama(src, dst, false), Eigen >= 3.3 sign rule.
void umeyama_from_moments(const double mean_src[3], const double mean_tgt[3], const double sigma[9], float T[16],
                          double sv[3])
{
  double U[9], S[3], V[9];
  svd3(sigma, U, S, V);
  M3 u, v;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { u(r, c) = U[3 * r + c]; v(r, c) = V[3 * r + c]; }
  const double flip = (det(u) * det(v) < 0.0) ? -1.0 : 1.0;
  for (int r = 0; r < 3; ++r) u(r, 2) *= flip;
  const M3 R = mul(u, transpose(v));
  double Rp[3];
  mulv(R, mean_src, Rp);
  std::memset(T, 0, 16 * sizeof(float));
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T[r + 4 * c] = (float)R(r, c);
    T[12 + r] = (float)(mean_tgt[r] - Rp[r]);
  }
  T[15] = 1.0f;
  if (sv) { sv[0] = S[0]; sv[1] = S[1]; sv[2] = S[2]; }
}

int invert6(const double A[36], double Ainv[36])
{
  double W[6][12];
  for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { W[r][c] = A[6 * r + c]; W[r][6 + c] = (r == c) ? 1.0 : 0.0; }
  for (int k = 0; k < 6; ++k) {
    int piv = k;
    for (int r = k + 1; r < 6; ++r) if (std::fabs(W[r][k]) > std::fabs(W[piv][k])) piv = r;
    if (W[piv][k] == 0.0) return MVR_E_SINGULAR;
    if (piv != k) for (int c = 0; c < 12; ++c) std::swap(W[k][c], W[piv][c]);
    const double inv = 1.0 / W[k][k];
    for (int c = 0; c < 12; ++c) W[k][c] *= inv;
    for (int r = 0; r < 6; ++r) {
      if (r == k || W[r][k] == 0.0) continue;
      const double f = W[r][k];
      for (int c = 0; c < 12; ++c) W[r][c] -= f * W[k][c];
    }
  }
  for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Ainv[6 * r + c] = W[r][6 + c];
  return MVR_OK;
}

// x = A^-1 b for one right-hand side: elimination with partial pivoting on the 6 x 7 augmented matrix (a
// quarter of the arithmetic of forming the inverse; the LUM loop solves one such system per edge and per vertex
// in every iteration)
int solve6(const double A[36], const double b[6], double x[6])
{
  double W[6][7];
  for (int r = 0; r < 6; ++r) { for (int c = 0; c < 6; ++c) W[r][c] = A[6 * r + c]; W[r][6] = b[r]; }
  for (int k = 0; k < 6; ++k) {
    int piv = k;
    for (int r = k + 1; r < 6; ++r) if (std::fabs(W[r][k]) > std::fabs(W[piv][k])) piv = r;
    if (W[piv][k] == 0.0) return MVR_E_SINGULAR;
    if (piv != k) for (int c = k; c < 7; ++c) std::swap(W[k][c], W[piv][c]);
    const double inv = 1.0 / W[k][k];
    for (int r = k + 1; r < 6; ++r) {
      const double f = W[r][k] * inv;
      if (f == 0.0) continue;
      for (int c = k + 1; c < 7; ++c) W[r][c] -= f * W[k][c];
    }
  }
  for (int k = 5; k >= 0; --k) {
    double s = W[k][6];
    for (int c = k + 1; c < 6; ++c) s -= W[k][c] * x[c];
    x[k] = s / W[k][k];
  }
  return MVR_OK;
}

// x = A^-1 b for a SYMMETRIC POSITIVE DEFINITE 6 x 6 (the normal matrix MM of LUM::computeEdge): Cholesky without
// pivot search -- a third of the dependent divisions and none of the row swaps of solve6, which the LUM loop
// called 12 times per iteration (0.2 of its 0.5 us per edge).  Falls back to solve6 when a pivot is not safely
// positive (degenerate correspondences: the caller's NaN / singular handling stays as it was).
int solve6_spd(const double A[36], const double b[6], double x[6])
{
  double L[6][6], y[6], dinv[6];
  double amax = 0.0;
  for (int k = 0; k < 6; ++k) amax = std::max(amax, std::fabs(A[7 * k]));
  for (int j = 0; j < 6; ++j) {
    double d = A[7 * j];
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
    if (!(d > 1e-13 * amax)) return solve6(A, b, x);
    const double ljj = std::sqrt(d), inv = 1.0 / ljj;
    L[j][j] = ljj; dinv[j] = inv;
    for (int i = j + 1; i < 6; ++i) {
      double v = A[6 * i + j];
      for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
      L[i][j] = v * inv;
    }
  }
  // (the substitutions multiply by the reciprocals the factorisation has: twelve divisions off the dependent chain)
  for (int i = 0; i < 6; ++i) { double v = b[i]; for (int k = 0; k < i; ++k) v -= L[i][k] * y[k]; y[i] = v * dinv[i]; }
  for (int i = 5; i >= 0; --i) { double v = y[i]; for (int k = i + 1; k < 6; ++k) v -= L[k][i] * x[k]; x[i] = v * dinv[i]; }
  return MVR_OK;
}

// Extension (K10): pcl::registration::TransformationEstimationPointToPlaneLLS --
// x = (A^T A)^-1 A^T b with x = (alpha, beta, gamma, tx, ty, tz), then
// constructTransformationMatrix: R = Rz(gamma) Ry(beta) Rx(alpha), t.
int p2plane_solve(const double ata_upper[21], const double atb[6], float T[16])
{
  double A[36], Ainv[36], x[6] = {0, 0, 0, 0, 0, 0};
  int t = 0;
  for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) { A[6 * r + c] = ata_upper[t]; A[6 * c + r] = ata_upper[t]; ++t; }
  if (invert6(A, Ainv) != MVR_OK) return MVR_E_SINGULAR;
  for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) x[r] += Ainv[6 * r + c] * atb[c];
  const double ca = std::cos(x[0]), sa = std::sin(x[0]), cb = std::cos(x[1]), sb = std::sin(x[1]);
  const double cg = std::cos(x[2]), sg = std::sin(x[2]);
  std::memset(T, 0, 16 * sizeof(float));
  T[0] = (float)(cg * cb);  T[4] = (float)(-sg * ca + cg * sb * sa);  T[8]  = (float)(sg * sa + cg * sb * ca);   T[12] = (float)x[3];
  T[1] = (float)(sg * cb);  T[5] = (float)(cg * ca + sg * sb * sa);   T[9]  = (float)(-cg * sa + sg * sb * ca);  T[13] = (float)x[4];
  T[2] = (float)(-sb);      T[6] = (float)(cb * sa);                  T[10] = (float)(cb * ca);                  T[14] = (float)x[5];
  T[15] = 1.0f;
  return MVR_OK;
}

int solve_dense(int n, double *A, double *b)
{
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int r = k + 1; r < n; ++r) if (std::fabs(A[(size_t)r * n + k]) > std::fabs(A[(size_t)piv * n + k])) piv = r;
    if (A[(size_t)piv * n + k] == 0.0) return MVR_E_SINGULAR;
    if (piv != k) { for (int c = 0; c < n; ++c) std::swap(A[(size_t)k * n + c], A[(size_t)piv * n + c]); std::swap(b[k], b[piv]); }
    for (int r = k + 1; r < n; ++r) {
      const double f = A[(size_t)r * n + k] / A[(size_t)k * n + k];
      if (f == 0.0) continue;
      for (int c = k; c < n; ++c) A[(size_t)r * n + c] -= f * A[(size_t)k * n + c];
      b[r] -= f * b[k];
    }
  }
  for (int k = n - 1; k >= 0; --k) {
    double s = b[k];
    for (int c = k + 1; c < n; ++c) s -= A[(size_t)k * n + c] * b[c];
    b[k] = s / A[(size_t)k * n + k];
  }
  return MVR_OK;
}

// Symmetric positive definite systems (the LUM normal equations G x = B are one unless the graph is
// degenerate): right-looking Cholesky U^T U on a scratch copy of the upper triangle.  Every inner loop is
// an axpy over a contiguous row segment (vectorisable without re-associating sums), rows are only as long
// as their last non-zero and zero multipliers are skipped, so the block-banded matrix of a ring of views
// costs O(n b^2) instead of O(n^3).  Falls back to the pivoted elimination (solve_dense) when a pivot is
// not safely positive.
// row_end (optional): one past the last column that can be non-zero in row j of the upper triangle, when the
// caller knows the structure (saves the scan of the matrix).
int solve_spd(int n, double *A, double *b, const int *row_end = nullptr)
{
  // Scratch: rows of `stride` doubles; row j is valid on [j, end[j]) and ZERO up to j + pad, so the row updates
  // below run over whole 4-wide vectors without a scalar tail (the rows of a ring's normal equations are at most
  // 12 long: a counted loop with a remainder cost more than the arithmetic).  Element by element the operations
  // are the ones of the scalar loop (multiply, then subtract: no contraction), so the bits are the same.
  static thread_local std::vector<double> U, dinv, rhs;
  static thread_local std::vector<int> end;           // one past the last non-zero of row j (upper part)
  if (end.size() < (size_t)n) { end.resize((size_t)n); dinv.resize((size_t)n); }
  double amax = 0.0;
  int width = 1;
  for (int j = 0; j < n; ++j) {
    amax = std::max(amax, std::fabs(A[(size_t)j * n + j]));
    int e = j + 1;
    if (row_end) e = std::max(e, std::min(n, row_end[j]));
    else for (int k = n - 1; k > j; --k) if (A[(size_t)j * n + k] != 0.0) { e = k + 1; break; }
    end[j] = e;
    width = std::max(width, e - j);
  }
  // fill-in never reaches beyond the widest row's span from the pivot, so every row stays within `width` of its diagonal
  const int pad = ((width + 3) & ~3) + 4;
  const size_t stride = (size_t)n + (size_t)pad;
  if (U.size() < (size_t)n * stride) U.resize((size_t)n * stride);
  if (rhs.size() < stride) rhs.resize(stride);
  for (int j = 0; j < n; ++j) {
    double *Uj = &U[(size_t)j * stride];
    std::memset(Uj + j, 0, (size_t)pad * sizeof(double));
    std::memcpy(Uj + j, &A[(size_t)j * n + j], (size_t)(end[j] - j) * sizeof(double));
  }
  bool ok = amax > 0.0;
  for (int j = 0; j < n && ok; ++j) {
    double *Uj = &U[(size_t)j * stride];
    const double d = Uj[j];
    if (!(d > 1e-13 * amax)) { ok = false; break; }
    const double ujj = std::sqrt(d), inv = 1.0 / ujj;
    const int ej = end[j];
    Uj[j] = ujj; dinv[j] = inv;
    const __m256d vinv = _mm256_set1_pd(inv);
    for (int k = j + 1; k < ej; k += 4) _mm256_storeu_pd(Uj + k, _mm256_mul_pd(_mm256_loadu_pd(Uj + k), vinv));
    for (int i = j + 1; i < ej; ++i) {
      const double f = Uj[i];
      if (f == 0.0) continue;
      double *Ui = &U[(size_t)i * stride];
      if (end[i] < ej) end[i] = ej;                     // fill-in (the row is zero there already)
      const __m256d vf = _mm256_set1_pd(f);
      for (int k = i; k < ej; k += 4)
        _mm256_storeu_pd(Ui + k, _mm256_sub_pd(_mm256_loadu_pd(Ui + k), _mm256_mul_pd(vf, _mm256_loadu_pd(Uj + k))));
    }
  }
  if (!ok) return solve_dense(n, A, b);
  double *y = rhs.data();
  std::memcpy(y, b, (size_t)n * sizeof(double));
  std::memset(y + n, 0, (size_t)pad * sizeof(double));
  for (int i = 0; i < n; ++i) {                       // U^T y = b, column-oriented: axpy over row i of U
    const double yi = y[i] * dinv[i];                 // (a multiply keeps the divider off the dependent chain)
    y[i] = yi;
    const double *Ui = &U[(size_t)i * stride];
    const __m256d vy = _mm256_set1_pd(yi);
    for (int k = i + 1; k < end[i]; k += 4)
      _mm256_storeu_pd(y + k, _mm256_sub_pd(_mm256_loadu_pd(y + k), _mm256_mul_pd(_mm256_loadu_pd(Ui + k), vy)));
  }
  for (int i = n - 1; i >= 0; --i) {                  // U x = y (a sum: kept in order)
    const double *Ui = &U[(size_t)i * stride];
    double sacc = y[i];
    for (int k = i + 1; k < end[i]; ++k) sacc -= Ui[k] * y[k];
    y[i] = sacc * dinv[i];
  }
  std::memcpy(b, y, (size_t)n * sizeof(double));
  return MVR_OK;
}

// The same factorisation on BAND storage, for systems whose upper rows reach at most W entries from the diagonal (a ring or
// chain of views: W = 12, one block beside the diagonal one): row j lives in band[j * S .. + W) as U(j, j .. j + W), zero up
// to S = 2 W, so every row update below is three fixed 4-wide operations per 12 columns -- no counted loops, no scan for the
// row ends, no dense n x n matrix to clear and copy (a 66 x 66 solve: 7k -> 3k cycles).  Element by element the operations
// and their order are those of solve_spd (multiply, then subtract; the back substitution in column order); the only
// additions are updates by an exact zero.  false: a pivot is not safely positive (the caller takes the dense route).
template <int W>
bool solve_spd_band(int n, double *band, double *b, double *dinv, double *y)
{
  constexpr int S = 2 * W;
  static_assert(W % 4 == 0, "rows are walked four columns at a time");
  double amax = 0.0;
  for (int j = 0; j < n; ++j) amax = std::max(amax, std::fabs(band[(size_t)j * S]));
  if (!(amax > 0.0)) return false;
  for (int j = 0; j < n; ++j) {
    double *Uj = band + (size_t)j * S;
    const double d = Uj[0];
    if (!(d > 1e-13 * amax)) return false;
    const double ujj = std::sqrt(d), inv = 1.0 / ujj;
    dinv[j] = inv;
    const __m256d vinv = _mm256_set1_pd(inv);
    for (int k = 0; k < W; k += 4) _mm256_storeu_pd(Uj + k, _mm256_mul_pd(_mm256_loadu_pd(Uj + k), vinv));
    Uj[0] = ujj;
    const int last = std::min(W, n - j);
    for (int i = 1; i < last; ++i) {
      const double f = Uj[i];
      if (f == 0.0) continue;
      double *Ui = band + (size_t)(j + i) * S;
      const __m256d vf = _mm256_set1_pd(f);
      for (int k = 0; k < W; k += 4)          // columns j + i + k: U(j, .) from offset i on (zero beyond the band)
        _mm256_storeu_pd(Ui + k, _mm256_sub_pd(_mm256_loadu_pd(Ui + k), _mm256_mul_pd(vf, _mm256_loadu_pd(Uj + i + k))));
    }
  }
  std::memcpy(y, b, (size_t)n * sizeof(double));
  std::memset(y + n, 0, (size_t)S * sizeof(double));
  for (int i = 0; i < n; ++i) {                       // U^T y = b, column-oriented: axpy over row i of U
    const double yi = y[i] * dinv[i];
    const double *Ui = band + (size_t)i * S;
    const __m256d vy = _mm256_set1_pd(yi);
    // (entries 1 .. W-1 of the row against y[i + 1 ..]; entry 0 is the diagonal: that lane is put back below)
    for (int k = 0; k < W; k += 4)
      _mm256_storeu_pd(y + i + k, _mm256_sub_pd(_mm256_loadu_pd(y + i + k), _mm256_mul_pd(_mm256_loadu_pd(Ui + k), vy)));
    y[i] = yi;
  }
  for (int i = n - 1; i >= 0; --i) {                  // U x = y (a sum: kept in order)
    const double *Ui = band + (size_t)i * S;
    double sacc = y[i];
    const int last = std::min(W, n - i);
    for (int k = 1; k < last; ++k) sacc -= Ui[k] * y[i + k];
    y[i] = sacc * dinv[i];
  }
  std::memcpy(b, y, (size_t)n * sizeof(double));
  return true;
}

// The normal equations of a RING or CHAIN of views are block tridiagonal (6 x 6 blocks: a view couples to the next and the
// previous one only; the fixed view 0 just adds to its neighbours' diagonal blocks).  Block Cholesky with everything of a
// block step in registers: D_v = L L^T, y_v = L^-1 b_v, W = L^-1 O_v, D_{v+1} -= W^T W, b_{v+1} -= W^T y_v; back substitution
// x_v = L^-T (y_v - W_v x_{v+1}).  All loops have constant trip counts (6): the compiler unrolls them, no row ends, no zero
// tests, no memory traffic beyond the blocks themselves -- what is left is the chain of 6 (nb) dependent square roots and
// divisions.  The ARITHMETIC differs from the row-wise elimination of solve_spd / solve_spd_band in the order of the updates
// (results agree to rounding: tests/test_host.py), so it is one route for every caller, not a choice per call.
// D: nb blocks (row-major, symmetric, destroyed), O: nb - 1 coupling blocks G(v, v + 1), b: nb * 6 in / out.
// false: a pivot is not safely positive (the caller takes the generic route).
// Blocks are stored with rows of EIGHT doubles (six used): a row is two 4-wide vectors, and the two O(6^3) parts -- W = L^-1 O
// and the update D_{v+1} -= W^T W -- run a row at a time.
static bool solve_chain6(int nb, double *D, double *O, double *b)
{
  constexpr int R = 8, B2 = 6 * R;        // row stride, block size
  double amax = 0.0;
  for (int v = 0; v < nb; ++v) for (int k = 0; k < 6; ++k) amax = std::max(amax, std::fabs(D[B2 * (size_t)v + (R + 1) * k]));
  if (!(amax > 0.0)) return false;
  const double tiny = 1e-13 * amax;
  // W = L^-1 C (C: six rows of eight, overwritten), then Dn -= W^T W, bn -= W^T y: the two O(6^3) parts of a block step, a row of
  // two 4-wide vectors at a time.  A: the factorised block (L below the diagonal), dinv: its inverted diagonal.
  auto couple = [](const double *A, const double *dinv, const double *y, double *W, double *Dn, double *bn) {
    __m256d w0[6], w1[6];
    for (int i = 0; i < 6; ++i) {
      __m256d r0 = _mm256_loadu_pd(W + R * i), r1 = _mm256_loadu_pd(W + R * i + 4);
      for (int k = 0; k < i; ++k) {
        const __m256d l = _mm256_set1_pd(A[R * i + k]);
        r0 = _mm256_sub_pd(r0, _mm256_mul_pd(l, w0[k])); r1 = _mm256_sub_pd(r1, _mm256_mul_pd(l, w1[k]));
      }
      const __m256d di = _mm256_set1_pd(dinv[i]);
      w0[i] = _mm256_mul_pd(r0, di); w1[i] = _mm256_mul_pd(r1, di);
      _mm256_storeu_pd(W + R * i, w0[i]); _mm256_storeu_pd(W + R * i + 4, w1[i]);
    }
    for (int r = 0; r < 6; ++r) {            // Dn -= W^T W (row r: sum over k of W[k][r] * row k of W), bn -= W^T y
      __m256d a0 = _mm256_loadu_pd(Dn + R * r), a1 = _mm256_loadu_pd(Dn + R * r + 4);
      double t = 0.0;
      for (int k = 0; k < 6; ++k) {
        const double wkr = W[R * k + r];
        const __m256d f = _mm256_set1_pd(wkr);
        a0 = _mm256_sub_pd(a0, _mm256_mul_pd(f, w0[k])); a1 = _mm256_sub_pd(a1, _mm256_mul_pd(f, w1[k]));
        t += wkr * y[k];
      }
      _mm256_storeu_pd(Dn + R * r, a0); _mm256_storeu_pd(Dn + R * r + 4, a1);
      bn[r] -= t;
    }
  };
  // one block by itself: D_v = L L^T, y_v = L^-1 b_v, and (next) the coupling to block v + 1
  auto step_single = [&](int v, bool next) -> bool {
    double *A = D + B2 * (size_t)v;          // becomes L (lower triangle), diagonal inverted in dinv
    double dinv[6];
    for (int j = 0; j < 6; ++j) {
      double d = A[(R + 1) * j];
      for (int k = 0; k < j; ++k) d -= A[R * j + k] * A[R * j + k];
      if (!(d > tiny)) return false;
      const double l = std::sqrt(d), inv = l * (1.0 / d);      // 1 / sqrt(d): the root and the reciprocal run side by side
      A[(R + 1) * j] = l; dinv[j] = inv;
      for (int i = j + 1; i < 6; ++i) {
        double sacc = A[R * i + j];
        for (int k = 0; k < j; ++k) sacc -= A[R * i + k] * A[R * j + k];
        A[R * i + j] = sacc * inv;
      }
    }
    double *y = b + 6 * (size_t)v;
    for (int i = 0; i < 6; ++i) {              // y = L^-1 b
      double sacc = y[i];
      for (int k = 0; k < i; ++k) sacc -= A[R * i + k] * y[k];
      y[i] = sacc * dinv[i];
    }
    if (next) couple(A, dinv, y, O + B2 * (size_t)v, D + B2 * (size_t)(v + 1), b + 6 * (size_t)(v + 1));
    for (int j = 0; j < 6; ++j) A[(R + 1) * j] = dinv[j];      // (the back substitution wants the inverted diagonal)
    return true;
  };
  // What this routine costs is its chain of 6 nb dependent pivots (a square root and a division each).  So the chain is eliminated
  // from BOTH ends at once (a twisted factorisation): block s upwards and block nb - 1 - s downwards are two independent
  // factorisations, run as the two lanes of 128-bit vectors -- the same instructions, half the chain; the one or two blocks in the
  // middle take both sides' updates and are factorised last.  Lane 0 does exactly what step_single does.
  const int pairs = (nb - 1) / 2;
  for (int s = 0; s < pairs; ++s) {
    const int va = s, vb = nb - 1 - s;
    double *A = D + B2 * (size_t)va, *Bk = D + B2 * (size_t)vb;
    double *ya = b + 6 * (size_t)va, *yb = b + 6 * (size_t)vb;
    __m128d L2[6][6], dinv2[6], y2[6];
    const __m128d tiny2 = _mm_set1_pd(tiny), one2 = _mm_set1_pd(1.0);
    for (int j = 0; j < 6; ++j) {
      __m128d d = _mm_set_pd(Bk[(R + 1) * j], A[(R + 1) * j]);
      for (int k = 0; k < j; ++k) d = _mm_sub_pd(d, _mm_mul_pd(L2[j][k], L2[j][k]));
      if (_mm_movemask_pd(_mm_cmpgt_pd(d, tiny2)) != 3) return false;
      const __m128d l = _mm_sqrt_pd(d), inv = _mm_mul_pd(l, _mm_div_pd(one2, d));
      L2[j][j] = l; dinv2[j] = inv;
      for (int i = j + 1; i < 6; ++i) {
        __m128d sacc = _mm_set_pd(Bk[R * i + j], A[R * i + j]);
        for (int k = 0; k < j; ++k) sacc = _mm_sub_pd(sacc, _mm_mul_pd(L2[i][k], L2[j][k]));
        L2[i][j] = _mm_mul_pd(sacc, inv);
      }
    }
    for (int i = 0; i < 6; ++i) {              // y = L^-1 b, both
      __m128d sacc = _mm_set_pd(yb[i], ya[i]);
      for (int k = 0; k < i; ++k) sacc = _mm_sub_pd(sacc, _mm_mul_pd(L2[i][k], y2[k]));
      y2[i] = _mm_mul_pd(sacc, dinv2[i]);
    }
    double da[6], db[6];
    for (int i = 0; i < 6; ++i) {
      for (int j = 0; j <= i; ++j) { _mm_storel_pd(A + R * i + j, L2[i][j]); _mm_storeh_pd(Bk + R * i + j, L2[i][j]); }
      _mm_storel_pd(ya + i, y2[i]); _mm_storeh_pd(yb + i, y2[i]);
      _mm_storel_pd(da + i, dinv2[i]); _mm_storeh_pd(db + i, dinv2[i]);
    }
    // upwards: W = La^-1 O_va onto block va + 1; downwards: U = Lb^-1 O_(vb-1)^T onto block vb - 1 (U replaces O_(vb-1), row-major)
    couple(A, da, ya, O + B2 * (size_t)va, D + B2 * (size_t)(va + 1), b + 6 * (size_t)(va + 1));
    {
      double *Ob = O + B2 * (size_t)(vb - 1), Ct[B2];
      for (int i = 0; i < 6; ++i) { for (int c = 0; c < 6; ++c) Ct[R * i + c] = Ob[R * c + i]; Ct[R * i + 6] = 0.0; Ct[R * i + 7] = 0.0; }
      std::memcpy(Ob, Ct, sizeof Ct);
      couple(Bk, db, yb, Ob, D + B2 * (size_t)(vb - 1), b + 6 * (size_t)(vb - 1));
    }
    for (int j = 0; j < 6; ++j) { A[(R + 1) * j] = da[j]; Bk[(R + 1) * j] = db[j]; }
  }
  const int lo = pairs, hi = nb - 1 - pairs;      // the middle: one block (nb odd) or two
  for (int v = lo; v <= hi; ++v) if (!step_single(v, v < hi)) return false;
  // back substitution: the middle first, then outwards on both sides
  auto back = [&](int v, const double *Wc, const double *xn) {      // x_v = L^-T (y_v - Wc x_n)
    const double *A = D + B2 * (size_t)v;
    double *x = b + 6 * (size_t)v;
    if (Wc) {
      for (int i = 0; i < 6; ++i) {
        double t = 0.0;
        for (int c = 0; c < 6; ++c) t += Wc[R * i + c] * xn[c];
        x[i] -= t;
      }
    }
    for (int i = 5; i >= 0; --i) {
      double sacc = x[i];
      for (int k = i + 1; k < 6; ++k) sacc -= A[R * k + i] * x[k];
      x[i] = sacc * A[(R + 1) * i];
    }
  };
  for (int v = hi; v >= lo; --v) back(v, v < hi ? O + B2 * (size_t)v : nullptr, b + 6 * (size_t)(v + 1));
  for (int s = pairs - 1; s >= 0; --s) {
    const int va = s, vb = nb - 1 - s;
    back(va, O + B2 * (size_t)va, b + 6 * (size_t)(va + 1));
    back(vb, O + B2 * (size_t)(vb - 1), b + 6 * (size_t)(vb - 1));
  }
  return true;
}

}  // namespace mvr

using namespace mvr;

extern "C" {

#define API __attribute__((visibility("default")))

API int mvr_umeyama_from_moments(const mvr_pair_moments_t *mom, float T[16], double sv[3])
{
  if (!mom || !T) return MVR_E_ARG;
  if (mom->n < 3.0) return MVR_E_NOCORR;
  umeyama_from_moments(mom->mean_src, mom->mean_tgt, mom->sigma, T, sv);
  return MVR_OK;
}

API int mvr_moments_from_moments2(const mvr_pair_moments2_t *m2, mvr_pair_moments_t *out)
{
  if (!m2 || !out) return MVR_E_ARG;
  std::memset(out, 0, sizeof *out);
  out->n = m2->n;
  if (m2->n <= 0) return MVR_OK;
  const double inv = 1.0 / m2->n;
  double mp[3], mq[3];
  for (int k = 0; k < 3; ++k) { mp[k] = m2->sp[k] * inv; mq[k] = m2->sq[k] * inv; out->mean_src[k] = mp[k] + m2->origin[k]; out->mean_tgt[k] = mq[k] + m2->origin[k]; }
  // sum |p-q|^2 = tr(spp) - 2 tr(spq) + tr(sqq)
  out->mse = ((m2->spp[0] + m2->spp[3] + m2->spp[5]) - 2.0 * (m2->spq[0] + m2->spq[4] + m2->spq[8]) +
              (m2->sqq[0] + m2->sqq[3] + m2->sqq[5])) * inv;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) out->sigma[3 * r + c] = m2->spq[3 * c + r] * inv - mq[r] * mp[c];
  return MVR_OK;
}

// pcl::getTransformation(x, y, z, roll, pitch, yaw) = Translation * Rz(yaw) Ry(pitch) Rx(roll)
// (the six trigonometric values of a pose, computed once per LUM iteration and vertex: the matrix and the incidence
// correction of the same pose both need them -- 116 libm calls per iteration were half of its time)
struct PoseTrig { double cr, sr, cp, sp, cy, sy; };
static inline PoseTrig pose_trig(const double pose[6])
{
  PoseTrig t;
  t.cr = std::cos(pose[3]); t.sr = std::sin(pose[3]);
  t.cp = std::cos(pose[4]); t.sp = std::sin(pose[4]);
  t.cy = std::cos(pose[5]); t.sy = std::sin(pose[5]);
  return t;
}
static inline void pose_to_mat4_trig(const double pose[6], const PoseTrig &g, double T[16])
{
  const double cr = g.cr, sr = g.sr, cp = g.cp, sp = g.sp, cy = g.cy, sy = g.sy;
  std::memset(T, 0, 16 * sizeof(double));
  T[0] = cy * cp; T[4] = cy * sp * sr - sy * cr; T[8]  = sy * sr + cy * sp * cr; T[12] = pose[0];
  T[1] = sy * cp; T[5] = cy * cr + sy * sp * sr; T[9]  = sy * sp * cr - cy * sr; T[13] = pose[1];
  T[2] = -sp;     T[6] = cp * sr;                T[10] = cp * cr;                T[14] = pose[2];
  T[15] = 1.0;
}
API void mvr_pose_to_mat4(const double pose[6], double T[16])
{
  pose_to_mat4_trig(pose, pose_trig(pose), T);
}

// LUM::computeEdge from raw moments.  With p' = p - o,
// q' = q - o the compounded points are a' = Rs p' + cs, b' = Rt q' + ct
// (cs = Rs o + ts - o), so every sum over aver = (a+b)/2 and diff = a-b is an
// algebraic function of {n, sum p', sum q', sum p'p'^T, sum q'q'^T, sum p'q'^T}.
// Work in the shifted frame (small magnitudes), shift MM/MZ back at the end.
static int lum_edge_from_moments_T(const mvr_pair_moments2_t *m2, const double Ts[16], const double Tt[16], double MM[36],
                                   double MZ[6], double *ss);

API int mvr_lum_edge_from_moments(const mvr_pair_moments2_t *m2, const double pose_s[6], const double pose_t[6],
                                  double MM[36], double MZ[6], double *ss)
{
  if (!m2 || !pose_s || !pose_t || !MM || !MZ || !ss) return MVR_E_ARG;
  double Ts[16], Tt[16];
  mvr_pose_to_mat4(pose_s, Ts); mvr_pose_to_mat4(pose_t, Tt);
  return lum_edge_from_moments_T(m2, Ts, Tt, MM, MZ, ss);
}

// the same with the two vertex poses already as 4x4 (LUM::compute converts every vertex once per iteration, not
// once per edge end: the twelve sin / cos per edge were a third of its cost)
static int lum_edge_from_moments_T(const mvr_pair_moments2_t *m2, const double Ts[16], const double Tt[16], double MM[36],
                                   double MZ[6], double *ss)
{
  std::memset(MM, 0, 36 * sizeof(double)); std::memset(MZ, 0, 6 * sizeof(double)); *ss = 0.0;
  const double n = m2->n;
  if (n < 3.0) return MVR_E_NOCORR;
  M3 Rs, Rt;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Rs(r, c) = Ts[r + 4 * c]; Rt(r, c) = Tt[r + 4 * c]; }
  const double *o = m2->origin;
  double cs[3], ct[3], t3[3];
  mulv(Rs, o, t3); for (int k = 0; k < 3; ++k) cs[k] = t3[k] + Ts[12 + k] - o[k];
  mulv(Rt, o, t3); for (int k = 0; k < 3; ++k) ct[k] = t3[k] + Tt[12 + k] - o[k];
  double Rsp[3], Rtq[3];
  mulv(Rs, m2->sp, Rsp); mulv(Rt, m2->sq, Rtq);
  double sa[3], sb[3];      // sum a', sum b'
  for (int k = 0; k < 3; ++k) { sa[k] = Rsp[k] + n * cs[k]; sb[k] = Rtq[k] + n * ct[k]; }
  M3 spq; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) spq(r, c) = m2->spq[3 * r + c];
  // S_aa = Rs spp Rs^T + (Rs sp) cs^T + cs (Rs sp)^T + n cs cs^T
  M3 Saa = add(add(mul(mul(Rs, sym6(m2->spp)), transpose(Rs)), outer(Rsp, cs)), add(outer(cs, Rsp), outer(cs, cs), n));
  M3 Sbb = add(add(mul(mul(Rt, sym6(m2->sqq)), transpose(Rt)), outer(Rtq, ct)), add(outer(ct, Rtq), outer(ct, ct), n));
  // S_ab = Rs spq Rt^T + (Rs sp) ct^T + cs (Rt sq)^T + n cs ct^T
  M3 Sab = add(add(mul(mul(Rs, spq), transpose(Rt)), outer(Rsp, ct)), add(outer(cs, Rtq), outer(cs, ct), n));
  M3 Sba = transpose(Sab);
  // shifted-frame sums
  double sav[3], sdf[3];
  for (int k = 0; k < 3; ++k) { sav[k] = 0.5 * (sa[k] + sb[k]); sdf[k] = sa[k] - sb[k]; }
  M3 Savav, Savdf;   // sum av' av'^T ; sum av' diff^T
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
    Savav(r, c) = 0.25 * (Saa(r, c) + Sab(r, c) + Sba(r, c) + Sbb(r, c));
    Savdf(r, c) = 0.5 * (Saa(r, c) - Sab(r, c) + Sba(r, c) - Sbb(r, c));
  }
  const double tr_dfdf = (Saa(0, 0) + Saa(1, 1) + Saa(2, 2)) - 2.0 * (Sab(0, 0) + Sab(1, 1) + Sab(2, 2)) +
                         (Sbb(0, 0) + Sbb(1, 1) + Sbb(2, 2));
  // absolute-frame sums: av = av' + o
  double Sx[3]; M3 Sxx, Sxd;
  for (int k = 0; k < 3; ++k) Sx[k] = sav[k] + n * o[k];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
    Sxx(r, c) = Savav(r, c) + o[r] * sav[c] + sav[r] * o[c] + n * o[r] * o[c];
    Sxd(r, c) = Savdf(r, c) + o[r] * sdf[c];
  }
#define M(r, c) MM[6 * (r) + (c)]
  M(0, 4) = -Sx[1]; M(0, 5) = Sx[2]; M(1, 3) = -Sx[2]; M(1, 4) = Sx[0]; M(2, 3) = Sx[1]; M(2, 5) = -Sx[0];
  M(3, 4) = -Sxx(0, 2); M(3, 5) = -Sxx(0, 1); M(4, 5) = -Sxx(1, 2);
  M(3, 3) = Sxx(1, 1) + Sxx(2, 2); M(4, 4) = Sxx(0, 0) + Sxx(1, 1); M(5, 5) = Sxx(0, 0) + Sxx(2, 2);
  M(0, 0) = M(1, 1) = M(2, 2) = n;
  for (int r = 0; r < 6; ++r) for (int c = r + 1; c < 6; ++c) M(c, r) = M(r, c);
#undef M
  MZ[0] = sdf[0]; MZ[1] = sdf[1]; MZ[2] = sdf[2];
  MZ[3] = Sxd(1, 2) - Sxd(2, 1);   // sum (y dz - z dy)
  MZ[4] = Sxd(0, 1) - Sxd(1, 0);   // sum (x dy - y dx)
  MZ[5] = Sxd(2, 0) - Sxd(0, 2);   // sum (z dx - x dz)
  double D[6] = {0, 0, 0, 0, 0, 0};
  if (solve6_spd(MM, MZ, D) != MVR_OK) { *ss = NAN; return MVR_OK; }      // D = MM^-1 MZ
  // residual e = diff - Dt - C av, C = [[0,-D4,D5],[D4,0,-D3],[-D5,D3,0]]; in
  // the shifted frame e = diff - (Dt + C o) - C av'
  M3 Cm = zero3();
  Cm(0, 1) = -D[4]; Cm(0, 2) = D[5]; Cm(1, 0) = D[4]; Cm(1, 2) = -D[3]; Cm(2, 0) = -D[5]; Cm(2, 1) = D[3];
  double Co[3], Dt[3];
  mulv(Cm, o, Co);
  for (int k = 0; k < 3; ++k) Dt[k] = D[k] + Co[k];
  const M3 CtC = mul(transpose(Cm), Cm);
  double tr_ctc_savav = 0, tr_c_savdf = 0;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { tr_ctc_savav += CtC(r, c) * Savav(c, r); tr_c_savdf += Cm(r, c) * Savdf(c, r); }
  double Csav[3];
  mulv(Cm, sav, Csav);
  const double dt2 = Dt[0] * Dt[0] + Dt[1] * Dt[1] + Dt[2] * Dt[2];
  const double dt_sdf = Dt[0] * sdf[0] + Dt[1] * sdf[1] + Dt[2] * sdf[2];
  const double dt_csav = Dt[0] * Csav[0] + Dt[1] * Csav[1] + Dt[2] * Csav[2];
  *ss = tr_dfdf + n * dt2 + tr_ctc_savav - 2.0 * dt_sdf - 2.0 * tr_c_savdf + 2.0 * dt_csav;
  return MVR_OK;
}

// ---- the same computeEdge for FOUR edges at once (one edge per AVX2 lane).  LUM::compute spends most of an iteration
// in the 12 (or 36) edges' small dense algebra -- a dozen 3 x 3 products and a 6 x 6 Cholesky each, all of it one long
// dependency chain per edge; four independent edges side by side fill the chain's bubbles.  Every lane executes exactly
// the operations of lum_edge_from_moments_T in the same order (IEEE add / mul / div / sqrt per lane, no contraction):
// the results are bit-identical to the scalar function, which tests/test_host.py asserts.  A group in which any lane
// needs one of the scalar function's special paths (fewer than 3 pairs, an unsafe Cholesky pivot) is recomputed by it.
namespace {
typedef double v4d __attribute__((ext_vector_type(4)));
typedef long v4i __attribute__((ext_vector_type(4)));
inline v4d vsplat(double x) { return v4d{x, x, x, x}; }
inline v4d vsqrt(v4d x) { return __builtin_elementwise_sqrt(x); }
inline v4d vabs(v4d x) { return __builtin_elementwise_abs(x); }
inline v4d vmax(v4d a, v4d b) { const v4i m = a < b; return v4d{m[0] ? b[0] : a[0], m[1] ? b[1] : a[1], m[2] ? b[2] : a[2], m[3] ? b[3] : a[3]}; }      // std::max(a, b): (a < b) ? b : a
struct M3v { v4d a[3][3]; v4d &operator()(int r, int c) { return a[r][c]; } const v4d &operator()(int r, int c) const { return a[r][c]; } };
inline M3v vzero3() { M3v m; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) m(r, c) = vsplat(0.0); return m; }
inline M3v vmul(const M3v &A, const M3v &B)
{
  M3v C = vzero3();
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) C(r, c) += A(r, k) * B(k, c);
  return C;
}
inline M3v vtranspose(const M3v &A) { M3v T; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) T(r, c) = A(c, r); return T; }
inline M3v vadd(const M3v &A, const M3v &B) { M3v C; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C(r, c) = A(r, c) + vsplat(1.0) * B(r, c); return C; }
inline M3v vadd_s(const M3v &A, const M3v &B, v4d sb) { M3v C; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C(r, c) = A(r, c) + sb * B(r, c); return C; }
inline M3v vouter(const v4d u[3], const v4d v[3]) { M3v C; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C(r, c) = u[r] * v[c]; return C; }
inline void vmulv(const M3v &A, const v4d v[3], v4d out[3]) { for (int r = 0; r < 3; ++r) out[r] = A(r, 0) * v[0] + A(r, 1) * v[1] + A(r, 2) * v[2]; }
inline M3v vsym6(const v4d s[6]) { M3v m; m(0, 0) = s[0]; m(0, 1) = m(1, 0) = s[1]; m(0, 2) = m(2, 0) = s[2]; m(1, 1) = s[3]; m(1, 2) = m(2, 1) = s[4]; m(2, 2) = s[5]; return m; }

// solve6_spd, four systems at once; false if any lane's pivot is not safely positive (the caller falls back)
inline bool solve6_spd_x4(const v4d A[36], const v4d b[6], v4d x[6])
{
  v4d L[6][6], y[6], dinv[6];
  v4d amax = vsplat(0.0);
  for (int k = 0; k < 6; ++k) amax = vmax(amax, vabs(A[7 * k]));
  for (int j = 0; j < 6; ++j) {
    v4d d = A[7 * j];
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
    const v4i okm = d > vsplat(1e-13) * amax;
    if (!(okm[0] && okm[1] && okm[2] && okm[3])) return false;
    const v4d ljj = vsqrt(d), inv = vsplat(1.0) / ljj;
    L[j][j] = ljj; dinv[j] = inv;
    for (int i = j + 1; i < 6; ++i) {
      v4d v = A[6 * i + j];
      for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
      L[i][j] = v * inv;
    }
  }
  for (int i = 0; i < 6; ++i) { v4d v = b[i]; for (int k = 0; k < i; ++k) v -= L[i][k] * y[k]; y[i] = v * dinv[i]; }
  for (int i = 5; i >= 0; --i) { v4d v = y[i]; for (int k = i + 1; k < 6; ++k) v -= L[k][i] * x[k]; x[i] = v * dinv[i]; }
  return true;
}

// m2 / Ts / Tt: four edges (lane l = edge l of the group); outputs row-major per lane.  false: use the scalar function.
bool lum_edge_from_moments_x4(const mvr_pair_moments2_t *const m2[4], const double *const Ts[4], const double *const Tt[4],
                              double MM[4][36], double MZ[4][6], double ss[4])
{
  for (int l = 0; l < 4; ++l) if (m2[l]->n < 3.0) return false;
#define LANES(expr) v4d{[&](int l) { return (expr); }(0), [&](int l) { return (expr); }(1), [&](int l) { return (expr); }(2), [&](int l) { return (expr); }(3)}
  const v4d n = LANES(m2[l]->n);
  M3v Rs, Rt;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Rs(r, c) = LANES(Ts[l][r + 4 * c]); Rt(r, c) = LANES(Tt[l][r + 4 * c]); }
  v4d o[3], tsv[3], ttv[3], sp[3], sq[3], spp[6], sqq[6];
  for (int k = 0; k < 3; ++k) { o[k] = LANES(m2[l]->origin[k]); tsv[k] = LANES(Ts[l][12 + k]); ttv[k] = LANES(Tt[l][12 + k]); sp[k] = LANES(m2[l]->sp[k]); sq[k] = LANES(m2[l]->sq[k]); }
  for (int k = 0; k < 6; ++k) { spp[k] = LANES(m2[l]->spp[k]); sqq[k] = LANES(m2[l]->sqq[k]); }
  M3v spq; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) spq(r, c) = LANES(m2[l]->spq[3 * r + c]);
#undef LANES
  v4d cs[3], ct[3], t3[3];
  vmulv(Rs, o, t3); for (int k = 0; k < 3; ++k) cs[k] = t3[k] + tsv[k] - o[k];
  vmulv(Rt, o, t3); for (int k = 0; k < 3; ++k) ct[k] = t3[k] + ttv[k] - o[k];
  v4d Rsp[3], Rtq[3];
  vmulv(Rs, sp, Rsp); vmulv(Rt, sq, Rtq);
  v4d sa[3], sb[3];
  for (int k = 0; k < 3; ++k) { sa[k] = Rsp[k] + n * cs[k]; sb[k] = Rtq[k] + n * ct[k]; }
  M3v Saa = vadd(vadd(vmul(vmul(Rs, vsym6(spp)), vtranspose(Rs)), vouter(Rsp, cs)), vadd_s(vouter(cs, Rsp), vouter(cs, cs), n));
  M3v Sbb = vadd(vadd(vmul(vmul(Rt, vsym6(sqq)), vtranspose(Rt)), vouter(Rtq, ct)), vadd_s(vouter(ct, Rtq), vouter(ct, ct), n));
  M3v Sab = vadd(vadd(vmul(vmul(Rs, spq), vtranspose(Rt)), vouter(Rsp, ct)), vadd_s(vouter(cs, Rtq), vouter(cs, ct), n));
  M3v Sba = vtranspose(Sab);
  v4d sav[3], sdf[3];
  for (int k = 0; k < 3; ++k) { sav[k] = vsplat(0.5) * (sa[k] + sb[k]); sdf[k] = sa[k] - sb[k]; }
  M3v Savav, Savdf;
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
    Savav(r, c) = vsplat(0.25) * (Saa(r, c) + Sab(r, c) + Sba(r, c) + Sbb(r, c));
    Savdf(r, c) = vsplat(0.5) * (Saa(r, c) - Sab(r, c) + Sba(r, c) - Sbb(r, c));
  }
  const v4d tr_dfdf = (Saa(0, 0) + Saa(1, 1) + Saa(2, 2)) - vsplat(2.0) * (Sab(0, 0) + Sab(1, 1) + Sab(2, 2)) +
                      (Sbb(0, 0) + Sbb(1, 1) + Sbb(2, 2));
  v4d Sx[3]; M3v Sxx, Sxd;
  for (int k = 0; k < 3; ++k) Sx[k] = sav[k] + n * o[k];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
    Sxx(r, c) = Savav(r, c) + o[r] * sav[c] + sav[r] * o[c] + n * o[r] * o[c];
    Sxd(r, c) = Savdf(r, c) + o[r] * sdf[c];
  }
  v4d M[36], Z[6], D[6];
  for (int k = 0; k < 36; ++k) M[k] = vsplat(0.0);
#define MV(r, c) M[6 * (r) + (c)]
  MV(0, 4) = -Sx[1]; MV(0, 5) = Sx[2]; MV(1, 3) = -Sx[2]; MV(1, 4) = Sx[0]; MV(2, 3) = Sx[1]; MV(2, 5) = -Sx[0];
  MV(3, 4) = -Sxx(0, 2); MV(3, 5) = -Sxx(0, 1); MV(4, 5) = -Sxx(1, 2);
  MV(3, 3) = Sxx(1, 1) + Sxx(2, 2); MV(4, 4) = Sxx(0, 0) + Sxx(1, 1); MV(5, 5) = Sxx(0, 0) + Sxx(2, 2);
  MV(0, 0) = MV(1, 1) = MV(2, 2) = n;
  for (int r = 0; r < 6; ++r) for (int c = r + 1; c < 6; ++c) MV(c, r) = MV(r, c);
#undef MV
  Z[0] = sdf[0]; Z[1] = sdf[1]; Z[2] = sdf[2];
  Z[3] = Sxd(1, 2) - Sxd(2, 1);
  Z[4] = Sxd(0, 1) - Sxd(1, 0);
  Z[5] = Sxd(2, 0) - Sxd(0, 2);
  if (!solve6_spd_x4(M, Z, D)) return false;
  M3v Cm = vzero3();
  Cm(0, 1) = -D[4]; Cm(0, 2) = D[5]; Cm(1, 0) = D[4]; Cm(1, 2) = -D[3]; Cm(2, 0) = -D[5]; Cm(2, 1) = D[3];
  v4d Co[3], Dt[3];
  vmulv(Cm, o, Co);
  for (int k = 0; k < 3; ++k) Dt[k] = D[k] + Co[k];
  const M3v CtC = vmul(vtranspose(Cm), Cm);
  v4d tr_ctc_savav = vsplat(0.0), tr_c_savdf = vsplat(0.0);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { tr_ctc_savav += CtC(r, c) * Savav(c, r); tr_c_savdf += Cm(r, c) * Savdf(c, r); }
  v4d Csav[3];
  vmulv(Cm, sav, Csav);
  const v4d dt2 = Dt[0] * Dt[0] + Dt[1] * Dt[1] + Dt[2] * Dt[2];
  const v4d dt_sdf = Dt[0] * sdf[0] + Dt[1] * sdf[1] + Dt[2] * sdf[2];
  const v4d dt_csav = Dt[0] * Csav[0] + Dt[1] * Csav[1] + Dt[2] * Csav[2];
  const v4d ssv = tr_dfdf + n * dt2 + tr_ctc_savav - vsplat(2.0) * dt_sdf - vsplat(2.0) * tr_c_savdf + vsplat(2.0) * dt_csav;
  for (int l = 0; l < 4; ++l) {
    for (int k = 0; k < 36; ++k) MM[l][k] = M[k][l];
    for (int k = 0; k < 6; ++k) MZ[l][k] = Z[k][l];
    ss[l] = ssv[l];
  }
  return true;
}
}  // namespace

// test hook: the four-lane computeEdge next to the scalar one (tests/test_host.py compares the bytes)
API int mvr_lum_edge_from_moments_x4(const mvr_pair_moments2_t *m2 /* [4] */, const double *pose_s /* [4][6] */,
                                     const double *pose_t /* [4][6] */, double *MM /* [4][36] */, double *MZ /* [4][6] */, double *ss /* [4] */)
{
  if (!m2 || !pose_s || !pose_t || !MM || !MZ || !ss) return MVR_E_ARG;
  double Ts[4][16], Tt[4][16];
  const mvr_pair_moments2_t *mp[4]; const double *ps[4], *pt[4];
  for (int l = 0; l < 4; ++l) { mvr_pose_to_mat4(pose_s + 6 * l, Ts[l]); mvr_pose_to_mat4(pose_t + 6 * l, Tt[l]); mp[l] = m2 + l; ps[l] = Ts[l]; pt[l] = Tt[l]; }
  double M[4][36], Z[4][6], S[4];
  if (!lum_edge_from_moments_x4(mp, ps, pt, M, Z, S)) return MVR_E_NOCORR;        // a lane needs the scalar function's special paths
  std::memcpy(MM, M, sizeof M); std::memcpy(MZ, Z, sizeof Z); std::memcpy(ss, S, sizeof S);
  return MVR_OK;
}

// LUM::incidenceCorrection (pcl/registration/impl/lum.hpp): H(X) with
//   d(R(theta) p + t)/dX = M(p') H(X),   p' = R p + t,   R = Rx Ry Rz (Borrmann et al.),
// M = [I | ex x p', ez x p', ey x p'] being the matrix LUM::computeEdge builds its sums from (rotational unknowns in
// the order x, z, y).  Rows 3..5: the rotation vector of the angle increments, w = dtx ex + dty Rx ey + dtz Rx Ry ez;
// column 5 of the top block: t x (Rx Ry ez).  (SURVEY recalled that column with the pitch's sin/cos swapped; this is
// the form for which the identity is exact -- tests/test_host.py checks it against a numeric Jacobian.)
static void lum_incidence(const double pose[6], double out[36])
{
  std::memset(out, 0, 36 * sizeof(double));
  for (int k = 0; k < 6; ++k) out[7 * k] = 1.0;
  const double cx = std::cos(pose[3]), sx = std::sin(pose[3]), cy = std::cos(pose[4]), sy = std::sin(pose[4]);
  out[4] = pose[1] * sx - pose[2] * cx;
  out[5] = pose[1] * cx * cy + pose[2] * sx * cy;
  out[6 + 3] = pose[2];
  out[6 + 4] = -pose[0] * sx;
  out[6 + 5] = -pose[0] * cx * cy + pose[2] * sy;
  out[12 + 3] = -pose[1];
  out[12 + 4] = pose[0] * cx;
  out[12 + 5] = -pose[0] * sx * cy - pose[1] * sy;
  out[18 + 5] = sy;
  out[24 + 4] = sx;
  out[24 + 5] = cx * cy;
  out[30 + 4] = cx;
  out[30 + 5] = -sx * cy;
}

API void mvr_lum_incidence(const double pose[6], double H[36]) { lum_incidence(pose, H); }

// x = H(pose)^-1 b without forming H: rows 3..5 of H only couple the three rotational unknowns
//   x3 + sy x5 = b3,   sx x4 + cx cy x5 = b4,   cx x4 - sx cy x5 = b5      (determinant of the 2 x 2: -cy)
// and rows 0..2 are the identity plus the top-right block times those three.  (Eleven general 6 x 6 eliminations
// per LUM iteration were a fifth of its time.)  false: cos(pitch) = 0, H is singular.
static bool solve_incidence(const double pose[6], const PoseTrig &g, const double b[6], double x[6])
{
  const double cx = g.cr, sx = g.sr, cy = g.cp, sy = g.sp;
  if (std::fabs(cy) < 1e-300) return false;
  // [sx, cx cy; cx, -sx cy] (x4, x5)^T = (b4, b5)^T
  const double det = -sx * sx * cy - cx * cx * cy;      // = -cy
  x[4] = (-sx * cy * b[4] - cx * cy * b[5]) / det;
  x[5] = (sx * b[5] - cx * b[4]) / det;
  x[3] = b[3] - sy * x[5];
  x[0] = b[0] - ((pose[1] * sx - pose[2] * cx) * x[4] + (pose[1] * cx * cy + pose[2] * sx * cy) * x[5]);
  x[1] = b[1] - (pose[2] * x[3] + (-pose[0] * sx) * x[4] + (-pose[0] * cx * cy + pose[2] * sy) * x[5]);
  x[2] = b[2] - ((-pose[1]) * x[3] + (pose[0] * cx) * x[4] + (-pose[0] * sx * cy - pose[1] * sy) * x[5]);
  return true;
}

// LUM::compute (App. A.6) on per-edge moments.
API int mvr_lum_compute(int n, int ne, const int *es, const int *et, const mvr_pair_moments2_t *m2,
                        int max_iterations, double threshold, double *poses, int *iters)
{
  if (iters) *iters = 0;
  if (n < 2 || ne < 0 || !es || !et || !m2 || !poses) return MVR_E_ARG;
  for (int e = 0; e < ne; ++e) if (es[e] < 0 || es[e] >= n || et[e] < 0 || et[e] >= n) return MVR_E_ARG;
  const int dim = 6 * (n - 1);
  // Everything that depends on the graph alone -- who is whose neighbour, the structure of G, which solver runs -- and every work
  // array are kept from call to call (per thread): a registration calls this once per pass with the same views and edges, on the
  // critical path between two passes (the GPU waits for the poses), and looking the edges up and allocating a dozen vectors was
  // ~25 us of a 53 us call.  The arithmetic is untouched.
  struct Nbr { int vj, e; double sign; };
  struct Plan {
    int n = -1, ne = -1; std::vector<int> es, et;
    std::vector<double> G, B, cinv, cinvd, Dc, Oc, band, bscratch, Tv;
    std::vector<int> eidx, row_end; std::vector<char> efwd;
    std::vector<std::vector<Nbr>> nbrs;
    std::vector<PoseTrig> trig;
    bool banded = false, chain = false;
  };
  static thread_local Plan plan;
  constexpr int kBandW = 12;
  static const bool force_dense = std::getenv("MVR_LUM_DENSE") != nullptr;       // (tests: the two row-wise routes give the same bits)
  static const bool force_band = std::getenv("MVR_LUM_BAND") != nullptr;         // (tests: the block route agrees with them to rounding)
  const bool same_graph = plan.n == n && plan.ne == ne && std::equal(es, es + ne, plan.es.begin()) && std::equal(et, et + ne, plan.et.begin());
  if (!same_graph) {
    plan.n = n; plan.ne = ne; plan.es.assign(es, es + ne); plan.et.assign(et, et + ne);
    // edge between an (unordered) vertex pair, looked up once: first as (s, t), then as (t, s)
    plan.eidx.assign((size_t)n * n, -1); plan.efwd.assign((size_t)n * n, 0);
    for (int vi = 0; vi < n; ++vi)
      for (int vj = 0; vj < n; ++vj) {
        int e = -1; bool fwd = false;
        for (int k = 0; k < ne && e < 0; ++k) if (es[k] == vi && et[k] == vj) { e = k; fwd = true; }
        for (int k = 0; k < ne && e < 0; ++k) if (es[k] == vj && et[k] == vi) e = k;
        plan.eidx[(size_t)vi * n + vj] = e; plan.efwd[(size_t)vi * n + vj] = fwd;
      }
    // structure of G (the same in every iteration): block row vi - 1 reaches as far as its highest neighbour
    plan.row_end.assign((size_t)dim, 0);
    for (int vi = 1; vi < n; ++vi) {
      int hi = vi;
      for (int vj = 1; vj < n; ++vj) if (plan.eidx[(size_t)vi * n + vj] >= 0) hi = std::max(hi, vj);
      for (int r = 0; r < 6; ++r) plan.row_end[6 * (vi - 1) + r] = 6 * hi;
    }
    // a ring or a chain (every neighbour of a view is the next or the previous one, or the fixed view 0): the upper rows of G
    // reach 12 entries from the diagonal at most -- assembled and factorised on band storage (solve_spd_band)
    int reach = 0;
    for (int j = 0; j < dim; ++j) reach = std::max(reach, plan.row_end[(size_t)j] - j);
    plan.banded = reach <= kBandW && !force_dense;
    plan.chain = plan.banded && !force_band && n >= 2;                            // (reach <= 12: a view's neighbours are the next, the previous and view 0)
    // every view's edges in ascending order of the neighbour (the order all three assemblies add the diagonal blocks up in)
    plan.nbrs.assign((size_t)n, std::vector<Nbr>());
    for (int vi = 1; vi < n; ++vi)
      for (int vj = 0; vj < n; ++vj) {
        const int e = plan.eidx[(size_t)vi * n + vj];
        if (e >= 0) plan.nbrs[(size_t)vi].push_back(Nbr{vj, e, plan.efwd[(size_t)vi * n + vj] ? 1.0 : -1.0});
      }
    plan.B.assign((size_t)dim, 0.0); plan.cinv.assign((size_t)ne * 36, 0.0); plan.cinvd.assign((size_t)ne * 6, 0.0);
    plan.Dc.clear(); plan.Oc.clear();
    if (plan.chain) { plan.Dc.resize((size_t)(n - 1) * 48 + 8); plan.Oc.resize((size_t)std::max(n - 2, 1) * 48 + 8); }      // (blocks of 6 rows of 8: solve_chain6)
    plan.Tv.assign((size_t)n * 16, 0.0); plan.trig.assign((size_t)n, PoseTrig());
  }
  std::vector<double> &G = plan.G, &B = plan.B, &cinv = plan.cinv, &cinvd = plan.cinvd, &Dc = plan.Dc, &Oc = plan.Oc, &band = plan.band, &bscratch = plan.bscratch, &Tv = plan.Tv;
  const std::vector<int> &eidx = plan.eidx, &row_end = plan.row_end; const std::vector<char> &efwd = plan.efwd;
  const std::vector<std::vector<Nbr>> &nbrs = plan.nbrs;
  std::vector<PoseTrig> &trig = plan.trig;
  const bool banded = plan.banded, chain = plan.chain;
  int it = 0;
  for (; it < max_iterations; ++it) {
    for (int v = 0; v < n; ++v) { trig[(size_t)v] = pose_trig(poses + 6 * v); pose_to_mat4_trig(poses + 6 * v, trig[(size_t)v], &Tv[(size_t)v * 16]); }
    auto store_edge = [&](int e, int rc, const double *MM, const double *MZ, double ss) {
      if (rc != MVR_OK || ss < 0.0000000000001 || !std::isfinite(ss)) {
        std::fill(cinv.begin() + 36 * e, cinv.begin() + 36 * (e + 1), 0.0);
        std::fill(cinvd.begin() + 6 * e, cinvd.begin() + 6 * (e + 1), 0.0);
      } else {
        for (int k = 0; k < 36; ++k) cinv[36 * e + k] = MM[k] * (1.0 / ss);
        for (int k = 0; k < 6; ++k) cinvd[6 * e + k] = MZ[k] * (1.0 / ss);
      }
    };
    int e = 0;
    for (; e + 4 <= ne; e += 4) {           // four edges per AVX2 pass (bit-identical to the scalar function)
      const mvr_pair_moments2_t *mp[4]; const double *ps[4], *pt[4];
      for (int l = 0; l < 4; ++l) { mp[l] = &m2[e + l]; ps[l] = &Tv[(size_t)es[e + l] * 16]; pt[l] = &Tv[(size_t)et[e + l] * 16]; }
      double MM4[4][36], MZ4[4][6], ss4[4];
      if (lum_edge_from_moments_x4(mp, ps, pt, MM4, MZ4, ss4)) {
        for (int l = 0; l < 4; ++l) store_edge(e + l, MVR_OK, MM4[l], MZ4[l], ss4[l]);
      } else {
        for (int l = 0; l < 4; ++l) {
          double MM[36], MZ[6], ss;
          const int rc = lum_edge_from_moments_T(mp[l], ps[l], pt[l], MM, MZ, &ss);
          store_edge(e + l, rc, MM, MZ, ss);
        }
      }
    }
    for (; e < ne; ++e) {
      double MM[36], MZ[6], ss;
      const int rc = lum_edge_from_moments_T(&m2[e], &Tv[(size_t)es[e] * 16], &Tv[(size_t)et[e] * 16], MM, MZ, &ss);
      store_edge(e, rc, MM, MZ, ss);
    }
    bool solved = false;
    if (chain) {
      std::fill(Dc.begin(), Dc.end(), 0.0); std::fill(Oc.begin(), Oc.end(), 0.0); std::fill(B.begin(), B.end(), 0.0);
      for (int vi = 1; vi < n; ++vi)
        for (const Nbr &nb : nbrs[(size_t)vi]) {
          const double *ci = &cinv[36 * (size_t)nb.e];
          double *Dv = &Dc[48 * (size_t)(vi - 1)];
          for (int r = 0; r < 6; ++r) for (int k = 0; k < 6; ++k) Dv[8 * r + k] += ci[6 * r + k];
          if (nb.vj == vi + 1) { double *Ov = &Oc[48 * (size_t)(vi - 1)]; for (int r = 0; r < 6; ++r) for (int k = 0; k < 6; ++k) Ov[8 * r + k] = -ci[6 * r + k]; }
          for (int r = 0; r < 6; ++r) B[6 * (vi - 1) + r] += nb.sign * cinvd[6 * nb.e + r];
        }
      solved = solve_chain6(n - 1, Dc.data(), Oc.data(), B.data());
      if (!solved) std::fill(B.begin(), B.end(), 0.0);
    }
    if (!solved && banded) {
      constexpr int S = 2 * kBandW;
      band.assign((size_t)(dim + kBandW) * 2 * kBandW, 0.0); bscratch.resize((size_t)2 * dim + 4 * kBandW);      // (only if this route runs)
      std::fill(B.begin(), B.end(), 0.0);
      for (int vi = 1; vi < n; ++vi)
        for (const Nbr &nb : nbrs[(size_t)vi]) {    // (the order the dense assembly adds the diagonal blocks up in)
          const int vj = nb.vj, e = nb.e;
          const double *ci = &cinv[36 * (size_t)e];
          for (int r = 0; r < 6; ++r) {
            double *row = &band[(size_t)(6 * (vi - 1) + r) * S];
            if (vj > vi) for (int cc = 0; cc < 6; ++cc) row[6 * (vj - vi) + cc - r] = -ci[6 * r + cc];
            for (int cc = r; cc < 6; ++cc) row[cc - r] += ci[6 * r + cc];
            B[6 * (vi - 1) + r] += nb.sign * cinvd[6 * e + r];
          }
        }
      // (a failed factorisation -- a pivot not safely positive -- falls back to the dense route below, which assembles G itself)
      solved = solve_spd_band<kBandW>(dim, band.data(), B.data(), bscratch.data(), bscratch.data() + dim);
      if (!solved) std::fill(B.begin(), B.end(), 0.0);
    }
    if (!solved) {
    G.assign((size_t)dim * dim, 0.0); std::fill(B.begin(), B.end(), 0.0);
    for (int vi = 1; vi < n; ++vi)
      for (int vj = 0; vj < n; ++vj) {
        const int e = eidx[(size_t)vi * n + vj];
        const bool fwd = efwd[(size_t)vi * n + vj] != 0;
        if (e < 0) continue;
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 6; ++c) {
            if (vj > 0) G[(size_t)(6 * (vi - 1) + r) * dim + 6 * (vj - 1) + c] = -cinv[36 * e + 6 * r + c];
            G[(size_t)(6 * (vi - 1) + r) * dim + 6 * (vi - 1) + c] += cinv[36 * e + 6 * r + c];
          }
        for (int r = 0; r < 6; ++r) B[6 * (vi - 1) + r] += (fwd ? 1.0 : -1.0) * cinvd[6 * e + r];
      }
    if (solve_spd(dim, G.data(), B.data(), row_end.data()) != MVR_OK) { if (iters) *iters = it; return MVR_E_SINGULAR; }
    }
    double sum = 0.0;
    for (int vi = 1; vi < n; ++vi) {
      double sol[6], dp[6], nrm = 0.0;
      if (!solve_incidence(poses + 6 * vi, trig[(size_t)vi], &B[6 * (vi - 1)], sol)) continue;       // incidence^-1 * X_vi (the pose is still the one the iteration began with)
      for (int r = 0; r < 6; ++r) { dp[r] = -sol[r]; nrm += sol[r] * sol[r]; }
      sum += std::sqrt(nrm);
      for (int r = 0; r < 6; ++r) poses[6 * vi + r] += dp[r];
    }
    if (sum <= threshold * (double)(n - 1)) { ++it; break; }
  }
  if (iters) *iters = it;
  return MVR_OK;
}

// The whole host side of one global step (registrator.cpp:650-662) in one call:
// per-pair Umeyama + residual from the all-reduced edge table, LUM::compute,
// pose_v <- LUM_v (as Eigen::Affine3f) * pose_v.  rows: ne x 32 doubles
// {n, origin[3], sp, sq, spp, sqq, spq, 0}; poses: n_views x 16 column-major, in/out.
API int mvr_ring_host_step(int n_views, int ne, const int *es, const int *et, const double *rows, const double origin[3],
                           int lum_iterations, double *poses, double *lum_pose /* n_views*6, out */,
                           float *pair_T /* ne*16, out, may be NULL */, double *pair_n /* ne */, double *pair_mse /* ne */,
                           int *lum_iters)
{
  if (n_views < 2 || ne < 0 || !es || !et || !rows || !origin || !poses || !lum_pose) return MVR_E_ARG;
  static thread_local std::vector<mvr_pair_moments2_t> m2;      // (kept from call to call: this runs between two passes, the GPU waiting)
  m2.resize((size_t)ne);
  for (int e = 0; e < ne; ++e) {
    std::memcpy(&m2[e], rows + 32 * (size_t)e, sizeof(mvr_pair_moments2_t));
    for (int k = 0; k < 3; ++k) m2[e].origin[k] = origin[k];     // a constant, not a sum over ranks
    mvr_pair_moments_t pm;
    mvr_moments_from_moments2(&m2[e], &pm);
    if (pair_n) pair_n[e] = pm.n;
    if (pair_mse) pair_mse[e] = pm.mse;
    if (pair_T) {
      float T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      if (pm.n >= 3.0) umeyama_from_moments(pm.mean_src, pm.mean_tgt, pm.sigma, T, nullptr);
      std::memcpy(pair_T + 16 * (size_t)e, T, sizeof T);
    }
  }
  std::fill(lum_pose, lum_pose + 6 * (size_t)n_views, 0.0);
  const int rc = mvr_lum_compute(n_views, ne, es, et, m2.data(), lum_iterations, 0.0, lum_pose, lum_iters);
  if (rc != MVR_OK) return rc;
  for (int v = 1; v < n_views; ++v) {
    double L[16], Lf[16], P[16];
    mvr_pose_to_mat4(lum_pose + 6 * (size_t)v, L);
    for (int k = 0; k < 16; ++k) Lf[k] = (double)(float)L[k];        // lum.getTransformation(i) is an Eigen::Affine3f
    mvr_mat4d_mul(Lf, poses + 16 * (size_t)v, P);
    std::memcpy(poses + 16 * (size_t)v, P, sizeof P);
  }
  return MVR_OK;
}

// min |A x - b|, A rows x 3 (row-major), by Givens rotations: the rows are folded one at a time into a 3 x 4
// triangular system [R | c] (a streaming QR: no storage beyond the triangle, any number of rows).
namespace {
struct Tri3 {
  double r[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  void add(const double a[3], double rhs)
  {
    double row[4] = {a[0], a[1], a[2], rhs};
    for (int k = 0; k < 3; ++k) {
      if (row[k] == 0.0) continue;
      const double h = std::hypot(r[k][k], row[k]), c = r[k][k] / h, s = row[k] / h;
      for (int j = k; j < 4; ++j) {
        const double t = c * r[k][j] + s * row[j];
        row[j] = -s * r[k][j] + c * row[j];
        r[k][j] = t;
      }
    }
  }
  bool solve(double x[3]) const
  {
    for (int k = 2; k >= 0; --k) {
      if (std::fabs(r[k][k]) < 1e-300) return false;
      double acc = r[k][3];
      for (int j = k + 1; j < 3; ++j) acc -= r[k][j] * x[j];
      x[k] = acc / r[k][k];
    }
    return true;
  }
};
}  // namespace

// Registrator::refineAxis (mvr/src/registrator.cpp:402-455; math_solvers::least_squares, math_solvers.cpp:12-38).
// The turntable axis is the common fixed direction of the registered poses, (R_i - I) x = 0, made inhomogeneous by
// the row u + v + w = 1 as the reference does; the pivot their common fixed point, (R_i - I) p = -t_i, with p_y
// pinned to the current pivot's.  Both are 3-unknown least-squares problems: solved by a streaming Givens QR.
API int mvr_refine_axis(int n, const double *poses, float pivot_y, float axis_out[3], float pivot_out[3])
{
  if (n <= 0 || !poses || !axis_out || !pivot_out) return MVR_E_ARG;
  Tri3 ax, pv;
  for (int i = 0; i < n; ++i) {
    const double *P = poses + 16 * (size_t)i;            // column-major [R t]: R(j,k) = P[j + 4 k]
    for (int j = 0; j < 3; ++j) {
      const double row[3] = {P[j] - (j == 0), P[j + 4] - (j == 1), P[j + 8] - (j == 2)};
      ax.add(row, 0.0);
      pv.add(row, -P[12 + j]);
    }
  }
  const double ones[3] = {1, 1, 1}, ey[3] = {0, 1, 0};
  ax.add(ones, 1.0);
  pv.add(ey, (double)pivot_y);
  double a[3], p[3];
  if (!ax.solve(a) || !pv.solve(p)) return MVR_E_SINGULAR;
  // osg::Vec3 (float) semantics of the reference: the solution is stored as floats, the axis normalised in float
  float v[3] = {(float)a[0], (float)a[1], (float)a[2]};
  const float nrm = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (nrm > 0.0f) { const float inv = 1.0f / nrm; v[0] *= inv; v[1] *= inv; v[2] *= inv; }
  for (int k = 0; k < 3; ++k) { axis_out[k] = v[k]; pivot_out[k] = (float)p[k]; }
  return MVR_OK;
}

API double mvr_turntable_angle(int view, int n_views)
{
  // point_cloud.cpp:409: ((view<7)?(-view):(12-view))*M_PI/6, generalised to V views
  const int k = (view <= n_views / 2) ? -view : (n_views - view);
  return (double)k * (2.0 * M_PI / (double)n_views);
}

API void mvr_axis_rotation(const double pivot[3], const double axis[3], double angle, double T[16])
{
  const double n = std::sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
  const double x = axis[0] / n, y = axis[1] / n, z = axis[2] / n, c = std::cos(angle), s = std::sin(angle), k = 1.0 - c;
  const double R[3][3] = {{c + x * x * k, x * y * k - z * s, x * z * k + y * s},
                          {y * x * k + z * s, c + y * y * k, y * z * k - x * s},
                          {z * x * k - y * s, z * y * k + x * s, c + z * z * k}};
  std::memset(T, 0, 16 * sizeof(double));
  for (int r = 0; r < 3; ++r) {
    for (int cc = 0; cc < 3; ++cc) T[r + 4 * cc] = R[r][cc];
    T[12 + r] = pivot[r] - (R[r][0] * pivot[0] + R[r][1] * pivot[1] + R[r][2] * pivot[2]);
  }
  T[15] = 1.0;
}

API void mvr_mat4d_mul(const double A[16], const double B[16], double C[16])
{
  double R[16];
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i) {
      double s = A[i] * B[4 * j];
      s = s + A[i + 4] * B[4 * j + 1];
      s = s + A[i + 8] * B[4 * j + 2];
      s = s + A[i + 12] * B[4 * j + 3];
      R[i + 4 * j] = s;
    }
  std::memcpy(C, R, sizeof R);
}

API void mvr_mat4f_mul(const float A[16], const float B[16], float C[16])
{
  float R[16];
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i) {
      float s = A[i] * B[4 * j];
      s = s + A[i + 4] * B[4 * j + 1];
      s = s + A[i + 8] * B[4 * j + 2];
      s = s + A[i + 12] * B[4 * j + 3];
      R[i + 4 * j] = s;
    }
  std::memcpy(C, R, sizeof R);
}

}  // extern "C"
