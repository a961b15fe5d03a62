// Host-side small dense algebra: the 4x4 pose helpers whose results feed the GPU queries
// (these must round exactly like include/slam6d/globals.icc does), and the closed-form /
// small linear solves of the icp6Dminimizer family, fed from merged pair sums instead of a
// vector<PtPair>.  The solves are microseconds; they stay on the host.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "tdtk_internal.h"

namespace tdtk {

void m4identity(double* M)
{
  for (int k = 0; k < 16; k++) M[k] = (k % 5 == 0) ? 1.0 : 0.0;
}

// globals.icc:501-531
void euler_to_matrix4(const double* rPos, const double* th, double* a)
{
  const double sx = std::sin(th[0]), cx = std::cos(th[0]);
  const double sy = std::sin(th[1]), cy = std::cos(th[1]);
  const double sz = std::sin(th[2]), cz = std::cos(th[2]);
  a[0] = cy * cz;
  a[1] = sx * sy * cz + cx * sz;
  a[2] = -cx * sy * cz + sx * sz;
  a[3] = 0.0;
  a[4] = -cy * sz;
  a[5] = -sx * sy * sz + cx * cz;
  a[6] = cx * sy * sz + sx * cz;
  a[7] = 0.0;
  a[8] = sy;
  a[9] = -sx * cy;
  a[10] = cx * cy;
  a[11] = 0.0;
  a[12] = rPos[0]; a[13] = rPos[1]; a[14] = rPos[2];
  a[15] = 1;
}

// globals.icc:541-576
void matrix4_to_euler(const double* a, double* th, double* rPos)
{
  if (a[0] > 0.0) th[1] = std::asin(a[8]);
  else th[1] = M_PI - std::asin(a[8]);
  const double C = std::cos(th[1]);
  if (std::fabs(C) > 0.005) {
    double trX = a[10] / C, trY = -a[9] / C;
    th[0] = std::atan2(trY, trX);
    trX = a[0] / C; trY = -a[4] / C;
    th[2] = std::atan2(trY, trX);
  } else {
    th[0] = 0.0;
    th[2] = std::atan2(a[1], a[5]);
  }
  if (rPos) { rPos[0] = a[12]; rPos[1] = a[13]; rPos[2] = a[14]; }
}

namespace {
// 3x3 minor of a 4x4 (row r and column c removed), laid out like M4_submat does
// (globals.icc:715-730: out[di*3+dj] = in[si*4+sj]).
inline void minor3(const double* M, int r, int c, double* out)
{
  int k = 0;
  for (int i = 0; i < 4; i++) {
    if (i == r) continue;
    for (int j = 0; j < 4; j++) {
      if (j == c) continue;
      out[k++] = M[i * 4 + j];
    }
  }
}
// association of globals.icc:388-395
inline double det3(const double* m)
{
  return m[0] * (m[4] * m[8] - m[7] * m[5]) - m[1] * (m[3] * m[8] - m[6] * m[5]) +
         m[2] * (m[3] * m[7] - m[6] * m[4]);
}
}  // namespace

// globals.icc:738-785.  The query of every NN search is transform3(M4inv(dalignxf), t); to
// hit the same fp64 bits the cofactor expansion is evaluated in the reference's order:
// det = sum_n (M[n]*det3(minor(0,n)))*(+1,-1,..), out[i+4j] = det3(minor(i,j))*sign/det.
int m4inv(const double* Min, double* Mout)
{
  double sub[9];
  double det = 0.0, s = 1.0;
  for (int n = 0; n < 4; n++, s *= -1.0) {
    minor3(Min, 0, n, sub);
    det += Min[n] * det3(sub) * s;
  }
  if (std::fabs(det) < 0.00000000000005) {
    m4identity(Mout);
    return 0;
  }
  double tmp[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      const int sign = 1 - ((i + j) % 2) * 2;
      minor3(Min, i, j, sub);
      tmp[i + j * 4] = (det3(sub) * sign) / det;
    }
  std::memcpy(Mout, tmp, sizeof tmp);
  return 1;
}

// globals.icc:298-328: Mout[c*4+r] = ((M1[r]*M2[c*4] + M1[4+r]*M2[c*4+1]) + M1[8+r]*M2[c*4+2])
// + M1[12+r]*M2[c*4+3]
void mmult(const double* M1, const double* M2, double* Mout)
{
  double t[16];
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++)
      t[c * 4 + r] = M1[r] * M2[c * 4] + M1[4 + r] * M2[c * 4 + 1] + M1[8 + r] * M2[c * 4 + 2] +
                     M1[12 + r] * M2[c * 4 + 3];
  std::memcpy(Mout, t, sizeof t);
}

// ---------------------------------------------------------------------------------------
// dense helpers
// ---------------------------------------------------------------------------------------
bool invert_dense(int n, const double* A, double* Ainv)
{
  std::vector<double> a(A, A + (size_t)n * n);
  std::vector<double> b((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) b[(size_t)i * n + i] = 1.0;
  for (int c = 0; c < n; c++) {
    int piv = c;
    double best = std::fabs(a[(size_t)c * n + c]);
    for (int r = c + 1; r < n; r++)
      if (std::fabs(a[(size_t)r * n + c]) > best) { best = std::fabs(a[(size_t)r * n + c]); piv = r; }
    if (best == 0.0) return false;
    if (piv != c)
      for (int k = 0; k < n; k++) {
        std::swap(a[(size_t)piv * n + k], a[(size_t)c * n + k]);
        std::swap(b[(size_t)piv * n + k], b[(size_t)c * n + k]);
      }
    const double inv = 1.0 / a[(size_t)c * n + c];
    for (int r = 0; r < n; r++) {
      if (r == c) continue;
      const double f = a[(size_t)r * n + c] * inv;
      if (f == 0.0) continue;
      for (int k = 0; k < n; k++) {
        a[(size_t)r * n + k] -= f * a[(size_t)c * n + k];
        b[(size_t)r * n + k] -= f * b[(size_t)c * n + k];
      }
    }
    for (int k = 0; k < n; k++) {
      a[(size_t)c * n + k] *= inv;
      b[(size_t)c * n + k] *= inv;
    }
  }
  std::memcpy(Ainv, b.data(), sizeof(double) * (size_t)n * n);
  return true;
}

// dot product of two contiguous rows with 8 independent partial sums: legal to vectorise without
// -ffast-math (the association is fixed by the code), ~8x the throughput of the serial chain
static inline double dot8(const double* __restrict__ a, const double* __restrict__ b, int n)
{
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
  int k = 0;
  for (; k + 8 <= n; k += 8) {
    s0 += a[k] * b[k];         s1 += a[k + 1] * b[k + 1];
    s2 += a[k + 2] * b[k + 2]; s3 += a[k + 3] * b[k + 3];
    s4 += a[k + 4] * b[k + 4]; s5 += a[k + 5] * b[k + 5];
    s6 += a[k + 6] * b[k + 6]; s7 += a[k + 7] * b[k + 7];
  }
  double t = 0;
  for (; k < n; k++) t += a[k] * b[k];
  return (((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7))) + t;
}

// Cholesky A = L L^T on a dense row-major copy, row-oriented (every inner product runs over two
// contiguous rows) and envelope-aware: fill-in stays right of each row's first non-zero, so the
// products start there.  The graph-SLAM matrix is a chain of 6x6 blocks plus a few loop-closure
// blocks; its envelope is a small fraction of the square (what CSparse exploits in the reference).
// Fails when a pivot drops below `floor` (the reference's choldc gives up at 1e-7,
// globals.icc:820-848).
static bool cholesky_solve(int n, std::vector<double>& a, const double* b, double* x, double floor)
{
  std::vector<int> first(n);
  for (int i = 0; i < n; i++) {
    int f = 0;
    while (f < i && a[(size_t)i * n + f] == 0.0) ++f;
    first[i] = f;
  }
  for (int i = 0; i < n; i++) {
    double* ri = &a[(size_t)i * n];
    const int fi = first[i];
    for (int j = fi; j < i; j++) {
      const double* rj = &a[(size_t)j * n];
      const int k0 = fi > first[j] ? fi : first[j];
      ri[j] = (ri[j] - (k0 < j ? dot8(ri + k0, rj + k0, j - k0) : 0.0)) / rj[j];
    }
    const double d = ri[i] - dot8(ri + fi, ri + fi, i - fi);
    if (!(d >= floor)) return false;
    ri[i] = std::sqrt(d);
  }
  std::vector<double> y(n);
  for (int i = 0; i < n; i++) {
    const int fi = first[i];
    y[i] = (b[i] - dot8(&a[(size_t)i * n + fi], y.data() + fi, i - fi)) / a[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) x[i] = y[i];
  for (int i = n - 1; i >= 0; i--) {   // back substitution, column sweep over the envelope
    x[i] /= a[(size_t)i * n + i];
    const double xi = x[i];
    const double* ri = &a[(size_t)i * n];
    for (int k = first[i]; k < i; k++) x[k] -= ri[k] * xi;
  }
  return true;
}

// graphSlam6D::solveSparseCholesky(GraphMatrix*, B): entries with |v| <= drop are not
// entered into the sparse matrix (graphSlam6D.cc:495); the system is SPD, so a dense
// Cholesky gives CSparse's answer to rounding.
bool solve_spd_dense(int n, const double* G, const double* B, double* x, double drop)
{
  std::vector<double> a((size_t)n * n);
  for (size_t k = 0; k < (size_t)n * n; k++) a[k] = (std::fabs(G[k]) > drop) ? G[k] : 0.0;
  std::vector<double> b(B, B + n);
  return cholesky_solve(n, a, b.data(), x, 1e-300);
}

// ---------------------------------------------------------------------------------------
// symmetric eigen / SVD by Jacobi rotations (4x4 and 3x3: a handful of sweeps)
// ---------------------------------------------------------------------------------------
template <int N>
static void jacobi_eigen(double A[N][N], double V[N][N], double w[N])
{
  for (int i = 0; i < N; i++)
    for (int j = 0; j < N; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < N; i++) {
      diag += A[i][i] * A[i][i];
      for (int j = i + 1; j < N; j++) off += A[i][j] * A[i][j];
    }
    if (off <= 1e-40 * diag || off == 0.0) break;
    for (int p = 0; p < N; p++)
      for (int q = p + 1; q < N; q++) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = ((theta >= 0) ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < N; k++) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = c * akp - s * akq;
          A[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < N; k++) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = c * apk - s * aqk;
          A[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < N; k++) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < N; i++) w[i] = A[i][i];
}

// one-sided (Hestenes) Jacobi SVD of a 3x3: H = U diag(sv) V^T
static void svd3(const double H[3][3], double U[3][3], double sv[3], double V[3][3])
{
  double A[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { A[i][j] = H[i][j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 64; sweep++) {
    bool rotated = false;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 3; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int k = 0; k < 3; k++) {
          alpha += A[k][p] * A[k][p];
          beta += A[k][q] * A[k][q];
          gamma += A[k][p] * A[k][q];
        }
        if (gamma == 0.0 || std::fabs(gamma) <= 1e-18 * std::sqrt(alpha * beta)) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = ((zeta >= 0) ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < 3; k++) {
          const double ap = A[k][p], aq = A[k][q];
          A[k][p] = c * ap - s * aq;
          A[k][q] = s * ap + c * aq;
          const double vp = V[k][p], vq = V[k][q];
          V[k][p] = c * vp - s * vq;
          V[k][q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  for (int j = 0; j < 3; j++) {
    double nrm = std::sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
    sv[j] = nrm;
    for (int k = 0; k < 3; k++) U[k][j] = (nrm > 0) ? A[k][j] / nrm : 0.0;
  }
  // complete U for a (numerically) rank-deficient H: replace a null column by the cross
  // product of the other two so that U stays orthogonal
  for (int j = 0; j < 3; j++) {
    if (sv[j] > 1e-300) continue;
    const int a = (j + 1) % 3, b = (j + 2) % 3;
    U[0][j] = U[1][a] * U[2][b] - U[2][a] * U[1][b];
    U[1][j] = U[2][a] * U[0][b] - U[0][a] * U[2][b];
    U[2][j] = U[0][a] * U[1][b] - U[1][a] * U[0][b];
  }
}

static inline double det3x3(const double R[3][3])
{
  return R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
         R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
}

// write R (row-major 3x3) and t = cm - R cd into a column-major 4x4
static void compose(const double R[3][3], const double cm[3], const double cd[3], double* xf)
{
  m4identity(xf);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) xf[c * 4 + r] = R[r][c];
  for (int r = 0; r < 3; r++) xf[12 + r] = cm[r] - R[r][0] * cd[0] - R[r][1] * cd[1] - R[r][2] * cd[2];
}

// small-angle rotation of icp6D_APX / icp6D_NAPX (icp6Dapx.cc:277-295): the solved x are
// taken as the sines of the three Euler angles
static void apx_rotation(const double x[3], double R[3][3])
{
  const double sx = x[0], cx = std::sqrt(1.0 - sx * sx);
  const double sy = x[1], cy = std::sqrt(1.0 - sy * sy);
  const double sz = x[2], cz = std::sqrt(1.0 - sz * sz);
  R[0][0] = cy * cz;               R[0][1] = -cy * sz;               R[0][2] = sy;
  R[1][0] = sx * sy * cz + cx * sz; R[1][1] = -sx * sy * sz + cx * cz; R[1][2] = -sx * cy;
  R[2][0] = -cx * sy * cz + sx * sz; R[2][1] = cx * sy * sz + sx * cz; R[2][2] = cx * cy;
}

int align_from_sums(int algo, const tdtk_pair_sums& s, double alignxf[16], double* rms, std::string& err)
{
  m4identity(alignxf);
  if (s.n == 0) { err = "no point pairs"; if (rms) *rms = 0; return TDTK_ESOLVE; }
  const double n = (double)s.n;
  const double* cm = s.centroid_m;
  const double* cd = s.centroid_d;
  double R[3][3];

  if (algo == TDTK_ALGO_QUAT) {
    // Horn 1987.  S[i][j] = (1/n) sum (p2-cd)_i (p1-cm)_j  (icp6Dquat.cc:62-98) = Si[j*3+i]/n
    double S[3][3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) S[i][j] = s.Si[j * 3 + i] / n;
    const double tr = S[0][0] + S[1][1] + S[2][2];
    double Q[4][4];
    Q[0][0] = tr;
    Q[0][1] = Q[1][0] = S[1][2] - S[2][1];
    Q[0][2] = Q[2][0] = S[2][0] - S[0][2];
    Q[0][3] = Q[3][0] = S[0][1] - S[1][0];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Q[i + 1][j + 1] = S[i][j] + S[j][i] - (i == j ? tr : 0.0);
    double V[4][4], w[4];
    jacobi_eigen<4>(Q, V, w);
    int best = 0;
    for (int k = 1; k < 4; k++)
      if (w[k] > w[best]) best = k;
    double q[4] = {V[0][best], V[1][best], V[2][best], V[3][best]};
    const double len = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (double& v : q) v /= len;
    const double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
    R[0][0] = q00 + q11 - q22 - q33;
    R[1][1] = q00 - q11 + q22 - q33;
    R[2][2] = q00 - q11 - q22 + q33;
    R[0][1] = 2.0 * (q[1] * q[2] - q[0] * q[3]);
    R[1][0] = 2.0 * (q[1] * q[2] + q[0] * q[3]);
    R[0][2] = 2.0 * (q[1] * q[3] + q[0] * q[2]);
    R[2][0] = 2.0 * (q[1] * q[3] - q[0] * q[2]);
    R[1][2] = 2.0 * (q[2] * q[3] - q[0] * q[1]);
    R[2][1] = 2.0 * (q[2] * q[3] + q[0] * q[1]);
    compose(R, cm, cd, alignxf);
    if (rms) *rms = std::sqrt(s.sum / n);
    return TDTK_OK;
  }

  if (algo == TDTK_ALGO_SVD) {
    // Arun et al.  H(j,k) = sum (p2-cd)_j (p1-cm)_k (icp6Dsvd.cc:84-90) = Si[k*3+j];
    // R = V U^T, reflection repaired on the weakest singular direction (icp6Dsvd.cc:103-116).
    double H[3][3], U[3][3], V[3][3], sv[3];
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) H[j][k] = s.Si[k * 3 + j];
    svd3(H, U, sv, V);
    auto vut = [&]() {
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) R[r][c] = V[r][0] * U[c][0] + V[r][1] * U[c][1] + V[r][2] * U[c][2];
    };
    vut();
    if (det3x3(R) < 0) {
      int weakest = 0;
      for (int k = 1; k < 3; k++)
        if (sv[k] < sv[weakest]) weakest = k;
      for (int r = 0; r < 3; r++) V[r][weakest] = -V[r][weakest];
      vut();
    }
    compose(R, cm, cd, alignxf);
    if (rms) *rms = std::sqrt(s.sum / n);
    return TDTK_OK;
  }

  if (algo == TDTK_ALGO_APX) {
    if (s.n <= 3) { if (rms) *rms = 0; return TDTK_OK; }  // icp6Dapx.cc:42-46: identity
    std::vector<double> A(9);
    const double* a = s.apx_A;
    A[0] = a[0]; A[1] = a[1]; A[2] = a[2];
    A[3] = a[1]; A[4] = a[3]; A[5] = a[4];
    A[6] = a[2]; A[7] = a[4]; A[8] = a[5];
    double x[3];
    if (!cholesky_solve(3, A, s.apx_B, x, 1.0e-7)) {
      err = "Couldn't find transform.";
      if (rms) *rms = -1.0;
      return TDTK_ESOLVE;
    }
    apx_rotation(x, R);
    compose(R, cm, cd, alignxf);
    if (rms) *rms = std::sqrt(s.sum / n);
    return TDTK_OK;
  }

  if (algo == TDTK_ALGO_NAPX) {
    std::vector<double> A(36);
    int k = 0;
    for (int i = 0; i < 6; i++)
      for (int j = i; j < 6; j++) { A[i * 6 + j] = A[j * 6 + i] = s.napx_A[k++]; }
    double x[6];
    if (!cholesky_solve(6, A, s.napx_B, x, 1.0e-7)) {
      err = "Couldn't find transform.";
      if (rms) *rms = -1.0;
      return TDTK_ESOLVE;
    }
    apx_rotation(x, R);
    // t = x[3..5] + cd - R cd   (icp6Dnapx.cc:140-145)
    const double cmx[3] = {x[3] + cd[0], x[4] + cd[1], x[5] + cd[2]};
    compose(R, cmx, cd, alignxf);
    if (rms) *rms = std::sqrt(s.napx_sum / n);
    return TDTK_OK;
  }

  err = "This minimization algorithm is not implemented";
  return TDTK_EINVAL;
}

}  // namespace tdtk
