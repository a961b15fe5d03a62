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

// factorisation + the two triangular solves on the skyline; cloned per instruction set (resolved once, at load
// time): the dot products are 8 fixed partial sums, so wider registers change the speed, not one bit of the result
__attribute__((target_clones("avx512f", "avx2", "default")))
static bool skyline_factor_solve(int n, const int* first, const size_t* off, double* sky, const double* B, double* y, double* x)
{
  const double floor = 1e-300;
  for (int i = 0; i < n; i++) {
    double* ri = sky + off[i] - first[i];     // ri[k] valid for first[i] <= k <= i
    const int fi = first[i];
    for (int j = fi; j < i; j++) {
      const double* rj = sky + off[j] - first[j];
      const int k0 = fi > first[j] ? fi : first[j];
      ri[j] = (ri[j] - (k0 < j ? dot8(ri + k0, rj + k0, j - k0) : 0.0)) / rj[j];
    }
    const double d = ri[i] - dot8(ri + fi, ri + fi, i - fi);
    if (!(d >= floor)) return false;
    ri[i] = std::sqrt(d);
  }
  for (int i = 0; i < n; i++) {
    const int fi = first[i];
    const double* ri = sky + off[i] - fi;
    y[i] = (B[i] - dot8(ri + fi, y + fi, i - fi)) / ri[i];
  }
  for (int i = n - 1; i >= 0; i--) x[i] = y[i];
  for (int i = n - 1; i >= 0; i--) {   // back substitution, column sweep over the envelope
    const double* ri = sky + off[i] - first[i];
    x[i] /= ri[i];
    const double xi = x[i];
    for (int k = first[i]; k < i; k++) x[k] -= ri[k] * xi;
  }
  return true;
}

// the same factorisation for a caller that has filled the skyline itself (tdtk_lum_assemble_solve)
bool skyline_solve(int n, const int* first, const size_t* off, double* sky, const double* B, double* y, double* x)
{
  return skyline_factor_solve(n, first, off, sky, B, y, x);
}

// graphSlam6D::solveSparseCholesky(GraphMatrix*, B): entries with |v| <= drop are not
// entered into the sparse matrix (graphSlam6D.cc:495); the system is SPD, so a dense
// Cholesky gives CSparse's answer to rounding.
bool solve_spd_dense(int n, const double* G, const double* B, double* x, double drop)
{
  // The factor lives in skyline storage (row i holds columns first[i]..i, where first[i] is the row's first entry
  // that survives the filter): the same products in the same order as cholesky_solve on the filtered dense copy,
  // without writing 2 x n^2 doubles per call -- 63 poses, 84 links: 1.1 MB twice, more time than the factorisation.
  // The buffers are kept per host thread, so a LUM round does not fault in fresh pages every time.
  thread_local std::vector<int> first;
  thread_local std::vector<size_t> off;
  thread_local std::vector<double> sky, y;
  first.resize(n); off.resize((size_t)n + 1);
  size_t total = 0;
  for (int i = 0; i < n; i++) {
    const double* gi = G + (size_t)i * n;
    int f = 0;
    while (f < i && !(std::fabs(gi[f]) > drop)) ++f;
    first[i] = f;
    off[i] = total;
    total += (size_t)(i - f + 1);
  }
  off[n] = total;
  sky.resize(total);
  for (int i = 0; i < n; i++) {
    const double* gi = G + (size_t)i * n;
    double* ri = sky.data() + off[i] - first[i];
    for (int k = first[i]; k <= i; k++) ri[k] = (std::fabs(gi[k]) > drop) ? gi[k] : 0.0;
  }
  y.resize(n);
  return skyline_factor_solve(n, first.data(), off.data(), sky.data(), B, y.data(), x);
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


// ---------------------------------------------------------------------------------------
// The serial-only minimizers (-a 3,4,5,7,8,9).  Each of them is a sum over the pairs of
// products of two linear forms of w = (p1 ; p2), so all of them follow from the second-moment
// matrix of w.  The device accumulates it in shifted form; here it is n, the two centroids, the
// centred cross block Si and the centred diagonal blocks mom_mm / mom_dd (TDTK_WANT_MOM2).
// ---------------------------------------------------------------------------------------
namespace {

struct Mom {
  double n, mu[6], C[6][6];
  explicit Mom(const tdtk_pair_sums& s)
  {
    n = (double)s.n;
    for (int a = 0; a < 3; a++) { mu[a] = s.centroid_m[a]; mu[3 + a] = s.centroid_d[a]; }
    int q = 0;
    for (int a = 0; a < 3; a++)
      for (int b = a; b < 3; b++, q++) {
        C[a][b] = C[b][a] = s.mom_mm[q];
        C[3 + a][3 + b] = C[3 + b][3 + a] = s.mom_dd[q];
      }
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) C[a][3 + b] = C[3 + b][a] = s.Si[a * 3 + b];
  }
  // sum over pairs of (a.w)
  double lin(const double a[6]) const
  {
    double v = 0;
    for (int i = 0; i < 6; i++) v += a[i] * mu[i];
    return n * v;
  }
  // sum over pairs of (a.w)(b.w)
  double quad(const double a[6], const double b[6]) const
  {
    double v = 0, am = 0, bm = 0;
    for (int i = 0; i < 6; i++) {
      double r = 0;
      for (int j = 0; j < 6; j++) r += C[i][j] * b[j];
      v += a[i] * r;
      am += a[i] * mu[i];
      bm += b[i] * mu[i];
    }
    return v + n * am * bm;
  }
};

// solve the n x n system A x = b (what newmat's A.i() * b amounts to)
bool solve_dense(int n, const double* A, const double* b, double* x)
{
  std::vector<double> inv((size_t)n * n);
  if (!invert_dense(n, A, inv.data())) return false;
  for (int i = 0; i < n; i++) {
    double v = 0;
    for (int j = 0; j < n; j++) v += inv[(size_t)i * n + j] * b[j];
    x[i] = v;
  }
  return true;
}

void skew(const double v[3], double S[3][3])
{
  S[0][0] = 0;     S[0][1] = -v[2]; S[0][2] = v[1];
  S[1][0] = v[2];  S[1][1] = 0;     S[1][2] = -v[0];
  S[2][0] = -v[1]; S[2][1] = v[0];  S[2][2] = 0;
}

void rt_to_gl(const double R[3][3], const double t[3], double* xf)
{
  m4identity(xf);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) xf[c * 4 + r] = R[r][c];
  for (int r = 0; r < 3; r++) xf[12 + r] = t[r];
}

// the row-major 4x4 pose matrix of icp6Dlumeuler.cc:144-156 / 184-196
void euler_T(const double x[3], const double th[3], double T[16])
{
  const double cx = std::cos(th[0]), cy = std::cos(th[1]), cz = std::cos(th[2]);
  const double sx = std::sin(th[0]), sy = std::sin(th[1]), sz = std::sin(th[2]);
  for (int i = 0; i < 16; i++) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
  T[3] = x[0]; T[7] = x[1]; T[11] = x[2];
  T[0] = cy * cz;                 T[1] = -cy * sz;                T[2] = sy;
  T[4] = cz * sx * sy + cx * sz;  T[5] = cx * cz - sx * sy * sz;  T[6] = -cy * sx;
  T[8] = sx * sz - cx * cz * sy;  T[9] = cz * sx + cx * sy * sz;  T[10] = cx * cy;
}

// (p*p - q.q) I + 2 q q^T + 2 p [q]x with translation x (icp6Dlumquat.cc:168-180, 190-203), row-major
void quat_T(const double x[3], double p, const double q[3], double T[16])
{
  double Cq[3][3];
  skew(q, Cq);
  const double qq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  for (int i = 0; i < 16; i++) T[i] = 0.0;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T[r * 4 + c] = (r == c ? (p * p - qq) : 0.0) + 2.0 * q[r] * q[c] + 2.0 * p * Cq[r][c];
    T[r * 4 + 3] = x[r];
  }
  T[15] = 1.0;
}

// T_inc = T1 * T2^-1 (row-major 4x4) written out column-major
bool tinc_to_gl(const double T1[16], const double T2[16], double* xf)
{
  double T2i[16], Ti[16];
  if (!invert_dense(4, T2, T2i)) return false;
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) {
      double v = 0;
      for (int k = 0; k < 4; k++) v += T1[r * 4 + k] * T2i[k * 4 + c];
      Ti[r * 4 + c] = v;
    }
  m4identity(xf);
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) xf[c * 4 + r] = Ti[r * 4 + c];
    xf[12 + r] = Ti[r * 4 + 3];
  }
  return true;
}

// Matrix4ToQuat, globals.icc:1032-1075
void matrix4_to_quat(const double* mat, double quat[4], double t[3])
{
  double S, X, Y, Z, W;
  const double T = 1 + mat[0] + mat[5] + mat[10];
  if (T > 0.00000001) {
    S = std::sqrt(T) * 2;
    X = (mat[9] - mat[6]) / S; Y = (mat[2] - mat[8]) / S; Z = (mat[4] - mat[1]) / S; W = 0.25 * S;
  } else if (mat[0] > mat[5] && mat[0] > mat[10]) {
    S = std::sqrt(1.0 + mat[0] - mat[5] - mat[10]) * 2;
    X = 0.25 * S; Y = (mat[4] + mat[1]) / S; Z = (mat[2] + mat[8]) / S; W = (mat[9] - mat[6]) / S;
  } else if (mat[5] > mat[10]) {
    S = std::sqrt(1.0 + mat[5] - mat[0] - mat[10]) * 2;
    X = (mat[4] + mat[1]) / S; Y = 0.25 * S; Z = (mat[9] + mat[6]) / S; W = (mat[2] - mat[8]) / S;
  } else {
    S = std::sqrt(1.0 + mat[10] - mat[0] - mat[5]) * 2;
    X = (mat[2] + mat[8]) / S; Y = (mat[9] + mat[6]) / S; Z = 0.25 * S; W = (mat[4] - mat[1]) / S;
  }
  quat[0] = W; quat[1] = -X; quat[2] = -Y; quat[3] = -Z;
  const double l = std::sqrt(quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2] + quat[3] * quat[3]);
  for (int i = 0; i < 4; i++) quat[i] /= l;
  t[0] = mat[12]; t[1] = mat[13]; t[2] = mat[14];
}

}  // namespace (helpers above are used below through the exported wrappers as well)

// icp6D_HELIX::computeRt (icp6Dhelix.cc:144-206) for one scan's six unknowns
void helix_compute_rt(const double ccs[6], double* alignxf)
{
  double R[3][3];
  const double c[3] = {-ccs[0], -ccs[1], -ccs[2]}, cs[3] = {-ccs[3], -ccs[4], -ccs[5]};
  const double CLength = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
  const double rotationCheck = c[0] * cs[0] + c[1] * cs[1] + c[2] * cs[2];
  const double angle = std::atan(CLength);
  const double g[3] = {c[0] / CLength, c[1] / CLength, c[2] / CLength};
  const double sinAngle = std::sin(-angle / 2);
  const double b0 = std::cos(-angle / 2), b1 = g[0] * sinAngle, b2 = g[1] * sinAngle, b3 = g[2] * sinAngle;
  R[0][0] = b0 * b0 + b1 * b1 - b2 * b2 - b3 * b3; R[0][1] = 2 * (b1 * b2 + b0 * b3); R[0][2] = 2 * (b1 * b3 - b0 * b2);
  R[1][0] = 2 * (b1 * b2 - b0 * b3); R[1][1] = b0 * b0 - b1 * b1 + b2 * b2 - b3 * b3; R[1][2] = 2 * (b2 * b3 + b0 * b1);
  R[2][0] = 2 * (b1 * b3 + b0 * b2); R[2][1] = 2 * (b2 * b3 - b0 * b1); R[2][2] = b0 * b0 - b1 * b1 - b2 * b2 + b3 * b3;
  const double den = b0 * b0 + b1 * b1 + b2 * b2 + b3 * b3;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[i][j] /= den;
  const double skewValue = rotationCheck / (CLength * CLength);
  double gs[3];
  for (int i = 0; i < 3; i++) gs[i] = (cs[i] - c[i] * skewValue) / CLength;
  const double pT[3] = {g[1] * gs[2] - g[2] * gs[1], g[2] * gs[0] - g[0] * gs[2], g[0] * gs[1] - g[1] * gs[0]};
  double t[3];
  for (int i = 0; i < 3; i++)
    t[i] = -(R[i][0] * pT[0] + R[i][1] * pT[1] + R[i][2] * pT[2]) + g[i] * (skewValue * angle) + pT[i];
  m4identity(alignxf);
  for (int r = 0; r < 3; r++) {
    for (int cc = 0; cc < 3; cc++) alignxf[cc * 4 + r] = R[r][cc];
    alignxf[12 + r] = t[r];
  }
}

// Matrix4ToQuat / QuatToMatrix4 (globals.icc:1032-1075, 988-1022)
void matrix4_to_quat_t(const double* mat, double quat[4], double t[3]);
void quat_to_matrix4(const double quat[4], const double t[3], double* mat)
{
  const double q11 = quat[1] * quat[1], q22 = quat[2] * quat[2], q33 = quat[3] * quat[3];
  const double q03 = quat[0] * quat[3], q13 = quat[1] * quat[3], q23 = quat[2] * quat[3];
  const double q02 = quat[0] * quat[2], q12 = quat[1] * quat[2], q01 = quat[0] * quat[1];
  mat[0] = 1 - 2 * (q22 + q33); mat[5] = 1 - 2 * (q11 + q33); mat[10] = 1 - 2 * (q11 + q22);
  mat[4] = 2.0 * (q12 - q03); mat[1] = 2.0 * (q12 + q03);
  mat[8] = 2.0 * (q13 + q02); mat[2] = 2.0 * (q13 - q02);
  mat[9] = 2.0 * (q23 - q01); mat[6] = 2.0 * (q23 + q01);
  mat[3] = mat[7] = mat[11] = 0.0;
  mat[12] = t ? t[0] : 0.0; mat[13] = t ? t[1] : 0.0; mat[14] = t ? t[2] : 0.0;
  mat[15] = 1.0;
}

// icp6D_APX::computeRt (icp6Dapx.cc:310-335): small-angle rotation from the three solved sines + dx
void apx_compute_rt(const double x[3], const double dx[3], double* a)
{
  const double sx = x[0], sy = x[1], sz = x[2];
  const double cx = std::sqrt(1.0 - sx * sx), cy = std::sqrt(1.0 - sy * sy), cz = std::sqrt(1.0 - sz * sz);
  for (int i = 0; i < 16; i++) a[i] = 0.0;
  a[0] = cy * cz; a[1] = sx * sy * cz + cx * sz; a[2] = -cx * sy * cz + sx * sz;
  a[4] = -cy * sz; a[5] = -sx * sy * sz + cx * cz; a[6] = cx * sy * sz + sx * cz;
  a[8] = sy; a[9] = -sx * cy; a[10] = cx * cy;
  a[12] = dx[0]; a[13] = dx[1]; a[14] = dx[2];
  a[15] = 1.0;
}

namespace {

// unit basis forms: e(i) picks w_i; P1 = w[0..2], P2 = w[3..5]
struct Form {
  double a[6];
  Form() { for (double& v : a) v = 0; }
  static Form e(int i, double f = 1.0) { Form r; r.a[i] = f; return r; }
  Form operator+(const Form& o) const { Form r; for (int i = 0; i < 6; i++) r.a[i] = a[i] + o.a[i]; return r; }
  Form operator-(const Form& o) const { Form r; for (int i = 0; i < 6; i++) r.a[i] = a[i] - o.a[i]; return r; }
  Form operator*(double f) const { Form r; for (int i = 0; i < 6; i++) r.a[i] = a[i] * f; return r; }
};

int align_serial_only(int algo, const tdtk_pair_sums& s, const double pose[16], double alignxf[16], std::string& err)
{
  const Mom M(s);
  const double n = M.n;
  const double* cm = s.centroid_m;
  const double* cd = s.centroid_d;
  auto Q = [&](const Form& a, const Form& b) { return M.quad(a.a, b.a); };
  auto L = [&](const Form& a) { return M.lin(a.a); };
  double R[3][3];

  if (algo == TDTK_ALGO_ORTHO) {
    // icp6Dortho.cc:86-121: H = sum m' d'^T, R = H (H^T H)^(-1/2)
    double H[3][3], HH[3][3], V[3][3], w[3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) H[i][j] = s.Si[i * 3 + j];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) HH[i][j] = H[0][i] * H[0][j] + H[1][i] * H[1][j] + H[2][i] * H[2][j];
    jacobi_eigen<3>(HH, V, w);
    double P[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int k = 0; k < 3; k++) {
      if (!(w[k] > 0)) { err = "ORTHO: singular correlation matrix"; return TDTK_ESOLVE; }
      const double f = 1.0 / std::sqrt(w[k]);
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) P[i][j] += V[i][k] * V[j][k] * f;
    }
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) R[i][j] = H[i][0] * P[0][j] + H[i][1] * P[1][j] + H[i][2] * P[2][j];
    compose(R, cm, cd, alignxf);
    return TDTK_OK;
  }

  if (algo == TDTK_ALGO_DUAL) {
    // icp6Ddual.cc:71-121.  X = sum m d^T (raw), cr = sum m x d
    double X[3][3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) X[i][j] = Q(Form::e(i), Form::e(3 + j));
    const double tr = X[0][0] + X[1][1] + X[2][2];
    const double cr[3] = {X[1][2] - X[2][1], X[2][0] - X[0][2], X[0][1] - X[1][0]};
    double C1[4][4] = {}, C2[4][4] = {};
    C1[0][0] = tr;
    for (int i = 0; i < 3; i++) {
      C1[0][1 + i] = -cr[i];
      C1[1 + i][0] = -cr[i];
      for (int j = 0; j < 3; j++) C1[1 + i][1 + j] = X[i][j] + X[j][i] - (i == j ? tr : 0.0);
    }
    double sm[3], sd[3], smd[3];
    for (int i = 0; i < 3; i++) { sm[i] = n * cm[i]; sd[i] = n * cd[i]; smd[i] = sm[i] + sd[i]; }
    double Sk[3][3];
    skew(smd, Sk);
    for (int i = 0; i < 3; i++) {
      C2[0][1 + i] = sm[i] - sd[i];
      C2[1 + i][0] = sd[i] - sm[i];
      for (int j = 0; j < 3; j++) C2[1 + i][1 + j] = -Sk[i][j];
    }
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) { C1[i][j] *= -2.0; C2[i][j] *= 2.0; }
    double A[4][4], V[4][4], w[4];
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) {
        double v = 0;
        for (int k = 0; k < 4; k++) v += C2[k][i] * C2[k][j];
        A[i][j] = (v * 1.0 / (2.0 * n) - C1[i][j] - C1[j][i]) * 0.5;
      }
    jacobi_eigen<4>(A, V, w);
    int best = 0;   // SVD(A): first column of U = the direction of the largest singular value
    for (int k = 1; k < 4; k++)
      if (std::fabs(w[k]) > std::fabs(w[best])) best = k;
    const double qd[4] = {V[0][best], V[1][best], V[2][best], V[3][best]};
    const double q[3] = {qd[1], qd[2], qd[3]};
    double Cq[3][3];
    skew(q, Cq);
    double sv[4];
    for (int i = 0; i < 4; i++) {
      double v = 0;
      for (int k = 0; k < 4; k++) v += C2[i][k] * qd[k];
      sv[i] = v * (-1.0) / (2.0 * n);
    }
    double Qm[4][4];
    Qm[0][0] = qd[0];
    for (int i = 0; i < 3; i++) {
      Qm[0][1 + i] = q[i];
      Qm[1 + i][0] = -q[i];
      for (int j = 0; j < 3; j++) Qm[1 + i][1 + j] = (i == j ? qd[0] : 0.0) + Cq[i][j];
    }
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = Qm[1 + i][0] * sv[0] + Qm[1 + i][1] * sv[1] + Qm[1 + i][2] * sv[2] + Qm[1 + i][3] * sv[3];
    const double qq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) R[i][j] = (i == j ? (qd[0] * qd[0] - qq) : 0.0) + q[i] * q[j] * 2.0 + Cq[i][j] * qd[0] * 2.0;
    rt_to_gl(R, t, alignxf);
    return TDTK_OK;
  }

  if (algo == TDTK_ALGO_HELIX) {
    // icp6Dhelix.cc:69-141
    const Form x2 = Form::e(3), y2 = Form::e(4), z2 = Form::e(5);
    const Form dX = Form::e(3) - Form::e(0), dY = Form::e(4) - Form::e(1), dZ = Form::e(5) - Form::e(2);
    double B[36] = {};
    auto set = [&](int r, int c, double v) { B[r * 6 + c] = B[c * 6 + r] = v; };
    set(3, 3, n); set(4, 4, n); set(5, 5, n);
    set(0, 4, -L(z2)); set(1, 3, L(z2));
    set(0, 5, L(y2));  set(2, 3, -L(y2));
    set(2, 4, L(x2));  set(1, 5, -L(x2));
    set(0, 1, -Q(y2, x2)); set(0, 2, -Q(z2, x2)); set(1, 2, -Q(z2, y2));
    set(0, 0, Q(z2, z2) + Q(y2, y2)); set(1, 1, Q(z2, z2) + Q(x2, x2)); set(2, 2, Q(x2, x2) + Q(y2, y2));
    const double bd[6] = {-Q(z2, dY) + Q(y2, dZ), Q(z2, dX) - Q(x2, dZ), -Q(y2, dX) + Q(x2, dY), L(dX), L(dY), L(dZ)};
    double ccs[6];
    if (!solve_dense(6, B, bd, ccs)) { err = "HELIX: singular system"; return TDTK_ESOLVE; }
    helix_compute_rt(ccs, alignxf);
    return TDTK_OK;
  }

  if (algo == TDTK_ALGO_LUMEULER) {
    // icp6Dlumeuler.cc:47-229
    double rPos[3], rPosTheta[3];
    matrix4_to_euler(pose, rPosTheta, rPos);
    const Form x = (Form::e(0) + Form::e(3)) * 0.5, y = (Form::e(1) + Form::e(4)) * 0.5, z = (Form::e(2) + Form::e(5)) * 0.5;
    const Form dx = Form::e(0) - Form::e(3), dy = Form::e(1) - Form::e(4), dz = Form::e(2) - Form::e(5);
    const double MZ[6] = {L(dx), L(dy), L(dz), -Q(z, dy) + Q(y, dz), -Q(y, dx) + Q(x, dy), Q(z, dx) - Q(x, dz)};
    double MM[36] = {};
    auto set = [&](int r, int c, double v) { MM[r * 6 + c] = MM[c * 6 + r] = v; };
    const double xx = Q(x, x), yy = Q(y, y), zz = Q(z, z);
    set(0, 0, n); set(1, 1, n); set(2, 2, n);
    set(3, 3, yy + zz); set(4, 4, xx + yy); set(5, 5, xx + zz);
    set(0, 4, -L(y)); set(0, 5, L(z));
    set(1, 3, -L(z)); set(1, 4, L(x));
    set(2, 3, L(y));  set(2, 5, -L(x));
    set(3, 4, -Q(x, z)); set(3, 5, -Q(x, y)); set(4, 5, -Q(y, z));
    double Ehat[6];
    if (!solve_dense(6, MM, MZ, Ehat)) { err = "LUMEULER: singular system"; return TDTK_ESOLVE; }
    const double cosx = std::cos(rPosTheta[0]), cosy = std::cos(rPosTheta[1]);
    const double sinx = std::sin(rPosTheta[0]), siny = std::sin(rPosTheta[1]);
    const double tx = rPos[0], ty = rPos[1], tz = rPos[2];
    double T1[16], T2[16];
    euler_T(rPos, rPosTheta, T1);
    double H[36];
    for (int i = 0; i < 36; i++) H[i] = (i % 7 == 0) ? 1.0 : 0.0;
    H[0 * 6 + 4] = -tz * cosx + ty * sinx; H[0 * 6 + 5] = ty * cosx * cosy + tz * cosy * sinx;
    H[1 * 6 + 3] = tz; H[1 * 6 + 4] = -tx * sinx; H[1 * 6 + 5] = -tx * cosx * cosy + tz * siny;
    H[2 * 6 + 3] = -ty; H[2 * 6 + 4] = tx * cosx; H[2 * 6 + 5] = -tx * cosy * sinx - ty * siny;
    H[3 * 6 + 5] = siny; H[4 * 6 + 4] = sinx; H[4 * 6 + 5] = cosx * cosy; H[5 * 6 + 4] = cosx; H[5 * 6 + 5] = -cosy * sinx;
    double HE[6];
    if (!solve_dense(6, H, Ehat, HE)) { err = "LUMEULER: singular pose Jacobian"; return TDTK_ESOLVE; }
    const double X[6] = {rPos[0] - HE[0], rPos[1] - HE[1], rPos[2] - HE[2],
                         rPosTheta[0] - HE[3], rPosTheta[1] - HE[4], rPosTheta[2] - HE[5]};
    euler_T(X, X + 3, T2);
    if (!tinc_to_gl(T1, T2, alignxf)) { err = "LUMEULER: singular pose"; return TDTK_ESOLVE; }
    return TDTK_OK;
  }

  if (algo == TDTK_ALGO_LUMQUAT) {
    // icp6Dlumquat.cc:40-231; x is p1.x, not the midpoint (icp6Dlumquat.cc:90, sic)
    double quat[4], t[3];
    matrix4_to_quat(pose, quat, t);
    const Form x = Form::e(0), y = (Form::e(1) + Form::e(4)) * 0.5, z = (Form::e(2) + Form::e(5)) * 0.5;
    const Form dx = Form::e(0) - Form::e(3), dy = Form::e(1) - Form::e(4), dz = Form::e(2) - Form::e(5);
    const double MZ[7] = {L(dx), L(dy), L(dz), Q(x, dx) + Q(y, dy) + Q(z, dz), Q(z, dy) - Q(y, dz),
                          Q(x, dz) - Q(z, dx), Q(y, dx) - Q(x, dy)};
    double MM[49] = {};
    auto set = [&](int r, int c, double v) { MM[r * 7 + c] = MM[c * 7 + r] = v; };
    const double xx = Q(x, x), yy = Q(y, y), zz = Q(z, z), sx = L(x), sy = L(y), sz = L(z);
    set(0, 0, n); set(1, 1, n); set(2, 2, n);
    set(3, 3, xx + yy + zz); set(4, 4, yy + zz); set(5, 5, xx + zz); set(6, 6, xx + yy);
    set(0, 3, sx); set(0, 5, -sz); set(0, 6, sy);
    set(1, 3, sy); set(1, 4, sz);  set(1, 6, -sx);
    set(2, 3, sz); set(2, 4, -sy); set(2, 5, sx);
    set(4, 5, -Q(x, y)); set(4, 6, -Q(x, z)); set(5, 6, -Q(y, z));
    double Ehat[7];
    if (!solve_dense(7, MM, MZ, Ehat)) { err = "LUMQUAT: singular system"; return TDTK_ESOLVE; }
    const double p = quat[0], q = quat[1], r = quat[2], sq = quat[3];
    const double X0 = t[0], Y0 = t[1], Z0 = t[2];
    const double U[4][4] = {{p, q, r, sq}, {q, -p, sq, -r}, {r, -sq, -p, q}, {sq, r, -q, -p}};
    const double T[3][4] = {
        {p * X0 + sq * Y0 - r * Z0, q * X0 + r * Y0 + sq * Z0, r * X0 - q * Y0 + p * Z0, sq * X0 - p * Y0 - q * Z0},
        {-sq * X0 + p * Y0 + q * Z0, -r * X0 + q * Y0 - p * Z0, q * X0 + r * Y0 + sq * Z0, p * X0 + sq * Y0 - r * Z0},
        {r * X0 - q * Y0 + p * Z0, -sq * X0 + p * Y0 + q * Z0, -p * X0 - sq * Y0 + r * Z0, q * X0 + r * Y0 - sq * Z0}};
    double H[49] = {};
    for (int i = 0; i < 3; i++) {
      H[i * 7 + i] = 1.0;
      for (int j = 0; j < 4; j++) H[i * 7 + 3 + j] = T[i][j] * (-2.0);
    }
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) H[(3 + i) * 7 + 3 + j] = U[i][j] * 2.0;
    double HE[7];
    if (!solve_dense(7, H, Ehat, HE)) { err = "LUMQUAT: singular pose Jacobian"; return TDTK_ESOLVE; }
    const double Xhat[7] = {X0, Y0, Z0, p, q, r, sq};
    double X[7];
    for (int i = 0; i < 7; i++) X[i] = Xhat[i] - HE[i];
    double T1[16], T2[16];
    const double qv1[3] = {q, r, sq};
    quat_T(t, p, qv1, T1);
    quat_T(X, X[3], X + 4, T2);
    if (!tinc_to_gl(T1, T2, alignxf)) { err = "LUMQUAT: singular pose"; return TDTK_ESOLVE; }
    return TDTK_OK;
  }

  if (algo == TDTK_ALGO_QUAT_SCALE) {
    // icp6Dquatscale.cc:37-161: Horn's quaternion + scale = sqrt(sum |m'|^2 / sum |d'|^2)
    double S[3][3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) S[i][j] = s.Si[j * 3 + i] / n;
    const double tr = S[0][0] + S[1][1] + S[2][2];
    double Qm[4][4], V[4][4], w[4];
    Qm[0][0] = tr;
    Qm[0][1] = Qm[1][0] = S[1][2] - S[2][1];
    Qm[0][2] = Qm[2][0] = S[2][0] - S[0][2];
    Qm[0][3] = Qm[3][0] = S[0][1] - S[1][0];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) Qm[i + 1][j + 1] = S[i][j] + S[j][i] - (i == j ? tr : 0.0);
    jacobi_eigen<4>(Qm, V, w);
    int best = 0;
    for (int k = 1; k < 4; k++)
      if (w[k] > w[best]) best = k;
    double q[4] = {V[0][best], V[1][best], V[2][best], V[3][best]};
    const double len = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (double& v : q) v /= len;
    const double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
    R[0][0] = q00 + q11 - q22 - q33; R[1][1] = q00 - q11 + q22 - q33; R[2][2] = q00 - q11 - q22 + q33;
    R[0][1] = 2.0 * (q[1] * q[2] - q[0] * q[3]); R[1][0] = 2.0 * (q[1] * q[2] + q[0] * q[3]);
    R[0][2] = 2.0 * (q[1] * q[3] + q[0] * q[2]); R[2][0] = 2.0 * (q[1] * q[3] - q[0] * q[2]);
    R[1][2] = 2.0 * (q[2] * q[3] - q[0] * q[1]); R[2][1] = 2.0 * (q[2] * q[3] + q[0] * q[1]);
    const double smm = s.mom_mm[0] + s.mom_mm[3] + s.mom_mm[5], sdd = s.mom_dd[0] + s.mom_dd[3] + s.mom_dd[5];
    if (!(sdd > 0)) { err = "QUAT_SCALE: degenerate data cloud"; return TDTK_ESOLVE; }
    const double scale = std::sqrt(smm / sdd);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) R[i][j] *= scale;
    compose(R, cm, cd, alignxf);
    return TDTK_OK;
  }
  err = "This minimization algorithm is not implemented";
  return TDTK_EINVAL;
}

}  // namespace

void matrix4_to_quat_t(const double* mat, double quat[4], double t[3]) { matrix4_to_quat(mat, quat, t); }

int align_from_sums(int algo, const tdtk_pair_sums& s, double alignxf[16], double* rms, std::string& err)
{
  double pose[16];
  std::memcpy(pose, alignxf, sizeof pose);   // LUMEULER / LUMQUAT: the current scan's transMat (icp6D.cc:237-241)
  m4identity(alignxf);
  if (s.n == 0) { err = "no point pairs"; if (rms) *rms = 0; return TDTK_ESOLVE; }
  const double n = (double)s.n;
  if (algo == TDTK_ALGO_ORTHO || algo == TDTK_ALGO_DUAL || algo == TDTK_ALGO_HELIX || algo == TDTK_ALGO_LUMEULER ||
      algo == TDTK_ALGO_LUMQUAT || algo == TDTK_ALGO_QUAT_SCALE) {
    if (rms) *rms = std::sqrt(s.sum / n);
    return align_serial_only(algo, s, pose, alignxf, err);
  }
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
