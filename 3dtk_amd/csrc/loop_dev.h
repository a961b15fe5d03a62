// Device side of the ICP loop that runs without the host (round 6): the solve every workgroup of a small-batch search launch
// makes before it searches (kernels.hip: loop_prologue), and the same arithmetic run once for tdtk_icp_device_solve (solve.hip).
//
// icp6D::match (icp6D.cc:104-285) on a scan of a few ten thousand reduced points is two short kernels an iteration -- and a
// host round trip between them: the pair sums land in pinned memory, the host wakes, solves Horn's 4x4 eigenproblem
// (icp6Dquat.cc:38-144), launches the next search.  At 15K points that trip is a third of the iteration.  Here:
//   * k_final's reduction of the partial rows, in k_final's association (256 row-striding adders per column, wave sums, a
//     4-entry pass), so the 17 sums are the sums the host path reads, bit for bit;
//   * finish_sums' centroids and centred covariance (api.cpp), Horn's matrix as linalg.cpp builds it;
//   * the largest eigenvalue of that traceless symmetric 4x4 by Newton's iteration on its characteristic polynomial, started
//     from an upper bound (all four roots are real, so Newton from above descends monotonically onto the largest) -- the
//     reference takes the same polynomial and solves it in closed form (icp6Dquat.cc:405-513, Ferrari);
//   * the eigenvector: the best-conditioned column of adj(Q - lambda I) polished by two inverse-iteration steps against
//     (mu I - Q), mu a hair above lambda (symmetric positive definite: LDL^T without pivoting);
//   * rotation from the quaternion, t = cm - R cd, RMS, the reference's stopping rule (icp6D.cc:266-279).
// Rotation and translation agree with linalg.cpp's Jacobi to ~1e-13 (tests: K6 pose 1e-9, dat/ B1 trace 1e-9, pair counts
// equal); a solve that does not come out finite raises ICP_ROW_NEED_HOST and the host finishes that iteration from the row's
// sums with its own solver.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace tdtk {

// The 2x2 sub-determinants of a 4x4 (rows 0,1 -> s, rows 2,3 -> c): determinant and adjugate in ~100 multiplications,
// every index a constant (nothing here may end up in scratch or LDS: one lane runs this, the iteration waits for it).
struct Sym4 { double m00, m01, m02, m03, m11, m12, m13, m22, m23, m33; };
struct Minors { double s0, s1, s2, s3, s4, s5, c0, c1, c2, c3, c4, c5; };
__device__ __forceinline__ Minors minors_of(const Sym4& m)
{
  Minors r;                                   // m10 = m01, m20 = m02, m21 = m12, m30 = m03, m31 = m13, m32 = m23
  r.s0 = m.m00 * m.m11 - m.m01 * m.m01; r.s1 = m.m00 * m.m12 - m.m01 * m.m02; r.s2 = m.m00 * m.m13 - m.m01 * m.m03;
  r.s3 = m.m01 * m.m12 - m.m11 * m.m02; r.s4 = m.m01 * m.m13 - m.m11 * m.m03; r.s5 = m.m02 * m.m13 - m.m12 * m.m03;
  r.c5 = m.m22 * m.m33 - m.m23 * m.m23; r.c4 = m.m12 * m.m33 - m.m13 * m.m23; r.c3 = m.m12 * m.m23 - m.m13 * m.m22;
  r.c2 = m.m02 * m.m33 - m.m03 * m.m23; r.c1 = m.m02 * m.m23 - m.m03 * m.m22; r.c0 = m.m02 * m.m13 - m.m03 * m.m12;
  return r;
}
__device__ __forceinline__ double det_of(const Minors& r)
{
  return r.s0 * r.c5 - r.s1 * r.c4 + r.s2 * r.c3 + r.s3 * r.c2 - r.s4 * r.c1 + r.s5 * r.c0;
}
// adjugate of a symmetric 4x4 (symmetric too): the ten entries
__device__ __forceinline__ Sym4 adj_of(const Sym4& m, const Minors& r)
{
  Sym4 a;
  a.m00 = m.m11 * r.c5 - m.m12 * r.c4 + m.m13 * r.c3;
  a.m01 = -m.m01 * r.c5 + m.m02 * r.c4 - m.m03 * r.c3;
  a.m02 = m.m13 * r.s5 - m.m23 * r.s4 + m.m33 * r.s3;
  a.m03 = -m.m12 * r.s5 + m.m22 * r.s4 - m.m23 * r.s3;
  a.m11 = m.m00 * r.c5 - m.m02 * r.c2 + m.m03 * r.c1;
  a.m12 = -m.m03 * r.s5 + m.m23 * r.s2 - m.m33 * r.s1;
  a.m13 = m.m02 * r.s5 - m.m22 * r.s2 + m.m23 * r.s1;
  a.m22 = m.m03 * r.s4 - m.m13 * r.s2 + m.m33 * r.s0;
  a.m23 = -m.m02 * r.s4 + m.m12 * r.s2 - m.m23 * r.s0;
  a.m33 = m.m02 * r.s3 - m.m12 * r.s1 + m.m22 * r.s0;
  return a;
}

// unit eigenvector of the largest eigenvalue of the symmetric traceless 4x4 Q (Horn's N); false: not finite
__device__ __forceinline__ bool largest_eigenvector(const Sym4& Q, double q[4])
{
  const double f2 = Q.m00 * Q.m00 + Q.m11 * Q.m11 + Q.m22 * Q.m22 + Q.m33 * Q.m33 +
                    2.0 * (Q.m01 * Q.m01 + Q.m02 * Q.m02 + Q.m03 * Q.m03 + Q.m12 * Q.m12 + Q.m13 * Q.m13 + Q.m23 * Q.m23);
  if (!(f2 > 0.0)) {                         // the zero matrix (or a NaN): identity, as Jacobi's V = I gives
    q[0] = 1.0; q[1] = q[2] = q[3] = 0.0;
    return f2 == 0.0;
  }
  const double inv = 1.0 / sqrt(f2);
  Sym4 B;                                    // |B|_F = 1: every eigenvalue in [-1, 1]
  B.m00 = Q.m00 * inv; B.m01 = Q.m01 * inv; B.m02 = Q.m02 * inv; B.m03 = Q.m03 * inv; B.m11 = Q.m11 * inv;
  B.m12 = Q.m12 * inv; B.m13 = Q.m13 * inv; B.m22 = Q.m22 * inv; B.m23 = Q.m23 * inv; B.m33 = Q.m33 * inv;
  // det(x I - B) = x^4 + c2 x^2 + c1 x + c0 (trace 0): c2 = -tr(B^2) / 2, c1 = -tr adj B, c0 = det B
  const Minors mb = minors_of(B);
  const Sym4 ab = adj_of(B, mb);
  const double c0 = det_of(mb), c1 = -(ab.m00 + ab.m11 + ab.m22 + ab.m33);
  const double c2 = -0.5 * (B.m00 * B.m00 + B.m11 * B.m11 + B.m22 * B.m22 + B.m33 * B.m33 +
                            2.0 * (B.m01 * B.m01 + B.m02 * B.m02 + B.m03 * B.m03 + B.m12 * B.m12 + B.m13 * B.m13 + B.m23 * B.m23));
  // Newton from above.  Upper bounds of the largest root: the Frobenius norm (1) and the largest Gershgorin disc edge -- for
  // the small rotations of an ICP iteration B is close to diagonal and the second is a few steps from the root.
  const double g0 = B.m00 + fabs(B.m01) + fabs(B.m02) + fabs(B.m03), g1 = B.m11 + fabs(B.m01) + fabs(B.m12) + fabs(B.m13);
  const double g2 = B.m22 + fabs(B.m02) + fabs(B.m12) + fabs(B.m23), g3 = B.m33 + fabs(B.m03) + fabs(B.m13) + fabs(B.m23);
  double x = fmin(1.0, fmax(fmax(g0, g1), fmax(g2, g3)));
  for (int it = 0; it < 100; it++) {
    const double p = ((x * x + c2) * x + c1) * x + c0;
    const double dp = (4.0 * x * x + 2.0 * c2) * x + c1;
    if (!(dp > 0.0)) break;                  // (at or beyond a multiple largest root)
    const double xn = x - p / dp;
    if (!(xn < x)) break;                    // no further descent: x is the root to rounding
    x = xn;
  }
  if (!(x == x)) return false;
  // start vector: the column of adj(B - x I) = c v v^T with the largest diagonal entry (|v_k| largest)
  Sym4 M = B;
  M.m00 -= x; M.m11 -= x; M.m22 -= x; M.m33 -= x;
  const Sym4 am = adj_of(M, minors_of(M));
  const double d0 = fabs(am.m00), d1 = fabs(am.m11), d2 = fabs(am.m22), d3 = fabs(am.m33);
  double v0, v1, v2, v3;
  if (d0 >= d1 && d0 >= d2 && d0 >= d3) { v0 = am.m00; v1 = am.m01; v2 = am.m02; v3 = am.m03; }
  else if (d1 >= d2 && d1 >= d3) { v0 = am.m01; v1 = am.m11; v2 = am.m12; v3 = am.m13; }
  else if (d2 >= d3) { v0 = am.m02; v1 = am.m12; v2 = am.m22; v3 = am.m23; }
  else { v0 = am.m03; v1 = am.m13; v2 = am.m23; v3 = am.m33; }
  double vn = sqrt(v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3);
  if (!(vn > 0.0)) { v0 = 1.0; v1 = v2 = v3 = 0.0; vn = 1.0; }             // (a multiple root: adj vanishes)
  { const double r = 1.0 / vn; v0 *= r; v1 *= r; v2 *= r; v3 *= r; }
  // two inverse-iteration steps against A = mu I - B, mu = x + 1e-9 (SPD: its smallest eigenvalue is mu - lambda_max > 0),
  // A = L D L^T without pivoting
  const double mu = x + 1e-9;
  const double D0 = mu - B.m00, i0 = 1.0 / D0;
  const double l10 = -B.m01 * i0, l20 = -B.m02 * i0, l30 = -B.m03 * i0;
  const double D1 = (mu - B.m11) - l10 * l10 * D0, i1 = 1.0 / D1;
  const double l21 = (-B.m12 - l20 * l10 * D0) * i1, l31 = (-B.m13 - l30 * l10 * D0) * i1;
  const double D2 = (mu - B.m22) - l20 * l20 * D0 - l21 * l21 * D1, i2 = 1.0 / D2;
  const double l32 = (-B.m23 - l30 * l20 * D0 - l31 * l21 * D1) * i2;
  const double D3 = (mu - B.m33) - l30 * l30 * D0 - l31 * l31 * D1 - l32 * l32 * D2, i3 = 1.0 / D3;
  if (D0 > 0.0 && D1 > 0.0 && D2 > 0.0 && D3 > 0.0) {
#pragma unroll
    for (int step = 0; step < 2; step++) {
      double y0 = v0, y1 = v1 - l10 * y0, y2 = v2 - l20 * y0 - l21 * y1, y3 = v3 - l30 * y0 - l31 * y1 - l32 * y2;
      y0 *= i0; y1 *= i1; y2 *= i2; y3 *= i3;
      y2 -= l32 * y3;
      y1 -= l21 * y2 + l31 * y3;
      y0 -= l10 * y1 + l20 * y2 + l30 * y3;
      const double yn = sqrt(y0 * y0 + y1 * y1 + y2 * y2 + y3 * y3);
      if (!(yn > 0.0) || !(yn < 1e300)) break;
      const double r = 1.0 / yn;
      v0 = y0 * r; v1 = y1 * r; v2 = y2 * r; v3 = y3 * r;
    }
  }
  q[0] = v0; q[1] = v1; q[2] = v2; q[3] = v3;
  return v0 == v0 && v1 == v1 && v2 == v2 && v3 == v3;
}

// the 17 base sums about `shift` -> alignxf, RMS (finish_sums of api.cpp + align_from_sums' QUAT branch of linalg.cpp)
__device__ __forceinline__ bool loop_quat_align(const double* acc, const double shift[3], double xf[16], double& rms)
{
  const double n = acc[ACC_N];
  const double* Sm = acc + ACC_SM;
  const double* Sd = acc + ACC_SD;
  double cm[3], cd[3], Si[9];
  for (int a = 0; a < 3; a++) {
    cm[a] = shift[a] + Sm[a] / n;
    cd[a] = shift[a] + Sd[a] / n;
  }
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) Si[a * 3 + b] = acc[ACC_P + a * 3 + b] - Sm[a] * Sd[b] / n;
  double S[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) S[i][j] = Si[j * 3 + i] / n;
  const double tr = S[0][0] + S[1][1] + S[2][2];
  Sym4 Q;                                    // Horn's N, as linalg.cpp builds it
  Q.m00 = tr;
  Q.m01 = S[1][2] - S[2][1];
  Q.m02 = S[2][0] - S[0][2];
  Q.m03 = S[0][1] - S[1][0];
  Q.m11 = S[0][0] + S[0][0] - tr; Q.m12 = S[0][1] + S[1][0]; Q.m13 = S[0][2] + S[2][0];
  Q.m22 = S[1][1] + S[1][1] - tr; Q.m23 = S[1][2] + S[2][1];
  Q.m33 = S[2][2] + S[2][2] - tr;
  double q[4];
  if (!largest_eigenvector(Q, q)) return false;
  const double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
  double R[3][3];
  R[0][0] = q00 + q11 - q22 - q33;
  R[1][1] = q00 - q11 + q22 - q33;
  R[2][2] = q00 - q11 - q22 + q33;
  R[0][1] = 2.0 * (q[1] * q[2] - q[0] * q[3]);
  R[1][0] = 2.0 * (q[1] * q[2] + q[0] * q[3]);
  R[0][2] = 2.0 * (q[1] * q[3] + q[0] * q[2]);
  R[2][0] = 2.0 * (q[1] * q[3] - q[0] * q[2]);
  R[1][2] = 2.0 * (q[2] * q[3] - q[0] * q[1]);
  R[2][1] = 2.0 * (q[2] * q[3] + q[0] * q[1]);
  for (int k = 0; k < 16; k++) xf[k] = (k % 5 == 0) ? 1.0 : 0.0;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) xf[c * 4 + r] = R[r][c];
  for (int r = 0; r < 3; r++) xf[12 + r] = cm[r] - R[r][0] * cd[0] - R[r][1] * cd[1] - R[r][2] * cd[2];
  rms = sqrt(acc[ACC_SUM] / n);
  bool fin = rms == rms;
  for (int k = 0; k < 16; k++) fin = fin && (fabs(xf[k]) < 1e300);
  return fin;
}


// The rows of pair sums of one launch, stored [column][row] (17 columns of `rows` doubles: what chunk_pair_sums writes inside
// the loop, so that a wave's load here is one stretch of memory), added up by a 256-thread workgroup exactly as k_final's
// workgroup adds up a column: thread t adds rows t, t + 256; wave sums; ((w0 + w1) + w2) + w3 -- the sums the stepped loop
// reads, bit for bit.  rows <= ICP_LOOP_MAX_ROWS = 512, i.e. at most two rows per thread, and all 34 loads are in flight before
// the first is waited for: the addresses are clamped into the array and the loads unconditional (a load behind a per-lane
// condition gets a branch and a wait of its own from the compiler -- seventeen dependent trips, 5 us, measured).
// Result in out[17] (shared memory), valid behind the function's last barrier.
__device__ __forceinline__ void loop_reduce_rows(const double* __restrict__ P, const int rows, double (*red)[ICP_LOOP_COLS], double* out)
{
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const bool h0 = t < rows, h1 = t + 256 < rows;
  const int r0 = h0 ? t : rows - 1, r1 = h1 ? t + 256 : rows - 1;
  double v0[ICP_LOOP_COLS], v1[ICP_LOOP_COLS];
#pragma unroll
  for (int k = 0; k < ICP_LOOP_COLS; k++) {
    v0[k] = P[(size_t)k * rows + r0];
    v1[k] = P[(size_t)k * rows + r1];
  }
  double v[ICP_LOOP_COLS];
#pragma unroll
  for (int k = 0; k < ICP_LOOP_COLS; k++) {
    v[k] = 0.0;                       // (k_final: s = 0; s += row t; s += row t + 256)
    v[k] += h0 ? v0[k] : 0.0;
    if (h1) v[k] += v1[k];
  }
  // the seventeen wave sums step by step side by side: written column by column the compiler makes 17 x 6 dependent LDS
  // permutes of it, each waited for (5 us, measured); this way a step's 34 permutes are in flight together
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double u[ICP_LOOP_COLS];
#pragma unroll
    for (int k = 0; k < ICP_LOOP_COLS; k++) u[k] = __shfl_down(v[k], off, 64);
#pragma unroll
    for (int k = 0; k < ICP_LOOP_COLS; k++) v[k] += u[k];
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < ICP_LOOP_COLS; k++) red[wv][k] = v[k];
  }
  __syncthreads();
  if (t < ICP_LOOP_COLS) out[t] = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
  __syncthreads();
}

// What the host did between two launches: sums -> (status, alignxf, rms) under the loop's history; `iter` = the iteration the
// sums belong to.  Every thread of the workgroup computes the same values.
struct LoopSolve { int status; double rms; double xf[16]; };
__device__ __forceinline__ LoopSolve loop_solve(const double* sums, const double shift[3], const double ret_prev, const double ret_prev_prev,
                                                const double eps, const int iter, const int max_iter)
{
  LoopSolve o;
  o.rms = 0.0;
#pragma unroll
  for (int k = 0; k < 16; k++) o.xf[k] = (k % 5 == 0) ? 1.0 : 0.0;
  o.status = ICP_ROW_CONTINUE;
  if (!((unsigned long long)(sums[ACC_N] + 0.5) > 3ull)) {
    o.status = ICP_ROW_FEW_PAIRS;                         // icp6D.cc:235-243: fewer than four pairs end the loop
  } else if (!loop_quat_align(sums, shift, o.xf, o.rms)) {
    o.status = ICP_ROW_NEED_HOST;
  } else if (fabs(o.rms - ret_prev) < eps && fabs(o.rms - ret_prev_prev) < eps && iter != max_iter - 1) {
    o.status = ICP_ROW_CONVERGED;                         // icp6D.cc:266-279
  } else if (iter == max_iter - 1) {
    o.status = ICP_ROW_LAST;
  }
  return o;
}

// one thread: the iteration's row into the record (pinned host memory; the host watches the READY word)
__device__ __forceinline__ void loop_write_row(IcpLoopDev* lp, const int iter, const LoopSolve& o, const double* sums)
{
  double* row = lp->rows_host + (size_t)(iter % lp->row_cap) * ICP_ROW;
  row[ICP_ROW_N] = sums[ACC_N];
  row[ICP_ROW_RMS] = o.rms;
#pragma unroll
  for (int k = 0; k < 16; k++) row[ICP_ROW_XF + k] = o.xf[k];
  row[ICP_ROW_STATUS] = (double)o.status;
#pragma unroll
  for (int k = 0; k < ICP_LOOP_COLS; k++) row[ICP_ROW_ACC + k] = sums[k];
  __hip_atomic_store(&row[ICP_ROW_READY], (double)(iter + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace tdtk
