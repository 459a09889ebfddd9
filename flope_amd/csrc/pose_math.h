// Small-matrix pose math shared by the head kernels (device) and the host test
// harness (tests/host_harness): special Procrustes, yaw nullification.
#pragma once
#include <math.h>
#if defined(__HIPCC__)
#define FLOPE_HD __host__ __device__
#else
#define FLOPE_HD
#endif

// ---------------------------------------------------------------------------
// Special orthogonal Procrustes of a 3x3 M:  R = argmax_{R in SO(3)} tr(R^T M)
// = U diag(1,1,det(U V^T)) V^T.  Solved as Horn's quaternion eigenproblem: with
// R(q) quadratic in the unit quaternion q = (w,x,y,z), tr(R(q)^T M) = q^T N q for the
// symmetric 4x4 N below; q is the eigenvector of the largest eigenvalue (cyclic
// Jacobi in fp64).  No SVD, no sign fix-up, det(R) = +1 by construction.  Undefined
// exactly where the reference is (repeated top eigenvalue <=> sigma2 + sigma3 = 0).
FLOPE_HD inline void procrustes3x3_jacobi(const float* M, float* R) {
  double A[4][4], V[4][4];
  const double m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[3], m11 = M[4], m12 = M[5], m20 = M[6], m21 = M[7],
               m22 = M[8];
  A[0][0] = m00 + m11 + m22;  A[1][1] = m00 - m11 - m22;  A[2][2] = -m00 + m11 - m22;  A[3][3] = -m00 - m11 + m22;
  A[0][1] = A[1][0] = m21 - m12;  A[0][2] = A[2][0] = m02 - m20;  A[0][3] = A[3][0] = m10 - m01;
  A[1][2] = A[2][1] = m01 + m10;  A[1][3] = A[3][1] = m02 + m20;  A[2][3] = A[3][2] = m12 + m21;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep) {
    double offn = 0.0, diagn = 0.0;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) (i == j ? diagn : offn) += A[i][j] * A[i][j];
    if (offn <= 1e-30 * diagn || offn == 0.0) break;
    for (int pp = 0; pp < 3; ++pp)
      for (int qq = pp + 1; qq < 4; ++qq) {
        const double apq = A[pp][qq];
        if (apq == 0.0) continue;
        const double theta = (A[qq][qq] - A[pp][pp]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 4; ++k) {             // A <- A J
          const double akp = A[k][pp], akq = A[k][qq];
          A[k][pp] = c * akp - s * akq;
          A[k][qq] = s * akp + c * akq;
        }
        for (int k = 0; k < 4; ++k) {             // A <- J^T A
          const double apk = A[pp][k], aqk = A[qq][k];
          A[pp][k] = c * apk - s * aqk;
          A[qq][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 4; ++k) {             // V <- V J
          const double vkp = V[k][pp], vkq = V[k][qq];
          V[k][pp] = c * vkp - s * vkq;
          V[k][qq] = s * vkp + c * vkq;
        }
      }
  }
  // column of the largest eigenvalue, picked with selects: a runtime index into A / V would put both arrays into scratch memory
  double bv = A[0][0], w = V[0][0], x = V[1][0], y = V[2][0], z = V[3][0];
#pragma unroll
  for (int i = 1; i < 4; ++i) {
    const bool gt = A[i][i] > bv;
    bv = gt ? A[i][i] : bv;
    w = gt ? V[0][i] : w; x = gt ? V[1][i] : x; y = gt ? V[2][i] : y; z = gt ? V[3][i] : z;
  }
  const double n = 1.0 / sqrt(w * w + x * x + y * y + z * z);
  w *= n; x *= n; y *= n; z *= n;
  R[0] = (float)(1.0 - 2.0 * (y * y + z * z));  R[1] = (float)(2.0 * (x * y - z * w));  R[2] = (float)(2.0 * (x * z + y * w));
  R[3] = (float)(2.0 * (x * y + z * w));  R[4] = (float)(1.0 - 2.0 * (x * x + z * z));  R[5] = (float)(2.0 * (y * z - x * w));
  R[6] = (float)(2.0 * (x * z - y * w));  R[7] = (float)(2.0 * (y * z + x * w));  R[8] = (float)(1.0 - 2.0 * (x * x + y * y));
}

// Fast path of the same eigenproblem (the Jacobi solve above costs ~40 us of serial fp64 per
// launch): N is traceless, so its characteristic polynomial is
//     P(l) = l^4 - 2|M|_F^2 l^2 - 8 det(M) l + det(N),
// all roots real, the largest one l* = s1 + s2 + sign(det M) s3 <= sqrt(3)|M|_F.  Newton from that
// upper bound converges monotonically (fp32 steps to get close, fp64 steps to finish); the
// eigenvector is the best-conditioned column of adj(N - l* I) (rank-3 matrix => adj = c q q^T).
// Returns false (caller falls back to Jacobi) when the top eigenvalue is not safely isolated.
FLOPE_HD inline double det3(double a, double b, double c, double d, double e, double f, double g, double h, double i) {
  return a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
}

// minor(R, J) of a 4x4 with compile-time row / column: rows != R, columns != J in ascending order (the order of the runtime
// gather this replaces -- same operands, same det3, same bits)
template <int R, int J> FLOPE_HD inline double minor4(const double (&n)[4][4]) {
  constexpr int a0 = R == 0 ? 1 : 0, a1 = R <= 1 ? 2 : 1, a2 = R <= 2 ? 3 : 2;
  constexpr int b0 = J == 0 ? 1 : 0, b1 = J <= 1 ? 2 : 1, b2 = J <= 2 ? 3 : 2;
  return det3(n[a0][b0], n[a0][b1], n[a0][b2], n[a1][b0], n[a1][b1], n[a1][b2], n[a2][b0], n[a2][b1], n[a2][b2]);
}
template <int R> FLOPE_HD inline void cofactor_row(const double (&n)[4][4], double* q) {
  const double m0 = minor4<R, 0>(n), m1 = minor4<R, 1>(n), m2 = minor4<R, 2>(n), m3 = minor4<R, 3>(n);
  q[0] = ((R + 0) & 1) ? -m0 : m0; q[1] = ((R + 1) & 1) ? -m1 : m1; q[2] = ((R + 2) & 1) ? -m2 : m2; q[3] = ((R + 3) & 1) ? -m3 : m3;
}

FLOPE_HD inline bool procrustes3x3_newton(const float* M, float* R) {
  double m[9], fro2 = 0.0;
  for (int i = 0; i < 9; ++i) { m[i] = M[i]; fro2 += m[i] * m[i]; }
  if (!(fro2 > 1e-60) || !(fro2 < 1e60)) return false;
  const double inv = 1.0 / sqrt(fro2);               // the rotation is scale invariant: work with |M|_F = 1
  for (int i = 0; i < 9; ++i) m[i] *= inv;
  double n[4][4];
  n[0][0] = m[0] + m[4] + m[8];  n[1][1] = m[0] - m[4] - m[8];  n[2][2] = -m[0] + m[4] - m[8];  n[3][3] = -m[0] - m[4] + m[8];
  n[0][1] = n[1][0] = m[7] - m[5];  n[0][2] = n[2][0] = m[2] - m[6];  n[0][3] = n[3][0] = m[3] - m[1];
  n[1][2] = n[2][1] = m[1] + m[3];  n[1][3] = n[3][1] = m[2] + m[6];  n[2][3] = n[3][2] = m[5] + m[7];
  const double detM = det3(m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8]);
  const double c2 = -2.0, c1 = -8.0 * detM;
  const double c0 =
      n[0][0] * det3(n[1][1], n[1][2], n[1][3], n[2][1], n[2][2], n[2][3], n[3][1], n[3][2], n[3][3]) -
      n[0][1] * det3(n[1][0], n[1][2], n[1][3], n[2][0], n[2][2], n[2][3], n[3][0], n[3][2], n[3][3]) +
      n[0][2] * det3(n[1][0], n[1][1], n[1][3], n[2][0], n[2][1], n[2][3], n[3][0], n[3][1], n[3][3]) -
      n[0][3] * det3(n[1][0], n[1][1], n[1][2], n[2][0], n[2][1], n[2][2], n[3][0], n[3][1], n[3][2]);
  float lf = 1.7320509f, c1f = (float)c1, c0f = (float)c0;
  for (int it = 0; it < 10; ++it) {
    const float l2 = lf * lf;
    const float pv = (l2 - 2.f) * l2 + c1f * lf + c0f, dv = (4.f * l2 - 4.f) * lf + c1f;
    if (!(dv > 1e-6f)) break;
    lf -= pv / dv;
  }
  double l = (double)lf * (1.0 + 1e-6) + 1e-9;        // stay on the upper side for the monotone fp64 polish
  bool ok = false;
  double dl_last = 1.0;
  for (int it = 0; it < 8; ++it) {
    const double l2 = l * l;
    const double pv = (l2 + c2) * l2 + c1 * l + c0, dv = (4.0 * l2 + 2.0 * c2) * l + c1;
    if (!(dv > 1e-9)) return false;                    // top eigenvalue (nearly) repeated: gauge-degenerate input
    const double dl = pv / dv;
    l -= dl; dl_last = dl;
    if (fabs(dl) <= 4e-16 * l) { ok = true; break; }
  }
  // r03: 0.37 % of random inputs never met the 2-ulp test -- with P'(l*) ~ 0.1 the rounding noise of P (1e-16) is a step of
  // 1e-15, forever -- and took the Jacobi fallback, whose ~10 us one lane then imposed on its whole launch.  Eight quadratic
  // steps that end on a step <= 1e-13 l have converged to that noise floor (seven orders below the float32 result's own).
  if (!ok && !(fabs(dl_last) <= 1e-13 * l)) return false;
  for (int i = 0; i < 4; ++i) n[i][i] -= l;
  // diagonal cofactors of the rank-3 matrix: pick the row with the largest one
  const double d0 = det3(n[1][1], n[1][2], n[1][3], n[2][1], n[2][2], n[2][3], n[3][1], n[3][2], n[3][3]);
  const double d1 = det3(n[0][0], n[0][2], n[0][3], n[2][0], n[2][2], n[2][3], n[3][0], n[3][2], n[3][3]);
  const double d2 = det3(n[0][0], n[0][1], n[0][3], n[1][0], n[1][1], n[1][3], n[3][0], n[3][1], n[3][3]);
  const double d3 = det3(n[0][0], n[0][1], n[0][2], n[1][0], n[1][1], n[1][2], n[2][0], n[2][1], n[2][2]);
  int r = 0; double best = fabs(d0);
  if (fabs(d1) > best) { best = fabs(d1); r = 1; }
  if (fabs(d2) > best) { best = fabs(d2); r = 2; }
  if (fabs(d3) > best) { best = fabs(d3); r = 3; }
  if (!(best > 1e-7)) return false;                    // eigen-gap too small for the adjugate to be trustworthy
  // cofactor row r of (N - l I): q_j = (-1)^(r+j) minor(r, j).  One statically indexed instance per r (r03: with r as a runtime
  // index the gather of the minors went through scratch memory -- a dozen dependent round trips, most of this function's 15 us)
  double q[4];
  switch (r) {
    case 0: cofactor_row<0>(n, q); break;
    case 1: cofactor_row<1>(n, q); break;
    case 2: cofactor_row<2>(n, q); break;
    default: cofactor_row<3>(n, q); break;
  }
  double w = q[0], x = q[1], y = q[2], z = q[3];
  const double nn = 1.0 / sqrt(w * w + x * x + y * y + z * z);
  w *= nn; x *= nn; y *= nn; z *= nn;
  R[0] = (float)(1.0 - 2.0 * (y * y + z * z));  R[1] = (float)(2.0 * (x * y - z * w));  R[2] = (float)(2.0 * (x * z + y * w));
  R[3] = (float)(2.0 * (x * y + z * w));  R[4] = (float)(1.0 - 2.0 * (x * x + z * z));  R[5] = (float)(2.0 * (y * z - x * w));
  R[6] = (float)(2.0 * (x * z - y * w));  R[7] = (float)(2.0 * (y * z + x * w));  R[8] = (float)(1.0 - 2.0 * (x * x + y * y));
  return true;
}

FLOPE_HD inline void procrustes3x3(const float* M, float* R) {
  if (!procrustes3x3_newton(M, R)) procrustes3x3_jacobi(M, R);
}

// nullify_yaw (mvg.py:240-251 with scipy extrinsic 'zyx'): zeroing the first Euler angle
// equals R' = R * Rz(a)^T with a = atan2(-R01, R00): column 2 is
// kept, R'01 = 0.
FLOPE_HD inline void nullify_yaw3x3(const float* R, float* O) {
  const double c0 = R[0], s0 = -R[1];
  double n = sqrt(c0 * c0 + s0 * s0);
  double c = 1.0, s = 0.0;
  if (n > 0.0) { c = c0 / n; s = s0 / n; }
  // Rz(a)^T = [[c, s, 0], [-s, c, 0], [0, 0, 1]]
  for (int i = 0; i < 3; ++i) {
    const double a = R[i * 3], b = R[i * 3 + 1];
    O[i * 3] = (float)(a * c - b * s);
    O[i * 3 + 1] = (float)(a * s + b * c);
    O[i * 3 + 2] = R[i * 3 + 2];
  }
}

