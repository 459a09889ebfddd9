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
FLOPE_HD inline void procrustes3x3(const float* M, float* R) {
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
  int best = 0;
  for (int i = 1; i < 4; ++i)
    if (A[i][i] > A[best][best]) best = i;
  double w = V[0][best], x = V[1][best], y = V[2][best], z = V[3][best];
  const double n = 1.0 / sqrt(w * w + x * x + y * y + z * z);
  w *= n; x *= n; y *= n; z *= n;
  R[0] = (float)(1.0 - 2.0 * (y * y + z * z));  R[1] = (float)(2.0 * (x * y - z * w));  R[2] = (float)(2.0 * (x * z + y * w));
  R[3] = (float)(2.0 * (x * y + z * w));  R[4] = (float)(1.0 - 2.0 * (x * x + z * z));  R[5] = (float)(2.0 * (y * z - x * w));
  R[6] = (float)(2.0 * (x * z - y * w));  R[7] = (float)(2.0 * (y * z + x * w));  R[8] = (float)(1.0 - 2.0 * (x * x + y * y));
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

