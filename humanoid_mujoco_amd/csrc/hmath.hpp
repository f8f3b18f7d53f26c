// hmath.hpp — small fp64 host-side vector/quaternion helpers used by the model compiler and
// set_const.  Conventions follow the reference API (simulation/mujoco/include/mujoco/mujoco.h:
// 1033-1075): quaternions are (w,x,y,z); 3x3 matrices are row-major.
#pragma once
#include <cmath>

namespace hb {
namespace hm {

inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline void cross(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
inline double norm3(const double* a) { return std::sqrt(dot3(a, a)); }
inline double normalize3(double* a) {
  double n = norm3(a);
  if (n < 1e-15) { a[0] = 1; a[1] = 0; a[2] = 0; return n; }
  a[0] /= n; a[1] /= n; a[2] /= n;
  return n;
}
inline void normalize4(double* q) {
  double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < 1e-15) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  for (int i = 0; i < 4; i++) q[i] /= n;
}
inline void mul_quat(double* r, const double* a, const double* b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
inline void quat2mat(double* m, const double* q) {
  double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
  double q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3], q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2 * (q12 - q03); m[2] = 2 * (q13 + q02);
  m[3] = 2 * (q12 + q03); m[5] = 2 * (q23 - q01);
  m[6] = 2 * (q13 - q02); m[7] = 2 * (q23 + q01);
}
inline void rot_vec_quat(double* r, const double* v, const double* q) {
  double m[9];
  quat2mat(m, q);
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2],
         z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
inline void axis_angle2quat(double* q, const double* axis, double angle) {
  double s = std::sin(angle * 0.5);
  q[0] = std::cos(angle * 0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
// rotation matrix (row-major, proper rotation) to quaternion
inline void mat2quat(double* q, const double* m) {
  double tr = m[0] + m[4] + m[8];
  if (tr > 0) {
    double s = std::sqrt(tr + 1.0) * 2;
    q[0] = 0.25 * s; q[1] = (m[7] - m[5]) / s; q[2] = (m[2] - m[6]) / s; q[3] = (m[3] - m[1]) / s;
  } else if (m[0] > m[4] && m[0] > m[8]) {
    double s = std::sqrt(1.0 + m[0] - m[4] - m[8]) * 2;
    q[0] = (m[7] - m[5]) / s; q[1] = 0.25 * s; q[2] = (m[1] + m[3]) / s; q[3] = (m[2] + m[6]) / s;
  } else if (m[4] > m[8]) {
    double s = std::sqrt(1.0 + m[4] - m[0] - m[8]) * 2;
    q[0] = (m[2] - m[6]) / s; q[1] = (m[1] + m[3]) / s; q[2] = 0.25 * s; q[3] = (m[5] + m[7]) / s;
  } else {
    double s = std::sqrt(1.0 + m[8] - m[0] - m[4]) * 2;
    q[0] = (m[3] - m[1]) / s; q[1] = (m[2] + m[6]) / s; q[2] = (m[5] + m[7]) / s; q[3] = 0.25 * s;
  }
  normalize4(q);
}
// quaternion rotating +z onto vec
inline void z2quat(double* q, const double* vec_in) {
  double vec[3] = {vec_in[0], vec_in[1], vec_in[2]};
  normalize3(vec);
  double z[3] = {0, 0, 1}, axis[3];
  cross(axis, z, vec);
  double s = norm3(axis);
  if (s < 1e-10) { axis[0] = 1; axis[1] = 0; axis[2] = 0; }
  else { axis[0] /= s; axis[1] /= s; axis[2] /= s; }
  double ang = std::atan2(s, vec[2]);
  axis_angle2quat(q, axis, ang);
}
// symmetric 3x3 eigen-decomposition by cyclic Jacobi; A = V diag(w) V^T, V columns = eigenvectors
inline void eig3(const double A_in[9], double w[3], double V[9]) {
  double A[9];
  for (int i = 0; i < 9; i++) { A[i] = A_in[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    if (off < 1e-300) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        double apq = A[3 * p + q];
        if (std::fabs(apq) < 1e-300) continue;
        double theta = (A[3 * q + q] - A[3 * p + p]) / (2 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
        double c = 1 / std::sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < 3; k++) {  // A = A*J
          double akp = A[3 * k + p], akq = A[3 * k + q];
          A[3 * k + p] = c * akp - s * akq; A[3 * k + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; k++) {  // A = J^T*A
          double apk = A[3 * p + k], aqk = A[3 * q + k];
          A[3 * p + k] = c * apk - s * aqk; A[3 * q + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; k++) {
          double vkp = V[3 * k + p], vkq = V[3 * k + q];
          V[3 * k + p] = c * vkp - s * vkq; V[3 * k + q] = s * vkp + c * vkq;
        }
      }
  }
  w[0] = A[0]; w[1] = A[4]; w[2] = A[8];
}

}  // namespace hm
}  // namespace hb
