// Minimal fixed-size linear algebra with Eigen's spelling, used ONLY when Eigen itself is not on
// the include path (it is an un-vendored submodule of the reference and absent from this image).
// With Eigen available, include/vacancy/common.h includes <Eigen/Geometry> instead and this file
// is not used.  Only what the voxel-carving API touches is provided.
//
// Evaluation orders follow Eigen 3.3/3.4 so that host-side camera arithmetic matches the
// reference bit for bit: 3-term sums associate a0 + (a1 + a2) (redux_novec_unroller),
// Affine inverse = cofactor inverse of the linear part and -(inv * t), Quaternion ->
// rotation without normalisation.
#pragma once

#include <cmath>

namespace Eigen {

template <typename T, int N>
struct Vec {
  T v[N];
  // Like Eigen's fixed-size vectors: NOT initialised.  (std::vector<Vector3f>::resize(n) then touches nothing, and the
  // facade fills an 800 MB voxel mesh from several threads instead of zeroing it first on one.)
  Vec() {}
  static Vec Zero() { Vec r; for (int i = 0; i < N; ++i) r.v[i] = T(0); return r; }
  Vec(T a, T b) { static_assert(N == 2, ""); v[0] = a; v[1] = b; }
  Vec(T a, T b, T c) { static_assert(N == 3, ""); v[0] = a; v[1] = b; v[2] = c; }
  T& operator[](int i) { return v[i]; }
  const T& operator[](int i) const { return v[i]; }
  T& operator()(int i) { return v[i]; }
  const T& operator()(int i) const { return v[i]; }
  T& x() { return v[0]; }
  T& y() { return v[1]; }
  T& z() { static_assert(N >= 3, ""); return v[2]; }
  const T& x() const { return v[0]; }
  const T& y() const { return v[1]; }
  const T& z() const { static_assert(N >= 3, ""); return v[2]; }
  Vec operator+(const Vec& o) const { Vec r; for (int i = 0; i < N; ++i) r.v[i] = v[i] + o.v[i]; return r; }
  Vec operator-(const Vec& o) const { Vec r; for (int i = 0; i < N; ++i) r.v[i] = v[i] - o.v[i]; return r; }
  Vec operator-() const { Vec r; for (int i = 0; i < N; ++i) r.v[i] = -v[i]; return r; }
  Vec operator*(T s) const { Vec r; for (int i = 0; i < N; ++i) r.v[i] = v[i] * s; return r; }
  Vec operator/(T s) const { Vec r; for (int i = 0; i < N; ++i) r.v[i] = v[i] / s; return r; }
  Vec& operator+=(const Vec& o) { for (int i = 0; i < N; ++i) v[i] += o.v[i]; return *this; }
  T squaredNorm() const {
    if (N == 3) return v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]);
    T s = T(0);
    for (int i = 0; i < N; ++i) s += v[i] * v[i];
    return s;
  }
  T norm() const { return std::sqrt(squaredNorm()); }
  Vec normalized() const { T n2 = squaredNorm(); return n2 > T(0) ? (*this) / std::sqrt(n2) : *this; }
  Vec cross(const Vec& b) const {
    static_assert(N == 3, "");
    return Vec(v[1] * b.v[2] - v[2] * b.v[1], v[2] * b.v[0] - v[0] * b.v[2], v[0] * b.v[1] - v[1] * b.v[0]);
  }
  template <typename U>
  Vec<U, N> cast() const { Vec<U, N> r; for (int i = 0; i < N; ++i) r.v[i] = static_cast<U>(v[i]); return r; }
};
template <typename T, int N>
Vec<T, N> operator*(T s, const Vec<T, N>& a) { return a * s; }

template <typename T>
struct Mat3 {
  T m[3][3];
  Mat3() { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = T(0); }
  static Mat3 Identity() { Mat3 r; r.m[0][0] = r.m[1][1] = r.m[2][2] = T(1); return r; }
  T& operator()(int i, int j) { return m[i][j]; }
  const T& operator()(int i, int j) const { return m[i][j]; }
  Vec<T, 3> operator*(const Vec<T, 3>& p) const {
    Vec<T, 3> r;
    for (int i = 0; i < 3; ++i) r[i] = m[i][0] * p[0] + (m[i][1] * p[1] + m[i][2] * p[2]);
    return r;
  }
  Vec<T, 3> col(int j) const { return Vec<T, 3>(m[0][j], m[1][j], m[2][j]); }
  void set_col(int j, const Vec<T, 3>& c) { for (int i = 0; i < 3; ++i) m[i][j] = c[i]; }
  template <typename U>
  Mat3<U> cast() const { Mat3<U> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = static_cast<U>(m[i][j]); return r; }
  // compute_inverse_size3: cofactors, det = c00*m00 + (c10*m10 + c20*m20)
  Mat3 inverse() const {
    auto cof = [&](int i, int j) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      return m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1];
    };
    const T c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
    const T det = c00 * m[0][0] + (c10 * m[1][0] + c20 * m[2][0]);
    const T invdet = T(1) / det;
    Mat3 r;
    r.m[0][0] = c00 * invdet; r.m[0][1] = c10 * invdet; r.m[0][2] = c20 * invdet;
    r.m[1][0] = cof(0, 1) * invdet; r.m[1][1] = cof(1, 1) * invdet; r.m[1][2] = cof(2, 1) * invdet;
    r.m[2][0] = cof(0, 2) * invdet; r.m[2][1] = cof(1, 2) * invdet; r.m[2][2] = cof(2, 2) * invdet;
    return r;
  }
};

// 4 x 4 homogeneous matrix: only what the look-at helper c2w(position, target, up, Matrix4*) fills
template <typename T>
struct Mat4 {
  T m[4][4];
  Mat4() { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m[i][j] = T(0); }
  static Mat4 Identity() { Mat4 r; for (int i = 0; i < 4; ++i) r.m[i][i] = T(1); return r; }
  T& operator()(int i, int j) { return m[i][j]; }
  const T& operator()(int i, int j) const { return m[i][j]; }
};

template <typename T>
struct Affine3 {
  Mat3<T> R;
  Vec<T, 3> t;
  static Affine3 Identity() { Affine3 a; a.R = Mat3<T>::Identity(); a.t = Vec<T, 3>::Zero(); return a; }
  const Mat3<T>& linear() const { return R; }
  Mat3<T>& linear() { return R; }
  const Vec<T, 3>& translation() const { return t; }
  Vec<T, 3>& translation() { return t; }
  Affine3 inverse() const { Affine3 r; r.R = R.inverse(); r.t = -(r.R * t); return r; }
  // res = translation; res += linear * p
  Vec<T, 3> operator*(const Vec<T, 3>& p) const {
    Vec<T, 3> r;
    for (int i = 0; i < 3; ++i) r[i] = t[i] + (R.m[i][0] * p[0] + (R.m[i][1] * p[1] + R.m[i][2] * p[2]));
    return r;
  }
  template <typename U>
  Affine3<U> cast() const { Affine3<U> r; r.R = R.template cast<U>(); r.t = t.template cast<U>(); return r; }
};

struct Quaterniond {
  double qx = 0, qy = 0, qz = 0, qw = 1;
  double& x() { return qx; }
  double& y() { return qy; }
  double& z() { return qz; }
  double& w() { return qw; }
  Mat3<double> toRotationMatrix() const {
    const double tx = 2.0 * qx, ty = 2.0 * qy, tz = 2.0 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    Mat3<double> r;
    r(0, 0) = 1.0 - (tyy + tzz); r(0, 1) = txy - twz; r(0, 2) = txz + twy;
    r(1, 0) = txy + twz; r(1, 1) = 1.0 - (txx + tzz); r(1, 2) = tyz - twx;
    r(2, 0) = txz - twy; r(2, 1) = tyz + twx; r(2, 2) = 1.0 - (txx + tyy);
    return r;
  }
};

struct Translation3d {
  Vec<double, 3> p = Vec<double, 3>::Zero();
  double& x() { return p[0]; }
  double& y() { return p[1]; }
  double& z() { return p[2]; }
};
inline Affine3<double> operator*(const Translation3d& t, const Quaterniond& q) {
  Affine3<double> a;
  a.R = q.toRotationMatrix();
  a.t = t.p;
  return a;
}

typedef Vec<float, 2> Vector2f;
typedef Vec<int, 2> Vector2i;
typedef Vec<float, 3> Vector3f;
typedef Vec<int, 3> Vector3i;
typedef Vec<double, 3> Vector3d;
typedef Mat3<float> Matrix3f;
typedef Mat3<double> Matrix3d;
typedef Mat4<float> Matrix4f;
typedef Mat4<double> Matrix4d;
typedef Affine3<double> Affine3d;
typedef Affine3<float> Affine3f;

}  // namespace Eigen
