// vacancy host API (MI355X build): common includes and helpers.
// Counterpart of the reference's include/vacancy/common.h (radians/degrees :32-47, look-at c2w
// :51-75, zfill :77-82).
#pragma once

#if defined(__has_include)
#if __has_include(<Eigen/Geometry>)
#include <Eigen/Geometry>
#define VACANCY_HAVE_EIGEN 1
#endif
#endif
#ifndef VACANCY_HAVE_EIGEN
#include "vacancy/linalg.h"
#endif

#include <iomanip>
#include <sstream>
#include <string>

#include "vacancy/log.h"

namespace vacancy {

template <typename T>
T radians(T deg) { return deg * static_cast<T>(0.01745329251994329576923690768489); }
template <typename T>
T degrees(T rad) { return rad * static_cast<T>(57.295779513082320876798154814105); }

// camera-to-world pose looking from `position` at `target` (z forward, y down, x right)
inline Eigen::Affine3d c2w(const Eigen::Vector3d& position, const Eigen::Vector3d& target,
                           const Eigen::Vector3d& up) {
  const Eigen::Vector3d zc = (target - position).normalized();
  const Eigen::Vector3d xc = zc.cross(up).normalized();
  const Eigen::Vector3d yc = zc.cross(xc);
  Eigen::Affine3d pose = Eigen::Affine3d::Identity();
#ifdef VACANCY_HAVE_EIGEN
  pose.linear().col(0) = xc; pose.linear().col(1) = yc; pose.linear().col(2) = zc;
#else
  pose.linear().set_col(0, xc); pose.linear().set_col(1, yc); pose.linear().set_col(2, zc);
#endif
  pose.translation() = position;
  return pose;
}

// The reference's two look-at forms (include/vacancy/common.h:51-75): rotation only, and the 4 x 4 pose.
#ifdef VACANCY_HAVE_EIGEN
template <typename T>
void c2w(const Eigen::Matrix<T, 3, 1>& position, const Eigen::Matrix<T, 3, 1>& target,
         const Eigen::Matrix<T, 3, 1>& up, Eigen::Matrix<T, 3, 3>* R) {
  R->col(2) = (target - position).normalized();
  R->col(0) = R->col(2).cross(up).normalized();
  R->col(1) = R->col(2).cross(R->col(0));
}
template <typename T>
void c2w(const Eigen::Matrix<T, 3, 1>& position, const Eigen::Matrix<T, 3, 1>& target,
         const Eigen::Matrix<T, 3, 1>& up, Eigen::Matrix<T, 4, 4>* pose) {
  *pose = Eigen::Matrix<T, 4, 4>::Identity();
  Eigen::Matrix<T, 3, 3> R;
  c2w(position, target, up, &R);
  pose->topLeftCorner(3, 3) = R;
  pose->topRightCorner(3, 1) = position;
}
#else
template <typename T>
void c2w(const Eigen::Vec<T, 3>& position, const Eigen::Vec<T, 3>& target, const Eigen::Vec<T, 3>& up,
         Eigen::Mat3<T>* R) {
  const Eigen::Vec<T, 3> zc = (target - position).normalized();
  const Eigen::Vec<T, 3> xc = zc.cross(up).normalized();
  R->set_col(2, zc);
  R->set_col(0, xc);
  R->set_col(1, zc.cross(xc));
}
template <typename T>
void c2w(const Eigen::Vec<T, 3>& position, const Eigen::Vec<T, 3>& target, const Eigen::Vec<T, 3>& up,
         Eigen::Mat4<T>* pose) {
  *pose = Eigen::Mat4<T>::Identity();
  Eigen::Mat3<T> R;
  c2w(position, target, up, &R);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) (*pose)(i, j) = R(i, j);
    (*pose)(i, 3) = position[i];
  }
}
#endif

template <typename T>
std::string zfill(const T& val, int num = 5) {
  std::ostringstream s;
  s << std::setfill('0') << std::setw(num) << val;
  return s.str();
}

}  // namespace vacancy
