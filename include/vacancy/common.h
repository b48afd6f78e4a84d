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

template <typename T>
std::string zfill(const T& val, int num = 5) {
  std::ostringstream s;
  s << std::setfill('0') << std::setw(num) << val;
  return s.str();
}

}  // namespace vacancy
