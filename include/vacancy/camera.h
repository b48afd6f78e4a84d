// Cameras of the carving API.  Right-handed, z forward, y down, x right (OpenCV convention),
// like the reference's include/vacancy/camera.h.  Only what VoxelCarver::Carve reads is kept:
// size, pose (w2c = c2w.inverse(), reference camera.cc:25,39-42) and Project (:131-137, :201-205).
#pragma once

#include <cmath>

#include "vacancy/common.h"

namespace vacancy {

class Camera {
 public:
  Camera() : width_(-1), height_(-1), c2w_(Eigen::Affine3d::Identity()), w2c_(Eigen::Affine3d::Identity()) {}
  Camera(int width, int height) : Camera() { width_ = width; height_ = height; }
  Camera(int width, int height, const Eigen::Affine3d& c2w) : width_(width), height_(height) { set_c2w(c2w); }
  virtual ~Camera() {}
  int width() const { return width_; }
  int height() const { return height_; }
  void set_size(int width, int height) { width_ = width; height_ = height; }
  const Eigen::Affine3d& c2w() const { return c2w_; }
  const Eigen::Affine3d& w2c() const { return w2c_; }
  void set_c2w(const Eigen::Affine3d& c2w) { c2w_ = c2w; w2c_ = c2w_.inverse(); }
  // camera space -> image plane
  // (the only virtual the carve path of the reference calls, voxel_carver.cc:460.  The device evaluates the two
  // projections the reference ships -- PinholeCamera and OrthoCamera, recognised by dynamic_cast -- itself; Carve() with
  // any other subclass fails loudly: `false` + LOGE, never a silently wrong projection)
  virtual void Project(const Eigen::Vector3f& camera_p, Eigen::Vector2f* image_p) const = 0;

 protected:
  int width_, height_;
  Eigen::Affine3d c2w_, w2c_;
};

class PinholeCamera : public Camera {
 public:
  PinholeCamera() : principal_point_(-1, -1), focal_length_(-1, -1) {}
  PinholeCamera(int width, int height) : Camera(width, height), principal_point_(-1, -1), focal_length_(-1, -1) {}
  PinholeCamera(int width, int height, float fov_y_deg) : Camera(width, height) { init_fov(fov_y_deg); }
  PinholeCamera(int width, int height, const Eigen::Affine3d& c2w)
      : Camera(width, height, c2w), principal_point_(-1, -1), focal_length_(-1, -1) {}
  PinholeCamera(int width, int height, const Eigen::Affine3d& c2w, float fov_y_deg)
      : Camera(width, height, c2w) { init_fov(fov_y_deg); }
  PinholeCamera(int width, int height, const Eigen::Affine3d& c2w, const Eigen::Vector2f& principal_point,
                const Eigen::Vector2f& focal_length)
      : Camera(width, height, c2w), principal_point_(principal_point), focal_length_(focal_length) {}
  const Eigen::Vector2f& principal_point() const { return principal_point_; }
  const Eigen::Vector2f& focal_length() const { return focal_length_; }
  void set_principal_point(const Eigen::Vector2f& p) { principal_point_ = p; }
  void set_focal_length(const Eigen::Vector2f& f) { focal_length_ = f; }
  // same focal length per pixel in x and y (reference camera.cc:106-120)
  void set_fov_y(float fov_y_deg) {
    focal_length_[1] = height_ * 0.5f / static_cast<float>(std::tan(radians<float>(fov_y_deg) * 0.5));
    focal_length_[0] = focal_length_[1];
  }
  void set_fov_x(float fov_x_deg) {
    focal_length_[0] = width_ * 0.5f / static_cast<float>(std::tan(radians<float>(fov_x_deg) * 0.5));
    focal_length_[1] = focal_length_[0];
  }
  void Project(const Eigen::Vector3f& p, Eigen::Vector2f* q) const override {
    (*q)[0] = focal_length_[0] / p[2] * p[0] + principal_point_[0];
    (*q)[1] = focal_length_[1] / p[2] * p[1] + principal_point_[1];
  }

 private:
  void init_fov(float fov_y_deg) {
    principal_point_[0] = width_ * 0.5f - 0.5f;
    principal_point_[1] = height_ * 0.5f - 0.5f;
    set_fov_y(fov_y_deg);
  }
  Eigen::Vector2f principal_point_, focal_length_;
};

class OrthoCamera : public Camera {
 public:
  OrthoCamera() {}
  OrthoCamera(int width, int height) : Camera(width, height) {}
  OrthoCamera(int width, int height, const Eigen::Affine3d& c2w) : Camera(width, height, c2w) {}
  void Project(const Eigen::Vector3f& p, Eigen::Vector2f* q) const override { (*q)[0] = p[0]; (*q)[1] = p[1]; }
};

}  // namespace vacancy
