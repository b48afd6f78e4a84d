// Compiled with -fsyntax-only against tests/eigen_decl/Eigen/Geometry (declarations only): every public header
// of the facade in its VACANCY_HAVE_EIGEN form, and the two look-at templates of common.h instantiated.
#include "vacancy/camera.h"
#include "vacancy/common.h"
#include "vacancy/image.h"
#include "vacancy/mesh.h"
#include "vacancy/sharded_voxel_carver.h"
#include "vacancy/voxel_carver.h"

#ifndef VACANCY_HAVE_EIGEN
#error "the Eigen branch of include/vacancy/common.h was not taken"
#endif

// a user's Camera subclass that overrides Project() only -- all the reference's carve path calls (camera.h:39-40)
class UserCamera : public vacancy::Camera {
 public:
  void Project(const Eigen::Vector3f& camera_p, Eigen::Vector2f* image_p) const override {
    (*image_p)[0] = camera_p[0];
    (*image_p)[1] = camera_p[1];
  }
};

void use(const Eigen::Vector3d& p, const Eigen::Vector3d& t, const Eigen::Vector3d& up) {
  Eigen::Affine3d pose = vacancy::c2w(p, t, up);
  Eigen::Matrix<double, 3, 3> R;
  vacancy::c2w(p, t, up, &R);
  Eigen::Matrix<double, 4, 4> T;
  vacancy::c2w(p, t, up, &T);
  vacancy::PinholeCamera cam(640, 480, pose, 45.0f);
  Eigen::Vector2f q;
  cam.Project(Eigen::Vector3f(0.f, 0.f, 1.f), &q);
  vacancy::VoxelCarverOption opt;
  opt.bb_max = Eigen::Vector3f(1.f, 1.f, 1.f);
  vacancy::VoxelCarver carver(opt);
  UserCamera user;
  vacancy::Image1f sdf(4, 4);
  (void)carver.Carve(user, sdf);  // compiles; returns false at run time (unsupported subclass)
}
