// vacancy::VoxelCarver on MI355X.  Same public API as the reference's
// include/vacancy/voxel_carver.h (option structs :20-60, class :95-118, free functions :120-128);
// the voxel grid lives in HBM behind the C-ABI of vacancy_hip.h instead of a std::vector<Voxel>.
#pragma once

#include <memory>
#include <vector>

#include "vacancy/camera.h"
#include "vacancy/common.h"
#include "vacancy/image.h"
#include "vacancy/mesh.h"

namespace vacancy {

enum class VoxelUpdate { kMax = 0, kWeightedAverage = 1 };
enum class SdfInterpolation { kNn = 0, kBilinear = 1 };
enum class UpdateOutsideImage { kNone = 0, kMax = 1 };

struct InvalidSdf {
  static const float kVal;  // std::numeric_limits<float>::lowest()
};

struct VoxelUpdateOption {
  VoxelUpdate voxel_update{VoxelUpdate::kMax};
  SdfInterpolation sdf_interp{SdfInterpolation::kBilinear};
  UpdateOutsideImage update_outside{UpdateOutsideImage::kNone};
  int voxel_max_update_num{255};
  float voxel_update_weight{1.0f};
  bool use_truncation{false};
  float truncation_band{0.1f};
};

struct VoxelCarverOption {
  Eigen::Vector3f bb_max = Eigen::Vector3f(0.0f, 0.0f, 0.0f);  // (the reference leaves both uninitialised)
  Eigen::Vector3f bb_min = Eigen::Vector3f(0.0f, 0.0f, 0.0f);
  float resolution{0.1f};
  bool sdf_minmax_normalize{true};
  VoxelUpdateOption update_option;
};

// One voxel as the reference stores it (include/vacancy/voxel_carver.h:62-72).  On the device the grid is
// a structure of arrays (vacancy_hip.h); this is the host-side view a VoxelGrid snapshot hands out.
struct Voxel {
  Eigen::Vector3i index{-1, -1, -1};      // voxel index
  int id{-1};
  Eigen::Vector3f pos{0.0f, 0.0f, 0.0f};  // center of voxel
  float sdf{0.0f};                        // Signed Distance Function (SDF) value
  int update_num{0};
  bool outside{false};
  bool on_surface{false};
  Voxel();
  ~Voxel();
};

// The reference's VoxelGrid (:74-93) as a HOST container: Init() lays out the voxels exactly like
// VoxelGrid::Init (voxel_carver.cc:276-345); VoxelCarver::Download(VoxelGrid*) fills sdf / update_num
// from the device-resident grid.  Carving never touches it.
class VoxelGrid {
 public:
  VoxelGrid();
  ~VoxelGrid();
  bool Init(const Eigen::Vector3f& bb_max, const Eigen::Vector3f& bb_min, float resolution);
  const Eigen::Vector3i& voxel_num() const;
  const Voxel& get(int x, int y, int z) const;
  Voxel* get_ptr(int x, int y, int z);
  float resolution() const;
  void ResetOnSurface();
  bool initialized() const;

 private:
  std::vector<Voxel> voxels_;
  Eigen::Vector3f bb_max_;
  Eigen::Vector3f bb_min_;
  float resolution_{-1.0f};
  Eigen::Vector3i voxel_num_{0, 0, 0};
  int xy_slice_num_{0};
};

class VoxelCarver {
 public:
  VoxelCarver();
  explicit VoxelCarver(VoxelCarverOption option);
  ~VoxelCarver();
  VoxelCarver(const VoxelCarver&) = delete;
  VoxelCarver& operator=(const VoxelCarver&) = delete;

  void set_option(VoxelCarverOption option);
  void set_device(int device_id);  // default 0
  bool Init();
  bool Carve(const Camera& camera, const Image1b& silhouette, const Eigen::Vector2i& roi_min,
             const Eigen::Vector2i& roi_max, Image1f* sdf);
  bool Carve(const Camera& camera, const Eigen::Vector2i& roi_min, const Eigen::Vector2i& roi_max,
             const Image1f& sdf);
  bool Carve(const Camera& camera, const Image1b& silhouette, Image1f* sdf);
  bool Carve(const Camera& camera, const Image1b& silhouette);
  bool Carve(const Camera& camera, const Image1f& sdf);
  // All views in one fused pass over the grid.  The first form is the reference's signature
  // (voxel_carver.h:113); Camera is abstract there as here, so callers hold cameras by pointer --
  // examples.cc:108-128 keeps std::shared_ptr<Camera> -- and the other two forms take those directly.
  bool Carve(const std::vector<Camera>& cameras, const std::vector<Image1b>& silhouettes);
  bool Carve(const std::vector<std::shared_ptr<Camera>>& cameras, const std::vector<Image1b>& silhouettes);
  bool Carve(const std::vector<const Camera*>& cameras, const std::vector<Image1b>& silhouettes);
  void ExtractVoxel(Mesh* mesh, bool inside_empty = false);
  void ExtractIsoSurface(Mesh* mesh, double iso_level = 0.0, bool linear_interp = true);

  // grid access for host-side consumers: global dims and the voxel state in id order
  Eigen::Vector3i voxel_num() const;
  bool Download(std::vector<float>* sdf, std::vector<int>* update_num) const;
  // host snapshot of the grid in the reference's own types: grid->Init(option) + sdf / update_num of every voxel
  bool Download(VoxelGrid* grid) const;

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

void DistanceTransformL1(const Image1b& mask, const Eigen::Vector2i& roi_min, const Eigen::Vector2i& roi_max,
                         Image1f* dist);
void MakeSignedDistanceField(const Image1b& mask, const Eigen::Vector2i& roi_min, const Eigen::Vector2i& roi_max,
                             Image1f* dist, bool minmax_normalize, bool use_truncation, float truncation_band);
void SignedDistance2Color(const Image1f& sdf, Image3b* vis_sdf, float min_negative_d, float max_positive_d);

}  // namespace vacancy
