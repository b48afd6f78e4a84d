// The reference's demo (examples.cc:75-152) on the MI355X path: 6 masks + TUM poses of data/
// bunny, 10 mm voxels; per view carve -> marching cubes (interpolated and not), PLY out.
// Usage: bunny <data_dir> <out_dir> [resolution] [n_slabs]
//   n_slabs > 0 additionally runs the same views through ShardedVoxelCarver (that many z-slabs on
//   device 0) and checks that its stitched mesh is identical to the single-context one.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "vacancy/sharded_voxel_carver.h"
#include "vacancy/voxel_carver.h"

// TUM trajectory line: id tx ty tz qx qy qz qw -> pose = Translation * Quaternion (examples.cc:36-50)
static bool LoadTumPoses(const std::string& path, std::vector<Eigen::Affine3d>* poses) {
  std::ifstream in(path);
  std::string line;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    std::vector<std::string> tok;
    for (std::string t; std::getline(ss, t, ' ');) tok.push_back(t);
    if (tok.size() != 8) {
      vacancy::LOGE("wrong tum format\n");
      return false;
    }
    Eigen::Translation3d t;
    t.x() = std::atof(tok[1].c_str());
    t.y() = std::atof(tok[2].c_str());
    t.z() = std::atof(tok[3].c_str());
    Eigen::Quaterniond q;
    q.x() = std::atof(tok[4].c_str());
    q.y() = std::atof(tok[5].c_str());
    q.z() = std::atof(tok[6].c_str());
    q.w() = std::atof(tok[7].c_str());
    poses->push_back(t * q);
  }
  return !poses->empty();
}

int main(int argc, char* argv[]) {
  const std::string data_dir = argc > 1 ? argv[1] : "../data/";
  const std::string out_dir = argc > 2 ? argv[2] : data_dir;
  const float resolution = argc > 3 ? (float)std::atof(argv[3]) : 10.0f;
  const int n_slabs = argc > 4 ? std::atoi(argv[4]) : 0;
  const bool planned = argc > 5 && std::string(argv[5]) == "planned";
  std::vector<Eigen::Affine3d> poses;
  if (!LoadTumPoses(data_dir + "/tumpose.txt", &poses)) return 1;

  vacancy::VoxelCarverOption option;
  option.bb_min = Eigen::Vector3f(-250.000000f, -344.586151f, -129.982697f);
  option.bb_max = Eigen::Vector3f(250.000000f, 150.542343f, 257.329224f);
  const float bb_offset = 20.0f;  // keep the boundary clean
  for (int i = 0; i < 3; ++i) {
    option.bb_min[i] -= bb_offset;
    option.bb_max[i] += bb_offset;
  }
  option.resolution = resolution;
  vacancy::VoxelCarver carver(option);
  if (!carver.Init()) return 2;

  std::unique_ptr<vacancy::ShardedVoxelCarver> sharded;
  if (n_slabs > 0) {
    sharded.reset(new vacancy::ShardedVoxelCarver(option, {0}, n_slabs));
    if (planned) {  // cuts of equal predicted cost for the six views about to be carved (vcy_plan_z_slabs)
      std::vector<std::shared_ptr<vacancy::PinholeCamera>> cams;
      std::vector<const vacancy::Camera*> cam_ptrs;
      std::vector<vacancy::Image1b> sils(poses.size() < 6 ? poses.size() : 6);
      for (size_t i = 0; i < sils.size(); ++i) {
        cams.push_back(std::make_shared<vacancy::PinholeCamera>(320, 240, poses[i], Eigen::Vector2f(159.3f, 127.65f),
                                                                Eigen::Vector2f(258.65f, 258.25f)));
        cam_ptrs.push_back(cams.back().get());
        if (!sils[i].Load(data_dir + "/mask_" + vacancy::zfill(i) + ".png")) return 3;
      }
      if (!sharded->PlanPartition(cam_ptrs, sils)) return 8;
    }
    if (!sharded->Init()) return 5;
    std::printf("BOUNDS");
    for (int z : sharded->z_bounds()) std::printf(" %d", z);
    std::printf("\n");
  }

  const int width = 320, height = 240;
  std::shared_ptr<vacancy::Camera> camera = std::make_shared<vacancy::PinholeCamera>(
      width, height, Eigen::Affine3d::Identity(), Eigen::Vector2f(159.3f, 127.65f), Eigen::Vector2f(258.65f, 258.25f));

  for (size_t i = 0; i < 6 && i < poses.size(); ++i) {
    camera->set_c2w(poses[i]);
    const std::string num = vacancy::zfill(i);
    vacancy::Image1b silhouette;
    if (!silhouette.Load(data_dir + "/mask_" + num + ".png")) return 3;
    vacancy::Image1f sdf;
    if (!carver.Carve(*camera, silhouette, &sdf)) return 4;
    vacancy::Image3b vis;
    vacancy::SignedDistance2Color(sdf, &vis, -1.0f, 1.0f);
    vis.WritePng(out_dir + "/sdf_" + num + ".png");

    if (sharded && !sharded->Carve(*camera, silhouette)) return 6;

    vacancy::Mesh mesh;
    carver.ExtractVoxel(&mesh);  // cube per voxel: large and slow to write, like the reference says
    mesh.WritePlyBinary(out_dir + "/voxel_" + num + ".ply");
    const size_t voxel_verts = mesh.vertices().size();
    if (sharded) {  // ExtractVoxel over the slabs (both predicates) == the single context's, array for array
      bool same = true;
      for (int pass = 0; pass < 2 && same; ++pass) {
        vacancy::Mesh a, b;
        sharded->ExtractVoxel(&a, pass == 1);
        carver.ExtractVoxel(&b, pass == 1);
        same = a.vertices().size() == b.vertices().size() && a.vertex_indices().size() == b.vertex_indices().size() &&
               (pass == 1 || b.vertices().size() == voxel_verts);
        for (size_t k = 0; same && k < a.vertices().size(); ++k)
          for (int q = 0; q < 3; ++q) same = same && a.vertices()[k][q] == b.vertices()[k][q];
        for (size_t k = 0; same && k < a.vertex_indices().size(); ++k)
          for (int q = 0; q < 3; ++q) same = same && a.vertex_indices()[k][q] == b.vertex_indices()[k][q];
      }
      std::printf("VOXELSHARDED view %zu identical %d\n", i, same ? 1 : 0);
    }
    carver.ExtractIsoSurface(&mesh, 0.0);
    mesh.WritePly(out_dir + "/surface_" + num + ".ply");
    const size_t nv = mesh.vertices().size(), nf = mesh.vertex_indices().size();
    double sum[3] = {0, 0, 0};
    for (const auto& v : mesh.vertices())
      for (int k = 0; k < 3; ++k) sum[k] += v[k];
    if (sharded) {
      vacancy::Mesh sm;
      sharded->ExtractIsoSurface(&sm, 0.0);
      bool same = sm.vertices().size() == mesh.vertices().size() &&
                  sm.vertex_indices().size() == mesh.vertex_indices().size();
      for (size_t k = 0; same && k < sm.vertices().size(); ++k)
        for (int a = 0; a < 3; ++a) same = same && sm.vertices()[k][a] == mesh.vertices()[k][a];
      for (size_t k = 0; same && k < sm.vertex_indices().size(); ++k)
        for (int a = 0; a < 3; ++a) same = same && sm.vertex_indices()[k][a] == mesh.vertex_indices()[k][a];
      std::printf("SHARDED view %zu slabs %d identical %d\n", i, sharded->slab_count(), same ? 1 : 0);
    }
    carver.ExtractIsoSurface(&mesh, 0.0, false);
    mesh.WritePly(out_dir + "/surface_nointerp_" + num + ".ply");
    std::printf("RESULT view %zu verts %zu faces %zu nointerp_verts %zu nointerp_faces %zu vsum %.6f %.6f %.6f "
                "voxel_verts %zu\n",
                i, nv, nf, mesh.vertices().size(), mesh.vertex_indices().size(), sum[0], sum[1], sum[2], voxel_verts);
  }

  // The same six views through the batch overload (cameras held by shared_ptr, as examples.cc does) on a
  // second carver, read back as the reference's VoxelGrid: touched voxels, negative voxels, sum of update_num.
  {
    std::vector<std::shared_ptr<vacancy::Camera>> cameras;
    std::vector<vacancy::Image1b> silhouettes(6);
    for (size_t i = 0; i < 6 && i < poses.size(); ++i) {
      cameras.push_back(std::make_shared<vacancy::PinholeCamera>(width, height, poses[i], Eigen::Vector2f(159.3f, 127.65f),
                                                                 Eigen::Vector2f(258.65f, 258.25f)));
      if (!silhouettes[i].Load(data_dir + "/mask_" + vacancy::zfill(i) + ".png")) return 3;
    }
    silhouettes.resize(cameras.size());
    vacancy::VoxelCarver batch(option);
    vacancy::VoxelGrid grid;
    if (!batch.Init() || !batch.Carve(cameras, silhouettes) || !batch.Download(&grid)) return 7;
    const Eigen::Vector3i n = grid.voxel_num();
    long long touched = 0, negative = 0, updates = 0;
    for (int z = 0; z < n[2]; ++z)
      for (int y = 0; y < n[1]; ++y)
        for (int x = 0; x < n[0]; ++x) {
          const vacancy::Voxel& v = grid.get(x, y, z);
          if (v.update_num < 1) continue;
          ++touched;
          negative += v.sdf < 0 ? 1 : 0;
          updates += v.update_num;
        }
    std::printf("GRID touched %lld negative %lld updates %lld\n", touched, negative, updates);
    // ... and through ShardedVoxelCarver's batch overload: the slabs share ONE producer of SDF images
    // (vcy_carve_batch_silhouettes_sharded) -- same mesh as the single context
    if (n_slabs > 0) {
      vacancy::ShardedVoxelCarver sb(option, {0}, n_slabs);
      std::vector<const vacancy::Camera*> ptrs;
      for (const auto& c : cameras) ptrs.push_back(c.get());
      vacancy::Mesh a, b;
      if (!sb.Init() || !sb.Carve(ptrs, silhouettes)) return 9;
      sb.ExtractIsoSurface(&a, 0.0);
      batch.ExtractIsoSurface(&b, 0.0);
      bool same = a.vertices().size() == b.vertices().size() && a.vertex_indices().size() == b.vertex_indices().size();
      for (size_t k = 0; same && k < a.vertices().size(); ++k)
        for (int q = 0; q < 3; ++q) same = same && a.vertices()[k][q] == b.vertices()[k][q];
      for (size_t k = 0; same && k < a.vertex_indices().size(); ++k)
        for (int q = 0; q < 3; ++q) same = same && a.vertex_indices()[k][q] == b.vertex_indices()[k][q];
      std::printf("BATCHSHARDED slabs %d verts %zu identical %d\n", sb.slab_count(), a.vertices().size(), same ? 1 : 0);
    }
  }
  return 0;
}
