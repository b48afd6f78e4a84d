// CPU-only check of the facade's host utilities (no GPU calls): PNG decode of the bunny masks,
// TUM pose -> w2c arithmetic.  Prints values that tests/test_host.py compares with fixtures.
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "vacancy/camera.h"
#include "vacancy/image.h"
#include "vacancy/mesh.h"
#include "vacancy/sharded_voxel_carver.h"
#include "vacancy/voxel_carver.h"

namespace {
// A user's Camera: only Project() overridden -- all the reference's carve path calls (camera.h:39-40,
// voxel_carver.cc:460).  It must compile against the facade, and Carve() must refuse it loudly (the device
// evaluates PinholeCamera and OrthoCamera only).
class FisheyeCamera : public vacancy::Camera {
 public:
  FisheyeCamera(int w, int h) : vacancy::Camera(w, h) {}
  void Project(const Eigen::Vector3f& p, Eigen::Vector2f* q) const override {
    const float r = std::sqrt(p[0] * p[0] + p[1] * p[1]), th = std::atan2(r, p[2]);
    (*q)[0] = 100.0f * th * (r > 0 ? p[0] / r : 0.0f) + 0.5f * width_;
    (*q)[1] = 100.0f * th * (r > 0 ? p[1] / r : 0.0f) + 0.5f * height_;
  }
};
}  // namespace

int main(int argc, char* argv[]) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  if (argc > 2 && std::string(argv[2]) == "gpu") {
    // needs a device: a carver that works for the two known cameras and says no to a third
    vacancy::VoxelCarverOption opt;
    opt.bb_min = Eigen::Vector3f(-16.f, -16.f, -16.f);
    opt.bb_max = Eigen::Vector3f(16.f, 16.f, 16.f);
    opt.resolution = 1.0f;
    vacancy::VoxelCarver carver(opt);
    if (!carver.Init()) return 3;
    vacancy::Image1f sdf(64, 48);
    for (float& v : *sdf.data_ptr()) v = 0.25f;
    Eigen::Translation3d t;
    t.x() = 0.0; t.y() = 0.0; t.z() = -80.0;
    Eigen::Quaterniond q;
    q.x() = 0.0; q.y() = 0.0; q.z() = 0.0; q.w() = 1.0;
    vacancy::PinholeCamera pin(64, 48, t * q, 60.0f);
    vacancy::OrthoCamera ortho(64, 48, t * q);
    FisheyeCamera fish(64, 48);
    const bool a = carver.Carve(pin, sdf), b = carver.Carve(ortho, sdf), c = carver.Carve(fish, sdf);
    std::vector<vacancy::Image1b> sil(1, vacancy::Image1b(64, 48));
    const bool d = carver.Carve(std::vector<const vacancy::Camera*>{&fish}, sil);
    std::vector<float> s;
    std::vector<int> n;
    const bool e = carver.Download(&s, &n);
    long long touched = 0;
    for (int k : n) touched += k > 0;
    std::printf("CUSTOMCAM %d %d %d %d %d %lld\n", a ? 1 : 0, b ? 1 : 0, c ? 1 : 0, d ? 1 : 0, e ? 1 : 0, touched);
    return 0;
  }
  if (argc > 3 && std::string(argv[2]) == "xvtime") {
    // needs a device: what the class API's extractions cost per call, Mesh included (a fresh Mesh per view, as
    // examples.cc:117-149 has it).   host_selftest <data dir> xvtime <resolution>
    std::vector<Eigen::Affine3d> poses;
    {
      std::FILE* fp = std::fopen((dir + "/tumpose.txt").c_str(), "r");
      if (!fp) return 9;
      int id;
      double t[3], q[4];
      while (std::fscanf(fp, "%d %lf %lf %lf %lf %lf %lf %lf", &id, &t[0], &t[1], &t[2], &q[0], &q[1], &q[2], &q[3]) == 8) {
        Eigen::Translation3d tr;
        tr.x() = t[0]; tr.y() = t[1]; tr.z() = t[2];
        Eigen::Quaterniond qu;
        qu.x() = q[0]; qu.y() = q[1]; qu.z() = q[2]; qu.w() = q[3];
        poses.push_back(tr * qu);
      }
      std::fclose(fp);
    }
    vacancy::VoxelCarverOption option;
    option.bb_min = Eigen::Vector3f(-270.000000f, -364.586151f, -149.982697f);
    option.bb_max = Eigen::Vector3f(270.000000f, 170.542343f, 277.329224f);
    option.resolution = (float)std::atof(argv[3]);
    vacancy::VoxelCarver carver(option);
    if (!carver.Init()) return 3;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const int n_slabs = argc > 4 ? std::atoi(argv[4]) : 0;  // > 0: the same views through ShardedVoxelCarver (z-slabs on device 0)
    for (int rep = 0; rep < 2; ++rep) {
      vacancy::VoxelCarver c2(option);
      if (!c2.Init()) return 3;
      std::unique_ptr<vacancy::ShardedVoxelCarver> sh;
      if (n_slabs > 0) {
        sh.reset(new vacancy::ShardedVoxelCarver(option, {0}, n_slabs));
        if (!sh->Init()) return 6;
      }
      for (size_t i = 0; i < 6 && i < poses.size(); ++i) {
        vacancy::PinholeCamera cam(320, 240, poses[i], Eigen::Vector2f(159.3f, 127.65f), Eigen::Vector2f(258.65f, 258.25f));
        vacancy::Image1b sil;
        if (!sil.Load(dir + "/mask_" + vacancy::zfill(i) + ".png")) return 4;
        vacancy::Image1f sdf;
        double t0 = now();
        if (!c2.Carve(cam, sil, &sdf)) return 5;
        const double t_carve = now() - t0;
        t0 = now();
        vacancy::Mesh voxels;
        c2.ExtractVoxel(&voxels);
        const double t_xv = now() - t0;
        t0 = now();
        vacancy::Mesh surface;
        c2.ExtractIsoSurface(&surface, 0.0);
        const double t_mc = now() - t0;
        if (sh) {
          if (!sh->Carve(cam, sil)) return 7;
          t0 = now();
          vacancy::Mesh sm;
          sh->ExtractIsoSurface(&sm, 0.0);
          const double t_s = now() - t0;
          t0 = now();
          vacancy::Mesh sv;
          sh->ExtractVoxel(&sv);
          std::printf("XVTIME rep %d view %zu: %d slabs: ExtractIsoSurface %.3f ms (%zu vertices, identical %d), ExtractVoxel %.2f ms (%zu vertices)\n", rep, i,
                      sh->slab_count(), t_s, sm.vertices().size(),
                      sm.vertices().size() == surface.vertices().size() && sm.vertex_indices().size() == surface.vertex_indices().size() ? 1 : 0,
                      now() - t0, sv.vertices().size());
        }
        std::printf("XVTIME rep %d view %zu: Carve(silhouette, &sdf) %.3f ms, ExtractVoxel %.2f ms (%zu vertices), ExtractIsoSurface %.3f ms (%zu vertices)\n", rep, i,
                    t_carve, t_xv, voxels.vertices().size(), t_mc, surface.vertices().size());
      }
    }
    return 0;
  }
  if (argc > 3 && std::string(argv[2]) == "io") {
    // Host outputs nothing else looks at (rows f3 / f4): tests/test_host.py reads every byte of these files back.
    //   <out>/mesh_ascii.ply, mesh_binary.ply, mesh_empty.ply: the same small mesh through both writers
    //   <out>/sdf_<i>.f32 (written by the test) -> SignedDistance2Color -> <out>/vis_<i>.rgb (raw) and vis_<i>.png
    //   PNGRT lines: WritePng -> Load gives the pixels back (1 and 3 channels)
    const std::string out = argv[3];
    vacancy::Mesh mesh;
    std::vector<Eigen::Vector3f> v;
    std::vector<Eigen::Vector3i> f;
    for (int i = 0; i < 5000; ++i)  // (values with short and long %g forms, negative zero, denormal-free)
      v.push_back(Eigen::Vector3f(0.1f * i - 250.0f, 1.0f / (1.0f + i), (i % 7) * -1234.5678f));
    for (int i = 0; i < 9000; ++i) f.push_back(Eigen::Vector3i(i % 5000, (i * 7 + 1) % 5000, (i * 13 + 2) % 5000));
    mesh.set_vertices(v);
    mesh.set_vertex_indices(f);
    bool ok = mesh.WritePly(out + "/mesh_ascii.ply") && mesh.WritePlyBinary(out + "/mesh_binary.ply");
    vacancy::Mesh empty;
    ok = ok && empty.WritePlyBinary(out + "/mesh_empty.ply");
    ok = ok && !mesh.WritePlyBinary(out + "/no/such/dir/x.ply");
    std::printf("PLY %d %zu %zu\n", ok ? 1 : 0, v.size(), f.size());
    for (int i = 0; i < 6; ++i) {
      vacancy::Image1b m;
      if (!m.Load(dir + "/mask_" + vacancy::zfill(i) + ".png")) return 1;
      vacancy::Image1f sdf(m.width(), m.height());
      std::FILE* fp = std::fopen((out + "/sdf_" + std::to_string(i) + ".f32").c_str(), "rb");
      if (!fp) return 4;
      const size_t n = sdf.data().size();
      if (std::fread(sdf.data_ptr()->data(), sizeof(float), n, fp) != n) return 5;
      std::fclose(fp);
      vacancy::Image3b vis;
      // (the ranges examples.cc passes: -1 .. 1 for normalised fields; view 3 with a narrower one so that both clamps fire)
      const float lo = i == 3 ? -0.25f : -1.0f, hi = i == 3 ? 0.125f : 1.0f;
      vacancy::SignedDistance2Color(sdf, &vis, lo, hi);
      fp = std::fopen((out + "/vis_" + std::to_string(i) + ".rgb").c_str(), "wb");
      if (!fp) return 6;
      std::fwrite(vis.data().data(), 1, vis.data().size(), fp);
      std::fclose(fp);
      bool rt = vis.WritePng(out + "/vis_" + std::to_string(i) + ".png");
      vacancy::Image3b back;
      rt = rt && back.Load(out + "/vis_" + std::to_string(i) + ".png") && back.width() == vis.width() &&
           back.height() == vis.height() && back.data() == vis.data();
      bool rt1 = m.WritePng(out + "/mask_" + std::to_string(i) + ".png");
      vacancy::Image1b back1;
      rt1 = rt1 && back1.Load(out + "/mask_" + std::to_string(i) + ".png") && back1.data() == m.data() &&
            back1.width() == m.width();
      vacancy::Image3b wrong;
      const bool refuses = !wrong.Load(out + "/mask_" + std::to_string(i) + ".png");  // 1 channel into a 3-channel image
      std::printf("PNGRT %d %d %d %d\n", i, rt ? 1 : 0, rt1 ? 1 : 0, refuses ? 1 : 0);
    }
    vacancy::Image1b none;
    std::printf("PNGEMPTY %d\n", none.WritePng(out + "/none.png") ? 1 : 0);
    return 0;
  }
  for (int i = 0; i < 6; ++i) {
    vacancy::Image1b m;
    if (!m.Load(dir + "/mask_" + vacancy::zfill(i) + ".png")) return 1;
    unsigned long long sum = 0;
    for (unsigned char p : m.data()) sum += p;
    std::printf("MASK %d %d %d %llu\n", i, m.width(), m.height(), sum);
  }
  // view 2 of tumpose.txt
  Eigen::Translation3d t;
  t.x() = 710.836121; t.y() = -48.510956; t.z() = 31.836634;
  Eigen::Quaterniond q;
  q.x() = 0.0; q.y() = -0.707107; q.z() = 0.0; q.w() = 0.707107;
  vacancy::PinholeCamera cam(320, 240, t * q, Eigen::Vector2f(159.3f, 127.65f), Eigen::Vector2f(258.65f, 258.25f));
  const Eigen::Affine3f w2c = cam.w2c().cast<float>();
  std::printf("W2C");
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) std::printf(" %.9g", w2c.linear()(i, j));
    std::printf(" %.9g", w2c.translation()[i]);
  }
  std::printf("\n");
  // VoxelGrid::Init on the bunny bounding box (examples.cc:91-99) at resolution 10: dims and an FNV-1a-64
  // hash of the voxel centres in id order (res10_pos)
  {
    Eigen::Vector3f bb_min(-250.000000f, -344.586151f, -129.982697f), bb_max(250.000000f, 150.542343f, 257.329224f);
    for (int i = 0; i < 3; ++i) {
      bb_min[i] -= 20.0f;
      bb_max[i] += 20.0f;
    }
    vacancy::VoxelGrid grid;
    if (grid.initialized() || !grid.Init(bb_max, bb_min, 10.0f) || !grid.initialized()) return 2;
    const Eigen::Vector3i n = grid.voxel_num();
    unsigned long long h = 1469598103934665603ull;
    bool ok = true;
    int id = 0;
    for (int z = 0; z < n[2]; ++z)
      for (int y = 0; y < n[1]; ++y)
        for (int x = 0; x < n[0]; ++x, ++id) {
          const vacancy::Voxel& v = grid.get(x, y, z);
          ok = ok && v.id == id && v.index[0] == x && v.index[1] == y && v.index[2] == z && v.update_num == 0 &&
               v.sdf == vacancy::InvalidSdf::kVal && !v.on_surface && !v.outside;
          for (int k = 0; k < 3; ++k) {
            const float f = v.pos[k];
            const unsigned char* b = reinterpret_cast<const unsigned char*>(&f);
            for (int q = 0; q < 4; ++q) h = (h ^ b[q]) * 1099511628211ull;
          }
        }
    grid.get_ptr(1, 2, 3)->on_surface = true;
    grid.ResetOnSurface();
    ok = ok && !grid.get(1, 2, 3).on_surface && grid.resolution() == 10.0f;
    vacancy::VoxelGrid bad;
    ok = ok && !bad.Init(bb_min, bb_max, 10.0f) && !bad.Init(bb_max, bb_min, 0.0f);  // inverted box, zero resolution
    // a box thinner than one voxel: the reference returns true with an empty grid (voxel_carver.cc:292-345)
    vacancy::VoxelGrid thin;
    ok = ok && thin.Init(Eigen::Vector3f(100.f, 100.f, 5.f), Eigen::Vector3f(0.f, 0.f, 0.f), 10.0f) &&
         !thin.initialized() && thin.voxel_num()[2] == 0 && thin.voxel_num()[0] == 10;
    std::printf("GRID %d %d %d %016llx %d\n", n[0], n[1], n[2], h, ok ? 1 : 0);
  }
  // the reference's look-at forms (common.h:51-75) against the Affine one
  {
    const Eigen::Vector3d pos(0.3, -1.2, 2.5), target(0.1, 0.2, -0.4), up(0.0, -1.0, 0.0);
    const Eigen::Affine3d pose = vacancy::c2w(pos, target, up);
    Eigen::Matrix3d R;
    vacancy::c2w(pos, target, up, &R);
    Eigen::Matrix4d T;
    vacancy::c2w(pos, target, up, &T);
    bool same = T(3, 0) == 0.0 && T(3, 1) == 0.0 && T(3, 2) == 0.0 && T(3, 3) == 1.0;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) same = same && R(i, j) == pose.linear()(i, j) && T(i, j) == R(i, j);
      same = same && T(i, 3) == pos[i];
    }
    std::printf("C2W %d\n", same ? 1 : 0);
  }
  vacancy::PinholeCamera fov(1280, 720, 60.0f);
  std::printf("FOCAL %.9g %.9g %.9g\n", fov.focal_length()[0], fov.principal_point()[0], fov.principal_point()[1]);
  return 0;
}
