// vacancy::VoxelCarver facade: the reference's class API (src/vacancy/voxel_carver.cc:367-543)
// forwarding to the HIP path through the C-ABI of include/vacancy_hip.h.
#include "vacancy/voxel_carver.h"

#include <chrono>
#include <cstring>
#include <limits>

#include "vacancy_hip.h"
#include "mesh_copy.h"

namespace vacancy {

const float InvalidSdf::kVal = std::numeric_limits<float>::lowest();

namespace {
double NowMs() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

vcy_view ToView(const Camera& camera, const Eigen::Vector2i& roi_min, const Eigen::Vector2i& roi_max, int width,
                int height, bool* ok) {
  vcy_view v;
  std::memset(&v, 0, sizeof(v));
  const Eigen::Affine3f w2c = camera.w2c().cast<float>();  // reference voxel_carver.cc:438
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) v.w2c[4 * i + j] = w2c.linear()(i, j);
    v.w2c[4 * i + 3] = w2c.translation()[i];
  }
  // Camera::Project is virtual in the reference (camera.h:39-40, called at voxel_carver.cc:460); the device knows the
  // two projections the reference implements.  Anything else is refused, never projected with fx = fy = 0.
  if (const PinholeCamera* p = dynamic_cast<const PinholeCamera*>(&camera)) {
    v.fx = p->focal_length()[0];
    v.fy = p->focal_length()[1];
    v.cx = p->principal_point()[0];
    v.cy = p->principal_point()[1];
  } else if (dynamic_cast<const OrthoCamera*>(&camera)) {
    v.is_ortho = 1;
  } else {
    *ok = false;
    LOGE("VoxelCarver::Carve unsupported Camera subclass: the HIP path projects PinholeCamera and OrthoCamera only\n");
  }
  v.roi_min[0] = roi_min[0];
  v.roi_min[1] = roi_min[1];
  v.roi_max[0] = roi_max[0];
  v.roi_max[1] = roi_max[1];
  v.width = width;
  v.height = height;
  return v;
}

// Per-view Carve() calls only queue their view; the loop the reference times as "VoxelCarver::Carve main loop"
// (voxel_carver.cc:435,492-493) runs when the queue is applied -- inside the next call that reads the state.  The
// library keeps HIP-event times of every fused launch ("carvetimer"); they are logged here, one line per launch, in the
// reference's words, by the calls that apply the queue.
// The timer costs three event records per fused launch: it is only on while the log level emits LOGI (checked at every
// call that applies the queue, so a level changed after Init is followed; `*timer_on` is the facade's view of the param).
void LogAppliedCarves(vcy_ctx* ctx, bool* timer_on) {
  const bool want = get_log_level() <= LogLevel::kInfo;
  if (*timer_on) {
    // every record the library holds (its log ends at 8192 launches, vcy_carve_log), not a first screenful of them
    constexpr int kCap = 8192;
    std::vector<float> pre(kCap), ker(kCap);
    std::vector<int32_t> first(kCap);
    int n = 0;
    if (vcy_carve_log(ctx, kCap, nullptr, pre.data(), ker.data(), first.data(), &n, 1) == VCY_OK && want) {
      double ms = 0.0;
      for (int i = 0; i < n; ++i) {
        if (first[i] && i > 0) {
          LOGI("VoxelCarver::Carve main loop %02f\n", ms);
          ms = 0.0;
        }
        ms += static_cast<double>(pre[i]) + ker[i];
      }
      if (n > 0) LOGI("VoxelCarver::Carve main loop %02f\n", ms);
    }
  }
  if (want != *timer_on) {
    vcy_set_param(ctx, "carvetimer", want ? 1 : 0);
    *timer_on = want;
  }
}
}  // namespace

Voxel::Voxel() {}
Voxel::~Voxel() {}

VoxelGrid::VoxelGrid() {}
VoxelGrid::~VoxelGrid() {}

// The container of reference voxel_carver.cc:276-345.  Its sizes and voxel centres come from the library -- the
// arithmetic of VoxelGrid::Init lives in ONE place, next to the device's axis tables (vcy_compute_dims,
// vcy_axis_positions; vcy_download_positions returns the same values bit for bit) -- and are only laid out here.
bool VoxelGrid::Init(const Eigen::Vector3f& bb_max, const Eigen::Vector3f& bb_min, float resolution) {
  const float mn[3] = {bb_min.x(), bb_min.y(), bb_min.z()}, mx[3] = {bb_max.x(), bb_max.y(), bb_max.z()};
  int32_t n[3];
  if (vcy_compute_dims(mn, mx, resolution, n) != VCY_OK) {  // (resolution, bounding box: the reference's checks and texts)
    LOGE("%s\n", vcy_last_error());
    return false;
  }
  if (static_cast<long long>(n[0]) * n[1] * n[2] > std::numeric_limits<int>::max()) {  // 32-bit ids, :298-301
    LOGE("too many voxels\n");
    return false;
  }
  const bool empty = n[0] <= 0 || n[1] <= 0 || n[2] <= 0;  // (the reference returns true with no voxels, :292-345)
  std::vector<float> axis[3];
  for (int a = 0; a < 3 && !empty; ++a) {
    axis[a].resize(static_cast<size_t>(n[a]));
    if (vcy_axis_positions(mn, mx, resolution, a, axis[a].data()) != VCY_OK) return false;
  }
  bb_max_ = bb_max;
  bb_min_ = bb_min;
  resolution_ = resolution;
  voxel_num_ = Eigen::Vector3i(n[0], n[1], n[2]);
  xy_slice_num_ = n[0] * n[1];
  voxels_.clear();
  if (empty) return true;
  voxels_.resize(static_cast<size_t>(n[0]) * n[1] * n[2]);
  size_t id = 0;
  for (int z = 0; z < n[2]; z++)
    for (int y = 0; y < n[1]; y++)
      for (int x = 0; x < n[0]; x++, id++) {
        Voxel& voxel = voxels_[id];
        voxel.index = Eigen::Vector3i(x, y, z);
        voxel.id = static_cast<int>(id);
        voxel.pos = Eigen::Vector3f(axis[0][x], axis[1][y], axis[2][z]);
        voxel.sdf = InvalidSdf::kVal;
      }
  return true;
}
const Eigen::Vector3i& VoxelGrid::voxel_num() const { return voxel_num_; }
const Voxel& VoxelGrid::get(int x, int y, int z) const {
  return voxels_[static_cast<size_t>(z) * xy_slice_num_ + (static_cast<size_t>(y) * voxel_num_.x() + x)];
}
Voxel* VoxelGrid::get_ptr(int x, int y, int z) {
  return &voxels_[static_cast<size_t>(z) * xy_slice_num_ + (static_cast<size_t>(y) * voxel_num_.x() + x)];
}
float VoxelGrid::resolution() const { return resolution_; }
void VoxelGrid::ResetOnSurface() {
  for (Voxel& v : voxels_) v.on_surface = false;
}
bool VoxelGrid::initialized() const { return !voxels_.empty(); }

struct VoxelCarver::Impl {
  VoxelCarverOption option;
  vcy_ctx* ctx = nullptr;
  int device = 0;
  bool carve_timer = false;  // "carvetimer" as last set by this facade (LogAppliedCarves)
  ~Impl() { vcy_destroy(ctx); }
};

VoxelCarver::VoxelCarver() : impl_(new Impl) {}
VoxelCarver::VoxelCarver(VoxelCarverOption option) : impl_(new Impl) { set_option(option); }
VoxelCarver::~VoxelCarver() {}

void VoxelCarver::set_option(VoxelCarverOption option) { impl_->option = option; }
void VoxelCarver::set_device(int device_id) { impl_->device = device_id; }

bool VoxelCarver::Init() {
  vcy_destroy(impl_->ctx);
  impl_->ctx = nullptr;
  const VoxelCarverOption& o = impl_->option;
  vcy_carver_option c;
  std::memset(&c, 0, sizeof(c));
  for (int i = 0; i < 3; ++i) {
    c.bb_max[i] = o.bb_max[i];
    c.bb_min[i] = o.bb_min[i];
  }
  c.resolution = o.resolution;
  c.sdf_minmax_normalize = o.sdf_minmax_normalize ? 1 : 0;
  c.update_option.voxel_update = static_cast<int>(o.update_option.voxel_update);
  c.update_option.sdf_interp = static_cast<int>(o.update_option.sdf_interp);
  c.update_option.update_outside = static_cast<int>(o.update_option.update_outside);
  c.update_option.voxel_max_update_num = o.update_option.voxel_max_update_num;
  c.update_option.voxel_update_weight = o.update_option.voxel_update_weight;
  c.update_option.use_truncation = o.update_option.use_truncation ? 1 : 0;
  c.update_option.truncation_band = o.update_option.truncation_band;
  if (vcy_create(&c, impl_->device, 0, -1, &impl_->ctx) != VCY_OK) {
    LOGE("%s\n", vcy_last_error());
    return false;
  }
  // one context holds the whole grid: no slab merge, so the mesh needs no edge keys (the reference's
  // MarchingCubes returns vertices and faces)
  vcy_set_param(impl_->ctx, "meshkeys", 0);
  impl_->carve_timer = get_log_level() <= LogLevel::kInfo;  // (three events per fused launch: only when they are logged)
  vcy_set_param(impl_->ctx, "carvetimer", impl_->carve_timer ? 1 : 0);
  return true;
}

bool VoxelCarver::Carve(const Camera& camera, const Image1b& silhouette, const Eigen::Vector2i& roi_min,
                        const Eigen::Vector2i& roi_max, Image1f* sdf) {
  if (!impl_->ctx) {
    LOGE("VoxelCarver::Carve voxel grid has not been initialized\n");
    return false;
  }
  sdf->Init(silhouette.width(), silhouette.height(), 0.0f);
  bool known = true;
  const vcy_view v = ToView(camera, roi_min, roi_max, silhouette.width(), silhouette.height(), &known);
  if (!known) return false;
  const double t0 = NowMs();
  // The view is QUEUED by the library (vcy_set_param "defer"): this call returns after the silhouette has
  // been uploaded and its SDF built and downloaded; queued views are carved together by one fused
  // launch when the state is next needed (Extract*, Download).  No vcy_sync here -- it would flush the
  // queue after every view and turn the fused pass into one pass per view.
  const int rc = vcy_carve_silhouette(impl_->ctx, &v, silhouette.data().data(), sdf->data_ptr()->data());
  LOGI("VoxelCarver::Carve make SDF + enqueue %02f\n", NowMs() - t0);
  if (rc != VCY_OK) LOGE("%s\n", vcy_last_error());
  return rc == VCY_OK;
}

bool VoxelCarver::Carve(const Camera& camera, const Eigen::Vector2i& roi_min, const Eigen::Vector2i& roi_max,
                        const Image1f& sdf) {
  if (!impl_->ctx) {
    LOGE("VoxelCarver::Carve voxel grid has not been initialized\n");
    return false;
  }
  bool known = true;
  const vcy_view v = ToView(camera, roi_min, roi_max, sdf.width(), sdf.height(), &known);
  if (!known) return false;
  const double t0 = NowMs();
  const int rc = vcy_carve(impl_->ctx, &v, sdf.data().data());  // copied and queued, see above
  LOGI("VoxelCarver::Carve enqueue %02f\n", NowMs() - t0);
  if (rc != VCY_OK) LOGE("%s\n", vcy_last_error());
  return rc == VCY_OK;
}

bool VoxelCarver::Carve(const Camera& camera, const Image1b& silhouette, Image1f* sdf) {
  return Carve(camera, silhouette, Eigen::Vector2i(0, 0),
               Eigen::Vector2i(silhouette.width() - 1, silhouette.height() - 1), sdf);
}

bool VoxelCarver::Carve(const Camera& camera, const Image1b& silhouette) {
  Image1f sdf(camera.width(), camera.height());
  return Carve(camera, silhouette, &sdf);
}

bool VoxelCarver::Carve(const Camera& camera, const Image1f& sdf) {
  return Carve(camera, Eigen::Vector2i(0, 0), Eigen::Vector2i(sdf.width() - 1, sdf.height() - 1), sdf);
}

bool VoxelCarver::Carve(const std::vector<Camera>& cameras, const std::vector<Image1b>& silhouettes) {
  std::vector<const Camera*> ptrs(cameras.size());
  for (size_t i = 0; i < cameras.size(); ++i) ptrs[i] = &cameras[i];
  return Carve(ptrs, silhouettes);
}

bool VoxelCarver::Carve(const std::vector<std::shared_ptr<Camera>>& cameras, const std::vector<Image1b>& silhouettes) {
  std::vector<const Camera*> ptrs(cameras.size());
  for (size_t i = 0; i < cameras.size(); ++i) ptrs[i] = cameras[i].get();
  return Carve(ptrs, silhouettes);
}

bool VoxelCarver::Carve(const std::vector<const Camera*>& cameras, const std::vector<Image1b>& silhouettes) {
  if (!impl_->ctx || cameras.size() != silhouettes.size() || cameras.empty()) return false;
  const int n = static_cast<int>(cameras.size());
  std::vector<vcy_view> views(n);
  std::vector<const uint8_t*> masks(n);
  for (int i = 0; i < n; ++i) {
    const Image1b& s = silhouettes[i];
    bool known = true;
    views[i] = ToView(*cameras[i], Eigen::Vector2i(0, 0), Eigen::Vector2i(s.width() - 1, s.height() - 1), s.width(),
                      s.height(), &known);
    if (!known) return false;
    masks[i] = s.data().data();
  }
  // masks are streamed to the device, SDFs built there, views fused in chunks of 32
  const bool ok = vcy_carve_batch_silhouettes(impl_->ctx, n, views.data(), masks.data()) == VCY_OK;
  if (!ok) LOGE("%s\n", vcy_last_error());
  LogAppliedCarves(impl_->ctx, &impl_->carve_timer);
  return ok;
}

void VoxelCarver::ExtractIsoSurface(Mesh* mesh, double iso_level, bool linear_interp) {
  mesh->Clear();
  if (!impl_->ctx) return;
  const double t0 = NowMs();
  vcy_mesh m;
  if (vcy_extract_iso(impl_->ctx, iso_level, linear_interp ? 1 : 0, &m) != VCY_OK) {
    LOGE("%s\n", vcy_last_error());
    vcy_mesh_free(&m);
    LogAppliedCarves(impl_->ctx, &impl_->carve_timer);  // (the queue may have been applied before the failure)
    return;
  }
  LogAppliedCarves(impl_->ctx, &impl_->carve_timer);
  static_assert(sizeof(Eigen::Vector3f) == 3 * sizeof(float), "packed vector layout");
  static_assert(sizeof(Eigen::Vector3i) == 3 * sizeof(int), "packed vector layout");
  std::vector<Eigen::Vector3f>* v = mesh->mutable_vertices();
  std::vector<Eigen::Vector3i>* f = mesh->mutable_vertex_indices();
  detail::CopyTriples(v, m.vertices, static_cast<size_t>(m.n_vertices));
  detail::CopyTriples(f, m.faces, static_cast<size_t>(m.n_faces));
  vcy_mesh_free(&m);
  LOGI("MarchingCubes %02f\n", NowMs() - t0);
}

void VoxelCarver::ExtractVoxel(Mesh* mesh, bool inside_empty) {
  mesh->Clear();
  if (!impl_->ctx) return;
  const double t0 = NowMs();
  // (the library's host threads fill the Mesh's own vectors: no library-owned copy of the mesh in between)
  typedef detail::MeshArrays<Eigen::Vector3f, Eigen::Vector3i> Arrays;
  Arrays arrays{mesh->mutable_vertices(), mesh->mutable_vertex_indices()};
  if (vcy_extract_voxel_into(impl_->ctx, inside_empty ? 1 : 0, &Arrays::Provide, &arrays) != VCY_OK) {
    LOGE("%s\n", vcy_last_error());
    mesh->Clear();
    LogAppliedCarves(impl_->ctx, &impl_->carve_timer);
    return;
  }
  LogAppliedCarves(impl_->ctx, &impl_->carve_timer);
  LOGI("VoxelCarver::ExtractVoxel %02f\n", NowMs() - t0);
}

Eigen::Vector3i VoxelCarver::voxel_num() const {
  int32_t d[3] = {0, 0, 0};
  if (impl_->ctx) vcy_grid_dims(impl_->ctx, d);
  return Eigen::Vector3i(d[0], d[1], d[2]);
}

bool VoxelCarver::Download(std::vector<float>* sdf, std::vector<int>* update_num) const {
  if (!impl_->ctx) return false;
  const Eigen::Vector3i n = voxel_num();
  const size_t total = static_cast<size_t>(n[0]) * n[1] * n[2];
  if (sdf) sdf->resize(total);
  if (update_num) update_num->resize(total);
  const bool ok = vcy_download(impl_->ctx, sdf ? sdf->data() : nullptr, update_num ? update_num->data() : nullptr) == VCY_OK;
  LogAppliedCarves(impl_->ctx, &impl_->carve_timer);
  return ok;
}

bool VoxelCarver::Download(VoxelGrid* grid) const {
  if (!impl_->ctx || !grid) return false;
  const VoxelCarverOption& o = impl_->option;
  if (!grid->Init(o.bb_max, o.bb_min, o.resolution)) return false;
  const Eigen::Vector3i n = grid->voxel_num();
  if (n[0] != voxel_num()[0] || n[1] != voxel_num()[1] || n[2] != voxel_num()[2]) return false;
  std::vector<float> sdf;
  std::vector<int> update_num;
  if (!Download(&sdf, &update_num)) return false;
  size_t i = 0;
  for (int z = 0; z < n[2]; ++z)
    for (int y = 0; y < n[1]; ++y)
      for (int x = 0; x < n[0]; ++x, ++i) {
        Voxel* v = grid->get_ptr(x, y, z);
        v->sdf = sdf[i];
        v->update_num = update_num[i];
      }
  return true;
}

void DistanceTransformL1(const Image1b& mask, const Eigen::Vector2i& roi_min, const Eigen::Vector2i& roi_max,
                         Image1f* dist) {
  dist->Init(mask.width(), mask.height(), 0.0f);
  const int32_t rmin[2] = {roi_min[0], roi_min[1]}, rmax[2] = {roi_max[0], roi_max[1]};
  if (vcy_distance_transform_l1(mask.data().data(), mask.width(), mask.height(), rmin, rmax,
                                dist->data_ptr()->data()) != VCY_OK)
    LOGE("%s\n", vcy_last_error());
}

void MakeSignedDistanceField(const Image1b& mask, const Eigen::Vector2i& roi_min, const Eigen::Vector2i& roi_max,
                             Image1f* dist, bool minmax_normalize, bool use_truncation, float truncation_band) {
  dist->Init(mask.width(), mask.height(), 0.0f);
  const int32_t rmin[2] = {roi_min[0], roi_min[1]}, rmax[2] = {roi_max[0], roi_max[1]};
  if (vcy_make_sdf(mask.data().data(), mask.width(), mask.height(), rmin, rmax, minmax_normalize, use_truncation,
                   truncation_band, dist->data_ptr()->data()) != VCY_OK)
    LOGE("%s\n", vcy_last_error());
}

// blue inside, red outside, white at the silhouette (reference voxel_carver.cc:239-267)
void SignedDistance2Color(const Image1f& sdf, Image3b* vis, float min_negative_d, float max_positive_d) {
  vis->Init(sdf.width(), sdf.height());
  for (int y = 0; y < sdf.height(); ++y)
    for (int x = 0; x < sdf.width(); ++x) {
      const float d = sdf.at(x, y, 0);
      uint8_t* px = vis->at(x, y);
      if (d > 0) {
        float k = (max_positive_d - d) / max_positive_d;
        k = std::min(std::max(k, 0.0f), 1.0f);
        px[0] = 255;
        px[1] = px[2] = static_cast<uint8_t>(255 * k);
      } else {
        float k = (d - min_negative_d) / (-min_negative_d);
        k = std::min(std::max(k, 0.0f), 1.0f);
        px[0] = px[1] = static_cast<uint8_t>(255 * k);
        px[2] = 255;
      }
    }
}

}  // namespace vacancy
