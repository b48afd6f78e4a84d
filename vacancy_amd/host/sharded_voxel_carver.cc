// ShardedVoxelCarver: z-slab sharding over the C-ABI contexts of include/vacancy_hip.h.
#include "vacancy/sharded_voxel_carver.h"

#include <cstring>
#include <future>
#include <limits>
#include <string>
#include <unordered_map>

#include "vacancy_hip.h"
#include "mesh_copy.h"

namespace vacancy {

namespace {

vcy_view MakeView(const Camera& camera, int width, int height, bool* ok) {
  vcy_view v;
  std::memset(&v, 0, sizeof(v));
  const Eigen::Affine3f w2c = camera.w2c().cast<float>();
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
  v.roi_max[0] = width - 1;
  v.roi_max[1] = height - 1;
  v.width = width;
  v.height = height;
  return v;
}

struct KeyHash {
  size_t operator()(const std::pair<int64_t, int64_t>& k) const {
    return std::hash<int64_t>()(k.first * 1000003 + (k.second - k.first));
  }
};

}  // namespace

struct ShardedVoxelCarver::Impl {
  VoxelCarverOption option;
  std::vector<int> devices;
  int per_device = 1;
  std::vector<int> wanted_bounds;  // PlanPartition / set_z_bounds; empty: equal thickness
  std::vector<int> bounds;         // of the slabs that exist
  int64_t slice = 0;               // voxels per z slice of the grid (nx * ny)
  std::vector<vcy_ctx*> slabs;  // in z order
  bool peer_copy_halo = false;
  ~Impl() {
    for (vcy_ctx* c : slabs) vcy_destroy(c);
  }
};

ShardedVoxelCarver::ShardedVoxelCarver(VoxelCarverOption option, std::vector<int> device_ids, int slabs_per_device)
    : impl_(new Impl) {
  impl_->option = option;
  impl_->devices = device_ids.empty() ? std::vector<int>{0} : device_ids;
  impl_->per_device = slabs_per_device < 1 ? 1 : slabs_per_device;
}
ShardedVoxelCarver::~ShardedVoxelCarver() {}

void ShardedVoxelCarver::set_halo_transport(HaloTransport t) { impl_->peer_copy_halo = t == HaloTransport::kPeerCopy; }

int ShardedVoxelCarver::slab_count() const { return static_cast<int>(impl_->slabs.size()); }

namespace {
vcy_carver_option ToC(const VoxelCarverOption& o) {
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
  return c;
}
}  // namespace

void ShardedVoxelCarver::set_z_bounds(const std::vector<int>& z_bounds) { impl_->wanted_bounds = z_bounds; }
const std::vector<int>& ShardedVoxelCarver::z_bounds() const { return impl_->bounds; }

bool ShardedVoxelCarver::PlanPartition(const std::vector<const Camera*>& cameras,
                                       const std::vector<Image1b>& silhouettes) {
  if (cameras.empty() || cameras.size() != silhouettes.size()) return false;
  const vcy_carver_option c = ToC(impl_->option);
  const int count = static_cast<int>(impl_->devices.size()) * impl_->per_device;
  if (count < 2) return true;  // one slab: nothing to cut
  vcy_ctx* planner = nullptr;
  if (vcy_create(&c, impl_->devices[0], 0, 8, &planner) != VCY_OK) {  // (its 8-slice slab is never touched)
    LOGE("%s\n", vcy_last_error());
    return false;
  }
  const int n = static_cast<int>(cameras.size());
  std::vector<vcy_view> views(n);
  std::vector<float*> imgs(n, nullptr);
  bool ok = true;
  for (int i = 0; i < n && ok; ++i) {
    views[i] = MakeView(*cameras[i], silhouettes[i].width(), silhouettes[i].height(), &ok);
    if (!ok) break;
    // MakeSignedDistanceField as Carve() will build it (voxel_carver.cc:405-408), on the device
    ok = vcy_make_sdf_device(planner, silhouettes[i].data().data(), views[i].width, views[i].height, views[i].roi_min,
                             views[i].roi_max, c.sdf_minmax_normalize, c.update_option.use_truncation,
                             c.update_option.truncation_band, &imgs[i]) == VCY_OK;
  }
  std::vector<int32_t> b(static_cast<size_t>(count) + 1, 0);
  if (ok)
    ok = vcy_plan_z_slabs(planner, n, views.data(), imgs.data(), count, 0, 0.0f, b.data(), nullptr, 0, nullptr) == VCY_OK;
  if (!ok) LOGE("PlanPartition: %s\n", vcy_last_error());
  for (float* p : imgs)
    if (p) vcy_device_free(planner, p);
  vcy_destroy(planner);
  if (ok) impl_->wanted_bounds.assign(b.begin(), b.end());
  return ok;
}

bool ShardedVoxelCarver::Init() {
  for (vcy_ctx* c : impl_->slabs) vcy_destroy(c);
  impl_->slabs.clear();
  impl_->bounds.clear();
  const vcy_carver_option c = ToC(impl_->option);
  int32_t dims[3];
  if (vcy_compute_dims(c.bb_min, c.bb_max, c.resolution, dims) != VCY_OK) {
    LOGE("%s\n", vcy_last_error());
    return false;
  }
  const int ndev = static_cast<int>(impl_->devices.size());
  int count = ndev * impl_->per_device;
  while (count > 1 && dims[2] / count < 2) --count;  // every slab needs >= 2 slices
  const int base = dims[2] / count, rem = dims[2] % count;
  std::vector<int> bounds(static_cast<size_t>(count) + 1, dims[2]);
  for (int s = 0; s < count; ++s) bounds[s] = s * base + (s < rem ? s : rem);
  {  // cuts from PlanPartition / set_z_bounds, if they fit this grid and slab count
    const std::vector<int>& w = impl_->wanted_bounds;
    bool fits = static_cast<int>(w.size()) == count + 1 && w.front() == 0 && w.back() == dims[2];
    for (size_t s = 0; fits && s + 1 < w.size(); ++s) fits = w[s + 1] - w[s] >= 2;
    if (fits) bounds = w;
    else if (!w.empty()) LOGW("ShardedVoxelCarver: the given z bounds do not fit %d slabs of this grid; equal thickness\n", count);
  }
  impl_->bounds = bounds;
  impl_->slice = static_cast<int64_t>(dims[0]) * dims[1];
  for (int s = 0; s < count; ++s) {
    const int z0 = bounds[s], z1 = bounds[s + 1];
    vcy_ctx* ctx = nullptr;
    if (vcy_create(&c, impl_->devices[s % ndev], z0, z1, &ctx) != VCY_OK) {  // cyclic deal
      LOGE("%s\n", vcy_last_error());
      return false;
    }
    // Several slabs are driven from one host thread here and in vcy_carve_batch_silhouettes_sharded: a launch of few
    // views over a carved slab must not make that thread wait for the slab's live-workgroup count ("livesync" 1, worth
    // 0.1 ms per launch on a whole 1024^3 grid) before it can enqueue the next slab's work.
    if (count > 1) (void)vcy_set_param(ctx, "livesync", 0);
    impl_->slabs.push_back(ctx);
  }
  return true;
}

bool ShardedVoxelCarver::Carve(const Camera& camera, const Image1b& silhouette) {
  return Carve(std::vector<const Camera*>{&camera}, std::vector<Image1b>{silhouette});
}

bool ShardedVoxelCarver::Carve(const std::vector<const Camera*>& cameras, const std::vector<Image1b>& silhouettes) {
  if (impl_->slabs.empty() || cameras.size() != silhouettes.size() || cameras.empty()) return false;
  const int n = static_cast<int>(cameras.size());
  std::vector<vcy_view> views(n);
  std::vector<const uint8_t*> masks(n);
  for (int i = 0; i < n; ++i) {
    bool known = true;
    views[i] = MakeView(*cameras[i], silhouettes[i].width(), silhouettes[i].height(), &known);
    if (!known) return false;
    masks[i] = silhouettes[i].data().data();
  }
  // Several views: the devices SHARE the producer (vcy_carve_batch_silhouettes_sharded: device r uploads and transforms
  // the silhouettes r, r + R, ... of every chunk of 32, one RCCL all-gather per chunk hands every device all the SDF
  // images, every slab carves from its device's copy) -- the reference calls MakeSignedDistanceField once per view
  // (voxel_carver.cc:405-408), and so does a node of GPUs.  Without librccl (VCY_ERR_UNSUPPORTED) every slab builds
  // its own, below.
  if (n > 1) {
    const int rc = vcy_carve_batch_silhouettes_sharded(impl_->slabs.data(), static_cast<int>(impl_->slabs.size()), n,
                                                       views.data(), masks.data());
    if (rc == VCY_OK) return true;
    if (rc != VCY_ERR_UNSUPPORTED) {
      LOGE("sharded carve failed: %s\n", vcy_last_error());
      return false;
    }
    LOGW("ShardedVoxelCarver: %s; every slab builds its own SDF images\n", vcy_last_error());
  }
  // one host thread per slab: a context is single-threaded, different contexts are independent.
  // vcy_last_error() is per thread: the worker hands its message back with the status.
  std::vector<std::future<std::pair<int, std::string>>> jobs;
  for (vcy_ctx* ctx : impl_->slabs)
    jobs.push_back(std::async(std::launch::async, [ctx, n, &views, &masks]() {
      // one view: queued by the library, carved together with the following calls (vcy_set_param "defer")
      const int rc = n == 1 ? vcy_carve_silhouette(ctx, &views[0], masks[0], nullptr)
                            : vcy_carve_batch_silhouettes(ctx, n, views.data(), masks.data());
      return std::make_pair(rc, rc == VCY_OK ? std::string() : std::string(vcy_last_error()));
    }));
  bool ok = true;
  for (auto& j : jobs) {
    const std::pair<int, std::string> r = j.get();
    if (r.first != VCY_OK) {
      LOGE("sharded carve failed: %s\n", r.second.c_str());
      ok = false;
    }
  }
  return ok;
}

// the two slices below every slab: ONE RCCL all-gather over the devices that hold slabs
// (vcy_halo_allgather); peer-to-peer copies only when asked for (set_halo_transport)
bool ShardedVoxelCarver::ExchangeHalo() {
  const size_t ns = impl_->slabs.size();
  if (impl_->peer_copy_halo) {
    for (size_t s = 0; s < ns; ++s)
      if (vcy_halo_copy_from(impl_->slabs[s], s ? impl_->slabs[s - 1] : nullptr) != VCY_OK) {
        LOGE("%s\n", vcy_last_error());
        return false;
      }
  } else if (vcy_halo_allgather(impl_->slabs.data(), static_cast<int>(ns)) != VCY_OK) {
    LOGE("%s\n", vcy_last_error());
    return false;
  }
  return true;
}

void ShardedVoxelCarver::ExtractVoxel(Mesh* mesh, bool inside_empty) {
  mesh->Clear();
  const size_t ns = impl_->slabs.size();
  if (ns == 0) return;
  // UpdateOnSurface (extract_voxel.cc:15-79) compares a voxel with its -z neighbour: the slice below a slab
  if (inside_empty && !ExchangeHalo()) return;
  std::vector<int64_t*> ids(ns, nullptr);
  std::vector<int64_t> counts(ns, 0);
  std::vector<std::string> errors(ns);
  std::vector<std::future<int>> jobs;
  for (size_t s = 0; s < ns; ++s)
    jobs.push_back(std::async(std::launch::async, [this, s, inside_empty, &ids, &counts, &errors]() {
      const int rc = vcy_extract_voxel_ids(impl_->slabs[s], inside_empty ? 1 : 0, &ids[s], &counts[s]);
      if (rc != VCY_OK) errors[s] = vcy_last_error();
      return rc;
    }));
  bool ok = true;
  for (auto& j : jobs) ok = (j.get() == VCY_OK) && ok;
  for (const std::string& e : errors)
    if (!e.empty()) LOGE("sharded ExtractVoxel failed: %s\n", e.c_str());
  if (ok) {
    // the kept voxels of the whole grid in scan order = the slabs' lists in z order; ONE cube drifts through them
    std::vector<int64_t> all;
    for (size_t s = 0; s < ns; ++s) all.insert(all.end(), ids[s], ids[s] + counts[s]);
    const vcy_carver_option c = ToC(impl_->option);
    typedef detail::MeshArrays<Eigen::Vector3f, Eigen::Vector3i> Arrays;
    Arrays arrays{mesh->mutable_vertices(), mesh->mutable_vertex_indices()};
    if (vcy_voxel_cubes_into(&c, static_cast<int64_t>(all.size()), all.data(), &Arrays::Provide, &arrays) != VCY_OK) {
      LOGE("%s\n", vcy_last_error());
      mesh->Clear();
    }
  }
  for (int64_t* p : ids) vcy_ids_free(p);
}

void ShardedVoxelCarver::ExtractIsoSurface(Mesh* mesh, double iso_level, bool linear_interp) {
  mesh->Clear();
  const size_t ns = impl_->slabs.size();
  if (ns == 0) return;
  if (!ExchangeHalo()) return;
  std::vector<vcy_mesh> parts(ns);
  std::vector<std::string> errors(ns);
  std::vector<std::future<int>> jobs;
  for (size_t s = 0; s < ns; ++s)
    jobs.push_back(std::async(std::launch::async, [this, s, iso_level, linear_interp, &parts, &errors]() {
      const int rc = vcy_extract_iso(impl_->slabs[s], iso_level, linear_interp ? 1 : 0, &parts[s]);
      if (rc != VCY_OK) errors[s] = vcy_last_error();
      return rc;
    }));
  bool ok = true;
  for (auto& j : jobs) ok = (j.get() == VCY_OK) && ok;
  for (const std::string& e : errors)
    if (!e.empty()) LOGE("sharded extraction failed: %s\n", e.c_str());
  if (ok) {
    // stitch: a slab's first n_foreign vertices are owned by the slab below -> look them up by edge key
    std::vector<Eigen::Vector3f>* V = mesh->mutable_vertices();
    std::vector<Eigen::Vector3i>* F = mesh->mutable_vertex_indices();
    std::unordered_map<std::pair<int64_t, int64_t>, int, KeyHash> prev;
    int64_t offset = 0;
    {
      size_t total_v = 0, total_f = 0;
      for (const vcy_mesh& m : parts) total_v += static_cast<size_t>(m.n_vertices - m.n_foreign_vertices), total_f += static_cast<size_t>(m.n_faces);
      V->reserve(total_v);
      F->reserve(total_f);
    }
    for (size_t s = 0; s < ns && ok; ++s) {
      const vcy_mesh& m = parts[s];
      const int64_t nfo = m.n_foreign_vertices, nown = m.n_vertices - nfo;
      std::vector<int> remap(static_cast<size_t>(m.n_vertices));
      for (int64_t i = 0; i < nfo; ++i) {
        auto it = prev.find({m.edge_keys[2 * i], m.edge_keys[2 * i + 1]});
        if (it == prev.end()) {
          LOGE("sharded merge: shared-plane vertex without an owner\n");
          ok = false;
          break;
        }
        remap[i] = it->second;
      }
      for (int64_t i = 0; i < nown; ++i) remap[nfo + i] = static_cast<int>(offset + i);
      const size_t v0 = V->size(), f0 = F->size();
      detail::CopyTriples(V, m.vertices + 3 * nfo, static_cast<size_t>(nown), v0);
      F->resize(f0 + m.n_faces);
      for (int64_t i = 0; i < m.n_faces; ++i)
        (*F)[f0 + i] = Eigen::Vector3i(remap[m.faces[3 * i]], remap[m.faces[3 * i + 1]], remap[m.faces[3 * i + 2]]);
      prev.clear();
      if (s + 1 < ns) {
        // Only vertices on this slab's top plane can be referenced from above: edges with both voxels in the slice below
        // the next slab (vcy_mesh::n_foreign_vertices; keys are (lower, higher) GLOBAL voxel ids).  Entering every vertex
        // made the merge 35 ms for a 600 K-vertex mesh in 8 slabs, where the extraction itself takes 2.
        const int64_t lo = static_cast<int64_t>(impl_->bounds[s + 1] - 1) * impl_->slice, hi = lo + impl_->slice;
        for (int64_t i = 0; i < nown; ++i) {
          const int64_t k0 = m.edge_keys[2 * (nfo + i)], k1 = m.edge_keys[2 * (nfo + i) + 1];
          if (k0 >= lo && k1 < hi) prev[{k0, k1}] = static_cast<int>(offset + i);
        }
      }
      offset += nown;
    }
  }
  if (!ok) mesh->Clear();
  for (vcy_mesh& m : parts) vcy_mesh_free(&m);
}

}  // namespace vacancy
