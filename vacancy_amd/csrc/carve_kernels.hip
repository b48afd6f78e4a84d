// K1: carve -- fuses one or more views into the device voxel slab.
//
// Replaces the main loop of VoxelCarver::Carve(camera, roi_min, roi_max, sdf)
// (reference src/vacancy/voxel_carver.cc:442-491).  Per voxel and view, in the reference's
// exact float operation order (no FMA contraction; IEEE divide):
//   pc  = t + (R0*px + (R1*py + R2*pz))          voxel_carver.cc:453 (Eigen Affine3f*Vector3f)
//   skip if pc.z < 0                              :456
//   u,v = (f/pc.z)*pc.xy + c   | pc.xy (ortho)    camera.cc:131-137 | :201-205
//   outside ROI -> skip | dist = max_sdf          :464-472
//   dist = bilinear | nearest sample              :16-76
//   skip if use_truncation && dist < -1           :478
//   first touch / kMax / kWeightedAverage         :482-488, :78-95
//
// Memory-bound path, no MFMA.  Algorithmic traffic: 4 B/voxel/view (kMax: the fp32 sdf;
// update_num is only touched when the voxel changes) or 4+cnt B (weighted average).
#include <algorithm>
#include <vector>

#include "carve_common.h"

namespace vcy {

// ---- kernel A: one thread per voxel, one view per launch (generic, every mode) ---------
//
// Grid: x over (y, x-segment), y over the slab's z slices (a 1-D grid of a 2048^3 slab would exceed
// the 2^32 threads a launch dimension may hold); a row is a (y, z) line of nx voxels, x fastest, so
// a wave reads 64 consecutive floats of the slab (256 B) and the SDF taps of neighbouring lanes fall
// in the same few cache lines.
template <typename CountT, bool RT, int UPDATE, int INTERP, int OUTSIDE, bool TRUNC, bool ORTHO>
__global__ __launch_bounds__(256) void carve_view_kernel(GridParams g, ViewParams v, ModeParams m,
                                                         int segs_per_row) {
  const int y = blockIdx.x / segs_per_row;
  const int seg = blockIdx.x - y * segs_per_row;
  const int x = seg * 256 + threadIdx.x;
  if (x >= g.nx) return;
  const int zl = blockIdx.y + blockIdx.z * 65535;
  if (zl >= g.nz_local) return;
  const int64_t row = (int64_t)zl * g.ny + y;
  const int64_t idx = row * g.nx + x;

  CountT* __restrict__ cnt = (CountT*)g.cnt;
  int n = (int)cnt[idx];
  if (n > g.max_update_num) return;  // voxel_carver.cc:447-450
  float dist;
  if (!view_distance<RT, INTERP, OUTSIDE, TRUNC, ORTHO>(v, m, g.px[x], g.py[y], g.pz[g.z0 + zl], &dist))
    return;
  float s = g.sdf[idx];
  if (fuse<RT, UPDATE>(m, g.weight, dist, s, n)) {
    g.sdf[idx] = s;
    cnt[idx] = (CountT)n;
  }
}

// max over the whole SDF buffer (voxel_carver.cc:436), only needed for update_outside=kMax
// (gridDim.x blocks per image, out[blockIdx.x] = maximum of that block's share; a second launch over the
// partial maxima finishes -- one block per image took 0.3 ms per 1280 x 720 image)
constexpr int kMaxReduceBlocks = 64;
__global__ void max_reduce_kernel(const float* __restrict__ p, int64_t n, float* out) {
  __shared__ float sm[256];
  float m = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, p[i]);
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = sm[0];
}

static void fill_view(const vcy_view& in, const float* sdf_dev, float max_sdf, ViewParams* v) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) v->r[i][j] = in.w2c[4 * i + j];
    v->t[i] = in.w2c[4 * i + 3];
  }
  v->fx = in.fx;
  v->fy = in.fy;
  v->cx = in.cx;
  v->cy = in.cy;
  v->roi_min_xi = in.roi_min[0];
  v->roi_min_yi = in.roi_min[1];
  v->roi_max_xi = in.roi_max[0];
  v->roi_max_yi = in.roi_max[1];
  v->roi_min_x = (float)in.roi_min[0];
  v->roi_min_y = (float)in.roi_min[1];
  v->roi_max_x = (float)in.roi_max[0];
  v->roi_max_y = (float)in.roi_max[1];
  v->width = in.width;
  v->height = in.height;
  v->max_sdf = max_sdf;
  v->sdf = sdf_dev;
}

template <typename CountT>
static void launch_view(vcy_ctx* c, const GridParams& g, const ViewParams& v, const ModeParams& m) {
  const int segs = (c->nx + 255) / 256;
  const int nzl = c->nz_local();
  const dim3 grid((unsigned)(c->ny * segs), (unsigned)std::min(nzl, 65535), (unsigned)((nzl + 65534) / 65535)), block(256);
  const bool is_default = m.update == VCY_UPDATE_MAX && m.interp == VCY_INTERP_BILINEAR &&
                          m.outside == VCY_OUTSIDE_NONE && !m.trunc && !m.ortho;
  const bool is_tsdf = m.update == VCY_UPDATE_WEIGHTED_AVERAGE && m.interp == VCY_INTERP_BILINEAR &&
                       m.outside == VCY_OUTSIDE_NONE && m.trunc && !m.ortho;
  if (is_default) {
    hipLaunchKernelGGL((carve_view_kernel<CountT, false, VCY_UPDATE_MAX, VCY_INTERP_BILINEAR,
                                          VCY_OUTSIDE_NONE, false, false>),
                       grid, block, 0, c->stream, g, v, m, segs);
  } else if (is_tsdf) {
    hipLaunchKernelGGL((carve_view_kernel<CountT, false, VCY_UPDATE_WEIGHTED_AVERAGE,
                                          VCY_INTERP_BILINEAR, VCY_OUTSIDE_NONE, true, false>),
                       grid, block, 0, c->stream, g, v, m, segs);
  } else {
    hipLaunchKernelGGL((carve_view_kernel<CountT, true, 0, 0, 0, false, false>), grid, block, 0,
                       c->stream, g, v, m, segs);
  }
}

// The views as the kernels read them: max over the whole SDF buffer resolved (voxel_carver.cc:436; only
// update_outside = kMax reads it).
static int resolve_views(vcy_ctx* c, int n_views, const vcy_view* views, const float* const* sdf_dev,
                         std::vector<ViewParams>* out) {
  const vcy_update_option& u = c->opt.update_option;
  // max over the whole SDF buffer (voxel_carver.cc:436); only update_outside = kMax reads it
  std::vector<float> max_sdf((size_t)n_views, 0.0f);
  if (u.update_outside == VCY_OUTSIDE_MAX) {
    float* d_max = nullptr;
    VCY_HIP_CHECK(hipMalloc(&d_max, sizeof(float) * (size_t)n_views * (1 + kMaxReduceBlocks)));
    float* d_part = d_max + n_views;
    for (int i = 0; i < n_views; ++i) {
      const int64_t npx = (int64_t)views[i].width * views[i].height;
      hipLaunchKernelGGL(max_reduce_kernel, dim3(kMaxReduceBlocks), dim3(256), 0, c->stream, sdf_dev[i], npx,
                         d_part + (size_t)i * kMaxReduceBlocks);
      hipLaunchKernelGGL(max_reduce_kernel, dim3(1), dim3(256), 0, c->stream, d_part + (size_t)i * kMaxReduceBlocks,
                         (int64_t)kMaxReduceBlocks, d_max + i);
    }
    hipError_t e = hipMemcpyAsync(max_sdf.data(), d_max, sizeof(float) * (size_t)n_views,
                                  hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_max);
    if (e != hipSuccess) {
      set_error("max reduce failed: %s", hipGetErrorString(e));
      return VCY_ERR_HIP;
    }
  }
  out->resize((size_t)n_views);
  for (int i = 0; i < n_views; ++i) fill_view(views[i], sdf_dev[i], max_sdf[i], &(*out)[(size_t)i]);
  return VCY_OK;
}


// Applies the views queued by the per-view entry points, in order, as one batch.  `from_carve`: the caller is
// a carve entry point, whose own return value carries a failure to the `if (!Carve())` of the host loop; any
// other caller (an extraction, a download ...) cannot, so the failure is kept for the next carve call.
int flush_pending(vcy_ctx* c, bool from_carve) {
  if (c->pending.empty()) return VCY_OK;
  std::vector<vcy_ctx::PendingView> todo;
  todo.swap(c->pending);  // launch_carve flushes first: nothing left to recurse on
  std::vector<vcy_view> views(todo.size());
  std::vector<const float*> ptrs(todo.size());
  for (size_t i = 0; i < todo.size(); ++i) {
    views[i] = todo[i].view;
    ptrs[i] = todo[i].d_sdf;
  }
  VCY_HIP_CHECK(hipSetDevice(c->device));
  // the halo slices were invalidated when the views were queued; halos installed since then were taken
  // from a neighbour that has applied the same views and stay valid
  const bool halo_valid = c->halo_valid;
  const int rc = launch_carve(c, (int)todo.size(), views.data(), ptrs.data());
  if (rc != VCY_OK && !from_carve) {  // also reported by the next carve call (vacancy_hip.h, vcy_carve)
    c->deferred_rc = rc;
    c->deferred_msg = vcy_last_error();
  }
  c->halo_valid = halo_valid;
  // stream order: a buffer handed out again is only written after this launch
  for (auto& t : todo) c->sdf_pool.emplace_back(t.d_sdf, t.bytes);
  // image sizes that keep changing would let idle buffers pile up: keep at most two queues' worth
  // (hipFree waits for the device, so a buffer still read by the launch above is safe to free)
  while (c->sdf_pool.size() > 64) {
    (void)hipFree(c->sdf_pool.front().first);
    c->sdf_pool.erase(c->sdf_pool.begin());
  }
  return rc;
}

int launch_carve(vcy_ctx* c, int n_views, const vcy_view* views, const float* const* sdf_dev) {
  {
    const int rcf = flush_pending(c, true);  // earlier per-view calls come first
    if (rcf != VCY_OK) return rcf;
  }
  if (c->inject_fail > 0) {  // test hook (vcy_set_param "inject_carve_failure"): this application of views fails
    --c->inject_fail;
    set_error("injected failure (vcy_set_param inject_carve_failure)");
    return VCY_ERR_INTERNAL;
  }
  const vcy_update_option& u = c->opt.update_option;
  GridParams g;
  g.sdf = c->owned_slab_sdf();
  g.cnt = nullptr;  // set per launch below: the counter array may be widened between two chunks (ensure_count_width)
  g.px = c->d_px;
  g.py = c->d_py;
  g.pz = c->d_pz;
  g.nx = c->nx;
  g.ny = c->ny;
  g.z0 = c->z0;
  g.nz_local = c->nz_local();
  g.max_update_num = u.voxel_max_update_num;
  g.weight = u.voxel_update_weight;
  if ((int64_t)c->ny * ((c->nx + 255) / 256) > 0xffffffLL) {
    set_error("slab too large for one launch");
    return VCY_ERR_TOO_MANY_VOXELS;
  }

  std::vector<ViewParams> vp;
  {
    const int rcv = resolve_views(c, n_views, views, sdf_dev, &vp);
    if (rcv != VCY_OK) return rcv;
  }

  const bool fused = c->use_fused && fused_eligible(c, n_views, views);
  if (!fused) {
    int rcm = materialize(c);
    if (rcm != VCY_OK) return rcm;
  }
  if (fused) {
    c->fused_ortho = views[0].is_ortho != 0;
    const int chunk = fused_max_views();
    for (int i = 0; i < n_views; i += chunk) {
      const int m = std::min(chunk, n_views - i);
      int rc = ensure_count_width(c, c->views_carved + m);  // (update_num <= views applied: u8 up to the 255th view)
      if (rc != VCY_OK) return rc;
      g.cnt = c->owned_slab_cnt();
      rc = launch_carve_fused(c, g, m, &vp[i]);
      if (rc != VCY_OK) return rc;
      c->views_carved += m;
    }
  } else {
    c->brick_min_valid = false;  // (the per-view kernel does not keep the brick minima)
    for (int i = 0; i < n_views; ++i) {
      ModeParams m{u.voxel_update, u.sdf_interp, u.update_outside, u.use_truncation ? 1 : 0,
                   views[i].is_ortho ? 1 : 0, 0};
      { const int rcw = ensure_count_width(c, c->views_carved + 1); if (rcw != VCY_OK) return rcw; }
      g.cnt = c->owned_slab_cnt();
      if (c->cnt_bytes == 1) launch_view<uint8_t>(c, g, vp[i], m);
      else if (c->cnt_bytes == 2) launch_view<uint16_t>(c, g, vp[i], m);
      else launch_view<uint32_t>(c, g, vp[i], m);
      VCY_HIP_CHECK(hipGetLastError());
      c->views_carved += 1;
    }
  }
  c->halo_valid = false;
  return VCY_OK;
}

// The contiguous partition of L brick layers into n_slabs parts (every part at least one layer) that minimises the
// largest part's cost and, among those, the sum of the squares: dynamic programming over (parts, layers), O(n_slabs L^2)
// -- L is nz / 8, a few hundred.  z_bounds[0 .. n_slabs] in slices (layer * 8; the last one nz).  Host arithmetic only
// (vcy_partition_layers exposes it: the CPU tests check it against brute force).  Every slab must hold at least two
// slices (the halo exchange sends a slab's last two): when the last layer is a single slice (nz % 8 == 1) the last part
// takes at least two layers -- the caller guarantees n_slabs <= L - 1 then.
void partition_layers(const double* cost, int L, int n_slabs, int nz, int32_t* z_bounds) {
  const bool tail_single = nz - (L - 1) * 8 == 1 && L > 1;
  std::vector<double> pre((size_t)L + 1, 0.0);
  for (int l = 0; l < L; ++l) pre[(size_t)l + 1] = pre[(size_t)l] + cost[l];
  // f[s][i]: best (largest part, sum of squares) for the first i layers in s parts
  struct Val { double mx, sq; };
  auto better = [](const Val& a, const Val& b) { return a.mx < b.mx * (1.0 - 1e-12) || (a.mx <= b.mx * (1.0 + 1e-12) && a.sq < b.sq); };
  const Val inf{1e300, 1e300};
  std::vector<std::vector<Val>> f((size_t)n_slabs + 1, std::vector<Val>((size_t)L + 1, inf));
  std::vector<std::vector<int>> from((size_t)n_slabs + 1, std::vector<int>((size_t)L + 1, -1));
  f[0][0] = Val{0.0, 0.0};
  for (int sidx = 1; sidx <= n_slabs; ++sidx)
    for (int i = sidx; i <= L - (n_slabs - sidx); ++i)
      for (int j = sidx - 1; j < i; ++j) {
        if (f[(size_t)sidx - 1][(size_t)j].mx >= 1e299) continue;
        if (tail_single && i == L && j == L - 1) continue;  // (a last part of one slice)
        const double part = pre[(size_t)i] - pre[(size_t)j];
        const Val v{std::max(f[(size_t)sidx - 1][(size_t)j].mx, part), f[(size_t)sidx - 1][(size_t)j].sq + part * part};
        if (better(v, f[(size_t)sidx][(size_t)i])) {
          f[(size_t)sidx][(size_t)i] = v;
          from[(size_t)sidx][(size_t)i] = j;
        }
      }
  int i = L;
  z_bounds[n_slabs] = nz;
  for (int sidx = n_slabs; sidx >= 1; --sidx) {
    i = from[(size_t)sidx][(size_t)i];
    z_bounds[sidx - 1] = i * 8;
  }
}

// Cuts the grid into n_slabs z-slabs of equal PREDICTED carve cost for these views (vcy_plan_z_slabs).  Boundaries are
// whole brick layers (8 slices): a slab that ends inside a layer pays that layer's waves in full, and so does the next.
// cost(layer) = brick_cost * bricks + estimated (brick, view) pairs processed (plan_layer_pairs); the contiguous
// partition that minimises the largest part -- and, among those, the sum of squares -- by dynamic programming.
// A wave's fixed work (record, state, write-back) in units of one ESTIMATED processed view: a least-squares fit of
// step time = fixed + a * bricks + b * estimated pairs over 23 slabs of 64 ... 1024 slices of the benchmark scene at
// steady clocks gives a / b = 2.54 (2.11 against the pairs really processed; the estimate runs 12 % high), a fixed
// 0.04 ms per launch, and the model within 2.5 % of every slab (profiles/r04/planner_fit.txt).
constexpr float kPlanBrickCost = 2.55f;
int plan_z_slabs(vcy_ctx* c, int n_views, const vcy_view* views, const float* const* sdf_dev, int n_slabs, int stride,
                 float brick_cost, int32_t* z_bounds, double* layer_cost, int max_layers, int* n_layers) {
  const int L = (c->nz + 7) / 8;
  const int usable = (c->nz - (L - 1) * 8 == 1 && L > 1) ? L - 1 : L;  // (a one-slice last layer cannot be a slab)
  if (n_slabs < 1 || n_slabs > usable || (n_slabs > 1 && c->nz < 2 * n_slabs)) {
    set_error("cannot cut %d brick layers (%d slices) into %d slabs of whole layers and at least 2 slices", L, c->nz,
              n_slabs);
    return VCY_ERR_INVALID_ARG;
  }
  std::vector<double> cost((size_t)L, 1.0);
  if (fused_eligible(c, n_views, views) && c->use_fused && n_views <= fused_max_views()) {
    std::vector<ViewParams> vp;
    int rc = resolve_views(c, n_views, views, sdf_dev, &vp);
    if (rc != VCY_OK) return rc;
    c->fused_ortho = views[0].is_ortho != 0;
    std::vector<double> pairs;
    int64_t bricks = 0;
    rc = plan_layer_pairs(c, n_views, vp.data(), stride > 0 ? stride : 2, &pairs, &bricks);
    if (rc != VCY_OK) return rc;
    const double bc = brick_cost > 0.0f ? brick_cost : kPlanBrickCost;
    for (int l = 0; l < L; ++l) cost[(size_t)l] = bc * (double)bricks + pairs[(size_t)l];
    // (a last layer of fewer than 8 slices costs its waves in full: nothing to scale)
  }  // (else: the per-view kernel, whose cost does not depend on the scene -- equal layers)
  if (layer_cost)
    for (int l = 0; l < std::min(L, max_layers); ++l) layer_cost[l] = cost[(size_t)l];
  if (n_layers) *n_layers = L;
  partition_layers(cost.data(), L, n_slabs, c->nz, z_bounds);
  return VCY_OK;
}

}  // namespace vcy
