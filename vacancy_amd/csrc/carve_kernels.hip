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

#include "vcy_internal.h"

namespace vcy {

struct ViewParams {
  float r[3][3];   // w2c rotation, row-major
  float t[3];
  float fx, fy, cx, cy;
  float roi_min_x, roi_min_y, roi_max_x, roi_max_y;  // (float)int, as the reference's int->float compare
  int roi_min_xi, roi_min_yi, roi_max_xi, roi_max_yi;
  int width;
  float max_sdf;
  const float* sdf;
};

struct GridParams {
  float* sdf;
  void* cnt;
  const float* px;
  const float* py;
  const float* pz;
  int nx, ny;
  int z0;        // global z of local slice 0
  int nz_local;
  int max_update_num;
  float weight;
};

struct ModeParams {
  int update, interp, outside, trunc, ortho;
};

// ---- sampling, shared by every carve kernel --------------------------------------------

__device__ __forceinline__ float tap(const float* __restrict__ s, int width, int x, int y) {
  return s[(int64_t)width * y + x];
}

// Returns false when the voxel must be skipped for this view.
template <bool RT, int INTERP, int OUTSIDE, bool TRUNC, bool ORTHO>
__device__ __forceinline__ bool view_distance(const ViewParams& v, const ModeParams& m, float px,
                                              float py, float pz, float* dist_out) {
  const int interp = RT ? m.interp : INTERP;
  const int outside = RT ? m.outside : OUTSIDE;
  const bool trunc = RT ? (m.trunc != 0) : TRUNC;
  const bool ortho = RT ? (m.ortho != 0) : ORTHO;

  float pc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float c0 = v.r[i][0] * px;
    const float c1 = v.r[i][1] * py;
    const float c2 = v.r[i][2] * pz;
    pc[i] = v.t[i] + (c0 + (c1 + c2));
  }
  if (pc[2] < 0.0f) return false;
  float u, w;
  if (ortho) {
    u = pc[0];
    w = pc[1];
  } else {
    u = v.fx / pc[2] * pc[0] + v.cx;
    w = v.fy / pc[2] * pc[1] + v.cy;
  }
  // NaN image coordinates (pc.z == 0 and pc.x|y == 0) are undefined in the reference;
  // skipped here and in the oracle.
  if (u != u || w != w) return false;

  float dist;
  if (u < v.roi_min_x || w < v.roi_min_y || v.roi_max_x < u || v.roi_max_y < w) {
    if (outside == VCY_OUTSIDE_NONE) return false;
    dist = v.max_sdf;
  } else if (interp == VCY_INTERP_NN) {
    int xi = (int)roundf(u);
    int yi = (int)roundf(w);
    xi = max(xi, v.roi_min_xi);
    yi = max(yi, v.roi_min_yi);
    xi = min(xi, v.roi_max_xi);
    yi = min(yi, v.roi_max_yi);
    dist = tap(v.sdf, v.width, xi, yi);
  } else {
    const float fu = floorf(u), fw = floorf(w);
    int x0 = (int)fu, y0 = (int)fw;
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = max(x0, v.roi_min_xi);
    y0 = max(y0, v.roi_min_yi);
    x1 = min(x1, v.roi_max_xi);
    y1 = min(y1, v.roi_max_yi);
    const float lu = u - (float)x0;
    const float lv = w - (float)y0;
    const float s00 = tap(v.sdf, v.width, x0, y0);
    const float s10 = tap(v.sdf, v.width, x1, y0);
    const float s01 = tap(v.sdf, v.width, x0, y1);
    const float s11 = tap(v.sdf, v.width, x1, y1);
    const float a = (1.0f - lu) * (1.0f - lv) * s00;
    const float b = lu * (1.0f - lv) * s10;
    const float c = (1.0f - lu) * lv * s01;
    const float d = lu * lv * s11;
    dist = ((a + b) + c) + d;
  }
  if (trunc && dist < -1.0f) return false;
  *dist_out = dist;
  return true;
}

// Applies one sample to the voxel state held in registers.  Returns true if it changed.
template <bool RT, int UPDATE>
__device__ __forceinline__ bool fuse(const ModeParams& m, float weight, float dist, float& sdf,
                                     int& n) {
  const int update = RT ? m.update : UPDATE;
  if (n < 1) {  // first touch, voxel_carver.cc:482-486
    sdf = dist;
    n = 1;
    return true;
  }
  if (update == VCY_UPDATE_MAX) {  // UpdateVoxelMax, :78-86
    if (dist > sdf) {
      sdf = dist;
      n = n + 1;
      return true;
    }
    return false;
  }
  // UpdateVoxelWeightedAverage, :88-95
  const float inv_denom = 1.0f / (weight * (float)(n + 1));
  sdf = (weight * (float)n * sdf + weight * dist) * inv_denom;
  n = n + 1;
  return true;
}

// ---- kernel A: one thread per voxel, one view per launch (generic, every mode) ---------
//
// Grid: 1-D over (row, x-segment); a row is a (y, z) line of nx voxels, x fastest, so a
// wave reads 64 consecutive floats of the slab (256 B) and the SDF taps of neighbouring
// lanes fall in the same few cache lines.
template <typename CountT, bool RT, int UPDATE, int INTERP, int OUTSIDE, bool TRUNC, bool ORTHO>
__global__ __launch_bounds__(256) void carve_view_kernel(GridParams g, ViewParams v, ModeParams m,
                                                         int segs_per_row) {
  const int64_t row = blockIdx.x / segs_per_row;
  const int seg = blockIdx.x - (int)(row * segs_per_row);
  const int x = seg * 256 + threadIdx.x;
  if (x >= g.nx) return;
  const int zl = (int)(row / g.ny);
  const int y = (int)(row - (int64_t)zl * g.ny);
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
__global__ void max_reduce_kernel(const float* __restrict__ p, int64_t n, float* out) {
  __shared__ float sm[256];
  float m = -INFINITY;
  for (int64_t i = threadIdx.x; i < n; i += 256) m = fmaxf(m, p[i]);
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sm[threadIdx.x] = fmaxf(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sm[0];
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
  v->max_sdf = max_sdf;
  v->sdf = sdf_dev;
}

template <typename CountT>
static void launch_view(vcy_ctx* c, const GridParams& g, const ViewParams& v, const ModeParams& m) {
  const int segs = (c->nx + 255) / 256;
  const int64_t rows = (int64_t)c->ny * c->nz_local();
  const dim3 grid((unsigned)(rows * segs)), block(256);
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

int launch_carve(vcy_ctx* c, int n_views, const vcy_view* views, const float* const* sdf_dev) {
  const vcy_update_option& u = c->opt.update_option;
  GridParams g;
  g.sdf = c->owned_slab_sdf();
  g.cnt = c->owned_slab_cnt();
  g.px = c->d_px;
  g.py = c->d_py;
  g.pz = c->d_pz;
  g.nx = c->nx;
  g.ny = c->ny;
  g.z0 = c->z0;
  g.nz_local = c->nz_local();
  g.max_update_num = u.voxel_max_update_num;
  g.weight = u.voxel_update_weight;
  if ((int64_t)c->ny * c->nz_local() * ((c->nx + 255) / 256) > 0x7fffffffLL) {
    set_error("slab too large for one launch");
    return VCY_ERR_TOO_MANY_VOXELS;
  }

  float* d_max = nullptr;
  if (u.update_outside == VCY_OUTSIDE_MAX) VCY_HIP_CHECK(hipMalloc(&d_max, sizeof(float)));

  for (int i = 0; i < n_views; ++i) {
    ModeParams m{u.voxel_update, u.sdf_interp, u.update_outside, u.use_truncation ? 1 : 0,
                 views[i].is_ortho ? 1 : 0};
    float max_sdf = 0.0f;
    if (d_max) {
      const int64_t npx = (int64_t)views[i].width * views[i].height;
      hipLaunchKernelGGL(max_reduce_kernel, dim3(1), dim3(256), 0, c->stream, sdf_dev[i], npx, d_max);
      VCY_HIP_CHECK(hipMemcpyAsync(&max_sdf, d_max, sizeof(float), hipMemcpyDeviceToHost, c->stream));
      VCY_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    ViewParams v;
    fill_view(views[i], sdf_dev[i], max_sdf, &v);
    if (c->cnt_bytes == 1) launch_view<uint8_t>(c, g, v, m);
    else if (c->cnt_bytes == 2) launch_view<uint16_t>(c, g, v, m);
    else launch_view<uint32_t>(c, g, v, m);
    VCY_HIP_CHECK(hipGetLastError());
    c->views_carved += 1;
  }
  c->halo_valid = false;
  if (d_max) {
    VCY_HIP_CHECK(hipStreamSynchronize(c->stream));
    VCY_HIP_CHECK(hipFree(d_max));
  }
  return VCY_OK;
}

}  // namespace vcy
