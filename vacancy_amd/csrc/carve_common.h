// Device code shared by the carve kernels: parameter blocks, the exact per-sample arithmetic
// (projection, ROI test, SDF sampling) and the voxel update rules.  Reference lines are cited
// at each step; float operation order is the reference's (no FMA contraction).
#pragma once

#include "vcy_internal.h"

namespace vcy {

struct ViewParams {
  float r[3][3];   // w2c rotation, row-major
  float t[3];
  float fx, fy, cx, cy;
  float roi_min_x, roi_min_y, roi_max_x, roi_max_y;  // (float)int, as the reference's int->float compare
  int roi_min_xi, roi_min_yi, roi_max_xi, roi_max_yi;
  int width, height;
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
  int div_level;  // host side only: which division sequence the fused kernel is instantiated with
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
  // Reference test (voxel_carver.cc:464-465): outside iff u < roi_min.x || v < roi_min.y ||
  // roi_max.x < u || roi_max.y < v.  Written as its complement so that NaN coordinates
  // (pc.z == 0 with pc.x|y == 0; undefined behaviour in the reference) count as outside.
  const bool inside = u >= v.roi_min_x && w >= v.roi_min_y && u <= v.roi_max_x && w <= v.roi_max_y;
  float dist;
  if (!inside) {
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


}  // namespace vcy
