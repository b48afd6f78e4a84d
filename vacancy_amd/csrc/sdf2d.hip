// 2-D silhouette SDF builder -- the step in front of the carve kernel.
//
// Replaces DistanceTransformL1 / MakeSignedDistanceField (reference
// src/vacancy/voxel_carver.cc:102-237).  The reference runs a two-pass 4-neighbour chamfer
// raster scan, which is an exact L1 (city-block) distance transform restricted to the ROI.
// L1 distance is separable, so this implementation does it as two independent 1-D min-plus
// sweeps in integers (rows, then columns) -- every row/column is independent work (the
// shape a GPU wants) and the integer result converts to exactly the
// floats the reference's `+1.0f` chain produces.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "vcy_internal.h"

namespace vcy {

namespace {

constexpr int kInf = 1 << 29;  // > any in-image L1 distance, and kInf + small never overflows

// dist[y][x] = L1 distance, inside the ROI, from (x,y) to the nearest ROI pixel with
// seed(x,y) == true; kInf when the ROI holds no seed.  `seed_is_255` selects which pixels
// are sources: false -> mask != 255 are sources (distance measured inside the silhouette),
// true -> mask == 255 are sources.
void l1_distance_int(const uint8_t* mask, int w, const int32_t* rmin, const int32_t* rmax,
                     bool seed_is_255, std::vector<int>* out) {
  const int x0 = rmin[0], y0 = rmin[1], x1 = rmax[0], y1 = rmax[1];
  const int rw = x1 - x0 + 1, rh = y1 - y0 + 1;
  std::vector<int>& d = *out;
  d.assign((size_t)rw * rh, kInf);
  // rows
  for (int y = 0; y < rh; ++y) {
    const uint8_t* m = mask + (size_t)(y + y0) * w + x0;
    int* r = d.data() + (size_t)y * rw;
    int run = kInf;
    for (int x = 0; x < rw; ++x) {
      const bool seed = (m[x] == 255) == seed_is_255;
      run = seed ? 0 : std::min(run + 1, kInf);
      r[x] = run;
    }
    run = kInf;
    for (int x = rw - 1; x >= 0; --x) {
      run = std::min(r[x], std::min(run + 1, kInf));
      r[x] = run;
    }
  }
  // columns
  for (int y = 1; y < rh; ++y) {
    int* r = d.data() + (size_t)y * rw;
    const int* up = r - rw;
    for (int x = 0; x < rw; ++x) r[x] = std::min(r[x], std::min(up[x] + 1, kInf));
  }
  for (int y = rh - 2; y >= 0; --y) {
    int* r = d.data() + (size_t)y * rw;
    const int* dn = r + rw;
    for (int x = 0; x < rw; ++x) r[x] = std::min(r[x], std::min(dn[x] + 1, kInf));
  }
}

inline float to_float_dist(int v) {
  return v >= kInf ? std::numeric_limits<float>::max() : (float)v;
}

}  // namespace

void host_distance_transform_l1(const uint8_t* mask, int w, int h, const int32_t* rmin,
                                const int32_t* rmax, float* out) {
  // voxel_carver.cc:104: Init(width, height, 0.0f); pixels outside the ROI stay 0.
  std::fill(out, out + (size_t)w * h, 0.0f);
  std::vector<int> d;
  l1_distance_int(mask, w, rmin, rmax, /*seed_is_255=*/false, &d);
  const int rw = rmax[0] - rmin[0] + 1;
  for (int y = rmin[1]; y <= rmax[1]; ++y)
    for (int x = rmin[0]; x <= rmax[0]; ++x)
      out[(size_t)y * w + x] = to_float_dist(d[(size_t)(y - rmin[1]) * rw + (x - rmin[0])]);
}

void host_make_sdf(const uint8_t* mask, int w, int h, const int32_t* rmin, const int32_t* rmax,
                   bool normalize, bool truncate, float band, float* out) {
  const int rw = rmax[0] - rmin[0] + 1;
  std::fill(out, out + (size_t)w * h, 0.0f);
  std::vector<int> din, dout;
  l1_distance_int(mask, w, rmin, rmax, false, &din);   // inside: distance to the nearest non-255
  l1_distance_int(mask, w, rmin, rmax, true, &dout);   // outside: distance to the nearest 255
  float mx = 0.0f, mn = 0.0f;  // pixels outside the ROI are 0 and take part in min/max (:205-212)
  bool roi_is_everything = rmin[0] == 0 && rmin[1] == 0 && rmax[0] == w - 1 && rmax[1] == h - 1;
  bool first = roi_is_everything;
  for (int y = rmin[1]; y <= rmax[1]; ++y) {
    for (int x = rmin[0]; x <= rmax[0]; ++x) {
      const size_t r = (size_t)(y - rmin[1]) * rw + (x - rmin[0]);
      float v;
      if (mask[(size_t)y * w + x] == 255) {
        v = to_float_dist(din[r]);
        if (v > 0) v *= -1;  // :176-182
      } else {
        v = to_float_dist(dout[r]);  // :197-203
      }
      out[(size_t)y * w + x] = v;
      if (first) {
        mx = mn = v;
        first = false;
      } else {
        mx = std::max(mx, v);
        mn = std::min(mn, v);
      }
    }
  }
  if (normalize) {  // :205-222
    const float abs_max = std::max(std::abs(mx), std::abs(mn));
    if (abs_max > std::numeric_limits<float>::min()) {
      const float norm = 1.0f / abs_max;
      for (int y = rmin[1]; y <= rmax[1]; ++y)
        for (int x = rmin[0]; x <= rmax[0]; ++x) out[(size_t)y * w + x] *= norm;
    }
  }
  if (truncate) {  // :225-236
    for (int y = rmin[1]; y <= rmax[1]; ++y)
      for (int x = rmin[0]; x <= rmax[0]; ++x) {
        float& d = out[(size_t)y * w + x];
        if (-band >= d) d = kInvalidSdf;
        else d = std::min(1.0f, d / band);
      }
  }
}

}  // namespace vcy
