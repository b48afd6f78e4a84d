/*
 * vacancy_oracle.cc -- CPU oracle for the voxel-carving hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see vacancy_oracle.h).  Build:
 *   g++ -std=c++17 -O2 -ffp-contract=off -fopenmp -shared -fPIC   (oracle/Makefile)
 * -ffp-contract=off is mandatory: the reference is plain -O2 x86-64 (no FMA), and a
 * fused multiply-add changes the bunny mesh.
 *
 * What this restates (all paths relative to /root/reference):
 *   src/vacancy/voxel_carver.cc:16-95    SDF sampling + voxel update rules
 *   src/vacancy/voxel_carver.cc:102-237  DistanceTransformL1 / MakeSignedDistanceField
 *   src/vacancy/voxel_carver.cc:276-345  VoxelGrid::Init
 *   src/vacancy/voxel_carver.cc:375-392  VoxelCarver::Init validation
 *   src/vacancy/voxel_carver.cc:415-496  VoxelCarver::Carve main loop
 *   src/vacancy/marching_cubes.cc:25-228 VertexInterp + MarchingCubes
 *   src/vacancy/camera.cc:39-42,114-137,201-205  w2c, fov, Project
 *   include/vacancy/common.h:51-75       look-at c2w
 *   examples.cc:36-50                    TUM pose -> Affine3d
 * The voxel record keeps the reference's 40-byte AoS layout
 * (include/vacancy/voxel_carver.h:62-72) so that, timed, it costs what the reference
 * costs on a CPU.
 *
 * PINNING.  The reference ships no tests and no golden vectors, and it cannot be
 * compiled in this image: include/vacancy/common.h:21 includes Eigen/Geometry and
 * third_party/eigen is an empty, un-vendored submodule (.gitmodules:4-6; version
 * unpinned).  oracle/_ref is therefore NOT built.  The oracle is pinned against the
 * known-answer table of SURVEY.md Appendix C (voxel counts, sums of sdf, sums of
 * update_num, marching-cubes vertex/face counts per view for four option modes, 2-D SDF
 * statistics, final vertex sums, w2c matrices), which the survey stage recorded from the
 * reference's own sources run on data/ bunny; tests/test_oracle_bunny.py checks every
 * row.  What stays UNPINNED by anything the reference provides is the internal
 * evaluation order of the four Eigen operations on the path (listed at their
 * restatements below); for those, parity rests on Eigen's published algorithm.
 */
#include "vacancy_oracle.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <utility>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// include/vacancy/voxel_carver.h:62-72 (Vector3i, int, Vector3f, float, int, bool, bool)
struct Voxel {
  int index[3] = {-1, -1, -1};
  int id = -1;
  float pos[3] = {0.0f, 0.0f, 0.0f};
  float sdf = 0.0f;
  int update_num = 0;
  bool outside = false;
  bool on_surface = false;
};
static_assert(sizeof(Voxel) == 40, "reference Voxel is 40 bytes");

// voxel_carver.cc:100
const float kInvalidSdf = std::numeric_limits<float>::lowest();

double now_ms() {
  using clk = std::chrono::steady_clock;
  return std::chrono::duration<double, std::milli>(clk::now().time_since_epoch()).count();
}

// marching_cubes_lut.cc:42-298, unpacked from the shared data file.
const char* const kCaseStrings[256] = {
#include "vacancy_mc_cases.inc"
};
struct McTables {
  int edge[256];
  int tri[256][16];
  McTables() {
    for (int c = 0; c < 256; ++c) {
      const char* s = kCaseStrings[c];
      int n = 0, used = 0;
      for (; s[n]; ++n) {
        int v = (s[n] <= '9') ? s[n] - '0' : s[n] - 'a' + 10;
        tri[c][n] = v;
        used |= 1 << v;
      }
      for (int k = n; k < 16; ++k) tri[c][k] = -1;
      edge[c] = used;  // kEdgeTable[c] == set of edges the triangles of case c use
    }
  }
};
const McTables& tables() {
  static McTables t;
  return t;
}

}  // namespace

struct orc_grid {
  std::vector<Voxel> voxels;
  float bb_max[3], bb_min[3];
  float resolution;
  int n[3];
  int xy;
  vcy_update_option opt;
  const Voxel& get(int x, int y, int z) const { return voxels[z * xy + (y * n[0] + x)]; }
  Voxel* get_ptr(int x, int y, int z) { return &voxels[z * xy + (y * n[0] + x)]; }
};

extern "C" {

int orc_sizeof_voxel(void) { return (int)sizeof(Voxel); }

// The tables MarchingCubes() below walks, as the reference lays them out (kEdgeTable[256],
// kTriTable[256][16], marching_cubes_lut.cc:15-298): pinned by tests/test_mc_tables.py against
// tests/golden/mc_tables.bin, which oracle/_ref dumped from the reference's own translation unit.
void orc_mc_tables(int* edge256, int* tri256x16) {
  const McTables& t = tables();
  for (int c = 0; c < 256; ++c) {
    edge256[c] = t.edge[c];
    for (int k = 0; k < 16; ++k) tri256x16[c * 16 + k] = t.tri[c][k];
  }
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int orc_omp_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

orc_grid* orc_grid_create(const vcy_carver_option* o) {
  // VoxelCarver::Init, voxel_carver.cc:375-389
  if (o->update_option.voxel_max_update_num < 1) return nullptr;
  if (o->update_option.voxel_update_weight < std::numeric_limits<float>::min()) return nullptr;
  if (o->update_option.truncation_band < std::numeric_limits<float>::min()) return nullptr;
  // VoxelGrid::Init, voxel_carver.cc:278-287
  if (o->resolution < std::numeric_limits<float>::min()) return nullptr;
  if (o->bb_max[0] <= o->bb_min[0] || o->bb_max[1] <= o->bb_min[1] ||
      o->bb_max[2] <= o->bb_min[2])
    return nullptr;

  orc_grid* g = new orc_grid;
  g->opt = o->update_option;
  for (int i = 0; i < 3; ++i) {
    g->bb_max[i] = o->bb_max[i];
    g->bb_min[i] = o->bb_min[i];
  }
  g->resolution = o->resolution;
  float diff[3];
  for (int i = 0; i < 3; ++i) {
    diff[i] = g->bb_max[i] - g->bb_min[i];                       // :292
    g->n[i] = static_cast<int>(diff[i] / g->resolution);         // :294-296
  }
  // :298-301 (the reference multiplies in int; do it in 64 bits and keep the limit)
  long long total = (long long)g->n[0] * g->n[1] * g->n[2];
  if (total > std::numeric_limits<int>::max() || total <= 0) {
    delete g;
    return nullptr;
  }
  g->xy = g->n[0] * g->n[1];
  g->voxels.resize((size_t)total);
  const float offset = g->resolution * 0.5f;                     // :308
  const int nx = g->n[0], ny = g->n[1], nz = g->n[2];
#pragma omp parallel for schedule(dynamic, 1)
  for (int z = 0; z < nz; z++) {
    float z_pos = diff[2] * (static_cast<float>(z) / static_cast<float>(nz)) + g->bb_min[2] + offset;
    for (int y = 0; y < ny; y++) {
      float y_pos = diff[1] * (static_cast<float>(y) / static_cast<float>(ny)) + g->bb_min[1] + offset;
      for (int x = 0; x < nx; x++) {
        float x_pos = diff[0] * (static_cast<float>(x) / static_cast<float>(nx)) + g->bb_min[0] + offset;
        Voxel* v = g->get_ptr(x, y, z);
        v->index[0] = x;
        v->index[1] = y;
        v->index[2] = z;
        v->id = z * g->xy + (y * nx + x);                        // :333
        v->pos[0] = x_pos;
        v->pos[1] = y_pos;
        v->pos[2] = z_pos;
        v->sdf = kInvalidSdf;                                    // :339
      }
    }
  }
  return g;
}

// A "grid" of arbitrary voxel centres (nx = n, ny = nz = 1): lets tests carve a SAMPLE of the
// voxels of a grid that is too large for the CPU (1024^3 and up) with the same loop.
orc_grid* orc_grid_from_positions(const vcy_update_option* uo, const float* pos, int n) {
  orc_grid* g = new orc_grid;
  g->opt = *uo;
  for (int i = 0; i < 3; ++i) g->bb_max[i] = g->bb_min[i] = 0.0f;
  g->resolution = 1.0f;
  g->n[0] = n;
  g->n[1] = g->n[2] = 1;
  g->xy = n;
  g->voxels.resize((size_t)n);
  for (int i = 0; i < n; ++i) {
    Voxel& v = g->voxels[i];
    v.index[0] = i;
    v.index[1] = v.index[2] = 0;
    v.id = i;
    for (int k = 0; k < 3; ++k) v.pos[k] = pos[3 * i + k];
    v.sdf = kInvalidSdf;
  }
  return g;
}

// Voxel::pos along one axis, voxel_carver.cc:308-326
void orc_axis_positions(float bb_min, float bb_max, float resolution, float* out, int* n_out) {
  const float diff = bb_max - bb_min;
  const int n = static_cast<int>(diff / resolution);
  const float offset = resolution * 0.5f;
  for (int i = 0; i < n; ++i) out[i] = diff * (static_cast<float>(i) / static_cast<float>(n)) + bb_min + offset;
  *n_out = n;
}

void orc_grid_destroy(orc_grid* g) { delete g; }

void orc_grid_dims(const orc_grid* g, int32_t dims[3]) {
  for (int i = 0; i < 3; ++i) dims[i] = g->n[i];
}

void orc_grid_download(const orc_grid* g, float* sdf, int32_t* update_num) {
  for (size_t i = 0; i < g->voxels.size(); ++i) {
    if (sdf) sdf[i] = g->voxels[i].sdf;
    if (update_num) update_num[i] = g->voxels[i].update_num;
  }
}

void orc_grid_upload(orc_grid* g, const float* sdf, const int32_t* update_num) {
  for (size_t i = 0; i < g->voxels.size(); ++i) {
    if (sdf) g->voxels[i].sdf = sdf[i];
    if (update_num) g->voxels[i].update_num = update_num[i];
  }
}

void orc_grid_positions(const orc_grid* g, float* pos) {
  for (size_t i = 0; i < g->voxels.size(); ++i)
    for (int k = 0; k < 3; ++k) pos[3 * i + k] = g->voxels[i].pos[k];
}

/* ---------------------------------------------------------------- carve -- */

namespace {

inline float sdf_at(const float* sdf, int width, int x, int y) {
  return sdf[width * y + x];  // image.h:65-74, one channel
}

// SdfInterpolationNn, voxel_carver.cc:16-38
inline float interp_nn(float u, float v, const float* sdf, int width, const int32_t* roi_min,
                       const int32_t* roi_max) {
  int xi = static_cast<int>(std::round(u));
  int yi = static_cast<int>(std::round(v));
  if (xi < roi_min[0]) xi = roi_min[0];
  if (yi < roi_min[1]) yi = roi_min[1];
  if (roi_max[0] < xi) xi = roi_max[0];
  if (roi_max[1] < yi) yi = roi_max[1];
  return sdf_at(sdf, width, xi, yi);
}

// SdfInterpolationBiliner, voxel_carver.cc:40-76
inline float interp_bilinear(float u, float v, const float* sdf, int width,
                             const int32_t* roi_min, const int32_t* roi_max) {
  int x0 = static_cast<int>(std::floor(u));
  int y0 = static_cast<int>(std::floor(v));
  int x1 = x0 + 1;
  int y1 = y0 + 1;
  if (x0 < roi_min[0]) x0 = roi_min[0];
  if (y0 < roi_min[1]) y0 = roi_min[1];
  if (roi_max[0] < x1) x1 = roi_max[0];
  if (roi_max[1] < y1) y1 = roi_max[1];
  float lu = u - x0;
  float lv = v - y0;
  float dist = (1.0f - lu) * (1.0f - lv) * sdf_at(sdf, width, x0, y0) +
               lu * (1.0f - lv) * sdf_at(sdf, width, x1, y0) +
               (1.0f - lu) * lv * sdf_at(sdf, width, x0, y1) +
               lu * lv * sdf_at(sdf, width, x1, y1);
  return dist;
}

}  // namespace

// Which way the three products of a row of Affine3f * Vector3f are summed.  0 (what the oracle and the device use):
// t + (c0 + (c1 + c2)), Eigen's unrolled redux.  1: t + ((c0 + c1) + c2).  2: ((t + c0) + c1) + c2.
// 1 and 2 exist ONLY to measure how much rests on that unpinned assumption (tests/test_association_exposure.py,
// tests/golden/association_exposure.json); nothing else selects them, and the product has no such switch.
static int g_association = 0;
void orc_set_association(int mode) { g_association = (mode == 1 || mode == 2) ? mode : 0; }

double orc_carve(orc_grid* g, const vcy_view* view, const float* sdf) {
  const vcy_update_option& opt = g->opt;
  const double t0 = now_ms();
  // voxel_carver.cc:436 -- max over the WHOLE buffer, not just the ROI
  const size_t npx = (size_t)view->width * view->height;
  const float max_sdf = *std::max_element(sdf, sdf + npx);
  const float* M = view->w2c;  // row-major 3x4, already cast to float (:438)
  const int nx = g->n[0], ny = g->n[1], nz = g->n[2];
  const int32_t* roi_min = view->roi_min;
  const int32_t* roi_max = view->roi_max;
  const int assoc = g_association;
#pragma omp parallel for schedule(dynamic, 1)
  for (int z = 0; z < nz; z++) {
    for (int y = 0; y < ny; y++) {
      for (int x = 0; x < nx; x++) {
        Voxel* voxel = g->get_ptr(x, y, z);
        if (voxel->outside || voxel->update_num > opt.voxel_max_update_num) continue;  // :447-450

        // :453  Eigen Affine3f * Vector3f.  [Eigen-internal order, UNPINNED]:
        // res = translation; res += linear * p, each row a 3-term redux
        // c0 + (c1 + c2) (Eigen redux_novec_unroller splits 3 = 1 + 2).
        const float* p = voxel->pos;
        float pc[3];
        for (int i = 0; i < 3; ++i) {
          float c0 = M[4 * i + 0] * p[0];
          float c1 = M[4 * i + 1] * p[1];
          float c2 = M[4 * i + 2] * p[2];
          if (assoc == 0) pc[i] = M[4 * i + 3] + (c0 + (c1 + c2));
          else if (assoc == 1) pc[i] = M[4 * i + 3] + ((c0 + c1) + c2);   // exposure measurement only, see orc_set_association
          else pc[i] = ((M[4 * i + 3] + c0) + c1) + c2;
        }
        if (pc[2] < 0) continue;  // :456

        float u, v;
        if (view->is_ortho) {  // camera.cc:201-205
          u = pc[0];
          v = pc[1];
        } else {  // camera.cc:131-137
          u = view->fx / pc[2] * pc[0] + view->cx;
          v = view->fy / pc[2] * pc[1] + view->cy;
        }
        float dist = kInvalidSdf;  // :462
        // :464-465.  The reference tests `u < roi_min.x || v < roi_min.y || roi_max.x < u ||
        // roi_max.y < v`; this is its complement, which differs only for NaN coordinates
        // (pc.z == 0 with pc.x or pc.y == 0): there the reference feeds NaN to floor()/int
        // conversion (undefined behaviour); oracle and device both treat them as outside.
        const bool inside = u >= roi_min[0] && v >= roi_min[1] && u <= roi_max[0] && v <= roi_max[1];
        if (!inside) {
          if (opt.update_outside == VCY_OUTSIDE_NONE) {
            continue;
          } else if (opt.update_outside == VCY_OUTSIDE_MAX) {
            dist = max_sdf;
          }
        } else {
          dist = (opt.sdf_interp == VCY_INTERP_NN)
                     ? interp_nn(u, v, sdf, view->width, roi_min, roi_max)
                     : interp_bilinear(u, v, sdf, view->width, roi_min, roi_max);
        }
        if (opt.use_truncation && dist < -1.0f) continue;  // :478

        if (voxel->update_num < 1) {  // :482-486
          voxel->sdf = dist;
          voxel->update_num++;
          continue;
        }
        if (opt.voxel_update == VCY_UPDATE_MAX) {  // UpdateVoxelMax :78-86
          if (dist > voxel->sdf) {
            voxel->sdf = dist;
            voxel->update_num++;
          }
        } else {  // UpdateVoxelWeightedAverage :88-95
          const float w = opt.voxel_update_weight;
          const float inv_denom = 1.0f / (w * (voxel->update_num + 1));
          voxel->sdf = (w * voxel->update_num * voxel->sdf + w * dist) * inv_denom;
          voxel->update_num++;
        }
      }
    }
  }
  return now_ms() - t0;
}

/* ------------------------------------------------------------- 2-D SDF -- */

void orc_distance_transform_l1(const uint8_t* mask, int width, int height,
                               const int32_t roi_min[2], const int32_t roi_max[2], float* dist) {
  const float kMax = std::numeric_limits<float>::max();
  auto D = [&](int x, int y) -> float& { return dist[(size_t)width * y + x]; };
  auto Mk = [&](int x, int y) -> uint8_t { return mask[(size_t)width * y + x]; };
  std::fill(dist, dist + (size_t)width * height, 0.0f);  // :104
  for (int y = roi_min[1]; y <= roi_max[1]; y++)          // :107-114
    for (int x = roi_min[0]; x <= roi_max[0]; x++)
      if (Mk(x, y) == 255) D(x, y) = kMax;
  // forward pass :117-141
  for (int y = roi_min[1] + 1; y <= roi_max[1]; y++) {
    float up = D(roi_min[0], y - 1);
    if (up < kMax) D(roi_min[0], y) = std::min(up + 1.0f, D(roi_min[0], y));
  }
  for (int x = roi_min[0] + 1; x <= roi_max[0]; x++) {
    float left = D(x - 1, roi_min[1]);
    if (left < kMax) D(x, roi_min[1]) = std::min(left + 1.0f, D(x, roi_min[1]));
  }
  for (int y = roi_min[1] + 1; y <= roi_max[1]; y++)
    for (int x = roi_min[0] + 1; x <= roi_max[0]; x++) {
      float m = std::min(D(x, y - 1), D(x - 1, y));
      if (m < kMax) D(x, y) = std::min(m + 1.0f, D(x, y));
    }
  // backward pass :144-166
  for (int y = roi_max[1] - 1; roi_min[1] <= y; y--) {
    float down = D(roi_max[0], y + 1);
    if (down < kMax) D(roi_max[0], y) = std::min(down + 1.0f, D(roi_max[0], y));
  }
  for (int x = roi_max[0] - 1; roi_min[0] <= x; x--) {
    float right = D(x + 1, roi_max[1]);
    if (right < kMax) D(x, roi_max[1]) = std::min(right + 1.0f, D(x, roi_max[1]));
  }
  for (int y = roi_max[1] - 1; roi_min[1] <= y; y--)
    for (int x = roi_max[0] - 1; roi_min[0] <= x; x--) {
      float m = std::min(D(x, y + 1), D(x + 1, y));
      if (m < kMax) D(x, y) = std::min(m + 1.0f, D(x, y));
    }
}

void orc_make_sdf(const uint8_t* mask, int width, int height, const int32_t roi_min[2],
                  const int32_t roi_max[2], int minmax_normalize, int use_truncation,
                  float truncation_band, float* sdf) {
  const size_t npx = (size_t)width * height;
  auto S = [&](int x, int y) -> float& { return sdf[(size_t)width * y + x]; };
  orc_distance_transform_l1(mask, width, height, roi_min, roi_max, sdf);  // :175
  for (int y = roi_min[1]; y <= roi_max[1]; y++)                           // :176-182
    for (int x = roi_min[0]; x <= roi_max[0]; x++)
      if (S(x, y) > 0) S(x, y) *= -1;

  std::vector<uint8_t> inv(mask, mask + npx);                              // :184-193
  for (int y = roi_min[1]; y <= roi_max[1]; y++)
    for (int x = roi_min[0]; x <= roi_max[0]; x++) {
      uint8_t& m = inv[(size_t)width * y + x];
      m = (m == 255) ? 0 : 255;
    }
  std::vector<float> pos(npx);
  orc_distance_transform_l1(inv.data(), width, height, roi_min, roi_max, pos.data());  // :196
  for (int y = roi_min[1]; y <= roi_max[1]; y++)                                        // :197-203
    for (int x = roi_min[0]; x <= roi_max[0]; x++)
      if (inv[(size_t)width * y + x] == 255) S(x, y) = pos[(size_t)width * y + x];

  if (minmax_normalize) {  // :205-222, min/max over the whole buffer
    float max_d = *std::max_element(sdf, sdf + npx);
    float min_d = *std::min_element(sdf, sdf + npx);
    float abs_max = std::max(std::abs(max_d), std::abs(min_d));
    if (abs_max > std::numeric_limits<float>::min()) {
      float norm = 1.0f / abs_max;
      for (int y = roi_min[1]; y <= roi_max[1]; y++)
        for (int x = roi_min[0]; x <= roi_max[0]; x++) S(x, y) *= norm;
    }
  }
  if (use_truncation) {  // :225-236
    for (int y = roi_min[1]; y <= roi_max[1]; y++)
      for (int x = roi_min[0]; x <= roi_max[0]; x++) {
        float& d = S(x, y);
        if (-truncation_band >= d)
          d = kInvalidSdf;
        else
          d = std::min(1.0f, d / truncation_band);
      }
  }
}

/* ------------------------------------------------------ marching cubes -- */

namespace {

// VertexInterp, marching_cubes.cc:25-57
void vertex_interp(double iso, const Voxel& a, const Voxel& b, float* p, bool linear) {
  if (!linear) {  // :54-56
    p[0] = a.pos[0]; p[1] = a.pos[1]; p[2] = a.pos[2];
    return;
  }
  double v1 = a.sdf, v2 = b.sdf;
  if (std::abs(iso - v1) < 0.00001) { p[0] = a.pos[0]; p[1] = a.pos[1]; p[2] = a.pos[2]; return; }
  if (std::abs(iso - v2) < 0.00001) { p[0] = b.pos[0]; p[1] = b.pos[1]; p[2] = b.pos[2]; return; }
  if (std::abs(v1 - v2) < 0.00001) { p[0] = a.pos[0]; p[1] = a.pos[1]; p[2] = a.pos[2]; return; }
  double mu = (iso - v1) / (v2 - v1);
  for (int k = 0; k < 3; ++k)
    p[k] = static_cast<float>(a.pos[k] +
                              mu * (static_cast<double>(b.pos[k]) - static_cast<double>(a.pos[k])));
}

}  // namespace

// z_begin/z_end select the cell layers [max(z_begin,1), z_end) (a cell is named by its max
// corner).  The whole grid is (0, nz): exactly the reference loop.  For a slab with z_begin > 0
// the layer below it (z_begin-1) is walked first without emitting faces, which is what a rank
// of the multi-GPU path knows from its two halo slices: the vertices it creates on the shared
// plane (both key voxels in slice z_begin-1) are reported first as n_foreign_vertices.
static double marching_cubes_range(const orc_grid* g, double iso, int linear_interp, int z_begin,
                                   int z_end, vcy_mesh* out) {
  const double t0 = now_ms();
  const McTables& T = tables();
  std::vector<std::array<float, 3>> vertices;
  std::vector<std::array<int, 3>> faces;
  std::vector<std::pair<int, int>> keys;
  std::map<std::pair<int, int>, int> ids2vertex;  // :78
  const int nx = g->n[0], ny = g->n[1];
  // corner pairs (interp order) and key order per edge, marching_cubes.cc:138-197
  static const int ea[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3};
  static const int eb[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
  static const int ka[12] = {0, 1, 3, 0, 4, 5, 7, 4, 0, 1, 2, 3};
  static const int kb[12] = {1, 2, 2, 3, 5, 6, 6, 7, 4, 5, 6, 7};
  const int z_own = std::max(z_begin, 1);
  const int z_first = (z_begin > 1) ? z_begin - 1 : z_own;  // ghost layer
  for (int z = z_first; z < z_end; z++) {
    const bool ghost = z < z_own;
    for (int y = 1; y < ny; y++) {
      for (int x = 1; x < nx; x++) {
        if (g->get(x, y, z).update_num < 1) continue;  // :88-90
        const Voxel* v[8];
        v[0] = &g->get(x - 1, y - 1, z - 1);            // :93-101
        v[1] = &g->get(x, y - 1, z - 1);
        v[2] = &g->get(x, y, z - 1);
        v[3] = &g->get(x - 1, y, z - 1);
        v[4] = &g->get(x - 1, y - 1, z);
        v[5] = &g->get(x, y - 1, z);
        v[6] = &g->get(x, y, z);
        v[7] = &g->get(x - 1, y, z);
        bool invalid = false;                            // :103-112
        for (int i = 0; i < 8; ++i) invalid |= (v[i]->sdf == kInvalidSdf);
        if (invalid) continue;
        int cube = 0;                                    // :121-128 (float promoted to double)
        for (int i = 0; i < 8; ++i)
          if (v[i]->sdf < iso) cube |= 1 << i;
        const int em = T.edge[cube];
        if (em == 0) continue;                           // :131-133
        float vert[12][3];
        std::pair<int, int> key[12];
        for (int e = 0; e < 12; ++e) {
          if (!(em & (1 << e))) continue;
          vertex_interp(iso, *v[ea[e]], *v[eb[e]], vert[e], linear_interp != 0);
          key[e] = std::make_pair(v[ka[e]]->id, v[kb[e]]->id);
        }
        for (int i = 0; T.tri[cube][i] != -1; i += 3) {  // :199-218
          std::array<int, 3> face;
          for (int j = 0; j < 3; j++) {
            const int e = T.tri[cube][i + (2 - j)];
            auto it = ids2vertex.find(key[e]);
            if (it == ids2vertex.end()) {
              face[j] = static_cast<int>(vertices.size());
              vertices.push_back({vert[e][0], vert[e][1], vert[e][2]});
              keys.push_back(key[e]);
              ids2vertex.insert(std::make_pair(key[e], face[j]));
            } else {
              face[j] = it->second;
            }
          }
          if (!ghost) faces.push_back(face);
        }
      }
    }
  }
  // drop the ghost-layer vertices that are not on the shared plane (they touch slice
  // z_begin-2 and no cell of the slab refers to them) and renumber
  std::vector<int> remap(vertices.size(), -1);
  std::vector<size_t> keep;
  const long long lo = (z_begin > 1) ? (long long)(z_begin - 1) * g->xy : 0;
  for (size_t i = 0; i < vertices.size(); ++i) {
    if (keys[i].first < lo) continue;
    remap[i] = (int)keep.size();
    keep.push_back(i);
  }
  out->n_vertices = (int64_t)keep.size();
  out->n_faces = (int64_t)faces.size();
  out->vertices = (float*)std::malloc(sizeof(float) * 3 * std::max<size_t>(1, keep.size()));
  out->faces = (int32_t*)std::malloc(sizeof(int32_t) * 3 * std::max<size_t>(1, faces.size()));
  out->edge_keys = (int64_t*)std::malloc(sizeof(int64_t) * 2 * std::max<size_t>(1, keep.size()));
  for (size_t k = 0; k < keep.size(); ++k) {
    const size_t i = keep[k];
    for (int c = 0; c < 3; ++c) out->vertices[3 * k + c] = vertices[i][c];
    out->edge_keys[2 * k] = keys[i].first;
    out->edge_keys[2 * k + 1] = keys[i].second;
  }
  for (size_t i = 0; i < faces.size(); ++i)
    for (int c = 0; c < 3; ++c) out->faces[3 * i + c] = remap[faces[i][c]];
  out->n_foreign_vertices = 0;
  return now_ms() - t0;
}

double orc_marching_cubes(const orc_grid* g, double iso, int linear_interp, vcy_mesh* out) {
  return marching_cubes_range(g, iso, linear_interp, 0, g->n[2], out);
}

double orc_marching_cubes_slab(const orc_grid* g, double iso, int linear_interp, int z_begin, int z_end,
                               vcy_mesh* out) {
  const double ms = marching_cubes_range(g, iso, linear_interp, z_begin, z_end, out);
  // Foreign vertices = the plane vertices created while the ghost layer was walked; they are a
  // prefix of the kept vertices.  Count them by walking the ghost layer alone.
  if (z_begin > 1) {
    vcy_mesh ghost_only;
    marching_cubes_range(g, iso, linear_interp, z_begin, z_begin, &ghost_only);
    out->n_foreign_vertices = ghost_only.n_vertices;
    orc_mesh_free(&ghost_only);
  }
  return ms;
}

/* ------------------------------------------------------------ ExtractVoxel -- */
// src/vacancy/extract_voxel.cc:15-79 (UpdateOnSurface), :258-317 (ExtractVoxel) and
// src/vacancy/mesh.cc:728-798 (MakeCube; R = I, t = 0 leave the corners at +-h exactly).
double orc_extract_voxel(orc_grid* g, int inside_empty, vcy_mesh* out) {
  const double t0 = now_ms();
  const int nx = g->n[0], ny = g->n[1], nz = g->n[2];
  const float h = g->resolution / 2;
  float cube[24][3];
  auto set = [&](int i, float x, float y, float z) { cube[i][0] = x; cube[i][1] = y; cube[i][2] = z; };
  auto cpy = [&](int i, int j) { for (int k = 0; k < 3; ++k) cube[i][k] = cube[j][k]; };
  set(0, -h, h, -h); set(1, h, h, -h); set(2, h, h, h); set(3, -h, h, h);
  set(4, -h, -h, -h); set(5, h, -h, -h); set(6, h, -h, h); set(7, -h, -h, h);
  cpy(8, 1); cpy(9, 2); cpy(10, 6); cpy(11, 5);
  cpy(12, 0); cpy(13, 3); cpy(14, 7); cpy(15, 4);
  cpy(16, 0); cpy(17, 1); cpy(18, 5); cpy(19, 4);
  cpy(20, 3); cpy(21, 2); cpy(22, 6); cpy(23, 7);
  static const int kFaces[12][3] = {{0, 2, 1}, {0, 3, 2}, {4, 5, 6}, {4, 6, 7}, {8, 9, 10}, {8, 10, 11},
                                    {12, 14, 13}, {12, 15, 14}, {16, 17, 18}, {16, 18, 19}, {20, 22, 21}, {20, 23, 22}};
  if (inside_empty) {
    for (Voxel& v : g->voxels) v.on_surface = false;  // ResetOnSurface
    constexpr float e = std::numeric_limits<float>::min();
    auto visit = [&](const Voxel& prev, Voxel* v) {
      if (v->update_num < 1 || prev.update_num < 1) return;
      if (v->sdf * prev.sdf < 0) v->on_surface = true;
      if (std::abs(v->sdf) < e) v->on_surface = true;
    };
    for (int z = 0; z < nz; z++) for (int y = 0; y < ny; y++) for (int x = 1; x < nx; x++)
      visit(g->get(x - 1, y, z), g->get_ptr(x, y, z));
    for (int z = 0; z < nz; z++) for (int x = 0; x < nx; x++) for (int y = 1; y < ny; y++)
      visit(g->get(x, y - 1, z), g->get_ptr(x, y, z));
    for (int y = 0; y < ny; y++) for (int x = 0; x < nx; x++) for (int z = 1; z < nz; z++)
      visit(g->get(x, y, z - 1), g->get_ptr(x, y, z));
  }
  std::vector<float> verts;
  std::vector<int> faces;
  for (int z = 0; z < nz; z++)
    for (int y = 0; y < ny; y++)
      for (int x = 0; x < nx; x++) {
        const Voxel& v = g->get(x, y, z);
        if (inside_empty) {
          if (!v.on_surface) continue;
        } else if (v.sdf > 0 || v.update_num < 1) {
          continue;
        }
        for (int i = 0; i < 24; ++i) for (int k = 0; k < 3; ++k) cube[i][k] += v.pos[k];     // Translate(pos)
        const int off = (int)(verts.size() / 3);
        for (int i = 0; i < 24; ++i) for (int k = 0; k < 3; ++k) verts.push_back(cube[i][k]);
        for (int i = 0; i < 12; ++i) for (int k = 0; k < 3; ++k) faces.push_back(kFaces[i][k] + off);
        for (int i = 0; i < 24; ++i) for (int k = 0; k < 3; ++k) cube[i][k] += -v.pos[k];    // Translate(-pos)
      }
  out->n_vertices = (int64_t)(verts.size() / 3);
  out->n_faces = (int64_t)(faces.size() / 3);
  out->n_foreign_vertices = 0;
  out->vertices = (float*)std::malloc(sizeof(float) * std::max<size_t>(3, verts.size()));
  out->faces = (int32_t*)std::malloc(sizeof(int32_t) * std::max<size_t>(3, faces.size()));
  out->edge_keys = (int64_t*)std::malloc(sizeof(int64_t) * 2);
  std::memcpy(out->vertices, verts.data(), sizeof(float) * verts.size());
  std::memcpy(out->faces, faces.data(), sizeof(int32_t) * faces.size());
  return now_ms() - t0;
}

void orc_mesh_free(vcy_mesh* m) {
  std::free(m->vertices);
  std::free(m->faces);
  std::free(m->edge_keys);
  std::memset(m, 0, sizeof(*m));
}

/* ------------------------------------------------------ pose arithmetic -- */

// examples.cc:36-50: pose = Translation3d(t) * Quaterniond(q).
// [Eigen-internal, UNPINNED] QuaternionBase::toRotationMatrix, no normalisation:
void orc_pose_from_tum(const double t[3], const double q[4], double c2w[12]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  double R[9] = {1.0 - (tyy + tzz), txy - twz,         txz + twy,
                 txy + twz,         1.0 - (txx + tzz), tyz - twx,
                 txz - twy,         tyz + twx,         1.0 - (txx + tyy)};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) c2w[4 * i + j] = R[3 * i + j];
    c2w[4 * i + 3] = t[i];
  }
}

// camera.cc:25,41: w2c = c2w.inverse().
// [Eigen-internal, UNPINNED] Transform<double,3,Affine>::inverse(): cofactor inverse of
// the 3x3 linear part (compute_inverse_size3: det = redux of cofactors_col0 .* col0 with
// the 1+2 split), translation = -(inv * t) with the same 3-term redux per row.
void orc_affine_inverse(const double c2w[12], double w2c[12]) {
  auto m = [&](int i, int j) { return c2w[4 * i + j]; };
  auto cof = [&](int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m(i1, j1) * m(i2, j2) - m(i1, j2) * m(i2, j1);
  };
  const double c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
  const double det = c00 * m(0, 0) + (c10 * m(1, 0) + c20 * m(2, 0));
  const double invdet = 1.0 / det;
  double inv[3][3];
  inv[0][0] = c00 * invdet;
  inv[0][1] = c10 * invdet;
  inv[0][2] = c20 * invdet;
  inv[1][0] = cof(0, 1) * invdet;
  inv[1][1] = cof(1, 1) * invdet;
  inv[1][2] = cof(2, 1) * invdet;
  inv[2][0] = cof(0, 2) * invdet;
  inv[2][1] = cof(1, 2) * invdet;
  inv[2][2] = cof(2, 2) * invdet;
  const double t[3] = {c2w[3], c2w[7], c2w[11]};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) w2c[4 * i + j] = inv[i][j];
    w2c[4 * i + 3] = -(inv[i][0] * t[0] + (inv[i][1] * t[1] + inv[i][2] * t[2]));
  }
}

// include/vacancy/common.h:51-75 with T = double.
// [Eigen-internal, UNPINNED] normalized() = v / sqrt(squaredNorm), squaredNorm redux 1+2.
void orc_lookat_c2w(const double position[3], const double target[3], const double up[3],
                    double c2w[12]) {
  auto normalize = [](double v[3]) {
    double n2 = v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]);
    if (n2 > 0) {
      double n = std::sqrt(n2);
      v[0] /= n; v[1] /= n; v[2] /= n;
    }
  };
  auto cross = [](const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
  };
  double c2[3] = {target[0] - position[0], target[1] - position[1], target[2] - position[2]};
  normalize(c2);
  double c0[3];
  cross(c2, up, c0);
  normalize(c0);
  double c1[3];
  cross(c2, c0, c1);
  for (int i = 0; i < 3; ++i) {
    c2w[4 * i + 0] = c0[i];
    c2w[4 * i + 1] = c1[i];
    c2w[4 * i + 2] = c2[i];
    c2w[4 * i + 3] = position[i];
  }
}

void orc_affine_to_float(const double m[12], float out[12]) {
  for (int i = 0; i < 12; ++i) out[i] = static_cast<float>(m[i]);  // voxel_carver.cc:438 cast<float>()
}

// PinholeCamera::set_fov_y, camera.cc:114-120; radians<float>, common.h:32-38
float orc_focal_from_fov_y(int height, float fov_y_deg) {
  float rad = fov_y_deg * static_cast<float>(0.01745329251994329576923690768489);
  return height * 0.5f / static_cast<float>(std::tan(rad * 0.5));
}

}  // extern "C"
