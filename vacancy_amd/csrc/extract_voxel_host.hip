// ExtractVoxel -- the cube-per-voxel visualisation mesh.
//
// Replaces ExtractVoxel() / UpdateOnSurface() (reference src/vacancy/extract_voxel.cc:258-317,
// :15-79) with MakeCube (src/vacancy/mesh.cc:728-798).  It stays on the HOST, on the downloaded
// voxel state, and that is not a shortcut: the reference moves ONE cube mesh to every kept voxel
// and back (`Translate(pos)` ... `Translate(-pos)`), so each emitted corner carries the float
// rounding of all earlier kept voxels -- a serial dependence through the whole scan that has to be
// replayed in order to match the reference bit for bit.  What is parallel (the keep / on-surface
// predicate) is a byte mask here.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "vcy_internal.h"

extern "C" int vcy_extract_voxel(vcy_ctx* c, int inside_empty, vcy_mesh* out) {
  using namespace vcy;
  if (!c || !out) {
    set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  std::memset(out, 0, sizeof(*out));
  if (c->z0 != 0 || c->z1 != c->nz) {
    set_error("vcy_extract_voxel needs the whole grid in one context");
    return VCY_ERR_UNSUPPORTED;
  }
  const int nx = c->nx, ny = c->ny, nz = c->nz;
  const size_t n = (size_t)nx * ny * nz;
  std::vector<float> sdf(n);
  std::vector<int32_t> cnt(n);
  int rc = vcy_download(c, sdf.data(), cnt.data());
  if (rc != VCY_OK) return rc;
  std::vector<float> px(nx), py(ny), pz(nz);
  VCY_HIP_CHECK(hipMemcpy(px.data(), c->d_px, sizeof(float) * nx, hipMemcpyDeviceToHost));
  VCY_HIP_CHECK(hipMemcpy(py.data(), c->d_py, sizeof(float) * ny, hipMemcpyDeviceToHost));
  VCY_HIP_CHECK(hipMemcpy(pz.data(), c->d_pz, sizeof(float) * nz, hipMemcpyDeviceToHost));

  // which voxels get a cube
  std::vector<uint8_t> keep(n, 0);
  if (inside_empty) {
    // sign change (or |sdf| < FLT_MIN) against the -x, -y or -z neighbour, both touched
    const float tiny = std::numeric_limits<float>::min();
    const size_t stride[3] = {1, (size_t)nx, (size_t)nx * ny};
    for (int z = 0; z < nz; ++z)
      for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) {
          const size_t i = ((size_t)z * ny + y) * nx + x;
          if (cnt[i] < 1) continue;
          const int coord[3] = {x, y, z};
          bool on = false;
          for (int a = 0; a < 3 && !on; ++a) {
            if (coord[a] == 0 || cnt[i - stride[a]] < 1) continue;
            on = (sdf[i] * sdf[i - stride[a]] < 0) || (std::fabs(sdf[i]) < tiny);
          }
          keep[i] = on;
        }
  } else {
    for (size_t i = 0; i < n; ++i) keep[i] = !(sdf[i] > 0 || cnt[i] < 1);
  }
  size_t kept = 0;
  for (size_t i = 0; i < n; ++i) kept += keep[i];
  if (kept * 24 > (size_t)std::numeric_limits<int32_t>::max()) {
    set_error("voxel mesh too large for 32-bit indices");
    return VCY_ERR_TOO_MANY_VOXELS;
  }

  // unit cube of MakeCube(resolution): 6 quads x 4 corners, 12 triangles
  const float h = c->opt.resolution / 2;
  static const int8_t sgn[24][3] = {
      {-1, 1, -1}, {1, 1, -1},  {1, 1, 1},   {-1, 1, 1},  {-1, -1, -1}, {1, -1, -1},  {1, -1, 1},  {-1, -1, 1},
      {1, 1, -1},  {1, 1, 1},   {1, -1, 1},  {1, -1, -1}, {-1, 1, -1},  {-1, 1, 1},   {-1, -1, 1}, {-1, -1, -1},
      {-1, 1, -1}, {1, 1, -1},  {1, -1, -1}, {-1, -1, -1}, {-1, 1, 1},  {1, 1, 1},    {1, -1, 1},  {-1, -1, 1}};
  static const int8_t tri[12][3] = {{0, 2, 1},    {0, 3, 2},    {4, 5, 6},    {4, 6, 7},    {8, 9, 10},   {8, 10, 11},
                                    {12, 14, 13}, {12, 15, 14}, {16, 17, 18}, {16, 18, 19}, {20, 22, 21}, {20, 23, 22}};
  float cube[24][3];
  for (int i = 0; i < 24; ++i)
    for (int k = 0; k < 3; ++k) cube[i][k] = sgn[i][k] < 0 ? -h : h;

  out->n_vertices = (int64_t)kept * 24;
  out->n_faces = (int64_t)kept * 12;
  out->vertices = (float*)std::malloc(sizeof(float) * 3 * std::max<size_t>(1, kept * 24));
  out->faces = (int32_t*)std::malloc(sizeof(int32_t) * 3 * std::max<size_t>(1, kept * 12));
  out->edge_keys = (int64_t*)std::malloc(sizeof(int64_t) * 2);
  float* v = out->vertices;
  int32_t* f = out->faces;
  int32_t base = 0;
  size_t i = 0;
  for (int z = 0; z < nz; ++z)
    for (int y = 0; y < ny; ++y)
      for (int x = 0; x < nx; ++x, ++i) {
        if (!keep[i]) continue;
        const float p[3] = {px[x], py[y], pz[z]};
        // the drifting cube: move to the voxel, emit, move back (extract_voxel.cc:292-310)
        for (int q = 0; q < 24; ++q)
          for (int k = 0; k < 3; ++k) {
            cube[q][k] += p[k];
            *v++ = cube[q][k];
          }
        for (int t = 0; t < 12; ++t)
          for (int k = 0; k < 3; ++k) *f++ = tri[t][k] + base;
        for (int q = 0; q < 24; ++q)
          for (int k = 0; k < 3; ++k) cube[q][k] += -p[k];
        base += 24;
      }
  return VCY_OK;
}
