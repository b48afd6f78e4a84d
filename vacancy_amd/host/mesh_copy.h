// Internal to libvacancy.so: the library's mesh arrays (page-locked, pooled) into the std::vectors the reference's
// Mesh API hands out (include/vacancy/mesh.h: vertices(), vertex_indices()).
#pragma once

#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>
#include <cstdint>
#if defined(__linux__)
#include <sys/mman.h>
#endif

#include "vacancy/common.h"

namespace vacancy {
namespace detail {

// Huge pages for a large fresh buffer where the kernel hands them out on request: 400 faults instead of 200 000.
inline void RequestHugePages(void* p, size_t bytes) {
#if defined(__linux__)
  if (bytes < ((size_t)8 << 20)) return;
  const uintptr_t a0 = (reinterpret_cast<uintptr_t>(p) + ((size_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1);
  const uintptr_t a1 = (reinterpret_cast<uintptr_t>(p) + bytes) & ~(((uintptr_t)2 << 20) - 1);
  if (a1 > a0) (void)madvise(reinterpret_cast<void*>(a0), a1 - a0, MADV_HUGEPAGE);
#else
  (void)p, (void)bytes;
#endif
}

// vcy_mesh_arrays_fn for a Mesh: the library's host threads write the voxel mesh straight into the Mesh's vectors
// (vcy_extract_voxel_into / vcy_voxel_cubes_into) -- resize() initialises nothing, see CopyTriples.
template <typename VV, typename VF>
struct MeshArrays {
  std::vector<VV>* vertices;
  std::vector<VF>* faces;
  static int Provide(void* user, int64_t n_vertices, int64_t n_faces, float** v, int32_t** f) {
    static_assert(sizeof(VV) == 3 * sizeof(float) && sizeof(VF) == 3 * sizeof(int32_t), "packed vector layout");
    MeshArrays* a = static_cast<MeshArrays*>(user);
    a->vertices->resize(static_cast<size_t>(n_vertices));
    a->faces->resize(static_cast<size_t>(n_faces));
    RequestHugePages(a->vertices->data(), sizeof(VV) * static_cast<size_t>(n_vertices));
    RequestHugePages(a->faces->data(), sizeof(VF) * static_cast<size_t>(n_faces));
    *v = reinterpret_cast<float*>(a->vertices->data());
    *f = reinterpret_cast<int32_t*>(a->faces->data());
    return 0;
  }
};

// dst[0, n) = the n packed triples at src.  `dst->resize(n)` initialises nothing (Eigen's fixed-size vectors have no
// default value, and neither has the stand-in of vacancy/linalg.h), so for a large mesh -- ExtractVoxel's is 24 vertices
// and 12 triangles per kept voxel, 0.8 - 2 GB for the bunny at resolution 2.5 -- the pages of the fresh vector are
// first touched by the copying threads, in parallel, instead of being zeroed by one thread and then copied over
// (class API, per call: 190 -> about 30 ms where the library itself takes 4.5).
template <typename V, typename S>
void CopyTriples(std::vector<V>* dst, const S* src, size_t n, size_t offset = 0) {
  static_assert(sizeof(V) == 3 * sizeof(S), "packed vector layout");
  dst->resize(offset + n);
  if (n == 0) return;
  char* d = reinterpret_cast<char*>(dst->data() + offset);
  const char* s = reinterpret_cast<const char*>(src);
  const size_t bytes = n * sizeof(V);
  constexpr size_t kPerThread = (size_t)16 << 20;  // a thread per 16 MiB, at most 16 (a container's quota, not the machine's cores)
  const size_t nthreads = std::min<size_t>(16, std::max<size_t>(1, std::min<size_t>(bytes / kPerThread, std::thread::hardware_concurrency())));
  if (nthreads <= 1) {
    std::memcpy(d, s, bytes);
    return;
  }
  RequestHugePages(d, bytes);
  const size_t piece = ((bytes + nthreads - 1) / nthreads + 4095) & ~(size_t)4095;
  std::vector<std::thread> pool;
  for (size_t t = 0; t < nthreads; ++t) {
    const size_t b0 = std::min(bytes, t * piece), b1 = std::min(bytes, b0 + piece);
    if (b1 > b0) pool.emplace_back([=]() { std::memcpy(d + b0, s + b0, b1 - b0); });
  }
  for (std::thread& th : pool) th.join();
}

}  // namespace detail
}  // namespace vacancy
