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
#if defined(__linux__)
  {  // huge pages for the fresh buffer where the kernel hands them out on request: 400 faults instead of 200 000
    const uintptr_t a0 = (reinterpret_cast<uintptr_t>(d) + ((size_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1);
    const uintptr_t a1 = (reinterpret_cast<uintptr_t>(d) + bytes) & ~(((uintptr_t)2 << 20) - 1);
    if (a1 > a0) (void)madvise(reinterpret_cast<void*>(a0), a1 - a0, MADV_HUGEPAGE);
  }
#endif
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
