// ExtractVoxel -- the cube-per-voxel visualisation mesh.
//
// Replaces ExtractVoxel() / UpdateOnSurface() (reference src/vacancy/extract_voxel.cc:258-317,
// :15-79) with MakeCube (src/vacancy/mesh.cc:728-798).
//
// What is parallel runs on the device, on the resident state: the keep predicate of every voxel
// (`sdf <= 0 && update_num >= 1`, or the on-surface sign tests against the -x / -y / -z neighbours) is
// balloted into a bit plane, counted per block, scanned, and the kept voxel ids are compacted in scan
// order.  Only that list crosses PCIe (8 B per kept voxel instead of 5-8 B per voxel of the grid).
//
// What is serial stays on the host, and that is not a shortcut: the reference moves ONE cube mesh to
// every kept voxel and back (`Translate(pos)` ... `Translate(-pos)`, extract_voxel.cc:292-310), so each
// emitted corner carries the float rounding of all earlier kept voxels -- a dependence through the whole
// scan that has to be replayed in order to match bit for bit.  The 24 corners of the cube only hold two
// values per axis (-h and +h), so the replay is six scalar chains, not seventy-two.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include <emmintrin.h>  // the host half of ExtractVoxel fills its 800 MB mesh with streaming stores (SSE2: any x86-64)

#include "vcy_internal.h"

namespace vcy {
namespace {

typedef unsigned long long u64;

// fl(a * b) < 0 for floats, whatever the denormal mode of the multiplier: the exact product (it fits a
// double) is negative and does not round to -0, i.e. is beyond half of the smallest denormal 2^-149.
__device__ __forceinline__ bool product_negative(float a, float b) {
  const double p = (double)a * (double)b;
  return p < 0.0 && fabs(p) > 0x1p-150;
}

// Keep bit of voxel i = blockIdx.x * 256 + threadIdx.x, one 64-bit word per wave, kept voxels per block.
//   SURFACE false: !(sdf > 0 || update_num < 1)                         (extract_voxel.cc:283-286)
//   SURFACE true : UpdateOnSurface (:15-79): the voxel and its -x, -y or -z neighbour are both touched
//                  and (sdf * neighbour.sdf < 0 or |sdf| < FLT_MIN)
template <typename CountT, bool SURFACE>
__global__ __launch_bounds__(256) void xv_keep_kernel(const float* __restrict__ sdf, const CountT* __restrict__ cnt,
                                                      int nx, int ny, int64_t n, int has_below, u64* __restrict__ bits,
                                                      u64* __restrict__ block_counts) {
  __shared__ int sm[4];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool keep = false;
  if (i < n) {
    const float s = sdf[i];
    const bool touched = cnt[i] >= 1;
    if (!SURFACE) {
      keep = !(s > 0.0f || !touched);
    } else if (touched) {
      const int64_t row = i / nx;
      const int x = (int)(i - row * nx);
      const int y = (int)(row % ny);
      const int64_t slice = (int64_t)nx * ny;
      const bool tiny = (__float_as_uint(s) & 0x7fffffffu) < 0x00800000u;  // |sdf| < FLT_MIN
      if (x > 0 && cnt[i - 1] >= 1) keep = keep || tiny || product_negative(s, sdf[i - 1]);
      if (y > 0 && cnt[i - nx] >= 1) keep = keep || tiny || product_negative(s, sdf[i - nx]);
      // (a z-slab above another one: the slice below its first is the halo slice z0 - 1, stored right below the slab)
      if ((i >= slice || has_below) && cnt[i - slice] >= 1) keep = keep || tiny || product_negative(s, sdf[i - slice]);
    }
  }
  const u64 word = __ballot(keep);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    bits[i >> 6] = word;  // (i of lane 0 is a multiple of 64; words beyond n are whole zero words)
    sm[wave] = __popcll(word);
  }
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = (u64)(sm[0] + sm[1] + sm[2] + sm[3]);
}

// ids[rank] = i for every kept voxel, rank = its number in scan order
__global__ __launch_bounds__(256) void xv_compact_kernel(const u64* __restrict__ bits, const u64* __restrict__ block_offs,
                                                         int64_t* __restrict__ ids) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u64* w = bits + (int64_t)blockIdx.x * 4;
  u64 rank = block_offs[blockIdx.x];
  for (int k = 0; k < wave; ++k) rank += (u64)__popcll(w[k]);
  const u64 word = w[wave];
  if (!((word >> lane) & 1ull)) return;
  rank += (u64)__popcll(word & ((1ull << lane) - 1ull));
  ids[rank] = (int64_t)blockIdx.x * 256 + threadIdx.x;
}

}  // namespace

int device_exclusive_scan_u64(unsigned long long* d, int64_t n, unsigned long long* d_total,
                              unsigned long long* scratch, hipStream_t stream);  // mc_kernels.hip

}  // namespace vcy

// Kept voxel ids on the host: page-locked memory from the mesh pool (mesh_host_alloc) -- the copy from the device runs at
// the link's rate and the pages are there already (a fresh std::vector cost 11 ms for the 36 MB of the bunny's first view:
// zero-fill, page faults and a staged copy), and vcy_extract_voxel_ids hands the array to its caller as it is.
struct HostIds {
  int64_t* p = nullptr;
  size_t n = 0;
  HostIds() = default;
  HostIds(const HostIds&) = delete;
  HostIds& operator=(const HostIds&) = delete;
  ~HostIds() { vcy::mesh_host_free(p); }
  int64_t* release() {
    int64_t* q = p;
    p = nullptr, n = 0;
    return q;
  }
};

// (grow-only device scratch kept by the context: an extraction per view allocated and freed twice per call)
static int xv_reserve(void** buf, size_t* cap, size_t need, hipStream_t s) {
  using namespace vcy;
  if (*cap >= need) return VCY_OK;
  VCY_HIP_CHECK(hipStreamSynchronize(s));
  if (*buf) (void)hipFree(*buf);
  *buf = nullptr, *cap = 0;
  VCY_HIP_CHECK(hipMalloc(buf, need + need / 8));
  *cap = need + need / 8;
  return VCY_OK;
}

// Kept voxel ids of this context's slab, GLOBAL ids in scan order (device predicate + compaction).
static int kept_voxel_ids(vcy_ctx* c, int inside_empty, HostIds* out_ids) {
  using namespace vcy;
  VCY_HIP_CHECK(hipSetDevice(c->device));
  {
    const int rcf = flush_pending(c);  // queued views are part of the state
    if (rcf != VCY_OK) return rcf;
  }
  // the on-surface test looks at the voxel below: for a slab above another one that is the halo slice z0 - 1
  if (inside_empty && c->halo_lo > 0 && !c->halo_valid) {
    set_error("halo slices not installed (call vcy_halo_unpack / vcy_halo_allgather before extracting from a slab)");
    return VCY_ERR_NOT_INITIALIZED;
  }
  const int nx = c->nx, ny = c->ny, nz = c->nz_local();
  const int64_t n = (int64_t)nx * ny * nz;
  hipStream_t s = c->stream;
  HostIds& ids = *out_ids;
  vcy::mesh_host_free(ids.release());
  // ---- device: keep bits -> block counts -> scan -> kept voxel ids ---------------------------------
  if (!c->fresh) {  // a fresh grid is untouched everywhere: nothing is kept under either predicate
    const int64_t nblocks = (n + 255) / 256;
    auto align = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t sz_bits = align(sizeof(u64) * (size_t)nblocks * 4);
    const size_t sz_counts = align(sizeof(u64) * ((size_t)nblocks + 1));
    const size_t sz_scan = align(sizeof(u64) * ((size_t)nblocks / 1024 + 64) * 2);
    {
      const int rcs = xv_reserve(&c->d_xv_scratch, &c->xv_scratch_bytes, sz_bits + sz_counts + sz_scan + 256, s);
      if (rcs != VCY_OK) return rcs;
    }
    char* d_scratch = (char*)c->d_xv_scratch;
    u64* d_bits = (u64*)d_scratch;
    u64* d_counts = (u64*)(d_scratch + sz_bits);
    u64* d_scan = (u64*)(d_scratch + sz_bits + sz_counts);
    u64* d_total = (u64*)(d_scratch + sz_bits + sz_counts + sz_scan);
    int64_t* d_ids = nullptr;
    int rc = VCY_OK;
#define XV_TRY(expr)                                                                                  \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess && rc == VCY_OK) {                                                           \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);           \
      rc = VCY_ERR_HIP;                                                                               \
    }                                                                                                 \
  } while (0)
    const float* sdf = c->owned_slab_sdf();
    const void* cnt = c->owned_slab_cnt();
#define XV_KEEP(CT, SURF)                                                                                       \
  hipLaunchKernelGGL((xv_keep_kernel<CT, SURF>), dim3((unsigned)nblocks), dim3(256), 0, s, sdf, (const CT*)cnt, \
                     nx, ny, n, c->halo_lo > 0 ? 1 : 0, d_bits, d_counts)
#define XV_KEEP_S(CT)                                      \
  do {                                                     \
    if (inside_empty) XV_KEEP(CT, true); else XV_KEEP(CT, false); \
  } while (0)
    if (c->cnt_bytes == 1) XV_KEEP_S(uint8_t);
    else if (c->cnt_bytes == 2) XV_KEEP_S(uint16_t);
    else XV_KEEP_S(uint32_t);
#undef XV_KEEP_S
#undef XV_KEEP
    XV_TRY(hipGetLastError());
    if (rc == VCY_OK) rc = device_exclusive_scan_u64(d_counts, nblocks, d_total, d_scan, s);
    u64 kept = 0;
    if (rc == VCY_OK) XV_TRY(hipMemcpyAsync(&kept, d_total, sizeof(u64), hipMemcpyDeviceToHost, s));
    if (rc == VCY_OK) XV_TRY(hipStreamSynchronize(s));
    if (rc == VCY_OK && kept * 24 > (u64)std::numeric_limits<int32_t>::max()) {
      set_error("voxel mesh too large for 32-bit indices");
      rc = VCY_ERR_TOO_MANY_VOXELS;
    }
    if (rc == VCY_OK && kept > 0) {
      rc = xv_reserve(&c->d_xv_ids, &c->xv_ids_bytes, sizeof(int64_t) * (size_t)kept, s);
      d_ids = (int64_t*)c->d_xv_ids;
      if (rc == VCY_OK) {
        ids.p = (int64_t*)mesh_host_alloc(sizeof(int64_t) * (size_t)kept);
        if (!ids.p) {
          set_error("out of host memory for the kept voxel ids");
          rc = VCY_ERR_INTERNAL;
        }
      }
      if (rc == VCY_OK) {
        ids.n = (size_t)kept;
        hipLaunchKernelGGL(xv_compact_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, d_bits, d_counts, d_ids);
        XV_TRY(hipGetLastError());
        XV_TRY(hipMemcpyAsync(ids.p, d_ids, sizeof(int64_t) * (size_t)kept, hipMemcpyDeviceToHost, s));
        XV_TRY(hipStreamSynchronize(s));
      }
    }
#undef XV_TRY
    if (rc != VCY_OK) return rc;
  }
  const int64_t first = (int64_t)c->z0 * c->slice;
  if (first)
    for (size_t t = 0; t < ids.n; ++t) ids.p[t] += first;
  return VCY_OK;
}

// The serial half of ExtractVoxel (extract_voxel.cc:290-311): ONE cube mesh (MakeCube, mesh.cc:728-798) translated to
// every kept voxel and back, in scan order.  Host arithmetic only (this file is built with -ffp-contract=off).
// VCY_XV_TIMING=1 in the environment: the phases of an ExtractVoxel call on stderr (development aid)
static bool xv_timing() {
  static const bool on = std::getenv("VCY_XV_TIMING") != nullptr;
  return on;
}
static double xv_now() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// (`arrays` null: the mesh in library-owned arrays, *out; else into what the caller's callback returns, *out only counts)
static int cubes_from_ids(const float* px, const float* py, const float* pz, int nx, int ny, float resolution,
                          const int64_t* ids, size_t kept, vcy_mesh* out, vcy_mesh_arrays_fn arrays = nullptr,
                          void* arrays_user = nullptr) {
  using namespace vcy;
  // unit cube of MakeCube(resolution): 6 quads x 4 corners, 12 triangles
  const float h = resolution / 2;
  static const int8_t sgn[24][3] = {
      {-1, 1, -1}, {1, 1, -1},  {1, 1, 1},   {-1, 1, 1},  {-1, -1, -1}, {1, -1, -1},  {1, -1, 1},  {-1, -1, 1},
      {1, 1, -1},  {1, 1, 1},   {1, -1, 1},  {1, -1, -1}, {-1, 1, -1},  {-1, 1, 1},   {-1, -1, 1}, {-1, -1, -1},
      {-1, 1, -1}, {1, 1, -1},  {1, -1, -1}, {-1, -1, -1}, {-1, 1, 1},  {1, 1, 1},    {1, -1, 1},  {-1, -1, 1}};
  static const int8_t tri[12][3] = {{0, 2, 1},    {0, 3, 2},    {4, 5, 6},    {4, 6, 7},    {8, 9, 10},   {8, 10, 11},
                                    {12, 14, 13}, {12, 15, 14}, {16, 17, 18}, {16, 18, 19}, {20, 22, 21}, {20, 23, 22}};
  if ((unsigned long long)kept * 24 > (unsigned long long)std::numeric_limits<int32_t>::max()) {
    set_error("voxel mesh too large for 32-bit indices");
    return VCY_ERR_TOO_MANY_VOXELS;
  }
  out->n_vertices = (int64_t)kept * 24;
  out->n_faces = (int64_t)kept * 12;
  if (kept == 0) return VCY_OK;  // an empty mesh has no arrays
  const double t_a = xv_now();
  if (arrays) {
    float* v_dst = nullptr;
    int32_t* f_dst = nullptr;
    if (arrays(arrays_user, out->n_vertices, out->n_faces, &v_dst, &f_dst) != 0 || !v_dst || !f_dst) {
      std::memset(out, 0, sizeof(*out));
      set_error("the caller's callback gave no arrays for the voxel mesh");
      return VCY_ERR_INTERNAL;
    }
    out->vertices = v_dst, out->faces = f_dst;  // (the caller's: cleared again before returning)
  } else {
    out->vertices = (float*)mesh_host_alloc(sizeof(float) * 3 * kept * 24);
    out->faces = (int32_t*)mesh_host_alloc(sizeof(int32_t) * 3 * kept * 12);
    if (!out->vertices || !out->faces) {
      mesh_host_free(out->vertices);
      mesh_host_free(out->faces);
      std::memset(out, 0, sizeof(*out));
      set_error("out of host memory for the voxel mesh");
      return VCY_ERR_INTERNAL;
    }
  }
  const double t_b = xv_now();
  // corner value per axis and sign: every corner with the same (axis, sign) went through the same
  // additions, so the reference's 72 running coordinates are these six.
  // Pass 1, serial -- the chain itself: the six values of every kept voxel after Translate(pos) (24 bytes per voxel).
  // Pass 2, host threads -- what is independent once those are known: 24 vertices and 12 triangles per voxel, 432 bytes
  // (round 5: one loop did both at 98 ns per voxel, 182 ms per view of the bunny at resolution 2.5 -- 1.86 M kept
  // voxels, an 800 MB mesh).
  // The chain is serial by construction -- every voxel's corners carry the float rounding of all kept voxels before it --
  // but what it carries is six floats whose history can be guessed cheaply (see do_chunk).  Round 6 therefore runs the list
  // in chunks on host threads, SPECULATIVELY: a chunk takes a guessed incoming state, computes its corners and fills its
  // vertices and triangles; afterwards the chunks are checked in order -- the state a chunk assumed must equal, bit for
  // bit, the state its predecessor really left -- and a chunk whose assumption was wrong is done again from the true
  // state.  Exact whatever the data (the test suite's permuted id lists included); 21 -> 7 ms per call for the
  // bunny's 1.86 M kept voxels at resolution 2.5 (one loop doing everything: 100 ms in round 4).
  struct Axes {
    const float *px, *py, *pz;
    int64_t nx, slice;
  } ax{px, py, pz, nx, (int64_t)nx * ny};
  // the chain over ids[t0, t1) from `state` (lo[3], hi[3]); corners written when cr != nullptr
  auto run_chain = [&ax, ids](size_t t0, size_t t1, float state[6], float* cr) {
    int64_t z = 0, y = 0;  // row of the previous voxel: the ids of a scan are ascending, so a division is rarely needed
    float lo[3] = {state[0], state[1], state[2]}, hi[3] = {state[3], state[4], state[5]};
    for (size_t t = t0; t < t1; ++t) {
      const int64_t i = ids[t];
      if (i < z * ax.slice || i >= (z + 1) * ax.slice) z = (i >= (z + 1) * ax.slice && i < (z + 2) * ax.slice) ? z + 1 : i / ax.slice;
      const int64_t r = i - z * ax.slice;
      if (r < y * ax.nx || r >= (y + 1) * ax.nx) y = (r >= (y + 1) * ax.nx && r < (y + 2) * ax.nx) ? y + 1 : r / ax.nx;
      const int64_t x = r - y * ax.nx;
      const float p[3] = {ax.px[x], ax.py[y], ax.pz[z]};
      for (int k = 0; k < 3; ++k) {
        const float clo = lo[k] + p[k], chi = hi[k] + p[k];  // Translate(pos)   (built with -ffp-contract=off)
        if (cr) cr[k] = clo, cr[3 + k] = chi;
        lo[k] = clo + -p[k];                                  // Translate(-pos)
        hi[k] = chi + -p[k];
      }
      if (cr) cr += 6;
    }
    for (int k = 0; k < 3; ++k) state[k] = lo[k], state[3 + k] = hi[k];
  };
  // A voxel's 72 vertex floats are 18 vectors of four: vector j starts at component j mod 3, so it is one of three
  // rotations of (x, y, z, x) of the low corner blended with the same rotation of the high corner by a constant mask;
  // its 36 indices are 9 constant vectors plus 24 t.  Written with STREAMING stores: the arrays are written once, front
  // to back, and never read here -- ordinary stores made every line a read (for ownership) and a write-back, 1.6 GB of
  // traffic for an 800 MB mesh, and 108 scalar stores per voxel (profiles/r06/extract_voxel_phases.txt).
  // (16-byte alignment: both arrays start on a page or malloc boundary and a voxel is 288 / 144 bytes.)
  struct FillTables {
    __m128 sel_hi[18];
    __m128i tri4[9];
  };
  static const FillTables tab = [] {
    FillTables t;
    for (int j = 0; j < 18; ++j) {
      alignas(16) uint32_t m[4];
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * j + e;
        m[e] = sgn[i / 3][i % 3] < 0 ? 0u : 0xffffffffu;
      }
      t.sel_hi[j] = _mm_castsi128_ps(_mm_load_si128((const __m128i*)m));
    }
    for (int j = 0; j < 9; ++j) {
      alignas(16) int32_t q[4];
      for (int e = 0; e < 4; ++e) q[e] = tri[(4 * j + e) / 3][(4 * j + e) % 3];
      t.tri4[j] = _mm_load_si128((const __m128i*)q);
    }
    return t;
  }();
  // (VCY_XV_SCALAR_FILL in the environment: the scalar loop, for comparisons)
  static const bool scalar_fill = std::getenv("VCY_XV_SCALAR_FILL") != nullptr;
  const bool aligned16 = (((uintptr_t)out->vertices | (uintptr_t)out->faces) & 15) == 0 && !scalar_fill;
  auto fill = [&](size_t t0, size_t t1, const float* cr) {  // (cr: readable up to cr[6 (t1 - t0)], one float past the end)
    float* v = out->vertices + 72 * t0;
    int32_t* f = out->faces + 36 * t0;
    if (!aligned16) {  // (never with the library's own arrays; kept for a caller's)
      for (size_t t = t0; t < t1; ++t, cr += 6) {
        for (int q = 0; q < 24; ++q)
          for (int k = 0; k < 3; ++k) *v++ = cr[(sgn[q][k] < 0 ? 0 : 3) + k];
        const int32_t base = (int32_t)(24 * t);
        for (int q = 0; q < 12; ++q)
          for (int k = 0; k < 3; ++k) *f++ = tri[q][k] + base;
      }
      return;
    }
    for (size_t t = t0; t < t1; ++t, cr += 6, v += 72, f += 36) {
      const __m128 lo = _mm_loadu_ps(cr), hi = _mm_loadu_ps(cr + 3);  // (x, y, z, -)
      const __m128 L[3] = {_mm_shuffle_ps(lo, lo, _MM_SHUFFLE(0, 2, 1, 0)), _mm_shuffle_ps(lo, lo, _MM_SHUFFLE(1, 0, 2, 1)),
                           _mm_shuffle_ps(lo, lo, _MM_SHUFFLE(2, 1, 0, 2))};
      const __m128 H[3] = {_mm_shuffle_ps(hi, hi, _MM_SHUFFLE(0, 2, 1, 0)), _mm_shuffle_ps(hi, hi, _MM_SHUFFLE(1, 0, 2, 1)),
                           _mm_shuffle_ps(hi, hi, _MM_SHUFFLE(2, 1, 0, 2))};
      for (int j = 0; j < 18; ++j) {
        const __m128 m = tab.sel_hi[j];
        _mm_stream_ps(v + 4 * j, _mm_or_ps(_mm_and_ps(m, H[j % 3]), _mm_andnot_ps(m, L[j % 3])));
      }
      const __m128i base = _mm_set1_epi32((int32_t)(24 * t));
      for (int j = 0; j < 9; ++j) _mm_stream_si128((__m128i*)(f + 4 * j), _mm_add_epi32(tab.tri4[j], base));
    }
    _mm_sfence();  // (the stores above are weakly ordered: visible before this thread is joined)
  };
  // (at most 16 threads: hardware_concurrency() counts the cores of the machine, not what a container's quota allows)
  const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  const size_t nthreads = kept < 65536 ? 1 : (size_t)hw;
  constexpr size_t kWarm = 1024, kSample = 8192, kPiece = 4096;  // corners of kPiece voxels at a time stay in the thread's cache
  const size_t nchunks = nthreads;
  const size_t per = (kept + nchunks - 1) / nchunks;
  struct Chunk {
    size_t t0, t1;
    float assumed[6], left[6];
  };
  std::vector<Chunk> chunks(nchunks);
  const float start[6] = {-h, -h, -h, h, h, h};
  auto do_chunk = [&](Chunk& c, const float* in_state) {  // in_state: the true state, or null = warm up to a guess
    float st[6];
    if (in_state) {
      std::memcpy(st, in_state, sizeof(st));
    } else {
      std::memcpy(st, start, sizeof(st));
      // What the state remembers of its history is how coarse the grids were it has been rounded to: lo' = fl(fl(lo + p)
      // - p) rounds lo to a multiple of ulp(lo + p) and is the identity once lo is on that grid, so the voxels that matter
      // are the ones with the largest |p| per axis so far, wherever in the list they were.  The guess therefore walks an
      // evenly spaced SAMPLE of everything before the chunk (it meets every y and z the scan has been through, and whole
      // ranges of x) and then the kWarm voxels right before it; the check below decides whether that was enough.
      const size_t head = c.t0 > kWarm ? c.t0 - kWarm : 0;
      if (head > 0) {
        const size_t step = std::max<size_t>(1, head / kSample);
        for (size_t t = 0; t < head; t += step) run_chain(t, t + 1, st, nullptr);
      }
      run_chain(head, c.t0, st, nullptr);
      // (test hook VCY_TEST_XV_BAD_GUESS: every guess is off by one ulp, so that the check and the second pass run)
      static const bool bad_guess = std::getenv("VCY_TEST_XV_BAD_GUESS") != nullptr;
      if (bad_guess) st[0] = std::nextafterf(st[0], 0.0f);
    }
    std::memcpy(c.assumed, st, sizeof(st));
    float piece[6 * kPiece + 4];  // (+ the float fill()'s last vector load reads past the corners)
    piece[6 * kPiece] = 0.0f;
    for (size_t t = c.t0; t < c.t1; t += kPiece) {
      const size_t te = std::min(c.t1, t + kPiece);
      run_chain(t, te, st, piece);
      fill(t, te, piece);
    }
    std::memcpy(c.left, st, sizeof(st));
  };
  for (size_t w = 0; w < nchunks; ++w) chunks[w].t0 = std::min(kept, w * per), chunks[w].t1 = std::min(kept, chunks[w].t0 + per);
  if (nthreads <= 1) {
    do_chunk(chunks[0], start);
  } else {
    std::vector<std::thread> pool;
    for (size_t w = 0; w < nchunks; ++w)
      if (chunks[w].t1 > chunks[w].t0) pool.emplace_back([&, w]() { do_chunk(chunks[w], w == 0 ? start : nullptr); });
    for (std::thread& th : pool) th.join();
  }
  const double t_c = xv_now();
  // in order: did every chunk start from what its predecessor left?  (bit patterns: -0.0f and 0.0f are different states)
  size_t redone = 0;
  for (size_t w = 1; w < nchunks; ++w) {
    if (chunks[w].t1 <= chunks[w].t0) continue;
    if (std::memcmp(chunks[w].assumed, chunks[w - 1].left, sizeof(float) * 6) != 0) {
      do_chunk(chunks[w], chunks[w - 1].left);
      ++redone;
    }
  }
  if (xv_timing())
    std::fprintf(stderr, "vcy xv: %zu kept voxels: host buffers %.2f ms, chain + fill in %zu chunks on %zu threads %.2f ms, "
                         "check + %zu chunks done again %.2f ms\n",
                 kept, t_b - t_a, nchunks, nthreads, t_c - t_b, redone, xv_now() - t_c);
  if (arrays) out->vertices = nullptr, out->faces = nullptr;  // (the caller's arrays: nothing for vcy_mesh_free here)
  return VCY_OK;
}

extern "C" int vcy_extract_voxel_ids(vcy_ctx* c, int inside_empty, int64_t** ids_out, int64_t* n_out) {
  using namespace vcy;
  if (!c || !ids_out || !n_out) {
    set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  *ids_out = nullptr;
  *n_out = 0;
  HostIds ids;
  const int rc = kept_voxel_ids(c, inside_empty, &ids);
  if (rc != VCY_OK) return rc;
  if (ids.n == 0) return VCY_OK;
  *n_out = (int64_t)ids.n;
  *ids_out = ids.release();  // (freed by vcy_ids_free: back to the pool)
  return VCY_OK;
}

extern "C" void vcy_ids_free(int64_t* ids) { vcy::mesh_host_free(ids); }

static int voxel_cubes_impl(const vcy_carver_option* o, int64_t n_ids, const int64_t* ids, vcy_mesh* out,
                            vcy_mesh_arrays_fn arrays, void* user) {
  using namespace vcy;
  if (!o || !out || n_ids < 0 || (n_ids > 0 && !ids)) {
    set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  std::memset(out, 0, sizeof(*out));
  int32_t dims[3];
  int rc = vcy_compute_dims(o->bb_min, o->bb_max, o->resolution, dims);
  if (rc != VCY_OK) return rc;
  if (dims[0] <= 0 || dims[1] <= 0 || dims[2] <= 0) {
    if (n_ids == 0) return VCY_OK;
    set_error("voxel ids for an empty grid");
    return VCY_ERR_INVALID_ARG;
  }
  const int64_t total = (int64_t)dims[0] * dims[1] * dims[2];
  for (int64_t t = 0; t < n_ids; ++t)
    if (ids[t] < 0 || ids[t] >= total) {
      set_error("voxel id %lld outside the grid", (long long)ids[t]);
      return VCY_ERR_INVALID_ARG;
    }
  std::vector<float> axis[3];
  for (int a = 0; a < 3; ++a) {
    axis[a].resize((size_t)dims[a]);
    rc = vcy_axis_positions(o->bb_min, o->bb_max, o->resolution, a, axis[a].data());
    if (rc != VCY_OK) return rc;
  }
  return cubes_from_ids(axis[0].data(), axis[1].data(), axis[2].data(), dims[0], dims[1], o->resolution, ids, (size_t)n_ids, out,
                        arrays, user);
}

extern "C" int vcy_voxel_cubes(const vcy_carver_option* o, int64_t n_ids, const int64_t* ids, vcy_mesh* out) {
  return voxel_cubes_impl(o, n_ids, ids, out, nullptr, nullptr);
}

extern "C" int vcy_voxel_cubes_into(const vcy_carver_option* o, int64_t n_ids, const int64_t* ids, vcy_mesh_arrays_fn arrays,
                                    void* user) {
  if (!arrays) {
    vcy::set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  vcy_mesh counts;
  return voxel_cubes_impl(o, n_ids, ids, &counts, arrays, user);
}

static int extract_voxel_impl(vcy_ctx* c, int inside_empty, vcy_mesh* out, vcy_mesh_arrays_fn arrays, void* user) {
  using namespace vcy;
  if (!c || !out) {
    set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  std::memset(out, 0, sizeof(*out));
  if (c->z0 != 0 || c->z1 != c->nz) {
    // ONE cube drifts through the kept voxels of the whole grid (extract_voxel.cc:290-311): the slabs' id lists
    // are concatenated in z order and walked by vcy_voxel_cubes (ShardedVoxelCarver::ExtractVoxel)
    set_error("vcy_extract_voxel needs the whole grid in one context; for z-slabs: vcy_extract_voxel_ids per slab, then vcy_voxel_cubes");
    return VCY_ERR_UNSUPPORTED;
  }
  HostIds ids;
  const double t_a = xv_now();
  const int rc = kept_voxel_ids(c, inside_empty, &ids);
  if (rc != VCY_OK) return rc;
  if (xv_timing()) std::fprintf(stderr, "vcy xv: kept voxel ids (device predicate + compaction + D2H) %.2f ms\n", xv_now() - t_a);
  std::vector<float> py((size_t)c->ny);
  VCY_HIP_CHECK(hipMemcpy(py.data(), c->d_py, sizeof(float) * (size_t)c->ny, hipMemcpyDeviceToHost));
  return cubes_from_ids(c->h_px, py.data(), c->h_pz, c->nx, c->ny, c->opt.resolution, ids.p, ids.n, out, arrays, user);
}

extern "C" int vcy_extract_voxel(vcy_ctx* c, int inside_empty, vcy_mesh* out) {
  return extract_voxel_impl(c, inside_empty, out, nullptr, nullptr);
}

extern "C" int vcy_extract_voxel_into(vcy_ctx* c, int inside_empty, vcy_mesh_arrays_fn arrays, void* user) {
  if (!arrays) {
    vcy::set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  vcy_mesh counts;
  return extract_voxel_impl(c, inside_empty, &counts, arrays, user);
}
