// C-ABI entry points of libvacancy_hip.so: lifetime, state access, halo, helpers.
// The carving and extraction kernels live in carve_kernels.hip / mc_kernels.hip.
#include <algorithm>
#include <cstdarg>
#include <cstring>
#include <chrono>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

#include "vcy_internal.h"
#include "build_id.h"  // VCY_SOURCE_HASH, written by the Makefile

namespace vcy {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

// ---- host buffers of returned meshes ---------------------------------------------------------
// The arrays of a vcy_mesh are page-locked host memory from a small process-wide pool: the mesh download
// (130 MB at 1024^3) is then one DMA at PCIe rate instead of a staged copy into freshly faulted pages, and
// vcy_mesh_free hands the buffers back for the next extraction.  Pageable memory is the fallback when
// pinning fails.
namespace {
struct HostBuf { void* p; size_t bytes; bool pinned; };
std::mutex g_mesh_mutex;
std::vector<HostBuf> g_mesh_live, g_mesh_idle;
constexpr size_t kMeshIdleCap = (size_t)3 << 30;  // idle bytes kept for reuse
}  // namespace

void* mesh_host_alloc(size_t bytes, bool* pinned_out) {
  if (pinned_out) *pinned_out = false;
  if (bytes == 0) bytes = 16;
  std::lock_guard<std::mutex> lock(g_mesh_mutex);
  size_t best = g_mesh_idle.size();
  for (size_t i = 0; i < g_mesh_idle.size(); ++i)
    if (g_mesh_idle[i].bytes >= bytes && g_mesh_idle[i].bytes <= 2 * bytes + 4096 &&
        (best == g_mesh_idle.size() || g_mesh_idle[i].bytes < g_mesh_idle[best].bytes))
      best = i;
  HostBuf b;
  if (best < g_mesh_idle.size()) {
    b = g_mesh_idle[best];
    g_mesh_idle.erase(g_mesh_idle.begin() + (long)best);
  } else {
    b.bytes = bytes + bytes / 8 + 4096;  // headroom: the next view's mesh is usually a little different
    b.p = nullptr;
    // (portable + mapped: the pool is shared by the contexts of every device of the process, and mc_emit writes small
    // meshes into these arrays from whichever device extracts -- "mcdirect")
    b.pinned = hipHostMalloc(&b.p, b.bytes, hipHostMallocPortable | hipHostMallocMapped) == hipSuccess && b.p != nullptr;
    if (!b.pinned) {
      (void)hipGetLastError();
      b.p = std::malloc(b.bytes);
      if (!b.p) return nullptr;
    }
  }
  g_mesh_live.push_back(b);
  if (pinned_out) *pinned_out = b.pinned;  // (page-locked: kernels can write it directly, mc_emit)
  return b.p;
}

void mesh_host_free(void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> lock(g_mesh_mutex);
  for (size_t i = 0; i < g_mesh_live.size(); ++i) {
    if (g_mesh_live[i].p != p) continue;
    const HostBuf b = g_mesh_live[i];
    g_mesh_live.erase(g_mesh_live.begin() + (long)i);
    size_t idle = 0;
    for (const HostBuf& q : g_mesh_idle) idle += q.bytes;
    if (b.pinned && idle + b.bytes <= kMeshIdleCap) {
      g_mesh_idle.push_back(b);
    } else if (b.pinned) {
      (void)hipHostFree(b.p);
    } else {
      std::free(b.p);
    }
    return;
  }
  std::free(p);  // not ours (never happens for meshes this library returned)
}

// sdf = lowest(), update_num = 0 over slab + halo (reference voxel_carver.cc:339, Voxel ctor)
__global__ void fill_f32_kernel(float* __restrict__ p, float v, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

// update_num from one counter width to another (lazy widening of vcy_ctx::d_cnt; halo packs travel in the final
// width).  Narrowing saturates: it only happens to the two halo slices a slab receives, whose counters are read as
// `update_num >= 1` and nothing else (marching_cubes.cc:88-90, extract_voxel.cc:283-286).
template <typename S, typename D>
__global__ __launch_bounds__(256) void convert_counts_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t n) {
  int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  constexpr unsigned cap = sizeof(D) == 1 ? 255u : (sizeof(D) == 2 ? 65535u : 0xffffffffu);
  for (; i < n; i += stride) {
    if (i + 4 <= n) {
      S v[4];
      __builtin_memcpy(v, src + i, sizeof(v));  // (both arrays are 16-byte aligned and i is a multiple of 4)
      D o[4];
      for (int k = 0; k < 4; ++k) o[k] = (D)min((unsigned)v[k], cap);
      __builtin_memcpy(dst + i, o, sizeof(o));
    } else {
      for (int64_t k = i; k < n; ++k) dst[k] = (D)min((unsigned)src[k], cap);
    }
  }
}

int convert_counts(hipStream_t stream, const void* src, int sb, void* dst, int db, int64_t n) {
  if (n <= 0) return VCY_OK;
  if (sb == db) {
    VCY_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)n * sb, hipMemcpyDeviceToDevice, stream));
    return VCY_OK;
  }
  const dim3 grid((unsigned)std::min<int64_t>((n + 1023) / 1024, 256 * 32));
#define VCY_CONV(S, D) hipLaunchKernelGGL((convert_counts_kernel<S, D>), grid, dim3(256), 0, stream, (const S*)src, (D*)dst, n)
  if (sb == 1 && db == 2) VCY_CONV(uint8_t, uint16_t);
  else if (sb == 1 && db == 4) VCY_CONV(uint8_t, uint32_t);
  else if (sb == 2 && db == 4) VCY_CONV(uint16_t, uint32_t);
  else if (sb == 2 && db == 1) VCY_CONV(uint16_t, uint8_t);
  else if (sb == 4 && db == 1) VCY_CONV(uint32_t, uint8_t);
  else if (sb == 4 && db == 2) VCY_CONV(uint32_t, uint16_t);
  else {
    set_error("convert_counts: unsupported widths %d -> %d", sb, db);
    return VCY_ERR_INTERNAL;
  }
#undef VCY_CONV
  VCY_HIP_CHECK(hipGetLastError());
  return VCY_OK;
}

// Bytes a counter needs to hold values up to max_count (never more than the final width of the options).
int count_width_for(const vcy_ctx* c, int64_t max_count) {
  if (!c->lazy_count) return c->cnt_bytes_wire;
  const int64_t cap = (int64_t)c->opt.update_option.voxel_max_update_num + 1;  // voxel_carver.cc:447-450
  const int64_t m = std::min(max_count, cap);
  const int w = m <= 255 ? 1 : (m <= 65535 ? 2 : 4);
  return std::min(w, c->cnt_bytes_wire);
}

// Switches d_cnt to `bytes` per counter, converting what it holds (nothing on a fresh slab).  The array of the other
// width is KEPT (d_cnt_spare) once both exist: a vcy_reset followed by a carve across the 256th view used to pay two
// allocations of 1 - 2 GB, two device-wide synchronisations (hipFree) and a pipeline stall per cycle.  The conversion is
// ordered on the context's stream like every other access to the counters, so nothing waits here either.
static int set_count_width(vcy_ctx* c, int bytes) {
  if (bytes == c->cnt_bytes) return VCY_OK;
  const int64_t nvox = c->slice * (int64_t)(c->halo_lo + c->nz_local());
  const size_t need = (size_t)nvox * bytes;
  void* d_new = nullptr;
  size_t new_cap = 0;
  if (c->d_cnt_spare && c->cnt_spare_cap >= need) {
    d_new = c->d_cnt_spare;
    new_cap = c->cnt_spare_cap;
    c->d_cnt_spare = nullptr;
    c->cnt_spare_cap = 0;
  } else {
    VCY_HIP_CHECK(hipMalloc(&d_new, need));
    new_cap = need;
  }
  int rc = VCY_OK;
  if (!c->fresh) {
    rc = convert_counts(c->stream, c->d_cnt, c->cnt_bytes, d_new, bytes, nvox);
  } else if (c->halo_lo && c->halo_valid) {
    rc = convert_counts(c->stream, c->d_cnt, c->cnt_bytes, d_new, bytes, c->slice * (int64_t)c->halo_lo);
  }
  if (rc != VCY_OK) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(d_new);
    return rc;
  }
  // the old array becomes the spare (a smaller spare that was passed over goes back to the allocator)
  if (c->d_cnt_spare) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->d_cnt_spare);
  }
  c->d_cnt_spare = c->d_cnt;
  c->cnt_spare_cap = c->cnt_cap;
  c->d_cnt = d_new;
  c->cnt_cap = new_cap;
  c->cnt_bytes = bytes;
  return VCY_OK;
}

int ensure_count_width(vcy_ctx* c, int64_t max_count) {
  const int w = count_width_for(c, max_count);
  if (w <= c->cnt_bytes) return VCY_OK;
  return set_count_width(c, w);
}

static void discard_pending(vcy_ctx* c) {
  for (auto& t : c->pending) c->sdf_pool.emplace_back(t.d_sdf, t.bytes);
  c->pending.clear();
}

int fill_state(vcy_ctx* c) {
  discard_pending(c);  // whatever they would have carved is wiped
  c->deferred_rc = VCY_OK;
  c->deferred_msg.clear();
  c->fresh = true;  // written lazily, see vcy_ctx::fresh
  c->brick_min_valid = false;
  if (c->h_live_hint) c->h_live_hint[0] = c->h_live_hint[1] = 0;
  c->views_carved = 0;
  c->halo_valid = false;
  c->cnt_implied = true;
  // counters start over at one byte (fresh: nothing to convert; the wide array is kept as the spare)
  if (c->d_cnt && c->cnt_bytes != count_width_for(c, 0)) return set_count_width(c, count_width_for(c, 0));
  return VCY_OK;
}

int materialize(vcy_ctx* c) {
  {
    const int rcf = flush_pending(c);  // every reader of the state comes through here
    if (rcf != VCY_OK) return rcf;
  }
  if (!c->fresh) return VCY_OK;
  // only the owned slab: halo slices are written by vcy_halo_install / _unpack
  const int64_t n = c->slab_voxels();
  const int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(fill_f32_kernel, dim3(grid), dim3(256), 0, c->stream, c->owned_slab_sdf(), kInvalidSdf, n);
  VCY_HIP_CHECK(hipGetLastError());
  VCY_HIP_CHECK(hipMemsetAsync(c->owned_slab_cnt(), 0, (size_t)n * c->cnt_bytes, c->stream));
  c->fresh = false;
  return VCY_OK;
}

// Next slot of the carve timer's event log (vcy_set_param "carvetimer"); the events of a slot are created once and
// re-used after the log has been cleared.  -1 when the log is full (the launch is then simply not recorded) or an
// event cannot be created.
constexpr int kCarveLogMax = 8192;
int carve_log_open(vcy_ctx* c, bool first_chunk) {
  if (first_chunk) c->carve_log_last_dropped = false;
  if (c->carve_log_n >= kCarveLogMax) {
    ++c->carve_log_dropped;
    c->carve_log_last_dropped = true;
    return -1;
  }
  if ((size_t)c->carve_log_n == c->carve_log.size()) {
    vcy_ctx::CarveStamp st{{nullptr, nullptr, nullptr}, false};
    for (int k = 0; k < 3; ++k)
      if (hipEventCreate(&st.ev[k]) != hipSuccess) {
        (void)hipGetLastError();
        for (int q = 0; q < k; ++q) (void)hipEventDestroy(st.ev[q]);
        return -1;
      }
    c->carve_log.push_back(st);
  }
  const int i = c->carve_log_n++;
  c->carve_log[(size_t)i].first_chunk = first_chunk;
  if (first_chunk) c->carve_log_last = i;
  return i;
}

// Voxel::pos along one axis, reference voxel_carver.cc:308-326:
//   pos = diff * ((float)i / (float)n) + bb_min + resolution * 0.5f   (left to right)
// Host IEEE arithmetic, this TU is built with -ffp-contract=off.  The ONE place the expression lives: the device's
// axis tables (vcy_create), vcy_axis_positions and through it the C++ facade's VoxelGrid come from here.
static void axis_positions(float bb_min, float bb_max, float resolution, int n, float* out) {
  const float offset = resolution * 0.5f;
  const float diff = bb_max - bb_min;
  for (int i = 0; i < n; ++i) out[i] = diff * (static_cast<float>(i) / static_cast<float>(n)) + bb_min + offset;
}

static int dims_from_option(const float bb_min[3], const float bb_max[3], float res, int32_t n[3],
                            bool allow_empty = false) {
  // VoxelGrid::Init, reference voxel_carver.cc:278-301
  if (res < std::numeric_limits<float>::min()) {
    set_error("resolution must be positive %f", res);
    return VCY_ERR_INVALID_ARG;
  }
  if (bb_max[0] <= bb_min[0] || bb_max[1] <= bb_min[1] || bb_max[2] <= bb_min[2]) {
    set_error("input bounding box is invalid");
    return VCY_ERR_INVALID_ARG;
  }
  for (int i = 0; i < 3; ++i) {
    const float diff = bb_max[i] - bb_min[i];
    n[i] = static_cast<int>(diff / res);
  }
  if (n[0] <= 0 || n[1] <= 0 || n[2] <= 0) {
    // The reference accepts a box thinner than one voxel and builds an EMPTY grid (voxel_carver.cc:292-345: the
    // loops do not run, Init returns true); the host container does the same (vcy_compute_dims, VoxelGrid::Init).
    // A device context over no voxels is refused (vcy_create).
    if (allow_empty) return VCY_OK;
    set_error("grid has an empty axis (%d,%d,%d)", n[0], n[1], n[2]);
    return VCY_ERR_INVALID_ARG;
  }
  // The reference refuses more than INT_MAX voxels (32-bit ids, voxel_carver.cc:298-301).
  // Device indices are 64-bit; a single xy slice must still fit 31 bits.
  if ((int64_t)n[0] * n[1] > std::numeric_limits<int>::max()) {
    set_error("too many voxels in one xy slice");
    return VCY_ERR_TOO_MANY_VOXELS;
  }
  return VCY_OK;
}

}  // namespace vcy

using namespace vcy;

namespace {
std::mutex g_ctx_count_mutex;
int g_ctx_count = 0;
// idle page-locked mesh buffers are only worth keeping while a context may extract again
void mesh_pool_trim() {
  std::lock_guard<std::mutex> lock(g_mesh_mutex);
  for (const HostBuf& b : g_mesh_idle) (void)hipHostFree(b.p);
  g_mesh_idle.clear();
}
}  // namespace

extern "C" {

static int check_view_static(const vcy_view* v);

const char* vcy_last_error(void) { return g_last_error.c_str(); }
const char* vcy_version(void) { return "vacancy_amd 0.3 (gfx950) src:" VCY_SOURCE_HASH; }

int vcy_device_count(int* count) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return VCY_ERR_NO_DEVICE;
  }
  *count = n;
  return VCY_OK;
}

int vcy_compute_dims(const float bb_min[3], const float bb_max[3], float resolution,
                     int32_t dims[3]) {
  return dims_from_option(bb_min, bb_max, resolution, dims, true);
}

int vcy_axis_positions(const float bb_min[3], const float bb_max[3], float resolution, int axis, float* out) {
  if (!bb_min || !bb_max || !out || axis < 0 || axis > 2) {
    set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  int32_t n[3];
  const int rc = dims_from_option(bb_min, bb_max, resolution, n);
  if (rc != VCY_OK) return rc;
  axis_positions(bb_min[axis], bb_max[axis], resolution, n[axis], out);
  return VCY_OK;
}

int vcy_create(const vcy_carver_option* o, int device_id, int z_begin, int z_end, vcy_ctx** out) {
  if (!o || !out) {
    set_error("null argument");
    return VCY_ERR_INVALID_ARG;
  }
  *out = nullptr;
  // VoxelCarver::Init, reference voxel_carver.cc:376-389
  if (o->update_option.voxel_max_update_num < 1) {
    set_error("voxel_max_update_num must be positive");
    return VCY_ERR_INVALID_ARG;
  }
  if (o->update_option.voxel_update_weight < std::numeric_limits<float>::min()) {
    set_error("voxel_update_weight must be positive");
    return VCY_ERR_INVALID_ARG;
  }
  if (o->update_option.truncation_band < std::numeric_limits<float>::min()) {
    set_error("truncation_band must be positive");
    return VCY_ERR_INVALID_ARG;
  }
  const vcy_update_option& u = o->update_option;
  if (u.voxel_update < 0 || u.voxel_update > 1 || u.sdf_interp < 0 || u.sdf_interp > 1 ||
      u.update_outside < 0 || u.update_outside > 1) {
    set_error("unknown enum value in update_option");
    return VCY_ERR_INVALID_ARG;
  }
  int32_t n[3];
  int rc = dims_from_option(o->bb_min, o->bb_max, o->resolution, n);
  if (rc != VCY_OK) return rc;
  if (z_end < 0) z_end = n[2];
  if (z_begin < 0 || z_begin >= z_end || z_end > n[2]) {
    set_error("invalid z-slab [%d,%d) for nz=%d", z_begin, z_end, n[2]);
    return VCY_ERR_INVALID_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("no HIP device available (there is no CPU fallback)");
    return VCY_ERR_NO_DEVICE;
  }
  if (device_id < 0 || device_id >= ndev) {
    set_error("device %d out of range (%d devices)", device_id, ndev);
    return VCY_ERR_NO_DEVICE;
  }
  VCY_HIP_CHECK(hipSetDevice(device_id));

  vcy_ctx* c = new vcy_ctx;
  c->device = device_id;
  c->opt = *o;
  c->nx = n[0];
  c->ny = n[1];
  c->nz = n[2];
  c->z0 = z_begin;
  c->z1 = z_end;
  c->slice = (int64_t)n[0] * n[1];
  c->halo_lo = (z_begin > 0) ? 2 : 0;
  if (c->halo_lo && z_begin < 2) {
    delete c;
    set_error("a non-first slab must start at z >= 2");
    return VCY_ERR_INVALID_ARG;
  }
  // update_num never exceeds voxel_max_update_num + 1 (voxel_carver.cc:447-450)
  const int64_t max_cnt = (int64_t)u.voxel_max_update_num + 1;
  c->cnt_bytes_wire = max_cnt <= 255 ? 1 : (max_cnt <= 65535 ? 2 : 4);
  c->cnt_bytes = 1;  // widened when the views applied (or uploaded counts) need it, ensure_count_width
  {
    const char* e = std::getenv("VCY_MC_TIMING");  // development aid: host-side phases of every extraction on stderr
    if (e && e[0] == '1') c->mc_timing = 1;
  }

  auto fail = [&](int code) {
    vcy_destroy(c);
    return code;
  };
#define VCY_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      set_error("%s failed: %s", #expr, hipGetErrorString(_e));                               \
      return fail(VCY_ERR_HIP);                                                               \
    }                                                                                         \
  } while (0)
  VCY_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  c->own_stream = true;
  VCY_TRY(hipEventCreate(&c->ev_begin));
  VCY_TRY(hipEventCreate(&c->ev_end));
  const int64_t nvox = c->slice * (int64_t)(c->halo_lo + c->nz_local());
  VCY_TRY(hipMalloc(&c->d_sdf, (size_t)nvox * sizeof(float)));
  VCY_TRY(hipMalloc(&c->d_cnt, (size_t)nvox * c->cnt_bytes));
  c->cnt_cap = (size_t)nvox * c->cnt_bytes;
  VCY_TRY(hipMalloc(&c->d_px, sizeof(float) * n[0]));
  VCY_TRY(hipMalloc(&c->d_py, sizeof(float) * n[1]));
  VCY_TRY(hipMalloc(&c->d_pz, sizeof(float) * n[2]));

  // Voxel::pos per axis (axis_positions above)
  float* d_axis[3] = {c->d_px, c->d_py, c->d_pz};
  for (int a = 0; a < 3; ++a) {
    std::vector<float> p(n[a]);
    axis_positions(o->bb_min[a], o->bb_max[a], o->resolution, n[a], p.data());
    VCY_TRY(hipMemcpy(d_axis[a], p.data(), sizeof(float) * n[a], hipMemcpyHostToDevice));
    if (a == 0) {
      c->h_px = new float[n[0]];
      std::memcpy(c->h_px, p.data(), sizeof(float) * n[0]);
      c->h_px_min = p.front();
      c->h_px_max = p.back();
    } else if (a == 1) {
      c->h_py_min = p.front();
      c->h_py_max = p.back();
    }
    if (a == 2) {
      c->h_pz = new float[n[2]];
      std::memcpy(c->h_pz, p.data(), sizeof(float) * n[2]);
    }
  }
  rc = fill_state(c);
  if (rc != VCY_OK) return fail(rc);
  VCY_TRY(hipStreamSynchronize(c->stream));
#undef VCY_TRY
  {
    std::lock_guard<std::mutex> lock(g_ctx_count_mutex);
    ++g_ctx_count;
    c->counted = true;
  }
  *out = c;
  return VCY_OK;
}


void vcy_destroy(vcy_ctx* c) {
  if (!c) return;
  {
    std::lock_guard<std::mutex> lock(g_ctx_count_mutex);
    if (c->counted && --g_ctx_count == 0) mesh_pool_trim();
  }
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  (void)hipFree(c->d_sdf);
  (void)hipFree(c->d_cnt);
  (void)hipFree(c->d_cnt_spare);
  (void)hipFree(c->d_halo_tmp);
  (void)hipFree(c->d_xv_scratch);
  (void)hipFree(c->d_xv_ids);
  (void)hipFree(c->d_px);
  (void)hipFree(c->d_py);
  (void)hipFree(c->d_pz);
  (void)hipFree(c->d_mc_tables);
  (void)hipFree(c->d_mc_scratch);
  (void)hipFree(c->d_stream_pool);
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
  for (int k = 0; k < 2; ++k) {
    if (c->ev_ready[k]) (void)hipEventDestroy(c->ev_ready[k]);
    if (c->ev_consumed[k]) (void)hipEventDestroy(c->ev_consumed[k]);
    if (c->ev_uploaded[k]) (void)hipEventDestroy(c->ev_uploaded[k]);
  }
  (void)hipFree(c->d_mc_out);
  (void)hipFree(c->d_mc_flags);
  if (c->h_mc_report) (void)hipHostFree(c->h_mc_report);
  (void)hipFree(c->d_mc_cells);
  (void)hipFree(c->d_fused_scratch);
  for (int q = 0; q < 2; ++q) {
    if (c->ev_fused_stage[q]) (void)hipEventDestroy(c->ev_fused_stage[q]);
    if (c->h_fused_stage[q]) (void)hipHostFree(c->h_fused_stage[q]);
  }
  (void)hipFree(c->d_wmax);
  (void)hipFree(c->d_records);
  (void)hipFree(c->d_wg_list);
  (void)hipFree(c->d_pair_count);
  for (hipEvent_t ev : c->stream_events) (void)hipEventDestroy(ev);
  if (c->h_live_hint) (void)hipHostFree(c->h_live_hint);
  for (auto& st : c->carve_log)
    for (int k = 0; k < 3; ++k)
      if (st.ev[k]) (void)hipEventDestroy(st.ev[k]);
  (void)hipFree(c->d_brick_min);
  for (auto& t : c->pending) (void)hipFree(t.d_sdf);
  for (auto& t : c->sdf_pool) (void)hipFree(t.first);
  (void)hipFree(c->d_sil_scratch);
  delete[] c->h_pz;
  delete[] c->h_px;
  if (c->ev_mc_begin) (void)hipEventDestroy(c->ev_mc_begin);
  if (c->ev_mc_end) (void)hipEventDestroy(c->ev_mc_end);
  if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
  if (c->ev_end) (void)hipEventDestroy(c->ev_end);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int vcy_grid_dims(const vcy_ctx* c, int32_t dims[3]) {
  if (!c) return VCY_ERR_NOT_INITIALIZED;
  dims[0] = c->nx;
  dims[1] = c->ny;
  dims[2] = c->nz;
  return VCY_OK;
}

int vcy_slab_range(const vcy_ctx* c, int32_t zr[2]) {
  if (!c) return VCY_ERR_NOT_INITIALIZED;
  zr[0] = c->z0;
  zr[1] = c->z1;
  return VCY_OK;
}

int vcy_set_stream(vcy_ctx* c, void* s) {
  if (!c) return VCY_ERR_NOT_INITIALIZED;
  VCY_HIP_CHECK(hipSetDevice(c->device));
  { const int rcf = flush_pending(c); if (rcf != VCY_OK) return rcf; }
  if (c->stream) VCY_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (c->own_stream && c->stream) VCY_HIP_CHECK(hipStreamDestroy(c->stream));
  c->stream = (hipStream_t)s;
  c->own_stream = false;
  return VCY_OK;
}

int vcy_set_param(vcy_ctx* c, const char* name, int value) {
  if (!c || !name) return VCY_ERR_INVALID_ARG;
  if (std::strcmp(name, "inject_carve_failure") == 0) {  // test hook; does not touch the queue
    c->inject_fail = value > 0 ? value : 0;
    return VCY_OK;
  }
  { const int rcf = flush_pending(c); if (rcf != VCY_OK) return rcf; }  // queued views keep the old setting
  if (std::strcmp(name, "defer") == 0) {
    c->defer = value != 0;
    return VCY_OK;
  }
  if (std::strcmp(name, "fused") == 0) {
    c->use_fused = value != 0;
    return VCY_OK;
  }
  if (std::strcmp(name, "tile") == 0) {
    if (value < 0 || value > 2) return VCY_ERR_INVALID_ARG;
    c->tile_mode = value;
    return VCY_OK;
  }
  if (std::strcmp(name, "shortdiv") == 0) {
    c->use_short_div = value != 0;
    return VCY_OK;
  }
  if (std::strcmp(name, "cull") == 0) {
    c->use_cull = value != 0;
    return VCY_OK;
  }
  if (std::strcmp(name, "mcsweep") == 0) {
    c->mc_sweep = value != 0;
    return VCY_OK;
  }
  if (std::strcmp(name, "carvetimer") == 0) {
    c->time_carve = value != 0;
    c->carve_log_n = c->carve_log_last = c->carve_log_dropped = 0;  // (the log starts over)
    c->carve_log_last_dropped = false;
    return VCY_OK;
  }
  if (std::strcmp(name, "paircount") == 0) {
    c->count_pairs = value != 0;
    return VCY_OK;
  }
  if (std::strcmp(name, "recordbytes") == 0) {
    c->record_bytes_max = value > 0 ? value : 0;
    return VCY_OK;
  }
  if (std::strcmp(name, "livelist") == 0) {
    c->use_live_list = value != 0;
    return VCY_OK;
  }
  if (std::strcmp(name, "prologue") == 0) {
    if (value < 0 || value > 2) return VCY_ERR_INVALID_ARG;
    c->prologue_mode = value;
    return VCY_OK;
  }
  if (std::strcmp(name, "livesync") == 0) {
    c->live_sync = value != 0;
    return VCY_OK;
  }
  if (std::strcmp(name, "coopstore") == 0) {
    c->coop_store = value < 0 ? -1 : (value != 0 ? 1 : 0);
    return VCY_OK;
  }
  if (std::strcmp(name, "listrecords") == 0) {
    c->list_records = value != 0;
    return VCY_OK;
  }
  if (std::strcmp(name, "eagerstate") == 0) {
    c->eager_state = value < 0 ? -1 : (value != 0 ? 1 : 0);
    return VCY_OK;
  }
  if (std::strcmp(name, "oneview") == 0) {
    c->one_view = value != 0 ? 1 : 0;
    return VCY_OK;
  }
  if (std::strcmp(name, "ntstore") == 0) {
    c->nt_store = value < 0 ? -1 : (value != 0 ? 1 : 0);
    return VCY_OK;
  }
  if (std::strcmp(name, "rowkernel") == 0) {
    c->row_kernel = value < 0 ? -1 : value;
    return VCY_OK;
  }
  if (std::strcmp(name, "mcskip") == 0) {
    c->mc_skip = value <= 0 ? 0 : (value >= 2 ? 2 : 1);
    return VCY_OK;
  }
  if (std::strcmp(name, "meshkeys") == 0) {
    c->mesh_keys = value != 0;
    return VCY_OK;
  }
  if (std::strcmp(name, "mcdirect") == 0) {  // bytes (of the guessed mesh) up to which mc_emit writes host memory directly
    c->mc_direct_bytes = value < 0 ? 0 : (int64_t)value;
    return VCY_OK;
  }
  if (std::strcmp(name, "mctiming") == 0) {
    c->mc_timing = value != 0 ? 1 : 0;
    return VCY_OK;
  }
  if (std::strcmp(name, "lazycount") == 0) {  // 0: counters at their final width from now on (round 4's layout)
    VCY_HIP_CHECK(hipSetDevice(c->device));
    c->lazy_count = value != 0;
    if (!c->lazy_count) return set_count_width(c, c->cnt_bytes_wire);
    return VCY_OK;
  }
  set_error("unknown parameter %s", name);
  return VCY_ERR_INVALID_ARG;
}

int vcy_get_param(vcy_ctx* c, const char* name, int* value) {
  if (!c || !name || !value) return VCY_ERR_INVALID_ARG;
  if (std::strcmp(name, "fused") == 0) *value = c->use_fused ? 1 : 0;
  else if (std::strcmp(name, "cull") == 0) *value = c->use_cull ? 1 : 0;
  else if (std::strcmp(name, "tile") == 0) *value = c->tile_mode;
  else if (std::strcmp(name, "defer") == 0) *value = c->defer ? 1 : 0;
  else if (std::strcmp(name, "shortdiv") == 0) *value = c->use_short_div ? 1 : 0;
  else if (std::strcmp(name, "div_level") == 0) *value = c->last_div_level;
  else if (std::strcmp(name, "mcsweep") == 0) *value = c->mc_sweep ? 1 : 0;
  else if (std::strcmp(name, "mcskip") == 0) *value = c->mc_skip;
  else if (std::strcmp(name, "rowkernel") == 0) *value = c->row_kernel;
  else if (std::strcmp(name, "ntstore") == 0) *value = c->nt_store;
  else if (std::strcmp(name, "oneview") == 0) *value = c->one_view;
  else if (std::strcmp(name, "eagerstate") == 0) *value = c->eager_state;
  else if (std::strcmp(name, "listrecords") == 0) *value = c->list_records;
  else if (std::strcmp(name, "mcdirect") == 0) *value = (int)std::min<int64_t>(c->mc_direct_bytes, 0x7fffffff);
  else if (std::strcmp(name, "livelist") == 0) *value = c->use_live_list ? 1 : 0;
  else if (std::strcmp(name, "prologue") == 0) *value = c->prologue_mode;
  else if (std::strcmp(name, "coopstore") == 0) *value = c->coop_store;
  else if (std::strcmp(name, "livesync") == 0) *value = c->live_sync ? 1 : 0;
  else if (std::strcmp(name, "brick_min_valid") == 0) *value = c->brick_min_valid && !c->fresh ? 1 : 0;
  else if (std::strcmp(name, "meshkeys") == 0) *value = c->mesh_keys ? 1 : 0;
  else if (std::strcmp(name, "lazycount") == 0) *value = c->lazy_count ? 1 : 0;
  else if (std::strcmp(name, "carvetimer") == 0) *value = c->time_carve ? 1 : 0;
  else if (std::strcmp(name, "carvelog_dropped") == 0) *value = c->carve_log_dropped;
  else if (std::strcmp(name, "count_bytes") == 0) *value = c->cnt_bytes;
  else if (std::strcmp(name, "count_bytes_final") == 0) *value = c->cnt_bytes_wire;
  else {
    set_error("unknown parameter %s", name);
    return VCY_ERR_INVALID_ARG;
  }
  return VCY_OK;
}

int vcy_get_stream(vcy_ctx* c, void** out) {
  if (!c || !out) return VCY_ERR_INVALID_ARG;
  *out = (void*)c->stream;
  return VCY_OK;
}

int vcy_sync(vcy_ctx* c) {
  if (!c) return VCY_ERR_NOT_INITIALIZED;
  VCY_HIP_CHECK(hipSetDevice(c->device));
  { const int rcf = flush_pending(c); if (rcf != VCY_OK) return rcf; }
  VCY_HIP_CHECK(hipStreamSynchronize(c->stream));
  return VCY_OK;
}

int vcy_timer_begin(vcy_ctx* c) {
  if (!c) return VCY_ERR_NOT_INITIALIZED;
  VCY_HIP_CHECK(hipEventRecord(c->ev_begin, c->stream));
  return VCY_OK;
}

int vcy_timer_end(vcy_ctx* c, float* ms) {
  if (!c) return VCY_ERR_NOT_INITIALIZED;
  { const int rcf = flush_pending(c); if (rcf != VCY_OK) return rcf; }  // queued views belong to the interval
  VCY_HIP_CHECK(hipEventRecord(c->ev_end, c->stream));
  VCY_HIP_CHECK(hipEventSynchronize(c->ev_end));
  VCY_HIP_CHECK(hipEventElapsedTime(ms, c->ev_begin, c->ev_end));
  return VCY_OK;
}

int vcy_last_carve_ms(vcy_ctx* c, float* prepass_ms, float* kernel_ms) {
  if (!c || !prepass_ms || !kernel_ms) return VCY_ERR_INVALID_ARG;
  VCY_HIP_CHECK(hipSetDevice(c->device));
  *prepass_ms = *kernel_ms = 0.0f;
  if (c->carve_log_last_dropped) {  // (never an older launch's times in its place)
    set_error("vcy_last_carve_ms: the event log is full (%d records); read it with vcy_carve_log(clear = 1)", kCarveLogMax);
    return VCY_ERR_INVALID_ARG;
  }
  for (int i = c->carve_log_last; i < c->carve_log_n; ++i) {  // the chunks of the last launch
    const vcy_ctx::CarveStamp& st = c->carve_log[(size_t)i];
    float a = 0.0f, b = 0.0f;
    VCY_HIP_CHECK(hipEventSynchronize(st.ev[2]));
    VCY_HIP_CHECK(hipEventElapsedTime(&a, st.ev[0], st.ev[1]));
    VCY_HIP_CHECK(hipEventElapsedTime(&b, st.ev[1], st.ev[2]));
    *prepass_ms += a;
    *kernel_ms += b;
  }
  return VCY_OK;
}

int vcy_carve_log(vcy_ctx* c, int max_records, float* begin_ms, float* prepass_ms, float* kernel_ms,
                  int32_t* first_chunk, int* n_records, int clear) {
  if (!c || !n_records || max_records < 0) return VCY_ERR_INVALID_ARG;
  VCY_HIP_CHECK(hipSetDevice(c->device));
  const int n = std::min(max_records, c->carve_log_n);
  for (int i = 0; i < n; ++i) {
    const vcy_ctx::CarveStamp& st = c->carve_log[(size_t)i];
    float t0 = 0.0f, a = 0.0f, b = 0.0f;
    VCY_HIP_CHECK(hipEventSynchronize(st.ev[2]));
    if (i > 0) VCY_HIP_CHECK(hipEventElapsedTime(&t0, c->carve_log[0].ev[0], st.ev[0]));
    VCY_HIP_CHECK(hipEventElapsedTime(&a, st.ev[0], st.ev[1]));
    VCY_HIP_CHECK(hipEventElapsedTime(&b, st.ev[1], st.ev[2]));
    if (begin_ms) begin_ms[i] = t0;
    if (prepass_ms) prepass_ms[i] = a;
    if (kernel_ms) kernel_ms[i] = b;
    if (first_chunk) first_chunk[i] = st.first_chunk ? 1 : 0;
  }
  *n_records = n;
  if (clear) {
    c->carve_log_n = c->carve_log_last = c->carve_log_dropped = 0;
    c->carve_log_last_dropped = false;
  }
  return VCY_OK;
}

int vcy_last_carve_pairs(vcy_ctx* c, int64_t* processed, int64_t* total, int64_t* per_layer, int max_layers, int* n_layers) {
  if (!c) return VCY_ERR_NOT_INITIALIZED;
  if (!c->count_pairs || !c->d_pair_count) {
    set_error("vcy_last_carve_pairs: no fused launch since vcy_set_param(\"paircount\", 1)");
    return VCY_ERR_INVALID_ARG;
  }
  VCY_HIP_CHECK(hipSetDevice(c->device));
  const int nbz = (c->nz_local() + 7) / 8;
  std::vector<unsigned long long> h((size_t)nbz, 0ull);
  VCY_HIP_CHECK(hipMemcpyAsync(h.data(), c->d_pair_count, sizeof(unsigned long long) * (size_t)nbz, hipMemcpyDeviceToHost, c->stream));
  VCY_HIP_CHECK(hipStreamSynchronize(c->stream));
  int64_t sum = 0;
  for (int l = 0; l < nbz; ++l) {
    sum += (int64_t)h[(size_t)l];
    if (per_layer && l < max_layers) per_layer[l] = (int64_t)h[(size_t)l];
  }
  if (processed) *processed = sum;
  if (total) *total = (int64_t)((c->nx + 7) / 8) * ((c->ny + 7) / 8) * nbz * c->pair_count_views;
  if (n_layers) *n_layers = nbz;
  return VCY_OK;
}

int vcy_partition_layers(const double* layer_cost, int n_layers, int n_slabs, int nz, int32_t* z_bounds) {
  if (!layer_cost || !z_bounds || n_layers < 1 || n_slabs < 1 || n_slabs > n_layers || nz <= (n_layers - 1) * 8 ||
      nz > n_layers * 8) {
    set_error("vcy_partition_layers: invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  if (n_slabs > 1 && n_layers > 1 && nz - (n_layers - 1) * 8 == 1 && n_slabs > n_layers - 1) {
    set_error("vcy_partition_layers: the last layer is a single slice and cannot be a slab of its own");
    return VCY_ERR_INVALID_ARG;
  }
  for (int l = 0; l < n_layers; ++l)
    if (!(layer_cost[l] >= 0.0) || !(layer_cost[l] < 1e280)) {
      set_error("vcy_partition_layers: layer costs must be finite and non-negative");
      return VCY_ERR_INVALID_ARG;
    }
  partition_layers(layer_cost, n_layers, n_slabs, nz, z_bounds);
  return VCY_OK;
}

int vcy_plan_z_slabs(vcy_ctx* c, int n_views, const vcy_view* views, const float* const* sdf_device, int n_slabs,
                     int sample_stride, float brick_cost, int32_t* z_bounds, double* layer_cost, int max_layers,
                     int* n_layers) {
  if (!c) return VCY_ERR_NOT_INITIALIZED;
  if (n_views <= 0 || !views || !sdf_device || !z_bounds) {
    set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  for (int i = 0; i < n_views; ++i) {
    const int rc = check_view_static(&views[i]);
    if (rc != VCY_OK) return rc;
    if (!sdf_device[i]) {
      set_error("null SDF pointer");
      return VCY_ERR_INVALID_ARG;
    }
  }
  VCY_HIP_CHECK(hipSetDevice(c->device));
  return plan_z_slabs(c, n_views, views, sdf_device, n_slabs, sample_stride, brick_cost, z_bounds, layer_cost,
                      max_layers, n_layers);
}

int vcy_last_stream_ms(vcy_ctx* c, float* produce_ms, float* carve_ms, float* wall_ms) {
  if (!c || !produce_ms || !carve_ms || !wall_ms) return VCY_ERR_INVALID_ARG;
  VCY_HIP_CHECK(hipSetDevice(c->device));
  *produce_ms = *carve_ms = 0.0f;
  for (int ci = 0; ci < c->stream_timed_chunks; ++ci) {
    float a = 0.0f, b = 0.0f;
    VCY_HIP_CHECK(hipEventElapsedTime(&a, c->stream_events[(size_t)4 * ci + 0], c->stream_events[(size_t)4 * ci + 1]));
    VCY_HIP_CHECK(hipEventElapsedTime(&b, c->stream_events[(size_t)4 * ci + 2], c->stream_events[(size_t)4 * ci + 3]));
    *produce_ms += a;
    *carve_ms += b;
  }
  *wall_ms = c->stream_wall_ms;
  return VCY_OK;
}

int vcy_selftest(vcy_ctx* c) {
  if (!c) return VCY_ERR_NOT_INITIALIZED;
  VCY_HIP_CHECK(hipSetDevice(c->device));
  return selftest_fused(c->stream);
}

int vcy_sdf_upload(vcy_ctx* c, const float* host, int w, int h, float** dev_out) {
  if (!c || !host || !dev_out || w <= 0 || h <= 0) {
    set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  VCY_HIP_CHECK(hipSetDevice(c->device));
  float* d = nullptr;
  VCY_HIP_CHECK(hipMalloc(&d, sizeof(float) * (size_t)w * h));
  hipError_t e = hipMemcpy(d, host, sizeof(float) * (size_t)w * h, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(d);
    set_error("hipMemcpy H2D failed: %s", hipGetErrorString(e));
    return VCY_ERR_HIP;
  }
  *dev_out = d;
  return VCY_OK;
}

int vcy_device_alloc(vcy_ctx* c, int64_t bytes, void** out) {
  if (!c || !out || bytes <= 0) return VCY_ERR_INVALID_ARG;
  VCY_HIP_CHECK(hipSetDevice(c->device));
  VCY_HIP_CHECK(hipMalloc(out, (size_t)bytes));
  return VCY_OK;
}

int vcy_memcpy_h2d(vcy_ctx* c, void* dst, const void* src, int64_t bytes) {
  if (!c || !dst || !src || bytes < 0) return VCY_ERR_INVALID_ARG;
  VCY_HIP_CHECK(hipSetDevice(c->device));
  VCY_HIP_CHECK(hipStreamSynchronize(c->stream));
  VCY_HIP_CHECK(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyHostToDevice));
  return VCY_OK;
}

int vcy_memcpy_d2h(vcy_ctx* c, void* dst, const void* src, int64_t bytes) {
  if (!c || !dst || !src || bytes < 0) return VCY_ERR_INVALID_ARG;
  VCY_HIP_CHECK(hipSetDevice(c->device));
  VCY_HIP_CHECK(hipStreamSynchronize(c->stream));
  VCY_HIP_CHECK(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
  return VCY_OK;
}

int vcy_reset(vcy_ctx* c) {
  if (!c) return VCY_ERR_NOT_INITIALIZED;
  VCY_HIP_CHECK(hipSetDevice(c->device));
  return fill_state(c);
}

int vcy_device_free(vcy_ctx* c, void* p) {
  if (!c) return VCY_ERR_NOT_INITIALIZED;
  VCY_HIP_CHECK(hipSetDevice(c->device));
  VCY_HIP_CHECK(hipStreamSynchronize(c->stream));
  VCY_HIP_CHECK(hipFree(p));
  return VCY_OK;
}

/* ---- state access ------------------------------------------------------- */

int vcy_download(vcy_ctx* c, float* sdf, int32_t* update_num) {
  if (!c) return VCY_ERR_NOT_INITIALIZED;
  VCY_HIP_CHECK(hipSetDevice(c->device));
  { int rcm = materialize(c); if (rcm != VCY_OK) return rcm; }
  VCY_HIP_CHECK(hipStreamSynchronize(c->stream));
  const int64_t n = c->slab_voxels();
  if (sdf)
    VCY_HIP_CHECK(hipMemcpy(sdf, c->owned_slab_sdf(), sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
  if (update_num) {
    std::vector<uint8_t> raw((size_t)n * c->cnt_bytes);
    VCY_HIP_CHECK(hipMemcpy(raw.data(), c->owned_slab_cnt(), raw.size(), hipMemcpyDeviceToHost));
    if (c->cnt_bytes == 1) {
      for (int64_t i = 0; i < n; ++i) update_num[i] = raw[i];
    } else if (c->cnt_bytes == 2) {
      const uint16_t* r = (const uint16_t*)raw.data();
      for (int64_t i = 0; i < n; ++i) update_num[i] = r[i];
    } else {
      std::memcpy(update_num, raw.data(), raw.size());
    }
  }
  return VCY_OK;
}

int vcy_upload(vcy_ctx* c, const float* sdf, const int32_t* update_num) {
  if (!c) return VCY_ERR_NOT_INITIALIZED;
  VCY_HIP_CHECK(hipSetDevice(c->device));
  { int rcm = materialize(c); if (rcm != VCY_OK) return rcm; }
  VCY_HIP_CHECK(hipStreamSynchronize(c->stream));
  const int64_t n = c->slab_voxels();
  if (sdf)
    VCY_HIP_CHECK(hipMemcpy(c->owned_slab_sdf(), sdf, sizeof(float) * (size_t)n, hipMemcpyHostToDevice));
  if (update_num) {
    const int64_t cap = (int64_t)c->opt.update_option.voxel_max_update_num + 1;
    int64_t mx = 0;
    for (int64_t i = 0; i < n; ++i) {
      const int32_t v = update_num[i];
      if (v < 0 || v > cap) {
        set_error("update_num[%lld]=%d outside [0, voxel_max_update_num+1]", (long long)i, v);
        return VCY_ERR_INVALID_ARG;
      }
      mx = v > mx ? v : mx;
    }
    { const int rcw = ensure_count_width(c, std::max<int64_t>(mx, c->views_carved)); if (rcw != VCY_OK) return rcw; }
    std::vector<uint8_t> raw((size_t)n * c->cnt_bytes);
    for (int64_t i = 0; i < n; ++i) {
      const int32_t v = update_num[i];
      if (c->cnt_bytes == 1) raw[i] = (uint8_t)v;
      else if (c->cnt_bytes == 2) ((uint16_t*)raw.data())[i] = (uint16_t)v;
      else ((int32_t*)raw.data())[i] = v;
    }
    VCY_HIP_CHECK(hipMemcpy(c->owned_slab_cnt(), raw.data(), raw.size(), hipMemcpyHostToDevice));
    c->views_carved = std::max<int64_t>(c->views_carved, mx);
  }
  c->halo_valid = false;
  c->cnt_implied = false;  // arbitrary state from outside
  c->brick_min_valid = false;
  return VCY_OK;
}

int vcy_download_positions(vcy_ctx* c, float* pos) {
  if (!c || !pos) return VCY_ERR_INVALID_ARG;
  VCY_HIP_CHECK(hipSetDevice(c->device));
  std::vector<float> px(c->nx), py(c->ny), pz(c->nz);
  VCY_HIP_CHECK(hipMemcpy(px.data(), c->d_px, sizeof(float) * c->nx, hipMemcpyDeviceToHost));
  VCY_HIP_CHECK(hipMemcpy(py.data(), c->d_py, sizeof(float) * c->ny, hipMemcpyDeviceToHost));
  VCY_HIP_CHECK(hipMemcpy(pz.data(), c->d_pz, sizeof(float) * c->nz, hipMemcpyDeviceToHost));
  int64_t i = 0;
  for (int z = c->z0; z < c->z1; ++z)
    for (int y = 0; y < c->ny; ++y)
      for (int x = 0; x < c->nx; ++x, ++i) {
        pos[3 * i + 0] = px[x];
        pos[3 * i + 1] = py[y];
        pos[3 * i + 2] = pz[z];
      }
  return VCY_OK;
}

/* ---- halo --------------------------------------------------------------- */
// Each rank contributes the LAST two xy-slices of its slab: [sdf slice z1-2][sdf slice
// z1-1][cnt slice z1-2][cnt slice z1-1].  Rank r installs rank r-1's contribution as its
// two halo slices z0-2, z0-1 (cells of layer z0 need slice z0-1; deciding which rank owns
// the marching-cubes vertices on plane z0-1 needs the validity of layer z0-1, i.e. slice
// z0-2 as well).

int64_t vcy_halo_bytes(const vcy_ctx* c) {
  if (!c) return 0;
  // (counters travel at their final width: a pack's size and layout do not depend on how many views a slab has seen)
  return 2 * c->slice * (int64_t)(sizeof(float) + c->cnt_bytes_wire);
}

int vcy_halo_pack(vcy_ctx* c, void* send) {
  if (!c || !send) return VCY_ERR_INVALID_ARG;
  if (c->nz_local() < 2) {
    set_error("a slab needs at least 2 slices to exchange halos");
    return VCY_ERR_INVALID_ARG;
  }
  VCY_HIP_CHECK(hipSetDevice(c->device));
  { int rcm = materialize(c); if (rcm != VCY_OK) return rcm; }
  const int64_t s = c->slice;
  const float* sdf_src = c->owned_slab_sdf() + (int64_t)(c->nz_local() - 2) * s;
  const char* cnt_src = (const char*)c->owned_slab_cnt() + (int64_t)(c->nz_local() - 2) * s * c->cnt_bytes;
  VCY_HIP_CHECK(hipMemcpyAsync(send, sdf_src, 2 * s * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  return convert_counts(c->stream, cnt_src, c->cnt_bytes, (char*)send + 2 * s * sizeof(float), c->cnt_bytes_wire, 2 * s);
}

int vcy_halo_install(vcy_ctx* c, const void* prev_pack) {
  if (!c) return VCY_ERR_INVALID_ARG;
  VCY_HIP_CHECK(hipSetDevice(c->device));
  if (c->halo_lo == 0) {
    c->halo_valid = true;  // first slab: nothing below
    return VCY_OK;
  }
  if (!prev_pack) return VCY_ERR_INVALID_ARG;
  const int64_t s = c->slice;
  const char* src = (const char*)prev_pack;
  VCY_HIP_CHECK(hipMemcpyAsync(c->d_sdf, src, 2 * s * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  { const int rcc = convert_counts(c->stream, src + 2 * s * sizeof(float), c->cnt_bytes_wire, c->d_cnt, c->cnt_bytes, 2 * s);
    if (rcc != VCY_OK) return rcc; }
  c->halo_valid = true;
  return VCY_OK;
}

int vcy_halo_copy_from(vcy_ctx* c, vcy_ctx* below) {
  if (!c) return VCY_ERR_INVALID_ARG;
  if (c->halo_lo == 0) {
    c->halo_valid = true;
    return VCY_OK;
  }
  if (!below || below->z1 != c->z0 || below->nx != c->nx || below->ny != c->ny ||
      below->cnt_bytes_wire != c->cnt_bytes_wire || below->nz_local() < 2) {
    set_error("vcy_halo_copy_from: `below` is not the slab that ends at z_begin (with >= 2 slices)");
    return VCY_ERR_INVALID_ARG;
  }
  VCY_HIP_CHECK(hipSetDevice(below->device));
  { int rcm = materialize(below); if (rcm != VCY_OK) return rcm; }
  VCY_HIP_CHECK(hipStreamSynchronize(below->stream));  // its carve must have finished
  VCY_HIP_CHECK(hipSetDevice(c->device));
  { int rcm = flush_pending(c); if (rcm != VCY_OK) return rcm; }
  const int64_t s = c->slice;
  const float* sdf_src = below->owned_slab_sdf() + (int64_t)(below->nz_local() - 2) * s;
  const char* cnt_src = (const char*)below->owned_slab_cnt() + (int64_t)(below->nz_local() - 2) * s * below->cnt_bytes;
  VCY_HIP_CHECK(hipMemcpyPeerAsync(c->d_sdf, c->device, sdf_src, below->device, 2 * s * sizeof(float), c->stream));
  if (below->cnt_bytes == c->cnt_bytes) {
    VCY_HIP_CHECK(hipMemcpyPeerAsync(c->d_cnt, c->device, cnt_src, below->device, 2 * s * c->cnt_bytes, c->stream));
  } else {
    // Slabs that have not seen the same number of views hold counters of different widths.  Neither array is
    // re-allocated for the exchange (the neighbour's would be, from THIS caller's thread, while its own driver thread may
    // be using it): its two slices travel as they are into a staging buffer of this context and are converted into this
    // slab's width behind the copy -- widening is exact, narrowing saturates, and halo counters are only ever read as
    // `update_num >= 1` (convert_counts_kernel).
    const size_t tmp_need = (size_t)(2 * s) * below->cnt_bytes;
    if (c->halo_tmp_bytes < tmp_need) {
      VCY_HIP_CHECK(hipStreamSynchronize(c->stream));
      (void)hipFree(c->d_halo_tmp);
      c->d_halo_tmp = nullptr;
      c->halo_tmp_bytes = 0;
      VCY_HIP_CHECK(hipMalloc(&c->d_halo_tmp, tmp_need));
      c->halo_tmp_bytes = tmp_need;
    }
    VCY_HIP_CHECK(hipMemcpyPeerAsync(c->d_halo_tmp, c->device, cnt_src, below->device, tmp_need, c->stream));
    const int rcc = convert_counts(c->stream, c->d_halo_tmp, below->cnt_bytes, c->d_cnt, c->cnt_bytes, 2 * s);
    if (rcc != VCY_OK) return rcc;
  }
  c->halo_valid = true;
  return VCY_OK;
}

int vcy_halo_unpack(vcy_ctx* c, const void* gathered, int rank, int world) {
  if (!c || !gathered || rank < 0 || rank >= world) return VCY_ERR_INVALID_ARG;
  VCY_HIP_CHECK(hipSetDevice(c->device));
  if (c->halo_lo == 0) {
    c->halo_valid = true;
    return VCY_OK;
  }
  if (rank == 0) {
    set_error("rank 0 must own z_begin == 0");
    return VCY_ERR_INVALID_ARG;
  }
  const int64_t s = c->slice;
  const char* src = (const char*)gathered + (int64_t)(rank - 1) * vcy_halo_bytes(c);
  VCY_HIP_CHECK(hipMemcpyAsync(c->d_sdf, src, 2 * s * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  { const int rcc = convert_counts(c->stream, src + 2 * s * sizeof(float), c->cnt_bytes_wire, c->d_cnt, c->cnt_bytes, 2 * s);
    if (rcc != VCY_OK) return rcc; }
  c->halo_valid = true;
  return VCY_OK;
}

/* ---- carving entry points ------------------------------------------------ */

static int check_view_static(const vcy_view* v) {
  if (!v || v->width <= 0 || v->height <= 0) {
    set_error("invalid view");
    return VCY_ERR_INVALID_ARG;
  }
  // The reference indexes sdf.at() unchecked (assert only); a ROI outside the image is
  // undefined there and rejected here.
  if (v->roi_min[0] < 0 || v->roi_min[1] < 0 || v->roi_max[0] >= v->width ||
      v->roi_max[1] >= v->height || v->roi_min[0] > v->roi_max[0] || v->roi_min[1] > v->roi_max[1]) {
    set_error("ROI [%d,%d]-[%d,%d] outside the %dx%d SDF image", v->roi_min[0], v->roi_min[1],
              v->roi_max[0], v->roi_max[1], v->width, v->height);
    return VCY_ERR_INVALID_ARG;
  }
  return VCY_OK;
}

static int check_view(const vcy_ctx* c, const vcy_view* v) {
  if (!c) {
    set_error("VoxelCarver::Carve voxel grid has not been initialized");
    return VCY_ERR_NOT_INITIALIZED;
  }
  if (c->deferred_rc != VCY_OK) {  // views queued by earlier calls failed to apply (see vcy_ctx::deferred_rc)
    vcy_ctx* m = const_cast<vcy_ctx*>(c);
    const int rc = m->deferred_rc;
    set_error("an earlier queued view failed: %s", m->deferred_msg.c_str());
    m->deferred_rc = VCY_OK;
    m->deferred_msg.clear();
    return rc;
  }
  return check_view_static(v);
}

}  // extern "C"
namespace vcy {
int check_carve_views(vcy_ctx* c, int n_views, const vcy_view* views) {
  for (int i = 0; i < n_views; ++i) {
    const int rc = check_view(c, &views[i]);
    if (rc != VCY_OK) return rc;
  }
  return VCY_OK;
}
}  // namespace vcy
extern "C" {

int vcy_carve_batch_device(vcy_ctx* c, int n_views, const vcy_view* views,
                           const float* const* sdf_device) {
  if (n_views <= 0 || !views || !sdf_device) {
    set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  for (int i = 0; i < n_views; ++i) {
    int rc = check_view(c, &views[i]);
    if (rc != VCY_OK) return rc;
    if (!sdf_device[i]) {
      set_error("null SDF pointer");
      return VCY_ERR_INVALID_ARG;
    }
  }
  VCY_HIP_CHECK(hipSetDevice(c->device));
  return launch_carve(c, n_views, views, sdf_device);
}

// An idle image buffer of at least `bytes` (from the pool, else newly allocated).
static int acquire_sdf_buffer(vcy_ctx* c, size_t bytes, float** out, size_t* cap) {
  for (size_t i = 0; i < c->sdf_pool.size(); ++i) {
    if (c->sdf_pool[i].second >= bytes) {
      *out = c->sdf_pool[i].first;
      *cap = c->sdf_pool[i].second;
      c->sdf_pool.erase(c->sdf_pool.begin() + (long)i);
      return VCY_OK;
    }
  }
  float* d = nullptr;
  VCY_HIP_CHECK(hipMalloc(&d, bytes));
  *out = d;
  *cap = bytes;
  return VCY_OK;
}

// Whether a view accepted by a per-view entry point may wait for a fused launch.
static bool can_defer(vcy_ctx* c, const vcy_view* view) {
  if (!c->defer || !c->use_fused || !fused_eligible(c, 1, view)) return false;
  return true;
}

#ifndef VCY_STAGE_THREADS
#define VCY_STAGE_THREADS 4   // (8 and 16 measured: the producer side of 32 silhouettes at 1280 x 720 stays at 1.45 - 1.5 ms)
#endif
constexpr int kStageThreads = VCY_STAGE_THREADS;  // host threads that copy silhouettes into page-locked staging and queue their DMAs
constexpr int kMaxPendingViews = 32;  // queued images held at most (3.7 MB each at 1280x720)

// Queues (view, private device image): flushes first if the queue is full or of the other projection
// model (one model per fused launch).
static int enqueue_view(vcy_ctx* c, const vcy_view* view, float* d_img, size_t cap) {
  int rc = VCY_OK;
  if (!c->pending.empty() && (c->pending.front().view.is_ortho != 0) != (view->is_ortho != 0)) rc = flush_pending(c, true);
  if (rc == VCY_OK) {
    c->pending.push_back(vcy_ctx::PendingView{*view, d_img, cap});
    c->halo_valid = false;
    if ((int)c->pending.size() >= kMaxPendingViews) rc = flush_pending(c, true);
  } else {
    c->sdf_pool.emplace_back(d_img, cap);
  }
  return rc;
}

int vcy_carve_device(vcy_ctx* c, const vcy_view* view, const float* sdf_device) {
  int rc = check_view(c, view);
  if (rc != VCY_OK) return rc;
  if (!sdf_device) {
    set_error("null SDF pointer");
    return VCY_ERR_INVALID_ARG;
  }
  if (!can_defer(c, view)) return vcy_carve_batch_device(c, 1, view, &sdf_device);
  // the caller may change or free its image after this returns: keep a copy (stream-ordered)
  VCY_HIP_CHECK(hipSetDevice(c->device));
  const size_t bytes = sizeof(float) * (size_t)view->width * view->height;
  float* d = nullptr;
  size_t cap = 0;
  rc = acquire_sdf_buffer(c, bytes, &d, &cap);
  if (rc != VCY_OK) return rc;
  const hipError_t e = hipMemcpyAsync(d, sdf_device, bytes, hipMemcpyDeviceToDevice, c->stream);
  if (e != hipSuccess) {
    c->sdf_pool.emplace_back(d, cap);
    set_error("SDF copy failed: %s", hipGetErrorString(e));
    return VCY_ERR_HIP;
  }
  return enqueue_view(c, view, d, cap);
}

int vcy_carve(vcy_ctx* c, const vcy_view* view, const float* sdf_host) {
  int rc = check_view(c, view);
  if (rc != VCY_OK) return rc;
  if (!sdf_host) {
    set_error("null SDF pointer");
    return VCY_ERR_INVALID_ARG;
  }
  if (!can_defer(c, view)) {
    float* d = nullptr;
    rc = vcy_sdf_upload(c, sdf_host, view->width, view->height, &d);
    if (rc != VCY_OK) return rc;
    rc = vcy_carve_batch_device(c, 1, view, (const float* const*)&d);
    int rc2 = vcy_device_free(c, d);  // synchronises the stream first
    return rc != VCY_OK ? rc : rc2;
  }
  VCY_HIP_CHECK(hipSetDevice(c->device));
  const size_t bytes = sizeof(float) * (size_t)view->width * view->height;
  float* d = nullptr;
  size_t cap = 0;
  rc = acquire_sdf_buffer(c, bytes, &d, &cap);
  if (rc != VCY_OK) return rc;
  // the previous user of a pooled buffer may be a launch still running: order the copy after it
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipMemcpy(d, sdf_host, bytes, hipMemcpyHostToDevice);  // caller's buffer is free on return
  if (e != hipSuccess) {
    c->sdf_pool.emplace_back(d, cap);
    set_error("hipMemcpy H2D failed: %s", hipGetErrorString(e));
    return VCY_ERR_HIP;
  }
  return enqueue_view(c, view, d, cap);
}

int vcy_carve_silhouette(vcy_ctx* c, const vcy_view* view, const uint8_t* mask, float* sdf_out) {
  int rc = check_view(c, view);
  if (rc != VCY_OK) return rc;
  if (!mask) {
    set_error("null silhouette");
    return VCY_ERR_INVALID_ARG;
  }
  VCY_HIP_CHECK(hipSetDevice(c->device));
  const vcy_update_option& u = c->opt.update_option;
  const size_t npx = (size_t)view->width * view->height;
  // device staging kept in the context: [mask u8][transform scratch]; the SDF goes to a pooled image
  const size_t off_scr = (npx + 255) / 256 * 256;
  const size_t need = off_scr + device_make_sdf_scratch_bytes(view->width, view->height);
  VCY_HIP_CHECK(hipStreamSynchronize(c->stream));  // the staging may still feed the previous call's kernels
  if (c->sil_scratch_bytes < need) {
    if (c->d_sil_scratch) VCY_HIP_CHECK(hipFree(c->d_sil_scratch));
    c->d_sil_scratch = nullptr;
    c->sil_scratch_bytes = 0;
    VCY_HIP_CHECK(hipMalloc(&c->d_sil_scratch, need));
    c->sil_scratch_bytes = need;
  }
  char* d = (char*)c->d_sil_scratch;
  float* d_img = nullptr;
  size_t cap = 0;
  rc = acquire_sdf_buffer(c, npx * sizeof(float), &d_img, &cap);
  if (rc != VCY_OK) return rc;
  auto give_back = [&](int code) {
    (void)hipStreamSynchronize(c->stream);
    c->sdf_pool.emplace_back(d_img, cap);
    return code;
  };
  if (hipMemcpy(d, mask, npx, hipMemcpyHostToDevice) != hipSuccess) {  // caller's mask is free on return
    set_error("mask upload failed");
    return give_back(VCY_ERR_HIP);
  }
  // MakeSignedDistanceField(silhouette, roi_min, roi_max, sdf, option_.sdf_minmax_normalize,
  //   use_truncation, truncation_band), reference voxel_carver.cc:405-408 -- on the device
  rc = device_make_sdf(c->stream, (const uint8_t*)d, view->width, view->height, view->roi_min, view->roi_max,
                       c->opt.sdf_minmax_normalize != 0, u.use_truncation != 0, u.truncation_band, d + off_scr, d_img);
  if (rc != VCY_OK) return give_back(rc);
  if (sdf_out) {
    hipError_t e = hipMemcpyAsync(sdf_out, d_img, npx * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
      set_error("sdf download failed: %s", hipGetErrorString(e));
      return give_back(VCY_ERR_HIP);
    }
  }
  if (can_defer(c, view)) return enqueue_view(c, view, d_img, cap);
  const float* img = d_img;
  rc = vcy_carve_batch_device(c, 1, view, &img);
  return give_back(rc);
}

int vcy_make_sdf_device(vcy_ctx* c, const uint8_t* mask_host, int w, int h, const int32_t rmin[2],
                        const int32_t rmax[2], int normalize, int truncate, float band, float** sdf_device_out) {
  if (!c || !mask_host || !sdf_device_out || w <= 0 || h <= 0 || rmin[0] < 0 || rmin[1] < 0 || rmax[0] >= w ||
      rmax[1] >= h || rmin[0] > rmax[0] || rmin[1] > rmax[1]) {
    set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  VCY_HIP_CHECK(hipSetDevice(c->device));
  const size_t npx = (size_t)w * h;
  char* tmp = nullptr;
  float* sdf = nullptr;
  const size_t off_scr = (npx + 255) / 256 * 256;
  VCY_HIP_CHECK(hipMalloc(&sdf, npx * sizeof(float)));
  if (hipMalloc(&tmp, off_scr + device_make_sdf_scratch_bytes(w, h)) != hipSuccess) {
    (void)hipFree(sdf);
    set_error("out of device memory");
    return VCY_ERR_HIP;
  }
  int rc = VCY_OK;
  if (hipMemcpyAsync(tmp, mask_host, npx, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = VCY_ERR_HIP;
  if (rc == VCY_OK)
    rc = device_make_sdf(c->stream, (const uint8_t*)tmp, w, h, rmin, rmax, normalize != 0, truncate != 0, band,
                         tmp + off_scr, sdf);
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(tmp);
  if (rc != VCY_OK) {
    (void)hipFree(sdf);
    return rc;
  }
  *sdf_device_out = sdf;
  return VCY_OK;
}

// MakeSignedDistanceField for n silhouettes in host memory into CALLER-owned device images (sdf_device_out[i]: w * h
// floats on the context's device): page-locked staging -> DMA -> device transform, in groups of 32.  Returns when the
// images are complete.  What a rank of a multi-GPU job calls for ITS share of the views (views r, r + G, ...) before the
// images are exchanged (vacancy_amd.dist.carve_silhouettes_sharded): every GPU building every SDF would leave the
// streamed path producer-bound at 8 GPUs.
int vcy_make_sdf_batch_device(vcy_ctx* c, int n_views, const vcy_view* views, const uint8_t* const* masks_host,
                              float* const* sdf_device_out) {
  if (!c) return VCY_ERR_NOT_INITIALIZED;
  if (n_views < 0 || (n_views > 0 && (!views || !masks_host || !sdf_device_out))) {
    set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  if (n_views == 0) return VCY_OK;
  size_t max_px = 0;
  for (int i = 0; i < n_views; ++i) {
    const int rc = check_view_static(&views[i]);
    if (rc != VCY_OK) return rc;
    if (!masks_host[i] || !sdf_device_out[i]) {
      set_error("null silhouette or output image");
      return VCY_ERR_INVALID_ARG;
    }
    max_px = std::max(max_px, (size_t)views[i].width * views[i].height);
  }
  VCY_HIP_CHECK(hipSetDevice(c->device));
  const vcy_update_option& u = c->opt.update_option;
  const int group = std::min(32, n_views);
  const size_t sz_mask = (max_px + 255) / 256 * 256;
  const size_t sz_scr = (device_make_sdf_scratch_bytes(1, (int)max_px) + 255) / 256 * 256;
  // staging of the streamed entry point, grown on demand (page-locking 64 MB per call would cost more than the work)
  const size_t need_dev = (size_t)group * (sz_mask + sz_scr), need_pin = (size_t)group * sz_mask;
  if (c->stream_pool_bytes < need_dev) {
    VCY_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->d_stream_pool) VCY_HIP_CHECK(hipFree(c->d_stream_pool));
    c->d_stream_pool = nullptr;
    c->stream_pool_bytes = 0;
    VCY_HIP_CHECK(hipMalloc(&c->d_stream_pool, need_dev));
    c->stream_pool_bytes = need_dev;
  }
  if (c->pinned_bytes < need_pin) {
    if (c->aux_stream) VCY_HIP_CHECK(hipStreamSynchronize(c->aux_stream));
    if (c->h_pinned) VCY_HIP_CHECK(hipHostFree(c->h_pinned));
    c->h_pinned = nullptr;
    c->pinned_bytes = 0;
    VCY_HIP_CHECK(hipHostMalloc(&c->h_pinned, need_pin, hipHostMallocDefault));
    c->pinned_bytes = need_pin;
  }
  char* d_tmp = (char*)c->d_stream_pool;
  void* h_stage = c->h_pinned;
  int rc = VCY_OK;
  hipStream_t st = c->stream;
  for (int first = 0; first < n_views && rc == VCY_OK; first += group) {
    const int m = std::min(group, n_views - first);
    std::vector<const uint8_t*> mptr((size_t)m);
    std::vector<float*> optr((size_t)m);
    if (first > 0 && hipStreamSynchronize(st) != hipSuccess) rc = VCY_ERR_HIP;  // staging and scratch are reused
    for (int j = 0; j < m && rc == VCY_OK; ++j) {
      const size_t npx = (size_t)views[first + j].width * views[first + j].height;
      std::memcpy((char*)h_stage + (size_t)j * sz_mask, masks_host[first + j], npx);
      if (hipMemcpyAsync(d_tmp + (size_t)j * sz_mask, (char*)h_stage + (size_t)j * sz_mask, npx, hipMemcpyHostToDevice, st) != hipSuccess)
        rc = VCY_ERR_HIP;
      mptr[(size_t)j] = (const uint8_t*)(d_tmp + (size_t)j * sz_mask);
      optr[(size_t)j] = sdf_device_out[first + j];
    }
    if (rc == VCY_OK)
      rc = device_make_sdf_batch(st, m, mptr.data(), views + first, c->opt.sdf_minmax_normalize != 0, u.use_truncation != 0,
                                 u.truncation_band, d_tmp + (size_t)group * sz_mask, sz_scr, optr.data());
    else
      set_error("vcy_make_sdf_batch_device: mask upload failed");
  }
  if (hipStreamSynchronize(st) != hipSuccess && rc == VCY_OK) {
    set_error("vcy_make_sdf_batch_device: %s", hipGetErrorString(hipGetLastError()));
    rc = VCY_ERR_HIP;
  }
  return rc;
}

// Streams n silhouettes through the device: masks are uploaded and turned into SDFs on a second
// stream in chunks of 32 views while the previous chunk is being fused into the grid on the
// context's stream (two sets of SDF buffers, ordered with events; BASELINE config 5).
int vcy_carve_batch_silhouettes(vcy_ctx* c, int n_views, const vcy_view* views,
                                const uint8_t* const* masks_host) {
  if (!c) {
    set_error("VoxelCarver::Carve voxel grid has not been initialized");
    return VCY_ERR_NOT_INITIALIZED;
  }
  if (n_views <= 0 || !views || !masks_host) {
    set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  size_t max_px = 0;
  for (int i = 0; i < n_views; ++i) {
    int rc = check_view(c, &views[i]);
    if (rc != VCY_OK) return rc;
    if (!masks_host[i]) {
      set_error("null silhouette");
      return VCY_ERR_INVALID_ARG;
    }
    max_px = std::max(max_px, (size_t)views[i].width * views[i].height);
  }
  VCY_HIP_CHECK(hipSetDevice(c->device));
  const vcy_update_option& u = c->opt.update_option;
  const int chunk = 32;  // per fused launch here: the next 32 silhouettes upload and transform meanwhile
  const int per_set = std::min(chunk, n_views);
  const size_t px_al = (max_px + 255) / 256 * 256;
  // [2 sets][per_set] SDF images + [2 sets][per_set] masks + [per_set] transform scratch; cached in
  // the context and grown on demand
  const size_t sz_sdf = px_al * sizeof(float), sz_mask = px_al;
  const size_t sz_scr = (device_make_sdf_scratch_bytes(1, (int)max_px) + 255) / 256 * 256;
  const size_t total = 2 * per_set * (sz_sdf + sz_mask) + per_set * sz_scr + 256;
  int rc = VCY_OK;
  auto fail_hip = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && rc == VCY_OK) {
      set_error("%s failed: %s", what, hipGetErrorString(e));
      rc = VCY_ERR_HIP;
    }
    return e != hipSuccess;
  };
  if (c->stream_pool_bytes < total) {
    VCY_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->d_stream_pool) VCY_HIP_CHECK(hipFree(c->d_stream_pool));
    c->d_stream_pool = nullptr;
    c->stream_pool_bytes = 0;
    VCY_HIP_CHECK(hipMalloc(&c->d_stream_pool, total));
    c->stream_pool_bytes = total;
  }
  // page-locked staging: pageable memory would be copied through the runtime's own bounce buffer by
  // one thread; here a few host threads fill it and the DMA engine takes it from there
  const size_t pinned_total = 2 * (size_t)per_set * sz_mask;
  if (c->pinned_bytes < pinned_total) {
    if (c->aux_stream) VCY_HIP_CHECK(hipStreamSynchronize(c->aux_stream));
    if (c->h_pinned) VCY_HIP_CHECK(hipHostFree(c->h_pinned));
    c->h_pinned = nullptr;
    c->pinned_bytes = 0;
    VCY_HIP_CHECK(hipHostMalloc(&c->h_pinned, pinned_total, hipHostMallocDefault));
    c->pinned_bytes = pinned_total;
  }
  if (!c->aux_stream) {
    fail_hip(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking), "hipStreamCreate");
    for (int k = 0; k < 2 && rc == VCY_OK; ++k) {
      fail_hip(hipEventCreateWithFlags(&c->ev_ready[k], hipEventDisableTiming), "hipEventCreate");
      fail_hip(hipEventCreateWithFlags(&c->ev_consumed[k], hipEventDisableTiming), "hipEventCreate");
      fail_hip(hipEventCreateWithFlags(&c->ev_uploaded[k], hipEventDisableTiming), "hipEventCreate");
    }
    if (rc != VCY_OK) return rc;
  }
  hipStream_t aux = c->aux_stream;
  // timing of the two sides (vcy_last_stream_ms): per chunk, events around its production (staging copy, H2D, SDF
  // build; on the producer stream) and around its carve (on the context's stream)
  const auto t_entry = std::chrono::steady_clock::now();
  const int n_chunks_t = (n_views + chunk - 1) / chunk;
  while ((int)c->stream_events.size() < 4 * n_chunks_t) {
    hipEvent_t ev = nullptr;
    if (fail_hip(hipEventCreate(&ev), "hipEventCreate")) return rc;
    c->stream_events.push_back(ev);
  }
  c->stream_timed_chunks = 0;
  char* pool = (char*)c->d_stream_pool;
  char* scratch = pool + 2 * per_set * (sz_sdf + sz_mask);
  auto sdf_buf = [&](int set, int j) { return (float*)(pool + ((size_t)set * per_set + j) * sz_sdf); };
  auto mask_buf = [&](int set, int j) {
    return (uint8_t*)(pool + 2 * per_set * sz_sdf + ((size_t)set * per_set + j) * sz_mask);
  };
  auto stage_buf = [&](int set, int j) { return (uint8_t*)c->h_pinned + ((size_t)set * per_set + j) * sz_mask; };
  const int n_chunks = (n_views + chunk - 1) / chunk;
  // producer for chunk ci: upload + SDF on the aux stream
  auto produce = [&](int ci) {
    const int set = ci & 1, first = ci * chunk, m = std::min(chunk, n_views - first);
    if (ci >= 2) {
      fail_hip(hipStreamWaitEvent(aux, c->ev_consumed[set], 0), "hipStreamWaitEvent");  // device buffers free
      fail_hip(hipEventSynchronize(c->ev_uploaded[set]), "hipEventSynchronize");        // staging free
    }
    std::vector<const uint8_t*> mptr(m);
    std::vector<float*> optr(m);
    for (int j = 0; j < m; ++j) {
      mptr[j] = mask_buf(set, j);
      optr[j] = sdf_buf(set, j);
    }
    fail_hip(hipEventRecord(c->stream_events[(size_t)4 * ci + 0], aux), "hipEventRecord");
    // host threads: copy silhouette j into the staging buffer, then queue its DMA
    const int n_thr = std::max(1, std::min(m, std::min(kStageThreads, (int)std::thread::hardware_concurrency())));
    std::vector<hipError_t> terr((size_t)n_thr, hipSuccess);
    auto worker = [&](int t) {
      (void)hipSetDevice(c->device);
      for (int j = t; j < m; j += n_thr) {
        const vcy_view& v = views[first + j];
        const size_t npx = (size_t)v.width * v.height;
        std::memcpy(stage_buf(set, j), masks_host[first + j], npx);
        const hipError_t e = hipMemcpyAsync(mask_buf(set, j), stage_buf(set, j), npx, hipMemcpyHostToDevice, aux);
        if (e != hipSuccess) terr[(size_t)t] = e;
      }
    };
    if (rc == VCY_OK) {
      std::vector<std::thread> pool_thr;
      for (int t = 1; t < n_thr; ++t) pool_thr.emplace_back(worker, t);
      worker(0);
      for (auto& th : pool_thr) th.join();
      for (int t = 0; t < n_thr; ++t) fail_hip(terr[(size_t)t], "mask upload");
    }
    fail_hip(hipEventRecord(c->ev_uploaded[set], aux), "hipEventRecord");
    if (rc == VCY_OK) {
      // MakeSignedDistanceField(...) for the whole chunk at once, reference voxel_carver.cc:405-408
      int r2 = device_make_sdf_batch(aux, m, mptr.data(), views + first, c->opt.sdf_minmax_normalize != 0,
                                     u.use_truncation != 0, u.truncation_band, scratch, sz_scr, optr.data());
      if (r2 != VCY_OK) rc = r2;
    }
    fail_hip(hipEventRecord(c->stream_events[(size_t)4 * ci + 1], aux), "hipEventRecord");
    fail_hip(hipEventRecord(c->ev_ready[set], aux), "hipEventRecord");
  };
  if (rc == VCY_OK) produce(0);
  for (int ci = 0; ci < n_chunks && rc == VCY_OK; ++ci) {
    const int set = ci & 1, first = ci * chunk, m = std::min(chunk, n_views - first);
    if (ci + 1 < n_chunks) produce(ci + 1);  // next chunk's SDFs build while this chunk carves
    if (rc != VCY_OK) break;
    fail_hip(hipStreamWaitEvent(c->stream, c->ev_ready[set], 0), "hipStreamWaitEvent");
    std::vector<const float*> ptrs(m);
    for (int j = 0; j < m; ++j) ptrs[j] = sdf_buf(set, j);
    fail_hip(hipEventRecord(c->stream_events[(size_t)4 * ci + 2], c->stream), "hipEventRecord");
    int r2 = launch_carve(c, m, views + first, ptrs.data());
    if (r2 != VCY_OK) rc = r2;
    fail_hip(hipEventRecord(c->stream_events[(size_t)4 * ci + 3], c->stream), "hipEventRecord");
    fail_hip(hipEventRecord(c->ev_consumed[set], c->stream), "hipEventRecord");
    if (rc == VCY_OK) c->stream_timed_chunks = ci + 1;
  }
  (void)hipStreamSynchronize(c->stream);
  (void)hipStreamSynchronize(aux);
  c->stream_wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_entry).count();
  return rc;
}

// Voxel state at arbitrary voxel ids (global ids of this slab), gathered on the device.
__global__ void gather_state_kernel(const float* __restrict__ sdf, const void* __restrict__ cnt, int cnt_bytes,
                                    const long long* __restrict__ ids, int64_t n, long long first_id,
                                    float* __restrict__ out_sdf, int* __restrict__ out_cnt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long local = ids[i] - first_id;
  out_sdf[i] = sdf[local];
  out_cnt[i] = cnt_bytes == 1 ? (int)((const uint8_t*)cnt)[local]
             : cnt_bytes == 2 ? (int)((const uint16_t*)cnt)[local] : ((const int*)cnt)[local];
}

int vcy_download_voxels(vcy_ctx* c, int64_t n, const int64_t* voxel_ids, float* sdf, int32_t* update_num) {
  if (!c || n < 0 || (n > 0 && (!voxel_ids || !sdf || !update_num))) {
    set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  if (n == 0) return VCY_OK;
  const int64_t first = (int64_t)c->z0 * c->slice, last = (int64_t)c->z1 * c->slice;
  for (int64_t i = 0; i < n; ++i)
    if (voxel_ids[i] < first || voxel_ids[i] >= last) {
      set_error("voxel id %lld outside this slab", (long long)voxel_ids[i]);
      return VCY_ERR_INVALID_ARG;
    }
  VCY_HIP_CHECK(hipSetDevice(c->device));
  { int rcm = materialize(c); if (rcm != VCY_OK) return rcm; }
  char* d = nullptr;
  VCY_HIP_CHECK(hipMalloc(&d, (size_t)n * 16));
  long long* d_ids = (long long*)d;
  float* d_s = (float*)(d + (size_t)n * 8);
  int* d_n = (int*)(d + (size_t)n * 12);
  hipError_t e = hipMemcpyAsync(d_ids, voxel_ids, (size_t)n * 8, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(gather_state_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                       c->owned_slab_sdf(), c->owned_slab_cnt(), c->cnt_bytes, d_ids, n, (long long)first, d_s, d_n);
    e = hipMemcpyAsync(sdf, d_s, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(update_num, d_n, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  if (e != hipSuccess) {
    set_error("vcy_download_voxels: %s", hipGetErrorString(e));
    return VCY_ERR_HIP;
  }
  return VCY_OK;
}

// Counts voxels whose state differs between two slabs (bit compare of sdf, value compare of update_num).
__device__ __forceinline__ int load_count(const void* cnt, int cnt_bytes, int64_t i) {
  return cnt_bytes == 1 ? (int)((const uint8_t*)cnt)[i]
       : cnt_bytes == 2 ? (int)((const uint16_t*)cnt)[i] : ((const int*)cnt)[i];
}

__global__ __launch_bounds__(256) void state_diff_kernel(const float* __restrict__ sa, const void* __restrict__ ca,
                                                         int cba, const float* __restrict__ sb,
                                                         const void* __restrict__ cb, int cbb, int64_t n,
                                                         unsigned long long* __restrict__ n_diff) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  unsigned local = 0;
  for (; i < n; i += stride) {
    const bool diff = __float_as_uint(sa[i]) != __float_as_uint(sb[i]) || load_count(ca, cba, i) != load_count(cb, cbb, i);
    local += diff ? 1u : 0u;
  }
  const unsigned long long m = __ballot(local != 0);
  if (m) {  // rare: serialise only when something differs
    for (int d = 32; d > 0; d >>= 1) local += __shfl_down(local, d, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(n_diff, (unsigned long long)local);
  }
}

int vcy_state_equal(vcy_ctx* a, vcy_ctx* b, int64_t* n_diff) {
  if (!a || !b || !n_diff) return VCY_ERR_INVALID_ARG;
  if (a->device != b->device || a->nx != b->nx || a->ny != b->ny || a->z0 != b->z0 || a->z1 != b->z1) {
    set_error("vcy_state_equal: the contexts do not own the same slab on the same device");
    return VCY_ERR_INVALID_ARG;
  }
  VCY_HIP_CHECK(hipSetDevice(a->device));
  { int rcm = materialize(a); if (rcm != VCY_OK) return rcm; }
  { int rcm = materialize(b); if (rcm != VCY_OK) return rcm; }
  VCY_HIP_CHECK(hipStreamSynchronize(b->stream));
  unsigned long long* d = nullptr;
  unsigned long long h = 0;
  VCY_HIP_CHECK(hipMalloc(&d, sizeof(unsigned long long)));
  hipError_t e = hipMemsetAsync(d, 0, sizeof(unsigned long long), a->stream);
  if (e == hipSuccess) {
    const int64_t n = a->slab_voxels();
    const unsigned grid = (unsigned)std::min<int64_t>((n + 255) / 256, 256 * 64);
    hipLaunchKernelGGL(state_diff_kernel, dim3(grid), dim3(256), 0, a->stream, a->owned_slab_sdf(), a->owned_slab_cnt(),
                       a->cnt_bytes, b->owned_slab_sdf(), b->owned_slab_cnt(), b->cnt_bytes, n, d);
    e = hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, a->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(a->stream);
  (void)hipFree(d);
  if (e != hipSuccess) {
    set_error("vcy_state_equal: %s", hipGetErrorString(e));
    return VCY_ERR_HIP;
  }
  *n_diff = (int64_t)h;
  return VCY_OK;
}

int vcy_distance_transform_l1(const uint8_t* mask, int w, int h, const int32_t rmin[2],
                              const int32_t rmax[2], float* out) {
  if (!mask || !out || w <= 0 || h <= 0 || rmin[0] < 0 || rmin[1] < 0 || rmax[0] >= w || rmax[1] >= h) {
    set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  host_distance_transform_l1(mask, w, h, rmin, rmax, out);
  return VCY_OK;
}

int vcy_make_sdf(const uint8_t* mask, int w, int h, const int32_t rmin[2], const int32_t rmax[2],
                 int normalize, int truncate, float band, float* out) {
  if (!mask || !out || w <= 0 || h <= 0 || rmin[0] < 0 || rmin[1] < 0 || rmax[0] >= w || rmax[1] >= h) {
    set_error("invalid argument");
    return VCY_ERR_INVALID_ARG;
  }
  host_make_sdf(mask, w, h, rmin, rmax, normalize != 0, truncate != 0, band, out);
  return VCY_OK;
}

int vcy_extract_iso(vcy_ctx* c, double iso, int linear_interp, vcy_mesh* out) {
  if (!c) {
    set_error("voxel grid has not been initialized");
    return VCY_ERR_NOT_INITIALIZED;
  }
  if (!out) return VCY_ERR_INVALID_ARG;
  std::memset(out, 0, sizeof(*out));
  VCY_HIP_CHECK(hipSetDevice(c->device));
  const auto t0 = std::chrono::steady_clock::now();
  { int rcm = materialize(c); if (rcm != VCY_OK) return rcm; }
  const int rc = extract_iso(c, iso, linear_interp, out);
  if (rc != VCY_OK) vcy_mesh_free(out);  // (whatever host arrays a failed extraction had already taken from the pool)
  c->last_extract_wall_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}

int vcy_last_extract_wall_ms(const vcy_ctx* c, float* wall_ms) {
  if (!c || !wall_ms) return VCY_ERR_INVALID_ARG;
  *wall_ms = c->last_extract_wall_ms;
  return VCY_OK;
}

int vcy_last_extract_ms(const vcy_ctx* c, float* device_ms) {
  if (!c || !device_ms) return VCY_ERR_INVALID_ARG;
  *device_ms = c->last_extract_device_ms;
  return VCY_OK;
}

void vcy_mesh_free(vcy_mesh* m) {
  if (!m) return;
  mesh_host_free(m->vertices);
  mesh_host_free(m->faces);
  mesh_host_free(m->edge_keys);
  std::memset(m, 0, sizeof(*m));
}

}  // extern "C"
