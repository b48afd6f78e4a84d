// Internal declarations shared by the HIP translation units of libvacancy_hip.so.
// gfx950 (MI355X) only; built with -ffp-contract=off (bit-exact parity with the
// reference's non-FMA x86 arithmetic).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "vacancy_hip.h"

namespace vcy {

void set_error(const char* fmt, ...);

#define VCY_HIP_CHECK(expr)                                                        \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      ::vcy::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                       __FILE__, __LINE__);                                        \
      return VCY_ERR_HIP;                                                          \
    }                                                                              \
  } while (0)

// std::numeric_limits<float>::lowest(), InvalidSdf::kVal (reference voxel_carver.cc:100)
constexpr float kInvalidSdf = -3.402823466e+38f;

}  // namespace vcy

// The device-resident voxel grid of one z-slab.
//
// Layout (structure of arrays, the reference's 40-byte AoS Voxel is never materialised):
//   sdf   float  [halo_lo + nz_local][ny][nx]   x fastest, one contiguous slab in HBM
//   cnt   u8/u16/u32 same shape                 Voxel::update_num
//   px/py/pz     float[nx]/[ny]/[nz]            Voxel::pos per axis (global index)
// halo_lo = 2 slices below the slab when z_begin > 0 (filled by vcy_halo_unpack before
// extraction), 0 otherwise.  Voxel::index/id are implicit, Voxel::outside is dead in the
// reference, Voxel::on_surface belongs to ExtractVoxel (host).
struct vcy_ctx {
  int device = 0;
  bool counted = false;               // registered in the process-wide context count (vcy_create succeeded)
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;

  vcy_carver_option opt{};
  int nx = 0, ny = 0, nz = 0;  // global dims
  int z0 = 0, z1 = 0;          // owned slab [z0, z1)
  int halo_lo = 0;             // slices stored below z0
  bool halo_valid = false;
  // Lazy initialisation: after vcy_create / vcy_reset the slab is KNOWN to be sdf = lowest(),
  // update_num = 0 without having been written.  The fused carve starts from that knowledge
  // (no state read, no fill pass); everything else calls materialize() first.
  bool fresh = false;
  int64_t slice = 0;           // nx*ny
  // Counter width, lazily widened (round 5): update_num of a voxel cannot exceed the number of views applied since the
  // fill, so the array is u8 while at most 255 views have been applied (or uploaded counts are <= 255), u16 up to 65535,
  // u32 beyond -- whatever voxel_max_update_num would allow in the end (cnt_bytes_wire, also the width of the halo
  // packs, whose layout therefore does not depend on how far a slab has got).  The default options (max 255 -> counts up
  // to 256) used to mean 6 B/voxel from the start; now 5 B/voxel until the 256th view.  ensure_count_width() widens
  // (one conversion pass, skipped on a fresh slab); vcy_reset goes back to u8.  vcy_set_param "lazycount" 0 keeps the
  // final width from the start.
  int cnt_bytes = 1;
  int cnt_bytes_wire = 1;
  bool lazy_count = true;

  float* d_sdf = nullptr;      // includes halo slices
  void* d_cnt = nullptr;
  void* d_cnt_spare = nullptr;        // the counter array of the OTHER width, kept once it exists (set_count_width): a
  size_t cnt_spare_cap = 0;           // reset / carve cycle across the 256th view then allocates and frees nothing
  size_t cnt_cap = 0;                 // bytes allocated behind d_cnt
  void* d_xv_scratch = nullptr;       // ExtractVoxel: keep bits, block counts, scan scratch (grow-only)
  size_t xv_scratch_bytes = 0;
  void* d_xv_ids = nullptr;           // ExtractVoxel: the kept voxel ids before their copy to the host (grow-only)
  size_t xv_ids_bytes = 0;
  void* d_halo_tmp = nullptr;         // two slices of a neighbour's counters at ITS width (vcy_halo_copy_from)
  size_t halo_tmp_bytes = 0;
  float* d_px = nullptr;
  float* d_py = nullptr;
  float* d_pz = nullptr;

  bool mesh_keys = true;              // vcy_extract_iso also returns the edge key of every vertex (vcy_set_param "meshkeys")
  int list_records = 1;               // "listrecords": the live list of a one-view launch carries the footprint records and per-wave live bits (0: workgroup ids only)
  int eager_state = -1;               // "eagerstate": few-view launches request a brick's state next to its footprint record instead of behind the early-return test (-1: when most workgroups were live last time, 0 never, 1 always)
  int one_view = 1;                   // "oneview": single-view launches take the kernel instance compiled for one view (carve_fused_kernel NB == 0)
  int nt_store = -1;                  // "ntstore": streaming stores in the cooperative write-back (-1 / 1: whenever it runs -- 0.5 - 1.5 % on single-view launches; 0 never)
  int row_kernel = 0;                 // "rowkernel": launches of up to this many views take the few-view flavour of the fused kernel (a wave walks the bricks of a row segment); -1 = up to 8, 0 = never (the default: measured slower, carve_fused.hip kRowBricks)
  int mc_skip = 1;                    // marching cubes: bricks whose kept minimum is above the iso level are not read (vcy_set_param "mcskip": 0 never, 1 where it pays -- rows of 1024 voxels and more --, 2 wherever possible)
  bool mc_sweep = false;              // marching cubes: cell search in one sweep with the bit planes in LDS where the row shape allows (vcy_set_param "mcsweep")
  int tile_mode = 0;                  // 0 auto, 1 the 16 x 16 pixel tile, 2 the 2048-pixel tile filled in place (vcy_set_param "tile")
  bool use_cull = true;               // vcy_set_param("cull", 0): never drop provably idle views
  // Views accepted by the per-view entry points (vcy_carve, vcy_carve_device, vcy_carve_silhouette) but
  // not applied yet: they are carved together, in order, by ONE fused launch when the state is next
  // needed (extraction, download, halo, timer, sync ...) or when 32 are waiting -- the reference's
  // `for each view: Carve()` loop then costs one pass over the grid instead of one per view.
  struct PendingView {
    vcy_view view;
    float* d_sdf;   // private device copy of the SDF image
    size_t bytes;
  };
  std::vector<PendingView> pending;
  std::vector<std::pair<float*, size_t>> sdf_pool;  // idle image buffers
  void* d_sil_scratch = nullptr;      // staging of vcy_carve_silhouette (mask + transform scratch)
  size_t sil_scratch_bytes = 0;
  bool defer = true;                  // vcy_set_param("defer", 0): apply every view at once
  // A queued view that failed to apply inside a call that cannot report it to a Carve() caller (an
  // extraction, a download ...) is remembered: the NEXT carve entry point returns it (then clears it).
  int deferred_rc = 0;
  int inject_fail = 0;                // test hook: the next n applications of views fail (vcy_set_param "inject_carve_failure")
  std::string deferred_msg;
  int last_div_level = 0;             // division variant of the last fused launch (vcy_get_param "div_level")
  bool use_short_div = true;          // vcy_set_param("shortdiv", 0): always the full division sequence
  bool use_fused = true;              // vcy_set_option("fused", 0) forces the per-view kernel
  float* h_px = nullptr;              // host copy of the x axis table (c0 tables of the fused carve)
  float h_px_min = 0, h_px_max = 0, h_py_min = 0, h_py_max = 0;  // extreme voxel centres
  float* h_pz = nullptr;              // host copy of d_pz (per-view z tables of the fused carve)
  void* d_fused_scratch = nullptr;    // view blocks + z tables of the fused carve kernel
  size_t fused_scratch_bytes = 0;
  bool cnt_implied = true;            // update_num == 0 implies sdf == lowest(): no vcy_upload since the fill
  float* d_brick_min = nullptr;       // min(sdf) of every 8x8x8 wave brick of the slab [bz][by][bxw], kept by the fused carve
  bool brick_min_valid = false;       // ... and current: no write to the state since has bypassed the fused kernel
  int* d_wg_list = nullptr;           // live workgroups of a carve launch of few views ([0] = count), live_workgroups_kernel
  size_t wg_list_bytes = 0;
  bool time_carve = false;            // vcy_set_param("carvetimer", 1): events around pre-pass and carve kernel of every fused launch
  struct CarveStamp {                 // one chunk of one fused launch
    hipEvent_t ev[3];                 // before window maxima / pre-pass, before the carve kernel, after it
    bool first_chunk;
  };
  std::vector<CarveStamp> carve_log;  // event triplets, created on demand (vcy_carve_log, vcy_last_carve_ms)
  int carve_log_n = 0;                // triplets recorded since the log was last cleared
  int carve_log_last = 0;             // index of the first chunk of the last launch
  int carve_log_dropped = 0;          // chunks that found the log full since it was cleared (vcy_get_param "carvelog_dropped")
  bool carve_log_last_dropped = false;  // ... the last launch among them: vcy_last_carve_ms has nothing to report
  int* h_live_hint = nullptr;         // page-locked {live workgroups, workgroups} of the last listed launch (a hint, see launch_carve_fused)
  int64_t live_list_age = 0;
  bool use_live_list = true;          // vcy_set_param("livelist", 0): every workgroup is launched and decides for itself
  bool live_sync = true;              // vcy_set_param("livesync", 0): the carve kernel of a listed launch starts every workgroup instead of waiting for the list's length
  int coop_store = -1;                // vcy_set_param("coopstore"): write-back of a workgroup's bricks through LDS in whole row segments; -1 = where it pays (launch_carve_fused), 0 / 1 = never / whenever possible
  bool count_pairs = false;           // vcy_set_param("paircount", 1): the fused kernel counts the (brick, view) pairs it processes
  unsigned long long* d_pair_count = nullptr;  // ... per brick layer of the slab, of the last fused launch (vcy_last_carve_pairs)
  int pair_count_layers = 0, pair_count_views = 0;
  int64_t record_bytes_max = 0;       // vcy_set_param("recordbytes", n): footprint records per launch chunk (0: 2 GiB)
  int prologue_mode = 0;              // vcy_set_param("prologue"): 0 / 2 footprint records from the pre-pass (chunked); 1 footprints in the carve kernel's prologue (slower at every shape measured)
  void* d_records = nullptr;          // footprint records of one fused launch, 8 bytes per (wave brick, view)
  size_t records_bytes = 0;
  float* d_wmax = nullptr;            // window-maximum planes of the views of one fused launch
  size_t wmax_bytes = 0;
  bool fused_ortho = false;           // projection model of the launch being prepared
  void* h_fused_stage[2] = {nullptr, nullptr};  // page-locked staging of the view blocks, taken in turn (prepare_views)
  hipEvent_t ev_fused_stage[2] = {nullptr, nullptr};  // "the copy out of staging buffer q has been made"
  size_t fused_stage_bytes = 0;
  int fused_stage_idx = 0;
  bool fused_cache_valid = false;     // d_fused_scratch holds what these view parameters give (launch_carve_fused)
  std::vector<char> fused_cache_vp;   // the ViewParams of the launch that filled it
  const float* fused_cache_wmax = nullptr;
  void* fused_cache_at = nullptr;
  bool fused_cache_bound = false, fused_cache_lower = false, fused_cache_ortho = false, fused_cache_samef = true;
  int fused_cache_max_quads = 0;
  int fused_cache_z[2] = {0, 0};
  void* d_stream_pool = nullptr;      // staging of vcy_carve_batch_silhouettes (masks, SDFs, scratch)
  size_t stream_pool_bytes = 0;
  hipStream_t aux_stream = nullptr;   // producer stream of the streamed batch (uploads + SDF build)
  hipEvent_t ev_ready[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr}, ev_uploaded[2] = {nullptr, nullptr};
  std::vector<hipEvent_t> stream_events;  // four per chunk of the last streamed batch (vcy_last_stream_ms)
  int stream_timed_chunks = 0;
  float stream_wall_ms = 0.0f;
  void* h_pinned = nullptr;           // page-locked staging of the silhouettes, two sets
  size_t pinned_bytes = 0;
  void* d_mc_tables = nullptr;        // marching-cubes case tables (mc_kernels.hip)
  void* d_mc_scratch = nullptr;       // bit planes, active words, offsets, per-cell info
  size_t mc_scratch_bytes = 0;
  void* d_mc_flags = nullptr;         // publication flags of the chained scans (mc_kernels.hip, scan_chained_kernel)
  uint32_t mc_scan_epoch = 0;         // ... and the epoch of the last scan (flags never hold a later one)
  void* h_mc_report = nullptr;        // 64 page-locked bytes mc_emit reports an extraction's counts in (extract_iso)
  int64_t mc_direct_bytes = (int64_t)32 << 20;  // "mcdirect": meshes guessed up to this size are written by mc_emit straight into host memory
  int mc_timing = 0;                  // "mctiming" 1 (or VCY_MC_TIMING=1): host-side phases of every extraction on stderr
  uint32_t mc_scan_tickets[2] = {0, 0};  // chunk tickets drawn so far from the two scan slots' counters (scan_chained_kernel)
  void* d_mc_cells = nullptr;         // per-active-cell arrays of the extraction
  size_t mc_cells_bytes = 0;
  void* d_mc_out = nullptr;           // device staging of the extracted mesh
  size_t mc_out_bytes = 0;
  int64_t mc_hint_cells = 0, mc_hint_verts = 0, mc_hint_faces = 0;  // sizes of the last extraction (extract_iso)
  float last_extract_device_ms = 0.0f;
  float last_extract_wall_ms = 0.0f;  // call entry -> mesh arrays in host memory
  hipEvent_t ev_mc_begin = nullptr, ev_mc_end = nullptr;  // the extraction's own timer

  // upper bound on any voxel's update_num (each carved view adds at most one)
  int64_t views_carved = 0;

  float* owned_slab_sdf() const { return d_sdf + (int64_t)halo_lo * slice; }
  void* owned_slab_cnt() const { return (char*)d_cnt + (int64_t)halo_lo * slice * cnt_bytes; }
  int nz_local() const { return z1 - z0; }
  int64_t slab_voxels() const { return slice * (int64_t)(z1 - z0); }
};

namespace vcy {

// carve_kernels.hip
int launch_carve(vcy_ctx* ctx, int n_views, const vcy_view* views, const float* const* sdf_dev);
// carve_fused.hip
struct GridParams;
struct ViewParams;
bool fused_eligible(const vcy_ctx* ctx, int n_views, const vcy_view* views);
int launch_carve_fused(vcy_ctx* ctx, const GridParams& g, int n_views, const ViewParams* vp);
int fused_max_views();
int plan_layer_pairs(vcy_ctx* ctx, int n_views, const ViewParams* vp, int stride, std::vector<double>* pairs,
                     int64_t* bricks_per_layer);  // the slab planner's estimate (carve_fused.hip)
void partition_layers(const double* cost, int n_layers, int n_slabs, int nz, int32_t* z_bounds);  // carve_kernels.hip
int plan_z_slabs(vcy_ctx* ctx, int n_views, const vcy_view* views, const float* const* sdf_dev, int n_slabs, int stride,
                 float brick_cost, int32_t* z_bounds, double* layer_cost, int max_layers, int* n_layers);  // carve_kernels.hip
int selftest_fused(hipStream_t stream);
int flush_pending(vcy_ctx* ctx, bool from_carve = false);   // applies vcy_ctx::pending (no-op when empty)
int check_carve_views(vcy_ctx* ctx, int n_views, const vcy_view* views);  // argument checks of the carve entry points (vcy_api.hip)
int carve_log_open(vcy_ctx* ctx, bool first_chunk);          // next slot of vcy_ctx::carve_log, or -1 (vcy_api.hip)
// mc_kernels.hip
int extract_iso(vcy_ctx* ctx, double iso, int linear_interp, vcy_mesh* out);
// sdf2d.hip
void host_distance_transform_l1(const uint8_t* mask, int w, int h, const int32_t* rmin,
                                const int32_t* rmax, float* out);
void host_make_sdf(const uint8_t* mask, int w, int h, const int32_t* rmin, const int32_t* rmax,
                   bool normalize, bool truncate, float band, float* out);
int device_make_sdf(hipStream_t stream, const uint8_t* mask_dev, int w, int h, const int32_t* rmin,
                    const int32_t* rmax, bool normalize, bool truncate, float band, void* scratch,
                    float* sdf_dev);
size_t device_make_sdf_scratch_bytes(int w, int h);
int device_make_sdf_batch(hipStream_t stream, int n, const uint8_t* const* masks_dev, const vcy_view* views,
                          bool normalize, bool truncate, float band, char* scratch, size_t scratch_stride,
                          float* const* sdf_dev);
// host arrays of returned meshes (page-locked pool, vcy_api.hip); released by vcy_mesh_free
void* mesh_host_alloc(size_t bytes, bool* pinned_out = nullptr);
void mesh_host_free(void* p);
// utility kernels (vcy_api.hip)
int ensure_count_width(vcy_ctx* ctx, int64_t max_count);  // d_cnt wide enough for counts up to max_count (vcy_api.hip)
int count_width_for(const vcy_ctx* ctx, int64_t max_count);
int convert_counts(hipStream_t stream, const void* src, int src_bytes, void* dst, int dst_bytes, int64_t n);  // saturating
int fill_state(vcy_ctx* ctx);   // marks the slab fresh (lazy)
int materialize(vcy_ctx* ctx);  // writes the fresh state to HBM if it is still pending

}  // namespace vcy
