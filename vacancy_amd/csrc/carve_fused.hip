// K1 (fused): carve up to 64 views per launch with the voxel state held in registers.
//
// Replaces the loop `for each view: Carve(camera, roi, sdf)` (reference voxel_carver.cc:516-528
// around :415-496).  Voxels are independent and every voxel sees its views in sequence order,
// so fusing views changes nothing but where the state lives.  A workgroup is four independent
// waves; each WAVE owns an 8x8x8 brick (lane = (y & 7) | (z << 3), 8 voxels along x per lane),
// loads sdf/update_num ONCE (not at all for a fresh grid), applies all views and writes back only
// what changed.  The four wave bricks of a workgroup are adjacent in x, so together they read
// 128-byte row segments.
//
// Per view the wave keeps the image footprint of its brick in a wave-private LDS tile: 16 x 16 raw pixels
// (kTileRaw: global memory -> LDS directly, double buffered, a sample = two ds_read2_b32), or for footprints
// beyond 15 x 15 pixels a raw tile of up to 2048 pixels with the footprint's own pitch, filled in place (kTileBig).
// Either way the reference's ROI clamps of x + 1 and y + 1 (voxel_carver.cc:51-66) are applied when the
// tile is filled, never per sample.  No workgroup barrier anywhere.  A voxel whose projection falls
// outside the staged tile (brick near the camera plane, footprint larger than the tile, outside the ROI)
// takes the generic global-memory path of carve_common.h, so SAMPLING never depends on the footprint
// estimate.  DROPPING a view for a brick does (see the kernel): it is only done when the
// footprint rectangle is provably a superset of every sample, with an explicit error margin.
//
// The arithmetic of a sample is the reference's, operation for operation (carve_common.h);
// the two divides fx/z, fy/z (camera.cc:133-136) use the same Newton sequence the compiler
// emits for IEEE division minus the exponent pre-scaling, which is a no-op for operands in
// [2^-60, 2^60]; anything outside that range takes the generic path.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <type_traits>
#include <vector>

#include "carve_common.h"

namespace vcy {

// The instances of carve_fused_kernel are compiled in TWO translation units of their own -- carve_fused_u8.hip (update_num
// in one byte) and carve_fused_u16.hip (two bytes), each `#define VCY_FUSED_PART` + `#include "carve_fused.hip"`: the
// kernel, its helpers and the launch_fused_* dispatch below up to launch_fused_1, then one exported function -- so that
// the 960 instantiations build in parallel halves (4.4 min in one unit).  This file compiled by itself is the host side
// and the small kernels; it instantiates no carve kernel.  (Pointers to types of the anonymous namespace cross as void*.)
void launch_fused_counts8(bool big, int update, bool trunc, bool samef, bool checkmax, unsigned grid_x, hipStream_t s,
                          const GridParams& g, const void* views, const float* c2, int nv, const ModeParams& m, int nbx, int nby,
                          int cull, int state_flags, const void* records, int64_t nbricks, float* bmin, const int* wgl,
                          unsigned long long* pcnt, int row_units);
void launch_fused_counts16(bool big, int update, bool trunc, bool samef, bool checkmax, unsigned grid_x, hipStream_t s,
                           const GridParams& g, const void* views, const float* c2, int nv, const ModeParams& m, int nbx, int nby,
                           int cull, int state_flags, const void* records, int64_t nbricks, float* bmin, const int* wgl,
                           unsigned long long* pcnt, int row_units);

namespace {

// Wave priority: everything but the runs over the voxels is short and ends in a memory request (view records,
// window lookups, the next tile, the state) whose latency nothing of this wave can cover; it runs at raised
// priority so that those requests leave as early as possible while the other waves of the SIMD are in their
// runs.  +2 ... 3 % in every mode.
#define VCY_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
// Waves per workgroup.  The waves never talk to each other, so 1 and 2 are just as correct; measured, they are
// 3 % slower in the default mode (four bricks adjacent in x start together and share rows and footprints).
#ifndef VCY_WG_WAVES
#define VCY_WG_WAVES 4   // (development builds: 8 is 7 % faster for single-view weighted-average launches, 7 % slower for the fused 32-view launch)
#endif
constexpr int kWgWaves = VCY_WG_WAVES;
constexpr int BX = 8 * kWgWaves, BY = 8, BZ = 8;  // voxels per workgroup: kWgWaves 8x8x8 wave bricks along x
constexpr int WX = 8;                    // wave brick is WX x BY x BZ, lane = (y & 7) | (z << 3), WX voxels per lane
constexpr int kMaxFusedViews = 64;         // one prologue lane per view
// Raw-pixel tile (the default: footprints up to 15 x 15 taps): 16 x 16 pixels of the image, pitch 16, 1 KB.
// The pixels go from global memory straight into LDS (global_load_lds_dword: lane L of the r-th load
// writes dword 64 r + L, i.e. pixel (L & 15, 4 r + (L >> 4))), two tiles per wave so that the next live
// view's footprint arrives while the current one is sampled -- no staging registers, no LDS stores, one
// address per PIXEL instead of four per quad.  A sample reads its four taps as two ds_read2_b32
// (offsets 0, 1 and 16, 17); columns / rows beyond the ROI repeat the edge pixel, which is the
// reference's clamp of x + 1 and y + 1 (voxel_carver.cc:51-66).
constexpr int kTileRaw = 16;
constexpr int kRawBuffers = 2;
template <int TQ>
constexpr int tile_f4_per_wave() { return TQ == kTileRaw ? kRawBuffers * 64 : TQ; }  // LDS of one wave, in float4
// The big tile: 8 KB per wave = 2048 raw pixels, pitch = width of the footprint, filled in place by LDS-direct
// loads (tile_fill; footprints up to ~44 x 44 pixels, voxels up to ~3 px); the second tap row is one address
// add away.
constexpr int kTileBig = 512;            // (in float4 units)
constexpr int kBigPixels = 4 * kTileBig;

// Cooperative write-back (state_flags bit 3, launch_carve_fused): the four waves of a workgroup hand their bricks' state
// to each other through LDS and every store instruction then writes whole 128-byte (sdf) / 64-byte (update_num) row
// segments instead of 64 scattered 16-byte pieces -- see the write-back of carve_fused_kernel.  Row pitches padded by
// 16 bytes so that the 16-byte LDS accesses of both directions spread over the banks.
constexpr int kCoopSdfPitch = 8 * VCY_WG_WAVES + 4;          // floats per row of the workgroup's 64 rows
template <typename CountT>
constexpr int coop_cnt_pitch() { return 8 * VCY_WG_WAVES + 16 / (int)sizeof(CountT); }  // counters per row
template <typename CountT>
constexpr size_t coop_lds_bytes() {
  return 64 * (size_t)kCoopSdfPitch * sizeof(float) + 64 * (size_t)coop_cnt_pitch<CountT>() * sizeof(CountT) +
         2 * VCY_WG_WAVES * sizeof(unsigned long long);  // (changed-lane masks, then "this wave takes part in the stores")
}

// ---- the few-view flavour (template parameter NB > 1 of carve_fused_kernel) -------------------------------------------
// The reference's own call pattern (examples.cc:117-149) carves ONE view per call, with an extraction in between: every
// view is a launch of its own, and such a launch spends more of a wave's life on the wave than on its 512 voxels -- block
// decode, axis loads, record unpack, tile request, barrier and write-back, 378 scalar + 290 vector instructions per brick
// next to the 290 of the voxels (profiles/r05/single_view_attribution.txt), on a CU with ONE scalar unit.  For launches of
// up to kRowMaxViews views a WAVE therefore walks NB consecutive bricks of one (y, z) row -- a "segment"; with NB = 4 the
// 32 x 8 x 8 block a workgroup of the NB = 1 kernel owns:
//   - block decode, axis loads, the records of all NB x nviews pairs (lane 8 j + v), the early-out test: once;
//   - the state of every live brick of the segment is requested up front with LDS-direct loads (global_load_lds: no
//     registers, no waits) into a staging area of the wave, and read from there when the brick's turn comes;
//   - ONE loop over the live (brick, view) pairs, in pair order: the tile of the next pair -- whether the next view of
//     this brick or the first live view of the next brick -- is in flight while this one is carved, exactly as the
//     NB = 1 kernel does between the views of its one brick;
//   - results go back to the staging area and leave it as whole row segments: 8 lanes store the 128 contiguous bytes
//     of sdf the segment has in one voxel row, 4 lanes its counters -- what the cooperative write-back gets from four
//     waves and a barrier, without the barrier.
// Waves never talk to each other; a workgroup is just kRowWaves of them.
// MEASURED (round 6, 1024^3 @1280x720, one view per launch; profiles/r06/row_kernel.txt): bit-identical to the NB = 1
// kernel, and SLOWER -- weighted average 3.17 ms per view against 2.63, first view 2.15 against 1.85, kMax 0.74 against
// 0.61.  The counters say why: a segment of four bricks costs 2062 vector + 1071 scalar instructions where four NB = 1
// waves cost 2400 + 1376 -- the decode, axis loads and barrier that are amortised were a seventh of the overhead, the
// rest is per brick and per pair whoever walks them -- while the staging area (14 KB per wave) leaves 2.7 waves per SIMD
// where the NB = 1 kernel has 5.7, and the run loops need the other waves to cover their LDS and scalar-load latencies.
// Two bricks per wave and four waves per workgroup (8 KB, 5 waves per SIMD) come closest (2.91 / 1.91 / 0.67 ms) and
// amortise next to nothing (592 + 324 per brick); workgroups of one or two waves are slower again (dispatch rate).
// So the flavour is OFF by default ("rowkernel" 0), kept and tested as the second implementation of few-view launches.
constexpr int kRowMaxViews = 8;          // pairs are numbered 8 j + v
#ifndef VCY_ROW_BRICKS
#define VCY_ROW_BRICKS 4
#endif
#ifndef VCY_ROW_WAVES
#define VCY_ROW_WAVES 2
#endif
constexpr int kRowBricks = VCY_ROW_BRICKS;
constexpr int kRowWaves = VCY_ROW_WAVES;
template <typename CountT, int NB>
constexpr size_t row_lds_bytes_per_wave() {  // two raw tiles, NB x 8 TileInfo, the state of NB bricks, NB changed-lane masks
  return (size_t)kRawBuffers * 1024 + (size_t)NB * kRowMaxViews * 56 /* sizeof(TileInfo) */ +
         (size_t)NB * 64 * WX * (sizeof(float) + sizeof(CountT)) + (size_t)NB * sizeof(unsigned long long);
}

constexpr int kWmaxPlanes = 2;           // window sizes 4 and 8
constexpr int kLiveListMaxViews = 8;      // launches of up to this many views over a carved grid list their live workgroups first
constexpr int64_t kRecordBytesMax = (int64_t)2 << 30;  // footprint records of one carve launch (see launch_carve_fused)


// tuning knobs of the select-free view loop (development builds override them, profiles/tools/build_variant.sh)
#ifndef VCY_FAST_GROUP
#define VCY_FAST_GROUP 4   // voxels whose LDS reads are in flight together
#endif
// Waves per SIMD the kernels are compiled for (register budget 512 / waves).  The kernels whose work is done by
// the select-free loop (raw tiles, no update_num limit in reach) need 57-59 VGPRs there; what
// wants more is the checked loop with its call of the generic sampler, which those kernels rarely enter.  They
// are compiled for 7 waves (72 VGPRs: a handful of spills, placed in the rare blocks by the branch weights at
// the loop selection; 8 waves spill in the tile staging as well and lose 15 %).  The others keep 5.
#ifndef VCY_WAVES
#define VCY_WAVES 7
#endif
#ifndef VCY_WAVES_CHECKED
#define VCY_WAVES_CHECKED 5
#endif
// The unit-weight weighted average keeps update_num as floats next to sdf and carries the brick-wide weights: at 7 waves
// its PROLOGUE spills 24 bytes per lane, which every wave executes -- 3 GB of scratch written back per single-view
// launch at 1024^3 (the L2 turns over every 8 us there), a third of what the launch has to write at all
// (profiles/r04/per_view_tsdf_pmc.txt).
#ifndef VCY_WAVES_WA
#define VCY_WAVES_WA 6
#endif
// The one-view instances (NB == 0) have no view loop to keep registers for: the unit-weight average fits 7 waves (first
// view on a fresh grid 1.61 -> 1.47 ms, later views 2.53 -> 2.50; 5 waves: 1.81 / 2.65 -- profiles/r06/one_view_waves.txt)
#ifndef VCY_WAVES_WA_ONE
#define VCY_WAVES_WA_ONE 7
#endif
// ... and kMax 8, which its one tile buffer makes room for in LDS (first view 1.38 -> 1.32 ms, later views 0.537 -> 0.520;
// the unit-weight average at 8: 2.50 -> 2.82, spills)
#ifndef VCY_WAVES_ONE
#define VCY_WAVES_ONE 8
#endif


// Development build only (-DVCY_PHASE_TIMING, profiles/tools/phase_timing.py): s_memtime ticks of every wave,
// accumulated per phase of the fused kernel.  Slots 0-6: prologue + state load, tile staging, select-free
// view, sure view, checked view, re-bounding after a change, write-back; 7-9: views taken by the three
// loops; 10: waves; 11: views that changed their brick; 12-14: parts of slot 0 (until the kernel arguments
// and axis tables are there, brick_footprints, state + first live set).
#ifdef VCY_PHASE_TIMING
__device__ unsigned long long g_phase_ticks[256][16];
#define VCY_PT_DECL unsigned long long pt_last = __builtin_amdgcn_s_memtime(), pt_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define VCY_PT(slot)                                                  \
  do {                                                                \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();       \
    pt_acc[slot] += t_ - pt_last;                                     \
    pt_last = t_;                                                     \
  } while (0)
#define VCY_PT_COUNT(slot) pt_acc[slot] += 1
#define VCY_PT_FLUSH(lane_)                                                                         \
  do {                                                                                              \
    if ((lane_) == 0)                                                                               \
      for (int q_ = 0; q_ < 16; ++q_) atomicAdd(&g_phase_ticks[blockIdx.x & 255][q_], pt_acc[q_]);  \
  } while (0)
#else
#define VCY_PT_DECL
#define VCY_PT(slot)
#define VCY_PT_COUNT(slot)
#define VCY_PT_FLUSH(lane_)
#endif

struct FusedView {
  ViewParams v;
  // Window maxima of the SDF image (built per launch by wmax_k4 / wmax_k8 below), or null:
  //   wmax[p * plane + y * width + x] = max of g over [x, x + k) x [y, y + k) clipped to the image,
  //   k = 4 (p = 0) or 8 (p = 1), g = the SDF value, or +inf where it is NaN / infinite.
  // The maximum over any pw x ph rectangle with min(pw, ph) >= k is then the maximum of
  // ceil(pw/k) * ceil(ph/k) entries (windows placed inside the rectangle, overlapping at the far
  // edges): the prologue bounds a footprint with a handful of loads instead of scanning it.
  const float* wmax;
  int wmax_plane;
  // Planes 2 and 3, when has_lower != 0: the same window maxima of -g, i.e. window MINIMA of the image
  // negated.  Only built for the truncating weighted average, where a tile whose every tap is provably
  // >= -1 needs no `dist < -1` test per sample (TileInfo::sure bit 1).
  int has_lower;
  // The planes are only filled inside wrect = {x0, y0, x1, y1} (x0, x1 multiples of 4), the image-space
  // bounding box of this context's slab plus a border wider than anything a footprint lookup reaches;
  // a z-slab of a sharded grid often sees a narrow band of the image.
  int wrect[4];
};
// c0_all[view][x brick][32]: the products c0 = R[i][0] * px[x] (one fp32 multiply per entry, done on the
// host) of the 8 voxels of one wave brick along x, laid out for wide scalar loads:
//   [2 k + 0] = R[0][0] px[x_k], [2 k + 1] = R[1][0] px[x_k]  (the (x, y) pair a packed add takes as one operand)
//   [16 + k]  = R[2][0] px[x_k]                               (two neighbours = one packed operand)
// 24 of 32 floats used (128-byte records); columns beyond nx repeat the last one.
constexpr int kC0Stride = 32;

struct TileInfo {
  float lo_x, hi_x, lo_y, hi_y;  // closed range of (u,v) whose taps are in the tile
  float pitchf;
  int base;                      // -(ty0*tw + tx0)
  int tx0, ty0, tw, nq;          // nq = tw*th quads; 0: no tile for this view
  int th;
  float inv_tw;                  // 1 / tw: q / tw == (int)((q + 0.5f) * inv_tw) for q < 2^12
  float ub;                      // upper bound of any sample taken from this tile (+inf: unknown)
  int sure;                      // bit 0: every voxel of the brick provably samples inside this tile;
                                 // bit 1: and every sample is provably >= -1 (no truncation skip possible)
};
static_assert(sizeof(TileInfo) == 56, "row_lds_bytes_per_wave");

// Correctly rounded n/d for normal operands away from the exponent limits: v_rcp_f32 plus the
// refinement steps of the standard fp32 division expansion (without v_div_scale/v_div_fixup).
__device__ __forceinline__ float div_fast(float n, float d) {
  float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  float q = n * r;
  const float e2 = __builtin_fmaf(-d, q, n);
  q = __builtin_fmaf(e2, r, q);
  const float e3 = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(e3, r, q);
}

// Shorter sequences for n / d.  They are NOT correct for every pair of floats, but for a given numerator
// they usually are for EVERY denominator: the host checks that exhaustively on the device, once per
// focal length (div_level below: all 2^23 significands in each of the 121 binades [2^-60, 2^61) the fast
// path admits), and only then selects the variant.  DIV 2: v_rcp_f32, one multiply, one correction;
// DIV 1: Newton step on the reciprocal first; DIV 0: the full IEEE expansion (div_fast).
// Plain (unpacked) fp32 throughout: on MI355X v_pk_*_f32 issue at half rate AND slow the scalar
// fp32 instructions around them (profiles/r02/valu_ubench.txt), while v_mul/v_fma_f32 issue every 2 cycles.
template <int DIV>
__device__ __forceinline__ float div_view(float n, float d) {
  if (DIV == 0) return div_fast(n, d);
  float r = __builtin_amdgcn_rcpf(d);
  if (DIV == 1) {
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
  }
  const float q = n * r;
  const float e2 = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(e2, r, q);
}

__device__ __forceinline__ float div_view1(int div, float n, float d) {  // scalar twin, for the checker
  float r = __builtin_amdgcn_rcpf(d);
  if (div == 1) {
    const float e = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
  }
  const float q = n * r;
  const float e2 = __builtin_fmaf(-d, q, n);
  return __builtin_fmaf(e2, r, q);
}

// 2^-60 <= z <= 2^60 (also false for negative z, NaN, inf, 0)
__device__ __forceinline__ bool in_fast_div_range(float z) {
  const unsigned lo = 0x21800000u;  // 2^-60
  const unsigned hi = 0x5d800000u;  // 2^60
  return (__float_as_uint(z) - lo) <= (hi - lo);
}

typedef const float __attribute__((address_space(1))) * gfloat_ptr;  // known-global loads
// base[idx] for idx < 2^30 with the BYTE offset formed in 32 bits: where `base` is uniform the load then takes a scalar
// base and one 32-bit vector offset (global_load_dword v, v, s[..]) instead of a 64-bit vector address
__device__ __forceinline__ float load_u32_index(gfloat_ptr base, unsigned idx) {
  typedef const char __attribute__((address_space(1))) * gchar_ptr;
  return *(gfloat_ptr)((gchar_ptr)base + (idx << 2));
}
typedef const float __attribute__((address_space(4))) * cfloat_ptr;  // read-only: scalar loads
// Generic sample for a voxel the staged tile does not cover (rare): global-memory taps and the
// full ROI / outside-image semantics of carve_common.h.  Kept out of line so that the hot loop
// stays small.
__device__ __attribute__((noinline)) bool sample_generic(const ViewParams* v, ModeParams m, float px,
                                                         float py, float pz, float* dist) {
  return view_distance<true, 0, 0, false, false>(*v, m, px, py, pz, dist);
}

// LDS traffic inside one wave needs ordering against the compiler only (DS ops of a wave are
// executed in issue order).
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Internal update mode: kWeightedAverage with voxel_update_weight == 1.0f (the default weight).
constexpr int kUpdateWaUnitWeight = 2;

// Correctly rounded 1.0f / m for the integers m = 1 .. 65536 (update_num + 1 of a u8 / u16 counter):
// v_rcp_f32 and ONE Newton step.  Unlike div_fast this is not correct for every float; that it is for
// every m in the range is checked exhaustively on the device by vcy_selftest (and by the GPU tests).
__device__ __forceinline__ float rcp_count(float m) {
  const float r = __builtin_amdgcn_rcpf(m);
  const float e = __builtin_fmaf(-m, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}

// Branch-free voxel update (select form of fuse() in carve_common.h): first touch
// (voxel_carver.cc:482-486), UpdateVoxelMax (:78-86) or UpdateVoxelWeightedAverage (:88-95).
template <int UPDATE>
__device__ __forceinline__ bool apply_sample(bool ok, float dist, float wgt, float& s, int& n) {
  if (UPDATE == VCY_UPDATE_MAX) {
    const bool take = ok && (n < 1 || dist > s);
    s = take ? dist : s;
    n += take ? 1 : 0;
    return take;
  } else if (UPDATE == kUpdateWaUnitWeight) {
    // voxel_update_weight == 1: w * x == x exactly, and the denominator is the integer n + 1
    const float inv_denom = rcp_count((float)(n + 1));
    const float avg = ((float)n * s + dist) * inv_denom;
    const float ns = (n < 1) ? dist : avg;
    s = ok ? ns : s;
    n += ok ? 1 : 0;
  } else {
    const float inv_denom = div_fast(1.0f, wgt * (float)(n + 1));
    const float avg = (wgt * (float)n * s + wgt * dist) * inv_denom;
    const float ns = (n < 1) ? dist : avg;
    s = ok ? ns : s;
    n += ok ? 1 : 0;
  }
  return ok;
}

// ---- update sequences of the fast path ---------------------------------------------------------
// Measured on MI355X (profiles/r02/valu_ubench.txt): a compare / select / carry chain through VCC (the
// VOPC / VOP2 encodings) issues in about 3.5 cycles per instruction and overlaps with full-rate fp32
// instructions of other voxels; the same chain through an arbitrary SGPR pair (VOP3 encodings, what the
// compiler picks once several voxels are in flight) takes 7 per instruction, and an EXEC-masked variant
// (v_cmpx) more.  The chains are therefore written out with VCC.
//
// UpdateVoxelMax for a voxel that has been touched before (voxel_carver.cc:78-86):
//   if (dist > sdf) { sdf = dist; ++update_num; }      -- NaN compares false, -0 == +0 stay put
// `took` accumulates the lanes that changed.
__device__ __forceinline__ void update_max_touched(float dist, float& s, int& n, unsigned long long& took) {
  asm("v_cmp_gt_f32_e32 vcc, %[d], %[s]\n\t"
      "s_or_b64 %[took], %[took], vcc\n\t"
      "v_cndmask_b32_e32 %[s], %[s], %[d], vcc\n\t"
      "v_addc_co_u32_e32 %[n], vcc, 0, %[n], vcc"
      : [s] "+v"(s), [n] "+v"(n), [took] "+s"(took)
      : [d] "v"(dist)
      : "vcc");
}

// UpdateVoxelWeightedAverage with voxel_update_weight == 1 (voxel_carver.cc:88-95) behind the truncation
// skip (:478), for a voxel whose counter is kept as a float `fn` (exact below 2^24):
//   if (!(dist < -1)) { sdf = (fn * sdf + dist) * (1 / (fn + 1)); fn += 1; }
// 1 / (fn + 1) is rcp_count() -- v_rcp_f32 and one Newton step, the correctly rounded quotient for every
// count a u8 / u16 counter can hold (vcy_selftest).  Requires "update_num == 0 implies sdf == lowest()"
// (state only ever written by the fill and the carve kernels): then the first touch needs no special
// case, (0 * sdf + dist) * 1 == dist bit for bit (0 * lowest() = -0, -0 + dist = dist).
template <bool TRUNC>
__device__ __forceinline__ void update_wa_unit(float dist, float& s, float& fn, unsigned long long& took) {
  const float f1 = fn + 1.0f;
  const float avg = (fn * s + dist) * rcp_count(f1);
  if (TRUNC) {
    asm("v_cmp_ngt_f32_e32 vcc, -1.0, %[d]\n\t"   // !(-1 > d)  ==  !(d < -1), true for NaN like the reference
        "s_or_b64 %[took], %[took], vcc\n\t"
        "v_cndmask_b32_e32 %[s], %[s], %[avg], vcc\n\t"
        "v_cndmask_b32_e32 %[fn], %[fn], %[f1], vcc"
        : [s] "+v"(s), [fn] "+v"(fn), [took] "+s"(took)
        : [d] "v"(dist), [avg] "v"(avg), [f1] "v"(f1)
        : "vcc");
  } else {
    s = avg;
    fn = f1;
  }
}

typedef float f4 __attribute__((ext_vector_type(4)));
typedef f4 __attribute__((address_space(3))) lds_float4;

// The weighted-average kernels keep update_num as a float in registers (exact below 2^24; it is converted
// at the load and the store of the brick): (float)n and (float)(n + 1) of the reference's formula are
// then fn and fn + 1 without conversions.
template <int UPDATE>
__device__ __forceinline__ bool apply_sample(bool ok, float dist, float wgt, float& s, float& fn) {
  const float f1 = fn + 1.0f;
  float avg;
  if (UPDATE == kUpdateWaUnitWeight) {
    avg = (fn * s + dist) * rcp_count(f1);
  } else {
    avg = (wgt * fn * s + wgt * dist) * div_fast(1.0f, wgt * f1);
  }
  const float ns = (fn < 1.0f) ? dist : avg;
  s = ok ? ns : s;
  fn = ok ? f1 : fn;
  return ok;
}

// q / d for 0 <= q < 4096, 1 <= d <= 1024, given inv = 1.0f / d: (q + 0.5) / d is never within
// 0.5 / d of an integer, far more than the float rounding of the product.
__device__ __forceinline__ int div_small(int q, float inv) { return (int)(((float)q + 0.5f) * inv); }

typedef float __attribute__((address_space(3))) lds_float;
typedef uint32_t __attribute__((address_space(3))) lds_u32;

// Raw tile of view `v` into the wave-private LDS buffer `buf` (256 floats): pixel (i, j) of the tile =
// image pixel (min(tx0 + i, roi_max.x), min(ty0 + j, roi_max.y)), for the th + 1 <= 16 rows the taps reach.
// Asynchronous: the data is in LDS once the wave's vmcnt has drained (raw_tile_wait).
__device__ __forceinline__ void raw_prefetch(const ViewParams& v, const TileInfo& ti, int lane, float* buf) {
#ifdef VCY_FLOOR_NO_TILE_LOADS  // development build (issue floor, profiles/tools/issue_floor.sh): the taps read whatever LDS holds
  return;
#endif
  // (opaque: lane >> 4 and lane & 15 are formed here, every time -- hoisted out of the view loop they were two more
  // registers live through every view, and the weighted-average kernels spilled exactly those to scratch)
  asm volatile("" : "+v"(lane));
  const int nq = __builtin_amdgcn_readfirstlane(ti.nq);
  if (nq == 0) return;
  const int th = __builtin_amdgcn_readfirstlane(ti.th);
  const int tx0 = __builtin_amdgcn_readfirstlane(ti.tx0);
  const int ty0 = __builtin_amdgcn_readfirstlane(ti.ty0);
  gfloat_ptr img = (gfloat_ptr)v.sdf;
  const unsigned width = (unsigned)v.width;
  lds_float* dst = (lds_float*)buf;
  if (tx0 + 15 <= v.roi_max_xi && ty0 + 15 <= v.roi_max_yi) {
    // The usual case (uniform test): the whole 16 x 16 window lies inside the ROI, nothing is clamped.  The
    // address is a scalar base per group of four rows plus one per-lane offset that only depends on the
    // image width: one vector instruction per load.
    const unsigned lane_off = __umul24(width, (unsigned)lane >> 4) + ((unsigned)lane & 15u);
    gfloat_ptr base = img + (__umul24(width, (unsigned)ty0) + (unsigned)tx0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (4 * r <= th)  // uniform: rows 4 r .. 4 r + 3 hold a tap row (taps reach rows 0 .. th)
        __builtin_amdgcn_global_load_lds(base + (size_t)(4 * r) * width + lane_off, dst + 64 * r, 4, 0, 0);
    }
    return;
  }
  const unsigned xx = (unsigned)min(tx0 + (lane & 15), v.roi_max_xi);
  const int yl = ty0 + (lane >> 4);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (4 * r <= th) {
      const unsigned yy = (unsigned)min(yl + 4 * r, v.roi_max_yi);
      __builtin_amdgcn_global_load_lds(img + (__umul24(width, yy) + xx), dst + 64 * r, 4, 0, 0);
    }
  }
}

__device__ __forceinline__ void raw_tile_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// min over the wave (NaN operands are ignored, like the `dist > s` test ignores them): six DPP
// v_min_f32 (row reduction, then row_bcast 15 / 31) and one readlane.  Written in assembly because
// the compiler expands a DPP move + canonicalise + min per step; the s_nop covers the VALU-write ->
// DPP-read hazard the assembler does not see inside an asm block.
__device__ __forceinline__ float wave_min(float v) {
#define VCY_DPP_MIN(CTRL) asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 " CTRL : "+v"(v))
  VCY_DPP_MIN("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
  VCY_DPP_MIN("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
  VCY_DPP_MIN("row_half_mirror row_mask:0xf bank_mask:0xf");
  VCY_DPP_MIN("row_mirror row_mask:0xf bank_mask:0xf");  // every lane of a 16-lane row holds the row minimum
  VCY_DPP_MIN("row_bcast:15 row_mask:0xa bank_mask:0xf");  // rows 1, 3 <- min(own, row 0 / 2)
  VCY_DPP_MIN("row_bcast:31 row_mask:0xc bank_mask:0xf");  // rows 2, 3 <- min(own, row 1): lane 63 = all
#undef VCY_DPP_MIN
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 63));
}

// Fills a whole (big) tile in place: pixel (i, j) of the (tw + 1) x (th + 1) window = image pixel
// (min(tx0 + i, roi_max.x), min(ty0 + j, roi_max.y)).  Every group of 64 consecutive tile pixels is one
// LDS-direct request (lane L -> tile element 64 r + L); all requests are issued before the one wait.
__device__ __forceinline__ void tile_fill(const ViewParams& v, const TileInfo& ti, int lane, float* tile) {
  const int nq = __builtin_amdgcn_readfirstlane(ti.nq);
  if (nq == 0) return;
  const int tw = __builtin_amdgcn_readfirstlane(ti.tw), th = __builtin_amdgcn_readfirstlane(ti.th);
  const int tx0 = __builtin_amdgcn_readfirstlane(ti.tx0);
  const int ty0 = __builtin_amdgcn_readfirstlane(ti.ty0);
  const float inv_pitch = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(ti.inv_tw)));
  const int pitch = tw + 1, npx = pitch * (th + 1);
  gfloat_ptr img = (gfloat_ptr)v.sdf;
  const unsigned width = (unsigned)v.width;
  lds_float* dst = (lds_float*)tile;
  for (int q0 = 0; q0 < npx; q0 += 64) {  // (uniform)
    const int q = q0 + lane;
    if (q < npx) {  // lanes beyond the window request nothing (and write nothing)
      const int j = div_small(q, inv_pitch), i = q - j * pitch;
      const unsigned xx = (unsigned)min(tx0 + i, v.roi_max_xi), yy = (unsigned)min(ty0 + j, v.roi_max_yi);
      __builtin_amdgcn_global_load_lds(img + (__umul24(width, yy) + xx), dst + q0, 4, 0, 0);
    }
  }
  raw_tile_wait();
}

// ---- window maxima (FusedView::wmax) --------------------------------------------------------
// Each thread produces four consecutive pixels of a row; blockIdx.y = view.  Rows are read as float4
// when the image allows it (width % 4 == 0, 16-byte aligned base), element-wise otherwise.  Pixels
// beyond the image border count as -inf, i.e. windows are clipped to the image.

// a[0..7] = row[x0 .. x0 + 7] (x0 % 4 == 0, x0 < w); row == nullptr: a row below the image
__device__ __forceinline__ void wmax_load8(const float* __restrict__ row, int x0, int w, bool vec, float a[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = -INFINITY;
  if (row == nullptr) return;
  if (vec) {
    const float4 lo = *(const float4*)(row + x0);
    a[0] = lo.x, a[1] = lo.y, a[2] = lo.z, a[3] = lo.w;
    if (x0 + 4 < w) {
      const float4 hi = *(const float4*)(row + x0 + 4);
      a[4] = hi.x, a[5] = hi.y, a[6] = hi.z, a[7] = hi.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (x0 + j < w) a[j] = row[x0 + j];
  }
}

__device__ __forceinline__ void wmax_store4(float* __restrict__ row, int x0, int w, bool vec, const float o[4]) {
  if (vec) {
    *(float4*)(row + x0) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (x0 + j < w) row[x0 + j] = o[j];
  }
}

// Plane 0: maxima of 4 x 4 windows of the image, non-finite pixels counted as +inf (a footprint
// holding one gives no bound: 0 * inf = NaN samples).  NEG: of the negated image, into plane 2.
template <bool NEG>
__global__ __launch_bounds__(256) void wmax_k4_kernel(const FusedView* __restrict__ views) {
  const FusedView& fv = views[blockIdx.y];
  const int w = fv.v.width, h = fv.v.height, wq = (fv.wrect[2] - fv.wrect[0]) >> 2;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (fv.wmax == nullptr || wq <= 0 || t >= wq * (fv.wrect[3] - fv.wrect[1])) return;
  const int yr = t / wq, y = fv.wrect[1] + yr, x0 = fv.wrect[0] + ((t - yr * wq) << 2);
  const float* img = fv.v.sdf;
  const bool vec = (w & 3) == 0 && (((uintptr_t)img | (uintptr_t)fv.wmax) & 15) == 0;
  float o[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float a[8];
    wmax_load8(y + r < h ? img + (size_t)(y + r) * w : nullptr, x0, w, vec, a);
#pragma unroll
    for (int j = 0; j < 8; ++j)  // NaN, +inf, -inf -> +inf; the -inf padding beyond the border stays
      if (x0 + j < w && y + r < h) a[j] = (fabsf(a[j]) <= 3.402823466e+38f) ? (NEG ? -a[j] : a[j]) : INFINITY;
    float p[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) p[j] = fmaxf(a[j], a[j + 1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = fmaxf(o[j], fmaxf(p[j], p[j + 2]));
  }
  wmax_store4(const_cast<float*>(fv.wmax) + (NEG ? 2 * (size_t)fv.wmax_plane : (size_t)0) + (size_t)y * w, x0, w, vec, o);
}

// Plane 1: maxima of 8 x 8 windows = the four 4 x 4 windows at offsets 0 / 4 of plane 0.
template <bool NEG>
__global__ __launch_bounds__(256) void wmax_k8_kernel(const FusedView* __restrict__ views) {
  const FusedView& fv = views[blockIdx.y];
  const int w = fv.v.width, h = fv.v.height, wq = (fv.wrect[2] - fv.wrect[0]) >> 2;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (fv.wmax == nullptr || wq <= 0 || t >= wq * (fv.wrect[3] - fv.wrect[1])) return;
  const int yr = t / wq, y = fv.wrect[1] + yr, x0 = fv.wrect[0] + ((t - yr * wq) << 2);
  const float* in = fv.wmax + (NEG ? 2 * (size_t)fv.wmax_plane : (size_t)0);  // (planes are multiples of 4 floats or vec is off)
  const bool vec = (w & 3) == 0 && ((uintptr_t)in & 15) == 0;
  float a[8], b[8], o[4];
  wmax_load8(in + (size_t)y * w, x0, w, vec, a);
  wmax_load8(y + 4 < h ? in + (size_t)(y + 4) * w : nullptr, x0, w, vec, b);
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = fmaxf(fmaxf(a[j], a[j + 4]), fmaxf(b[j], b[j + 4]));
  wmax_store4(const_cast<float*>(in) + (size_t)fv.wmax_plane + (size_t)y * w, x0, w, vec, o);
}

// Footprint of one brick in one view: the tile of SDF pixels its samples read, whether every sample provably
// lies inside it (`sure`), and bounds of those samples (TileInfo).
// The brick is convex, so the exact projections of its voxels lie in the hull of the exact
// projections of its 8 corners.  Corners and voxels are both COMPUTED with a few float operations;
// the rectangle is only trusted when an explicit first-order bound of those errors (err_u, err_w
// below) is well inside the margin added around the corner hull.  Nothing here needs the exact
// arithmetic of the samples: corners come from the linear form p000 + {0,ax} + {0,ay} + {0,az} and
// an approximate reciprocal.
template <bool SAMEF, int TQ, bool GEN>
__device__ __forceinline__ TileInfo footprint_of(const FusedView& fv, float xl, float xh, float yl, float yh, float zl_,
                                                 float zh, bool is_ortho, bool outside_max, bool want_bound,
                                                 bool want_lower, float* lower_out = nullptr) {
  const ViewParams& v = fv.v;
  const float xa = fmaxf(fabsf(xl), fabsf(xh)), ya = fmaxf(fabsf(yl), fabsf(yh)), za = fmaxf(fabsf(zl_), fabsf(zh));
  const bool ortho = GEN && is_ortho;
  float p0[3], ax[3], ay[3], az[3], mag[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    p0[i] = v.t[i] + (v.r[i][0] * xl + (v.r[i][1] * yl + v.r[i][2] * zl_));
    ax[i] = v.r[i][0] * (xh - xl);
    ay[i] = v.r[i][1] * (yh - yl);
    az[i] = v.r[i][2] * (zh - zl_);
    // magnitude of the terms of pc[i]: its computed value is within ~2^-21 * mag[i] of the exact one
    mag[i] = fabsf(v.t[i]) + (fabsf(v.r[i][0]) * xa + (fabsf(v.r[i][1]) * ya + fabsf(v.r[i][2]) * za));
  }
  float umin = INFINITY, umax = -INFINITY, wmin = INFINITY, wmax_ = -INFINITY, zmin = INFINITY, zmax = -INFINITY;
  int bad = 0;
  const float fx = v.fx, fy = SAMEF ? v.fx : v.fy;
  float pxy[4][3];  // p0, p0 + ax, p0 + ay, p0 + ax + ay
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    pxy[0][i] = p0[i];
    pxy[1][i] = p0[i] + ax[i];
    pxy[2][i] = p0[i] + ay[i];
    pxy[3][i] = pxy[1][i] + ay[i];
  }
#pragma unroll
  for (int corner = 0; corner < 8; ++corner) {
    float pc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) pc[i] = (corner & 4) ? pxy[corner & 3][i] + az[i] : pxy[corner & 3][i];
    float u = pc[0], w = pc[1];
    if (!ortho) {
      const float rz = __builtin_amdgcn_rcpf(pc[2]);  // (its operand range is checked on zmin / zmax below)
      u = __builtin_fmaf(fx * rz, pc[0], v.cx);
      w = __builtin_fmaf(fy * rz, pc[1], v.cy);
    }
    umin = fminf(umin, u);
    umax = fmaxf(umax, u);
    wmin = fminf(wmin, w);
    wmax_ = fmaxf(wmax_, w);
    zmin = fminf(zmin, pc[2]);
    zmax = fmaxf(zmax, pc[2]);
  }
  // in front of the camera, reciprocals finite and normal: every corner depth inside div_fast's range -- tested on the
  // smallest and the largest (a NaN depth would slip through fminf / fmaxf, but NaN / huge inputs end up in mag, next line)
  if (!ortho) bad |= !in_fast_div_range(zmin) || !in_fast_div_range(zmax);
  // finite inputs (NaN / huge values anywhere end up in mag), image coordinates of sane size
  bad |= !(mag[0] < 0x1p60f) || !(mag[1] < 0x1p60f) || !(mag[2] < 0x1p60f);
  bad |= !(umin > -1.0e6f) || !(umax < 1.0e6f) || !(wmin > -1.0e6f) || !(wmax_ < 1.0e6f);
  const float uabs = fmaxf(fabsf(umin), fabsf(umax)), wabs = fmaxf(fabsf(wmin), fabsf(wmax_));
  // |computed - exact| of an image coordinate, corner or voxel (first order, constants rounded up):
  //   pinhole  u = fx * X / Z + cx:  fx * dX / Z + |u - cx| * dZ / Z + rounding of the last operations,
  //            with dX <= 2^-21 mag_x, dZ <= 2^-21 mag_z and Z >= zmin;
  //   ortho    u = X:                dX.
  float err_u, err_w;
  if (ortho) {
    err_u = 0x1p-21f * mag[0];
    err_w = 0x1p-21f * mag[1];
  } else {
    bad |= !(zmin * 4.0f >= zmax);
    const float iz = 0x1p-21f * __builtin_amdgcn_rcpf(zmin) * 1.0001f;
    err_u = iz * (fx * mag[0] + (uabs + fabsf(v.cx)) * mag[2]) + 0x1p-21f * (uabs + fabsf(v.cx));
    err_w = iz * (fy * mag[1] + (wabs + fabsf(v.cy)) * mag[2]) + 0x1p-21f * (wabs + fabsf(v.cy));
  }
  const float margin = 0.125f;
  bad |= !(err_u <= 0.03125f) || !(err_w <= 0.03125f);  // corner error + voxel error <= margin / 2
  TileInfo ti;
  ti.lo_x = ti.lo_y = INFINITY;  // nothing passes the tile test
  ti.hi_x = ti.hi_y = -INFINITY;
  ti.pitchf = 0.0f;
  ti.base = 0;
  ti.tx0 = ti.ty0 = ti.tw = ti.nq = ti.th = 0;
  ti.inv_tw = 1.0f;
  ti.ub = INFINITY;  // never dropped
  ti.sure = 0;
  if (!bad) {
    const int tx0 = max((int)floorf(umin - margin), v.roi_min_xi);
    const int ty0 = max((int)floorf(wmin - margin), v.roi_min_yi);
    const int tx1 = min((int)floorf(umax + margin), v.roi_max_xi);
    const int ty1 = min((int)floorf(wmax_ + margin), v.roi_max_yi);
    const int tw = tx1 - tx0 + 1, th = ty1 - ty0 + 1;
    constexpr bool kRaw = TQ == kTileRaw;
    if (tw > 0 && th > 0 && (kRaw ? (tw <= 15 && th <= 15) : ((tw + 1) * (th + 1) <= kBigPixels))) {
      // Every computed (u, w) of the brick is within corner error + voxel error < margin of the corner
      // hull, so when the ROI clipped nothing it lies in [tx0, tx1 + 1) x [ty0, ty1 + 1); the depth
      // guard keeps every computed pc.z within a factor 2 of the corner range, inside div_fast's.
      const bool unclipped = (int)floorf(umin - margin) >= v.roi_min_xi && (int)floorf(wmin - margin) >= v.roi_min_yi &&
                             (int)floorf(umax + margin) < v.roi_max_xi && (int)floorf(wmax_ + margin) < v.roi_max_yi;
      const bool depth_ok = 0x1p-20f * mag[2] <= 0.25f * zmin && zmin >= 0x1p-58f && zmax <= 0x1p58f;
      // (|16 base| < 2^22: the fast path forms LDS addresses in the float pipeline, carve_view_fast)
      const int pitch = kRaw ? 16 : tw + 1;  // pixels per tile row
      const bool small_base = ty0 * pitch + tx0 < (1 << 18);
      // orthographic: no division, and the only depth test is the reference's `pc.z < 0` skip
      // (voxel_carver.cc:456): every computed pc.z of the brick is >= zmin - 2^-20 mag_z
      const bool depth_ok_ortho = zmin > 0x1p-20f * mag[2];
      ti.sure = (unclipped && (ortho ? depth_ok_ortho : depth_ok) && small_base) ? 1 : 0;
      ti.tx0 = tx0;
      ti.ty0 = ty0;
      ti.tw = tw;
      ti.th = th;
      ti.nq = tw * th;
      ti.inv_tw = 1.0f / (float)pitch;
      ti.pitchf = (float)pitch;
      ti.base = -(ty0 * pitch + tx0);
      ti.lo_x = (float)tx0;
      ti.lo_y = (float)ty0;
      // taps exist for floor(u) in [tx0, tx1]; at the ROI edge u == roi_max is still inside
      ti.hi_x = (tx1 == v.roi_max_xi) ? v.roi_max_x
                                      : __uint_as_float(__float_as_uint((float)(tx1 + 1)) - 1u);
      ti.hi_y = (ty1 == v.roi_max_yi) ? v.roi_max_y
                                      : __uint_as_float(__float_as_uint((float)(ty1 + 1)) - 1u);
      if (want_bound) {
        // maximum over every pixel a tap of this tile can read
        const int pw = min(tx1 + 1, v.roi_max_xi) - tx0 + 1;
        const int ph = min(ty1 + 1, v.roi_max_yi) - ty0 + 1;
        float m = -INFINITY;
        int has_nan = 0;
        gfloat_ptr wm = (gfloat_ptr)fv.wmax;
        // window maxima: k = 8 when both sides reach 8, else 4; nxw x nyw windows placed inside the rectangle
        // (a side shorter than k gets one window that sticks out of it: a maximum over more pixels is still
        // an upper bound, and the planes are filled well beyond any footprint, FusedView::wrect)
        const int L = min(pw, ph) >= 8 ? 3 : 2;
        const int k = 1 << L;
        const int nxw = (pw + k - 1) >> L, nyw = (ph + k - 1) >> L;
        // the largest window counts (up to 3) among the lanes that take the 3 x 3 path below: wave-uniform
        const bool small = nxw <= 3 && nyw <= 3;
        const int ux = __any(small && nxw >= 3) ? 3 : (__any(small && nxw >= 2) ? 2 : 1);
        const int uy = __any(small && nyw >= 3) ? 3 : (__any(small && nyw >= 2) ? 2 : 1);
        if (wm != nullptr) {
          gfloat_ptr lvl = wm + (L == 3 ? (size_t)fv.wmax_plane : (size_t)0);
          // (the 3 x 3 path indexes from `wm` itself with the plane folded into a 32-bit index: in the pre-pass the
          // view is uniform, so the load takes a scalar base and one vector offset instead of a 64-bit vector address;
          // images are at most 8192 x 8192 and there are four planes: < 2^28 elements.  Width and rows are below
          // 2^24: full-rate 24-bit multiplies.)
          const unsigned origin = __umul24((unsigned)v.width, (unsigned)ty0) + (unsigned)tx0 + (L == 3 ? (unsigned)fv.wmax_plane : 0u);
          if (nxw <= 3 && nyw <= 3) {
            // the usual case (footprints up to 24 pixels wide): as many lookups as the widest footprint among
            // the wave's views needs (uniform counts ux x uy, typically 2 x 2; narrower ones repeat their last
            // window), all requested before the first is used.  As a per-lane loop each load waited for the one
            // before; nine unconditional ones cost the memory system twice what is needed (measured at
            // 2048^3 x 64: 157 ms instead of 108).
            float t[9];
#pragma unroll
            for (int bq = 0; bq < 3; ++bq) {
              const unsigned ro = origin + __umul24((unsigned)v.width, (unsigned)min(bq << L, max(ph - k, 0)));
#pragma unroll
              for (int aq = 0; aq < 3; ++aq) {
                t[3 * bq + aq] = -INFINITY;
                if (aq < ux && bq < uy) t[3 * bq + aq] = load_u32_index(wm, ro + (unsigned)min(aq << L, max(pw - k, 0)));
              }
            }
#pragma unroll
            for (int q = 0; q < 9; ++q) m = fmaxf(m, t[q]);
          } else {
            for (int bq = 0; bq < nyw; ++bq) {
              gfloat_ptr row = lvl + (unsigned)v.width * (unsigned)(ty0 + min(bq << L, max(ph - k, 0)));
              for (int aq = 0; aq < nxw; ++aq) m = fmaxf(m, row[tx0 + min(aq << L, max(pw - k, 0))]);
            }
          }
        } else {  // no planes (out of memory for them): scan the rectangle
          gfloat_ptr img = (gfloat_ptr)v.sdf;
          for (int j = 0; j < ph; ++j) {
            gfloat_ptr row = img + ((unsigned)v.width * (unsigned)(ty0 + j) + (unsigned)tx0);
            for (int i = 0; i < pw; ++i) {
              const float t = row[i];
              has_nan |= !(fabsf(t) <= 3.402823466e+38f);  // NaN or +-inf: 0 * inf = NaN samples
              m = fmaxf(m, t);
            }
          }
        }
        // voxels projecting outside the ROI sample max_sdf instead (voxel_carver.cc:469-471)
        // (not in a `sure` tile: every sample of the brick lies inside it, hence inside the ROI)
        if (outside_max && !ti.sure) {
          has_nan |= !(fabsf(v.max_sdf) <= 3.402823466e+38f);
          m = fmaxf(m, v.max_sdf);
        }
        ti.ub = has_nan ? INFINITY : (__builtin_fmaf(fabsf(m), 0x1p-20f, m) + 1.0e-30f);
        // Lower bound of the samples, by the mirrored argument: with every tap >= mn the sample is
        // >= mn - 2^-22 |mn|.  If that is >= -1 no voxel of this tile is skipped by the truncation test
        // (`dist < -1`, voxel_carver.cc:478) and the test is compiled out of the run over it (sure bit 1).
        // Voxels outside the ROI are not an issue: a `sure` tile has none.
        // (not looked up for a tile the upper bound already drops: `ub < -1`, the view is never processed)
        if (want_lower && ti.sure && !(ti.ub < -1.0f) && wm != nullptr && fv.has_lower && nxw <= 3 && nyw <= 3) {
          const unsigned origin = __umul24((unsigned)v.width, (unsigned)ty0) + (unsigned)tx0 +
                                  (L == 3 ? 3u : 2u) * (unsigned)fv.wmax_plane;  // planes 2 / 3: of the negated image
          float t[9], mneg = -INFINITY;  // max of -g = -(min of g)
#pragma unroll
          for (int bq = 0; bq < 3; ++bq) {
            const unsigned ro = origin + __umul24((unsigned)v.width, (unsigned)min(bq << L, max(ph - k, 0)));
#pragma unroll
            for (int aq = 0; aq < 3; ++aq) {
              t[3 * bq + aq] = -INFINITY;
              if (aq < ux && bq < uy) t[3 * bq + aq] = load_u32_index(wm, ro + (unsigned)min(aq << L, max(pw - k, 0)));
            }
          }
#pragma unroll
          for (int q = 0; q < 9; ++q) mneg = fmaxf(mneg, t[q]);
          const float neg_lb = __builtin_fmaf(fabsf(mneg), 0x1p-20f, mneg);  // -(lower bound); +inf: none
          if (neg_lb <= 1.0f) ti.sure |= 2;
          if (lower_out) *lower_out = -neg_lb;  // (the slab planner, plan_cost_kernel)
        }
      }
    }
  }
  return ti;
}

// The whole FusedView record of view `vi` at once (ten 16-byte loads in flight, one wait): fields fetched where
// they are first needed cost a memory round trip each, behind every branch of footprint_of.
// (Pinned by the empty asm: the compiler would otherwise sink every load to its first use again.)
__device__ __forceinline__ FusedView load_fused_view(const FusedView* __restrict__ views, int vi) {
  static_assert(sizeof(FusedView) % 4 == 0, "FusedView is fetched dword by dword");
  constexpr int kViewDwords = (int)(sizeof(FusedView) / 4);
  typedef const uint32_t __attribute__((address_space(1))) * gu32_ptr;
  gu32_ptr src = (gu32_ptr)views + (size_t)vi * kViewDwords;
  uint32_t raw[kViewDwords];
#pragma unroll
  for (int q = 0; q < kViewDwords; ++q) raw[q] = src[q];
#pragma unroll
  for (int q = 0; q < kViewDwords; ++q) asm volatile("" : "+v"(raw[q]));
  FusedView fv;
  __builtin_memcpy(&fv, raw, sizeof(FusedView));
  return fv;
}

// (through an LDS-typed pointer: ds_write_b128, not flat stores)
__device__ __forceinline__ void store_tile_info(lds_u32* tinfo_lds, int vi, const TileInfo& ti) {
  static_assert(sizeof(TileInfo) % 4 == 0, "TileInfo is stored dword by dword");
  uint32_t w32[sizeof(TileInfo) / 4];
  __builtin_memcpy(w32, &ti, sizeof(TileInfo));
  lds_u32* dst = tinfo_lds + vi * (int)(sizeof(TileInfo) / 4);
#pragma unroll
  for (int q = 0; q < (int)(sizeof(TileInfo) / 4); ++q) dst[q] = w32[q];
}

// Prologue of the fused kernels that bound their footprints themselves (the big tile), out of line so that its
// registers do not add to the main loop's: lane vi handles view vi of the wave brick.
template <bool SAMEF, int TQ, bool GEN>
__device__ __attribute__((noinline)) float brick_footprints(const FusedView* __restrict__ views, int nviews, int lane,
                                                            float xl, float xh, float yl, float yh, float zl_, float zh,
                                                            bool is_ortho, bool outside_max, bool want_bound,
                                                            bool want_lower, lds_u32* tinfo_lds) {
  float ub_lane = INFINITY;
  if (lane < nviews) {
    const FusedView fv = load_fused_view(views, lane);
    const TileInfo ti = footprint_of<SAMEF, TQ, GEN>(fv, xl, xh, yl, yh, zl_, zh, is_ortho, outside_max, want_bound,
                                                     want_lower);
    store_tile_info(tinfo_lds, lane, ti);
    ub_lane = ti.ub;
  }
  return ub_lane;
}

// ---- footprint records (raw-tile kernels) ------------------------------------------------------
// What footprint_of finds for a (wave brick, view) pair does not depend on the voxel state, and inside the carve
// kernel it is the worst kind of work: one lane per view (half the wave idle at 32 views, 63 of 64 lanes for a
// single view), two dependent memory round trips before the wave can do anything else, and registers the run loops
// then have to live with.  The raw-tile kernels therefore take it from a pre-pass at full occupancy
// (footprint_records_kernel: one thread per pair, lane = brick along x, the view wave-uniform, so the view
// constants are scalar operands and the window lookups of neighbouring lanes fall into the same cache lines) that
// leaves 8 bytes per pair in memory, [view][brick]; the carve kernel's prologue is one 8-byte load per lane.
//   word 0: bits 31..6 upper bound of the samples (a float rounded UP to 26 bits: still a bound),
//           bits 3..0 th, bit 4 / 5: the tile ends at the ROI's last column / row (TileInfo::hi_x / hi_y)
//   word 1: bits 12..0 tx0, 25..13 ty0, 29..26 tw (0: no tile), 31..30 TileInfo::sure
// (raw tiles: tw, th <= 15; images up to 8192 x 8192: fused_eligible)
struct FootprintRecord {
  uint32_t w0, w1;
};

__device__ __forceinline__ FootprintRecord pack_footprint(const TileInfo& ti, const ViewParams& v) {
  uint32_t b = __float_as_uint(ti.ub);
  if (!(fabsf(ti.ub) <= 3.402823466e+38f)) b = 0x7f800000u;       // +inf / NaN: no bound
  else if (b & 0x80000000u) b &= ~63u;                            // negative: towards zero is up
  else b = (b + 63u) & ~63u;                                      // (may carry into +inf: no bound)
  FootprintRecord r;
  const int tx1 = ti.tx0 + ti.tw - 1, ty1 = ti.ty0 + ti.th - 1;
  r.w0 = b | (uint32_t)ti.th | (ti.nq && tx1 == v.roi_max_xi ? 16u : 0u) | (ti.nq && ty1 == v.roi_max_yi ? 32u : 0u);
  r.w1 = ti.nq ? ((uint32_t)ti.tx0 | ((uint32_t)ti.ty0 << 13) | ((uint32_t)ti.tw << 26) | ((uint32_t)ti.sure << 30)) : 0u;
  return r;
}

__device__ __forceinline__ TileInfo unpack_footprint(const FootprintRecord r) {
  TileInfo ti;
  const int tw = (int)((r.w1 >> 26) & 15u), th = (int)(r.w0 & 15u);
  const int tx0 = (int)(r.w1 & 8191u), ty0 = (int)((r.w1 >> 13) & 8191u);
  ti.ub = __uint_as_float(r.w0 & ~63u);
  ti.sure = (int)(r.w1 >> 30);
  ti.tx0 = tx0, ti.ty0 = ty0, ti.tw = tw, ti.th = (tw ? th : 0), ti.nq = tw * th;
  ti.pitchf = tw ? 16.0f : 0.0f;
  ti.inv_tw = tw ? 0.0625f : 1.0f;
  ti.base = tw ? -(ty0 * 16 + tx0) : 0;
  if (tw) {
    const int tx1 = tx0 + tw - 1, ty1 = ty0 + th - 1;
    ti.lo_x = (float)tx0, ti.lo_y = (float)ty0;
    // taps exist for floor(u) in [tx0, tx1]; at the ROI edge u == roi_max (== tx1) is still inside
    ti.hi_x = (r.w0 & 16u) ? (float)tx1 : __uint_as_float(__float_as_uint((float)(tx1 + 1)) - 1u);
    ti.hi_y = (r.w0 & 32u) ? (float)ty1 : __uint_as_float(__float_as_uint((float)(ty1 + 1)) - 1u);
  } else {
    ti.lo_x = ti.lo_y = INFINITY;  // nothing passes the tile test
    ti.hi_x = ti.hi_y = -INFINITY;
    ti.ub = INFINITY;
    ti.sure = 0;
  }
  return ti;
}

// Exact n / d for 32-bit unsigned n (Granlund-Montgomery, as in mc_kernels.hip): three integer instructions where the
// compiler's division by a run-time value takes about twenty.
struct FastDivU32 {
  uint32_t d, m, s1, s2;
};
__device__ __forceinline__ uint32_t fast_div_u32(uint32_t n, const FastDivU32& f) {
  const uint32_t t = __umulhi(n, f.m);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}
FastDivU32 make_fast_div_u32(uint32_t d) {
  FastDivU32 f;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;  // ceil(log2 d)
  f.d = d;
  f.m = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l < 1 ? 0 : l - 1;
  return f;
}

// Launch constants of carve_fused_kernel's block decode (which workgroup block of bricks a launch index is), computed on
// the host: a single-view launch runs 2 M waves that live a few microseconds, each CU has ONE scalar unit, and the five
// integer divisions by run-time values at the head of every wave -- 25 scalar instructions and a v_rcp_iflag round trip
// each -- were a fifth of the scalar work that bounds such a launch (profiles/r05/first_view_floor.txt).
struct BlockDecode {
  int total;                           // units (workgroup blocks; segments of the few-view flavour) of the launch
  int layer, q, rem, dealt;            // workgroups per brick layer, layer / 8, layer % 8, 8 q (layers of the launch)
  FastDivU32 dq, drem, dnbx, dnby;     // divisions by q, rem (1 when rem == 0: never used then), nbx, nby
};
BlockDecode make_block_decode(unsigned grid_x, int nbx, int nby) {
  BlockDecode d;
  d.total = (int)grid_x;
  d.layer = nbx * nby;
  d.q = d.layer >> 3;
  d.rem = d.layer & 7;
  d.dealt = 8 * d.q * (int)(grid_x / (unsigned)std::max(d.layer, 1));
  d.dq = make_fast_div_u32((uint32_t)std::max(d.q, 1));
  d.drem = make_fast_div_u32((uint32_t)std::max(d.rem, 1));
  d.dnbx = make_fast_div_u32((uint32_t)std::max(nbx, 1));
  d.dnby = make_fast_div_u32((uint32_t)std::max(nby, 1));
  return d;
}

// Pre-pass of the raw-tile carve kernels: blockIdx.y = view, thread = wave brick (linear, x fastest: the carve
// kernel's wave (bx, wave) of brick row (by, bz) is brick (bz * nby + by) * nbw + 4 bx + wave).
// VALU-bound (round 5: 441 vector instructions per pair, 0.41 of the 0.6 ms it takes at 1024^3 x 32 at two cycles each;
// profiles/r05/prepass.txt) -- hence the fast divisions of the brick number (div_nbw, div_nby: by nbw and nby, for launches of
// fewer than 2^32 bricks), the depth-range test on two values instead of eight, 24-bit multiplies and a scalar base
// for the window lookups in footprint_of.
template <bool SAMEF, bool GEN>
__global__ __launch_bounds__(256) void footprint_records_kernel(GridParams g, const FusedView* __restrict__ views,
                                                                int nbw, int nby, int64_t nbricks, ModeParams mode,
                                                                int want_bound, int want_lower,
                                                                FootprintRecord* __restrict__ records,
                                                                FastDivU32 div_nbw, FastDivU32 div_nby, int small32) {
  const int64_t brick = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (brick >= nbricks) return;
  const int vi = blockIdx.y;
  int bxw, by, bz;
  if (small32) {  // (uniform)
    const uint32_t b32 = (uint32_t)brick;
    const uint32_t rowb = fast_div_u32(b32, div_nbw);
    bxw = (int)(b32 - rowb * (uint32_t)nbw);
    const uint32_t zz = fast_div_u32(rowb, div_nby);
    by = (int)(rowb - zz * (uint32_t)nby);
    bz = (int)zz;
  } else {
    bxw = (int)(brick % nbw);
    const int64_t rowb = brick / nbw;
    by = (int)(rowb % nby);
    bz = (int)(rowb / nby);
  }
  const int x_lo = min(bxw * WX, g.nx - 1), x_hi = min(bxw * WX + WX - 1, g.nx - 1);
  const int y_hi = min(by * BY + BY - 1, g.ny - 1), z_hi = min(bz * BZ + BZ - 1, g.nz_local - 1);
  // (the view is uniform: its record arrives through scalar loads)
  const FusedView& fv = views[vi];
  gfloat_ptr ax = (gfloat_ptr)g.px, ay = (gfloat_ptr)g.py, az = (gfloat_ptr)g.pz;  // (uniform bases, 32-bit offsets)
  const TileInfo ti = footprint_of<SAMEF, kTileRaw, GEN>(
      fv, load_u32_index(ax, (unsigned)x_lo), load_u32_index(ax, (unsigned)x_hi), load_u32_index(ay, (unsigned)(by * BY)),
      load_u32_index(ay, (unsigned)y_hi), load_u32_index(az, (unsigned)(g.z0 + bz * BZ)),
      load_u32_index(az, (unsigned)(g.z0 + z_hi)), mode.ortho != 0, mode.outside == VCY_OUTSIDE_MAX, want_bound != 0,
      want_lower != 0);
  records[(int64_t)vi * nbricks + brick] = pack_footprint(ti, fv.v);
}

// Which workgroups of a carve launch have anything to do: a (wave brick, view) pair is dropped before the state is
// read when every sample is below the truncation limit or (kMax) not above the brick minimum the previous launch
// left (the early return of carve_fused_kernel, same test on the same records).  For a launch of few views over a
// carved grid -- the reference's `Carve(view); ExtractIsoSurface();` loop makes every view a launch of its own --
// most workgroups would only start, load 8 bytes and leave; 2 M such waves cost more than the bricks that do
// change.  This pass lists the workgroups with a live pair, list[0] = their number, list[1 ...] = their linear
// ids (any order); the carve kernel is launched over the full range and workgroups beyond list[0] leave at once.
// Entries WITH RECORDS (launches of one view, `entry_words` = kLiveEntryWords): list[0] = count, list[1] unused, then per
// listed workgroup {id, bit j = the brick of wave j is live, the kWgWaves footprint records} -- what a wave of the listed
// launch otherwise fetches AFTER it has learnt its workgroup id from the list: its record (a second round trip in a wave
// that lasts a handful) and the brick minimum for a test whose outcome is known here.
constexpr int kLiveThreads = 1024;
constexpr int kLiveEntryWords = 2 + 2 * kWgWaves;
__global__ __launch_bounds__(kLiveThreads) void live_workgroups_kernel(const FootprintRecord* __restrict__ recs, int64_t nbricks,
                                                                       int nviews, const float* __restrict__ bmin, int trunc,
                                                                       int nbx, int nby, int nbw, int nwg, int* __restrict__ list,
                                                                       int unit_bricks, int entry_words) {
  // (one atomic per block of 1024 workgroups: one per WAVE -- 8192 of them on one counter at 1024^3 -- took 78 us of a
  // 0.8 ms single-view launch, the serialised atomics, not the 25 MB it reads)
  __shared__ int wave_count[kLiveThreads / 64];
  __shared__ int block_base;
  const int wg = blockIdx.x * kLiveThreads + threadIdx.x;
  bool live = false;
  unsigned live_bricks = 0u;        // bit j: brick j of the unit has a live view
  FootprintRecord rec0[kWgWaves];   // (entries with records: view 0 of the workgroup's bricks)
  for (int j = 0; j < kWgWaves; ++j) rec0[j].w0 = 0u, rec0[j].w1 = 0u;
  if (wg < nwg) {
    const int bx = wg % nbx, r = wg / nbx;
    const int by = r % nby, bz = r / nby;
    constexpr int kUnitMax = kWgWaves > VCY_ROW_BRICKS ? kWgWaves : VCY_ROW_BRICKS;
#pragma unroll
    for (int j = 0; j < kUnitMax; ++j) {  // (kWgWaves bricks of a workgroup, or the segment of a wave: kRowBricks)
      if (j >= unit_bricks) break;
      const int bxw = bx * unit_bricks + j;
      if (bxw >= nbw) break;
      const int64_t brick = ((int64_t)bz * nby + by) * nbw + bxw;
      const float smin = bmin ? bmin[brick] : 0.0f;
      for (int v = 0; v < nviews; ++v) {
        const FootprintRecord rec = recs[(int64_t)v * nbricks + brick];
        const float ub = __uint_as_float(rec.w0 & ~63u);
        const bool drop = (trunc && ub < -1.0f) || (bmin != nullptr && ub <= smin);
        live = live || !drop;
        if (!drop) live_bricks |= 1u << j;
        if (v == 0 && j < kWgWaves) rec0[j] = rec;
      }
    }
  }
  const unsigned long long m = __ballot(live);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_count[wave] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int total = 0;
    for (int w = 0; w < kLiveThreads / 64; ++w) {
      const int c = wave_count[w];
      wave_count[w] = total;  // exclusive offsets of the waves
      total += c;
    }
    block_base = total ? atomicAdd(&list[0], total) : 0;
  }
  __syncthreads();
  if (live) {
    const int slot = block_base + wave_count[wave] + __popcll(m & ((1ull << lane) - 1ull));
    if (entry_words == 0) {
      list[1 + slot] = wg;
    } else {  // (8-byte aligned: the list is, and entry_words is even)
      int* e = list + 2 + (int64_t)slot * kLiveEntryWords;
      e[0] = wg, e[1] = (int)live_bricks;
      for (int j = 0; j < kWgWaves; ++j) e[2 + 2 * j] = (int)rec0[j].w0, e[3 + 2 * j] = (int)rec0[j].w1;
    }
  }
}

// GEN: nearest-neighbour sampling and/or an orthographic camera, selected at run time from `mode`
// (compiled out of the default bilinear + pinhole kernels, where the extra branches cost 16 %).
// NB: bricks per wave -- 1: a workgroup of kWgWaves waves, a brick each (fused launches of many views); kRowBricks: the
// few-view flavour described at kRowBricks above (raw tiles, records from the pre-pass, rows of whole bricks).
template <int UPDATE, bool CHECKMAX, int TQ, bool GEN, int NB>
constexpr int carve_waves_per_simd() {
  // (the few-view flavour is bounded by its LDS: 3 - 5 waves per SIMD, registers to spare)
  if (NB > 1) return NB > 2 ? 4 : 5;
  if (GEN || CHECKMAX || TQ != kTileRaw || UPDATE == VCY_UPDATE_WEIGHTED_AVERAGE) return VCY_WAVES_CHECKED;
  if (UPDATE == kUpdateWaUnitWeight) return NB == 0 ? VCY_WAVES_WA_ONE : VCY_WAVES_WA;
  return NB == 0 ? VCY_WAVES_ONE : VCY_WAVES;
}
template <typename CountT, int UPDATE, bool TRUNC, bool SAMEF, bool CHECKMAX, int TQ, bool GEN, int DIV, int NB = 1>
__global__ __launch_bounds__(64 * (NB > 1 ? kRowWaves : kWgWaves))
__attribute__((amdgpu_waves_per_eu(NB > 1 ? 1 : carve_waves_per_simd<UPDATE, CHECKMAX, TQ, GEN, NB>(),
                                   carve_waves_per_simd<UPDATE, CHECKMAX, TQ, GEN, NB>()))) void carve_fused_kernel(GridParams g,
                                                          const FusedView* __restrict__ views,
                                                          const float* __restrict__ c0_all,
                                                          int nviews_arg, ModeParams mode, int nbx,
                                                          int nby, BlockDecode bd, int cull_enabled, int state_flags,
                                                          const FootprintRecord* __restrict__ records,
                                                          int64_t nbricks, float* __restrict__ brick_min,
                                                          const int* __restrict__ wg_list,
                                                          unsigned long long* __restrict__ pair_count) {
  // brick_min[wave brick] (or null): min(sdf) over the brick as the carve kernels left it -- lowest() while a
  // voxel of it is untouched.  Written by every fused launch; READ (state_flags bit 2: every write to the state
  // since the slab was fresh went through a fused launch) to drop views before the state is loaded: a wave
  // whose every view is dropped returns without reading or writing anything, which is what makes the
  // reference's `Carve(); Extract(); Carve(); ...` pattern of single-view launches cheap.  Marching cubes skips
  // bricks that lie entirely outside the iso-surface with it (mc_bits).
  // state_flags: bit 0 = the slab is fresh (known sdf = lowest(), update_num = 0, never written);
  //              bit 1 = update_num == 0 implies sdf == lowest() (no vcy_upload since the fill)
  //              bit 3 = cooperative write-back through LDS (below)
  // NB == 0: a launch of ONE view with the NB = 1 structure (a brick per wave, cooperative write-back).  The reference's
  // own loop (examples.cc:117-149) makes every view such a launch; with the view count a compile-time 1 the footprint
  // record is a scalar load unpacked into registers (no TileInfo in LDS, no read-backs), and the view loop, its
  // next-view search, the second tile buffer's bookkeeping and the re-bounding after the view fold away.
  constexpr bool kOne = NB == 0;
#ifdef VCY_ONE_LDS  // development build: the one-view instance keeps its TileInfo in LDS like the general one
  constexpr bool kOneRegs = false;
#else
  constexpr bool kOneRegs = kOne;
#endif
  const int nviews = kOne ? 1 : nviews_arg;
  const int fresh = state_flags & 1;
  const bool implied = (state_flags & 2) != 0;
  const bool coop = NB <= 1 && (kWgWaves == 4 || kWgWaves == 8) && (state_flags & 8) != 0;
  const bool nt_store = (state_flags & 16) != 0;  // cooperative write-back with streaming stores
  constexpr bool kRows = NB > 1;  // the few-view flavour: this WAVE walks NB bricks of a row (kRowBricks)
  static_assert(!kRows || (TQ == kTileRaw && !CHECKMAX), "the few-view flavour: raw tiles, no update limit in reach");
  // dynamic LDS: [4 waves][TQ] quads, then [4 waves][nviews] TileInfo (sized by the launch), then the staging of the
  // cooperative write-back
  extern __shared__ float4 fused_lds[];
  constexpr bool kRaw = TQ == kTileRaw;                  // raw-pixel tiles, loaded straight into LDS
  // (NB == 0: ONE tile buffer -- there is no next view to fetch ahead -- and no TileInfo: 18.4 KB per workgroup with the
  // cooperative write-back's staging instead of 22.5, i.e. eight workgroups per CU where seven fit)
  constexpr int kTileF4 = NB == 0 ? 64 : tile_f4_per_wave<TQ>();
  // A view can be dropped for a whole wave brick when no voxel of the brick can change:
  //   - use_truncation and every sample is provably < -1 (voxel_carver.cc:478), or
  //   - kMax, every voxel already touched, and every sample is provably <= min(sdf) of the
  //     brick (UpdateVoxelMax only writes when dist > sdf, voxel_carver.cc:82).
  // "Provably": with every tap <= M and weights >= 0, monotonicity of IEEE rounding gives
  //   dist = fl(fl(fl(w00 s00 + w10 s10) + w01 s01) + w11 s11) <= the same expression with all taps = M,
  // and the four weights sum to 1 within 2^-23 (each is a product of u-floor(u), 1-(u-floor(u)) ...),
  // so dist <= M + 2^-22 |M| for either sign of M.  ub = M + 2^-20 |M| is that bound with slack.
  // Footprints holding a NaN or an infinity give no bound (0 * inf = NaN samples).
  // (Round 6 also bounded a `sure` view by the EXACT maximum of its staged tile -- one 16-byte LDS read per lane and a wave
  // reduction once the tile has landed -- against the window maxima's over-estimate: on the benchmark scenes it never
  // dropped a single pair more and cost 8 % (profiles/r06/exact_tile.txt).  The pairs that are processed without changing
  // anything are not lost to the windows sticking out of the footprint: a distance field varies by 1 - 2 % across a
  // footprint, and so does the brick's state; what is compared is the MAXIMUM of the one with the MINIMUM of the other.)
  constexpr bool kNeedBound = TRUNC || UPDATE == VCY_UPDATE_MAX;

  VCY_SETPRIO(3);
  const int tid = threadIdx.x;
  // (the wave index is uniform, which the compiler cannot see: keeps the LDS bases of the wave in SGPRs)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  VCY_PT_DECL;
  // NB > 1: every wave has its own region [tiles | TileInfo of the NB x 8 pairs | state of NB bricks | NB masks]
  constexpr size_t kRowWaveBytes = kRows ? row_lds_bytes_per_wave<CountT, NB>() : 0;
  static_assert(kRowWaveBytes % 16 == 0, "wave regions are 16-byte aligned");
  float4* tile = kRows ? (float4*)((char*)fused_lds + wave * kRowWaveBytes) : fused_lds + wave * kTileF4;
  TileInfo* tinfo = kRows ? (TileInfo*)((char*)tile + kRawBuffers * 1024)
                          : (TileInfo*)(fused_lds + kWgWaves * kTileF4) + wave * nviews;
  float* stage_s = (float*)((char*)tinfo + NB * kRowMaxViews * sizeof(TileInfo));   // [NB][64 rows][WX]
  CountT* stage_n = (CountT*)(stage_s + NB * 64 * WX);                                 // [NB][64 rows][WX]
  typedef unsigned long long __attribute__((address_space(3))) lds_u64_row;
  lds_u64_row* stage_mask = (lds_u64_row*)(unsigned long long*)(stage_n + NB * 64 * WX);  // [NB] changed lanes
  // (sizeof(TileInfo) * kWgWaves is a multiple of 16: the staging area is 16-byte aligned)
  static_assert((sizeof(TileInfo) * kWgWaves) % 16 == 0, "alignment of the cooperative write-back's staging");
  typedef CountT CountVec8 __attribute__((ext_vector_type(WX)));
  typedef CountVec8 __attribute__((address_space(3))) lds_countvec;
  typedef unsigned long long __attribute__((address_space(3))) lds_u64;
  float* coop_s = NB == 0 ? (float*)(fused_lds + kWgWaves * kTileF4)
                          : (float*)((TileInfo*)(fused_lds + kWgWaves * kTileF4) + kWgWaves * nviews);
  CountT* coop_n = (CountT*)(coop_s + 64 * kCoopSdfPitch);
  lds_u64* coop_mask = (lds_u64*)(unsigned long long*)(coop_n + 64 * coop_cnt_pitch<CountT>());
  // A wave that leaves early tells the others that none of its rows is to be stored and that it will not be there to
  // store rows of theirs (s_barrier only waits for the waves of the workgroup that have not ended; the LDS writes have
  // completed before the wave ends).
  auto coop_leave = [&]() {
    if (coop) {
      if (lane == 0) coop_mask[wave] = 0ull, coop_mask[kWgWaves + wave] = 0ull;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
  };
  TileInfo ti_one;  // NB == 0: the one view's tile, in registers
  auto tile_of = [&](int p) -> const TileInfo& {
    if constexpr (kOneRegs) return ti_one;
    else return tinfo[p];
  };
  int cur = 0;  // raw tiles: which of the wave's buffers holds the view being carved
  auto raw_buf = [&](int b) -> float* { return (float*)tile + 256 * b; };
  const int ly = lane & (BY - 1), lz = lane >> 3;
  // XCD-aware order: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs, so
  // workgroup b runs on XCD b % 8.  Give every XCD one contiguous eighth of the brick list: bricks
  // that follow each other on an XCD are neighbours in x and share SDF footprint pixels and
  // z-table entries in that XCD's private L2.
  // NB > 1: the unit of the launch is a SEGMENT (NB bricks of a row) and every wave takes one -- the waves of a
  // workgroup consecutive units of the same XCD's share (b mod 8 = blockIdx mod 8, the XCD the workgroup runs on)
  int b = kRows ? ((int)(blockIdx.x & 7u) + 8 * (kRowWaves * (int)(blockIdx.x >> 3) + wave)) : (int)blockIdx.x;
  const int* list_entry = nullptr;  // NB == 0, listed launch whose entries hold {id, live waves, records}: this workgroup's
  int list_live = 0;
  FootprintRecord list_rec;
  list_rec.w0 = 0u, list_rec.w1 = 0u;
  if (wg_list != nullptr) {  // only the workgroups live_workgroups_kernel listed (wg_list[0] of them)
    if (kRows) b = (int)blockIdx.x * kRowWaves + wave;
    if (b >= wg_list[0]) return;
    if (kOne && (state_flags & 64) != 0) {  // entries with records (live_workgroups_kernel)
      // (VECTOR loads that all lanes share, made uniform afterwards: 40 bytes per workgroup streamed through the scalar
      // cache evict the view record and the axis tables that every wave re-reads -- measured, like the records before)
      list_entry = wg_list + 2 + (int64_t)b * kLiveEntryWords;
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      typedef const u32x2 __attribute__((address_space(1))) * gent_ptr;
      const u32x2 head = ((gent_ptr)list_entry)[0], mine = ((gent_ptr)list_entry)[1 + wave];
      b = __builtin_amdgcn_readfirstlane((int)head.x);
      list_live = __builtin_amdgcn_readfirstlane((int)head.y);
      list_rec.w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mine.x);
      list_rec.w1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mine.y);
    } else {
      b = wg_list[1 + b];
    }
  } else {
    if (kRows && b >= bd.total) return;
#if !defined(VCY_XCD_LAYERS) && !defined(VCY_XCD_CONTIGUOUS)
    // Every XCD takes an eighth of EVERY brick layer -- q = layer / 8 consecutive workgroups, i.e. whole rows in (y, x)
    // order -- and a different eighth in every layer (chunk (xcd + layer) mod 8), so that each XCD sees every z and,
    // over 8 layers, every y range: balanced for a slab of few layers too (a rank's slab of an 8-GPU run has 16, and
    // dealing whole layers gives XCD 7 the two most expensive ones of an outer slab).  The workgroups a layer has
    // beyond a multiple of eight go round-robin as they come.
    // (layer, q, rem, dealt and the divisions by q and rem: launch constants from the host, BlockDecode)
    const int layer = bd.layer, q = bd.q, rem = bd.rem, dealt = bd.dealt;
    if (b < dealt) {
      const int xcd = b & 7, j = b >> 3;
      const int l = (int)fast_div_u32((uint32_t)j, bd.dq), within = j - l * q;
      b = l * layer + ((xcd + l) & 7) * q + within;
    } else {
      const int r = b - dealt, l = (int)fast_div_u32((uint32_t)r, bd.drem);
      b = l * layer + 8 * q + (r - l * rem);
    }
#elif defined(VCY_XCD_LAYERS)
    // (round 3's order, kept for A/B runs)  ... and those eighths must cost the same.  A contiguous eighth of the brick list is a z-slab, and with view
    // dropping the slabs through the object cost 1.6x the outer ones: six XCDs would wait for two.  So every XCD
    // gets whole brick LAYERS, dealt cyclically -- layers xcd, xcd + 8, xcd + 16 ... -- and walks each of them in
    // (y, x) order: neighbours in x and y still share footprint pixels in that XCD's L2, every XCD sees the same mix
    // of empty and busy regions.  (The layers beyond a multiple of eight go round-robin as they come.)
    const int layer = nbx * nby, nlayers = (int)gridDim.x / layer, full = (nlayers >> 3) * layer;
    if (b < full * 8) {
      const int xcd = b & 7, j = b >> 3;
      b = ((j / layer) * 8 + xcd) * layer + j % layer;
    }
#else
    const int nb = gridDim.x, per = nb >> 3;
    if (b < per * 8) b = (b & 7) * per + (b >> 3);
#endif
  }
  const int brow = (int)fast_div_u32((uint32_t)b, bd.dnbx);  // (b >= 0: a launch covers fewer than 2^31 workgroups)
  const int bx = b - brow * nbx;
  const int bz = (int)fast_div_u32((uint32_t)brow, bd.dnby);
  const int by = brow - bz * nby;
  // (the wave index is uniform, which the compiler cannot see: readfirstlane keeps the x tables in scalar loads)
  // (NB > 1: the origin of the segment's first brick; moves on with the brick being carved)
  int x_first = kRows ? bx * (NB * WX) : __builtin_amdgcn_readfirstlane(bx * BX + wave * WX);  // wave brick origin
#if defined(VCY_DEV_EXIT_AT) && VCY_DEV_EXIT_AT == 1
  if (x_first >= 0) {
    coop_leave();
    return;
  }
#endif
  if (x_first >= g.nx) {                    // (a wave may leave alone: see coop_leave)
    coop_leave();
    return;
  }
  if constexpr (kOne) {
    // (the early return below, decided by the list pass on the same record and the same brick minimum)
    if (list_entry != nullptr && ((list_live >> wave) & 1) == 0) {
      coop_leave();
      return;
    }
  }
  const int zl0 = bz * BZ;
  const int y_raw = by * BY + ly, zl_raw = zl0 + lz;
  const int y = min(y_raw, g.ny - 1), zl = min(zl_raw, g.nz_local - 1);  // clones for out-of-grid lanes
  const float py = g.py[y], pz = g.pz[g.z0 + zl];
  // Lane (y, z) walks the WX voxels of its x run: in pc = t + (c0 + (c1 + c2)) (reference association) the
  // inner sum c1 + c2 = R[:,1] y + R[:,2] z is the same for the whole run and computed once per view.
  const int nxp = (g.nx + WX - 1) & ~(WX - 1);
  const bool want_bound = kNeedBound && cull_enabled;

  // This wave brick's index in the launch (fewer than 2^31: launch_carve_fused).  Computed HERE, in uniform control
  // flow: a uniform value first computed inside a divergent branch (`if (lane < nviews)` below) reaches later uses
  // through a phi that the compiler must treat as divergent -- it then lives in a VGPR, and so did the address of the
  // c0 records that shares `nxp / WX` with it: the scalar loads of the run loops had become vector loads (-15 %).
  int brick_lin = (bz * nby + by) * (nxp / WX) + (x_first / WX);
  const int x_seg = x_first, brick_seg = brick_lin;  // (NB > 1: the segment's first brick)
  // "Eager" launches (state_flags bit 5; launch_carve_fused sets it for few-view launches whose workgroups are nearly all
  // live -- listed ones, or a weighted-average view that changes nearly every brick): the brick's state is requested HERE,
  // next to the footprint record, instead of behind the early-return test that needs the record first -- one memory round
  // trip less in the life of a wave that consists of little else.  (A wave that then returns early has read 2.5 KB for
  // nothing; y and zl are clamped and x_first < nx, so the addresses are inside the slab.)
  // Only in the instance compiled for ONE view: in the general one the ten registers, live across the prologue, put
  // lane spills into the run loops of the 32-view launch that never takes this path.
  typedef CountT CountVecE __attribute__((ext_vector_type(WX)));
  f4 eager_a = f4{0.f, 0.f, 0.f, 0.f}, eager_b = eager_a;
  CountVecE eager_c = CountVecE{};
  const bool eager = kOne && (state_flags & 32) != 0 && (state_flags & 1) == 0 && (g.nx & (WX - 1)) == 0;
  if constexpr (kOne) {
    if (eager) {
      const int64_t row_e = ((int64_t)zl * g.ny + y) * g.nx + x_first;
      eager_a = *(const f4*)(g.sdf + row_e);
      eager_b = *(const f4*)(g.sdf + row_e + 4);
      eager_c = *(const CountVecE*)((const CountT*)g.cnt + row_e);
    }
  }
  // ---- prologue: lane vi bounds the footprint of the wave brick in view vi (brick_footprints) -------
  float ub_lane;
#ifdef VCY_PHASE_TIMING
  {
    asm volatile("" ::"v"(py), "v"(pz));  // the axis tables have arrived
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();
    pt_acc[12] += t_ - pt_last;
  }
#endif
  // NB > 1: lane 8 j + v holds the pair (brick j of the segment, view v)
  const int pair_j = lane >> 3, pair_v = lane & 7;
  const bool pair_valid = kRows && pair_j < NB && pair_v < nviews && x_seg + pair_j * WX < g.nx;
  if constexpr (kRows) {
    ub_lane = INFINITY;
    if (pair_valid) {
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      typedef const u32x2 __attribute__((address_space(1))) * grec_ptr;
      const u32x2 raw = ((grec_ptr)records)[(int64_t)pair_v * nbricks + (brick_seg + pair_j)];
      FootprintRecord rec;
      rec.w0 = raw.x, rec.w1 = raw.y;
      const TileInfo ti = unpack_footprint(rec);
      store_tile_info((lds_u32*)tinfo, lane, ti);
      ub_lane = ti.ub;
    }
  } else if constexpr (kOneRegs) {
    // one view: the record of (view 0, this brick) is wave-uniform.  Fetched with a VECTOR load all lanes share and made
    // uniform afterwards: the records are streamed once, and as scalar loads they evicted the view record and the x
    // tables -- which every wave re-reads -- from the scalar cache (2.72 -> 3.38 ms per weighted-average view).
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef const u32x2 __attribute__((address_space(1))) * grec_ptr;
    FootprintRecord rec;
    if (list_entry != nullptr) {  // (arrived with the workgroup id)
      rec = list_rec;
    } else {
      const u32x2 raw = ((grec_ptr)records)[brick_lin];
      rec.w0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)raw.x), rec.w1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)raw.y);
    }
    ti_one = unpack_footprint(rec);
    ub_lane = ti_one.ub;
  } else if (kRaw && records != nullptr) {
    // raw tiles: the footprints come from the pre-pass (footprint_records_kernel), 8 bytes per view
    ub_lane = INFINITY;
    if (lane < nviews) {
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      typedef const u32x2 __attribute__((address_space(1))) * grec_ptr;
      const u32x2 raw = ((grec_ptr)records)[(int64_t)lane * nbricks + brick_lin];
      FootprintRecord rec;
      rec.w0 = raw.x, rec.w1 = raw.y;
      const TileInfo ti = unpack_footprint(rec);
      store_tile_info((lds_u32*)tinfo, lane, ti);
      ub_lane = ti.ub;
    }
  } else {
    // the big tile -- and raw tiles of a launch whose records would not fit (`records` null: 2048^3 x 64 views would
    // write and read back 8.6 GB of them in nine chunks; with 64 views every lane of this prologue has a view)
    const int x_lo = min(x_first, g.nx - 1), x_hi = min(x_first + WX - 1, g.nx - 1);
    const int y_hi = min(by * BY + BY - 1, g.ny - 1);
    const int z_hi = min(zl0 + BZ - 1, g.nz_local - 1);
    ub_lane = brick_footprints<SAMEF, TQ, GEN>(views, nviews, lane, g.px[x_lo], g.px[x_hi], g.py[by * BY], g.py[y_hi],
                                               g.pz[g.z0 + zl0], g.pz[g.z0 + z_hi], mode.ortho != 0,
                                               mode.outside == VCY_OUTSIDE_MAX, want_bound,
                                               want_bound && TRUNC && UPDATE != VCY_UPDATE_MAX, (lds_u32*)tinfo);
  }
  wave_lds_fence();
#if defined(VCY_DEV_EXIT_AT) && VCY_DEV_EXIT_AT == 2  // development build: where a wave's scalar instructions go (profiles/tools/salu_attribution.sh)
  {
    coop_leave();
    return;
  }
#endif
  const unsigned long long view_mask = kRows ? __ballot(pair_valid) : ((nviews >= 64) ? ~0ull : ((1ull << nviews) - 1ull));
  unsigned long long live = view_mask;  // pairs / views that may still change something (NB > 1: set here, from the kept minima)
  // (a launch covers fewer than 2^31 wave bricks: launch_carve_fused)
  // Views that cannot change this brick whatever its voxels hold now: every sample below the truncation limit, or
  // (kMax) not above the brick's minimum as the previous launch left it.  All of them: nothing to read or write.
#ifndef VCY_NO_EARLY_EXIT
  if (want_bound && !fresh && !(kOne && list_entry != nullptr)) {
    const bool have_min = UPDATE == VCY_UPDATE_MAX && (state_flags & 4) != 0 && brick_min != nullptr;
    if (TRUNC || have_min) {
      bool drop0 = TRUNC && ub_lane < -1.0f;
      if (have_min) {
        float smin0;
        if constexpr (kRows) smin0 = pair_valid ? brick_min[brick_seg + pair_j] : 0.0f;  // (this lane's brick)
        else smin0 = ((cfloat_ptr)brick_min)[brick_lin];  // (uniform: a scalar load)
        drop0 = drop0 || ub_lane <= smin0;  // (a brick with an untouched voxel holds lowest(): never true)
      }
      live = __ballot(!drop0) & view_mask;
      if (live == 0ull) {
        coop_leave();
        return;
      }
    }
  }
#endif
#ifdef VCY_PHASE_TIMING
  {
    asm volatile("" ::"v"(ub_lane));
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();
    pt_acc[13] += t_ - pt_last;  // (includes slot 12)
  }
#endif

#if defined(VCY_DEV_EXIT_AT) && VCY_DEV_EXIT_AT == 3  // development build: where a wave's scalar instructions go (profiles/tools/salu_attribution.sh)
  {
    coop_leave();
    return;
  }
#endif
  // ---- load the wave brick's state ----------------------------------------------------------
  CountT* __restrict__ cnt = (CountT*)g.cnt;
  // update_num in registers: an int for kMax, a float for the weighted-average modes (see apply_sample)
  constexpr bool kFloatCount = UPDATE != VCY_UPDATE_MAX;
  typedef typename std::conditional<kFloatCount, float, int>::type NT;
  float s[WX];
  NT n[WX];
  const int64_t row0 = ((int64_t)zl * g.ny + y) * g.nx;  // this lane's row; voxel k is at row0 + min(x_first + k, nx - 1)
  // The first view this brick will process is usually known BEFORE its state is: it is the first view the bounds do not
  // drop, and what the bounds are compared with -- the truncation limit, the brick minimum the previous launch left --
  // is already here.  Its tile is then requested right behind the state instead of after the state has arrived and been
  // looked at: one memory round trip less in a wave's chain, which is most of what a launch of ONE view consists of.
  // (kMax without valid minima: not known, vi_pre stays -1.  live_views() below decides as before; the request is
  // repeated there if it names another view -- it never does -- and loads complete in order, so the later one wins.)
  int vi_pre = -1;
  auto prefetch_first_tile = [&]() {
    if constexpr (kRaw) {
      const bool listed_live = kOne && list_entry != nullptr;  // (the list pass has decided: the one view is live)
      const bool have_min = UPDATE == VCY_UPDATE_MAX && !fresh && (state_flags & 4) != 0 && brick_min != nullptr;
      if (!(fresh || UPDATE != VCY_UPDATE_MAX || !want_bound || have_min || listed_live)) return;
      bool drop = false;
      if (want_bound && !listed_live) {
        if (TRUNC) drop = ub_lane < -1.0f;
        if (have_min) {
          const float smin0 = ((cfloat_ptr)brick_min)[brick_lin];
          // (lowest(): a voxel of the brick is untouched -- all_touched will be false and nothing is dropped by this rule)
          if (smin0 != kInvalidSdf) drop = drop || ub_lane <= smin0;
        }
      }
      const unsigned long long lp = __ballot(!drop) & view_mask;
      vi_pre = lp ? (__ffsll((long long)lp) - 1) : nviews;
      if (vi_pre < nviews) raw_prefetch(views[vi_pre].v, tile_of(vi_pre), lane, raw_buf(0));
    }
  };
  // rows are whole bricks when nx % 8 == 0: the run is one 32-byte (sdf) and one 8/16-byte (update_num) vector
  const bool vec_io = (g.nx & (WX - 1)) == 0;
  typedef CountT CountVec __attribute__((ext_vector_type(WX)));
  if constexpr (kRows) {
    // The state of every live brick of the segment, requested NOW with LDS-direct loads into the wave's staging area
    // (row = the carving lane that owns it, 8 voxels per row): no registers, no waits -- the first tile wait below covers
    // them (loads complete in order).  An sdf request r is one z slice of a brick: lane L -> dword L & 7 of row
    // 8 r + (L >> 3), i.e. eight 32-byte row pieces; the pieces of the NB bricks of a row are requested back to back, so
    // the memory system sees the row's 128 contiguous bytes together.  Counters: 8 (u8) or 16 (u16) bytes per row.
#pragma unroll
    for (int k = 0; k < WX; ++k) {
      s[k] = kInvalidSdf;
      n[k] = (NT)0;
    }
    if (lane < NB) stage_mask[lane] = 0ull;  // (a brick that is never begun is neither read nor stored)
    if (!fresh) {
      const int yl = min(by * BY + (lane >> 3), g.ny - 1);
      const unsigned off_s = (unsigned)yl * (unsigned)g.nx + (unsigned)(lane & 7);   // (floats; + slice base + brick origin)
      constexpr int kCntPerDword = 4 / (int)sizeof(CountT);            // counters per dword: 4 (u8) or 2 (u16)
      constexpr int kCntDwordsPerRow = WX / kCntPerDword;              // 2 or 4
      constexpr int kCntRowsPerReq = 64 / kCntDwordsPerRow;            // 32 or 16 rows per request
      constexpr int kCntReqs = 64 / kCntRowsPerReq;                    // 2 or 4 requests per brick
      const int crow = lane / kCntDwordsPerRow;                        // row within a request
      typedef const CountT __attribute__((address_space(1))) * gcnt_ptr;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (((live >> (8 * j)) & 0xffull) == 0ull) continue;  // (uniform) no live view: neither read nor written
        const int xb = x_seg + j * WX;
#pragma unroll
        for (int r = 0; r < BZ; ++r) {
          const int zr = min(zl0 + r, g.nz_local - 1);
          gfloat_ptr src = (gfloat_ptr)g.sdf + ((int64_t)zr * g.ny * g.nx + xb);
          __builtin_amdgcn_global_load_lds(src + off_s, (lds_float*)(stage_s + (j * 64 + 8 * r) * WX), 4, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < kCntReqs; ++r) {
          const int row = r * kCntRowsPerReq + crow;  // = ly | lz << 3 of the lane that carves it
          const int yr = min(by * BY + (row & 7), g.ny - 1), zr = min(zl0 + (row >> 3), g.nz_local - 1);
          gcnt_ptr src = (gcnt_ptr)cnt + (((int64_t)zr * g.ny + yr) * g.nx + xb + (lane % kCntDwordsPerRow) * kCntPerDword);
          __builtin_amdgcn_global_load_lds((const uint32_t __attribute__((address_space(1)))*)src,
                                           (lds_u32*)(uint32_t*)(stage_n + (j * 64 + r * kCntRowsPerReq) * WX), 4, 0, 0);
        }
      }
    }
  } else if (fresh) {  // a fresh slab is known to be untouched everywhere: nothing to read
#pragma unroll
    for (int k = 0; k < WX; ++k) {
      s[k] = kInvalidSdf;
      n[k] = (NT)0;
    }
  } else if (vec_io) {
    // (streaming LOADS of the state were measured too: 2.7 -> 5.3 ms per view, profiles/r06/nontemporal.txt)
    float4 a, b4;
    CountVec cv;
    if (kOne && eager) {  // (uniform) requested before the footprint record was looked at
      a = make_float4(eager_a.x, eager_a.y, eager_a.z, eager_a.w), b4 = make_float4(eager_b.x, eager_b.y, eager_b.z, eager_b.w);
      cv = eager_c;
    } else {
      a = *(const float4*)(g.sdf + row0 + x_first), b4 = *(const float4*)(g.sdf + row0 + x_first + 4);
      cv = *(const CountVec*)(cnt + row0 + x_first);
    }
    prefetch_first_tile();  // (behind the state's requests, in front of their first use)
    s[0] = a.x, s[1] = a.y, s[2] = a.z, s[3] = a.w, s[4] = b4.x, s[5] = b4.y, s[6] = b4.z, s[7] = b4.w;
#pragma unroll
    for (int k = 0; k < WX; ++k) n[k] = (NT)cv[k];
  } else {
#pragma unroll
    for (int k = 0; k < WX; ++k) {
      const int xk = min(x_first + k, g.nx - 1);
      s[k] = g.sdf[row0 + xk];
      n[k] = (NT)cnt[row0 + xk];
    }
  }

  // Every voxel of the brick touched (update_num >= 1)?  Wave-uniform; update_num never decreases, so
  // once true it stays true.  Selects the select-free update (update_max_touched) and one of the two
  // view-dropping rules.
  bool all_touched = false;
  auto refresh_all_touched = [&]() {
    if (UPDATE != VCY_UPDATE_MAX || all_touched) return;
    NT nmin = n[0];
#pragma unroll
    for (int k = 1; k < WX; ++k) nmin = min(nmin, n[k]);
    all_touched = __all(nmin >= (NT)1);
  };
  if (!fresh && !kRows) refresh_all_touched();
  // No voxel of the brick touched yet?  (Wave-uniform; true for every brick of a fresh slab.)  The first `sure`
  // view of such a brick is a plain store of the samples (carve_view_fast<FIRST>).
  bool none_touched = fresh != 0;
  auto refresh_none_touched = [&]() {
    NT nmax = n[0];
#pragma unroll
    for (int k = 1; k < WX; ++k) nmax = max(nmax, n[k]);
    none_touched = __all(nmax < (NT)1);
  };
  if (!kRows && !fresh && UPDATE == VCY_UPDATE_MAX && !all_touched) refresh_none_touched();
  // Weighted average: does every voxel of the brick carry the same update_num?  (Wave-uniform; true for a
  // fresh slab, and it stays true while every processed view updates every voxel -- the views whose tile
  // provably holds no sample below -1.)  Then the weights of the average, fn and 1 / (fn + 1), are the same
  // for the whole brick and are formed once per view instead of once per sample (carve_view_fast<UNIFORM>);
  // n[] is only brought up to date when the brick leaves this state, and at the write-back.
  bool uniform_cnt = false;
  float fnu = 0.0f;  // the common update_num (as a float, like n[])
  auto refresh_uniform_cnt = [&]() {
    if (UPDATE == VCY_UPDATE_MAX) return;
    if (fresh) {
      uniform_cnt = true;
      fnu = 0.0f;
    } else {
      const float f0 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint((float)n[0])));
      bool same = true;
#pragma unroll
      for (int k = 0; k < WX; ++k) same = same && (float)n[k] == f0;
      uniform_cnt = __all(same);
      fnu = f0;
    }
  };
  if (!kRows) refresh_uniform_cnt();
  auto leave_uniform = [&]() {
    if (!uniform_cnt) return;
    uniform_cnt = false;
#pragma unroll
    for (int k = 0; k < WX; ++k) n[k] = (NT)fnu;
  };
  // views that may still change something, as a wave-uniform bit mask
  int jc = -1;  // NB > 1: the brick of the segment whose state is in registers
  auto live_views = [&]() -> unsigned long long {
    bool drop = false;
    if (want_bound) {
      if (TRUNC) drop = ub_lane < -1.0f;
      if (UPDATE == VCY_UPDATE_MAX && all_touched) {
        float m = s[0];
#pragma unroll
        for (int k = 1; k < WX; ++k) m = fminf(m, s[k]);
        const float smin = wave_min(m);
        drop = drop || ub_lane <= smin;
      }
    }
    if constexpr (kRows) {  // (the bounds of the other bricks' pairs stand as they are)
      const unsigned long long cur_bits = 0xffull << (8 * jc);
      return (live & ~cur_bits) | (__ballot(!drop) & view_mask & cur_bits);
    }
    return __ballot(!drop) & view_mask;
  };
  const int vi_end = kRows ? 64 : nviews;  // "no further view / pair"
  auto next_view = [&](unsigned long long live, int after) -> int {
    const unsigned long long rest = (after >= 63) ? 0ull : (live & ~((2ull << after) - 1ull));
    return rest ? (__ffsll((long long)rest) - 1) : vi_end;
  };

  // Lanes whose voxels changed (update_num grows with every change), accumulated over the views: what the write-back
  // stores.  (Round 3 re-read update_num from memory and compared: a dependent round trip in every wave's chain.  It
  // turned out not to be what bounds a single-view launch -- see DESIGN section 8 -- but there is no reason to keep it.)
  unsigned long long changed_lanes = 0ull;
  if (!kRows) live = live_views();
  int vi = live ? (__ffsll((long long)live) - 1) : vi_end;
  if (kRaw && vi < vi_end && vi != vi_pre) raw_prefetch(views[kRows ? (vi & 7) : vi].v, tile_of(vi), lane, raw_buf(0));
  // NB > 1: the brick whose turn it is takes its state from the staging area (the LDS-direct requests above have
  // landed once the wave has waited for its first tile) and leaves it there again when the next brick begins
  typedef CountT CountVecR __attribute__((ext_vector_type(WX)));
  typedef CountVecR __attribute__((address_space(3))) lds_countvec_r;
  auto finish_brick = [&]() {
    if constexpr (kRows) {
      leave_uniform();
      if (brick_min != nullptr && implied) {
        float m = s[0];
#pragma unroll
        for (int k = 1; k < WX; ++k) m = fminf(m, s[k]);
        const float smin = wave_min(m);
        if (lane == 0) brick_min[brick_lin] = smin;
      }
      lds_float4* rs = (lds_float4*)(float4*)(stage_s + (jc * 64 + lane) * WX);
      rs[0] = f4{s[0], s[1], s[2], s[3]};
      rs[1] = f4{s[4], s[5], s[6], s[7]};
      CountVecR cv;
#pragma unroll
      for (int k = 0; k < WX; ++k) cv[k] = (CountT)n[k];
      *(lds_countvec_r*)(CountVecR*)(stage_n + (jc * 64 + lane) * WX) = cv;
      if (lane == 0) stage_mask[jc] = fresh ? ~0ull : changed_lanes;
    }
  };
  auto begin_brick = [&](int j) {
    if constexpr (kRows) {
      jc = j;
      x_first = x_seg + j * WX;
      brick_lin = brick_seg + j;
      changed_lanes = 0ull;
      if (fresh) {
#pragma unroll
        for (int k = 0; k < WX; ++k) {
          s[k] = kInvalidSdf;
          n[k] = (NT)0;
        }
      } else {
        const lds_float4* rs = (const lds_float4*)(float4*)(stage_s + (j * 64 + lane) * WX);
        const f4 a = rs[0], b4 = rs[1];
        const CountVecR cv = *(const lds_countvec_r*)(CountVecR*)(stage_n + (j * 64 + lane) * WX);
        s[0] = a.x, s[1] = a.y, s[2] = a.z, s[3] = a.w, s[4] = b4.x, s[5] = b4.y, s[6] = b4.z, s[7] = b4.w;
#pragma unroll
        for (int k = 0; k < WX; ++k) n[k] = (NT)cv[k];
      }
      all_touched = false;
      none_touched = fresh != 0;
      if (!fresh) {
        refresh_all_touched();
        if (UPDATE == VCY_UPDATE_MAX && !all_touched) refresh_none_touched();
      }
      refresh_uniform_cnt();
    }
  };
  VCY_PT(0);
  VCY_PT_COUNT(10);

#if defined(VCY_DEV_EXIT_AT) && VCY_DEV_EXIT_AT == 4  // development build: where a wave's scalar instructions go (profiles/tools/salu_attribution.sh)
  {
    coop_leave();
    return;
  }
#endif
  // ---- views ------------------------------------------------------------------------------
  int n_processed = 0;  // (wave-uniform: an SGPR; only read with "paircount" on)
  while (vi < vi_end) {
    if constexpr (kRows) {
      if ((vi >> 3) != jc) {  // (uniform) the next pair belongs to another brick of the segment
        raw_tile_wait();      // everything requested so far has landed: the state of every brick, this pair's tile
        wave_lds_fence();
        if (jc >= 0) finish_brick();
        begin_brick(vi >> 3);
        // the bounds of this brick's pairs against its state as it really is (the kept minima may be invalid or absent)
        live = live_views();
        if (((live >> vi) & 1ull) == 0ull) {
          vi = next_view(live, vi);
          // (the dropped pair's pixels may still be arriving in that buffer: loads complete in order)
          if (vi < vi_end) raw_prefetch(views[vi & 7].v, tile_of(vi), lane, raw_buf(cur));
          continue;
        }
      }
    }
    const int vv = kRows ? (vi & 7) : vi;  // the view of pair vi
    const ViewParams& v = views[vv].v;
    // this view's record of the wave brick's x products: (x, y) pairs at [2 k], z at [16 + k]
    cfloat_ptr c0 = (cfloat_ptr)(c0_all + ((size_t)vv * (nxp / WX) + (x_first / WX)) * kC0Stride);
    // stage this view's tile (wave-private: program order is enough)
    VCY_SETPRIO(3);
    wave_lds_fence();
    if (kRaw) {
      raw_tile_wait();  // this view's pixels have landed in raw_buf(cur)
    } else {
      tile_fill(v, tile_of(vi), lane, (float*)tile);
    }
    wave_lds_fence();
    // the next live view's tile is fetched while this one is computed
    int vnext = next_view(live, vi);
    if (kRaw && vnext < vi_end) raw_prefetch(views[kRows ? (vnext & 7) : vnext].v, tile_of(vnext), lane, raw_buf(cur ^ 1));
    ++n_processed;
    const float pitchf = tile_of(vi).pitchf;
    const int base = tile_of(vi).base;
    const int big_pitch = kRaw ? 16 : (int)pitchf;  // pixels per row of the big tile
    // the four taps of the sample whose upper left pixel is tile element idx
    const lds_float* rawcur = (const lds_float*)raw_buf(cur);
    auto quad_at = [&](unsigned idx) -> float4 {
      if constexpr (kRaw) {
        const lds_float* p = rawcur + idx;
        return make_float4(p[0], p[1], p[16], p[17]);
      } else {
        const lds_float* p = (const lds_float*)(float*)tile + idx;
        const lds_float* p2 = p + big_pitch;
        return make_float4(p[0], p[1], p2[0], p2[1]);
      }
    };

    const float lo_x = tile_of(vi).lo_x, hi_x = tile_of(vi).hi_x;
    const float lo_y = tile_of(vi).lo_y, hi_y = tile_of(vi).hi_y;
    const bool is_ortho = GEN && mode.ortho != 0, is_nn = GEN && mode.interp == VCY_INTERP_NN;
    // c1 + c2 of this lane's (y, z): the inner sum of pc = t + (c0 + (c1 + c2)) (voxel_carver.cc:453)
    const float h12x = v.r[0][1] * py + v.r[0][2] * pz, h12y = v.r[1][1] * py + v.r[1][2] * pz;
    const float h12z = v.r[2][1] * py + v.r[2][2] * pz;
    VCY_SETPRIO(0);
    VCY_PT(1);

    // Straight-line fast path for the 8 voxels of this thread (no divergent control flow, so
    // the eight LDS reads and the arithmetic interleave); voxels the tile does not cover are
    // only recorded here and handled below.  SURE: the prologue has proved that every voxel of the
    // brick samples inside this tile (TileInfo::sure), so the per-voxel tests are compiled out.
    auto carve_view = [&](auto sure_tag) {
      constexpr bool SURE = decltype(sure_tag)::value;
      bool slow[WX];
      bool any_slow = false;
      bool moved = false;  // some voxel of this lane changed
      // Every operation is the reference's, in its order, as plain fp32 instructions.
#pragma unroll
      for (int k = 0; k < WX; ++k) {
        const float pcz = v.t[2] + (c0[16 + k] + h12z);
        // pinhole: u = fx / z * x + cx (camera.cc:133-136); orthographic: u = x (camera.cc:201-205)
        float qx = 1.0f, qy = 1.0f;
        if (!is_ortho) {
          qx = div_view<DIV>(v.fx, pcz);
          qy = SAMEF ? qx : div_view<DIV>(v.fy, pcz);
        }
        const float pcx = v.t[0] + (c0[2 * k] + h12x), pcy = v.t[1] + (c0[2 * k + 1] + h12y);
        const float u = is_ortho ? pcx : qx * pcx + v.cx;
        const float w = is_ortho ? pcy : qy * pcy + v.cy;
        bool in_tile = true;
        if (!SURE) {
          // orthographic: only `pc.z < 0` is skipped (voxel_carver.cc:456)
          const bool zfast = is_ortho ? !(pcz < 0.0f) : in_fast_div_range(pcz);
          in_tile = zfast && u >= lo_x && u <= hi_x && w >= lo_y && w <= hi_y;
          slow[k] = !in_tile;
          any_slow = any_slow || !in_tile;
        }
        const float fu = floorf(u), fw = floorf(w);
        const float lu = u - fu, lv = w - fw;
        const float mu = 1.0f - lu, mv = 1.0f - lv;
        // any index is harmless when !in_tile (the sample is discarded); keep it inside the tile
        unsigned idx = (unsigned)((int)__builtin_fmaf(fw, pitchf, fu) + base);
        if (!SURE) idx = min(idx, (unsigned)(kRaw ? 256 - 18 : max(kBigPixels - big_pitch - 2, 0)));
        const float4 q = quad_at(idx);
        // ((1-lu)(1-lv)) s00 + (lu (1-lv)) s10 + ((1-lu) lv) s01 + (lu lv) s11, summed left to right (:69-73)
        float dist = ((((mu * mv) * q.x) + ((lu * mv) * q.y)) + ((mu * lv) * q.z)) + ((lu * lv) * q.w);
        if (is_nn) {
          // SdfInterpolationNn (voxel_carver.cc:16-38): round half away from zero == floor + (frac >= .5)
          // for the non-negative in-ROI coordinates; the quad already holds the ROI-clamped neighbours
          const float top = lu >= 0.5f ? q.y : q.x, bot = lu >= 0.5f ? q.w : q.z;
          dist = lv >= 0.5f ? bot : top;
        }
        bool ok = in_tile;
        if (TRUNC) ok = ok && !(dist < -1.0f);
        if (CHECKMAX) ok = ok && !(n[k] > (NT)g.max_update_num);
        moved = apply_sample<UPDATE>(ok, dist, g.weight, s[k], n[k]) || moved;
      }
      if (!SURE && any_slow) {
#pragma unroll
        for (int k = 0; k < WX; ++k) {
          if (slow[k]) {
            float dist = 0.0f;
            bool ok = sample_generic(&v, mode, g.px[min(x_first + k, g.nx - 1)], py, pz, &dist);
            if (CHECKMAX) ok = ok && !(n[k] > (NT)g.max_update_num);
            moved = apply_sample<UPDATE>(ok, dist, g.weight, s[k], n[k]) || moved;
          }
        }
      }
      const unsigned long long mv = __ballot(moved);
      changed_lanes |= mv;
      return mv != 0ull;
    };
    // ---- select-free fast path -------------------------------------------------------------
    // A `sure` tile (every sample provably inside it and inside div_view2's depth range) whose update
    // needs no per-voxel case distinction: kMax on a brick that is touched everywhere, or the unit-weight
    // average on a state with "update_num == 0 implies sdf == lowest()".  Same operations as above, in
    // the same order; what changes is what they cost on the SIMD:
    //  - the LDS byte address of the taps comes out of the float pipeline: a = fw * (bytes per tile row) +
    //    (fu * (bytes per element) + (element size * base + tile offset)), every term an integer below 2^22,
    //    evaluated in units of 2^-149 so that the bits of the result ARE the address (2 fma instead of fma,
    //    cvt, shift-add);
    //  - the two wave-uniform terms of that sum sit in VGPRs (back-to-back scalar operands halve the issue rate);
    //  - the update is a compare / select / carry chain through VCC (update_max_touched), or a plain store for
    //    a brick that has not been touched at all (FIRST).
    // (GEN kernels take them too: nearest-neighbour taps and orthographic projection are uniform branches
    // inside the run)
    constexpr bool kFastMax = UPDATE == VCY_UPDATE_MAX && !TRUNC && !CHECKMAX;
    constexpr bool kFastWa = UPDATE == kUpdateWaUnitWeight && !CHECKMAX;
    // general weights: only the brick-wide flavour (UNIFORM) of the run, where the weights are formed once per view
    constexpr bool kFastWaGeneral = UPDATE == VCY_UPDATE_WEIGHTED_AVERAGE && !CHECKMAX;
    // FIRST: no voxel of the brick has been touched yet (a fresh slab): the update is `sdf = dist, update_num = 1`
    // for every voxel (voxel_carver.cc:482-486), whatever the old value.
    // NOTRUNC: the prologue has proved that no sample of this tile is below -1 (TileInfo::sure bit 1): the
    // truncation test of the weighted average and its two selects are compiled out.
    // UNIFORM (implies NOTRUNC): every voxel has update_num == fnu before this view and is updated by it.
    auto carve_view_fast = [&](auto first_tag, auto notrunc_tag, auto uniform_tag) -> bool {
      constexpr bool FIRST = decltype(first_tag)::value;
      constexpr bool NOTRUNC = decltype(notrunc_tag)::value;
      constexpr bool UNIFORM = decltype(uniform_tag)::value;
      // the brick's common weights, in VGPRs (uniform values; opaque to the compiler so that they are not
      // folded back into scalar operands): (fn * sdf + dist) * (1 / (fn + 1)), voxel_carver.cc:88-95
      float fn_v = 0.0f, inv_v = 0.0f, wgt_v = 1.0f;
      if constexpr (UNIFORM) {
        const float f1 = fnu + 1.0f;
        asm volatile("v_mov_b32_e32 %0, %1" : "=v"(fn_v) : "s"(fnu));
        if constexpr (UPDATE == kUpdateWaUnitWeight) {
          inv_v = rcp_count(fn_v + 1.0f);
        } else {  // (w * n, w and 1 / (w * (n + 1)) of voxel_carver.cc:91-93)
          asm volatile("v_mov_b32_e32 %0, %1" : "=v"(wgt_v) : "s"(g.weight));
          inv_v = div_fast(1.0f, wgt_v * (fn_v + 1.0f));
          fn_v = wgt_v * fn_v;
        }
        fnu = f1;
      }
      // uniform -> VGPR (opaque to the compiler, which would otherwise fold them back into SGPR operands)
      float pitch16, cmagic;
      constexpr int kElemB = 4;                 // bytes per tile element (a pixel)
      // The address sum is carried out in units of 2^-149, i.e. in denormals (fp32 denormals are on for this
      // library and v_fma_f32 handles them at full rate): the bit pattern of the result IS the integer, no
      // mask or conversion needed.  The constant may be negative (base < 0); the final sum never is.
      constexpr float kAddrUnit = 0x1p-149f;
      {
        const float p16 = pitchf * ((float)kElemB * kAddrUnit);  // bytes per tile row; pitch <= 512: exact
        const unsigned lds_off = kRaw ? (unsigned)(size_t)rawcur : (unsigned)(size_t)(const lds_float*)(float*)tile;
        const int ci = kElemB * base + (int)lds_off;  // |16 base| < 2^22 (TileInfo::sure)
        const float cm = ci < 0 ? -__int_as_float(-ci) : __int_as_float(ci);
        asm volatile("v_mov_b32_e32 %0, %1" : "=v"(pitch16) : "s"(p16));
        asm volatile("v_mov_b32_e32 %0, %1" : "=v"(cmagic) : "s"(cm));
      }
      // Four voxels at a time.  Phase A: image coordinates, fractions and the LDS reads (in flight
      // together); phase B: weights, sample, update.
      unsigned long long took = 0;
      constexpr int kGroup = VCY_FAST_GROUP;
#pragma unroll
      for (int k0 = 0; k0 < WX; k0 += kGroup) {
        float lu[kGroup], lv[kGroup];
        f4 q[kGroup];
#pragma unroll
        for (int j = 0; j < kGroup; ++j) {
          const int k = k0 + j;
          const float pcz = v.t[2] + (c0[16 + k] + h12z);
          const float qx = div_view<DIV>(v.fx, pcz);
          const float qy = SAMEF ? qx : div_view<DIV>(v.fy, pcz);
          const float pcx = v.t[0] + (c0[2 * k] + h12x), pcy = v.t[1] + (c0[2 * k + 1] + h12y);
          float u = qx * pcx + v.cx, w = qy * pcy + v.cy;
          if constexpr (GEN) {
            if (is_ortho) u = pcx, w = pcy;  // (uniform) camera.cc:201-205
          }
          const float fu = floorf(u), fw = floorf(w);
          lu[j] = u - fu;
          lv[j] = w - fw;
          const float a = __builtin_fmaf(fw, pitch16, __builtin_fmaf(fu, (float)kElemB * kAddrUnit, cmagic));
          const unsigned addr = __float_as_uint(a);
          const lds_float* tp = (const lds_float*)(size_t)addr;
          if constexpr (kRaw) {
            q[j] = f4{tp[0], tp[1], tp[16], tp[17]};
          } else {
            const lds_float* tp2 = tp + big_pitch;
            q[j] = f4{tp[0], tp[1], tp2[0], tp2[1]};
          }
        }
#pragma unroll
        for (int j = 0; j < kGroup; ++j) {
          const int k = k0 + j;
          const float mu = 1.0f - lu[j], mv = 1.0f - lv[j];
          float dist =
              ((((mu * mv) * q[j].x) + ((lu[j] * mv) * q[j].y)) + ((mu * lv[j]) * q[j].z)) + ((lu[j] * lv[j]) * q[j].w);
          if constexpr (GEN) {
            if (is_nn) {  // (uniform) SdfInterpolationNn, as in the checked loop above
              const float top = lu[j] >= 0.5f ? q[j].y : q[j].x, bot = lu[j] >= 0.5f ? q[j].w : q[j].z;
              dist = lv[j] >= 0.5f ? bot : top;
            }
          }
          if constexpr (FIRST) {
            s[k] = dist;
            n[k] = (NT)1;
          } else if constexpr (kFastMax) {
            update_max_touched(dist, s[k], n[k], took);
          } else if constexpr (UNIFORM) {
            if constexpr (UPDATE == kUpdateWaUnitWeight) s[k] = (fn_v * s[k] + dist) * inv_v;
            else s[k] = (fn_v * s[k] + wgt_v * dist) * inv_v;
          } else if constexpr (kFastWa) {
            update_wa_unit<TRUNC && !NOTRUNC>(dist, s[k], n[k], took);
          }
        }
      }
      // every lane took every sample, unless the update was conditional (kMax on a touched brick, the truncating average)
      constexpr bool kConditional = !FIRST && !UNIFORM && (kFastMax || (kFastWa && TRUNC && !NOTRUNC));
      changed_lanes |= kConditional ? took : ~0ull;
      return (kFastMax && !FIRST) ? took != 0ull : true;
    };
    bool brick_moved;
    const int sure_bits = __builtin_amdgcn_readfirstlane(tile_of(vi).sure);
    const bool sure = (sure_bits & 1) != 0, never_truncated = (sure_bits & 2) != 0;
    // (Branch weights: the checked loops below are the rare ones in the kernels that have a select-free loop;
    // the register allocator then spills there, if anywhere, and not in the loops that do the work.)
    constexpr bool kHasFast = kFastMax || kFastWa || kFastWaGeneral;
    const bool fast_first = kFastMax && sure && none_touched;
    const bool fast_next = (kFastMax && sure && all_touched) || (kFastWa && sure && implied);
    // general weights: every voxel updated by this view and all counts equal -- a first touch stores the sample
    // (voxel_carver.cc:482-486), later views average with the brick's weights
    const bool fast_general = kFastWaGeneral && sure && uniform_cnt && (!TRUNC || never_truncated);
    if (kFastWaGeneral && __builtin_expect_with_probability(fast_general, 1, 0.9)) {
      if (fnu < 1.0f) {
        brick_moved = carve_view_fast(std::true_type{}, std::false_type{}, std::false_type{});
        fnu = 1.0f;
      } else {
        brick_moved = carve_view_fast(std::false_type{}, std::true_type{}, std::true_type{});
      }
      VCY_PT(2);
      VCY_PT_COUNT(7);
    } else if (__builtin_expect_with_probability(fast_next, kHasFast && !kFastWaGeneral, 0.9)) {
      // weighted average: no truncation test when it cannot fire, and brick-wide weights while the counts agree
      const bool all_updated = kFastWa && (!TRUNC || never_truncated);
      if (kFastWa && all_updated && uniform_cnt) {
        brick_moved = carve_view_fast(std::false_type{}, std::true_type{}, std::true_type{});
      } else if (kFastWa && all_updated) {
        brick_moved = carve_view_fast(std::false_type{}, std::true_type{}, std::false_type{});
      } else {
        if (kFastWa) leave_uniform();
        brick_moved = carve_view_fast(std::false_type{}, std::false_type{}, std::false_type{});
      }
      VCY_PT(2);
      VCY_PT_COUNT(7);
    } else if (__builtin_expect_with_probability(fast_first, kHasFast, 0.99)) {
      brick_moved = carve_view_fast(std::true_type{}, std::false_type{}, std::false_type{});
      VCY_PT(2);
      VCY_PT_COUNT(7);
    } else if (sure) {
      leave_uniform();
      brick_moved = carve_view(std::true_type{});
      VCY_PT(3);
      VCY_PT_COUNT(8);
    } else {
      leave_uniform();
      brick_moved = carve_view(std::false_type{});
      VCY_PT(4);
      VCY_PT_COUNT(9);
    }
    VCY_SETPRIO(3);
    if (brick_moved) VCY_PT_COUNT(11);
    none_touched = false;  // (a checked view may have touched only some voxels)
    if (!kOne) refresh_all_touched();  // (only the views that follow ask)

    // state moved: some of the remaining views may have become droppable (min(sdf) only grows)
    // (an unchanged brick leaves every bound comparison as it was)
    if (!kOne && want_bound && UPDATE == VCY_UPDATE_MAX && brick_moved) {
      live = live_views();
      const int v2 = next_view(live, vi);
      if (v2 != vnext) {
        vnext = v2;
        // (the dropped view's pixels may still be arriving in that buffer: loads complete in order)
        if (kRaw && vnext < vi_end) raw_prefetch(views[kRows ? (vnext & 7) : vnext].v, tile_of(vnext), lane, raw_buf(cur ^ 1));
      }
    }
    vi = vnext;
    cur ^= 1;
    VCY_PT(5);
  }

  // ---- write back what changed (update_num grows with every change) ----------------------------
  if constexpr (kRows) {
    if (jc >= 0) finish_brick();
    wave_lds_fence();
    if (pair_count != nullptr && lane == 0) atomicAdd(&pair_count[bz], (unsigned long long)n_processed);
    // Whole row segments: request i writes z slice i of the segment -- lane L the 16-byte chunk L & 7 of voxel row
    // 8 i + (L >> 3), so 8 lanes store the 128 contiguous bytes the segment has in that row (NB = 4) -- for the rows
    // whose carving lane changed (stage_mask of the chunk's brick; every row of a fresh slab).
    {
      const int yw = by * BY + (lane >> 3), c16 = lane & 7, jw = c16 >> 1;
      const unsigned long long mw = jw < NB ? stage_mask[jw] : 0ull;
      const bool col_ok = jw < NB && yw < g.ny && x_seg + 4 * c16 < g.nx;
#pragma unroll
      for (int i = 0; i < BZ; ++i) {
        const int row = 8 * i + (lane >> 3);
        if (zl0 + i < g.nz_local && col_ok && ((mw >> row) & 1ull) != 0ull) {
          const f4 q = *(const lds_float4*)(float4*)(stage_s + (jw * 64 + row) * WX + (c16 & 1) * 4);
          *(float4*)(g.sdf + (((int64_t)(zl0 + i) * g.ny + yw) * g.nx + x_seg + 4 * c16)) = make_float4(q.x, q.y, q.z, q.w);
        }
      }
      // counters: lane L the 8 (u8) / 16 (u16) bytes brick L & 3 has in voxel row 16 i + (L >> 2)
      const int jn = lane & 3;
      const unsigned long long mn = jn < NB ? stage_mask[jn] : 0ull;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 16 * i + (lane >> 2);
        const int yn = by * BY + (row & 7), zn = zl0 + (row >> 3);
        if (jn < NB && yn < g.ny && zn < g.nz_local && x_seg + WX * jn < g.nx && ((mn >> row) & 1ull) != 0ull) {
          const CountVecR cv = *(const lds_countvec_r*)(CountVecR*)(stage_n + (jn * 64 + row) * WX);
          *(CountVecR*)(cnt + (((int64_t)zn * g.ny + yn) * g.nx + x_seg + WX * jn)) = cv;
        }
      }
    }
    return;
  }
  leave_uniform();
  // Where this lane's run lies is worked out again from the thread id (opaque to the compiler): kept from the prologue
  // it occupies three registers through every view, and the weighted-average kernels are short of exactly those --
  // their loop pre-header spilled to scratch, which every wave executes.
  int tid_w = (int)threadIdx.x;
  asm volatile("" : "+v"(tid_w));
  const int lane_w = tid_w & 63;
  const int y_w = by * BY + (lane_w & (BY - 1)), zl_w = zl0 + (lane_w >> 3);
  const bool lane_valid_w = y_w < g.ny && zl_w < g.nz_local;
  const int64_t row0_w = ((int64_t)min(zl_w, g.nz_local - 1) * g.ny + min(y_w, g.ny - 1)) * g.nx;
  // ("paircount" 1: (brick, view) pairs processed, per brick layer of the launch -- what the slab planner's
  // estimate is checked against, and what bench.py reports as the fraction of pairs the scene leaves)
#if defined(VCY_DEV_EXIT_AT) && VCY_DEV_EXIT_AT == 5  // development build: where a wave's scalar instructions go (profiles/tools/salu_attribution.sh)
  {
    coop_leave();
    return;
  }
#endif
  if (pair_count != nullptr && lane == 0) atomicAdd(&pair_count[bz], (unsigned long long)n_processed);
#ifndef VCY_NO_BRICK_MIN_WRITE
  if (brick_min != nullptr && implied) {  // (lanes outside the grid hold copies of voxels inside it)
    float m = s[0];
#pragma unroll
    for (int k = 1; k < WX; ++k) m = fminf(m, s[k]);
    const float smin = wave_min(m);
    if (lane == 0) brick_min[brick_lin] = smin;
  }
#endif
  if (coop) {
    // Cooperative write-back.  A wave's own stores are 64 pieces of 16 bytes in 64 different rows; the 128-byte line of
    // a row is completed by the other three waves of the workgroup at other times, and in a launch that also READS the
    // state (a view over a carved grid) the L2 writes such lines back before they are complete: 10.4 GB written for
    // 6.4 GB of state at 1024^3 in weighted-average mode, and four write requests where one would do
    // (profiles/r04/per_view_tsdf_pmc.txt).  Here every wave leaves its runs in LDS (row = lane, columns of its brick),
    // and after one barrier the waves share out the 64 rows of the workgroup's 32 x 8 x 8 block: 8 lanes = one
    // 128-byte row segment of sdf, 4 lanes = one row segment of update_num.  A row is stored when the lane that owned
    // it changed (coop_mask: a wave that left early, or a lane outside the grid, owns none).
    const bool changed = lane_valid_w && (fresh != 0 || ((changed_lanes >> lane_w) & 1ull) != 0ull);
    const unsigned long long my_mask = __ballot(changed);
    {
      lds_float4* rs = (lds_float4*)(float4*)(coop_s + lane_w * kCoopSdfPitch + wave * WX);
      rs[0] = f4{s[0], s[1], s[2], s[3]};
      rs[1] = f4{s[4], s[5], s[6], s[7]};
      CountVec8 cv;
#pragma unroll
      for (int k = 0; k < WX; ++k) cv[k] = (CountT)n[k];
      *(lds_countvec*)(CountVec8*)(coop_n + lane_w * coop_cnt_pitch<CountT>() + wave * WX) = cv;
      if (lane == 0) coop_mask[wave] = my_mask, coop_mask[kWgWaves + wave] = 1ull;
    }
    __syncthreads();
    // the row groups are dealt to the waves that are still here (a wave whose every view was dropped has left)
    int n_here = 0, my_rank = 0;
#pragma unroll
    for (int w = 0; w < kWgWaves; ++w) {
      const int here = __builtin_amdgcn_readfirstlane((int)coop_mask[kWgWaves + w]);
      n_here += here;
      my_rank += (w < wave) ? here : 0;
    }
    const int xb = bx * BX;
    // sdf: a row of the block is 2 kWgWaves chunks of 16 bytes, an instruction covers 64 / (2 kWgWaves) rows
    constexpr int kSdfChunks = 2 * kWgWaves, kSdfRows = 64 / kSdfChunks;
    for (int gi = my_rank; gi < 64 / kSdfRows; gi += n_here) {
      const int r = gi * kSdfRows + lane / kSdfChunks, ch = lane % kSdfChunks;
      if ((coop_mask[ch >> 1] >> r) & 1ull) {
        const f4 v = *(lds_float4*)(float4*)(coop_s + r * kCoopSdfPitch + ch * 4);
        const int64_t rowg = ((int64_t)(zl0 + (r >> 3)) * g.ny + (by * BY + (r & 7))) * g.nx;
#ifndef VCY_DEV_SKIP_SDF_STORE  // (development builds: which of the two arrays the written bytes belong to)
        // (whole 128-byte row segments: as streaming stores when the launch asks for it -- state_flags bit 4.  A scalar
        // base with 32-bit offsets instead of these 64-bit row addresses was measured in round 6: 23 vector instructions
        // fewer per wave, 18 scalar ones more, +1 % in time -- profiles/r06/one_view.txt)
        if (nt_store) __builtin_nontemporal_store(v, (f4*)(g.sdf + rowg + xb + ch * 4));
        else *(float4*)(g.sdf + rowg + xb + ch * 4) = make_float4(v.x, v.y, v.z, v.w);
#else
        if (v.x == 1.2345e-30f) g.sdf[0] = v.y;
#endif
      }
    }
    // update_num: kWgWaves chunks of 8 counters per row, 64 / kWgWaves rows per instruction
    constexpr int kCntRows = 64 / kWgWaves;
    for (int gi = my_rank; gi < 64 / kCntRows; gi += n_here) {
      const int r = gi * kCntRows + lane / kWgWaves, ch = lane % kWgWaves;
      if ((coop_mask[ch] >> r) & 1ull) {
        const CountVec8 cv = *(lds_countvec*)(CountVec8*)(coop_n + r * coop_cnt_pitch<CountT>() + ch * WX);
        const int64_t rowg = ((int64_t)(zl0 + (r >> 3)) * g.ny + (by * BY + (r & 7))) * g.nx;
#ifndef VCY_DEV_SKIP_CNT_STORE
        if (nt_store) __builtin_nontemporal_store(cv, (CountVec8*)(cnt + rowg + xb + ch * WX));
        else *(CountVec8*)(cnt + rowg + xb + ch * WX) = cv;
#else
        if (cv[0] == (CountT)12345) cnt[0] = cv[1];
#endif
      }
    }
  } else if (lane_valid_w) {
    if (vec_io) {
      bool changed = fresh != 0;  // (a fresh slab has never been written: every voxel is stored)
#ifdef VCY_FLOOR_NO_STORES  // development build (issue floor): results stay live, nothing is stored
      {  // (every value stays live: a dead s[k] would take its whole update chain with it)
        float ssum = 0.0f, nsum = 0.0f;
#pragma unroll
        for (int k = 0; k < WX; ++k) ssum += s[k], nsum += (float)n[k];
        changed = ssum == 1.2345e-30f && nsum == 777.25f;
      }
#else
      changed = changed || ((changed_lanes >> lane_w) & 1ull) != 0ull;
#endif
      if (changed) {
#ifdef VCY_FLOOR_DUMMY_STORES  // development build: the same store instructions, all into 64 rows of ONE brick row (never reach HBM)
        const int64_t row0_ = ((int64_t)(lane >> 3) * g.ny + (lane & 7)) * g.nx;
        const int x_first_ = (x_first & 1023);
#define row0_w row0_
#define x_first x_first_
#endif
        *(float4*)(g.sdf + row0_w + x_first) = make_float4(s[0], s[1], s[2], s[3]);
        *(float4*)(g.sdf + row0_w + x_first + 4) = make_float4(s[4], s[5], s[6], s[7]);
        CountVec cv;
#pragma unroll
        for (int k = 0; k < WX; ++k) cv[k] = (CountT)n[k];
        *(CountVec*)(cnt + row0_w + x_first) = cv;
#ifdef VCY_FLOOR_DUMMY_STORES
#undef row0_w
#undef x_first
#endif
      }
    } else {
#pragma unroll
      for (int k = 0; k < WX; ++k) {
        if (x_first + k < g.nx) {
          const int64_t idx = row0_w + x_first + k;
          if (fresh || ((changed_lanes >> lane_w) & 1ull) != 0ull) {  // (unchanged voxels of a changed lane store what they hold)
            g.sdf[idx] = s[k];
            cnt[idx] = (CountT)n[k];
          }
        }
      }
    }
  }
  VCY_PT(6);
  VCY_PT_FLUSH(lane);
}

template <typename CountT, int UPDATE, bool TRUNC, bool SAMEF>
void launch_fused_4(bool big, bool checkmax, dim3 grid, hipStream_t s, const GridParams& g, const FusedView* dv,
                    const float* c2, int nv, const ModeParams& m, int nbx, int nby, int cull, int fresh,
                    const FootprintRecord* recs, int64_t nbricks, float* bmin, const int* wgl, unsigned long long* pcnt,
                    int row_units) {
  const bool gen = m.ortho != 0 || m.interp == VCY_INTERP_NN;
  if (row_units < 0) {
    // a launch of ONE view (NB == 0): the NB = 1 launch shape, the view count a compile-time constant
#define VCY_ONE(GEN_, DIV_)                                                                                          \
  hipLaunchKernelGGL((carve_fused_kernel<CountT, UPDATE, TRUNC, SAMEF, false, kTileRaw, GEN_, DIV_, 0>), grid,         \
                     dim3(64 * kWgWaves),                                                                            \
                     (size_t)kWgWaves * 64 * sizeof(float4) + ((fresh & 8) ? coop_lds_bytes<CountT>() : 0), s,          \
                     g, dv, c2, 1, m, nbx, nby, make_block_decode(grid.x, nbx, nby), cull, fresh, recs, nbricks, bmin, wgl, pcnt)
#ifdef VCY_DEV_BENCH_KERNELS_ONLY
    if (gen || m.div_level != 2 || !SAMEF || sizeof(CountT) != 1 || UPDATE == VCY_UPDATE_WEIGHTED_AVERAGE) {
      fprintf(stderr, "VCY_DEV_BENCH_KERNELS_ONLY: kernel variant not built\n");
      abort();
    }
    if constexpr (SAMEF && sizeof(CountT) == 1 && UPDATE != VCY_UPDATE_WEIGHTED_AVERAGE) VCY_ONE(false, 2);
#else
    if (gen) VCY_ONE(true, 0);
    else if (m.div_level == 2) VCY_ONE(false, 2);
    else if (m.div_level == 1) VCY_ONE(false, 1);
    else VCY_ONE(false, 0);
#endif
#undef VCY_ONE
    return;
  }
  if (row_units > 0) {
    // the few-view flavour: `grid` workgroups of kRowWaves waves, a segment of kRowBricks bricks per wave; `nbx` =
    // segments per brick row, `row_units` = segments of the launch (what the block decode deals to the XCDs)
    const BlockDecode bd = make_block_decode((unsigned)row_units, nbx, nby);
    const size_t lds = (size_t)kRowWaves * row_lds_bytes_per_wave<CountT, kRowBricks>();
#define VCY_ROWS(GEN_, DIV_)                                                                                         \
  hipLaunchKernelGGL((carve_fused_kernel<CountT, UPDATE, TRUNC, SAMEF, false, kTileRaw, GEN_, DIV_, kRowBricks>), grid, \
                     dim3(64 * kRowWaves), lds, s, g, dv, c2, nv, m, nbx, nby, bd, cull, fresh, recs, nbricks, bmin, wgl, pcnt)
#ifdef VCY_DEV_BENCH_KERNELS_ONLY
    if (gen || m.div_level != 2 || !SAMEF || sizeof(CountT) != 1 || UPDATE == VCY_UPDATE_WEIGHTED_AVERAGE) {
      fprintf(stderr, "VCY_DEV_BENCH_KERNELS_ONLY: kernel variant not built\n");
      abort();
    }
    if constexpr (SAMEF && sizeof(CountT) == 1 && UPDATE != VCY_UPDATE_WEIGHTED_AVERAGE) VCY_ROWS(false, 2);
#else
    if (gen) VCY_ROWS(true, 0);
    else if (m.div_level == 2) VCY_ROWS(false, 2);
    else if (m.div_level == 1) VCY_ROWS(false, 1);
    else VCY_ROWS(false, 0);
#endif
#undef VCY_ROWS
    return;
  }
#define VCY_FUSED(CM, TQ_, GEN_, DIV_)                                                                           \
  hipLaunchKernelGGL((carve_fused_kernel<CountT, UPDATE, TRUNC, SAMEF, CM, TQ_, GEN_, DIV_>), grid, dim3(64 * kWgWaves),  \
                     (size_t)kWgWaves * tile_f4_per_wave<TQ_>() * sizeof(float4) + (size_t)kWgWaves * nv * sizeof(TileInfo) + \
                         ((fresh & 8) ? coop_lds_bytes<CountT>() : 0), s,                                         \
                     g, dv, c2, nv, m, nbx, nby, make_block_decode(grid.x, nbx, nby), cull, fresh, recs, nbricks, bmin, wgl, pcnt)
#define VCY_FUSED_G(CM, TQ_)                                                                                     \
  do {                                                                                                           \
    if (gen) VCY_FUSED(CM, TQ_, true, 0);                                                                        \
    else if (m.div_level == 2) VCY_FUSED(CM, TQ_, false, 2);                                                     \
    else if (m.div_level == 1) VCY_FUSED(CM, TQ_, false, 1);                                                     \
    else VCY_FUSED(CM, TQ_, false, 0);                                                                           \
  } while (0)
#ifdef VCY_DEV_BENCH_KERNELS_ONLY
  // development builds (profiles/tools/build_variant.sh): only the instantiations bench.py launches,
  // a 20x shorter compile; anything else aborts
  // (the benchmark's 32 / 64 views fit one-byte counters: vcy_ctx::cnt_bytes, lazy widening)
  if (big || checkmax || gen || m.div_level != 2 || !SAMEF || sizeof(CountT) != 1 ||
      UPDATE == VCY_UPDATE_WEIGHTED_AVERAGE) {
    fprintf(stderr, "VCY_DEV_BENCH_KERNELS_ONLY: kernel variant not built\n");
    abort();
  }
  if constexpr (SAMEF && sizeof(CountT) == 1 && UPDATE != VCY_UPDATE_WEIGHTED_AVERAGE) VCY_FUSED(false, kTileRaw, false, 2);
#else
  if (big) {
    if (checkmax) VCY_FUSED_G(true, kTileBig); else VCY_FUSED_G(false, kTileBig);
  } else {
    if (checkmax) VCY_FUSED_G(true, kTileRaw); else VCY_FUSED_G(false, kTileRaw);
  }
#endif
#undef VCY_FUSED_G
#undef VCY_FUSED
}

template <typename CountT, int UPDATE>
void launch_fused_2(bool big, bool trunc, bool samef, bool checkmax, dim3 grid, hipStream_t s, const GridParams& g,
                    const FusedView* dv, const float* c2, int nv, const ModeParams& m, int nbx, int nby, int cull, int fresh,
                    const FootprintRecord* recs, int64_t nbricks, float* bmin, const int* wgl, unsigned long long* pcnt,
                    int row_units) {
  if (trunc) {
    if (samef) launch_fused_4<CountT, UPDATE, true, true>(big, checkmax, grid, s, g, dv, c2, nv, m, nbx, nby, cull, fresh, recs, nbricks, bmin, wgl, pcnt, row_units);
    else launch_fused_4<CountT, UPDATE, true, false>(big, checkmax, grid, s, g, dv, c2, nv, m, nbx, nby, cull, fresh, recs, nbricks, bmin, wgl, pcnt, row_units);
  } else {
    if (samef) launch_fused_4<CountT, UPDATE, false, true>(big, checkmax, grid, s, g, dv, c2, nv, m, nbx, nby, cull, fresh, recs, nbricks, bmin, wgl, pcnt, row_units);
    else launch_fused_4<CountT, UPDATE, false, false>(big, checkmax, grid, s, g, dv, c2, nv, m, nbx, nby, cull, fresh, recs, nbricks, bmin, wgl, pcnt, row_units);
  }
}

template <typename CountT>
void launch_fused_1(bool big, int update, bool trunc, bool samef, bool checkmax, dim3 grid, hipStream_t s,
                    const GridParams& g, const FusedView* dv, const float* c2, int nv, const ModeParams& m, int nbx, int nby, int cull, int fresh,
                    const FootprintRecord* recs, int64_t nbricks, float* bmin, const int* wgl, unsigned long long* pcnt,
                    int row_units) {
  if (update == VCY_UPDATE_MAX)
    launch_fused_2<CountT, VCY_UPDATE_MAX>(big, trunc, samef, checkmax, grid, s, g, dv, c2, nv, m, nbx, nby, cull, fresh, recs, nbricks, bmin, wgl, pcnt, row_units);
  else if (g.weight == 1.0f)
    launch_fused_2<CountT, kUpdateWaUnitWeight>(big, trunc, samef, checkmax, grid, s, g, dv, c2, nv, m, nbx, nby, cull, fresh, recs, nbricks, bmin, wgl, pcnt, row_units);
  else
    launch_fused_2<CountT, VCY_UPDATE_WEIGHTED_AVERAGE>(big, trunc, samef, checkmax, grid, s, g, dv, c2, nv, m, nbx, nby, cull, fresh, recs, nbricks, bmin, wgl, pcnt, row_units);
}

#ifdef VCY_FUSED_PART
}  // namespace

// (carve_fused_u8.hip / carve_fused_u16.hip) this unit's half of the kernel instances behind its one exported function
void VCY_FUSED_PART_FN(bool big, int update, bool trunc, bool samef, bool checkmax, unsigned grid_x, hipStream_t s,
                       const GridParams& g, const void* views, const float* c2, int nv, const ModeParams& m, int nbx, int nby,
                       int cull, int state_flags, const void* records, int64_t nbricks, float* bmin, const int* wgl,
                       unsigned long long* pcnt, int row_units) {
  launch_fused_1<VCY_FUSED_PART_TYPE>(big, update, trunc, samef, checkmax, dim3(grid_x), s, g, (const FusedView*)views, c2, nv, m,
                                      nbx, nby, cull, state_flags, (const FootprintRecord*)records, nbricks, bmin, wgl, pcnt,
                                      row_units);
}

}  // namespace vcy

#if defined(VCY_PHASE_TIMING) && VCY_FUSED_PART == 8
// development build only (the benchmark's kernels are the one-byte ones): reads (and optionally clears) the phase counters
// of the fused kernel -- in THIS unit, whose copy of g_phase_ticks its kernels write
extern "C" int vcy_debug_phase_ticks(unsigned long long* out12, int reset) {
  unsigned long long h[256][16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(vcy::g_phase_ticks), sizeof(h)) != hipSuccess) return -1;
  for (int q = 0; q < 16; ++q) {
    out12[q] = 0;
    for (int b = 0; b < 256; ++b) out12[q] += h[b][q];
  }
  if (reset) {
    std::memset(h, 0, sizeof(h));
    if (hipMemcpyToSymbol(HIP_SYMBOL(vcy::g_phase_ticks), h, sizeof(h)) != hipSuccess) return -1;
  }
  return 0;
}
#endif

#else  // !VCY_FUSED_PART: the host side and the small kernels

// Exhaustive check of the short division sequences for ONE numerator: every significand of the
// denominator (blockIdx.x * 256 + threadIdx.x) in every binade 2^-60 .. 2^60 (blockIdx.y) the fast path
// admits (in_fast_div_range), against the IEEE quotient.  bad[0]: DIV 2 differs somewhere, bad[1]: DIV 1.
__global__ __launch_bounds__(256) void div_verify_kernel(float n, unsigned* __restrict__ bad) {
  const unsigned sig = blockIdx.x * 256u + threadIdx.x;
  const unsigned expo = 127u - 60u + blockIdx.y;
  const float d = __uint_as_float((expo << 23) | sig);
  const float ref = n / d;  // -fhip-fp32-correctly-rounded-divide-sqrt
  const bool b2 = div_view1(2, n, d) != ref, b1 = div_view1(1, n, d) != ref;
  if (__any(b2) && (threadIdx.x & 63) == 0) atomicOr(&bad[0], 1u);
  if (__any(b1) && (threadIdx.x & 63) == 0) atomicOr(&bad[1], 1u);
}

__global__ void fill_lowest_kernel(float* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = kInvalidSdf;
}

std::mutex g_div_mutex;
std::map<std::pair<int, uint32_t>, int> g_div_cache;  // (device, numerator bits) -> level

// 2, 1 or 0: the shortest sequence of div_view2 that equals IEEE division by every admissible depth for
// this numerator on this device.  ~1 G cases, about a millisecond, once per distinct focal length.
int div_level(vcy_ctx* c, float n) {
  uint32_t bits;
  std::memcpy(&bits, &n, 4);
  const std::pair<int, uint32_t> key(c->device, bits);
  std::lock_guard<std::mutex> lock(g_div_mutex);
  auto it = g_div_cache.find(key);
  if (it != g_div_cache.end()) return it->second;
  int level = 0;
  unsigned* d_bad = nullptr;
  unsigned h_bad[2] = {1u, 1u};
  if (hipMalloc(&d_bad, 2 * sizeof(unsigned)) == hipSuccess) {
    hipError_t e = hipMemsetAsync(d_bad, 0, 2 * sizeof(unsigned), c->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(div_verify_kernel, dim3((1u << 23) / 256u, 121u), dim3(256), 0, c->stream, n, d_bad);
      e = hipMemcpyAsync(h_bad, d_bad, sizeof(h_bad), hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_bad);
    if (e == hipSuccess) level = h_bad[0] == 0 ? 2 : (h_bad[1] == 0 ? 1 : 0);
    else (void)hipGetLastError();
  } else {
    (void)hipGetLastError();
  }
  g_div_cache[key] = level;
  return level;
}

bool sane(float f) { return f >= 0x1p-40f && f <= 0x1p40f; }

}  // namespace

// The c0 records of a launch (FusedView layout note: c0_all[view][x brick][kC0Stride]) from the views' first rotation
// column and the x axis table, on the device: blockIdx.y = view, thread = (x brick, k).  One fp32 multiply per entry --
// the same IEEE product the host formed until round 5, when a launch with NEW views still waited for the previous
// launch, uploaded 0.5 MB of these from pageable memory and waited again (prepare_views).
namespace {
__global__ __launch_bounds__(256) void c0_records_kernel(const FusedView* __restrict__ views, const float* __restrict__ px,
                                                         int nx, int nbw, float* __restrict__ c2) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int b = t >> 3, k = t & 7;
  if (b >= nbw) return;
  const ViewParams& v = views[blockIdx.y].v;  // (uniform: scalar loads)
  const float p = px[min(b * WX + k, nx - 1)];
  float* rec = c2 + ((size_t)blockIdx.y * nbw + b) * kC0Stride;
  rec[2 * k + 0] = v.r[0][0] * p;
  rec[2 * k + 1] = v.r[1][0] * p;
  rec[16 + k] = v.r[2][0] * p;
  rec[24 + k] = 0.0f;  // (unused quarter of the 128-byte record)
}
}  // namespace

// True when the fused kernel can take these views (otherwise the per-view kernel does).
bool fused_eligible(const vcy_ctx* c, int n_views, const vcy_view* views) {
  const vcy_update_option& u = c->opt.update_option;
  if (count_width_for(c, c->views_carved + n_views) > 2) return false;  // (32-bit counters: the per-view kernel)
  if (u.voxel_update == VCY_UPDATE_WEIGHTED_AVERAGE && !sane(u.voxel_update_weight)) return false;
  for (int i = 0; i < n_views; ++i) {
    const vcy_view& v = views[i];
    if (v.is_ortho != views[0].is_ortho) return false;  // one projection model per launch
    if (!v.is_ortho && (!sane(v.fx) || !sane(v.fy))) return false;
    if (v.width > 8192 || v.height > 8192) return false;
  }
  return true;
}

// What a fused launch derives from its views, resident on the device (the context's staging buffer):
struct PreparedViews {
  float* d_c2;          // c0 records, [view][x brick][kC0Stride]
  FusedView* d_views;   // view blocks with their window-plane addresses and the rectangle the planes are built in
  bool samef;           // fx == fy in every view
  int max_quads;        // threads per view of the window-maximum kernels (0: no planes)
};

// `need_bound`: window-maximum planes for the view-dropping bounds (kMax or truncation, see the kernel; without the memory
// for them the kernel scans the footprints instead, results are the same either way).  `need_lower`: two more planes, of
// the negated image (FusedView::has_lower).  [zlo, zhi): the slices whose image-space bounding box the planes must
// cover -- the context's slab for a carve, the whole grid for the slab planner.
int prepare_views(vcy_ctx* c, int n_views, const ViewParams* vp, bool need_bound, bool need_lower, int zlo, int zhi,
                  PreparedViews* out) {
  // per-view records of c0 = R[i][0] * px[x] for every wave brick along x (layout: kC0Stride above);
  // columns beyond nx repeat the last one
  const int nxp = (c->nx + WX - 1) / WX * WX, nbw = nxp / WX;
  const size_t c2_floats = (size_t)n_views * nbw * kC0Stride;
  const size_t c2_bytes = c2_floats * sizeof(float);
  const size_t fv_bytes = sizeof(FusedView) * (size_t)n_views;
  const size_t vp_bytes = sizeof(ViewParams) * (size_t)n_views;
  const int planes = need_lower ? 2 * kWmaxPlanes : kWmaxPlanes;
  if (need_bound) {
    size_t total = 0;
    for (int vi = 0; vi < n_views; ++vi) total += ((size_t)planes * vp[vi].width * vp[vi].height + 3) & ~(size_t)3;
    if (c->wmax_bytes < total * sizeof(float)) {
      VCY_HIP_CHECK(hipStreamSynchronize(c->stream));
      if (c->d_wmax) (void)hipFree(c->d_wmax);
      c->d_wmax = nullptr;
      c->wmax_bytes = 0;
      if (hipMalloc(&c->d_wmax, total * sizeof(float)) == hipSuccess) c->wmax_bytes = total * sizeof(float);
      else { c->d_wmax = nullptr; (void)hipGetLastError(); }
    }
  }
  // staging buffer owned by the context, grown on demand
  if (c->fused_scratch_bytes < c2_bytes + fv_bytes) {
    VCY_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->d_fused_scratch) VCY_HIP_CHECK(hipFree(c->d_fused_scratch));
    c->d_fused_scratch = nullptr;
    c->fused_scratch_bytes = 0;
    c->fused_cache_valid = false;
    VCY_HIP_CHECK(hipMalloc(&c->d_fused_scratch, c2_bytes + fv_bytes));
    c->fused_scratch_bytes = c2_bytes + fv_bytes;
  }
  float* d_c2 = (float*)c->d_fused_scratch;
  FusedView* d_views = (FusedView*)((char*)c->d_fused_scratch + c2_bytes);
  // Everything derived from the views -- the c0 records (n_views * nx products), the image-space bounding box of the
  // slab in every view, the device copies -- depends on nothing but the views' parameters (image pointers included),
  // the planes' address and two mode flags: a launch with the SAME views as the last one (the reference's loop carves a
  // sequence of grids with one camera rig; the benchmark repeats its step) takes all of it as it is.  Comparing 4 KB
  // instead of rebuilding and comparing 0.5 MB took 0.1 ms of host time out of every launch, which is what a z-slab
  // of an 8-GPU run pays 10 % of its step for.
  const bool cached = c->fused_cache_valid && c->fused_cache_vp.size() == vp_bytes &&
                      std::memcmp(c->fused_cache_vp.data(), vp, vp_bytes) == 0 && c->fused_cache_wmax == c->d_wmax &&
                      c->fused_cache_bound == need_bound && c->fused_cache_lower == need_lower &&
                      c->fused_cache_ortho == c->fused_ortho && c->fused_cache_at == (void*)d_c2 &&
                      c->fused_cache_z[0] == zlo && c->fused_cache_z[1] == zhi;
  if (!cached) {
    // The view blocks are built in page-locked host memory -- two buffers taken in turn, each with an event that says
    // when the copy out of it has been made -- and the c0 records by a kernel behind that copy: a launch with new
    // views queues behind the previous one like any other work on the stream (round 5; until then this path waited
    // for the stream twice and uploaded the c0 records, 0.5 MB at 1024^3 x 32, from a host vector).
    static_assert(kC0Stride == 32 && WX == 8, "c0_records_kernel's record layout");
    if (c->fused_stage_bytes < fv_bytes) {
      for (int q = 0; q < 2; ++q) {
        if (c->ev_fused_stage[q]) (void)hipEventSynchronize(c->ev_fused_stage[q]);
        if (c->h_fused_stage[q]) (void)hipHostFree(c->h_fused_stage[q]);
        c->h_fused_stage[q] = nullptr;
      }
      c->fused_stage_bytes = 0;
      const size_t room = fv_bytes + fv_bytes / 2 + 4096;
      for (int q = 0; q < 2; ++q) {
        VCY_HIP_CHECK(hipHostMalloc(&c->h_fused_stage[q], room, hipHostMallocDefault));
        if (!c->ev_fused_stage[q]) VCY_HIP_CHECK(hipEventCreateWithFlags(&c->ev_fused_stage[q], hipEventDisableTiming));
      }
      c->fused_stage_bytes = room;
    }
    c->fused_stage_idx ^= 1;
    VCY_HIP_CHECK(hipEventSynchronize(c->ev_fused_stage[c->fused_stage_idx]));  // (the copy of two launches ago: long made)
    FusedView* fv = (FusedView*)c->h_fused_stage[c->fused_stage_idx];
    std::memset((void*)fv, 0, fv_bytes);
    bool samef = true;
    for (int vi = 0; vi < n_views; ++vi) {
      fv[vi].v = vp[vi];
      samef = samef && (vp[vi].fx == vp[vi].fy);
    }
    int max_quads = 0;  // threads of the window-maximum kernels, per view
    if (need_bound && c->d_wmax) {
      size_t off = 0;
      for (int vi = 0; vi < n_views; ++vi) {
        const int npx = vp[vi].width * vp[vi].height;
        fv[vi].wmax = c->d_wmax + off;
        fv[vi].wmax_plane = npx;
        fv[vi].has_lower = need_lower ? 1 : 0;
        off += ((size_t)planes * npx + 3) & ~(size_t)3;  // every view 16-byte aligned
        // image-space bounding box of the slab (double precision, 16 px border; the footprints the
        // kernel looks up lie within a fraction of a pixel of the exact hull, their windows inside them)
        const int w = vp[vi].width, h = vp[vi].height;
        int rx0 = 0, ry0 = 0, rx1 = w, ry1 = h;
        {
          const double X[2] = {c->h_px_min, c->h_px_max}, Y[2] = {c->h_py_min, c->h_py_max};
          const double Z[2] = {c->h_pz[zlo], c->h_pz[zhi - 1]};
          double umin = 1e300, umax = -1e300, wmin = 1e300, wmax = -1e300;
          bool whole = false;
          for (int cr = 0; cr < 8; ++cr) {
            const double x = X[cr & 1], y = Y[(cr >> 1) & 1], z = Z[cr >> 2];
            double pc[3];
            for (int i = 0; i < 3; ++i)
              pc[i] = (double)vp[vi].t[i] + ((double)vp[vi].r[i][0] * x + (double)vp[vi].r[i][1] * y + (double)vp[vi].r[i][2] * z);
            double uu = pc[0], ww = pc[1];
            if (!c->fused_ortho) {
              if (!(pc[2] > 1e-30)) { whole = true; break; }
              uu = (double)vp[vi].fx / pc[2] * pc[0] + vp[vi].cx;
              ww = (double)vp[vi].fy / pc[2] * pc[1] + vp[vi].cy;
            }
            if (!(std::fabs(uu) < 1e9) || !(std::fabs(ww) < 1e9)) { whole = true; break; }
            umin = std::min(umin, uu), umax = std::max(umax, uu);
            wmin = std::min(wmin, ww), wmax = std::max(wmax, ww);
          }
          if (!whole) {
            rx0 = std::max(0, (int)std::floor(umin) - 16) & ~3;
            ry0 = std::max(0, (int)std::floor(wmin) - 16);
            rx1 = std::min((w + 3) & ~3, ((int)std::floor(umax) + 16 + 4) & ~3);
            ry1 = std::min(h, (int)std::floor(wmax) + 16 + 1);
            if (rx1 < rx0) rx1 = rx0;
            if (ry1 < ry0) ry1 = ry0;
          } else {
            rx1 = (w + 3) & ~3;
          }
        }
        fv[vi].wrect[0] = rx0, fv[vi].wrect[1] = ry0, fv[vi].wrect[2] = rx1, fv[vi].wrect[3] = ry1;
        max_quads = std::max(max_quads, ((rx1 - rx0) / 4) * (ry1 - ry0));
      }
    }
    // (the scratch may still be read by the previous launch: the copy and the kernel are queued behind it on the stream)
    VCY_HIP_CHECK(hipMemcpyAsync(d_views, fv, fv_bytes, hipMemcpyHostToDevice, c->stream));
    VCY_HIP_CHECK(hipEventRecord(c->ev_fused_stage[c->fused_stage_idx], c->stream));
    hipLaunchKernelGGL(c0_records_kernel, dim3((unsigned)((nbw * WX + 255) / 256), (unsigned)n_views), dim3(256), 0, c->stream,
                       d_views, c->d_px, c->nx, nbw, d_c2);
    VCY_HIP_CHECK(hipGetLastError());
    c->fused_cache_vp.assign((const char*)vp, (const char*)vp + vp_bytes);
    c->fused_cache_wmax = c->d_wmax;
    c->fused_cache_bound = need_bound;
    c->fused_cache_lower = need_lower;
    c->fused_cache_ortho = c->fused_ortho;
    c->fused_cache_at = (void*)d_c2;
    c->fused_cache_z[0] = zlo, c->fused_cache_z[1] = zhi;
    c->fused_cache_samef = samef;
    c->fused_cache_max_quads = max_quads;
    c->fused_cache_valid = true;
  }
  out->d_c2 = d_c2;
  out->d_views = d_views;
  out->samef = c->fused_cache_samef;
  out->max_quads = c->fused_cache_max_quads;
  return VCY_OK;
}


void build_window_planes(vcy_ctx* c, const PreparedViews& pv, int n_views, bool need_lower) {
  if (pv.max_quads <= 0) return;
  const dim3 wgrid((unsigned)((pv.max_quads + 255) / 256), (unsigned)n_views);
  hipLaunchKernelGGL(wmax_k4_kernel<false>, wgrid, dim3(256), 0, c->stream, pv.d_views);
  hipLaunchKernelGGL(wmax_k8_kernel<false>, wgrid, dim3(256), 0, c->stream, pv.d_views);
  if (need_lower) {
    hipLaunchKernelGGL(wmax_k4_kernel<true>, wgrid, dim3(256), 0, c->stream, pv.d_views);
    hipLaunchKernelGGL(wmax_k8_kernel<true>, wgrid, dim3(256), 0, c->stream, pv.d_views);
  }
}

// Carves views[0..n_views) (n_views <= 32, max_sdf already resolved) in one launch.
int launch_carve_fused(vcy_ctx* c, const GridParams& g, int n_views, const ViewParams* vp) {
  const vcy_update_option& u = c->opt.update_option;
  const int nzl = c->nz_local();
  const int nxp = (c->nx + WX - 1) / WX * WX, nbw = nxp / WX;
  const bool need_bound = c->use_cull && (u.voxel_update == VCY_UPDATE_MAX || u.use_truncation);
  // (two more planes, of the negated image, for the truncating unit-weight average: FusedView::has_lower)
  const bool need_lower = need_bound && u.use_truncation && u.voxel_update == VCY_UPDATE_WEIGHTED_AVERAGE;
  PreparedViews pv;
  {
    const int rcp = prepare_views(c, n_views, vp, need_bound, need_lower, c->z0, c->z1, &pv);
    if (rcp != VCY_OK) return rcp;
  }
  float* d_c2 = pv.d_c2;
  FusedView* d_views = pv.d_views;
  const bool samef = pv.samef;

  // ("carvetimer" 1: every chunk of every fused launch leaves three events in the context's log -- before what runs
  // ahead of the carve kernel (window maxima, pre-pass, live list), before the carve kernel, after it -- read WITHOUT
  // having synchronised in between by vcy_carve_log; vcy_last_carve_ms sums the last launch's chunks)
  const bool timed = c->time_carve;
  int stamp = -1;
  // (a launch that fails between opening a record and its last event leaves no half-recorded triplet behind:
  // vcy_carve_log would fail on it, or report the times of whatever used those events before)
  struct LogGuard {
    vcy_ctx* c;
    int n0, last0;
    bool done;
    ~LogGuard() { if (!done) c->carve_log_n = n0, c->carve_log_last = last0; }
  } log_guard{c, c->carve_log_n, c->carve_log_last, false};
  if (timed) {
    stamp = carve_log_open(c, true);
    if (stamp >= 0) VCY_HIP_CHECK(hipEventRecord(c->carve_log[stamp].ev[0], c->stream));
  }
  build_window_planes(c, pv, n_views, need_lower);  // the images may have changed since the last call: rebuild every time
  const int nbx = (c->nx + BX - 1) / BX, nby = (c->ny + BY - 1) / BY, nbz = (nzl + BZ - 1) / BZ;
  if ((int64_t)nbx * nby * nbz > 0x7fffffffLL) {
    set_error("slab too large for one launch");
    return VCY_ERR_TOO_MANY_VOXELS;
  }
  ModeParams m{u.voxel_update, u.sdf_interp, u.update_outside, u.use_truncation ? 1 : 0, c->fused_ortho ? 1 : 0, 0};
  // shortest division sequence that is exact for every focal length of this batch (checked on the device)
  if (!c->fused_ortho && c->use_short_div) {
    int level = 2;
    for (int vi = 0; vi < n_views && level > 0; ++vi) {
      level = std::min(level, div_level(c, vp[vi].fx));
      if (vp[vi].fy != vp[vi].fx && level > 0) level = std::min(level, div_level(c, vp[vi].fy));
    }
    m.div_level = level;
  }
  c->last_div_level = m.div_level;
  // update_num can only exceed voxel_max_update_num after more than that many views
  const bool checkmax = c->views_carved + n_views > (int64_t)u.voxel_max_update_num;
  // Tile kind: footprint of an 8x8x8 wave brick in pixels ~ (8*sqrt(3)*pixels_per_voxel + 3)^2.  The raw
  // 16 x 16 pixel tile covers voxels up to ~0.85 px; the 2048-pixel tile filled in place up to ~3 px; wider
  // footprints take the generic path inside the kernel either way.
  bool big = false;
  {
    const float res = c->opt.resolution;
    float worst = 0.0f;
    for (int vi = 0; vi < n_views; ++vi) {
      // pixels per voxel at the centre of the slab (bricks much closer to the camera than that
      // overflow the tile and take the generic path on their own)
      const float X = 0.5f * (c->h_px_min + c->h_px_max), Y = 0.5f * (c->h_py_min + c->h_py_max);
      const float Z = 0.5f * (c->h_pz[c->z0] + c->h_pz[c->z1 - 1]);
      const float pz = vp[vi].t[2] + (vp[vi].r[2][0] * X + (vp[vi].r[2][1] * Y + vp[vi].r[2][2] * Z));
      const float f = std::max(vp[vi].fx, vp[vi].fy);
      // orthographic: one pixel per world unit
      worst = std::max(worst, c->fused_ortho ? res : (pz > 0.0f ? f * res / pz : INFINITY));
    }
    const float side = 8.0f * 1.7320508f * worst + 3.0f;
    big = side > 15.0f;
    if (c->tile_mode == 1) big = false;
    if (c->tile_mode == 2) big = true;
  }
  // min(sdf) per wave brick (vcy_ctx::d_brick_min): 4 bytes per 512 voxels, allocated on first use; without it
  // nothing is dropped before the state is read and marching cubes reads every brick
  if (!c->d_brick_min) {
    const size_t bytes = sizeof(float) * (size_t)nbw * nby * nbz;
    if (hipMalloc(&c->d_brick_min, bytes) != hipSuccess) {
      c->d_brick_min = nullptr;
      (void)hipGetLastError();
    }
    c->brick_min_valid = false;
  }
  if (!c->cnt_implied) c->brick_min_valid = false;
  // Not every wave of the launches below rewrites its entry: a wave whose every view lies below the truncation limit
  // returns before it has read anything, a workgroup left off the live list never starts.  That is harmless while the
  // entry was valid on entry (it still is).  When it was not -- the per-view kernel or a vcy_upload has written to the
  // state since, or the array has just been allocated over a slab that is not fresh -- every entry starts from
  // lowest(): "a voxel of this brick may be untouched", which never drops a view and never lets marching cubes skip
  // the brick.  (A fresh slab needs nothing: there every wave runs and stores its minimum.)
  if (c->d_brick_min && !c->fresh && !c->brick_min_valid && c->cnt_implied) {
    const int64_t nb = (int64_t)nbw * nby * nbz;
    hipLaunchKernelGGL(fill_lowest_kernel, dim3((unsigned)std::min<int64_t>((nb + 255) / 256, 4096)), dim3(256), 0,
                       c->stream, c->d_brick_min, nb);
    VCY_HIP_CHECK(hipGetLastError());
  }
  // Cooperative write-back (carve_fused_kernel): pays where a launch moves the state for little arithmetic -- few views
  // over a carved grid (1024^3, one view per launch: weighted average 4.39 -> 3.41 ms, kMax 0.89 -> 0.84), and the first
  // single-view launch on a fresh grid, which stores every voxel once (2.48 -> 1.95 ms: 6.4 GB at 3.3 instead of
  // 2.6 TB/s).  "coopstore": -1 that rule, 0 never, 1 always when the layout allows it: rows of whole bricks, four
  // waves per workgroup, raw tiles.
  // (kMax: only single-view launches -- with 4 or 8 views per launch the waves of a workgroup process different numbers
  // of views and the barrier costs 3 - 4 %, profiles/r04/coop_store_batches.txt; weighted average: up to 8 views, 0 ... +2 %;
  // a fresh grid: single-view launches in either mode -- the fused 32-view launch gained nothing from whole-segment
  // stores in round 3)
  // The few-view flavour (kRowBricks: a wave walks the bricks of a row segment): launches of up to kRowMaxViews views with
  // raw tiles and records from the pre-pass, over rows of whole bricks, no update limit in reach.  "rowkernel" -1 that
  // rule, 0 never (the NB = 1 kernel with its cooperative write-back), n > 0: launches of up to min(n, 8) views.
  const int row_views = c->row_kernel < 0 ? kRowMaxViews : std::min(c->row_kernel, kRowMaxViews);
  const bool rows = !big && !checkmax && (c->nx & (WX - 1)) == 0 && n_views <= row_views && c->prologue_mode != 1;
  const int nseg = (nbw + kRowBricks - 1) / kRowBricks;  // segments per brick row
  // a launch of ONE view through the kernel instance that knows it ("oneview" 0: the general instance)
  const bool one_view = c->one_view && n_views == 1 && !rows && !big && !checkmax && c->prologue_mode != 1;
  const bool coop_ok = (kWgWaves == 4 || kWgWaves == 8) && (c->nx & (WX - 1)) == 0 && !big && !rows;
  const int coop_views = c->fresh ? 1 : (u.voxel_update == VCY_UPDATE_MAX ? 1 : kLiveListMaxViews);
  const bool coop = coop_ok && (c->coop_store > 0 || (c->coop_store < 0 && n_views <= coop_views));
  // "ntstore": streaming stores whenever the cooperative write-back runs (0: never) -- whole 128-byte segments that this
  // launch does not read again: 0.5 - 1.5 % on single-view launches (profiles/r06/nontemporal.txt)
  const bool nt = coop && c->nt_store != 0;
  // (bit 5, "eager" state requests, is decided per chunk below: it depends on whether the launch is a listed one)
  const int state_flags_base = (c->fresh ? 1 : 0) | (c->cnt_implied ? 2 : 0) | (c->brick_min_valid && !c->fresh ? 4 : 0) |
                               (coop ? 8 : 0) | (nt ? 16 : 0);
  // Raw tiles: the footprint records of every (wave brick, view) pair come from a pre-pass (footprint_records_kernel),
  // 8 bytes per pair.  The slab is carved in chunks of whole brick layers so that the records of a chunk stay
  // below kRecordBytesMax (1024^3 x 32 views: 0.5 GiB, one chunk; 2048^3 x 64: nine).
  const int64_t layer_bricks = (int64_t)nbw * nby;
  int chunk_layers = nbz;
  // Footprints from the pre-pass (records) or from the carve kernel's own prologue?  The pre-pass: its threads are all
  // busy where the prologue would use nviews lanes of 64, it runs at full occupancy, and the records are what the live
  // list and the early return read.  Round 5 tried the prologue for the one launch whose records are large -- 2048^3 x 64
  // views: 8.6 GB of them, written and read back in chunks, 9.1 of the step's 81.8 ms; every lane of the prologue has a
  // view there and the step becomes ONE launch -- and measured 92.8 ms: the footprints cost 20 ms in the kernel (two
  // dependent round trips in front of every wave, at the carve kernel's occupancy) against 9 in the pre-pass
  // (profiles/r05/bench_2048x64_config4_*.json).  So: "prologue" 0 or 2 records (chunks of at most kRecordBytesMax: four
  // at that shape), 1 in the kernel.
  const int64_t rec_cap = c->record_bytes_max > 0 ? c->record_bytes_max : kRecordBytesMax;  // ("recordbytes": tests force several chunks)
  const bool in_kernel_prologue = !big && c->prologue_mode == 1;
  if (!big && !in_kernel_prologue) {
    const int64_t per_layer = layer_bricks * n_views * (int64_t)sizeof(FootprintRecord);
    const int64_t cap = rec_cap;
    chunk_layers = (int)std::max<int64_t>(1, std::min<int64_t>(nbz, cap / std::max<int64_t>(per_layer, 1)));
    const size_t need = (size_t)(per_layer * chunk_layers);
    if (c->records_bytes < need) {
      VCY_HIP_CHECK(hipStreamSynchronize(c->stream));
      if (c->d_records) (void)hipFree(c->d_records);
      c->d_records = nullptr;
      c->records_bytes = 0;
      VCY_HIP_CHECK(hipMalloc(&c->d_records, need));
      c->records_bytes = need;
    }
  }
  if (c->count_pairs) {  // "paircount" 1: one counter per brick layer of the slab, cleared by every launch
    if (c->pair_count_layers < nbz) {
      VCY_HIP_CHECK(hipStreamSynchronize(c->stream));
      if (c->d_pair_count) (void)hipFree(c->d_pair_count);
      c->d_pair_count = nullptr;
      c->pair_count_layers = 0;
      VCY_HIP_CHECK(hipMalloc(&c->d_pair_count, sizeof(unsigned long long) * (size_t)nbz));
      c->pair_count_layers = nbz;
    }
    VCY_HIP_CHECK(hipMemsetAsync(c->d_pair_count, 0, sizeof(unsigned long long) * (size_t)nbz, c->stream));
    c->pair_count_views = n_views;
  }
  const bool gen = m.ortho != 0 || m.interp == VCY_INTERP_NN;
  const bool want_lower = need_bound && u.use_truncation && u.voxel_update != VCY_UPDATE_MAX;
  for (int l0 = 0; l0 < nbz; l0 += chunk_layers) {
    const int layers = std::min(chunk_layers, nbz - l0);
    GridParams gc = g;  // this chunk: brick layers [l0, l0 + layers)
    gc.sdf = g.sdf + (int64_t)l0 * BZ * c->slice;
    gc.cnt = (char*)g.cnt + (int64_t)l0 * BZ * c->slice * c->cnt_bytes;
    gc.z0 = g.z0 + l0 * BZ;
    gc.nz_local = std::min(layers * BZ, nzl - l0 * BZ);
    const int64_t nbricks = layer_bricks * layers;
    FootprintRecord* recs = in_kernel_prologue ? nullptr : (FootprintRecord*)c->d_records;
    float* bmin = c->d_brick_min ? c->d_brick_min + (int64_t)l0 * layer_bricks : nullptr;
    unsigned long long* pcnt = c->count_pairs && c->d_pair_count ? c->d_pair_count + l0 : nullptr;  // ("paircount" 1)
    if (timed && l0 > 0) {
      stamp = carve_log_open(c, false);
      if (stamp >= 0) VCY_HIP_CHECK(hipEventRecord(c->carve_log[stamp].ev[0], c->stream));
    }
    if (!big && !in_kernel_prologue) {
      const dim3 pgrid((unsigned)((nbricks + 255) / 256), (unsigned)n_views);
      const FastDivU32 div_nbw = make_fast_div_u32((uint32_t)nbw), div_nby = make_fast_div_u32((uint32_t)nby);
#define VCY_PREPASS(SF, GN)                                                                                       \
  hipLaunchKernelGGL((footprint_records_kernel<SF, GN>), pgrid, dim3(256), 0, c->stream, gc, d_views, nbw, nby,   \
                     nbricks, m, need_bound ? 1 : 0, want_lower ? 1 : 0, recs, div_nbw, div_nby,                  \
                     nbricks < 0xffffffffLL ? 1 : 0)
      if (samef) { if (gen) VCY_PREPASS(true, true); else VCY_PREPASS(true, false); }
      else { if (gen) VCY_PREPASS(false, true); else VCY_PREPASS(false, false); }
#undef VCY_PREPASS
      VCY_HIP_CHECK(hipGetLastError());
    }
    // units of the launch: workgroup blocks of kWgWaves bricks, or the segments of the few-view flavour (one per wave)
    const int unit_bricks = rows ? kRowBricks : kWgWaves;
    const int units_x = rows ? nseg : nbx;
    const dim3 grid((unsigned)((int64_t)units_x * nby * layers));
    auto groups_of = [&](unsigned units) {  // workgroups that hold `units` listed units
      return rows ? dim3((units + kRowWaves - 1) / kRowWaves) : dim3(units);
    };
    // (unlisted launch of the few-view flavour: unit (g mod 8) + 8 (kRowWaves (g / 8) + wave) of workgroup g)
    dim3 launch_grid = rows ? dim3(8u * ((grid.x + 8u * kRowWaves - 1) / (8u * kRowWaves))) : grid;
    // few views over a carved grid: only the workgroups with a live (brick, view) pair (live_workgroups_kernel)
    const int* wgl = nullptr;
    int list_entry_words = 0;
    const bool have_min = u.voxel_update == VCY_UPDATE_MAX && (state_flags_base & 4) != 0 && bmin != nullptr;
    // (not when the list of the previous such launch held most workgroups anyway -- a weighted-average carve touches
    // nearly every brick with every view, and the list pass is then 4 % on top; the count arrives by an asynchronous
    // copy into page-locked memory and is only a hint: reading an older value is harmless)
    const bool list_pays = c->h_live_hint == nullptr || c->h_live_hint[1] <= 0 ||
                           (double)c->h_live_hint[0] < 0.6 * (double)c->h_live_hint[1];
    ++c->live_list_age;
    if (!big && recs != nullptr && c->use_live_list && need_bound && !c->fresh && n_views <= kLiveListMaxViews && (m.trunc != 0 || have_min) &&
        (list_pays || c->live_list_age % 16 == 0)) {  // (every 16th launch looks again)
      const int nwg = (int)grid.x;
      // (a launch of ONE view: entries {id, live waves, the four records} -- "listrecords" 0: ids only)
      list_entry_words = one_view && c->list_records != 0 ? kLiveEntryWords : 0;
      const size_t need = list_entry_words ? sizeof(int) * (2 + (size_t)nwg * kLiveEntryWords) : sizeof(int) * ((size_t)nwg + 1);  // (the hint below: a race with its copy is benign, it only
      // decides whether the NEXT launch lists its workgroups; with several chunks it reflects the last one)
      if (c->wg_list_bytes < need) {
        VCY_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (c->d_wg_list) (void)hipFree(c->d_wg_list);
        c->d_wg_list = nullptr;
        c->wg_list_bytes = 0;
        VCY_HIP_CHECK(hipMalloc(&c->d_wg_list, need));
        c->wg_list_bytes = need;
      }
      VCY_HIP_CHECK(hipMemsetAsync(c->d_wg_list, 0, sizeof(int), c->stream));
      hipLaunchKernelGGL(live_workgroups_kernel, dim3((unsigned)((nwg + kLiveThreads - 1) / kLiveThreads)), dim3(kLiveThreads), 0, c->stream, recs, nbricks,
                         n_views, have_min ? bmin : nullptr, m.trunc, units_x, nby, nbw, nwg, c->d_wg_list, unit_bricks,
                         list_entry_words);
      launch_grid = groups_of((unsigned)nwg);  // (every unit started unless the count below arrives)
      VCY_HIP_CHECK(hipGetLastError());
      wgl = c->d_wg_list;
      if (!c->h_live_hint && hipHostMalloc((void**)&c->h_live_hint, 2 * sizeof(int), hipHostMallocDefault) != hipSuccess) {
        c->h_live_hint = nullptr;
        (void)hipGetLastError();
      }
      if (c->h_live_hint) {
        c->h_live_hint[1] = nwg;
        (void)hipMemcpyAsync(&c->h_live_hint[0], c->d_wg_list, sizeof(int), hipMemcpyDeviceToHost, c->stream);
        // The carve kernel is launched over the listed workgroups only: the host waits for the count (the list pass is
        // 60 us of device time behind it) instead of starting every workgroup of the slab to have all but the listed
        // ones read list[0] and leave -- 512 K workgroups that do nothing else are 0.115 ms at 1024^3
        // (profiles/tools/per_view_floor.py), a sixth of a single-view launch.  ("livesync" 0: the full grid, no wait.)
        if (c->live_sync && hipStreamSynchronize(c->stream) == hipSuccess) {
          const int live = c->h_live_hint[0];
          if (live >= 0 && live <= nwg) launch_grid = groups_of((unsigned)live);
        } else {
          (void)hipGetLastError();
        }
      }
    }
    // "eagerstate": the workgroups of a one-view launch over a carved grid request their bricks' state next to the
    // footprint records when nearly all of them will need it: a listed launch (only live workgroups are started), or one
    // that skipped its list because the last one held most workgroups (-1 that rule, 0 never, 1 every one-view launch).
    // 1024^3, one view per launch: weighted average 2.51 -> 2.46 ms, kMax 0.521 -> 0.487 (profiles/r06/eager_state.txt).
    const bool nearly_all_live = wgl != nullptr ? launch_grid.x < grid.x || !list_pays : !list_pays;
    const bool eager_state = !c->fresh && one_view &&
                             (c->eager_state > 0 || (c->eager_state < 0 && nearly_all_live));
    const int state_flags = state_flags_base | (eager_state ? 32 : 0) | (wgl != nullptr && list_entry_words ? 64 : 0);
    if (stamp >= 0) VCY_HIP_CHECK(hipEventRecord(c->carve_log[stamp].ev[1], c->stream));
    if (launch_grid.x == 0) {
      // (no workgroup is live: nothing to launch)
    } else if (c->cnt_bytes == 1)
      launch_fused_counts8(big, u.voxel_update, m.trunc != 0, samef, checkmax, launch_grid.x, c->stream, gc, d_views,
                           d_c2, n_views, m, units_x, nby, c->use_cull ? 1 : 0, state_flags, recs, nbricks, bmin, wgl, pcnt,
                           rows ? (int)grid.x : (one_view ? -1 : 0));
    else
      launch_fused_counts16(big, u.voxel_update, m.trunc != 0, samef, checkmax, launch_grid.x, c->stream, gc, d_views,
                            d_c2, n_views, m, units_x, nby, c->use_cull ? 1 : 0, state_flags, recs, nbricks, bmin, wgl, pcnt,
                            rows ? (int)grid.x : (one_view ? -1 : 0));
    VCY_HIP_CHECK(hipGetLastError());
    if (stamp >= 0) VCY_HIP_CHECK(hipEventRecord(c->carve_log[stamp].ev[2], c->stream));
  }
  log_guard.done = true;
  c->fresh = false;  // the launches store every voxel of a fresh slab
  // (every wave that ran to its end wrote its entry; the others' entries were valid on entry or hold lowest(), see above)
  c->brick_min_valid = c->d_brick_min != nullptr && c->cnt_implied;
  return VCY_OK;
}

int fused_max_views() { return kMaxFusedViews; }

// ---- slab planner ---------------------------------------------------------------------------------------------
// What a brick layer of the grid will cost a fused carve of these views, BEFORE any slab exists: with view dropping the
// layers through the object cost 1.6x the outer ones, so z-slabs of equal thickness leave the GPUs of a node unequally
// loaded (the slowest rank of eight took 1.24x the mean).  The carve kernel's time is, to a good approximation, a fixed
// cost per wave brick plus a cost per (brick, view) pair it processes.  Which pairs it processes is decided by bounds:
// a view is dropped for a brick when every sample lies below the truncation limit, or (kMax, every voxel touched) not
// above the brick's current minimum.  The first is static.  The second depends on the state -- but min(sdf) of a brick
// after the views processed so far is at least the largest LOWER bound of their samples, so the same window planes
// that give the upper bounds (built here for the negated images as well) let one thread play the kernel's decisions
// for a brick: "processed" is counted where ub > the running maximum of the lower bounds.  That over-counts only views
// whose samples lie within one footprint's variation of the running maximum.  One thread per SAMPLED brick (every
// `stride`-th in x and y, every layer), the views in sequence; the counts are summed per layer.
template <bool SAMEF, bool GEN>
__global__ __launch_bounds__(256) void plan_cost_kernel(GridParams g, const FusedView* __restrict__ views, int nviews,
                                                        int nbw, int nby, int sxn, int syn, int stride, int64_t nsample,
                                                        ModeParams mode, unsigned long long* __restrict__ layer_pairs) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool valid = t < nsample;
  const int64_t tt = valid ? t : nsample - 1;  // (idle lanes repeat the last sample: footprint_of uses wave votes)
  const int sx = (int)(tt % sxn);
  const int64_t r = tt / sxn;
  const int sy = (int)(r % syn), bz = (int)(r / syn);
  const int bxw = min(sx * stride + stride / 2, nbw - 1), by = min(sy * stride + stride / 2, nby - 1);
  const int x_lo = min(bxw * WX, g.nx - 1), x_hi = min(bxw * WX + WX - 1, g.nx - 1);
  const int y_hi = min(by * BY + BY - 1, g.ny - 1), z_hi = min(bz * BZ + BZ - 1, g.nz_local - 1);
  const float xl = g.px[x_lo], xh = g.px[x_hi], yl = g.py[by * BY], yh = g.py[y_hi];
  const float zl = g.pz[g.z0 + bz * BZ], zh = g.pz[g.z0 + z_hi];
  const bool update_max = mode.update == VCY_UPDATE_MAX;
  float smin = -INFINITY;   // lower bound of min(sdf) of the brick once every voxel is touched
  bool touched = false;
  int count = 0;
  for (int vi = 0; vi < nviews; ++vi) {
    float lb = -INFINITY;
    const TileInfo ti = footprint_of<SAMEF, kTileRaw, GEN>(views[vi], xl, xh, yl, yh, zl, zh, mode.ortho != 0,
                                                            mode.outside == VCY_OUTSIDE_MAX, true, true, &lb);
    const bool drop = (mode.trunc != 0 && ti.ub < -1.0f) || (update_max && touched && ti.ub <= smin);
    if (!drop) {
      ++count;
      if (update_max && (ti.sure & 1)) {  // (a `sure` view touches every voxel of the brick)
        smin = touched ? fmaxf(smin, lb) : lb;
        touched = true;
      }
    }
  }
  if (!valid) count = 0;
  const int bz0 = __shfl(bz, 0, 64);
  if (__all(bz == bz0)) {  // (the usual case: one atomic per wave)
    for (int d = 32; d > 0; d >>= 1) count += __shfl_down(count, d, 64);
    if ((threadIdx.x & 63) == 0 && count) atomicAdd(&layer_pairs[bz0], (unsigned long long)count);
  } else if (count) {
    atomicAdd(&layer_pairs[bz], (unsigned long long)count);
  }
}

// pairs[l] = estimated (brick, view) pairs a fused carve of these views processes in brick layer l of the WHOLE grid,
// scaled from the sampled bricks to the layer's bricks; bricks_per_layer as the carve kernel counts them.
int plan_layer_pairs(vcy_ctx* c, int n_views, const ViewParams* vp, int stride, std::vector<double>* pairs,
                     int64_t* bricks_per_layer) {
  const vcy_update_option& u = c->opt.update_option;
  const int nxp = (c->nx + WX - 1) / WX * WX, nbw = nxp / WX, nby = (c->ny + BY - 1) / BY, nbz = (c->nz + BZ - 1) / BZ;
  *bricks_per_layer = (int64_t)nbw * nby;
  pairs->assign((size_t)nbz, (double)n_views * (double)nbw * nby);  // nothing dropped: every pair
  const bool drops = c->use_cull && (u.voxel_update == VCY_UPDATE_MAX || u.use_truncation);
  if (!drops) return VCY_OK;
  if (stride < 1) stride = 1;
  PreparedViews pv;
  {
    const int rcp = prepare_views(c, n_views, vp, true, true, 0, c->nz, &pv);
    if (rcp != VCY_OK) return rcp;
  }
  if (pv.max_quads <= 0) return VCY_OK;  // no memory for the planes: no estimate, equal layers
  build_window_planes(c, pv, n_views, true);
  GridParams g;
  std::memset(&g, 0, sizeof(g));
  g.px = c->d_px, g.py = c->d_py, g.pz = c->d_pz;
  g.nx = c->nx, g.ny = c->ny, g.z0 = 0, g.nz_local = c->nz;
  ModeParams m{u.voxel_update, u.sdf_interp, u.update_outside, u.use_truncation ? 1 : 0, c->fused_ortho ? 1 : 0, 0};
  const int sxn = (nbw + stride - 1) / stride, syn = (nby + stride - 1) / stride;
  const int64_t nsample = (int64_t)sxn * syn * nbz;
  unsigned long long* d_out = nullptr;
  VCY_HIP_CHECK(hipMalloc(&d_out, sizeof(unsigned long long) * (size_t)nbz));
  std::vector<unsigned long long> h((size_t)nbz, 0ull);
  hipError_t e = hipMemsetAsync(d_out, 0, sizeof(unsigned long long) * (size_t)nbz, c->stream);
  if (e == hipSuccess) {
    const bool gen = m.ortho != 0 || m.interp == VCY_INTERP_NN;
    const dim3 grid((unsigned)((nsample + 255) / 256));
#define VCY_PLAN(SF, GN)                                                                                           \
  hipLaunchKernelGGL((plan_cost_kernel<SF, GN>), grid, dim3(256), 0, c->stream, g, pv.d_views, n_views, nbw, nby, \
                     sxn, syn, stride, nsample, m, d_out)
    if (pv.samef) { if (gen) VCY_PLAN(true, true); else VCY_PLAN(true, false); }
    else { if (gen) VCY_PLAN(false, true); else VCY_PLAN(false, false); }
#undef VCY_PLAN
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_out, sizeof(unsigned long long) * (size_t)nbz, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d_out);
  if (e != hipSuccess) {
    set_error("slab planner: %s", hipGetErrorString(e));
    return VCY_ERR_HIP;
  }
  const double scale = (double)nbw * nby / ((double)sxn * syn);
  for (int l = 0; l < nbz; ++l) (*pairs)[(size_t)l] = (double)h[(size_t)l] * scale;
  return VCY_OK;
}


namespace {
__global__ void selftest_rcp_count_kernel(int* n_bad) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x + 1;  // 1 .. 65536
  const float x = (float)m;
  if (rcp_count(x) != 1.0f / x) atomicAdd(n_bad, 1);  // IEEE division (-fhip-fp32-correctly-rounded-divide-sqrt)
}
}  // namespace

// Device-side identities the fast paths of the fused kernel rest on; VCY_OK when all hold.
int selftest_fused(hipStream_t stream) {
  int* d_bad = nullptr;
  int h_bad = -1;
  VCY_HIP_CHECK(hipMalloc(&d_bad, sizeof(int)));
  hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(int), stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(selftest_rcp_count_kernel, dim3(256), dim3(256), 0, stream, d_bad);
    e = hipMemcpyAsync(&h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(d_bad);
  if (e != hipSuccess) {
    set_error("self test failed to run: %s", hipGetErrorString(e));
    return VCY_ERR_HIP;
  }
  if (h_bad != 0) {
    set_error("self test: rcp_count differs from IEEE division for %d of 65536 counts", h_bad);
    return VCY_ERR_INTERNAL;
  }
  return VCY_OK;
}

}  // namespace vcy

#endif  // VCY_FUSED_PART
