// K2-K4: marching cubes on the device slab, reproducing the reference's serial scan.
//
// Replaces MarchingCubes() (reference src/vacancy/marching_cubes.cc:63-228).  The reference
// walks cells z,y,x from 1, deduplicates vertices with a std::map keyed by the voxel-id pair
// of the cut edge, numbers vertices in order of first reference and faces in scan order.
// The same numbering falls out of data-parallel passes in raster order:
//
//   bits      ONE streaming pass over the voxels (the only pass that touches the 4-6 B/voxel
//             state): a wave ballots 64 consecutive x into three bit planes --
//             IN  = sdf < iso_level          (marching_cubes.cc:121-128, float promoted to double)
//             OK  = sdf != InvalidSdf::kVal  (:103-112)
//             TC  = update_num >= 1          (:88-90, tested on corner 6 only)
//   active    one thread per 64-cell word: the 8 corner planes of the cells are word shifts /
//             row offsets of IN and OK; active = valid & ~all_inside & any_inside.
//   (sweep    on request, "mcsweep" 1: bits + active in ONE kernel that walks z with the planes of two slices
//             in LDS; fewer bytes, not faster -- see mc_sweep_kernel.)
//   owner     a cut edge belongs to the FIRST active cell (scan order) among the <= 4 cells
//             that share it -- where the reference's map insert happens, which also fixes the
//             interpolation direction.  Active cells (sparse) find the
//             edges they own from 9 neighbour ACT bits; per-word counts -> wave/block
//             prefix sums -> per-block (vertex, triangle) totals.
//   scan      exclusive scan of the per-block totals.
//   emit      owned edges -> VertexInterp in fp64 (:25-57) -> vertices + edge keys;
//             triangles -> vertex ids through the owner cell of every corner edge.
//
// Multi-GPU: the layer of cells below the slab (z = z0-1, owned by the previous rank) is
// evaluated from the two halo slices as a ghost layer; vertices it owns on the shared plane
// are emitted first and counted in n_foreign_vertices, so a host merge can map them onto the
// previous rank's numbering by edge key.
//
// Memory-bound, no MFMA: 4 B (sdf) per cell algorithmic; real traffic = one read of sdf +
// update_num, everything else is 1 bit per voxel/cell, plus 12 B per vertex/triangle out.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "vcy_internal.h"

namespace vcy {

namespace {

typedef unsigned long long u64;

// ---- tables ------------------------------------------------------------------------------
const char* const kCaseStrings[256] = {
#include "vacancy_mc_cases.inc"
};

struct McTables {
  int8_t tri[256][16];     // edge numbers, -1 terminated (reference kTriTable)
  uint8_t ntri[256];
  uint16_t prec[256][12];  // prec[c][e] = edges whose vertex the serial scan creates before e's
};

void build_tables(McTables* t) {
  std::memset(t, 0, sizeof(*t));
  for (int c = 0; c < 256; ++c) {
    const char* s = kCaseStrings[c];
    int n = 0;
    for (; s[n]; ++n) t->tri[c][n] = (int8_t)((s[n] <= '9') ? s[n] - '0' : s[n] - 'a' + 10);
    for (int k = n; k < 16; ++k) t->tri[c][k] = -1;
    t->ntri[c] = (uint8_t)(n / 3);
    // creation order: triangles in table order, corners j=0..2 read entry i+(2-j)
    // (marching_cubes.cc:199-206)
    uint16_t seen = 0;
    for (int i = 0; i < n; i += 3)
      for (int j = 0; j < 3; ++j) {
        const int e = t->tri[c][i + (2 - j)];
        if (!(seen & (1 << e))) {
          t->prec[c][e] = seen;
          seen |= (uint16_t)(1 << e);
        }
      }
  }
}

// corner offsets relative to the cell's max corner (x,y,z), marching_cubes.cc:93-101
__device__ const int8_t kCornerOff[8][3] = {{-1, -1, -1}, {0, -1, -1}, {0, 0, -1}, {-1, 0, -1},
                                            {-1, -1, 0},  {0, -1, 0},  {0, 0, 0},  {-1, 0, 0}};
// interpolation argument order per edge (:138-197) and key order (always lower id first)
__device__ const int8_t kEdgeA[12] = {0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3};
__device__ const int8_t kEdgeB[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};
__device__ const int8_t kKeyA[12] = {0, 1, 3, 0, 4, 5, 7, 4, 0, 1, 2, 3};
__device__ const int8_t kKeyB[12] = {1, 2, 2, 3, 5, 6, 6, 7, 4, 5, 6, 7};

// For edge e of a cell: the earlier cells sharing it, in scan order, as (dx,dy,dl) and the
// number the edge has inside that cell.  n = 0 means the cell itself always owns it.
struct Share { int8_t n; int8_t d[3][3]; int8_t e[3]; };
__device__ const Share kShare[12] = {
    {3, {{0, -1, -1}, {0, 0, -1}, {0, -1, 0}}, {6, 4, 2}},    // e0
    {2, {{0, 0, -1}, {1, 0, -1}, {0, 0, 0}}, {5, 7, 0}},      // e1
    {2, {{0, 0, -1}, {0, 1, -1}, {0, 0, 0}}, {6, 4, 0}},      // e2
    {3, {{-1, 0, -1}, {0, 0, -1}, {-1, 0, 0}}, {5, 7, 1}},    // e3
    {1, {{0, -1, 0}, {0, 0, 0}, {0, 0, 0}}, {6, 0, 0}},       // e4
    {0, {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, {0, 0, 0}},        // e5
    {0, {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, {0, 0, 0}},        // e6
    {1, {{-1, 0, 0}, {0, 0, 0}, {0, 0, 0}}, {5, 0, 0}},       // e7
    {3, {{-1, -1, 0}, {0, -1, 0}, {-1, 0, 0}}, {10, 11, 9}},  // e8
    {2, {{0, -1, 0}, {1, -1, 0}, {0, 0, 0}}, {10, 11, 0}},    // e9
    {0, {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, {0, 0, 0}},        // e10
    {1, {{-1, 0, 0}, {0, 0, 0}, {0, 0, 0}}, {10, 0, 0}},      // e11
};

// The nine distinct neighbour cells kShare refers to, and kShare[e].d[k] as an index into them.
constexpr int kNbrCount = 9;
__device__ const int8_t kNbr[kNbrCount][3] = {{0, -1, -1}, {0, 0, -1}, {0, -1, 0}, {1, 0, -1}, {0, 1, -1},
                                              {-1, 0, -1}, {-1, 0, 0}, {-1, -1, 0}, {1, -1, 0}};
__device__ const int8_t kShareNbr[12][3] = {{0, 1, 2}, {1, 3, 0}, {1, 4, 0}, {5, 1, 6}, {2, 0, 0}, {0, 0, 0},
                                            {0, 0, 0}, {6, 0, 0}, {7, 2, 6}, {2, 8, 0}, {0, 0, 0}, {6, 0, 0}};

// Cell (x, y, z) is named by its max corner; bit b of word w of a row is x = 64*w + b.
// Cell rows: layer li = 0 is the ghost layer (z = zc0-1), li = l+1 the slab's own layer l;
// word index of (li, cy = y-1, w):  li == 0 ? cy*Wr + w : G + ((li-1)*Yc + cy)*Wr + w,
// Yc >= Y = rows a layer takes in the cell-word arrays (the sweep pads a layer to whole row groups, so that
// a group is a whole number of 256-word blocks; padding rows hold no active cell),
// G = ghost words rounded up to a whole block so that own cells start on a block boundary.
// Exact n / d for 32-bit unsigned n (Granlund-Montgomery): three integer instructions instead of the
// long 64-bit division sequence.
struct FastDiv {
  uint32_t d, m, s1, s2;
};

struct McParams {
  const float* sdf;   // slab incl. halo slices
  const void* cnt;
  const float* px;
  const float* py;
  const float* pz;
  const u64* in;      // bit planes [slice][y][Wr]
  const u64* ok;
  const u64* tc;
  int nx, ny;
  int nslices;        // stored voxel slices
  int Wr;             // 64-bit words per row
  int Y;              // cell rows per layer = ny-1
  int Yc;             // rows per layer in the cell-word arrays (>= Y)
  int L;              // own cell layers
  int zc0;            // global z of own layer 0
  int zs0;            // global z of stored slice 0
  int has_ghost;
  int64_t G;          // words reserved for the ghost layer
  int64_t nwords;     // G + L*Yc*Wr
  double iso;
  int linear;
  FastDiv div_row, div_layer;  // by Wr and by Yc * Wr; used when small32 (every word index < 2^32)
  int small32;
};

constexpr int kWordsPerBlock = 256;

// ---- pass 0: bit planes ----------------------------------------------------------------------
// One wave turns kBitsWordsPerWave consecutive 64-voxel words into plane words; all loads of a
// wave are issued before the first ballot so that a wave keeps 4 KB in flight (16 words: 1.32 ms per extraction at 1024^3 against 1.39 with 8 and 1.45 with 32).
#ifndef VCY_BITS_WORDS
#define VCY_BITS_WORDS 16
#endif
constexpr int kBitsWordsPerWave = VCY_BITS_WORDS;

// TC_FROM_OK: the state has only ever been written by the grid fill and the carve kernels, where
// update_num == 0 implies sdf == lowest(); then OK(corner 6) already implies TC(corner 6) and the TC
// plane may be any superset of OK -- OK itself, without reading update_num at all (see extract_iso).
template <typename CountT, bool ISO_F32, bool TC_FROM_OK>
__global__ __launch_bounds__(256) void mc_bits_kernel(const float* __restrict__ sdf,
                                                      const CountT* __restrict__ cnt, int nx, int Wr,
                                                      int64_t nwords, double iso, u64* __restrict__ in,
                                                      u64* __restrict__ ok, u64* __restrict__ tc) {
  const int lane = threadIdx.x & 63;
  const int64_t first = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * kBitsWordsPerWave;
  if (first >= nwords) return;
  u64 m_in = 0, m_ok = 0, m_tc = 0;
  if (nx == Wr * 64 && first + kBitsWordsPerWave <= nwords) {
    // rows are whole words: voxel index == word * 64 + lane, plain streaming
    const float* __restrict__ ps = sdf + first * 64 + lane;
    const CountT* __restrict__ pc = cnt + first * 64 + lane;
    float s[kBitsWordsPerWave];
    int n[kBitsWordsPerWave];
#pragma unroll
    for (int k = 0; k < kBitsWordsPerWave; ++k) {
      s[k] = __builtin_nontemporal_load(ps + k * 64);  // streamed once: keep it out of the caches (-4 %)
      n[k] = TC_FROM_OK ? 1 : (int)pc[k * 64];
    }
#pragma unroll
    for (int k = 0; k < kBitsWordsPerWave; ++k) {
      // `sdf < iso_level` promotes the float to double (marching_cubes.cc:121-128); when iso_level
      // is itself a float value the comparison is the same in single precision
      const u64 a = __ballot(ISO_F32 ? s[k] < (float)iso : (double)s[k] < iso);
      const u64 b = __ballot(s[k] != kInvalidSdf);
      const u64 c = TC_FROM_OK ? b : __ballot(n[k] >= 1);
      const bool mine = lane == k;
      m_in = mine ? a : m_in;
      m_ok = mine ? b : m_ok;
      m_tc = mine ? c : m_tc;
    }
  } else {
    // general rows (nx not a multiple of 64, or the tail of the array)
    int64_t row = first / Wr;
    int w = (int)(first - row * Wr);
    for (int k = 0; k < kBitsWordsPerWave; ++k) {
      const int x = w * 64 + lane;
      const bool live = first + k < nwords && x < nx;
      float sv = kInvalidSdf;
      int nv = 0;
      if (live) {
        sv = sdf[row * nx + x];
        nv = TC_FROM_OK ? 1 : (int)cnt[row * nx + x];
      }
      const u64 a = __ballot(live && (ISO_F32 ? sv < (float)iso : (double)sv < iso));
      const u64 b = __ballot(live && sv != kInvalidSdf);
      const u64 c = TC_FROM_OK ? b : __ballot(live && nv >= 1);
      if (lane == k) {
        m_in = a;
        m_ok = b;
        m_tc = c;
      }
      if (++w == Wr) {
        w = 0;
        ++row;
      }
    }
  }
  if (lane < kBitsWordsPerWave && first + lane < nwords) {
    in[first + lane] = m_in;
    ok[first + lane] = m_ok;
    if (tc != nullptr) tc[first + lane] = m_tc;  // null: the TC plane is the OK plane itself (extract_iso)
  }
}

// ---- pass 0 with the brick minima: one workgroup per row of bricks ---------------------------------
// The fused carve keeps min(sdf) of every 8 x 8 x 8 brick (vcy_ctx::d_brick_min; lowest() while a voxel of the
// brick is untouched).  A voxel of a brick whose minimum is above the iso level -- and not lowest() -- is outside
// and valid whatever it holds exactly, which is all these planes record (IN = 0, OK = 1), so it need not be read:
// in a carved grid that is most of the volume.  Testing this per wave of mc_bits_kernel costs more than it saves
// (a dependent memory round trip in front of every 1024 voxels: 1.51 ms per extraction at 1024^3 against 1.31
// reading everything), so here a WORKGROUP takes one row of bricks (by, bz) = 64 voxel rows: its threads fetch the
// row's minima once, ballot them into a bit mask in LDS, and every wave then walks 16 of the 64 rows with loads
// predicated by bits it already holds -- whole 64-voxel words of skipped bricks cost no instruction at all.
// For rows that are whole words (nx == 64 Wr) of an owned slab whose state implies TC == OK.
constexpr int kBricksMaxNbw = 1024;  // bricks along x (nx <= 8192)
template <bool ISO_F32>
__global__ __launch_bounds__(256) void mc_bits_bricks_kernel(const float* __restrict__ sdf, int ny, int nz, int Wr,
                                                             double iso, u64* __restrict__ in, u64* __restrict__ ok,
                                                             u64* __restrict__ tc, const float* __restrict__ bmin,
                                                             int nbw, int nby) {
  __shared__ u64 skip64[kBricksMaxNbw / 64 + 2];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int by = blockIdx.x, bz = blockIdx.y;
  // which bricks of this row need no read
  const float* __restrict__ brow = bmin + ((int64_t)bz * nby + by) * nbw;
  for (int b0 = 0; b0 < nbw; b0 += 256) {  // (uniform)
    const int b = b0 + (int)threadIdx.x;
    bool outside = false;
    if (b < nbw) {
      const float bm = brow[b];
      outside = (ISO_F32 ? bm > (float)iso : (double)bm > iso) && bm > kInvalidSdf;
    }
    const u64 m = __ballot(outside);
    if (lane == 0) skip64[(b0 >> 6) + wave] = m;
  }
  if (threadIdx.x < 2) skip64[((nbw + 255) / 256) * 4 + threadIdx.x] = 0ull;  // (read as the second half of the last chunk)
  __syncthreads();
  const int64_t nx = (int64_t)Wr * 64;
  for (int w0 = 0; w0 < Wr; w0 += 16) {  // 16 words = 128 bricks = two mask words
    const int nw = min(16, Wr - w0);
    u64 mlo = skip64[w0 >> 3], mhi = skip64[(w0 >> 3) + 1];
    mlo = ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(mlo >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)mlo);
    mhi = ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(mhi >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)mhi);
    // per lane: bit k = this lane's voxel of word k lies in a skipped brick; uniform: word k is skipped entirely
    uint32_t lane_skip = 0, word_skip = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const uint32_t byte = (uint32_t)((k < 8 ? mlo : mhi) >> (8 * (k & 7))) & 255u;
      word_skip |= (byte == 255u ? 1u : 0u) << k;
      lane_skip |= ((byte >> (lane >> 3)) & 1u) << k;
    }
    // This wave's 16 rows of the brick row, four at a time, and of each row only the words that hold a brick to be read
    // (`live`, uniform), four of those at a time: 16 loads in flight per pass.  (Round 3 walked a row's 16 words with
    // predicated loads, one row per memory round trip: in a carved grid a quarter of the words are live, so a wave had
    // four loads in flight and sixteen round trips in a row -- the pass moved its 1.4 GB at 3.6 TB/s.)
    const uint32_t live = ~word_skip & ((nw >= 32 ? 0u : (1u << nw)) - 1u);
    for (int i0 = 0; i0 < 16; i0 += 4) {
      int64_t rows[4];
      bool valid[4];
      const float* __restrict__ ps[4];
      u64 m_in[4], m_ok[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = wave * 16 + i0 + q;
        const int yy = by * 8 + (r & 7), zz = bz * 8 + (r >> 3);
        valid[q] = yy < ny && zz < nz;  // (uniform)
        rows[q] = (int64_t)zz * ny + yy;
        ps[q] = sdf + rows[q] * nx + (int64_t)w0 * 64 + lane;
        m_in[q] = 0ull;   // a skipped brick: outside ...
        m_ok[q] = ~0ull;  // ... and valid
      }
      uint32_t rest = live;
      while (rest != 0u) {  // (uniform)
        int kk[4];
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          kk[j] = 0;
          if (rest != 0u) {
            kk[j] = __builtin_ctz(rest);
            rest &= rest - 1u;
            cnt = j + 1;
          }
        }
        float sv[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            sv[q][j] = INFINITY;  // outside and valid
            if (valid[q] && j < cnt) {                                                  // (uniform)
              if (!((lane_skip >> kk[j]) & 1u)) sv[q][j] = __builtin_nontemporal_load(ps[q] + kk[j] * 64);
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (valid[q] && j < cnt) {                                                  // (uniform)
              const u64 a = __ballot(ISO_F32 ? sv[q][j] < (float)iso : (double)sv[q][j] < iso);
              const u64 b = __ballot(sv[q][j] != kInvalidSdf);
              const bool mine = lane == kk[j];
              m_in[q] = mine ? a : m_in[q];
              m_ok[q] = mine ? b : m_ok[q];
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (valid[q] && lane < nw) {
          const int64_t o = rows[q] * Wr + w0 + lane;
          in[o] = m_in[q];
          ok[o] = m_ok[q];
          if (tc != nullptr) tc[o] = m_ok[q];
        }
      }
    }
  }
}

// ---- shared cell-word helpers -----------------------------------------------------------------
__device__ __forceinline__ uint32_t fast_div(uint32_t n, const FastDiv& f) {
  const uint32_t t = __umulhi(n, f.m);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}

__device__ __forceinline__ bool decode_word(const McParams& p, int64_t cw, int* li, int* cy, int* w) {
  if (p.small32) {
    uint32_t r;
    if (cw < p.G) {
      if (cw >= (int64_t)p.Yc * p.Wr) return false;  // padding
      *li = 0;
      r = (uint32_t)cw;
    } else {
      const uint32_t q = (uint32_t)(cw - p.G);
      const uint32_t layer = fast_div(q, p.div_layer);
      *li = (int)layer + 1;
      r = q - layer * p.div_layer.d;
    }
    const uint32_t row = fast_div(r, p.div_row);
    *cy = (int)row;
    *w = (int)(r - row * p.div_row.d);
    return *cy < p.Y;  // (rows Y .. Yc-1 are padding)
  }
  int64_t r;
  if (cw < p.G) {
    if (cw >= (int64_t)p.Yc * p.Wr) return false;  // padding
    *li = 0;
    r = cw;
  } else {
    const int64_t q = cw - p.G;
    const int64_t layer = q / ((int64_t)p.Yc * p.Wr);
    *li = (int)layer + 1;
    r = q - layer * ((int64_t)p.Yc * p.Wr);
  }
  *cy = (int)(r / p.Wr);
  *w = (int)(r - (int64_t)(*cy) * p.Wr);
  return *cy < p.Y;
}

__device__ __forceinline__ int64_t word_index(const McParams& p, int li, int cy, int w) {
  return (li == 0 ? 0 : p.G + (int64_t)(li - 1) * p.Yc * p.Wr) + (int64_t)cy * p.Wr + w;
}

// padded per-cell index (info array)
__device__ __forceinline__ int64_t cell_slot(int64_t cw, int b) { return cw * 64 + b; }

// the 8 corner planes of the 64 cells of word (li, cy, w) and their validity
struct CellWord {
  u64 c[8];
  u64 valid;
};

__device__ __forceinline__ u64 shl1(const u64* row, int w) {
  // bit b = voxel x-1: shift towards higher x, carry from the previous word
  return (row[w] << 1) | (w > 0 ? row[w - 1] >> 63 : 0ull);
}

// rows of the four voxel lines a cell row touches: (y-1,z-1), (y,z-1), (y-1,z), (y,z)
struct CellRows {
  int64_t r00, r10, r01, r11;
};
__device__ __forceinline__ CellRows cell_rows(const McParams& p, int li, int cy) {
  const int z = p.zc0 + li - 1;   // global z of the max corner
  const int y = cy + 1;
  const int64_t rw = (int64_t)p.Wr;
  CellRows r;
  r.r11 = ((int64_t)(z - p.zs0) * p.ny + y) * rw;   // (y,   z)
  r.r01 = r.r11 - rw;                               // (y-1, z)
  r.r10 = r.r11 - (int64_t)p.ny * rw;               // (y,   z-1)
  r.r00 = r.r10 - rw;                               // (y-1, z-1)
  return r;
}

// the 8 corner planes (IN) of the 64 cells of word w
__device__ __forceinline__ void load_cell_in(const McParams& p, const CellRows& r, int w, CellWord* o) {
  o->c[0] = shl1(p.in + r.r00, w);
  o->c[1] = p.in[r.r00 + w];
  o->c[2] = p.in[r.r10 + w];
  o->c[3] = shl1(p.in + r.r10, w);
  o->c[4] = shl1(p.in + r.r01, w);
  o->c[5] = p.in[r.r01 + w];
  o->c[6] = p.in[r.r11 + w];
  o->c[7] = shl1(p.in + r.r11, w);
}

// which of the 64 cells are evaluated at all: corner 6 touched (marching_cubes.cc:88-90), no corner invalid (:103-112)
__device__ __forceinline__ u64 load_cell_valid(const McParams& p, const CellRows& r, int w) {
  u64 v = p.tc[r.r11 + w];
  v &= p.ok[r.r00 + w] & p.ok[r.r10 + w] & p.ok[r.r01 + w] & p.ok[r.r11 + w];
  v &= shl1(p.ok + r.r00, w) & shl1(p.ok + r.r10, w) & shl1(p.ok + r.r01, w) & shl1(p.ok + r.r11, w);
  // x = 0 is not a cell; bits beyond nx are zero in every plane already
  if (w == 0) v &= ~1ull;
  return v;
}

__device__ __forceinline__ void load_cell_word(const McParams& p, int li, int cy, int w, CellWord* o) {
  const CellRows r = cell_rows(p, li, cy);
  load_cell_in(p, r, w, o);
  o->valid = load_cell_valid(p, r, w);
}

__device__ __forceinline__ u64 active_mask(const CellWord& cw) {
  u64 all = cw.c[0], any = cw.c[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) {
    all &= cw.c[i];
    any |= cw.c[i];
  }
  return cw.valid & any & ~all;   // kEdgeTable[cube] != 0  (:131-133)
}

__device__ __forceinline__ int case_of(const CellWord& cw, int b) {
  int code = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) code |= (int)((cw.c[i] >> b) & 1ull) << i;
  return code;
}

__device__ __forceinline__ int cut_edges(int c) {
  // an edge is cut iff its two corners differ (== reference kEdgeTable[c])
  int m = 0;
#pragma unroll
  for (int e = 0; e < 12; ++e) m |= (((c >> kEdgeA[e]) ^ (c >> kEdgeB[e])) & 1) << e;
  return m;
}

__device__ __forceinline__ bool neighbour_active(const McParams& p, const u64* __restrict__ act, int li,
                                                 int cy, int x, int dx, int dy, int dl) {
  const int nl = li + dl, ncy = cy + dy, nxx = x + dx;
  if (nl < 0 || ncy < 0 || ncy >= p.Y || nxx < 1 || nxx >= p.nx) return false;
  return (act[word_index(p, nl, ncy, nxx >> 6)] >> (nxx & 63)) & 1ull;
}

__device__ __forceinline__ int owned_edges(const McParams& p, const u64* __restrict__ act, int code, int li,
                                           int cy, int x, int* nactive_out = nullptr) {
  const int cut = cut_edges(code);
  // the ACT bits of the nine neighbour cells an edge can be shared with, requested together (cells outside
  // the grid read word 0 and count as inactive), then pure bit logic
  u64 aw[kNbrCount];
  int bit[kNbrCount];
#pragma unroll
  for (int q = 0; q < kNbrCount; ++q) {
    const int nl = li + kNbr[q][2], ncy = cy + kNbr[q][1], nxx = x + kNbr[q][0];
    const bool there = nl >= 0 && ncy >= 0 && ncy < p.Y && nxx >= 1 && nxx < p.nx;
    bit[q] = there ? (nxx & 63) : -1;
    aw[q] = act[there ? word_index(p, nl, ncy, nxx >> 6) : 0];
  }
  int nactive = 0;
#pragma unroll
  for (int q = 0; q < kNbrCount; ++q) nactive |= (bit[q] >= 0 && ((aw[q] >> bit[q]) & 1ull)) ? (1 << q) : 0;
  int owned = 0;
#pragma unroll
  for (int e = 0; e < 12; ++e) {
    int sharers = 0;
#pragma unroll
    for (int k = 0; k < kShare[e].n; ++k) sharers |= 1 << kShareNbr[e][k];
    if ((cut & (1 << e)) && !(nactive & sharers)) owned |= 1 << e;
  }
  // a ghost cell only contributes vertices on the plane it shares with the slab (e4..e7)
  if (li == 0) owned &= 0xF0;
  if (nactive_out) *nactive_out = nactive;
  return owned;
}

// ---- block-level exclusive scan (256 threads = 4 waves) ---------------------------------------
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

__device__ __forceinline__ int block_exclusive_scan(int v, int* total, int* sm /*[4]*/) {
  const int incl = wave_inclusive_scan(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 63) sm[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int s = sm[w];
    if (w < wave) base += s;
    tot += s;
  }
  *total = tot;
  return base + incl - v;
}

// ---- pass 1: active cells, counted per word --------------------------------------------------
// A workgroup handles kActiveBlocks consecutive blocks of 256 words, and every thread requests the corner
// planes of all its words before it uses the first: the pass is one memory round trip deep and is bound by
// how many requests a CU issues, not by the bytes (1024^3, after the carry loads went: 0.164 ms with one block per
// workgroup, 0.166 with two, 0.186 with three, 0.216 with four).
#ifndef VCY_ACTIVE_BLOCKS
#define VCY_ACTIVE_BLOCKS 2
#endif
constexpr int kActiveBlocks = VCY_ACTIVE_BLOCKS;

__global__ __launch_bounds__(256) void mc_active_kernel(McParams p, u64* __restrict__ act,
                                                        uint32_t* __restrict__ word_cell_off,
                                                        u64* __restrict__ block_cells, int64_t nblocks) {
  __shared__ int sm[kActiveBlocks][4];
  // Plain order.  (Rounds 2-4 gave every XCD -- workgroup b runs on XCD b % 8 -- a contiguous eighth of the blocks, so
  // that the plane rows consecutive blocks and, one layer later, consecutive layers share stay in one L2.  Measured
  // against plain order in round 5: 136 against 131.5 us at 1024^3, 16.7 against 15.5 at 512^3, and mc_compact behind it
  // 45.8 against 41.3 -- the planes are 1 bit per voxel and every XCD's L2 holds the rows of its neighbours anyway;
  // profiles/r05/mc_lookback.txt.  VCY_ACTIVE_XCD_EIGHTHS restores the old order.)
  int64_t lg = blockIdx.x;
#ifdef VCY_ACTIVE_XCD_EIGHTHS
  {
    const int64_t per = gridDim.x >> 3;
    if (lg < per * 8) lg = (lg & 7) * per + (lg >> 3);
  }
#endif
  CellWord c[kActiveBlocks];
  CellRows r[kActiveBlocks];
  int w[kActiveBlocks];
  bool live[kActiveBlocks];
#pragma unroll
  for (int k = 0; k < kActiveBlocks; ++k) {
    const int64_t cw = (lg * kActiveBlocks + k) * 256 + threadIdx.x;
    int li = 0, cy = 0;
    w[k] = 0;
    live[k] = cw < p.nwords && decode_word(p, cw, &li, &cy, &w[k]) && (li > 0 || p.has_ghost);
    if (!live[k]) li = 1, cy = 0, w[k] = 0;  // (any valid word: the loads below are not branched around)
    r[k] = cell_rows(p, li, cy);
    // The pass is bound by the number of load instructions (reading the validity planes for every word as well
    // makes it 50 % slower): a thread loads its own word of the four voxel rows; the word before -- for x - 1 of
    // bit 0 -- is the one the lane before loaded (consecutive lanes hold consecutive words of a row), only lane 0
    // of a wave fetches it itself.
    const u64 m00 = p.in[r[k].r00 + w[k]], m10 = p.in[r[k].r10 + w[k]];
    const u64 m01 = p.in[r[k].r01 + w[k]], m11 = p.in[r[k].r11 + w[k]];
    uint32_t h00 = 0, h10 = 0, h01 = 0, h11 = 0;  // high halves of the words before
    const bool edge = (threadIdx.x & 63) == 0 && w[k] > 0;
    if (__ballot(edge) != 0 && edge) {  // (never when a row is 1 .. 64 words: lane 0 then starts a row)
      h00 = (uint32_t)(p.in[r[k].r00 + w[k] - 1] >> 32);
      h10 = (uint32_t)(p.in[r[k].r10 + w[k] - 1] >> 32);
      h01 = (uint32_t)(p.in[r[k].r01 + w[k] - 1] >> 32);
      h11 = (uint32_t)(p.in[r[k].r11 + w[k] - 1] >> 32);
    }
    const uint32_t s00 = __shfl_up((uint32_t)(m00 >> 32), 1, 64), s10 = __shfl_up((uint32_t)(m10 >> 32), 1, 64);
    const uint32_t s01 = __shfl_up((uint32_t)(m01 >> 32), 1, 64), s11 = __shfl_up((uint32_t)(m11 >> 32), 1, 64);
    if (!edge) h00 = s00, h10 = s10, h01 = s01, h11 = s11;
    if (w[k] == 0) h00 = h10 = h01 = h11 = 0;
    c[k].c[0] = (m00 << 1) | (h00 >> 31);
    c[k].c[1] = m00;
    c[k].c[2] = m10;
    c[k].c[3] = (m10 << 1) | (h10 >> 31);
    c[k].c[4] = (m01 << 1) | (h01 >> 31);
    c[k].c[5] = m01;
    c[k].c[6] = m11;
    c[k].c[7] = (m11 << 1) | (h11 >> 31);
  }
  // the offsets of all the workgroup's blocks with ONE barrier: wave scans, the wave totals through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int excl[kActiveBlocks];
#pragma unroll
  for (int k = 0; k < kActiveBlocks; ++k) {
    const int64_t cw = (lg * kActiveBlocks + k) * 256 + threadIdx.x;
    // a cell is active when its corners are neither all inside nor all outside (kEdgeTable != 0) and it
    // is valid; the validity planes are only read for the few words that have a candidate
    c[k].valid = ~0ull;
    u64 a = live[k] ? active_mask(c[k]) : 0ull;
    if (__ballot(a != 0) != 0) {  // (uniform: the whole wave reads, so that the words before can come from lane - 1)
      // corner 6 touched (marching_cubes.cc:88-90), no corner invalid (:103-112) -- load_cell_valid with the carry
      // words taken from the neighbouring lane, as for IN above
      const u64 o00 = p.ok[r[k].r00 + w[k]], o10 = p.ok[r[k].r10 + w[k]];
      const u64 o01 = p.ok[r[k].r01 + w[k]], o11 = p.ok[r[k].r11 + w[k]];
      const u64 t11 = p.tc[r[k].r11 + w[k]];
      uint32_t g00 = 0, g10 = 0, g01 = 0, g11 = 0;
      const bool edge = lane == 0 && w[k] > 0;
      if (edge) {
        g00 = (uint32_t)(p.ok[r[k].r00 + w[k] - 1] >> 32);
        g10 = (uint32_t)(p.ok[r[k].r10 + w[k] - 1] >> 32);
        g01 = (uint32_t)(p.ok[r[k].r01 + w[k] - 1] >> 32);
        g11 = (uint32_t)(p.ok[r[k].r11 + w[k] - 1] >> 32);
      }
      const uint32_t u00 = __shfl_up((uint32_t)(o00 >> 32), 1, 64), u10 = __shfl_up((uint32_t)(o10 >> 32), 1, 64);
      const uint32_t u01 = __shfl_up((uint32_t)(o01 >> 32), 1, 64), u11 = __shfl_up((uint32_t)(o11 >> 32), 1, 64);
      if (!edge) g00 = u00, g10 = u10, g01 = u01, g11 = u11;
      // (x = 0 is not a cell: bit 0 of word 0 is cleared, whatever its carry)
      u64 v = t11 & o00 & o10 & o01 & o11;
      v &= ((o00 << 1) | (g00 >> 31)) & ((o10 << 1) | (g10 >> 31)) & ((o01 << 1) | (g01 >> 31)) & ((o11 << 1) | (g11 >> 31));
      if (w[k] == 0) v &= ~1ull;
      a &= v;
    }
    if (cw < p.nwords) act[cw] = a;
    const int pc = __popcll(a);
    const int inc = wave_inclusive_scan(pc);
    excl[k] = inc - pc;
    if (lane == 63) sm[k][wave] = inc;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kActiveBlocks; ++k) {
    const int64_t lb = lg * kActiveBlocks + k;
    const int64_t cw = lb * 256 + threadIdx.x;
    int base = 0, total = 0;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int t = sm[k][v];
      if (v < wave) base += t;
      total += t;
    }
    if (cw < p.nwords) word_cell_off[cw] = (uint32_t)(base + excl[k]);
    if (threadIdx.x == 0 && lb < nblocks) block_cells[lb] = (u64)(unsigned)total;
  }
}

// ---- passes 0 + 1 in one sweep: the bit planes never reach memory ------------------------------------
// Taken when a voxel row is a power-of-two number of whole words (nx = 64, 128 ... 2048).  A workgroup owns R
// cell rows (R * Wr = K whole 256-word blocks) and walks `layers` cell layers in z.  Per step its four waves turn
// the R + 1 voxel rows of the next slice into IN / OK (/ TC) words in LDS -- sixteen requests in flight per lane,
// then the ballots, as in mc_bits -- and every thread evaluates its K cell words from the words of this slice and
// of the one before (LDS holds two slices).  What goes to memory is what the later passes read: ACT, the offsets
// per word and per block, and the IN words around ACTIVE cells (case_at reads nothing else of that plane: the
// sparse 32-byte sectors of the surface instead of two dense planes written and read back).  The state is read
// once plus the row shared by two row groups (1 / R) and the slice shared by two z chunks (1 / layers).
// update_num: READS_CNT reads it next to sdf (state set by vcy_upload); otherwise OK implies TC, except for the
// ghost layer of a slab, whose max corners lie in the slab below: those TC words come in `tc_ghost`.
struct SweepParams {
  int R;           // cell rows per workgroup
  int K;           // 256-word blocks per step = R * Wr / 256
  int groups;      // row groups per layer = Yc / R
  int layers;      // cell layers per workgroup
  int dl;          // cells whose max corner lies in stored slice s form layer li = s + dl
  int cnt_slices;  // READS_CNT: update_num is read for the stored slices below this one (all of them)
  int wshift;      // log2 Wr
};
constexpr int kSweepMaxK = 4;
constexpr int kSweepBatch = 16;
#ifndef VCY_SWEEP_SETS
#define VCY_SWEEP_SETS 2
#endif
constexpr int kSweepSets = VCY_SWEEP_SETS;  // register sets of kSweepBatch requests in flight per wave
#ifndef VCY_SWEEP_TARGET_WGS
#define VCY_SWEEP_TARGET_WGS 1024
#endif

// lane K of the result = the scalar `sval`, the other lanes keep `old` (the lane select must be an immediate: a
// second scalar register would be a second constant-bus operand)
template <int K>
__device__ __forceinline__ uint32_t write_lane(uint32_t sval, uint32_t old) {
  asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(sval), "n"(K));
  return old;
}
template <typename F, int... Ks>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, Ks...>) {
  (f(std::integral_constant<int, Ks>{}), ...);
}

template <typename CountT, bool ISO_F32, bool READS_CNT, int KMAX>
__global__ __launch_bounds__(256) void mc_sweep_kernel(McParams p, SweepParams q, u64* __restrict__ act,
                                                       uint32_t* __restrict__ word_cell_off,
                                                       u64* __restrict__ block_cells, u64* __restrict__ in_plane,
                                                       const u64* __restrict__ tc_ghost) {
  extern __shared__ u64 planes[];  // [2 slices][IN, OK (, TC)][(R + 1) * Wr]
  __shared__ int sm[2][KMAX][4];
  constexpr int kPlanes = READS_CNT ? 3 : 2;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Wr = p.Wr;
  const int plane_words = (q.R + 1) * Wr;
  // XCD-aware order (workgroup b runs on XCD b % 8): neighbouring row groups of one z chunk -- they share a
  // voxel row per slice -- run on the same XCD at about the same time
  int64_t lg = blockIdx.x;
  {
    const int64_t per = gridDim.x >> 3;
    if (lg < per * 8) lg = (lg & 7) * per + (lg >> 3);
  }
  const int chunk = (int)(lg / q.groups);
  const int cy0 = (int)(lg - (int64_t)chunk * q.groups) * q.R;
  const int li_a = chunk * q.layers;
  const int li_b = min(li_a + q.layers, p.L + 1);

  // Voxel rows cy0 .. cy0 + R of a stored slice -> plane words of LDS buffer (slice & 1).  Each wave takes a
  // quarter of the slice's words in batches of kSweepBatch, and the slices a workgroup needs are consecutive, so
  // the batches form ONE sequence per wave across the steps: two batches are always requested ahead (register
  // sets A and B), also over the barriers and the cell evaluation of a step.
  // (rows cy0 .. cy0 + R of a slice are contiguous in memory: word u of the group starts 64 u voxels after the
  // group's first voxel; words of rows beyond the grid are not requested and read as "outside, invalid")
  const int per_wave = (plane_words + 3) >> 2;
  const int nb = ((per_wave + kSweepBatch - 1) / kSweepBatch + kSweepSets - 1) / kSweepSets * kSweepSets;  // batches per wave and slice
  const int ubase = wave * per_wave;
  const int uend = min(min(ubase + per_wave, plane_words), (p.ny - cy0) * Wr);
  const bool first_has_cells = li_a >= 1 || p.has_ghost;
  const int s_first = li_a - q.dl - ((first_has_cells && li_a - q.dl >= 1) ? 1 : 0);
  const int s_last = li_b - 1 - q.dl;  // (<= nslices - 1 since li <= L)
  int rq_s = s_first, rq_j = 0;        // the next batch to request
  auto request = [&](float (&sv)[kSweepBatch], int (&nv)[kSweepBatch]) {
    // No branch around a request: words that do not exist (beyond the wave's share, the grid or the last slice)
    // read the group's first word instead and are masked in `ballots` -- with a single path the compiler counts
    // the outstanding requests exactly and waits for the older register set only.
    const int sl = min(rq_s, s_last);
    const int u0 = ubase + rq_j * kSweepBatch;
    const bool cnt_here = READS_CNT && sl < q.cnt_slices;
    const float* __restrict__ ps = p.sdf + ((int64_t)sl * p.ny + cy0) * p.nx + lane;
    const CountT* __restrict__ pc = (const CountT*)p.cnt + ((int64_t)sl * p.ny + cy0) * p.nx + lane;
#pragma unroll
    for (int k = 0; k < kSweepBatch; ++k) {
      const int u = u0 + k < uend ? u0 + k : 0;  // (scalar)
      // (streaming loads also for the rows two groups share: ordinary loads there cost the extraction 6 %, everywhere 8 %)
      sv[k] = __builtin_nontemporal_load(ps + (int64_t)u * 64);
      nv[k] = 1;
      if (cnt_here) nv[k] = (int)pc[(int64_t)u * 64];
    }
    if (++rq_j == nb) rq_j = 0, ++rq_s;
  };
  auto ballots = [&](int s, int j, const float (&sv)[kSweepBatch], const int (&nv)[kSweepBatch]) {
    u64* __restrict__ dst = planes + (size_t)(s & 1) * kPlanes * plane_words;
    const int u0 = ubase + j * kSweepBatch;
    if (u0 >= min(ubase + per_wave, plane_words)) return;
    // word k of the batch goes to lane k (v_writelane: the ballot is a scalar)
    uint32_t in_lo = 0, in_hi = 0, ok_lo = 0, ok_hi = 0, tc_lo = 0, tc_hi = 0;
    static_for([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      const bool there = u0 + k < uend;
      // (comparisons as in mc_bits_kernel)
      const u64 a = __ballot(there && (ISO_F32 ? sv[k] < (float)p.iso : (double)sv[k] < p.iso));
      const u64 b = __ballot(there && sv[k] != kInvalidSdf);
      in_lo = write_lane<k>((uint32_t)a, in_lo);
      in_hi = write_lane<k>((uint32_t)(a >> 32), in_hi);
      ok_lo = write_lane<k>((uint32_t)b, ok_lo);
      ok_hi = write_lane<k>((uint32_t)(b >> 32), ok_hi);
      if (READS_CNT) {
        const u64 c = __ballot(nv[k] >= 1);
        tc_lo = write_lane<k>((uint32_t)c, tc_lo);
        tc_hi = write_lane<k>((uint32_t)(c >> 32), tc_hi);
      }
    }, std::make_integer_sequence<int, kSweepBatch>{});
    const u64 m_in = ((u64)in_hi << 32) | in_lo, m_ok = ((u64)ok_hi << 32) | ok_lo, m_tc = ((u64)tc_hi << 32) | tc_lo;
    if (lane < kSweepBatch && u0 + lane < min(ubase + per_wave, plane_words)) {
      dst[u0 + lane] = m_in;
      dst[plane_words + u0 + lane] = m_ok;
      if (READS_CNT) dst[2 * plane_words + u0 + lane] = m_tc;
    }
  };
  float sa[kSweepBatch], sb[kSweepBatch], sc[kSweepBatch];
  int na[kSweepBatch], nbb[kSweepBatch], nc[kSweepBatch];
  request(sa, na);
  request(sb, nbb);
  if (kSweepSets == 3) request(sc, nc);
  auto fill = [&](int s) {
    for (int j = 0; j < nb; j += kSweepSets) {
      ballots(s, j, sa, na);
      request(sa, na);
      ballots(s, j + 1, sb, nbb);
      request(sb, nbb);
      if (kSweepSets == 3) {
        ballots(s, j + 2, sc, nc);
        request(sc, nc);
      }
    }
  };

  if (s_first < li_a - q.dl) fill(s_first);
  for (int li = li_a; li < li_b; ++li) {
    const int s = li - q.dl;  // stored slice of the max corners
    const bool cells = (li >= 1 || p.has_ghost) && s >= 1;
    fill(s);
    __syncthreads();
    const u64* __restrict__ cur = planes + (size_t)(s & 1) * kPlanes * plane_words;
    const u64* __restrict__ prv = planes + (size_t)((s & 1) ^ 1) * kPlanes * plane_words;
    const int par = li & 1;
    const int64_t cw0 = word_index(p, li, cy0, 0);
    int excl[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      excl[k] = 0;
      if (k < q.K) {
        const int i = k * 256 + threadIdx.x;
        const int row = i >> q.wshift, w = i & (Wr - 1);
        u64 m = 0;
        if (cells && cy0 + row < p.Y) {
          // voxel rows y-1 (r0) and y (r1) of slices z-1 (prv) and z (cur); corner order of load_cell_in
          const int r0 = row * Wr + w, r1 = r0 + Wr;
          const u64 i00 = prv[r0], i10 = prv[r1], i01 = cur[r0], i11 = cur[r1];
          u64 b00 = 0, b10 = 0, b01 = 0, b11 = 0;  // the words before (for x - 1 of bit 0)
          if (w > 0) b00 = prv[r0 - 1], b10 = prv[r1 - 1], b01 = cur[r0 - 1], b11 = cur[r1 - 1];
          const u64 s00 = (i00 << 1) | (b00 >> 63), s10 = (i10 << 1) | (b10 >> 63);
          const u64 s01 = (i01 << 1) | (b01 >> 63), s11 = (i11 << 1) | (b11 >> 63);
          const u64 all = i00 & i10 & i01 & i11 & s00 & s10 & s01 & s11;
          const u64 any = i00 | i10 | i01 | i11 | s00 | s10 | s01 | s11;
          m = any & ~all;  // kEdgeTable[cube] != 0 (:131-133)
          if (w == 0) m &= ~1ull;  // x = 0 is not a cell
          if (m) {
            // corner 6 touched (:88-90), no corner invalid (:103-112)
            const u64* __restrict__ okp = prv + plane_words;
            const u64* __restrict__ okc = cur + plane_words;
            const u64 o00 = okp[r0], o10 = okp[r1], o01 = okc[r0], o11 = okc[r1];
            u64 v = o00 & o10 & o01 & o11;
            u64 p00 = ~0ull, p10 = ~0ull, p01 = ~0ull, p11 = ~0ull;
            if (w > 0) p00 = okp[r0 - 1], p10 = okp[r1 - 1], p01 = okc[r0 - 1], p11 = okc[r1 - 1];
            v &= ((o00 << 1) | (p00 >> 63)) & ((o10 << 1) | (p10 >> 63)) & ((o01 << 1) | (p01 >> 63)) &
                 ((o11 << 1) | (p11 >> 63));
            if (READS_CNT) v &= cur[2 * plane_words + r1];
            // the ghost layer's max corners lie in a slice of another context: its TC words, made by mc_bits
            if (!READS_CNT && tc_ghost != nullptr && li == 0) v &= tc_ghost[(cy0 + row + 1) * Wr + w];
            m &= v;
          }
          if (m) {
            const int64_t g11 = ((int64_t)s * p.ny + (cy0 + row + 1)) * Wr + w;
            const int64_t g01 = g11 - Wr, g10 = g11 - (int64_t)p.ny * Wr, g00 = g10 - Wr;
            in_plane[g00] = i00;
            in_plane[g10] = i10;
            in_plane[g01] = i01;
            in_plane[g11] = i11;
            if ((m & 1ull) && w > 0) {
              in_plane[g00 - 1] = b00;
              in_plane[g10 - 1] = b10;
              in_plane[g01 - 1] = b01;
              in_plane[g11 - 1] = b11;
            }
          }
        }
        act[cw0 + i] = m;
        const int pc = __popcll(m);
        const int inc = wave_inclusive_scan(pc);
        excl[k] = inc - pc;
        if (lane == 63) sm[par][k][wave] = inc;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < q.K) {
        int base = 0, tot = 0;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) {
          const int t = sm[par][k][w4];
          if (w4 < wave) base += t;
          tot += t;
        }
        const int64_t cw = cw0 + k * 256 + threadIdx.x;
        word_cell_off[cw] = (uint32_t)(base + excl[k]);
        if (threadIdx.x == 0) block_cells[cw >> 8] = (u64)(unsigned)tot;
      }
    }
  }
}

// ---- pass 2: compact the active cells into a list (raster order is preserved) ------------------
// cell_list[i] = padded cell slot (word * 64 + bit).  The inverse map needs no array: the list index of
// the active cell (word cw, bit b) is block_cell_offs[cw >> 8] + word_cell_off[cw] + popcount of the
// active bits below b (list_index_of), three loads from arrays a few hundred times smaller than the grid.
constexpr int kCompactBlocks = 16;
__global__ __launch_bounds__(256) void mc_compact_kernel(McParams p, const u64* __restrict__ act,
                                                         const uint32_t* __restrict__ word_cell_off,
                                                         const u64* __restrict__ block_cell_offs,
                                                         const u64* __restrict__ total_cells, int64_t nblocks,
                                                         u64* __restrict__ cell_list, int64_t capacity) {
  // A workgroup walks kCompactBlocks blocks of 256 words; most hold no active cell at all, which two offsets
  // tell without touching the words (65 536 workgroups of one block each spent their time being dispatched).
  for (int k = 0; k < kCompactBlocks; ++k) {
    const int64_t blk = (int64_t)blockIdx.x * kCompactBlocks + k;
    if (blk >= nblocks) return;
    const u64 first = block_cell_offs[blk];
    const u64 next = (blk + 1 < nblocks) ? block_cell_offs[blk + 1] : *total_cells;
    if (next == first) continue;  // (uniform)
    const int64_t cw = blk * 256 + threadIdx.x;
    if (cw >= p.nwords) continue;
    u64 a = act[cw];
    int64_t i = (int64_t)first + word_cell_off[cw];
    while (a) {
      const int b = __ffsll((long long)a) - 1;
      a &= a - 1;
      if (i < capacity) cell_list[i] = (u64)cell_slot(cw, b);  // (a list sized from the last extraction may be short)
      ++i;
    }
  }
}

__device__ __forceinline__ int64_t list_index_of(const u64* __restrict__ act, const uint32_t* __restrict__ word_cell_off,
                                                 const u64* __restrict__ block_cell_offs, int64_t cw, int b) {
  return (int64_t)block_cell_offs[cw >> 8] + word_cell_off[cw] + __popcll(act[cw] & ((1ull << b) - 1ull));
}

// cube index of one cell from the IN plane (marching_cubes.cc:121-128)
__device__ __forceinline__ int case_at(const McParams& p, int li, int cy, int x) {
  const int z = p.zc0 + li - 1, y = cy + 1;
  const int64_t rw = (int64_t)p.Wr;
  const int64_t r11 = ((int64_t)(z - p.zs0) * p.ny + y) * rw;
  const int64_t r01 = r11 - rw, r10 = r11 - (int64_t)p.ny * rw, r00 = r10 - rw;
  const int w = x >> 6, b = x & 63, wp = (x - 1) >> 6, bp = (x - 1) & 63;  // x >= 1 for a cell
  auto bit = [&](int64_t row, int ww, int bb) -> int { return (int)((p.in[row + ww] >> bb) & 1ull); };
  return bit(r00, wp, bp) | bit(r00, w, b) << 1 | bit(r10, w, b) << 2 | bit(r10, wp, bp) << 3 |
         bit(r01, wp, bp) << 4 | bit(r01, w, b) << 5 | bit(r11, w, b) << 6 | bit(r11, wp, bp) << 7;
}

// ---- pass 3: edge ownership + counts, one thread per active cell ----------------------------------
// info[i] = owned edges (12 bits) | case << 12 | first vertex of the cell inside its block << 20
__global__ __launch_bounds__(256) void mc_owner_kernel(McParams p, const McTables* __restrict__ T,
                                                       const u64* __restrict__ act,
                                                       const u64* __restrict__ cell_list,
                                                       const u64* __restrict__ ncells_dev, int64_t capacity,
                                                       uint32_t* __restrict__ info,
                                                       uint16_t* __restrict__ nbr_active,
                                                       u64* __restrict__ block_counts) {
  __shared__ int sm[4];
  // the number of active cells is read where the scan left it: the host need not know it to launch this
  const int64_t ncells = min((int64_t)*ncells_dev, capacity);
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int nvert = 0, ntri = 0, owned = 0, code = 0, nactive = 0;
  if (i < ncells) {
    const int64_t slot = (int64_t)cell_list[i];
    int li, cy, w;
    decode_word(p, slot >> 6, &li, &cy, &w);
    const int x = w * 64 + (int)(slot & 63);
    code = case_at(p, li, cy, x);
    owned = owned_edges(p, act, code, li, cy, x, &nactive);
    nvert = __popc(owned);
    if (li > 0) ntri = T->ntri[code];
  }
  int tot_v, tot_t;
  const int off_v = block_exclusive_scan(nvert, &tot_v, sm);
  (void)block_exclusive_scan(ntri, &tot_t, sm);
  if (i < ncells) {
    info[i] = (uint32_t)owned | ((uint32_t)code << 12) | ((uint32_t)off_v << 20);
    // which of the nine neighbour cells are active: mc_emit finds the owners of this cell's other cut edges from it
    // without asking again (it used to re-read the nine ACT words -- and three more arrays for all nine)
    nbr_active[i] = (uint16_t)nactive;
  }
  if (threadIdx.x == 0) block_counts[blockIdx.x] = ((u64)(unsigned)tot_v << 32) | (u64)(unsigned)tot_t;
}

// ---- exclusive scan of packed block counts (used twice: cells per word block, (verts<<32 | tris)) ---------------------------
__global__ __launch_bounds__(256) void scan_chunks_kernel(u64* __restrict__ data, int64_t n,
                                                          u64* __restrict__ chunk_sums) {
  // 1024 elements per block, 4 per thread
  __shared__ u64 sm[256];
  const int64_t base = (int64_t)blockIdx.x * 1024 + (int64_t)threadIdx.x * 4;
  u64 v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = (base + k < n) ? data[base + k] : 0ull;
    s += v[k];
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    u64 t = (threadIdx.x >= d) ? sm[threadIdx.x - d] : 0ull;
    __syncthreads();
    sm[threadIdx.x] += t;
    __syncthreads();
  }
  u64 run = sm[threadIdx.x] - s;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < n) data[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 255) chunk_sums[blockIdx.x] = sm[255];
}

__global__ __launch_bounds__(256) void add_chunk_offsets_kernel(u64* __restrict__ data, int64_t n,
                                                                const u64* __restrict__ offs) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) data[i] += offs[i >> 10];
}

// The same in ONE launch (three launches otherwise: chunks, their sums, the offsets added -- each a dependent kernel
// boundary in the middle of an extraction made of short kernels): a workgroup scans 1024 elements, publishes its sum,
// and adds up the sums of the chunks before it, waiting for those that are not published yet.
// WHICH chunk a workgroup scans is a ticket drawn from a device-scope counter when the workgroup starts, not its
// blockIdx: chunk c is only ever waited for by chunks drawn AFTER it, i.e. by workgroups that started after the one
// holding c did -- and that one is running, it needs nothing but its own data to publish.  Forward progress therefore
// rests on no assumption about the order in which the dispatchers start workgroups, whatever else occupies the device
// (other contexts' streams, a persistent probe wave, a long carve): the lowest unpublished ticket always belongs to a
// resident workgroup.  One atomic per workgroup on one address retires at about one per 10 ns here; the grids of this
// kernel are at most kChainedScanMaxChunks = 1024 blocks (1024^3: 64 chunks for the word blocks, about 10 for the
// surface cells), i.e. 0.6 us and 0.1 us.  (That price is what ruled tickets out for a single-pass prefix over the
// 32 768 workgroups of mc_active / mc_owner, profiles/r05/mc_lookback.txt -- not for this kernel.)
// The counter is never cleared: the host passes the number of tickets drawn before this launch (`ticket_base`; every
// launched workgroup draws exactly one).  Publication is (sum, epoch): the epoch grows with every scan of a context, so
// the flags are never cleared either.  Every chunk reads all its predecessors -- quadratic, which is why this form is
// only taken up to kChainedScanMaxChunks.
constexpr int kChainedScanMaxChunks = 1024;
__global__ __launch_bounds__(256) void scan_chained_kernel(u64* __restrict__ data, int64_t n, u64* __restrict__ chunk_sums,
                                                           uint32_t* __restrict__ chunk_flags, uint32_t epoch,
                                                           u64* __restrict__ total, uint32_t* __restrict__ ticket,
                                                           uint32_t ticket_base) {
  __shared__ u64 sm[256];
  __shared__ u64 sm_before;
  __shared__ int sm_chunk;
  if (threadIdx.x == 0)
    sm_chunk = (int)(__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ticket_base);
  __syncthreads();
  const int chunk = sm_chunk;  // 0 .. gridDim.x - 1, each exactly once
  const int64_t base = (int64_t)chunk * 1024 + (int64_t)threadIdx.x * 4;
  u64 v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = (base + k < n) ? data[base + k] : 0ull;
    s += v[k];
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    u64 t = (threadIdx.x >= d) ? sm[threadIdx.x - d] : 0ull;
    __syncthreads();
    sm[threadIdx.x] += t;
    __syncthreads();
  }
  if (threadIdx.x == 255) {  // publish this chunk's sum
    __hip_atomic_store(&chunk_sums[chunk], sm[255], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&chunk_flags[chunk], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
  // the sums of the chunks before this one: thread t takes chunks t, t + 256, ...
  u64 before = 0;
  for (int c = threadIdx.x; c < chunk; c += 256) {
    while (__hip_atomic_load(&chunk_flags[c], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(1);
    before += __hip_atomic_load(&chunk_sums[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const u64 incl_local = sm[threadIdx.x];
  __syncthreads();
  sm[threadIdx.x] = before;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) sm[threadIdx.x] += sm[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) sm_before = sm[0];
  __syncthreads();
  u64 run = sm_before + incl_local - s;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < n) data[base + k] = run;
    run += v[k];
  }
  if (chunk == (int)gridDim.x - 1 && threadIdx.x == 255) *total = run;  // (the last element's inclusive value)
}

// What a chained scan needs from its caller: the publication flags of ONE scan slot (kChainedScanMaxChunks words), the
// slot's ticket counter and how many tickets have been drawn from it so far (advanced here).
struct ChainedScanSlot {
  uint32_t* flags;
  uint32_t* ticket;
  uint32_t* tickets_drawn;  // (host)
  uint32_t epoch;
};

// in-place exclusive scan; *d_total (device) receives the grand total.  `scratch` holds the chunk
// sums of every level (n/1024 + n/1024^2 + ... + a few elements).
int exclusive_scan_u64(u64* d, int64_t n, u64* d_total, u64* scratch, hipStream_t stream,
                       const ChainedScanSlot* slot = nullptr) {
  const int64_t nchunks = (n + 1023) / 1024;
  if (slot != nullptr && nchunks <= kChainedScanMaxChunks) {
    hipLaunchKernelGGL(scan_chained_kernel, dim3((unsigned)nchunks), dim3(256), 0, stream, d, n, scratch, slot->flags,
                       slot->epoch, d_total, slot->ticket, *slot->tickets_drawn);
    VCY_HIP_CHECK(hipGetLastError());
    *slot->tickets_drawn += (uint32_t)nchunks;  // (modulo 2^32, like the counter)
    return VCY_OK;
  }
  hipLaunchKernelGGL(scan_chunks_kernel, dim3((unsigned)nchunks), dim3(256), 0, stream, d, n, scratch);
  if (nchunks > 1) {
    int rc = exclusive_scan_u64(scratch, nchunks, d_total, scratch + nchunks, stream);
    if (rc != VCY_OK) return rc;
    hipLaunchKernelGGL(add_chunk_offsets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d, n,
                       scratch);
  } else {
    VCY_HIP_CHECK(hipMemcpyAsync(d_total, scratch, sizeof(u64), hipMemcpyDeviceToDevice, stream));
  }
  VCY_HIP_CHECK(hipGetLastError());
  return VCY_OK;
}

// ---- pass 4: emit, one thread per active cell -----------------------------------------------------
__device__ __forceinline__ int64_t vertex_id_of(const McTables* T, const uint32_t* __restrict__ info,
                                                const u64* __restrict__ block_offs, int64_t i, int edge) {
  const uint32_t inf = info[i];
  const int owned = inf & 0xFFF, code = (inf >> 12) & 0xFF, in_block = inf >> 20;
  return (int64_t)(block_offs[i >> 8] >> 32) + in_block + __popc(owned & T->prec[code][edge]);
}

// VertexInterp, marching_cubes.cc:25-57 (fp64, then cast)
__device__ __forceinline__ void vertex_interp(double iso, const float pa[3], const float pb[3], float va,
                                              float vb, bool linear, float out[3]) {
  if (!linear) {
    out[0] = pa[0]; out[1] = pa[1]; out[2] = pa[2];
    return;
  }
  const double v1 = va, v2 = vb;
  if (fabs(iso - v1) < 0.00001) { out[0] = pa[0]; out[1] = pa[1]; out[2] = pa[2]; return; }
  if (fabs(iso - v2) < 0.00001) { out[0] = pb[0]; out[1] = pb[1]; out[2] = pb[2]; return; }
  if (fabs(v1 - v2) < 0.00001) { out[0] = pa[0]; out[1] = pa[1]; out[2] = pa[2]; return; }
  const double mu = (iso - v1) / (v2 - v1);
#pragma unroll
  for (int k = 0; k < 3; ++k) out[k] = (float)((double)pa[k] + mu * ((double)pb[k] - (double)pa[k]));
}

// Output staging of one block of 256 active cells.  A block's vertices and triangles are contiguous
// ranges of the output arrays (cells are numbered in list order), so they are assembled in LDS and
// written out as whole rows of dwords instead of scattered 12- and 16-byte pieces (the scattered form
// wrote 4x the bytes it produced).  Blocks whose totals exceed the staging (dense noise) store directly.
constexpr int kEmitMaxVerts = 512;  // 6 KB + 8 KB of keys (a smooth surface has about one vertex per active cell)
constexpr int kEmitMaxTris = 768;   // 9 KB (about two triangles per active cell)

// (round 4: 98 VGPRs once the gather asked for the owners only, 123 before; compiled for 5 waves per SIMD -- a handful of
// spills -- 160 -> 146 us at 1024^3; for 6: 205 us.  Halving its loads changed nothing by itself: the kernel waits for
// its chain of dependent gathers, so occupancy is what helps.)
#ifndef VCY_EMIT_WAVES
#define VCY_EMIT_WAVES 5
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VCY_EMIT_WAVES, VCY_EMIT_WAVES))) void mc_emit_kernel(McParams p, const McTables* __restrict__ T,
                                                      const u64* __restrict__ act,
                                                      const u64* __restrict__ cell_list,
                                                      const u64* __restrict__ ncells_dev, int64_t capacity,
                                                      const uint32_t* __restrict__ word_cell_off,
                                                      const u64* __restrict__ block_cell_offs,
                                                      const uint32_t* __restrict__ info,
                                                      const uint16_t* __restrict__ nbr_active,
                                                      const u64* __restrict__ block_offs,
                                                      const u64* __restrict__ grand_total_dev, int64_t verts_capacity,
                                                      int64_t faces_capacity, float* __restrict__ verts,
                                                      long long* __restrict__ keys, int* __restrict__ faces,
                                                      const u64* __restrict__ ghost_cells_dev, u64* __restrict__ report) {
  const int64_t ncells = min((int64_t)*ncells_dev, capacity);
  const u64 grand_total = *grand_total_dev;
  // The counts the host needs, written by the LAST kernel of the chain straight into page-locked host memory (`report`):
  // active cells, how many of them are ghost cells, (vertices << 32 | triangles), and the vertices owned by ghost cells
  // = the vertex prefix of list entry `nghost`.  The host used to fetch these with three 8-byte copies into pageable
  // memory plus two more for the ghost prefix -- each a blocking round trip -- in the middle of every extraction.
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const u64 nc = *ncells_dev, ng = *ghost_cells_dev;
    u64 foreign = 0;
    if ((int64_t)nc <= capacity && ng > 0) {
      if (ng >= nc) foreign = grand_total >> 32;
      else foreign = (block_offs[ng >> 8] >> 32) + (info[ng] >> 20);
    }
    report[0] = nc;
    report[1] = ng;
    report[2] = grand_total;
    report[3] = foreign;
  }
  if ((int64_t)*ncells_dev > capacity) return;  // the host sees the same totals and runs again with room
  if ((int64_t)(grand_total >> 32) > verts_capacity || (int64_t)(grand_total & 0xFFFFFFFFull) > faces_capacity) return;
  __shared__ int sm[4];
  __shared__ float sv[3 * kEmitMaxVerts];
  __shared__ long long sk[2 * kEmitMaxVerts];
  __shared__ int sf[3 * kEmitMaxTris];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int li = 0, cy = 0, x = 0, owned = 0, code = 0, ntri = 0;
  uint32_t inf = 0;
  if (i < ncells) {
    const int64_t slot = (int64_t)cell_list[i];
    int w;
    decode_word(p, slot >> 6, &li, &cy, &w);
    x = w * 64 + (int)(slot & 63);
    inf = info[i];
    owned = inf & 0xFFF;
    code = (inf >> 12) & 0xFF;
    if (li > 0) ntri = T->ntri[code];
  }
  int tot_t;
  const int tri_off = block_exclusive_scan(ntri, &tot_t, sm);
  // this block's ranges of the output arrays (uniform)
  const u64 boff = block_offs[blockIdx.x];
  const u64 bnext = (blockIdx.x + 1 < gridDim.x) ? block_offs[blockIdx.x + 1] : grand_total;
  const int64_t vb = (int64_t)(boff >> 32), fb = (int64_t)(boff & 0xFFFFFFFFull);
  const int tot_v = (int)((int64_t)(bnext >> 32) - vb);
  const bool staged = tot_v <= kEmitMaxVerts && tot_t <= kEmitMaxTris;
  if (i < ncells) {
    const int y = cy + 1, z = p.zc0 + li - 1;
    const int64_t slice = (int64_t)p.nx * p.ny;
    const int vin = (int)(inf >> 20);  // first vertex of this cell inside the block
    const int cut = cut_edges(code);

    // ---- gather phase: everything this cell reads from memory, requested before anything is used ----
    // (a) its 8 corner values and the 6 axis coordinates
    float cval[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
      cval[c] = p.sdf[(int64_t)(z + kCornerOff[c][2] - p.zs0) * slice + (int64_t)(y + kCornerOff[c][1]) * p.nx +
                      (x + kCornerOff[c][0])];
    const float ax2[2] = {p.px[x - 1], p.px[x]}, ay2[2] = {p.py[y - 1], p.py[y]}, az2[2] = {p.pz[z - 1], p.pz[z]};
    // (b) rank tables of this case: prec[code][e] for the 12 edges
    uint16_t prec[12];  // (24 bytes, 8-byte aligned: three loads instead of twelve)
    {
      const u64* pw = reinterpret_cast<const u64*>(&T->prec[code][0]);
      const u64 pw0 = pw[0], pw1 = pw[1], pw2 = pw[2];
#pragma unroll
      for (int e = 0; e < 12; ++e) prec[e] = (uint16_t)((e < 4 ? pw0 : (e < 8 ? pw1 : pw2)) >> (16 * (e & 3)));
    }
    //     ... and its triangle row (16 edge numbers) as one 16-byte load
    const uint4 trow = *reinterpret_cast<const uint4*>(&T->tri[code][0]);
    // (c) the neighbour cells that OWN the cut edges this cell does not.  Which of the nine neighbours are active is
    //     known from mc_owner (nbr_active); the owner of a foreign edge is its first active sharer in scan order, so
    //     the set of owners is bit logic -- typically one to three cells -- and only for those are the three dependent
    //     rounds (ACT word -> list offsets -> info word and block offset) issued, four owners together.  (Round 3 ran
    //     the rounds for all nine neighbours: 45 loads per cell, most of them to element 0.)
    const int foreign = cut & ~owned;
    const int nactive = (int)nbr_active[i];
    int owner_of[12];    // foreign edge e: neighbour index of the cell that owns it (-1: none), its edge number there
    int owner_edge[12];
    int owners = 0;
#pragma unroll
    for (int e = 0; e < 12; ++e) {
      owner_of[e] = -1;
      owner_edge[e] = 0;
      bool found = false;
#pragma unroll
      for (int k = 0; k < kShare[e].n; ++k) {
        const int q = kShareNbr[e][k];
        const bool hit = !found && ((foreign >> e) & 1) && ((nactive >> q) & 1);
        owner_of[e] = hit ? q : owner_of[e];
        owner_edge[e] = hit ? (int)kShare[e].e[k] : owner_edge[e];
        found = found || hit;
      }
      if (found) owners |= 1 << owner_of[e];
    }
    // (dx, dy, dl) of neighbour q, from kNbr packed two bits per entry (value + 1)
    auto nbr_d = [](int q, int axis) -> int {
      constexpr uint32_t kPack[3] = {
          (1u << 0) | (1u << 2) | (1u << 4) | (2u << 6) | (1u << 8) | (0u << 10) | (0u << 12) | (0u << 14) | (2u << 16),   // dx
          (0u << 0) | (1u << 2) | (0u << 4) | (1u << 6) | (2u << 8) | (1u << 10) | (1u << 12) | (0u << 14) | (0u << 16),   // dy
          (0u << 0) | (0u << 2) | (1u << 4) | (0u << 6) | (0u << 8) | (0u << 10) | (1u << 12) | (1u << 14) | (1u << 16)};  // dl
      return (int)((kPack[axis] >> (2 * q)) & 3u) - 1;
    };
    constexpr int kSlots = 4;
    int sq[kSlots];        // the owners taken in this pass (-1: none)
    uint32_t sinf[kSlots];
    u64 sboff[kSlots];
    {
      int left = owners;
      int64_t ocw[kSlots];
      int obit[kSlots];
      u64 aw[kSlots];
#pragma unroll
      for (int t = 0; t < kSlots; ++t) {
        sq[t] = left ? (__ffs(left) - 1) : -1;
        left &= left - 1;
        const int q = max(sq[t], 0);
        const int nl = li + nbr_d(q, 2), ncy = cy + nbr_d(q, 1), ox = x + nbr_d(q, 0);
        // (an active neighbour lies inside the grid: mc_owner only counts those)
        ocw[t] = sq[t] >= 0 ? word_index(p, nl, ncy, ox >> 6) : 0;
        obit[t] = ox & 63;
        aw[t] = act[ocw[t]];
      }
      uint32_t wco[kSlots];
      u64 bco[kSlots];
#pragma unroll
      for (int t = 0; t < kSlots; ++t) {
        wco[t] = word_cell_off[ocw[t]];
        bco[t] = block_cell_offs[ocw[t] >> 8];
      }
#pragma unroll
      for (int t = 0; t < kSlots; ++t) {
        const int64_t oi = sq[t] >= 0 ? (int64_t)bco[t] + wco[t] + __popcll(aw[t] & ((1ull << obit[t]) - 1ull)) : 0;
        sinf[t] = info[oi];
        sboff[t] = block_offs[oi >> 8];
      }
    }
    // a fifth and further owner (rare: a cell most of whose cut edges belong to different earlier cells), one at a time
    auto fetch_owner = [&](int q, uint32_t* oinf, u64* oboff) {
      const int nl = li + nbr_d(q, 2), ncy = cy + nbr_d(q, 1), ox = x + nbr_d(q, 0);
      const int64_t cw = word_index(p, nl, ncy, ox >> 6);
      const int64_t oi = list_index_of(act, word_cell_off, block_cell_offs, cw, ox & 63);
      *oinf = info[oi];
      *oboff = block_offs[oi >> 8];
    };

    // ---- vertices of the edges this cell owns ----------------------------------------------------
#pragma unroll
    for (int e = 0; e < 12; ++e) {
      if (!(owned & (1 << e))) continue;
      const int ca = kEdgeA[e], cb = kEdgeB[e];
      const float pa[3] = {ax2[kCornerOff[ca][0] + 1], ay2[kCornerOff[ca][1] + 1], az2[kCornerOff[ca][2] + 1]};
      const float pb[3] = {ax2[kCornerOff[cb][0] + 1], ay2[kCornerOff[cb][1] + 1], az2[kCornerOff[cb][2] + 1]};
      float out[3];
      vertex_interp(p.iso, pa, pb, cval[ca], cval[cb], p.linear != 0, out);
      const int r = vin + __popc(owned & prec[e]);
      const int ka = kKeyA[e], kb = kKeyB[e];
      const long long k0 = (int64_t)(z + kCornerOff[ka][2]) * slice + (int64_t)(y + kCornerOff[ka][1]) * p.nx +
                           (x + kCornerOff[ka][0]);
      const long long k1 = (int64_t)(z + kCornerOff[kb][2]) * slice + (int64_t)(y + kCornerOff[kb][1]) * p.nx +
                           (x + kCornerOff[kb][0]);
      if (staged) {
        sv[3 * r + 0] = out[0];
        sv[3 * r + 1] = out[1];
        sv[3 * r + 2] = out[2];
        sk[2 * r + 0] = k0;
        sk[2 * r + 1] = k1;
      } else {
        const int64_t vid = vb + r;
        verts[3 * vid + 0] = out[0];
        verts[3 * vid + 1] = out[1];
        verts[3 * vid + 2] = out[2];
        if (keys != nullptr) {
          keys[2 * vid + 0] = k0;
          keys[2 * vid + 1] = k1;
        }
      }
    }

    // ---- the vertex of every cut edge: this cell's, or the first active sharer's (scan order) -------
    int evid[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) {
      int vid = -1;
      if (cut & (1 << e)) {
        if (owned & (1 << e)) {
          vid = (int)(vb + vin + __popc(owned & prec[e]));
        } else if (owner_of[e] >= 0) {
          uint32_t oi_inf = 0;
          u64 oi_boff = 0;
          bool have = false;
#pragma unroll
          for (int t = 0; t < kSlots; ++t) {
            const bool m = sq[t] == owner_of[e];
            oi_inf = m ? sinf[t] : oi_inf;
            oi_boff = m ? sboff[t] : oi_boff;
            have = have || m;
          }
          if (!have) fetch_owner(owner_of[e], &oi_inf, &oi_boff);
          const int o_owned = oi_inf & 0xFFF, o_code = (oi_inf >> 12) & 0xFF, o_in_block = oi_inf >> 20;
          vid = (int)((int64_t)(oi_boff >> 32) + o_in_block + __popc(o_owned & T->prec[o_code][owner_edge[e]]));
        }
      }
      evid[e] = vid;
    }

    // ---- triangles, marching_cubes.cc:199-218 (ghost cells have ntri == 0) -------------------------
    for (int t = 0; t < ntri; ++t) {
      for (int j = 0; j < 3; ++j) {
        const int idx = 3 * t + (2 - j);
        const uint32_t word = (idx < 4) ? trow.x : (idx < 8) ? trow.y : (idx < 12) ? trow.z : trow.w;
        const int e = (int)((word >> (8 * (idx & 3))) & 0xFFu);
        int vid = -1;
#pragma unroll
        for (int q = 0; q < 12; ++q) vid = (e == q) ? evid[q] : vid;
        if (staged) sf[3 * (tri_off + t) + j] = vid;
        else faces[3 * (fb + tri_off + t) + j] = vid;
      }
    }
  }
  if (!staged) return;  // uniform
  __syncthreads();
  // whole rows of dwords: 3 floats per vertex, 2 x 8 bytes per key pair, 3 ints per triangle
  float* gv = verts + 3 * vb;
  for (int k = threadIdx.x; k < 3 * tot_v; k += 256) gv[k] = sv[k];
  if (keys != nullptr) {  // (null: the caller does not merge slabs, "meshkeys" 0)
    long long* gk = keys + 2 * vb;
    for (int k = threadIdx.x; k < 2 * tot_v; k += 256) gk[k] = sk[k];
  }
  int* gf = faces + 3 * fb;
  for (int k = threadIdx.x; k < 3 * tot_t; k += 256) gf[k] = sf[k];
}

}  // namespace

// the same scan for the other compactions of the library (extract_voxel.hip)
int device_exclusive_scan_u64(unsigned long long* d, int64_t n, unsigned long long* d_total,
                              unsigned long long* scratch, hipStream_t stream) {
  return exclusive_scan_u64(d, n, d_total, scratch, stream);
}

// ---- host driver ----------------------------------------------------------------------------

int extract_iso(vcy_ctx* c, double iso, int linear_interp, vcy_mesh* out) {
  out->n_vertices = out->n_faces = out->n_foreign_vertices = 0;
  out->vertices = nullptr;  // an empty mesh has no arrays
  out->faces = nullptr;
  out->edge_keys = nullptr;
  if (c->halo_lo && !c->halo_valid) {
    set_error("halo slices not installed: call vcy_halo_pack / all-gather / vcy_halo_unpack first");
    return VCY_ERR_NOT_INITIALIZED;
  }
  McParams p;
  p.sdf = c->d_sdf;
  p.cnt = c->d_cnt;
  p.px = c->d_px;
  p.py = c->d_py;
  p.pz = c->d_pz;
  p.nx = c->nx;
  p.ny = c->ny;
  p.nslices = c->halo_lo + c->nz_local();
  p.Wr = (c->nx + 63) / 64;
  p.Y = c->ny - 1;
  p.zc0 = std::max(c->z0, 1);
  p.L = c->z1 - p.zc0;
  p.zs0 = c->z0 - c->halo_lo;
  p.has_ghost = c->halo_lo > 0 ? 1 : 0;
  p.iso = iso;
  p.linear = linear_interp;
  c->last_extract_device_ms = 0.0f;
  if (c->nx < 2 || p.Y <= 0 || p.L <= 0) return VCY_OK;  // no cells (reference loops do not run)
  // One sweep (mc_sweep_kernel) needs a voxel row that is a power-of-two number of whole words.  It moves 2 % fewer
  // bytes than the bit planes in memory (mc_bits + mc_active) but is not faster anywhere (sweep / planes, one box:
  // 256^3 0.172 / 0.133 ms, 512^3 0.291 / 0.260, 1024^3 1.26-1.27 / 1.23-1.24, 2048^3 9.55 / 9.18), so it is
  // taken only on request ("mcsweep" 1).
  const bool sweep = c->mc_sweep && c->nx == p.Wr * 64 && (p.Wr & (p.Wr - 1)) == 0 && p.Wr <= 32;
  SweepParams q{};
  p.Yc = p.Y;
  if (sweep) {
    q.R = std::max(32, kWordsPerBlock / p.Wr);
    q.K = q.R * p.Wr / kWordsPerBlock;  // <= kSweepMaxK
    p.Yc = (p.Y + q.R - 1) / q.R * q.R;
    q.groups = p.Yc / q.R;
    // enough workgroups to fill the GPU, few enough that the slice two z chunks share stays a small part
    const int64_t want = ((int64_t)(p.L + 1) * q.groups + VCY_SWEEP_TARGET_WGS - 1) / VCY_SWEEP_TARGET_WGS;
    q.layers = (int)std::min<int64_t>(std::max<int64_t>(want, 8), 64);
    q.dl = p.zs0 - p.zc0 + 1;
    q.cnt_slices = c->cnt_implied ? 0 : p.nslices;
    while ((1 << q.wshift) < p.Wr) ++q.wshift;
  }
  const int64_t ghost_words = (int64_t)p.Yc * p.Wr;
  p.G = (ghost_words + kWordsPerBlock - 1) / kWordsPerBlock * kWordsPerBlock;
  p.nwords = p.G + (int64_t)p.L * p.Yc * p.Wr;
  {
    auto make_div = [](uint32_t d) {
      FastDiv f;
      uint32_t l = 0;
      while ((1ull << l) < d) ++l;  // ceil(log2 d)
      f.d = d;
      f.m = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
      f.s1 = l < 1 ? l : 1;
      f.s2 = l < 1 ? 0 : l - 1;
      return f;
    };
    p.small32 = p.nwords < 0xffffffffLL && (int64_t)p.Yc * p.Wr < 0x7fffffffLL ? 1 : 0;
    p.div_row = make_div((uint32_t)p.Wr);
    p.div_layer = make_div(p.small32 ? (uint32_t)((int64_t)p.Yc * p.Wr) : 1u);
  }
  const int64_t nblocks64 = (p.nwords + kWordsPerBlock - 1) / kWordsPerBlock;
  const int64_t vox_rows = (int64_t)p.nslices * c->ny;
  const int64_t vox_words = vox_rows * p.Wr;
  if (nblocks64 > 0x7fffffffLL || (vox_words + 3) / 4 > 0x7fffffffLL) {
    set_error("too many cells for one launch");
    return VCY_ERR_TOO_MANY_VOXELS;
  }
  const unsigned nblocks = (unsigned)nblocks64;
  hipStream_t s = c->stream;

  if (!c->d_mc_tables) {
    McTables h;
    build_tables(&h);
    VCY_HIP_CHECK(hipMalloc(&c->d_mc_tables, sizeof(McTables)));
    VCY_HIP_CHECK(hipMemcpy(c->d_mc_tables, &h, sizeof(McTables), hipMemcpyHostToDevice));
  }
  const McTables* T = (const McTables*)c->d_mc_tables;

  // scratch, cached in the context (grown on demand): bit planes, ACT, per-word offsets, block counts
  // (3 bits per voxel + 12.5 B per 64 cells: 0.9 GB at 1024^3)
  auto align = [](size_t v) { return (v + 255) / 256 * 256; };
  const size_t sz_plane = align(sizeof(u64) * (size_t)vox_words);
  const size_t sz_act = align(sizeof(u64) * (size_t)p.nwords);
  const size_t sz_woff = align(sizeof(uint32_t) * (size_t)p.nwords);
  const size_t sz_counts = align(sizeof(u64) * ((size_t)nblocks + 1));
  const size_t sz_scan = align(sizeof(u64) * ((size_t)nblocks / 1024 + 64) * 2);
  const size_t sz_ghost = sweep ? align(sizeof(u64) * 3 * (size_t)c->ny * p.Wr) : 0;  // IN / OK / TC of one slice
  const size_t need = (sweep ? 1 : 3) * sz_plane + sz_ghost + sz_act + sz_woff + sz_counts + sz_scan + 256;
  if (c->mc_scratch_bytes < need) {
    VCY_HIP_CHECK(hipStreamSynchronize(s));
    if (c->d_mc_scratch) VCY_HIP_CHECK(hipFree(c->d_mc_scratch));
    c->d_mc_scratch = nullptr;
    c->mc_scratch_bytes = 0;
    VCY_HIP_CHECK(hipMalloc(&c->d_mc_scratch, need));
    c->mc_scratch_bytes = need;
  }
  char* base = (char*)c->d_mc_scratch;
  u64* d_in = (u64*)base;                     base += sz_plane;
  u64* d_ok = (u64*)base;                     base += sweep ? 0 : sz_plane;  // (the sweep keeps OK / TC in LDS)
  u64* d_tc = (u64*)base;                     base += sweep ? 0 : sz_plane;
  u64* d_ghost = (u64*)base;                  base += sz_ghost;
  u64* d_act = (u64*)base;                    base += sz_act;
  uint32_t* d_woff = (uint32_t*)base;         base += sz_woff;
  u64* d_wcounts = (u64*)base;                base += sz_counts;
  u64* d_scan = (u64*)base;                   base += sz_scan;
  // publication flags of the chained scans (scan_chained_kernel): an allocation of their own, zeroed once -- they
  // must never hold a FUTURE epoch, so they do not live in scratch whose layout changes with the extraction
  if (!c->d_mc_flags) {
    // [2 slots][kChainedScanMaxChunks] flags, then the two ticket counters
    const size_t fbytes = sizeof(uint32_t) * (2 * (size_t)kChainedScanMaxChunks + 2);
    VCY_HIP_CHECK(hipMalloc(&c->d_mc_flags, fbytes));
    VCY_HIP_CHECK(hipMemsetAsync(c->d_mc_flags, 0, fbytes, s));
    c->mc_scan_epoch = 0;
    c->mc_scan_tickets[0] = c->mc_scan_tickets[1] = 0;
  }
  uint32_t* d_flags = (uint32_t*)c->d_mc_flags;
  auto scan_slot = [&](int which) {
    ChainedScanSlot sl;
    sl.flags = d_flags + which * kChainedScanMaxChunks;
    sl.ticket = d_flags + 2 * kChainedScanMaxChunks + which;
    sl.tickets_drawn = &c->mc_scan_tickets[which];
    sl.epoch = ++c->mc_scan_epoch;
    return sl;
  };
  u64* d_total = (u64*)base;
  p.in = d_in;
  p.ok = d_ok;
  p.tc = d_tc;

  float* d_verts = nullptr;
  long long* d_keys = nullptr;
  int* d_faces = nullptr;
  auto cleanup = [&]() {};
#define MC_TRY(expr)                                                               \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      set_error("%s failed: %s", #expr, hipGetErrorString(_e));                    \
      cleanup();                                                                   \
      return VCY_ERR_HIP;                                                          \
    }                                                                              \
  } while (0)

  // the extraction has its own event pair: vcy_timer_begin / _end may bracket it
  if (!c->ev_mc_begin) {
    MC_TRY(hipEventCreate(&c->ev_mc_begin));
    MC_TRY(hipEventCreate(&c->ev_mc_end));
  }
  MC_TRY(hipEventRecord(c->ev_mc_begin, s));
  const bool iso_f32 = (double)(float)iso == iso;
  // the halo slices come from another context: their update_num is read; the owned slices need it
  // only if the state was ever set from outside (vcy_upload), see mc_bits_kernel
  auto launch_bits = [&](int64_t word0, int64_t nw, bool tc_from_ok, u64* o_in, u64* o_ok, u64* o_tc) {
    if (nw <= 0) return;
    const unsigned blocks = (unsigned)((nw + 4 * kBitsWordsPerWave - 1) / (4 * kBitsWordsPerWave));
    const float* sdf0 = c->d_sdf + word0 * 64;  // whole rows: only used when nx == Wr * 64 or word0 == 0
    const char* cnt0 = (const char*)c->d_cnt + word0 * 64 * c->cnt_bytes;
#define VCY_BITS(CT, F32, TCOK)                                                                          \
  hipLaunchKernelGGL((mc_bits_kernel<CT, F32, TCOK>), dim3(blocks), dim3(256), 0, s, sdf0, (const CT*)cnt0, \
                     c->nx, p.Wr, nw, iso, o_in, o_ok, o_tc)
#define VCY_BITS_F(CT, TCOK)                                                    \
  do {                                                                          \
    if (iso_f32) VCY_BITS(CT, true, TCOK); else VCY_BITS(CT, false, TCOK);      \
  } while (0)
    if (tc_from_ok) {
      if (c->cnt_bytes == 1) VCY_BITS_F(uint8_t, true);
      else if (c->cnt_bytes == 2) VCY_BITS_F(uint16_t, true);
      else VCY_BITS_F(uint32_t, true);
    } else {
      if (c->cnt_bytes == 1) VCY_BITS_F(uint8_t, false);
      else if (c->cnt_bytes == 2) VCY_BITS_F(uint16_t, false);
      else VCY_BITS_F(uint32_t, false);
    }
#undef VCY_BITS_F
#undef VCY_BITS
  };
  if (sweep) {
    const bool reads_cnt = q.cnt_slices > 0;
    // TC words of the slice the ghost layer's max corners lie in (stored slice 1), from its update_num
    const u64* d_tc_ghost = nullptr;
    if (!reads_cnt && c->halo_lo > 0) {
      const int64_t row_words = (int64_t)c->ny * p.Wr;
      launch_bits(row_words, row_words, false, d_ghost, d_ghost + row_words, d_ghost + 2 * row_words);
      d_tc_ghost = d_ghost + 2 * row_words;
    }
    const unsigned chunks = (unsigned)((p.L + 1 + q.layers - 1) / q.layers);
    const size_t lds = sizeof(u64) * 2 * (reads_cnt ? 3 : 2) * (size_t)(q.R + 1) * p.Wr;
#define VCY_SWEEP(CT, F32, RC, KM)                                                                                   \
  hipLaunchKernelGGL((mc_sweep_kernel<CT, F32, RC, KM>), dim3(chunks * (unsigned)q.groups), dim3(256), lds, s, p, q, d_act, \
                     d_woff, d_wcounts, d_in, d_tc_ghost)
#define VCY_SWEEP_K(CT, F32, RC)                      \
  do {                                                \
    if (q.K <= 1) VCY_SWEEP(CT, F32, RC, 1);          \
    else if (q.K == 2) VCY_SWEEP(CT, F32, RC, 2);     \
    else VCY_SWEEP(CT, F32, RC, kSweepMaxK);          \
  } while (0)
#define VCY_SWEEP_F(CT, RC)                                                       \
  do {                                                                            \
    if (iso_f32) VCY_SWEEP_K(CT, true, RC); else VCY_SWEEP_K(CT, false, RC);      \
  } while (0)
    if (reads_cnt) {
      if (c->cnt_bytes == 1) VCY_SWEEP_F(uint8_t, true);
      else if (c->cnt_bytes == 2) VCY_SWEEP_F(uint16_t, true);
      else VCY_SWEEP_F(uint32_t, true);
    } else {
      VCY_SWEEP_F(uint16_t, false);  // (update_num is not read)
    }
#undef VCY_SWEEP_F
#undef VCY_SWEEP_K
#undef VCY_SWEEP
  } else {
    const int64_t halo_words = (int64_t)c->halo_lo * c->ny * p.Wr;
    // a whole grid whose state implies TC == OK: no third plane at all
    const bool alias_tc = c->cnt_implied && c->nx == p.Wr * 64 && c->halo_lo == 0;
    if (alias_tc) {
      p.tc = d_ok;
      d_tc = nullptr;
    }
    if (c->cnt_implied && c->nx == p.Wr * 64) {
      launch_bits(0, halo_words, false, d_in, d_ok, d_tc);
      const int nbw = (c->nx + 7) / 8, nby = (c->ny + 7) / 8, nbz = (c->nz_local() + 7) / 8;
      // (rows of 16 words and more: at 512^3 -- 8 words, 4096 brick rows -- the dense pass is the faster one, 0.197 against
      // 0.210 ms per extraction in the default mode, 0.233 against 0.271 after a weighted-average carve; "mcskip" 2 forces
      // the brick rows on any size for the tests)
      if (c->mc_skip && (p.Wr >= 16 || c->mc_skip > 1) && c->brick_min_valid && !c->fresh && c->d_brick_min &&
          nbw <= kBricksMaxNbw && nbz <= 65535) {
        // the owned slices, bricks the carve kernels left entirely outside the surface not read ("mcskip")
        const float* sdf0 = c->d_sdf + halo_words * 64;
        u64* o_tc = d_tc ? d_tc + halo_words : nullptr;
        if (iso_f32)
          hipLaunchKernelGGL((mc_bits_bricks_kernel<true>), dim3((unsigned)nby, (unsigned)nbz), dim3(256), 0, s, sdf0, c->ny,
                             c->nz_local(), p.Wr, iso, d_in + halo_words, d_ok + halo_words, o_tc, c->d_brick_min, nbw, nby);
        else
          hipLaunchKernelGGL((mc_bits_bricks_kernel<false>), dim3((unsigned)nby, (unsigned)nbz), dim3(256), 0, s, sdf0, c->ny,
                             c->nz_local(), p.Wr, iso, d_in + halo_words, d_ok + halo_words, o_tc, c->d_brick_min, nbw, nby);
      } else {
        launch_bits(halo_words, vox_words - halo_words, true, d_in + halo_words, d_ok + halo_words,
                    d_tc ? d_tc + halo_words : nullptr);
      }
    } else {
      launch_bits(0, vox_words, false, d_in, d_ok, d_tc);
    }
    hipLaunchKernelGGL(mc_active_kernel, dim3((nblocks + kActiveBlocks - 1) / kActiveBlocks), dim3(256), 0, s, p, d_act,
                       d_woff, d_wcounts, (int64_t)nblocks);
  }
  MC_TRY(hipGetLastError());
  int rc;
  {
    const ChainedScanSlot sl = scan_slot(0);
    rc = exclusive_scan_u64(d_wcounts, nblocks, d_total, d_scan, s, &sl);
  }
  if (rc != VCY_OK) return rc;

  // ---- the surface cells ------------------------------------------------------------------------
  // How much comes next is data: the number of active cells sizes the list and the owner info, the numbers of
  // vertices and triangles the output arrays.  The kernels read those counts from device memory, so with the
  // sizes of this context's previous extraction as a guess (plus a quarter) the whole chain is enqueued
  // without the host reading anything back; the counts are fetched once at the end, and if a guess was too
  // small the chain runs again with the exact sizes -- which is also the path of the first extraction.
  struct CellBuffers {
    u64* list; uint32_t* info; uint16_t* nact; u64* counts; u64* scan; u64* total; unsigned blocks;
  };
  auto cell_buffers = [&](int64_t cap_cells, CellBuffers* b) -> int {
    b->blocks = (unsigned)((cap_cells + 255) / 256);
    const size_t sz_list = align(sizeof(u64) * (size_t)cap_cells);
    const size_t sz_info = align(sizeof(uint32_t) * (size_t)cap_cells);
    const size_t sz_nact = align(sizeof(uint16_t) * (size_t)cap_cells);
    const size_t sz_cc = align(sizeof(u64) * ((size_t)b->blocks + 1));
    const size_t sz_cs = align(sizeof(u64) * ((size_t)b->blocks / 1024 + 64) * 2);
    const size_t need2 = sz_list + sz_info + sz_nact + sz_cc + sz_cs + 256;
    if (c->mc_cells_bytes < need2) {
      MC_TRY(hipStreamSynchronize(s));
      if (c->d_mc_cells) MC_TRY(hipFree(c->d_mc_cells));
      c->d_mc_cells = nullptr;
      c->mc_cells_bytes = 0;
      MC_TRY(hipMalloc(&c->d_mc_cells, need2));
      c->mc_cells_bytes = need2;
    }
    char* b2 = (char*)c->d_mc_cells;
    b->list = (u64*)b2;                   b2 += sz_list;
    b->info = (uint32_t*)b2;              b2 += sz_info;
    b->nact = (uint16_t*)b2;              b2 += sz_nact;
    b->counts = (u64*)b2;                 b2 += sz_cc;
    b->scan = (u64*)b2;                   b2 += sz_cs;
    b->total = (u64*)b2;
    return VCY_OK;
  };
  // active cells -> list -> owner info + (vertices, triangles) per block -> offsets
  auto enqueue_owners = [&](const CellBuffers& b, int64_t cap_cells) -> int {
    hipLaunchKernelGGL(mc_compact_kernel, dim3((nblocks + kCompactBlocks - 1) / kCompactBlocks), dim3(256), 0, s, p, d_act,
                       d_woff, d_wcounts, d_total, (int64_t)nblocks, b.list, cap_cells);
    hipLaunchKernelGGL(mc_owner_kernel, dim3(b.blocks), dim3(256), 0, s, p, T, d_act, b.list, d_total, cap_cells, b.info,
                       b.nact, b.counts);
    MC_TRY(hipGetLastError());
    const ChainedScanSlot sl = scan_slot(1);
    return exclusive_scan_u64(b.counts, b.blocks, b.total, b.scan, s, &sl);
  };
  // The counts come back through 64 bytes of page-locked memory that mc_emit writes itself (see the kernel).
  if (!c->h_mc_report) {
    MC_TRY(hipHostMalloc((void**)&c->h_mc_report, 64, hipHostMallocPortable | hipHostMallocMapped));
    std::memset((void*)c->h_mc_report, 0, 64);
  }
  volatile u64* report = (volatile u64*)c->h_mc_report;
  const bool timing = c->mc_timing != 0;
  double t_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto now_us = []() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
  };
  const double t_begin = timing ? now_us() : 0.0;
  // Output arrays.  A mesh of up to "mcdirect" bytes (default 32 MiB) is written by mc_emit STRAIGHT into the page-locked
  // host arrays the caller receives (whole rows of dwords over PCIe while other blocks still compute): the call is then
  // one enqueue and one wait.  Larger meshes are staged in device memory and copied with exact sizes after the counts
  // are known (over-copying the guess's headroom would cost more than the second wait).  The kernel's stores cross PCIe
  // at 50 GB/s where the copy engine reaches 54, and what they overlap is the 65 us of emit arithmetic: 0.22 -> 0.20 ms
  // for a 5 MB mesh, 0.65 -> 0.62 for 23 MB (512^3), nothing for 92 MB (1024^3) -- profiles/r06/mc_wall.txt.
  bool direct = false;
  auto release_host = [&]() {
    mesh_host_free(out->vertices);
    mesh_host_free(out->faces);
    mesh_host_free(out->edge_keys);
    out->vertices = nullptr, out->faces = nullptr, out->edge_keys = nullptr;
  };
  auto enqueue_emit = [&](const CellBuffers& b, int64_t cap_cells, int64_t cap_v, int64_t cap_f) -> int {
    const size_t sz_v = align(sizeof(float) * 3 * (size_t)std::max<int64_t>(cap_v, 1));
    const size_t sz_k = align(sizeof(long long) * 2 * (size_t)std::max<int64_t>(cap_v, 1));
    const size_t sz_f = align(sizeof(int) * 3 * (size_t)std::max<int64_t>(cap_f, 1));
    direct = false;
    if ((int64_t)(sz_v + sz_f + (c->mesh_keys ? sz_k : 0)) <= c->mc_direct_bytes) {
      bool pinned = true, pk = true, pf = true;
      out->vertices = (float*)mesh_host_alloc(sz_v, &pinned);
      out->faces = (int32_t*)mesh_host_alloc(sz_f, &pf);
      if (c->mesh_keys) out->edge_keys = (int64_t*)mesh_host_alloc(sz_k, &pk);
      direct = out->vertices && out->faces && (!c->mesh_keys || out->edge_keys) && pinned && pf && pk;
      if (!direct) release_host();
    }
    if (direct) {
      d_verts = out->vertices;
      d_keys = c->mesh_keys ? (long long*)out->edge_keys : nullptr;
      d_faces = (int*)out->faces;
    } else {
      if (c->mc_out_bytes < sz_v + sz_k + sz_f) {
        MC_TRY(hipStreamSynchronize(s));
        if (c->d_mc_out) MC_TRY(hipFree(c->d_mc_out));
        c->d_mc_out = nullptr;
        c->mc_out_bytes = 0;
        MC_TRY(hipMalloc(&c->d_mc_out, sz_v + sz_k + sz_f));
        c->mc_out_bytes = sz_v + sz_k + sz_f;
      }
      d_verts = (float*)c->d_mc_out;
      d_keys = c->mesh_keys ? (long long*)((char*)c->d_mc_out + sz_v) : nullptr;
      d_faces = (int*)((char*)c->d_mc_out + sz_v + sz_k);
    }
    hipLaunchKernelGGL(mc_emit_kernel, dim3(b.blocks), dim3(256), 0, s, p, T, d_act, b.list, d_total, cap_cells, d_woff,
                       d_wcounts, b.info, b.nact, b.counts, b.total, cap_v, cap_f, d_verts, d_keys, d_faces,
                       d_wcounts + p.G / kWordsPerBlock, (u64*)c->h_mc_report);
    MC_TRY(hipGetLastError());
    return VCY_OK;
  };
  // number of active cells, and how many of them are ghost cells (words below G)
  int64_t ncells = 0, nghost = 0, nv = 0, nf = 0, nforeign = 0;
  CellBuffers cb{};
  bool done = false, end_recorded = false;
  auto with_headroom = [](int64_t v) { return v + v / 4 + 4096; };  // the next view's mesh is a little different
  auto read_report = [&]() {
    ncells = (int64_t)report[0];
    nghost = (int64_t)report[1];
    nv = (int64_t)(report[2] >> 32);
    nf = (int64_t)(report[2] & 0xFFFFFFFFull);
    nforeign = (int64_t)report[3];
  };
  if (timing) t_ph[0] = now_us();
  if (c->mc_hint_cells > 0) {
    const int64_t cap_cells = with_headroom(c->mc_hint_cells);
    const int64_t cap_v = with_headroom(c->mc_hint_verts), cap_f = with_headroom(c->mc_hint_faces);
    rc = cell_buffers(cap_cells, &cb);
    if (rc == VCY_OK) rc = enqueue_owners(cb, cap_cells);
    if (rc == VCY_OK) rc = enqueue_emit(cb, cap_cells, cap_v, cap_f);
    if (rc != VCY_OK) {
      release_host();
      return rc;
    }
    // (the end of the kernels: last_extract_device_ms is "kernels only")
    MC_TRY(hipEventRecord(c->ev_mc_end, s));
    end_recorded = true;
    if (timing) t_ph[1] = now_us();
    MC_TRY(hipStreamSynchronize(s));  // the ONE wait of an extraction whose mesh went straight to host memory
    if (timing) t_ph[2] = now_us();
    read_report();
    done = ncells <= cap_cells && nv <= cap_v && nf <= cap_f;
    if (ncells == 0) nv = nf = 0;
    if (!done) release_host();
  }
  if (!done) {
    // first extraction of a context, or a guess that was too small: the counts first, then buffers of the right size
    MC_TRY(hipMemcpyAsync((void*)&report[0], d_total, sizeof(u64), hipMemcpyDeviceToHost, s));
    MC_TRY(hipMemcpyAsync((void*)&report[1], d_wcounts + p.G / kWordsPerBlock, sizeof(u64), hipMemcpyDeviceToHost, s));
    MC_TRY(hipStreamSynchronize(s));
    ncells = (int64_t)report[0];
    nghost = (int64_t)report[1];
    nv = nf = nforeign = 0;
    if (ncells > 0xFFFFFFFFLL) {
      set_error("too many surface cells");
      return VCY_ERR_TOO_MANY_VOXELS;
    }
    if (ncells > 0) {
      // (buffers sized with the same headroom as the guesses, so that the next extraction does not reallocate)
      const int64_t cap_cells = with_headroom(ncells);
      rc = cell_buffers(cap_cells, &cb);
      if (rc == VCY_OK) rc = enqueue_owners(cb, cap_cells);
      if (rc != VCY_OK) return rc;
      MC_TRY(hipMemcpyAsync((void*)&report[2], cb.total, sizeof(u64), hipMemcpyDeviceToHost, s));
      MC_TRY(hipStreamSynchronize(s));
      nv = (int64_t)(report[2] >> 32);
      nf = (int64_t)(report[2] & 0xFFFFFFFFull);
      rc = enqueue_emit(cb, cap_cells, with_headroom(nv), with_headroom(nf));
      if (rc != VCY_OK) {
        release_host();
        return rc;
      }
      MC_TRY(hipEventRecord(c->ev_mc_end, s));
      end_recorded = true;
      MC_TRY(hipStreamSynchronize(s));
      read_report();
    } else {
      end_recorded = false;
    }
  }
  if (ncells > 0) out->n_foreign_vertices = nforeign;
  c->mc_hint_cells = ncells;
  c->mc_hint_verts = nv;
  c->mc_hint_faces = nf;
  if (!end_recorded) MC_TRY(hipEventRecord(c->ev_mc_end, s));
  MC_TRY(hipEventSynchronize(c->ev_mc_end));
  MC_TRY(hipEventElapsedTime(&c->last_extract_device_ms, c->ev_mc_begin, c->ev_mc_end));
  if (timing) t_ph[3] = now_us();

  if (direct && ncells > 0 && (nv > 0 || nf > 0)) {
    // the arrays are already where the caller reads them; an empty side has no array
    if (nv == 0) {
      mesh_host_free(out->vertices);
      mesh_host_free(out->edge_keys);
      out->vertices = nullptr, out->edge_keys = nullptr;
    }
    if (nf == 0) {
      mesh_host_free(out->faces);
      out->faces = nullptr;
    }
  } else {
    if (direct) release_host();  // (an empty mesh)
    // the mesh arrays: page-locked host buffers, three DMAs in flight on the context's stream
    if (nv > 0) {
      out->vertices = (float*)mesh_host_alloc(sizeof(float) * 3 * (size_t)nv);
      if (c->mesh_keys) out->edge_keys = (int64_t*)mesh_host_alloc(sizeof(int64_t) * 2 * (size_t)nv);
    }
    if (nf > 0) out->faces = (int32_t*)mesh_host_alloc(sizeof(int32_t) * 3 * (size_t)nf);
    if ((nv > 0 && (!out->vertices || (c->mesh_keys && !out->edge_keys))) || (nf > 0 && !out->faces)) {
      set_error("out of host memory for the mesh");
      return VCY_ERR_INTERNAL;
    }
    if (nv > 0) {
      MC_TRY(hipMemcpyAsync(out->vertices, d_verts, sizeof(float) * 3 * (size_t)nv, hipMemcpyDeviceToHost, s));
      if (c->mesh_keys)
        MC_TRY(hipMemcpyAsync(out->edge_keys, d_keys, sizeof(long long) * 2 * (size_t)nv, hipMemcpyDeviceToHost, s));
    }
    if (nf > 0) MC_TRY(hipMemcpyAsync(out->faces, d_faces, sizeof(int) * 3 * (size_t)nf, hipMemcpyDeviceToHost, s));
    if (nv > 0 || nf > 0) MC_TRY(hipStreamSynchronize(s));
  }
  out->n_vertices = nv;
  out->n_faces = nf;
  if (timing) {
    t_ph[4] = now_us();
    fprintf(stderr, "[vcy mc timing] setup %.1f us | enqueue %.1f | wait %.1f | events %.1f | mesh to host %.1f | total %.1f "
                    "(direct %d, %lld cells, %lld v, %lld f, kernels %.1f us)\n",
            t_ph[0] - t_begin, t_ph[1] - t_ph[0], t_ph[2] - t_ph[1], t_ph[3] - t_ph[2], t_ph[4] - t_ph[3], t_ph[4] - t_begin,
            direct ? 1 : 0, (long long)ncells, (long long)nv, (long long)nf, c->last_extract_device_ms * 1e3);
  }
#undef MC_TRY
  cleanup();
  return VCY_OK;
}

}  // namespace vcy
