// K2-K4: marching cubes on the device slab, reproducing the reference's serial scan.
//
// Replaces MarchingCubes() (reference src/vacancy/marching_cubes.cc:63-228).  The reference
// walks cells z,y,x from 1, deduplicates vertices with a std::map keyed by the voxel-id pair
// of the cut edge, numbers vertices in order of first reference and faces in scan order.
// The same numbering falls out of three data-parallel passes over cells in raster order:
//
//   classify  one thread per cell: validity (:88-112) + cube index (:121-128) -> 1 case byte,
//             triangles per 256-cell block.
//   owner     a cut edge belongs to the FIRST active cell (scan order) among the <= 4 cells
//             that share it -- that cell is where the reference's map insert happens, so it
//             also fixes the interpolation direction (Appendix D of SURVEY.md).  Each active
//             cell finds the edges it owns from its 9 earlier neighbours' case bytes;
//             wave prefix-sums give every cell its first vertex number inside the block.
//   scan      exclusive scan of the per-block (vertex, triangle) counts.
//   emit      owned edges -> VertexInterp in fp64 (:25-57) -> vertices + edge keys;
//             triangles -> vertex ids through the owner cell of every corner edge.
//
// Multi-GPU: the layer of cells below the slab (z = z0-1, owned by the previous rank) is
// classified from the two halo slices as a ghost layer; vertices it owns on the shared plane
// are emitted first and counted in n_foreign_vertices, so a host merge can map them onto the
// previous rank's numbering by edge key.
//
// Memory-bound, no MFMA: 4 B (sdf) per cell algorithmic read + 12 B per vertex/triangle out.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vcy_internal.h"

namespace vcy {

namespace {

// ---- tables ------------------------------------------------------------------------------
const char* const kCaseStrings[256] = {
#include "vacancy_mc_cases.inc"
};

struct McTables {
  int8_t tri[256][16];   // edge numbers, -1 terminated (reference kTriTable)
  uint8_t ntri[256];
  uint16_t prec[256][12];  // prec[c][e] = edges whose vertex the serial scan creates before e's
};

struct McScratch {
  McTables* d_tables = nullptr;
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
// number the edge has inside that cell.  count = 0 means the cell itself always owns it.
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

struct McParams {
  const float* sdf;   // slab incl. halo slices
  const void* cnt;
  const float* px;
  const float* py;
  const float* pz;
  int nx, ny;
  int X, Y;           // cells per row / column = nx-1, ny-1
  int L;              // own cell layers
  int zc0;            // global z of own layer 0 (its max-corner slice)
  int zs0;            // global z of stored slice 0 (= z0 - halo_lo)
  int has_ghost;      // ghost layer computed from halo slices (else all inactive)
  int64_t XY;
  int64_t G;          // cells reserved for the ghost layer: XY rounded up to a whole block,
                      // so that own-layer cells (and their vertices) start on a block boundary
  int64_t ncells;     // G + L*XY, ghost layer first
  double iso;
  int linear;
};

__device__ __forceinline__ bool case_active(uint8_t c) { return c != 0 && c != 255; }

__device__ __forceinline__ int cut_edges(int c) {
  // an edge is cut iff its two corners differ (== reference kEdgeTable[c])
  int m = 0;
#pragma unroll
  for (int e = 0; e < 12; ++e) m |= (((c >> kEdgeA[e]) ^ (c >> kEdgeB[e])) & 1) << e;
  return m;
}

// false for the padding cells between the ghost layer and the first own layer
__device__ __forceinline__ bool decode_cell(const McParams& p, int64_t c, int* cx, int* cy, int* l) {
  int64_t r;
  if (c < p.G) {
    if (c >= p.XY) return false;
    *l = -1;
    r = c;
  } else {
    const int64_t layer = (c - p.G) / p.XY;
    r = (c - p.G) - layer * p.XY;
    *l = (int)layer;
  }
  *cy = (int)(r / p.X);
  *cx = (int)(r - (int64_t)(*cy) * p.X);
  return true;
}

__device__ __forceinline__ int64_t cell_index(const McParams& p, int cx, int cy, int l) {
  return (l < 0 ? 0 : p.G + (int64_t)l * p.XY) + (int64_t)cy * p.X + cx;
}

// ---- block-level exclusive scan of small per-thread counts (256 threads = 4 waves) ---------
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

// ---- pass 1: classify ---------------------------------------------------------------------
template <typename CountT>
__global__ __launch_bounds__(256) void mc_classify_kernel(McParams p, const McTables* __restrict__ T,
                                                          uint8_t* __restrict__ cases,
                                                          unsigned long long* __restrict__ block_counts) {
  __shared__ int sm[4];
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int ntri = 0;
  int cx, cy, l;
  if (c < p.ncells && decode_cell(p, c, &cx, &cy, &l)) {
    uint8_t code = 0;
    if (l >= 0 || p.has_ghost) {
      const int x = cx + 1, y = cy + 1, z = p.zc0 + l;  // max corner, global z
      const int64_t slice = (int64_t)p.nx * p.ny;
      const int64_t base6 = (int64_t)(z - p.zs0) * slice + (int64_t)y * p.nx + x;
      const CountT* cnt = (const CountT*)p.cnt;
      if ((int)cnt[base6] >= 1) {  // marching_cubes.cc:88-90
        const float* s1 = p.sdf + base6;        // slice z
        const float* s0 = s1 - slice;           // slice z-1
        float v[8];
        v[0] = s0[-p.nx - 1];
        v[1] = s0[-p.nx];
        v[2] = s0[0];
        v[3] = s0[-1];
        v[4] = s1[-p.nx - 1];
        v[5] = s1[-p.nx];
        v[6] = s1[0];
        v[7] = s1[-1];
        bool invalid = false;
        int bits = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          invalid |= (v[i] == kInvalidSdf);                      // :103-112
          bits |= ((double)v[i] < p.iso) ? (1 << i) : 0;         // :121-128
        }
        code = invalid ? 0 : (uint8_t)bits;
      }
    }
    cases[c] = code;
    if (l >= 0) ntri = T->ntri[code];
  } else if (c < p.ncells) {
    cases[c] = 0;  // padding
  }
  int total;
  (void)block_exclusive_scan(ntri, &total, sm);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = (unsigned long long)(unsigned)total;
}

// ---- pass 2: edge ownership ---------------------------------------------------------------
__device__ __forceinline__ bool neighbour_active(const McParams& p, const uint8_t* cases, int cx,
                                                 int cy, int l, int dx, int dy, int dl) {
  const int nx_ = cx + dx, ny_ = cy + dy, nl = l + dl;
  if (nx_ < 0 || nx_ >= p.X || ny_ < 0 || ny_ >= p.Y || nl < -1) return false;
  return case_active(cases[cell_index(p, nx_, ny_, nl)]);
}

__device__ __forceinline__ int owned_edges(const McParams& p, const uint8_t* cases, int code, int cx,
                                           int cy, int l) {
  const int cut = cut_edges(code);
  int owned = 0;
#pragma unroll
  for (int e = 0; e < 12; ++e) {
    if (!(cut & (1 << e))) continue;
    bool earlier = false;
    for (int k = 0; k < kShare[e].n; ++k)
      earlier |= neighbour_active(p, cases, cx, cy, l, kShare[e].d[k][0], kShare[e].d[k][1],
                                  kShare[e].d[k][2]);
    if (!earlier) owned |= 1 << e;
  }
  // a ghost cell only contributes vertices on the plane it shares with the slab (e4..e7)
  if (l < 0) owned &= 0xF0;
  return owned;
}

__global__ __launch_bounds__(256) void mc_owner_kernel(McParams p, const uint8_t* __restrict__ cases,
                                                       uint32_t* __restrict__ info,
                                                       unsigned long long* __restrict__ block_counts) {
  __shared__ int sm[4];
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int owned = 0;
  bool active = false;
  int cx, cy, l;
  if (c < p.ncells && decode_cell(p, c, &cx, &cy, &l)) {
    const uint8_t code = cases[c];
    active = case_active(code);
    if (active) owned = owned_edges(p, cases, code, cx, cy, l);
  }
  const int nv = __popc(owned);
  int total;
  const int off = block_exclusive_scan(nv, &total, sm);
  if (active) info[c] = (uint32_t)owned | ((uint32_t)off << 12);
  if (threadIdx.x == 0) block_counts[blockIdx.x] |= ((unsigned long long)(unsigned)total) << 32;
}

// ---- pass 3: exclusive scan of packed (verts<<32 | tris) block counts ---------------------
__global__ __launch_bounds__(256) void scan_chunks_kernel(unsigned long long* __restrict__ data,
                                                          int64_t n,
                                                          unsigned long long* __restrict__ chunk_sums) {
  // 1024 elements per block, 4 per thread
  __shared__ unsigned long long sm[256];
  const int64_t base = (int64_t)blockIdx.x * 1024 + (int64_t)threadIdx.x * 4;
  unsigned long long v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = (base + k < n) ? data[base + k] : 0ull;
    s += v[k];
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    unsigned long long t = (threadIdx.x >= d) ? sm[threadIdx.x - d] : 0ull;
    __syncthreads();
    sm[threadIdx.x] += t;
    __syncthreads();
  }
  unsigned long long run = sm[threadIdx.x] - s;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < n) data[base + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 255) chunk_sums[blockIdx.x] = sm[255];
}

__global__ __launch_bounds__(256) void add_chunk_offsets_kernel(unsigned long long* __restrict__ data,
                                                                int64_t n,
                                                                const unsigned long long* __restrict__ offs) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) data[i] += offs[i >> 10];
}

// in-place exclusive scan; *total (device) receives the grand total
int exclusive_scan_u64(unsigned long long* d, int64_t n, unsigned long long* d_total,
                       hipStream_t stream) {
  const int64_t nchunks = (n + 1023) / 1024;
  unsigned long long* sums = nullptr;
  VCY_HIP_CHECK(hipMalloc(&sums, sizeof(unsigned long long) * (size_t)(nchunks + 1)));
  hipLaunchKernelGGL(scan_chunks_kernel, dim3((unsigned)nchunks), dim3(256), 0, stream, d, n, sums);
  int rc = VCY_OK;
  if (nchunks > 1) {
    rc = exclusive_scan_u64(sums, nchunks, d_total, stream);
    if (rc == VCY_OK)
      hipLaunchKernelGGL(add_chunk_offsets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                         stream, d, n, sums);
  } else {
    hipError_t e = hipMemcpyAsync(d_total, sums, sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) rc = VCY_ERR_HIP;
  }
  hipError_t e1 = hipStreamSynchronize(stream);
  hipError_t e2 = hipFree(sums);
  if (rc == VCY_OK && (e1 != hipSuccess || e2 != hipSuccess || hipGetLastError() != hipSuccess)) {
    set_error("scan failed");
    rc = VCY_ERR_HIP;
  }
  return rc;
}

// ---- pass 4: emit -------------------------------------------------------------------------
__device__ __forceinline__ int64_t vertex_id_of(const McParams& p, const McTables* T,
                                                const uint8_t* cases, const uint32_t* info,
                                                const unsigned long long* block_offs, int64_t cell,
                                                int edge) {
  const uint32_t inf = info[cell];
  const int owned = inf & 0xFFF;
  const int local = inf >> 12;
  const int64_t block_base = (int64_t)(block_offs[cell >> 8] >> 32);
  return block_base + local + __popc(owned & T->prec[cases[cell]][edge]);
}

// VertexInterp, marching_cubes.cc:25-57 (fp64, then cast)
__device__ __forceinline__ void vertex_interp(double iso, const float pa[3], const float pb[3],
                                              float va, float vb, bool linear, float out[3]) {
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
  for (int k = 0; k < 3; ++k)
    out[k] = (float)((double)pa[k] + mu * ((double)pb[k] - (double)pa[k]));
}

__global__ __launch_bounds__(256) void mc_emit_kernel(McParams p, const McTables* __restrict__ T,
                                                      const uint8_t* __restrict__ cases,
                                                      const uint32_t* __restrict__ info,
                                                      const unsigned long long* __restrict__ block_offs,
                                                      float* __restrict__ verts,
                                                      long long* __restrict__ keys,
                                                      int* __restrict__ faces) {
  __shared__ int sm[4];
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint8_t code = 0;
  int cx = 0, cy = 0, l = 0;
  bool active = false;
  if (c < p.ncells && decode_cell(p, c, &cx, &cy, &l)) {
    code = cases[c];
    active = case_active(code);
  }
  const int ntri = (active && l >= 0) ? T->ntri[code] : 0;
  int total;
  const int tri_off = block_exclusive_scan(ntri, &total, sm);
  if (!active) return;

  const int x = cx + 1, y = cy + 1, z = p.zc0 + l;
  const int64_t slice = (int64_t)p.nx * p.ny;

  // vertices of the edges this cell owns
  const uint32_t inf = info[c];
  const int owned = inf & 0xFFF;
  if (owned) {
    const int64_t vbase = (int64_t)(block_offs[c >> 8] >> 32) + (inf >> 12);
    for (int e = 0; e < 12; ++e) {
      if (!(owned & (1 << e))) continue;
      const int a = kEdgeA[e], b = kEdgeB[e];
      const int ax = x + kCornerOff[a][0], ay = y + kCornerOff[a][1], az = z + kCornerOff[a][2];
      const int bx = x + kCornerOff[b][0], by = y + kCornerOff[b][1], bz = z + kCornerOff[b][2];
      const float pa[3] = {p.px[ax], p.py[ay], p.pz[az]};
      const float pb[3] = {p.px[bx], p.py[by], p.pz[bz]};
      const float va = p.sdf[(int64_t)(az - p.zs0) * slice + (int64_t)ay * p.nx + ax];
      const float vb = p.sdf[(int64_t)(bz - p.zs0) * slice + (int64_t)by * p.nx + bx];
      float out[3];
      vertex_interp(p.iso, pa, pb, va, vb, p.linear != 0, out);
      const int64_t vid = vbase + __popc(owned & T->prec[code][e]);
      verts[3 * vid + 0] = out[0];
      verts[3 * vid + 1] = out[1];
      verts[3 * vid + 2] = out[2];
      const int ka = kKeyA[e], kb = kKeyB[e];
      keys[2 * vid + 0] = (int64_t)(z + kCornerOff[ka][2]) * slice +
                          (int64_t)(y + kCornerOff[ka][1]) * p.nx + (x + kCornerOff[ka][0]);
      keys[2 * vid + 1] = (int64_t)(z + kCornerOff[kb][2]) * slice +
                          (int64_t)(y + kCornerOff[kb][1]) * p.nx + (x + kCornerOff[kb][0]);
    }
  }
  if (ntri == 0) return;

  // triangles, marching_cubes.cc:199-218
  const int64_t fbase = (int64_t)(block_offs[c >> 8] & 0xFFFFFFFFull) + tri_off;
  for (int t = 0; t < ntri; ++t) {
    for (int j = 0; j < 3; ++j) {
      const int e = T->tri[code][3 * t + (2 - j)];
      int64_t vid;
      if (owned & (1 << e)) {
        vid = vertex_id_of(p, T, cases, info, block_offs, c, e);
      } else {
        vid = -1;
        for (int k = 0; k < kShare[e].n; ++k) {
          const int dx = kShare[e].d[k][0], dy = kShare[e].d[k][1], dl = kShare[e].d[k][2];
          if (neighbour_active(p, cases, cx, cy, l, dx, dy, dl)) {
            const int64_t oc = cell_index(p, cx + dx, cy + dy, l + dl);
            vid = vertex_id_of(p, T, cases, info, block_offs, oc, kShare[e].e[k]);
            break;
          }
        }
      }
      faces[3 * (fbase + t) + j] = (int)vid;
    }
  }
}

}  // namespace

// ---- host driver ----------------------------------------------------------------------------

template <typename CountT>
static void launch_classify(const McParams& p, const McTables* T, uint8_t* cases,
                            unsigned long long* counts, unsigned nblocks, hipStream_t s) {
  hipLaunchKernelGGL((mc_classify_kernel<CountT>), dim3(nblocks), dim3(256), 0, s, p, T, cases, counts);
}

int extract_iso(vcy_ctx* c, double iso, int linear_interp, vcy_mesh* out) {
  out->n_vertices = out->n_faces = out->n_foreign_vertices = 0;
  out->vertices = (float*)std::malloc(sizeof(float) * 3);
  out->faces = (int32_t*)std::malloc(sizeof(int32_t) * 3);
  out->edge_keys = (int64_t*)std::malloc(sizeof(int64_t) * 2);
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
  p.X = c->nx - 1;
  p.Y = c->ny - 1;
  p.zc0 = std::max(c->z0, 1);
  p.L = c->z1 - p.zc0;
  p.zs0 = c->z0 - c->halo_lo;
  p.has_ghost = c->halo_lo > 0 ? 1 : 0;
  p.iso = iso;
  p.linear = linear_interp;
  c->last_extract_device_ms = 0.0f;
  if (p.X <= 0 || p.Y <= 0 || p.L <= 0) return VCY_OK;  // no cells (reference loops do not run)
  p.XY = (int64_t)p.X * p.Y;
  p.G = (p.XY + 255) / 256 * 256;
  p.ncells = p.G + (int64_t)p.L * p.XY;
  const int64_t nblocks64 = (p.ncells + 255) / 256;
  if (nblocks64 > 0x7fffffffLL) {
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

  uint8_t* cases = nullptr;
  uint32_t* info = nullptr;
  unsigned long long* counts = nullptr;
  unsigned long long* d_total = nullptr;
  float *d_verts = nullptr;
  long long* d_keys = nullptr;
  int* d_faces = nullptr;
  auto cleanup = [&]() {
    (void)hipFree(cases);
    (void)hipFree(info);
    (void)hipFree(counts);
    (void)hipFree(d_total);
    (void)hipFree(d_verts);
    (void)hipFree(d_keys);
    (void)hipFree(d_faces);
  };
#define MC_TRY(expr)                                                               \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      set_error("%s failed: %s", #expr, hipGetErrorString(_e));                    \
      cleanup();                                                                   \
      return VCY_ERR_HIP;                                                          \
    }                                                                              \
  } while (0)
  MC_TRY(hipMalloc(&cases, (size_t)p.ncells));
  MC_TRY(hipMalloc(&info, sizeof(uint32_t) * (size_t)p.ncells));
  MC_TRY(hipMalloc(&counts, sizeof(unsigned long long) * (size_t)nblocks));
  MC_TRY(hipMalloc(&d_total, sizeof(unsigned long long)));

  MC_TRY(hipEventRecord(c->ev_begin, s));
  if (c->cnt_bytes == 1) launch_classify<uint8_t>(p, T, cases, counts, nblocks, s);
  else if (c->cnt_bytes == 2) launch_classify<uint16_t>(p, T, cases, counts, nblocks, s);
  else launch_classify<uint32_t>(p, T, cases, counts, nblocks, s);
  hipLaunchKernelGGL(mc_owner_kernel, dim3(nblocks), dim3(256), 0, s, p, cases, info, counts);
  MC_TRY(hipGetLastError());
  int rc = exclusive_scan_u64(counts, nblocks, d_total, s);
  if (rc != VCY_OK) {
    cleanup();
    return rc;
  }
  unsigned long long total = 0;
  MC_TRY(hipMemcpy(&total, d_total, sizeof(total), hipMemcpyDeviceToHost));
  const int64_t nv = (int64_t)(total >> 32), nf = (int64_t)(total & 0xFFFFFFFFull);

  // vertices owned by ghost cells come first in scan order; own cells start at block G/256
  unsigned long long first_own = 0;
  MC_TRY(hipMemcpy(&first_own, counts + p.G / 256, sizeof(first_own), hipMemcpyDeviceToHost));
  out->n_foreign_vertices = (int64_t)(first_own >> 32);
  MC_TRY(hipMalloc(&d_verts, sizeof(float) * 3 * (size_t)std::max<int64_t>(nv, 1)));
  MC_TRY(hipMalloc(&d_keys, sizeof(long long) * 2 * (size_t)std::max<int64_t>(nv, 1)));
  MC_TRY(hipMalloc(&d_faces, sizeof(int) * 3 * (size_t)std::max<int64_t>(nf, 1)));
  hipLaunchKernelGGL(mc_emit_kernel, dim3(nblocks), dim3(256), 0, s, p, T, cases, info, counts, d_verts,
                     d_keys, d_faces);
  MC_TRY(hipGetLastError());
  MC_TRY(hipEventRecord(c->ev_end, s));
  MC_TRY(hipEventSynchronize(c->ev_end));
  MC_TRY(hipEventElapsedTime(&c->last_extract_device_ms, c->ev_begin, c->ev_end));

  std::free(out->vertices);
  std::free(out->faces);
  std::free(out->edge_keys);
  out->vertices = (float*)std::malloc(sizeof(float) * 3 * (size_t)std::max<int64_t>(nv, 1));
  out->faces = (int32_t*)std::malloc(sizeof(int32_t) * 3 * (size_t)std::max<int64_t>(nf, 1));
  out->edge_keys = (int64_t*)std::malloc(sizeof(int64_t) * 2 * (size_t)std::max<int64_t>(nv, 1));
  if (nv > 0) {
    MC_TRY(hipMemcpy(out->vertices, d_verts, sizeof(float) * 3 * (size_t)nv, hipMemcpyDeviceToHost));
    MC_TRY(hipMemcpy(out->edge_keys, d_keys, sizeof(long long) * 2 * (size_t)nv, hipMemcpyDeviceToHost));
  }
  if (nf > 0)
    MC_TRY(hipMemcpy(out->faces, d_faces, sizeof(int) * 3 * (size_t)nf, hipMemcpyDeviceToHost));
  out->n_vertices = nv;
  out->n_faces = nf;
#undef MC_TRY
  cleanup();
  return VCY_OK;
}

}  // namespace vcy
