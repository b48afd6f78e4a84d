/*
 * vacancy_hip.h -- C-ABI of the MI355X-native voxel-carving hot path.
 *
 * This is the drop-in boundary.  The reference (unclearness/vacancy) has no FFI
 * layer of its own: its hot path is reached through the C++ class
 * vacancy::VoxelCarver (include/vacancy/voxel_carver.h:95-118).  Every entry
 * point below names the reference member/function it replaces; the C++ facade in
 * include/vacancy/ (this repo) keeps the reference's class API and forwards to
 * these symbols (see INTEGRATION.md for the exact binding).
 *
 * Conventions
 *   - plain C types only; no exceptions cross the boundary;
 *   - every function returns VCY_OK (0) or a negative vcy_status; the text of the
 *     last error on the calling thread is vcy_last_error();
 *   - inputs are caller-owned and never retained past the call unless the
 *     function name says "_device" (then the pointer is a HIP device pointer that
 *     must stay valid until vcy_sync());
 *   - outputs that the library allocates are released with the matching *_free;
 *   - one context = one GPU = one caller thread.  A context owns a z-slab
 *     [z_begin, z_end) of the global grid (the whole grid when world size is 1).
 *   - there is NO CPU fallback: if no HIP device is usable, vcy_create fails.
 */
#ifndef VACANCY_HIP_H_
#define VACANCY_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum vcy_status {
  VCY_OK = 0,
  VCY_ERR_INVALID_ARG = -1,   /* reference: `return false` + LOGE in Init()/Carve() */
  VCY_ERR_NOT_INITIALIZED = -2,
  VCY_ERR_TOO_MANY_VOXELS = -3,
  VCY_ERR_HIP = -4,           /* a HIP runtime call failed */
  VCY_ERR_NO_DEVICE = -5,
  VCY_ERR_UNSUPPORTED = -6,
  VCY_ERR_INTERNAL = -7       /* vcy_selftest: a device-side identity a fast path relies on does not hold */
} vcy_status;

/* vacancy::VoxelUpdate / SdfInterpolation / UpdateOutsideImage
 * (include/vacancy/voxel_carver.h:20-38) */
enum { VCY_UPDATE_MAX = 0, VCY_UPDATE_WEIGHTED_AVERAGE = 1 };
enum { VCY_INTERP_NN = 0, VCY_INTERP_BILINEAR = 1 };
enum { VCY_OUTSIDE_NONE = 0, VCY_OUTSIDE_MAX = 1 };

/* vacancy::VoxelUpdateOption (voxel_carver.h:43-52), same defaults. */
typedef struct vcy_update_option {
  int32_t voxel_update;          /* VCY_UPDATE_*           default kMax      */
  int32_t sdf_interp;            /* VCY_INTERP_*           default kBilinear */
  int32_t update_outside;        /* VCY_OUTSIDE_*          default kNone     */
  int32_t voxel_max_update_num;  /*                        default 255       */
  float   voxel_update_weight;   /*                        default 1.0f      */
  int32_t use_truncation;        /* bool                   default false     */
  float   truncation_band;       /*                        default 0.1f      */
} vcy_update_option;

/* vacancy::VoxelCarverOption (voxel_carver.h:54-60). */
typedef struct vcy_carver_option {
  float bb_max[3];
  float bb_min[3];
  float resolution;              /* default 0.1f */
  int32_t sdf_minmax_normalize;  /* bool, default true */
  vcy_update_option update_option;
} vcy_carver_option;

/* Everything Carve() reads from `const Camera&` plus the ROI:
 *   w2c      = camera.w2c().cast<float>() (voxel_carver.cc:438), row-major 3x4
 *   fx,fy,cx,cy = PinholeCamera focal_length / principal_point (camera.cc:131-137)
 *   is_ortho = OrthoCamera::Project (camera.cc:201-205)
 *   roi_min/roi_max = the Vector2i arguments of Carve (voxel_carver.cc:415-416)
 *   width,height = sdf.width()/height() (image.h:65-74 layout: data[w*y+x]) */
typedef struct vcy_view {
  float   w2c[12];
  float   fx, fy, cx, cy;
  int32_t is_ortho;
  int32_t roi_min[2];
  int32_t roi_max[2];
  int32_t width, height;
} vcy_view;

/* Output of marching cubes: what MarchingCubes() hands to
 * Mesh::set_vertices / set_vertex_indices (marching_cubes.cc:223-224), plus the
 * dedup key of every vertex (the sorted voxel-id pair, marching_cubes.cc:78). */
typedef struct vcy_mesh {
  int64_t  n_vertices;
  int64_t  n_faces;
  float*   vertices;   /* 3 * n_vertices, xyz                         */
  int32_t* faces;      /* 3 * n_faces, indices into vertices          */
  int64_t* edge_keys;  /* 2 * n_vertices, (lower id, higher id), GLOBAL voxel ids; NULL with "meshkeys" 0 */
  /* Multi-GPU only (0 for a whole grid): the first n_foreign_vertices entries duplicate
   * vertices that the previous z-slab owns (edges on the shared plane z_begin-1); a merge
   * maps them onto that slab's numbering by edge key. */
  int64_t  n_foreign_vertices;
} vcy_mesh;

typedef struct vcy_ctx vcy_ctx;

/* ---- lifetime ----------------------------------------------------------- */

/* Replaces VoxelCarver::set_option + VoxelCarver::Init (voxel_carver.cc:373-392)
 * and VoxelGrid::Init (voxel_carver.cc:276-345): validates the options with the
 * reference's rules, sizes the grid n[i] = (int)((bb_max-bb_min)[i]/resolution),
 * allocates the slab z in [z_begin, z_end) on `device_id` (pass z_begin=0,
 * z_end=-1 for the whole grid) and sets sdf = lowest(), update_num = 0. */
int vcy_create(const vcy_carver_option* option, int device_id, int z_begin,
               int z_end, vcy_ctx** out);
void vcy_destroy(vcy_ctx* ctx);

/* VoxelGrid::voxel_num() (voxel_carver.cc:347): global dims (nx,ny,nz). */
int vcy_grid_dims(const vcy_ctx* ctx, int32_t dims[3]);
/* The z-range this context owns. */
int vcy_slab_range(const vcy_ctx* ctx, int32_t z_range[2]);
/* Grid dims without creating a context (host arithmetic of VoxelGrid::Init, voxel_carver.cc:278-301).  A box thinner
 * than one voxel along an axis gives dims[axis] = 0 and VCY_OK -- the reference builds an empty grid there --; only
 * vcy_create refuses it (there is nothing to put in HBM). */
int vcy_compute_dims(const float bb_min[3], const float bb_max[3],
                     float resolution, int32_t dims[3]);

/* Voxel::pos along one axis (0 = x, 1 = y, 2 = z): out[i], i < dims[axis], as VoxelGrid::Init computes it
 * (voxel_carver.cc:308-326) -- the same host arithmetic the device's axis tables are built with; needs no GPU. */
int vcy_axis_positions(const float bb_min[3], const float bb_max[3], float resolution, int axis, float* out);

/* ---- carving ------------------------------------------------------------ */

/* Replaces bool VoxelCarver::Carve(const Camera&, const Vector2i& roi_min,
 * const Vector2i& roi_max, const Image1f& sdf) (voxel_carver.cc:415-496).
 * `sdf_host` is row-major float[height*width]; it is copied before the call returns.  The view is
 * queued and applied later, in order (see "defer" under vcy_set_param); argument errors are reported
 * here.  A failure while applying queued views is returned by the call that applies them AND, if that
 * call was not a carve entry point (an extraction, a download), once more by the next vcy_carve* call,
 * so a `for each view: if (!Carve()) ...` loop sees it like the reference's bool Carve() would. */
int vcy_carve(vcy_ctx* ctx, const vcy_view* view, const float* sdf_host);
/* Same, SDF image already resident in HBM on the context's device (copied, ordered on the context's
 * stream: do not overwrite it before the stream has passed this call). */
int vcy_carve_device(vcy_ctx* ctx, const vcy_view* view, const float* sdf_device);
/* Replaces the loop of Carve(const std::vector<Camera>&, ...)
 * (voxel_carver.cc:516-528) for pre-built SDFs: fuses `n_views` views in
 * sequence order with the voxel state held in registers across views (up to 64
 * views per kernel launch, more are split).  The result is bit-identical to
 * n_views calls of vcy_carve_device.  The images are read while the launch runs:
 * keep them unchanged until the context's stream has passed this call. */
int vcy_carve_batch_device(vcy_ctx* ctx, int n_views, const vcy_view* views,
                           const float* const* sdf_device);
/* Replaces bool VoxelCarver::Carve(const Camera&, const Image1b& silhouette,
 * const Vector2i& roi_min, const Vector2i& roi_max, Image1f* sdf)
 * (voxel_carver.cc:394-413): MakeSignedDistanceField then the carve.
 * `sdf_out_host` (may be NULL) receives the SDF image like the Image1f* does. */
int vcy_carve_silhouette(vcy_ctx* ctx, const vcy_view* view,
                         const uint8_t* mask_host, float* sdf_out_host);

/* Replaces bool VoxelCarver::Carve(const std::vector<Camera>&, const std::vector<Image1b>&)
 * (voxel_carver.cc:516-528) end to end: every silhouette is uploaded (8 bit), turned into its SDF
 * on the device and fused, in chunks of 32 views; the upload + SDF build of chunk i+1 runs on a
 * second stream while chunk i is carved.  Result identical to n calls of vcy_carve_silhouette. */
int vcy_carve_batch_silhouettes(vcy_ctx* ctx, int n_views, const vcy_view* views,
                                const uint8_t* const* masks_host);

/* The same over the z-slabs of ONE grid held by this process (`slabs`: contexts of the same option set, any order, on
 * one or several devices -- vacancy::ShardedVoxelCarver::Carve): the devices SHARE the producer.  Device r of R uploads
 * and transforms the views r, r + R, ... of every chunk of 32; one ncclAllGather per chunk (librccl, the communicators
 * of vcy_halo_allgather) hands every device all the images of the chunk (width * height * 4 bytes each); every slab
 * carves the chunk from its device's copy while the next chunk is produced and gathered.  Slabs that share a device
 * share its images.  Same result as vcy_carve_batch_silhouettes on every slab -- where every GPU would build every
 * SDF (voxel_carver.cc:516-528 calls MakeSignedDistanceField once per view, :405-408).  One host thread per device
 * inside the call; vcy_last_stream_ms reports per slab.  VCY_ERR_UNSUPPORTED when several devices are involved and
 * librccl cannot be loaded (call vcy_carve_batch_silhouettes per slab then). */
int vcy_carve_batch_silhouettes_sharded(vcy_ctx* const* slabs, int n_slabs, int n_views, const vcy_view* views,
                                        const uint8_t* const* masks_host);

/* MakeSignedDistanceField (voxel_carver.cc:169-237, as Carve calls it at :405-408 with the context's options) for
 * n silhouettes in host memory into CALLER-owned device images (sdf_device_out[i]: width * height floats on the
 * context's device).  Returns when the images are complete.  The producer share of one rank of a one-process-per-GPU
 * job (vacancy_amd.dist.carve_silhouettes_sharded: build views r, r + G, ..., all-gather, carve). */
int vcy_make_sdf_batch_device(vcy_ctx* ctx, int n_views, const vcy_view* views, const uint8_t* const* masks_host,
                              float* const* sdf_device_out);

/* Of the last vcy_carve_batch_silhouettes: milliseconds its producer side took (per chunk of 32 views: the copy of the
 * silhouettes into page-locked staging, their DMA and the SDF build, events on the producer stream), its consumer
 * side (the fused carve launches, events on the context's stream), both summed over the chunks, and the call's wall
 * time.  max(produce, carve) / wall says how much of the shorter side was hidden behind the longer. */
int vcy_last_stream_ms(vcy_ctx* ctx, float* produce_ms, float* carve_ms, float* wall_ms);

/* Replaces void DistanceTransformL1(...) (voxel_carver.cc:102-167). */
int vcy_distance_transform_l1(const uint8_t* mask, int width, int height,
                              const int32_t roi_min[2], const int32_t roi_max[2],
                              float* dist_out);
/* Replaces void MakeSignedDistanceField(...) (voxel_carver.cc:169-237). */
int vcy_make_sdf(const uint8_t* mask, int width, int height,
                 const int32_t roi_min[2], const int32_t roi_max[2],
                 int minmax_normalize, int use_truncation, float truncation_band,
                 float* sdf_out);

/* MakeSignedDistanceField on the device: uploads the 8-bit mask, builds the SDF in HBM and returns
 * the device image (release with vcy_device_free); feed it to vcy_carve_device / _batch_device. */
int vcy_make_sdf_device(vcy_ctx* ctx, const uint8_t* mask_host, int width, int height,
                        const int32_t roi_min[2], const int32_t roi_max[2], int minmax_normalize,
                        int use_truncation, float truncation_band, float** sdf_device_out);

/* ---- surface extraction ------------------------------------------------- */

/* Replaces void VoxelCarver::ExtractIsoSurface(Mesh*, double iso_level,
 * bool linear_interp) (voxel_carver.cc:540-543) = MarchingCubes()
 * (marching_cubes.cc:63-228).  Vertex order and face order equal the
 * reference's serial scan (first reference in z,y,x order). */
int vcy_extract_iso(vcy_ctx* ctx, double iso_level, int linear_interp,
                    vcy_mesh* out);
/* Replaces void VoxelCarver::ExtractVoxel(Mesh*, bool inside_empty) (voxel_carver.cc:530-538 ->
 * extract_voxel.cc:258-317): one cube (24 vertices, 12 triangles) per kept voxel.  Runs on the
 * host on the downloaded state (the reference's drifting-cube arithmetic is serial by
 * construction); needs the whole grid in one context.  edge_keys is unused. */
int vcy_extract_voxel(vcy_ctx* ctx, int inside_empty, vcy_mesh* out);
/* The two halves of ExtractVoxel for a grid cut into z-slabs (vacancy::ShardedVoxelCarver::ExtractVoxel).
 * vcy_extract_voxel_ids: the parallel half on the device -- the keep predicate of every voxel of this context's slab
 * (extract_voxel.cc:283-286, or with inside_empty UpdateOnSurface :15-79; a slab above another one reads the slice below
 * its first from its halo: vcy_halo_allgather / _unpack / _install first) and the compaction -- returns the kept voxels'
 * GLOBAL ids in scan order (library-owned, vcy_ids_free; *ids_out = NULL when none is kept).
 * vcy_voxel_cubes: the serial half on the host, no GPU needed -- the reference translates ONE cube mesh to every kept
 * voxel and back (extract_voxel.cc:290-311), so every corner carries the rounding of all earlier kept voxels: the
 * slabs' lists, concatenated in z order, are walked with one drifting cube, and the mesh equals the single-context
 * vcy_extract_voxel array for array. */
int vcy_extract_voxel_ids(vcy_ctx* ctx, int inside_empty, int64_t** ids_out, int64_t* n_out);
void vcy_ids_free(int64_t* ids);
int vcy_voxel_cubes(const vcy_carver_option* option, int64_t n_ids, const int64_t* ids, vcy_mesh* out);
/* The same two calls writing into arrays of the CALLER: once the sizes are known -- 24 vertices and 12 triangles per kept
 * voxel -- `arrays` is called exactly once (not at all for an empty mesh; a non-zero return is passed on as
 * VCY_ERR_INTERNAL) and returns where 3 * n_vertices floats and 3 * n_faces int32 go; the host threads that fill them are
 * the first to touch them.  What a class API whose Mesh owns std::vectors wants (vacancy::VoxelCarver::ExtractVoxel: no
 * library-owned copy of an 800 MB mesh in between).  16-byte aligned arrays are written with streaming stores. */
typedef int (*vcy_mesh_arrays_fn)(void* user, int64_t n_vertices, int64_t n_faces, float** vertices, int32_t** faces);
int vcy_extract_voxel_into(vcy_ctx* ctx, int inside_empty, vcy_mesh_arrays_fn arrays, void* user);
int vcy_voxel_cubes_into(const vcy_carver_option* option, int64_t n_ids, const int64_t* ids, vcy_mesh_arrays_fn arrays,
                         void* user);

void vcy_mesh_free(vcy_mesh* mesh);
/* Milliseconds the device kernels of the last vcy_extract_iso took (hipEvents on the
 * context's stream: classify + owner + scan + emit; the mesh download is not included).
 * This is the region the reference's MarchingCubes timer brackets, minus the copy into Mesh. */
int vcy_last_extract_ms(const vcy_ctx* ctx, float* device_ms);
/* Milliseconds from the entry of the last vcy_extract_iso to its return, i.e. until the mesh arrays are
 * in host memory: the region the reference's MarchingCubes timer brackets (marching_cubes.cc:65-66,
 * 226-227).  The arrays of a vcy_mesh are page-locked host memory from a pool owned by the library. */
int vcy_last_extract_wall_ms(const vcy_ctx* ctx, float* wall_ms);

/* ---- state access (tests, ExtractVoxel on the host, checkpoint) ---------- */

/* Copies the slab's voxel state to the host: sdf[nx*ny*nz_local] and
 * update_num[nx*ny*nz_local] in the reference's linear order
 * id = z*nx*ny + y*nx + x (voxel_carver.cc:333,349-355).  Either may be NULL. */
int vcy_download(vcy_ctx* ctx, float* sdf, int32_t* update_num);
int vcy_upload(vcy_ctx* ctx, const float* sdf, const int32_t* update_num);
/* Device-side comparison of two contexts that own the same slab of the same grid (on one device):
 * *n_diff = number of voxels whose (sdf bits, update_num) differ.  Nothing is downloaded -- this is
 * how the tests cross-check two kernel paths over a whole 1024^3 / 2048^3 grid. */
int vcy_state_equal(vcy_ctx* a, vcy_ctx* b, int64_t* n_diff);
/* Point query: state of `n` voxels given by global id (must lie in this context's slab). */
int vcy_download_voxels(vcy_ctx* ctx, int64_t n, const int64_t* voxel_ids, float* sdf, int32_t* update_num);
/* Voxel centres of the slab, 3 floats per voxel (Voxel::pos, voxel_carver.cc:315-337). */
int vcy_download_positions(vcy_ctx* ctx, float* pos);

/* ---- multi-GPU halo (one process per GPU; the exchange itself is one RCCL
 * all-gather issued by the caller on the buffers below) ------------------- */

/* Bytes one rank contributes: the LAST two xy-slices of its slab, (sdf, update_num).
 * Rank r consumes rank r-1's contribution as its slices z_begin-2, z_begin-1. */
int64_t vcy_halo_bytes(const vcy_ctx* ctx);
/* Packs this slab's boundary slices into `send_device` (vcy_halo_bytes bytes). */
int vcy_halo_pack(vcy_ctx* ctx, void* send_device);
/* Installs the neighbours' slices from the all-gathered buffer
 * (world * vcy_halo_bytes bytes, rank-major). */
int vcy_halo_unpack(vcy_ctx* ctx, const void* gathered_device, int rank, int world);
/* Same, given directly the pack (vcy_halo_bytes bytes, device) of the slab that ends at this
 * context's z_begin -- for layouts where slabs are not ordered by rank (several slabs per GPU). */
int vcy_halo_install(vcy_ctx* ctx, const void* prev_slab_pack_device);
/* Single-process multi-GPU: copies the last two slices of `below` (the slab that ends at ctx's
 * z_begin, on any device of this process) straight into ctx's halo (peer-to-peer over xGMI). */
int vcy_halo_copy_from(vcy_ctx* ctx, vcy_ctx* below);
/* Single-process multi-GPU, the north star's "single RCCL all-gather of boundary slabs before mesh
 * extraction" (the exchange step of MarchingCubes(), marching_cubes.cc:93-101 reads z-1): `slabs` are
 * ALL z-slabs of one grid in z order (slab i ends where slab i+1 begins), on any devices of this
 * process.  Every slab's pack goes into its device's send buffer, ONE ncclAllGather (one communicator
 * rank per distinct device, ncclCommInitAll; communicators and staging are cached per device list)
 * hands every device every pack, and each slab installs the pack of the slab below it.  librccl.so is
 * opened on first use; VCY_ERR_UNSUPPORTED if it cannot be loaded -- THIS function never falls back to
 * peer copies.  vcy_halo_copy_from is the explicit alternative, and the host layers above take it on exactly
 * that status only where they say so: Python's vacancy_amd.carver.halo_exchange (used by ShardedVoxelCarver and
 * dist.exchange_halo) then copies slab by slab and returns "backend": "peer copies ..."; the C++
 * ShardedVoxelCarver never does it on its own (set_halo_transport(kPeerCopy) asks for it). */
int vcy_halo_allgather(vcy_ctx* const* slabs, int n_slabs);
/* Releases what vcy_halo_allgather keeps between calls (communicators, streams and staging buffers per
 * device set); the next exchange builds them again.  Call when no exchange is in flight. */
void vcy_halo_shutdown(void);
/* One process per GPU WITHOUT torch -- the same single all-gather for a C++ host that runs one process per device
 * (the north star's host model; vacancy_amd/dist.py is its torch.distributed twin; no reference counterpart: the
 * reference is one OpenMP process).  vcy_comm_create: rank `rank` of `world` on `device_id`; rank 0 draws an
 * ncclUniqueId and every rank receives it through `rendezvous` -- "file:<path>" (a path private to the job on a
 * filesystem all ranks of the node see: rank 0 publishes the id by an atomic rename, the others poll for it) or
 * "tcp:<host>:<port>" (rank 0 listens, the others connect, retrying until it does) -- then ncclCommInitRank.  Blocks
 * until all ranks have joined or `timeout_ms` has passed (<= 0: 120 s).  world == 1 needs no rendezvous (NULL is fine).
 * vcy_halo_allgather_ranks: `my_slabs` = this rank's slab contexts in slab order -- the grid's slab ids rank,
 * rank + world, ... (vacancy_amd.dist.slabs_of_rank), every rank holding the SAME number; each packs its last two
 * slices, ONE ncclAllGather (rank-major), every slab installs the pack of the slab below it.
 * vcy_rendezvous_exchange is the rendezvous by itself (128 bytes from rank 0 to every rank): what the CPU tests run. */
typedef struct vcy_comm vcy_comm;
int vcy_comm_create(int rank, int world, int device_id, const char* rendezvous, int timeout_ms, vcy_comm** out);
void vcy_comm_destroy(vcy_comm* comm);
int vcy_halo_allgather_ranks(vcy_comm* comm, vcy_ctx* const* my_slabs, int n_my_slabs);
int vcy_rendezvous_exchange(int rank, int world, const char* rendezvous, void* payload128, int timeout_ms);
/* What the last vcy_halo_allgather of this process did, as text:
 * "backend=rccl op=ncclAllGather version=V ranks=R bytes_per_rank=B calls=N lib=..." or "none". */
const char* vcy_last_collective(void);

/* ---- device memory / stream / timing helpers ----------------------------- */

int vcy_device_count(int* count);
int vcy_sdf_upload(vcy_ctx* ctx, const float* sdf_host, int width, int height,
                   float** sdf_device_out);
int vcy_device_free(vcy_ctx* ctx, void* device_ptr);
int vcy_device_alloc(vcy_ctx* ctx, int64_t bytes, void** device_ptr_out);
int vcy_memcpy_h2d(vcy_ctx* ctx, void* dst_device, const void* src_host, int64_t bytes);
int vcy_memcpy_d2h(vcy_ctx* ctx, void* dst_host, const void* src_device, int64_t bytes);
/* Back to the state right after vcy_create (VoxelGrid::Init: sdf = lowest(), update_num = 0);
 * asynchronous on the context's stream. */
int vcy_reset(vcy_ctx* ctx);
/* Tuning knobs that never change results.  "fused" (default 1): 0 forces one kernel launch
 * per view (the generic kernel) instead of the fused multi-view kernel.  "cull" (default 1): 0 never
 * drops provably idle (brick, view) pairs.  "tile" (default 0 = chosen from the pixel footprint of a
 * voxel): 1 / 2 force the 16 x 16 pixel tile / the 2048-pixel tile of the fused kernel.  "defer" (default 1): the per-view
 * entry points vcy_carve / vcy_carve_device / vcy_carve_silhouette keep a private device copy of the image
 * and queue the view; queued views are carved together, in call order, by one fused launch when the
 * state is next needed (extraction, download, upload, halo, vcy_sync, vcy_timer_end, a batch call) or
 * when 32 wait, so the reference's `for each view: Carve()` loop costs one pass over the grid instead of
 * one per view.  0 carves every view before its call returns.  vcy_reset drops queued views.
 * "shortdiv" (default 1): fx / z in the fused kernel may use a 4- or 6-instruction sequence instead of
 * the full IEEE expansion, but only after the library has checked on the device, for each focal length
 * in use, that the sequence equals IEEE division for EVERY admissible depth (all 2^23 significands in 121
 * binades, about a millisecond once per focal length per process).
 * "mcsweep" (default 0): 1 makes vcy_extract_iso find the surface cells in one sweep over the state with the bit
 * planes in LDS when a voxel row is a power-of-two number of 64-voxel words (nx = 64 ... 2048): 2 % fewer bytes
 * moved than with the bit planes in memory (the default and the path of every other shape), 2 - 30 % more time.
 * "meshkeys" (default 1): vcy_extract_iso returns vcy_mesh.edge_keys; 0 leaves it NULL (nothing is computed for
 * it or copied: a third of the mesh bytes) -- for callers that do not merge z-slabs, i.e. what the reference's
 * MarchingCubes returns.
 * "mcskip" (default 1): vcy_extract_iso does not read bricks whose minimum -- kept per 8 x 8 x 8 brick by the fused
 * carve kernel -- lies above the iso level (they are outside the surface whatever they hold exactly); 0 reads every
 * brick; 1 skips them where that is the faster pass (voxel rows of 1024 and more: at 512^3 the dense pass wins by 7 - 14 %);
 * 2 skips them on any size (what the parity tests ask for).
 * "mcdirect" (default 33554432): vcy_extract_iso lets its last kernel write a mesh whose GUESSED size (the previous
 * extraction's counts + 25 %) is at most this many bytes straight into the page-locked host arrays it returns -- one
 * enqueue, one wait per call; larger meshes are staged in device memory and copied with their exact sizes.  0: always
 * staged.  Results identical.  "mctiming" 1 (or VCY_MC_TIMING=1 in the environment): the host-side phases of every
 * extraction on stderr.
 * "livelist" (default 1): a carve launch of up to 8 views over an already carved grid first lists the workgroups in
 * which some view can still change a voxel (bounds of the views' samples against the kept brick minima / the
 * truncation limit) and starts only those; 0 starts every workgroup and lets each decide for itself.
 * "livesync" (default 1): with the live list, the host waits for the list's length and launches the carve kernel over the
 * listed workgroups only (a single-view launch at 1024^3: 0.115 ms less than starting all 512 K workgroups to have the
 * others leave); 0 starts every workgroup and never waits.
 * "coopstore" (default -1): how a fused launch writes the state back.  1: the four waves of a workgroup exchange their
 * bricks through LDS and store whole 128-byte row segments; 0: every wave stores its own 16-byte pieces; -1: the first
 * for single-view launches (over a carved grid in the weighted-average modes: up to 8 views), the second otherwise.
 * Results are identical.
 * "rowkernel" (default 0): n > 0 or -1 (= 8) sends fused launches of up to n views through the few-view flavour of the
 * carve kernel -- a WAVE walks the four bricks of a row segment, their state requested up front with LDS-direct loads,
 * whole 128-byte row segments stored, no barrier -- instead of a workgroup of four waves with the cooperative write-back.
 * Results identical; measured slower (3.17 against 2.63 ms per weighted-average view at 1024^3: its staging area leaves
 * 2.7 waves per SIMD), so it is off by default and serves as the second implementation the tests compare with.
 * "oneview" (default 1): a fused launch of exactly ONE view -- what every call of the reference's per-view API becomes
 * when an extraction separates the views, examples.cc:117-149 -- takes a kernel instance compiled for one view (the
 * footprint record unpacked into registers, no view loop, no second tile buffer): 15 % fewer vector and 27 % fewer
 * scalar instructions per wave, 2.72 -> 2.43-2.53 ms per weighted-average view at 1024^3, the first view on a fresh grid
 * 1.73-1.92 -> 1.37-1.58, kMax 0.60 -> 0.53; 0: the general instance.  Results identical.
 * "listrecords" (default 1): the live list of a launch of ONE view holds, per listed workgroup, its id, which of its
 * waves are live and their four footprint records (40 bytes instead of 4): a wave learns its record with its id instead
 * of one memory round trip later, and needs no brick minimum for a test the list pass has made (kMax at 1024^3:
 * 0.506 -> 0.496 ms per view).  0: ids only.  Results identical.
 * "eagerstate" (default -1): launches of ONE view ("oneview" 1) over a carved grid request a brick's state next to its footprint
 * record, before the test that lets a wave leave without it, when nearly every started workgroup will need it: listed
 * launches (only live workgroups are started) and launches that skipped their list because the last one held most
 * workgroups; 0 never, 1 every such launch.  One memory round trip less per wave: 2.51 -> 2.35-2.46 ms per
 * weighted-average view at 1024^3, kMax 0.52 -> 0.49 (profiles/r06/eager_state.txt).  Results identical.
 * "ntstore" (default -1 = 1): the cooperative write-back stores its whole row segments as streaming stores (0: ordinary).
 * "recordbytes" (default 0 = 2 GiB): bytes of footprint records one carve launch may take; a launch whose records would
 * be larger is cut into chunks of whole brick layers (2048^3 x 64 views: 8.6 GB of records, four chunks) -- small values
 * let tests run the chunking on small grids.
 * "prologue" (default 0): where a raw-tile launch gets its footprints: 0 or 2 = from the pre-pass's records; 1 = computed
 * in the carve kernel's prologue, one lane per view (one launch whatever the size, and slower: 92.8 against 81.8 ms per
 * step at 2048^3 x 64 views).  Results identical.
 * "carvetimer" (default 0): 1 records HIP events around what runs before the carve kernel (window maxima, pre-pass)
 * and around the carve kernel of every fused launch (vcy_last_carve_ms, vcy_carve_log); setting it clears the log.
 * "lazycount" (default 1): update_num is stored in one byte until more than 255 views have been applied since the fill
 * (a voxel's count cannot exceed that number), then widened in one pass to what voxel_max_update_num needs; 0 allocates
 * the final width at once.  Results are identical; "count_bytes" / "count_bytes_final" read the widths back.
 * "paircount" (default 0): 1 makes every fused launch count the (brick, view) pairs it processes (vcy_last_carve_pairs).
 * "inject_carve_failure" (test hook, default 0): the next `value` applications of views fail with
 * VCY_ERR_INTERNAL before anything is launched -- how the tests exercise the error contract of vcy_carve. */
int vcy_set_param(vcy_ctx* ctx, const char* name, int value);
/* Reads a knob back ("fused", "cull", "tile", "defer", "shortdiv", "mcsweep", "mcskip", "mcdirect", "rowkernel", "oneview", "eagerstate", "listrecords", "ntstore", "livelist", "livesync", "coopstore", "meshkeys",
 * "lazycount", "carvetimer"), "count_bytes" / "count_bytes_final" / "carvelog_dropped" (see above), "div_level": the
 * division sequence the last fused launch was instantiated with (2: 4 instructions, 1: 6, 0: full IEEE expansion), or
 * "brick_min_valid": 1 while the brick minima describe the state (every write since the fill went through the fused kernel). */
int vcy_get_param(vcy_ctx* ctx, const char* name, int* value);
/* Use an existing hipStream_t (e.g. torch's current stream) for all launches. */
int vcy_set_stream(vcy_ctx* ctx, void* hip_stream);
int vcy_get_stream(vcy_ctx* ctx, void** hip_stream_out);
int vcy_sync(vcy_ctx* ctx);
/* hipEvent pair recorded on the context's stream around whatever is launched
 * between begin and end; vcy_timer_end synchronises and returns milliseconds. */
int vcy_timer_begin(vcy_ctx* ctx);
int vcy_timer_end(vcy_ctx* ctx, float* elapsed_ms);

/* With vcy_set_param(ctx, "carvetimer", 1): milliseconds (HIP events on the context's stream) of the last fused
 * carve launch, split into what runs before the carve kernel (footprint pre-pass, live-workgroup list) and the
 * carve kernel itself -- the kernel the roofline is quoted for.  Synchronises with that launch. */
int vcy_last_carve_ms(vcy_ctx* ctx, float* prepass_ms, float* kernel_ms);
/* The whole log "carvetimer" 1 keeps since it was set (or since the last call with clear != 0): one record per chunk
 * of every fused launch (first_chunk[i] != 0 starts a launch; a launch has one chunk unless its footprint records
 * exceed "recordbytes"), up to 8192 records -- later launches are not recorded, and counted:
 * vcy_get_param "carvelog_dropped" (a reader that divides by a step count must check it; vcy_last_carve_ms fails
 * rather than report an older launch).  begin_ms[i] = start of record i
 * (before its window maxima and pre-pass) since the start of record 0; prepass_ms[i] / kernel_ms[i] as for
 * vcy_last_carve_ms.  Any of the arrays may be NULL.  Nothing synchronises while the launches are issued: a sequence
 * of carve steps can be queued back to back and read here afterwards (this call waits for the recorded launches).
 * No reference counterpart (the reference's Timer brackets the loop, voxel_carver.cc:435,492-493). */
int vcy_carve_log(vcy_ctx* ctx, int max_records, float* begin_ms, float* prepass_ms, float* kernel_ms,
                  int32_t* first_chunk, int* n_records, int clear);

/* With vcy_set_param(ctx, "paircount", 1): how many (8 x 8 x 8 brick, view) pairs the last fused launch really
 * processed -- i.e. did not drop as provably idle -- in total and per brick layer of the slab (per_layer may be NULL;
 * *n_layers = layers of the slab), and how many pairs there are.  Synchronises with that launch.  What the scene
 * leaves of the work is visible here (bench.py: "pairs_processed_frac"), and the slab planner's estimate is checked
 * against it. */
int vcy_last_carve_pairs(vcy_ctx* ctx, int64_t* processed, int64_t* total, int64_t* per_layer, int max_layers,
                         int* n_layers);

/* Multi-GPU: where to cut the grid into `n_slabs` z-slabs so that a fused carve of THESE views costs every slab the
 * same (no reference counterpart: the reference's OpenMP loop balances itself with schedule(dynamic, 1),
 * voxel_carver.cc:439-441; z-slabs on different GPUs cannot).  With view dropping the brick layers through the object
 * cost about 1.6x the outer ones, so slabs of equal thickness leave the slowest of 8 GPUs 1.2x over the mean.  `ctx` is
 * any context of the grid on the device that holds the images (typically a small planning context, vcy_create(option,
 * device, 0, 8, ...), destroyed afterwards: only its axis tables and staging buffers are used, its slab is not touched);
 * the estimate -- per brick layer of the WHOLE grid: brick_cost x bricks + (brick, view) pairs a carve will process,
 * found by playing the kernel's drop decisions on upper and lower bounds of every `sample_stride`-th brick's samples
 * (0: every 2nd in x and y) -- is returned in layer_cost[0 .. *n_layers) if not NULL.  brick_cost <= 0: the library's
 * calibrated value.  z_bounds[0 .. n_slabs] receives the cuts (multiples of 8 slices, z_bounds[0] = 0,
 * z_bounds[n_slabs] = nz): the contiguous partition with the smallest largest part.  The same inputs give the same
 * cuts on every rank. */
int vcy_plan_z_slabs(vcy_ctx* ctx, int n_views, const vcy_view* views, const float* const* sdf_device, int n_slabs,
                     int sample_stride, float brick_cost, int32_t* z_bounds, double* layer_cost, int max_layers,
                     int* n_layers);

/* The cut itself, host arithmetic only (no GPU): the contiguous partition of `n_layers` brick layers with the given costs
 * into `n_slabs` parts -- every part at least one layer -- that minimises the largest part and, among those, the sum of
 * squares.  z_bounds[0 .. n_slabs] in slices (multiples of 8, the last one nz; nz in ((n_layers - 1) * 8, n_layers * 8]).
 * What vcy_plan_z_slabs applies to its estimate; callable with measured costs as well. */
int vcy_partition_layers(const double* layer_cost, int n_layers, int n_slabs, int nz, int32_t* z_bounds);

/* Device-side self test of the identities the fast paths rest on (no reference counterpart): the
 * two-instruction reciprocal used for update_num + 1 in the unit-weight weighted average equals the
 * IEEE quotient for every count a u8 / u16 counter can hold.  VCY_OK, or VCY_ERR_INTERNAL with the
 * mismatch in vcy_last_error(). */
int vcy_selftest(vcy_ctx* ctx);

/* Measured device-memory bandwidth of this GPU, for the roofline next to the vendor peak
 * ("print a measured device-memcpy/triad GB/s on the box"): a streaming read of
 * `bytes` (dword loads, eight in flight per lane, the access shape of the grid sweeps here) and a
 * device-to-device copy of `bytes` (counted as read + write).  Best of `reps`; GB/s = 1e9 B/s. */
int vcy_measure_bandwidth(int device_id, uint64_t bytes, int reps, double* read_gbs, double* copy_gbs);

/* Measurement aid, not on the carve path: the shader clock the device runs at WHILE other work is on it.  _start puts one
 * wave on a stream of its own that samples the shader clock counter against the constant 100 MHz reference every ~30 us
 * (at most max_samples samples, then it ends by itself); _stop ends it, frees everything and returns the clock over the
 * whole span (mean_hz) and over its second half (settled_hz: the device idles for a millisecond while the probe is set
 * up, and its clock takes some milliseconds of load to come back), the slowest / fastest interval, the number of samples
 * and the time they cover.  bench.py runs it
 * during six more steps of its workload queued right behind the timed region (a second active hardware queue costs the
 * carve kernel 1.7 %, so not inside it): the VALU issue fraction of the carve kernel is then priced at the clock of the
 * run that is reported, not at the clock of the box the counters were collected on. */
typedef struct vcy_clock_probe vcy_clock_probe;
int vcy_clock_probe_start(int device_id, int max_samples, vcy_clock_probe** out);
int vcy_clock_probe_stop(vcy_clock_probe* probe, double* mean_hz, double* settled_hz, double* min_hz, double* max_hz,
                         int* n_samples, double* covered_ms);

const char* vcy_last_error(void);
/* "vacancy_amd <version> (gfx950) src:<hash>": the hash covers every source file the library was built from
 * (profiles/counters.json entries are stamped with this string; bench.py drops counters of another build). */
const char* vcy_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VACANCY_HIP_H_ */
