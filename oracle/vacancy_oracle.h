/*
 * vacancy_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library.  The shipped path
 * (libvacancy_hip.so) never links, loads or calls it.
 *
 * The oracle is a CPU restatement of the reference's algorithm for the hot path
 * (VoxelCarver::Carve + MarchingCubes + the SDF builder + the pose arithmetic of
 * the bunny harness); every function cites the reference lines it follows.  See
 * vacancy_oracle.cc for how it is pinned.
 */
#ifndef VACANCY_ORACLE_H_
#define VACANCY_ORACLE_H_

#include <stdint.h>

#include "vacancy_hip.h" /* POD option/view structs only */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_grid orc_grid;

/* VoxelCarver::Init + VoxelGrid::Init.  NULL when the reference returns false. */
orc_grid* orc_grid_create(const vcy_carver_option* option);
/* Test helpers for grids too large for the CPU: carve an arbitrary SAMPLE of voxel centres with the
 * same loop, and the per-axis voxel centres of VoxelGrid::Init. */
orc_grid* orc_grid_from_positions(const vcy_update_option* uo, const float* pos, int n);
void orc_axis_positions(float bb_min, float bb_max, float resolution, float* out, int* n_out);
void orc_grid_destroy(orc_grid* g);
void orc_grid_dims(const orc_grid* g, int32_t dims[3]);
int orc_sizeof_voxel(void);
void orc_grid_download(const orc_grid* g, float* sdf, int32_t* update_num);
void orc_grid_upload(orc_grid* g, const float* sdf, const int32_t* update_num);
void orc_grid_positions(const orc_grid* g, float* pos);

/* VoxelCarver::Carve(camera, roi_min, roi_max, sdf), voxel_carver.cc:415-496.
 * Returns the milliseconds of the main loop (what the reference's Timer brackets). */
double orc_carve(orc_grid* g, const vcy_view* view, const float* sdf);

/* DistanceTransformL1 / MakeSignedDistanceField, voxel_carver.cc:102-237. */
void orc_distance_transform_l1(const uint8_t* mask, int width, int height,
                               const int32_t roi_min[2], const int32_t roi_max[2],
                               float* dist);
void orc_make_sdf(const uint8_t* mask, int width, int height,
                  const int32_t roi_min[2], const int32_t roi_max[2],
                  int minmax_normalize, int use_truncation, float truncation_band,
                  float* sdf);

/* MarchingCubes(), marching_cubes.cc:63-228.  Fills a vcy_mesh (malloc-ed, release
 * with orc_mesh_free); edge_keys are the std::map keys of the reference.  Returns the
 * milliseconds of the whole function. */
double orc_marching_cubes(const orc_grid* g, double iso_level, int linear_interp,
                          vcy_mesh* out);
/* The same loop restricted to the cell layers of a z-slab, with the layer below walked first as
 * a ghost layer (what one rank of the multi-GPU path computes); see the .cc. */
double orc_marching_cubes_slab(const orc_grid* g, double iso_level, int linear_interp,
                               int z_begin, int z_end, vcy_mesh* out);
/* ExtractVoxel(), extract_voxel.cc:258-317 (cube per kept voxel; edge_keys unused). */
double orc_extract_voxel(orc_grid* g, int inside_empty, vcy_mesh* out);
void orc_mesh_free(vcy_mesh* m);

/* Pose arithmetic of the harness (Eigen operations restated, see .cc). */
void orc_pose_from_tum(const double t[3], const double q_xyzw[4], double c2w[12]);
void orc_affine_inverse(const double c2w[12], double w2c[12]);
void orc_lookat_c2w(const double position[3], const double target[3],
                    const double up[3], double c2w[12]);
void orc_affine_to_float(const double m[12], float out[12]);
/* PinholeCamera::set_fov_y, camera.cc:114-120 */
float orc_focal_from_fov_y(int height, float fov_y_deg);

/* TEST-ONLY: association of the 3-term row sums of Affine3f * Vector3f in orc_carve (0 = the restated Eigen order,
 * the default; 1, 2 = the other two).  Used by tests/test_association_exposure.py to put a number on the one
 * assumption nothing in the reference pins; see the .cc. */
void orc_set_association(int mode);

int orc_omp_max_threads(void);
/* kEdgeTable[256] / kTriTable[256][16] as the oracle uses them (marching_cubes_lut.cc:15-298). */
void orc_mc_tables(int* edge256, int* tri256x16);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
