// ShardedVoxelCarver: the VoxelCarver API over several GPUs of one node from ONE process.
//
// The grid is cut into z-slabs (k per device, dealt cyclically; one per device by default).  Where it is cut
// matters: with view dropping the slabs through the object cost 1.6x the outer ones, so PlanPartition() places
// the cuts where a carve of the given views is predicted to cost every slab the same (vcy_plan_z_slabs);
// without it the slabs are of equal thickness.  Carving needs no exchange (a batch of views shares ONE producer of SDF
// images: device r builds views r, r + R, ..., one RCCL all-gather per chunk of 32 hands every device all of them,
// vcy_carve_batch_silhouettes_sharded), extraction needs the two slices below
// each slab -- ONE RCCL all-gather of every slab's boundary slices (vcy_halo_allgather: one
// communicator rank per device, ncclCommInitAll) -- and the per-slab meshes are stitched by edge key
// into exactly the mesh a single VoxelCarver returns.  (The one-process-per-GPU form of the same
// scheme is vacancy_amd/dist.py + bench.py, where the all-gather goes through torch.distributed.)
#pragma once

#include <memory>
#include <vector>

#include "vacancy/voxel_carver.h"

namespace vacancy {

class ShardedVoxelCarver {
 public:
  ShardedVoxelCarver(VoxelCarverOption option, std::vector<int> device_ids, int slabs_per_device = 1);
  ~ShardedVoxelCarver();
  ShardedVoxelCarver(const ShardedVoxelCarver&) = delete;
  ShardedVoxelCarver& operator=(const ShardedVoxelCarver&) = delete;

  // Before Init(): cuts of equal predicted carve cost for these views (a small planning context on the first
  // device builds their SDF images and plays the kernel's drop decisions per brick layer; about a millisecond
  // at 1024^3 x 32 views).  The views carved later need not be these -- a camera rig that stays put is the case it
  // is made for.  false (and equal thickness at Init) if the plan cannot be made.
  bool PlanPartition(const std::vector<const Camera*>& cameras, const std::vector<Image1b>& silhouettes);
  // ... or cuts from elsewhere: slab_count + 1 increasing z values from 0 to nz, every slab >= 2 slices.
  void set_z_bounds(const std::vector<int>& z_bounds);
  const std::vector<int>& z_bounds() const;  // of the slabs Init() created
  bool Init();
  int slab_count() const;
  // how the halo slices travel before extraction: the RCCL all-gather (default) or explicit
  // peer-to-peer copies slab by slab (vcy_halo_copy_from)
  enum class HaloTransport { kRcclAllGather, kPeerCopy };
  void set_halo_transport(HaloTransport t);
  // one view / a batch of views into every slab (slabs of different devices run concurrently)
  bool Carve(const Camera& camera, const Image1b& silhouette);
  bool Carve(const std::vector<const Camera*>& cameras, const std::vector<Image1b>& silhouettes);
  void ExtractIsoSurface(Mesh* mesh, double iso_level = 0.0, bool linear_interp = true);
  // VoxelCarver::ExtractVoxel over the slabs: the keep predicate and the compaction run on every slab's device, the
  // kept voxels of all slabs are walked in z order by ONE drifting cube on the host -- the reference's arithmetic
  // (extract_voxel.cc:290-311), so the mesh equals a single VoxelCarver's array for array.
  void ExtractVoxel(Mesh* mesh, bool inside_empty = false);

 private:
  bool ExchangeHalo();
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace vacancy
