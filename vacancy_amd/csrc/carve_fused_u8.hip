// One half of the instances of carve_fused_kernel: update_num in ONE byte (what every BASELINE configuration runs: the
// counters are one byte until more than 255 views have been applied since the fill).  The kernel, its helpers and the
// launch dispatch are carve_fused.hip's, included up to launch_fused_1; this unit exports launch_fused_counts8.
// (Two units so that the instantiations compile in parallel halves; reference: voxel_carver.cc:415-496, see carve_fused.hip.)
#define VCY_FUSED_PART 8
#define VCY_FUSED_PART_FN launch_fused_counts8
#define VCY_FUSED_PART_TYPE uint8_t
#include "carve_fused.hip"
