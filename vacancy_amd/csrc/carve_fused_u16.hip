// The other half of the instances of carve_fused_kernel: update_num in TWO bytes (voxel_max_update_num > 254 once more
// than 255 views have been applied).  See carve_fused_u8.hip; this unit exports launch_fused_counts16.
#define VCY_FUSED_PART 16
#define VCY_FUSED_PART_FN launch_fused_counts16
#define VCY_FUSED_PART_TYPE uint16_t
#include "carve_fused.hip"
