// vk_score_m6.hip -- vk_score_kernel, MODE 6 (see vk_score.hip.h): fp32 rows of any d (runtime K loop)
#include "vk_score.hip.h"

extern "C" hipError_t vk_launch_score_m6(const VkScoreParams *p, int32_t grid, size_t smem_bytes, hipStream_t stream) {
	return launch_score_gap<6, 0, false>(*p, grid, smem_bytes, stream);
}
