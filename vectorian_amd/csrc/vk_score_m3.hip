// vk_score_m3.hip -- vk_score_kernel, MODE 3 (see vk_score.hip.h)
#include "vk_score.hip.h"

extern "C" hipError_t vk_launch_score_m3(const VkScoreParams *p, int32_t grid, size_t smem_bytes, hipStream_t stream) {
	return launch_score_gap<3, 24, false>(*p, grid, smem_bytes, stream);
}

// d = 300 with the query tile in LDS instead of registers (experiment: VK_QLDS=1)
extern "C" hipError_t vk_launch_score_m3_300(const VkScoreParams *p, int32_t grid, size_t smem_bytes, hipStream_t stream) {
	return launch_score_gap<3, 10, true>(*p, grid, smem_bytes, stream);
}
