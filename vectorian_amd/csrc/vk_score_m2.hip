// vk_score_m2.hip -- vk_score_kernel, MODE 2 (see vk_score.hip.h)
#include "vk_score.hip.h"

extern "C" hipError_t vk_launch_score_m2(const VkScoreParams *p, int32_t grid, size_t smem_bytes, hipStream_t stream) {
	return launch_score_gap<2, 0, false>(*p, grid, smem_bytes, stream);
}
