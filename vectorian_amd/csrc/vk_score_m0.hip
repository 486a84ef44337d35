// vk_score_m0.hip -- vk_score_kernel, MODE 0 (see vk_score.hip.h)
#include "vk_score.hip.h"

extern "C" hipError_t vk_launch_score_m0(const VkScoreParams *p, int32_t grid, size_t smem_bytes, hipStream_t stream) {
	return launch_score_gap<0, 10, true>(*p, grid, smem_bytes, stream);
}
