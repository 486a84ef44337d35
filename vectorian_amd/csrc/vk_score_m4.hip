// vk_score_m4.hip -- vk_score_kernel, MODE 4 (see vk_score.hip.h): fp32 unit rows at d = 300 (19 blocks of 16 features)
#include "vk_score.hip.h"

extern "C" hipError_t vk_launch_score_m4(const VkScoreParams *p, int32_t grid, size_t smem_bytes, hipStream_t stream) {
	return launch_score_gap<4, 19, false>(*p, grid, smem_bytes, stream);
}
