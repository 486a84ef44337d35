// vk_score_m5.hip -- vk_score_kernel, MODE 5 (see vk_score.hip.h): any d >= 256, bf16 rows, eight K-steps in flight
#include "vk_score.hip.h"

extern "C" hipError_t vk_launch_score_m5(const VkScoreParams *p, int32_t grid, size_t smem_bytes, hipStream_t stream) {
	return launch_score_gap<5, 0, false>(*p, grid, smem_bytes, stream);
}
