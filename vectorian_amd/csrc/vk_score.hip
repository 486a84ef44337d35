// vk_score.hip -- dispatch of the scoring kernel over its MODE translation units, and the span kernel.
#include "vk_common.hip.h"

extern "C" hipError_t vk_launch_score_m0(const VkScoreParams *p, int32_t grid, size_t smem_bytes, hipStream_t stream);
extern "C" hipError_t vk_launch_score_m1(const VkScoreParams *p, int32_t grid, size_t smem_bytes, hipStream_t stream);
extern "C" hipError_t vk_launch_score_m5(const VkScoreParams *p, int32_t grid, size_t smem_bytes, hipStream_t stream);
extern "C" hipError_t vk_launch_score_m6(const VkScoreParams *p, int32_t grid, size_t smem_bytes, hipStream_t stream);
extern "C" hipError_t vk_launch_score_m2(const VkScoreParams *p, int32_t grid, size_t smem_bytes, hipStream_t stream);
extern "C" hipError_t vk_launch_score_m3(const VkScoreParams *p, int32_t grid, size_t smem_bytes, hipStream_t stream);
extern "C" hipError_t vk_launch_score_m3_300(const VkScoreParams *p, int32_t grid, size_t smem_bytes, hipStream_t stream);
extern "C" hipError_t vk_launch_score_m4(const VkScoreParams *p, int32_t grid, size_t smem_bytes, hipStream_t stream);

extern "C" hipError_t vk_launch_score(const VkScoreParams *pp, int32_t grid, size_t smem_bytes, hipStream_t stream) {
	const VkScoreParams &p = *pp;
	if (p.layout == VK_DEV_LAYOUT_STATIC) return vk_launch_score_m2(pp, grid, smem_bytes, stream);
	if (p.prec == 0 && p.nk32 == 10 && p.tail == 1) return p.q_mode3 ? vk_launch_score_m3_300(pp, grid, smem_bytes, stream) : vk_launch_score_m0(pp, grid, smem_bytes, stream);
	if (p.prec == 0 && p.nk32 == 24 && p.tail == 0) return vk_launch_score_m3(pp, grid, smem_bytes, stream);
	if (p.prec == 1 && p.nk32 == 19 && p.q_mode3) return vk_launch_score_m4(pp, grid, smem_bytes, stream);   // fp32 rows, 300-d
	if (p.prec == 1) return vk_launch_score_m6(pp, grid, smem_bytes, stream);
	if (p.prec == 0 && p.nk32 >= 8) return vk_launch_score_m5(pp, grid, smem_bytes, stream);   // wide rows: deeper load pipeline
	return vk_launch_score_m1(pp, grid, smem_bytes, stream);
}

// ---------------------------------------------------------------------------
// One-token slices against a one-token query (the span-embedding index: one vector per sentence / window,
// PartitionEmbeddingSim, vectorian/index.py:679-810): local alignment of a 1 x 1 matrix is the clipped cosine
// itself, so the kernel is the similarity tile alone: one MFMA tile = 16 slices, scores written 64 bytes at a
// time.  HBM-bound: d * 2 bytes per slice.
// ---------------------------------------------------------------------------

// A wave takes runs of VK_SPAN_RUN consecutive tiles (one tile per wave in turn left 300-d spans at 5.9 TB/s; runs: 6.4 - 6.6) and
// keeps all of a tile's loads in front of its MFMAs (sched_barrier in sim_tile).
constexpr int VK_SPAN_RUN = 4;

template <int NK32, bool TAIL>
__global__ __launch_bounds__(256) void vk_span_kernel(VkScoreParams p) {
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	QFrag<NK32, TAIL> qf;
	if constexpr (NK32 > 0) load_qfrag<NK32, TAIL>(qf, p.qtile, lane);
	const int64_t n_tiles = ((int64_t)p.n_sent + 15) >> 4;
	for (int64_t run = (int64_t)blockIdx.x * 4 + wv; run * VK_SPAN_RUN < n_tiles; run += (int64_t)gridDim.x * 4)
	for (int64_t tile = run * VK_SPAN_RUN; tile < (run + 1) * VK_SPAN_RUN && tile < n_tiles; tile++) {
		const uint8_t *tp = p.tiles + tile * p.tile_bytes;
		f32x4 acc;
		if constexpr (NK32 > 0) acc = sim_tile<NK32, TAIL, true>(qf, tp, lane);
		else acc = sim_tile_generic(p.qtile, tp, p.nk32, p.tail, lane, p.prec);
		const int64_t s_idx = tile * 16 + lane;
		if (lane < 16 && s_idx < p.n_sent) {
			const float raw = acc[0];                       // query column 0, token lane
			const float boost = p.boost ? p.boost[s_idx] : 1.0f;
			p.scores[s_idx] = (raw / p.ref_total) * boost;
			if (p.raw) p.raw[s_idx] = raw;   // null when nothing reads it (vk_query.cpp): a second output array costs a read stream 4 - 7 %
		}
	}
}

template <int NK32, bool TAIL>
static hipError_t launch_span(const VkScoreParams &p, hipStream_t stream) {
	int occ = 0, dev = 0, cus = 256;
	hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, vk_span_kernel<NK32, TAIL>, 256, 0);
	if (e != hipSuccess) return e;
	if (occ < 1) occ = 1;
	if (occ > 3) occ = 3;   // as vk_score_kernel: 12 waves per CU stream HBM fastest
	const char *ov = getenv("VK_BLOCKS_PER_CU");   // read per launch: tools/sweep_dims.py varies it inside one process
	if (ov && atoi(ov) > 0) occ = atoi(ov);
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	const int64_t runs = (((int64_t)p.n_sent + 15) / 16 + VK_SPAN_RUN - 1) / VK_SPAN_RUN;
	const int64_t want = (runs + 3) / 4 > 0 ? (runs + 3) / 4 : 1, cap = (int64_t)cus * occ;
	vk_span_kernel<NK32, TAIL><<<(int)(want < cap ? want : cap), 256, 0, stream>>>(p);
	return hipGetLastError();
}

extern "C" hipError_t vk_launch_span(const VkScoreParams *pp, hipStream_t stream) {
	const VkScoreParams &p = *pp;
	if (p.prec) return launch_span<0, false>(p, stream);
	if (p.nk32 == 10 && p.tail == 1) return launch_span<10, true>(p, stream);
	if (p.nk32 == 24 && p.tail == 0) return launch_span<24, false>(p, stream);
	if (p.nk32 == 12 && p.tail == 0) return launch_span<12, false>(p, stream);
	if (p.nk32 == 32 && p.tail == 0) return launch_span<32, false>(p, stream);
	return launch_span<0, false>(p, stream);
}
