// vk_score_batch.hip -- queries with common options sharing one pass over the token tiles.
#include "vk_common.hip.h"

// ---------------------------------------------------------------------------
// A batch of queries with common options over one pass of the corpus (contextual layout): every token tile is
// loaded ONCE and multiplied with the QB query tiles (A operands in LDS), the QB similarity strips of the
// wave's 4 slices stay in LDS and the DP runs once per query.  The corpus bytes per query drop by QB; the pass
// is bound by DP issue and by what four waves per CU can keep in flight, not by HBM.
//   LDS: QB query tiles (shared by the workgroup) + per wave QB strips [rows][LT] (+ one column history).
// ---------------------------------------------------------------------------

#define VK_QB_MAX 4

template <int GAP, int LT>
__global__ __launch_bounds__(256) void vk_score_batch_kernel(VkScoreBatchParams p) {
	extern __shared__ float4 vk_smem4[];
	float *smem = reinterpret_cast<float *>(vk_smem4);
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int qb_n = p.n_queries;
	const uint8_t *qlds = reinterpret_cast<const uint8_t *>(smem);
	for (int i = threadIdx.x; i < qb_n * (p.tile_bytes >> 4); i += 256)
		vk_smem4[i] = *reinterpret_cast<const float4 *>(p.qtiles + (size_t)i * 16);
	__syncthreads();
	smem += qb_n * (p.tile_bytes >> 2);
	const int strip = p.s_rows_per_wave * LT + 16;                 // floats of one query's strip
	float *S0 = smem + wv * p.lds_floats_per_wave;
	float *Hh = S0 + p.n_strips * strip;
	const int sigma = lane >> 4, v = lane & 15;

	DpArgs a;
	a.locality = p.locality; a.len_t = 0;
	a.gs = p.gs; a.gt = p.gt; a.a_s = p.a_s; a.a_t = p.a_t; a.open_s = p.open_s; a.open_t = p.open_t;
	a.ws = p.ws; a.wt = p.wt; a.wt0 = p.wt0;
	a.rwmd_symmetric = p.rwmd_symmetric; a.rwmd_normalize_bow = p.rwmd_normalize_bow; a.wmd_bound = 0;
	a.wrd_raw_total = 0.0f;

	constexpr int WSN = GAP == 6 ? 65 : 33;
	float wsr[WSN], wtr[LT];
	if (GAP == 3 || GAP == 6) {
#pragma unroll
		for (int k = 0; k < WSN; k++) wsr[k] = p.ws[k];
#pragma unroll
		for (int k = 0; k < LT; k++) wtr[k] = p.wt[k];
	}
	const int nfull = p.tail ? p.nk32 - 1 : p.nk32;

	const int n_groups = (p.n_sent + 3) >> 2;
	for (int grp = blockIdx.x * 4 + wv; grp < n_groups; grp += gridDim.x * 4) {
		const int s_idx = grp * 4 + sigma;
		const int i0 = s_idx < p.n_sent ? s_idx : p.n_sent;
		const int t_a = p.sent_start[i0], t_b = p.sent_end[i0];
		const int len = t_b - t_a;
		const int g_a = __builtin_amdgcn_readlane(t_a, 0);
		const int g_b = __builtin_amdgcn_readlane(t_b, 48);
		int maxlen = __builtin_amdgcn_readlane(len, 0);
		maxlen = max(maxlen, __builtin_amdgcn_readlane(len, 16));
		maxlen = max(maxlen, __builtin_amdgcn_readlane(len, 32));
		maxlen = max(maxlen, __builtin_amdgcn_readlane(len, 48));

		const int tile0 = g_a >> 4;
		const int ntiles = ((g_b + 15) >> 4) - tile0;
		const uint8_t *tp = p.tiles + (int64_t)tile0 * p.tile_bytes;
		// Other widths: ten K-steps at a time, no branches around the loads (clamped addresses, results zeroed by a select): a
		// branch would make the compiler wait for all outstanding loads at its join.  (Round 1 prefetched tile ti + 1 during the
		// MFMAs of tile ti, with a reload of each group's last tile and 80 registers of staging: 537 against 595 M pairs/s.)
		const int t_last = p.tail ? nfull : nfull - 1;
		auto load10 = [&](const uint8_t *tile, int t0, bf16x8 (&x)[10]) {
#pragma unroll
			for (int i = 0; i < 10; i++) {
				const int t = t0 + i;
				const int tc = t < t_last ? t : t_last;
				const bool half = tc == nfull;                       // the half-filled tail block: lanes 0..31 only
				const bf16x8 ld = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tile + tc * 1024 + (half ? (lane & 31) : lane) * 16));
				const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
				x[i] = (t <= t_last && !(half && lane >= 32)) ? ld : z;
			}
		};
		auto mfma10 = [&](int t0, const bf16x8 (&x)[10], f32x4 (&acc)[VK_QB_MAX]) {
#pragma unroll
			for (int qb = 0; qb < VK_QB_MAX; qb++)
				if (qb < qb_n) {
#pragma unroll
					for (int i = 0; i < 10; i++) {
						const int t = t0 + i;
						const int tc = t < t_last ? t : t_last;
						const bool half = tc == nfull;
						const bf16x8 q = *reinterpret_cast<const bf16x8 *>(qlds + (size_t)qb * p.tile_bytes + tc * 1024 + (half ? (lane & 31) : lane) * 16);
						acc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q, x[i], acc[qb], 0, 0, 0);   // x is zero past the last K-step
					}
				}
		};
		bf16x8 xb[10];
		for (int ti = 0; ti < ntiles; ti++) {
			f32x4 acc[VK_QB_MAX];
#pragma unroll
			for (int qb = 0; qb < VK_QB_MAX; qb++) acc[qb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
			if (p.nk32 == 10 && p.tail) {
				// 300-d rows: the tile's nine full K-steps and the half one, loaded once and multiplied with every query tile
				bf16x8 x[10];
#pragma unroll
				for (int t = 0; t < 9; t++) x[t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tp + t * 1024 + lane * 16));
				x[9] = load_half_block(tp + 9 * 1024, lane, true);
#pragma unroll
				for (int qb = 0; qb < VK_QB_MAX; qb++)
					if (qb < qb_n) {
						const uint8_t *qq = qlds + (size_t)qb * p.tile_bytes;
#pragma unroll
						for (int t = 0; t < 10; t++) {
							bf16x8 q = *reinterpret_cast<const bf16x8 *>(qq + t * 1024 + (t == 9 ? (lane & 31) : lane) * 16);
							if (t == 9) { const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0}; q = lane < 32 ? q : z; }
							acc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q, x[t], acc[qb], 0, 0, 0);
						}
					}
			} else {
				for (int t0 = 0; t0 < p.nk32; t0 += 10) {
					load10(tp, t0, xb);
					mfma10(t0, xb, acc);
				}
			}
#pragma unroll
			for (int qb = 0; qb < VK_QB_MAX; qb++)
				if (qb < qb_n && (lane >> 4) * 4 < LT) {
					f32x4 c4 = acc[qb];
					c4[0] = clip01(c4[0]); c4[1] = clip01(c4[1]); c4[2] = clip01(c4[2]); c4[3] = clip01(c4[3]);
					*reinterpret_cast<f32x4 *>(S0 + qb * strip + (ti * 16 + (lane & 15)) * LT + (lane >> 4) * 4) = c4;
				}
			tp += p.tile_bytes;
		}
		const int rowbase = t_a - tile0 * 16;
		wave_lds_fence();

		const int lenc = len > 0 ? len : 0;
		const int rb = len > 0 ? rowbase : 0;
		for (int qb = 0; qb < qb_n; qb++) {
			const float *S = S0 + qb * strip;
			a.len_t = p.len_t[qb];
			float raw;
			if constexpr (GAP == 0) raw = dp_linear<LT>(S, rb, lenc, maxlen, v, a);
			else if constexpr (GAP == 1) raw = dp_affine<LT>(S, rb, lenc, maxlen, v, a);
			else if constexpr (GAP == 2) raw = dp_general<LT>(S, Hh, p.h_rows, rb, lenc, maxlen, lane, a);
			else if constexpr (GAP == 3) raw = dp_general_reg<LT, 32>(S, rb, lenc, maxlen, v, a, wsr, wtr);
			else if constexpr (GAP == 6) raw = dp_general_reg<LT, 64>(S, rb, lenc, maxlen, v, a, wsr, wtr);
			else raw = rwmd_rows<LT>(S, rb, lenc, maxlen, v, a);
			if (v == 15 && s_idx < p.n_sent) {
				float val = VK_NEG_INF, r = VK_NEG_INF;
				if (len >= 1) {
					const float boost = p.boost ? p.boost[s_idx] : 1.0f;
					r = raw;
					val = (raw / (float)p.len_t[qb]) * boost;
				}
				p.scores[(int64_t)qb * p.n_sent + s_idx] = val;
				if (p.raw) p.raw[(int64_t)qb * p.n_sent + s_idx] = r;
			}
			if (GAP == 2) wave_lds_fence();
		}
		wave_lds_fence();
	}
}

template <int GAP>
static hipError_t launch_score_batch_lt(const VkScoreBatchParams &p, int lt, int grid, size_t smem, hipStream_t stream) {
	switch (lt) {
	case 4: vk_score_batch_kernel<GAP, 4><<<grid, 256, smem, stream>>>(p); break;
	case 8: vk_score_batch_kernel<GAP, 8><<<grid, 256, smem, stream>>>(p); break;
	case 12: vk_score_batch_kernel<GAP, 12><<<grid, 256, smem, stream>>>(p); break;
	default: vk_score_batch_kernel<GAP, 16><<<grid, 256, smem, stream>>>(p); break;
	}
	return hipGetLastError();
}

// one workgroup per CU (the strips of 4 queries fill the LDS); grid = CUs, or fewer for small corpora
extern "C" hipError_t vk_launch_score_batch(const VkScoreBatchParams *pp, int32_t lt, size_t smem, hipStream_t stream) {
	const VkScoreBatchParams &p = *pp;
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	const int64_t want = (((int64_t)p.n_sent + 3) / 4 + 3) / 4;
	const int per_cu = smem > 0 ? (int)((160 * 1024) / smem) : 1;
	const int64_t cap = (int64_t)cus * (per_cu < 1 ? 1 : per_cu > 3 ? 3 : per_cu);
	const int grid = (int)(want < cap ? want : cap);
	const void *fn = nullptr;
	hipError_t e = hipSuccess;
#define VK_BATCH_CASE(G) \
	case G: \
		switch (lt) { case 4: fn = (const void *)vk_score_batch_kernel<G, 4>; break; case 8: fn = (const void *)vk_score_batch_kernel<G, 8>; break; \
			case 12: fn = (const void *)vk_score_batch_kernel<G, 12>; break; default: fn = (const void *)vk_score_batch_kernel<G, 16>; break; } \
		if (smem > 64 * 1024 && (e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)) != hipSuccess) return e; \
		return launch_score_batch_lt<G>(p, lt, grid, smem, stream);
	switch (p.gap_mode) {
	VK_BATCH_CASE(0)
	VK_BATCH_CASE(1)
	VK_BATCH_CASE(2)
	VK_BATCH_CASE(3)
	VK_BATCH_CASE(6)
	VK_BATCH_CASE(4)
	default: return hipErrorNotSupported;
	}
#undef VK_BATCH_CASE
}
