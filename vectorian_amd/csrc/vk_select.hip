// vk_select.hip -- bounded result sets (top-k), candidate selection and the small per-row kernels around them.
#include "vk_common.hip.h"

#include <hipcub/hipcub.hpp>

// ---------------------------------------------------------------------------
// bounded result set: keys = (orderable(score) << 32) | sentence, descending.
// Order = score desc, sentence index desc (match/match_impl.h:8-42);
// admission score > min_score (metric/alignment.h:284).  Each block bitonic-sorts
// 2048 keys in LDS and emits its best k; stages repeat until one block is left.
// ---------------------------------------------------------------------------

#define VK_TOPK_CHUNK 2048

__device__ __forceinline__ uint32_t float_orderable(float f) {
	const uint32_t u = __builtin_bit_cast(uint32_t, f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ void bitonic_sort_desc_2048(uint64_t *keys) {
	for (int k = 2; k <= VK_TOPK_CHUNK; k <<= 1) {
		for (int j = k >> 1; j > 0; j >>= 1) {
			__syncthreads();
			for (int t = threadIdx.x; t < VK_TOPK_CHUNK / 2; t += blockDim.x) {
				const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
				const int ixj = i | j;
				const uint64_t a = keys[i], b = keys[ixj];
				const bool desc = (i & k) == 0;
				if (desc ? (a < b) : (a > b)) { keys[i] = b; keys[ixj] = a; }
			}
		}
	}
	__syncthreads();
}

__global__ __launch_bounds__(256) void vk_topk_scores_kernel(const float *__restrict__ scores, int64_t n, float min_score,
	int32_t k, uint64_t *__restrict__ out) {
	__shared__ uint64_t keys[VK_TOPK_CHUNK];
	const int64_t base = (int64_t)blockIdx.x * VK_TOPK_CHUNK;
	for (int t = threadIdx.x; t < VK_TOPK_CHUNK; t += blockDim.x) {
		const int64_t g = base + t;
		uint64_t key = 0;
		if (g < n) {
			const float s = scores[g];
			if (s > min_score) key = ((uint64_t)float_orderable(s) << 32) | (uint32_t)g;
		}
		keys[t] = key;
	}
	bitonic_sort_desc_2048(keys);
	for (int t = threadIdx.x; t < k; t += blockDim.x) out[(int64_t)blockIdx.x * k + t] = keys[t];
}

// keys whose aligned runs of `run` elements are sorted descending already (the output of an earlier stage): merging
// the runs pairwise -- one flip step, then half-cleaners, everything descending -- takes 21 compare-exchange steps for
// runs of 512 where the full sort takes 66
__device__ __forceinline__ void bitonic_merge_runs_desc_2048(uint64_t *keys, int run) {
	for (int size = run << 1; size <= VK_TOPK_CHUNK; size <<= 1) {
		const int half = size >> 1;
		__syncthreads();
		for (int t = threadIdx.x; t < VK_TOPK_CHUNK / 2; t += blockDim.x) {
			const int blk = t / half, off = t - blk * half;
			const int i = blk * size + off, ixj = blk * size + size - 1 - off;
			const uint64_t a = keys[i], b = keys[ixj];
			if (a < b) { keys[i] = b; keys[ixj] = a; }
		}
		for (int j = half >> 1; j > 0; j >>= 1) {
			__syncthreads();
			for (int t = threadIdx.x; t < VK_TOPK_CHUNK / 2; t += blockDim.x) {
				const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
				const int ixj = i | j;
				const uint64_t a = keys[i], b = keys[ixj];
				if (a < b) { keys[i] = b; keys[ixj] = a; }
			}
		}
	}
	__syncthreads();
}

__global__ __launch_bounds__(256) void vk_topk_keys_kernel(const uint64_t *__restrict__ in, int64_t n, int32_t k,
	uint64_t *__restrict__ out, int32_t run) {
	__shared__ uint64_t keys[VK_TOPK_CHUNK];
	const int64_t base = (int64_t)blockIdx.x * VK_TOPK_CHUNK;
	for (int t = threadIdx.x; t < VK_TOPK_CHUNK; t += blockDim.x) {
		const int64_t g = base + t;
		keys[t] = g < n ? in[g] : 0;
	}
	if (run > 0) bitonic_merge_runs_desc_2048(keys, run);
	else bitonic_sort_desc_2048(keys);
	for (int t = threadIdx.x; t < k; t += blockDim.x) out[(int64_t)blockIdx.x * k + t] = keys[t];
}

// ---------------------------------------------------------------------------
// bounded result set for k <= 64: one wave streams VK_TOPK_PER_WAVE elements and keeps its k best
// keys sorted across the lanes (lane 0 = best).  A batch of 64 candidates is tested against the
// current k-th key with one ballot; the few that pass are inserted by a shift across lanes.
// Two launches (n -> n/4096 * k -> k) replace the multi-stage bitonic sort.
// ---------------------------------------------------------------------------

#define VK_TOPK_PER_WAVE 4096

template <int FROM_KEYS>
__global__ __launch_bounds__(256) void vk_topk_wave_kernel(const float *__restrict__ scores, const uint64_t *__restrict__ keys_in,
	int64_t n, float min_score, int32_t k, int64_t per_wave, uint64_t *__restrict__ out) {
	const int lane = threadIdx.x & 63;
	const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const int64_t a = wave * per_wave;
	if (a >= n) return;
	const int64_t b = a + per_wave < n ? a + per_wave : n;
	uint64_t mine = 0;                       // sorted descending over lanes; 0 = empty
	uint64_t thr = 0;                        // key of lane k-1: candidates must exceed it
	for (int64_t base = a; base < b; base += 64) {
		const int64_t g = base + lane;
		uint64_t key = 0;
		if (g < b) {
			if (FROM_KEYS) key = keys_in[g];
			else {
				const float s = scores[g];
				if (s > min_score) key = ((uint64_t)float_orderable(s) << 32) | (uint32_t)g;
			}
		}
		bool pending = key > thr;
		for (;;) {
			const unsigned long long mask = __ballot(pending && key > thr);
			if (mask == 0) break;
			const int src = __builtin_ctzll(mask);
			const uint64_t nk = __shfl(key, src, 64);
			if (lane == src) pending = false;
			uint64_t up = __shfl_up(mine, 1, 64);
			if (lane == 0) up = ~0ull;
			mine = (mine >= nk) ? mine : ((up >= nk) ? nk : up);
			thr = __shfl(mine, k - 1, 64);
		}
	}
	if (lane < k) out[wave * k + lane] = mine;
}

// per-query selection over the [B x n] score matrix: blockIdx.y = query
template <int FROM_KEYS>
__global__ __launch_bounds__(256) void vk_topk_wave_batch_kernel(const float *__restrict__ scores, const uint64_t *__restrict__ keys_in,
	int64_t n, float min_score, int32_t k, int64_t per_wave, int64_t in_stride, int64_t out_stride, uint64_t *__restrict__ out) {
	const int lane = threadIdx.x & 63;
	const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const int64_t a = wave * per_wave;
	if (a >= n) return;
	const int64_t b = a + per_wave < n ? a + per_wave : n;
	const float *sc = scores ? scores + (int64_t)blockIdx.y * in_stride : nullptr;
	const uint64_t *ki = keys_in ? keys_in + (int64_t)blockIdx.y * in_stride : nullptr;
	uint64_t mine = 0, thr = 0;
	for (int64_t base = a; base < b; base += 64) {
		const int64_t g = base + lane;
		uint64_t key = 0;
		if (g < b) {
			if (FROM_KEYS) key = ki[g];
			else {
				const float s = sc[g];
				if (s > min_score) key = ((uint64_t)float_orderable(s) << 32) | (uint32_t)g;
			}
		}
		bool pending = key > thr;
		for (;;) {
			const unsigned long long mask = __ballot(pending && key > thr);
			if (mask == 0) break;
			const int src = __builtin_ctzll(mask);
			const uint64_t nk = __shfl(key, src, 64);
			if (lane == src) pending = false;
			uint64_t up = __shfl_up(mine, 1, 64);
			if (lane == 0) up = ~0ull;
			mine = (mine >= nk) ? mine : ((up >= nk) ? nk : up);
			thr = __shfl(mine, k - 1, 64);
		}
	}
	if (lane < k) out[(int64_t)blockIdx.y * out_stride + wave * k + lane] = mine;
}

// processed candidates leave the pool: their bound becomes -inf
__global__ void vk_mark_kernel(const uint64_t *__restrict__ keys, int32_t n, float *__restrict__ scores) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && keys[i] != 0) scores[(uint32_t)(keys[i] & 0xffffffffu)] = VK_NEG_INF;
}

// submatch_weight != 0 (reference_score, metric/alignment.h:84-106): the score divides the aligner score by
//   ref(m) = m + ((T - m) / T)^w (T - m),   m = weight of the matched query tokens, T = total weight,
// and m is known only after a traceback.  Stage 1 turns raw into an upper bound of the score: every matched
// pair contributes at most its token weight, so m >= raw; ref is convex in m with its minimum at m_star, so
// ref(m) >= ref(max(raw, m_star)) =: ref_lb(raw) (a margin covers powf).  Negative raw (GLOBAL) is largest
// over the largest ref = T.  Stage 2 (vk_query.cpp) retraces the candidates with the largest bounds.
__global__ void vk_submatch_bound_kernel(const float *__restrict__ raw, const float *__restrict__ boost, int64_t n,
	float total, float w, float m_star, float *__restrict__ scores) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float r = raw[i];
	if (!(r > VK_NEG_INF)) { scores[i] = VK_NEG_INF; return; }
	const float b = boost ? boost[i] : 1.0f;
	float ub;
	if (r <= 0.0f) ub = r / total;
	else {
		float m = fmaxf(r, m_star);
		m = fminf(m, total);
		const float ref = m + powf((total - m) / total, w) * (total - m);
		ub = r / (ref * (1.0f - 4e-6f));
	}
	scores[i] = ub * b;
}

extern "C" hipError_t vk_launch_submatch_bound(const float *raw, const float *boost, int64_t n, float total, float w, float m_star,
	float *scores, hipStream_t stream) {
	vk_submatch_bound_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(raw, boost, n, total, w, m_star, scores);
	return hipGetLastError();
}

// all slices whose bound can still enter the result set: keys (bound, row) of the rows with
// theta <= bound and bound > floor, appended in no particular order; *counter = how many qualify
__global__ void vk_select_ge_kernel(const float *__restrict__ scores, int64_t n, float theta, float floor_excl,
	uint64_t *__restrict__ keys_out, uint32_t *__restrict__ counter, uint32_t cap) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float sc = scores[i];
	if (sc >= theta && sc > floor_excl) {
		const uint32_t at = atomicAdd(counter, 1u);
		if (at < cap) keys_out[at] = ((uint64_t)float_orderable(sc) << 32) | (uint32_t)i;
	}
}

extern "C" hipError_t vk_launch_select_ge(const float *scores, int64_t n, float theta, float floor_excl, uint64_t *keys_out,
	uint32_t *counter, uint32_t cap, hipStream_t stream) {
	hipError_t e = hipMemsetAsync(counter, 0, 4, stream);
	if (e != hipSuccess) return e;
	vk_select_ge_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(scores, n, theta, floor_excl, keys_out, counter, cap);
	return hipGetLastError();
}

extern "C" hipError_t vk_launch_mark(const uint64_t *keys, int32_t n, float *scores, hipStream_t stream) {
	vk_mark_kernel<<<(n + 255) / 256, 256, 0, stream>>>(keys, n, scores);
	return hipGetLastError();
}

extern "C" hipError_t vk_launch_topk_scores(const float *scores, int64_t n, float min_score, int32_t k, uint64_t *out,
	int32_t *n_blocks_out, hipStream_t stream) {
	const int nb = (int)((n + VK_TOPK_CHUNK - 1) / VK_TOPK_CHUNK);
	vk_topk_scores_kernel<<<nb, 256, 0, stream>>>(scores, n, min_score, k, out);
	*n_blocks_out = nb;
	return hipGetLastError();
}

extern "C" hipError_t vk_launch_topk_keys(const uint64_t *in, int64_t n, int32_t k, uint64_t *out, int32_t *n_blocks_out,
	hipStream_t stream) {
	const int nb = (int)((n + VK_TOPK_CHUNK - 1) / VK_TOPK_CHUNK);
	// the input of every stage after the first is whole runs of k keys, each sorted descending (the tail is zeros)
	const bool runs = k >= 2 && k <= VK_TOPK_CHUNK / 2 && (k & (k - 1)) == 0 && n % k == 0;
	vk_topk_keys_kernel<<<nb, 256, 0, stream>>>(in, n, k, out, runs ? k : 0);
	*n_blocks_out = nb;
	return hipGetLastError();
}

extern "C" hipError_t vk_launch_topk_wave(const float *scores, const uint64_t *keys_in, int64_t n, float min_score, int32_t k,
	int64_t per_wave, uint64_t *out, int64_t *n_waves_out, hipStream_t stream) {
	const int64_t nw = (n + per_wave - 1) / per_wave;
	const unsigned grid = (unsigned)((nw + 3) / 4);
	if (keys_in) vk_topk_wave_kernel<1><<<grid, 256, 0, stream>>>(nullptr, keys_in, n, min_score, k, per_wave, out);
	else vk_topk_wave_kernel<0><<<grid, 256, 0, stream>>>(scores, nullptr, n, min_score, k, per_wave, out);
	*n_waves_out = nw;
	return hipGetLastError();
}

extern "C" hipError_t vk_launch_topk_wave_batch(const float *scores, const uint64_t *keys_in, int64_t n, float min_score, int32_t k,
	int64_t per_wave, int32_t n_queries, int64_t in_stride, int64_t out_stride, uint64_t *out, int64_t *n_waves_out, hipStream_t stream) {
	const int64_t nw = (n + per_wave - 1) / per_wave;
	const dim3 grid((unsigned)((nw + 3) / 4), (unsigned)n_queries);
	if (keys_in) vk_topk_wave_batch_kernel<1><<<grid, 256, 0, stream>>>(nullptr, keys_in, n, min_score, k, per_wave, in_stride, out_stride, out);
	else vk_topk_wave_batch_kernel<0><<<grid, 256, 0, stream>>>(scores, nullptr, n, min_score, k, per_wave, in_stride, out_stride, out);
	*n_waves_out = nw;
	return hipGetLastError();
}


// ---------------------------------------------------------------------------
// Result sets of more than VK_MAX_MATCHES entries: the keys of ALL slices, sorted (upstream's ResultSet holds whatever max_matches
// allows, result_set.h:32-68; the block selection above keeps at most 1,024 per 2,048 keys).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vk_keys_all_kernel(const float *__restrict__ scores, int64_t n, float min_score, uint64_t *__restrict__ out) {
	const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= n) return;
	const float s = scores[g];
	out[g] = s > min_score ? ((uint64_t)float_orderable(s) << 32) | (uint32_t)g : 0ull;
}

extern "C" hipError_t vk_launch_sort_all(const float *scores, int64_t n, float min_score, uint64_t *keys_a, uint64_t *keys_b, void *temp, size_t *temp_bytes,
	uint64_t **sorted_out, hipStream_t stream) {
	hipcub::DoubleBuffer<uint64_t> buf(keys_a, keys_b);
	if (!temp) return hipcub::DeviceRadixSort::SortKeysDescending(nullptr, *temp_bytes, buf, (int)n, 0, 64, stream);
	vk_keys_all_kernel<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(scores, n, min_score, keys_a);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return e;
	e = hipcub::DeviceRadixSort::SortKeysDescending(temp, *temp_bytes, buf, (int)n, 0, 64, stream);
	*sorted_out = buf.Current();
	return e;
}
