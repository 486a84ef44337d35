// vk_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the brute-force alignment search.
//
// Written for wave64 / MFMA / LDS of gfx950 only (no portability layer).
// Compiled with -ffp-contract=off: the DP recurrences must be the literal fp32
// add / subtract / max sequence of the oracle (oracle/vk_oracle.c).
//
// Reference functions realised here (paths relative to the reference tree):
//   similarity  : vectorian/sim/vector.py:66-78 (cosine of unit rows),
//                 vectorian/core/cpp/metric/metric.h:28-30 (clip),
//                 metric/contextual.cpp:26-63, metric/static.cpp:9-78
//   slices      : slice/contextual.h:65-67, slice/static.h:71-75, document.h:147-169
//   alignment   : metric/alignment.h:247-294 (make_match -> pyalign solve), :84-106 (reference_score),
//                 match/match.h:295-307 (Score)
//   result set  : result_set.h:32-93, match/match_impl.h:8-42
//
// Data layout in HBM ("tiles"): rows (token vectors, or vocabulary vectors) are stored
// as unit-norm bf16 in MFMA operand order.  A tile is 16 consecutive rows; for each
// K-step t of 32 features the tile holds one 1 KiB block in which lane l
// (l = 16*g + i) owns the 16 bytes  row i, features 32t + 8g .. 32t + 8g + 7.
// If d_pad % 32 == 16 the last block is a half block (512 bytes, g = 0, 1 only); lanes
// 32..63 feed zeros to that K-step.  A wave therefore reads a tile with NK perfectly
// coalesced global_load_dwordx4 and feeds the registers to v_mfma_f32_16x16x32_bf16
// without any shuffle or LDS staging.
//
// One MFMA opcode only on the accumulator chain: on gfx950 (ROCm 7.2 hipcc) a
// v_mfma_f32_16x16x16_bf16 whose SrcC is the vDst of the immediately preceding
// v_mfma_f32_16x16x32_bf16 reads stale accumulator registers -- the compiler inserts no
// wait states for that opcode change (tools/probe/mfma_hazard.hip reproduces it:
// 128 of 256 results differ).  Hence the K=16 tail is issued as a K=32 step.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "vk_device.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define VK_NEG_INF (-__builtin_inff())
#ifndef VK_RUN
#define VK_RUN 4   // consecutive groups of 4 slices per wave turn (vk_score_kernel)
#endif
#ifdef VK_DBG_NOINLINE
#define VK_DP_INLINE __attribute__((noinline))
#else
#define VK_DP_INLINE __forceinline__
#endif

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------

template <int CTRL>
__device__ __forceinline__ float dpp_f(float old, float src) {
	// lanes without a source lane keep `old` (bound_ctrl = 0)
	return __builtin_bit_cast(float,
		__builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), CTRL, 0xf, 0xf, false));
}
#define DPP_ROW_SHR1 0x111
#define DPP_ROW_SHR2 0x112
#define DPP_ROW_SHR4 0x114
#define DPP_ROW_SHR8 0x118

// max over each 16-lane DPP row; result valid in lane 15 of the row
__device__ __forceinline__ float row_max_to_lane15(float x) {
	x = fmaxf(x, dpp_f<DPP_ROW_SHR1>(x, x));
	x = fmaxf(x, dpp_f<DPP_ROW_SHR2>(x, x));
	x = fmaxf(x, dpp_f<DPP_ROW_SHR4>(x, x));
	x = fmaxf(x, dpp_f<DPP_ROW_SHR8>(x, x));
	return x;
}

__device__ __forceinline__ void wave_lds_fence() {
	// one wave's LDS operations execute in order; this only stops the compiler
	// from moving LDS accesses across the point and drains lgkmcnt
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float clip01(float x) {
	// xt::clip(sim, 0, 1); NaN -> 0 as the oracle does
	return fminf(fmaxf(x, 0.0f), 1.0f);
}

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float x) {
	uint32_t u = __builtin_bit_cast(uint32_t, x);
	if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
	u += 0x7fffu + ((u >> 16) & 1u);
	return (uint16_t)(u >> 16);
}

// ---------------------------------------------------------------------------
// corpus upload: L2-normalise rows (Vectors.normalized, vectorian/embedding/vectors.py:71-86),
// round to bf16 (RNE) and store in tile order.  One wave per row.
// ---------------------------------------------------------------------------

template <typename T> __device__ __forceinline__ float load_elem(const T *p);
template <> __device__ __forceinline__ float load_elem<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float load_elem<uint16_t>(const uint16_t *p) {
	return __builtin_bit_cast(float, ((uint32_t)*p) << 16);
}

template <typename T>
__global__ __launch_bounds__(256) void vk_pack_rows_kernel(
	const T *__restrict__ in, int64_t n_rows, int32_t d, int32_t d_pad, int64_t row0,
	uint8_t *__restrict__ tiles, float *__restrict__ mag_out, int32_t normalize, int32_t prec) {

	const int lane = threadIdx.x & 63;
	const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= n_rows) return;
	const T *row = in + r * (int64_t)d;

	float m = 1.0f;
	if (normalize || mag_out) {
		double acc = 0.0;
		for (int k = lane; k < d; k += 64) {
			const double x = (double)load_elem<T>(row + k);
			acc += x * x;
		}
		for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
		m = (float)sqrt(acc);
		if (m != m) m = 0.0f;
		if (mag_out && lane == 0) mag_out[row0 + r] = m;
	}

	const int64_t grow = row0 + r;
	const int64_t tile = grow >> 4;
	const int i = (int)(grow & 15);
	const int nk32 = d_pad >> 5;
	const int tile_bytes = prec ? d_pad * 64 : d_pad * 32;
	uint8_t *tp = tiles + tile * (int64_t)tile_bytes;
	if (prec) {
		// fp32 tiles (operand order of v_mfma_f32_16x16x4_f32): a block of 16 features is 1 KiB; lane 16 g + i owns
		// row i, features 16 b + 4 s + g for s = 0..3 (element s feeds MFMA step s)
		for (int k = lane; k < d_pad; k += 64) {
			float x = 0.0f;
			if (k < d) {
				x = load_elem<T>(row + k);
				if (normalize) {
					x = x / m;
					if (x != x) x = 0.0f;
				}
			}
			const int b = k >> 4, sidx = (k & 15) >> 2, g = k & 3;
			*reinterpret_cast<float *>(tp + b * 1024 + (g * 16 + i) * 16 + sidx * 4) = x;
		}
		return;
	}

	const int n8 = d_pad >> 3;                  // 8-element chunks; chunk c: K-step c>>2, lane group c&3
	(void)nk32;
	for (int c = lane; c < n8; c += 64) {
		const int k0 = c * 8;
		const int off = (c >> 2) * 1024 + ((c & 3) * 16 + i) * 16;
		uint16_t v[8];
		for (int j = 0; j < 8; j++) {
			float x = 0.0f;
			if (k0 + j < d) {
				x = load_elem<T>(row + k0 + j);
				if (normalize) {
					x = x / m;
					if (x != x) x = 0.0f;
				}
			}
			v[j] = f32_to_bf16_rne(x);
		}
		uint4 w;
		w.x = v[0] | ((uint32_t)v[1] << 16); w.y = v[2] | ((uint32_t)v[3] << 16);
		w.z = v[4] | ((uint32_t)v[5] << 16); w.w = v[6] | ((uint32_t)v[7] << 16);
		*reinterpret_cast<uint4 *>(tp + off) = w;
	}
}

// ---------------------------------------------------------------------------
// similarity of one 16-row tile against the query: S^T = Q * X^T on MFMA.
// A operand = query fragment (rows = query tokens), B operand = token tile
// (columns = tokens).  Result: lane l holds S[token l&15][query 4*(l>>4) + r], r=0..3.
// ---------------------------------------------------------------------------

// NK = number of K=32 steps (the last one half filled when HALF)
template <int NK, bool HALF>
struct QFrag {
	bf16x8 q[NK > 0 ? NK : 1];
};

__device__ __forceinline__ bf16x8 load_half_block(const uint8_t *__restrict__ p, int lane, bool nt) {
	// half block: 32 x 16 bytes; lanes 32..63 contribute zeros to the K-step
	const bf16x8 *src = reinterpret_cast<const bf16x8 *>(p + (lane & 31) * 16);
	bf16x8 x = nt ? __builtin_nontemporal_load(src) : *src;
	const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
	return lane < 32 ? x : z;
}

template <int NK, bool HALF>
__device__ __forceinline__ void load_qfrag(QFrag<NK, HALF> &f, const uint8_t *__restrict__ qtile, int lane) {
#pragma unroll
	for (int t = 0; t < NK; t++) {
		if (HALF && t == NK - 1) f.q[t] = load_half_block(qtile + t * 1024, lane, false);
		else f.q[t] = *reinterpret_cast<const bf16x8 *>(qtile + t * 1024 + lane * 16);
	}
}

template <int NK, bool HALF>
__device__ __forceinline__ f32x4 sim_tile(const QFrag<NK, HALF> &f, const uint8_t *__restrict__ tile, int lane) {
	bf16x8 x[NK > 0 ? NK : 1];
#pragma unroll
	for (int t = 0; t < NK; t++) {
		if (HALF && t == NK - 1) x[t] = load_half_block(tile + t * 1024, lane, true);
		else x[t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tile + t * 1024 + lane * 16));
	}
	f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
	for (int t = 0; t < NK; t++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.q[t], x[t], acc, 0, 0, 0);
	acc[0] = clip01(acc[0]); acc[1] = clip01(acc[1]); acc[2] = clip01(acc[2]); acc[3] = clip01(acc[3]);
	return acc;
}

// Large d (e.g. 768): the query tile is staged once per workgroup in LDS (NK KiB) and its fragments
// are re-read per K-step with ds_read_b128 (conflict-free: 64 consecutive 16-byte slots); the token
// tile's NK loads are all issued up front.  Same MFMA sequence as sim_tile.
template <int NK, bool HALF>
__device__ __forceinline__ f32x4 sim_tile_qlds(const uint8_t *__restrict__ qlds, const uint8_t *__restrict__ tile, int lane) {
	bf16x8 x[NK];
#pragma unroll
	for (int t = 0; t < NK; t++) {
		if (HALF && t == NK - 1) x[t] = load_half_block(tile + t * 1024, lane, true);
		else x[t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tile + t * 1024 + lane * 16));
	}
	f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
	for (int t = 0; t < NK; t++) {
		bf16x8 q = *reinterpret_cast<const bf16x8 *>(qlds + t * 1024 + ((HALF && t == NK - 1) ? (lane & 31) : lane) * 16);
		if (HALF && t == NK - 1) {
			const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
			q = lane < 32 ? q : z;
		}
		acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q, x[t], acc, 0, 0, 0);
	}
	acc[0] = clip01(acc[0]); acc[1] = clip01(acc[1]); acc[2] = clip01(acc[2]); acc[3] = clip01(acc[3]);
	return acc;
}

// any d: query fragments re-read per K-step (L1/L2 resident), runtime trip count.
// Same MFMA sequence as sim_tile, hence bit-identical similarities.
__device__ __forceinline__ f32x4 sim_tile_generic(const uint8_t *__restrict__ qtile, const uint8_t *__restrict__ tile,
	int nk, int half, int lane, int prec = 0) {
	f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
	if (prec) {
		// fp32 rows (the reference's own precision): nk blocks of 16 features, four v_mfma_f32_16x16x4_f32 per block
		int b = 0;
		for (; b + 4 <= nk; b += 4) {   // four blocks in flight
			f32x4 q[4], x[4];
#pragma unroll
			for (int i = 0; i < 4; i++) {
				x[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(tile + (b + i) * 1024 + lane * 16));
				q[i] = *reinterpret_cast<const f32x4 *>(qtile + (b + i) * 1024 + lane * 16);
			}
#pragma unroll
			for (int i = 0; i < 4; i++) {
#pragma unroll
				for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q[i][e], x[i][e], acc, 0, 0, 0);
			}
		}
		for (; b < nk; b++) {
			const f32x4 x = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(tile + b * 1024 + lane * 16));
			const f32x4 q = *reinterpret_cast<const f32x4 *>(qtile + b * 1024 + lane * 16);
#pragma unroll
			for (int e = 0; e < 4; e++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(q[e], x[e], acc, 0, 0, 0);
		}
		acc[0] = clip01(acc[0]); acc[1] = clip01(acc[1]); acc[2] = clip01(acc[2]); acc[3] = clip01(acc[3]);
		return acc;
	}
	const int nfull = half ? nk - 1 : nk;
	int t = 0;
	for (; t + 4 <= nfull; t += 4) {   // four K-steps in flight
		bf16x8 q[4], x[4];
#pragma unroll
		for (int i = 0; i < 4; i++) {
			x[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tile + (t + i) * 1024 + lane * 16));
			q[i] = *reinterpret_cast<const bf16x8 *>(qtile + (t + i) * 1024 + lane * 16);
		}
#pragma unroll
		for (int i = 0; i < 4; i++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q[i], x[i], acc, 0, 0, 0);
	}
	for (; t < nfull; t++) {
		const bf16x8 q = *reinterpret_cast<const bf16x8 *>(qtile + t * 1024 + lane * 16);
		const bf16x8 x = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tile + t * 1024 + lane * 16));
		acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q, x, acc, 0, 0, 0);
	}
	if (half) {
		const bf16x8 q = load_half_block(qtile + nfull * 1024, lane, false);
		const bf16x8 x = load_half_block(tile + nfull * 1024, lane, true);
		acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q, x, acc, 0, 0, 0);
	}
	acc[0] = clip01(acc[0]); acc[1] = clip01(acc[1]); acc[2] = clip01(acc[2]); acc[3] = clip01(acc[3]);
	return acc;
}

// TagWeightedSlice::similarity (vectorian/core/cpp/slice/static.h:237-264): S * weight(i, j) with
// weight = t_pos_weights[j] * (pos_s != pos_t ? 1 - penalty : 1); values <= threshold become 0.
__device__ __forceinline__ float tag_weighted(float s, float w, int pos_s, int pos_t, float keep, float thr) {
	float wgt = w;
	if (pos_s != pos_t) wgt *= keep;
	const float sc = s * wgt;
	return sc <= thr ? 0.0f : sc;
}

// ---------------------------------------------------------------------------
// DP over a group of 4 sentences: DPP row sigma = lane >> 4 is one sentence, lane
// v = lane & 15 is query column v + 1.  Rows (sentence tokens) are swept serially;
// within a row the left-to-right dependency is resolved by a monotone fixpoint chain
// on DPP row_shr:1, which reproduces the sequential recurrence bit for bit (max is
// exact and x -> x - g is monotone).
//
// S: wave-private LDS [rows][LT] (the LT = 4/8/12/16 padded query columns); sentence sigma's token i is row rowbase + i.
// Lanes v >= LT read into the next row: their values never reach a lower lane.
// Returns the aligner score (raw) in lane 15 of each DPP row.
// ---------------------------------------------------------------------------

struct DpArgs {
	int32_t locality;
	int32_t len_t;
	float gs, gt;          // linear: w(k) = g*k ; affine: extension cost b
	float a_s, a_t;        // affine: a
	float open_s, open_t;  // affine: a + b
	const float *ws;       // general: w_s[0..max_len]
	const float *wt;       // general: w_t[0..16]
	int32_t rwmd_symmetric, rwmd_normalize_bow, wmd_bound;
	float wrd_raw_total;   // WRD on raw magnitudes: sum of the query's magnitudes (0: masses are normalised)
};

// In-row dependency of the linear recurrence H[u][j] = max(c[j], H[u][j-1] - gt): unrolled,
// H[u][j] = max_k (c[j-k] - k gt), a prefix maximum with decay.  It is taken in log2(LT) doubling steps
// x <- max(x, row_shr:s(x) - s gt), s = 1, 2, 4, 8 (s gt is exact for powers of two), ten instructions
// instead of a chain of LT dependent subtractions.  The value may differ from the sequential
// recurrence in the last bit (c - 2 gt is rounded once, (c - gt) - gt twice); scores are compared at
// 1e-4, and the tracebacks of the winners come from vk_flow_kernel, which walks the recurrence
// sequentially.  The border column enters as H[u][0] - (v + 1) gt (non-LOCAL only).
template <int K>
__device__ __forceinline__ float shr_k(float old, float src) { return dpp_f<0x110 + K>(old, src); }

template <int LT>
__device__ __forceinline__ float decay_scan(float x, float g) {
	x = fmaxf(x, shr_k<1>(VK_NEG_INF, x) - g);
	if (LT > 2) x = fmaxf(x, shr_k<2>(VK_NEG_INF, x) - 2.0f * g);
	if (LT > 4) x = fmaxf(x, shr_k<4>(VK_NEG_INF, x) - 4.0f * g);
	if (LT > 8) x = fmaxf(x, shr_k<8>(VK_NEG_INF, x) - 8.0f * g);
	return x;
}

template <int LT>
__device__ __forceinline__ float dp_linear(const float *__restrict__ S, int rowbase, int len, int maxlen, int v, const DpArgs &a) {
	const bool is_local = a.locality == VK_DEV_LOCAL;
	const bool is_global = a.locality == VK_DEV_GLOBAL;
	const float floor0 = is_local ? 0.0f : VK_NEG_INF;
	const float gsb = is_global ? a.gs : 0.0f;   // border H[u][0] = -(gs*u) (GLOBAL) else 0
	const float gs = a.gs, gt = a.gt;
	const bool last_col = v == a.len_t - 1;
	const float gt_v1 = gt * (float)(v + 1);     // distance of this column from the border column

	float h = is_global ? -gt_v1 : 0.0f;  // H[0][v+1]
	float best = 0.0f;
	for (int u = 1; u <= maxlen; u++) {
		const bool act = u <= len;
		const float s = S[(rowbase + (act ? u - 1 : 0)) * LT + v];
		const float bprev = -(gsb * (float)(u - 1));
		const float bcur = -(gsb * (float)u);
		const float diag = dpp_f<DPP_ROW_SHR1>(bprev, h);
		float c = fmaxf(diag + s, floor0);
		c = fmaxf(c, h - gs);
		float hn = decay_scan<LT>(c, gt);
		if (!is_local) hn = fmaxf(hn, bcur - gt_v1);
		h = act ? hn : h;
		if (is_local || last_col) best = fmaxf(best, h);
	}
	float m;
	if (is_local) m = v < a.len_t ? best : 0.0f;
	else if (is_global) m = last_col ? h : VK_NEG_INF;
	else m = v < a.len_t ? fmaxf(h, last_col ? best : 0.0f) : 0.0f;   // last row U last column U border 0
	m = row_max_to_lane15(m);
	return (is_global) ? m : fmaxf(m, 0.0f);
}

// Gotoh, w(k) = a + b*k: E (gap over s tokens) lives in the lane, F (gap over query
// tokens) is resolved with the same fixpoint chain as H.
template <int LT>
__device__ VK_DP_INLINE float dp_affine(const float *__restrict__ S, int rowbase, int len, int maxlen, int v, const DpArgs &a) {
	const bool is_local = a.locality == VK_DEV_LOCAL;
	const bool is_global = a.locality == VK_DEV_GLOBAL;
	const float floor0 = is_local ? 0.0f : VK_NEG_INF;
	const float bs = a.gs, bt = a.gt, open_s = a.open_s, open_t = a.open_t;
	const float a_s = a.a_s, a_t = a.a_t;
	const bool last_col = v == a.len_t - 1;

	// borders (GLOBAL): H[0][j] = -(a_t + bt*j), H[u][0] = -(a_s + bs*u)
	float h = is_global ? -(a_t + bt * (float)(v + 1)) : 0.0f;
	float e = VK_NEG_INF;                       // E[0][j]
	float best = 0.0f;
	for (int u = 1; u <= maxlen; u++) {
		const bool act = u <= len;
		const float s = S[(rowbase + (act ? u - 1 : 0)) * LT + v];
		const float bprev = (is_global && u > 1) ? -(a_s + bs * (float)(u - 1)) : 0.0f;
		const float bcur = is_global ? -(a_s + bs * (float)u) : 0.0f;
		const float diag = dpp_f<DPP_ROW_SHR1>(bprev, h);
		const float en = fmaxf(h - open_s, e - bs);     // E[u][j]
		float c = fmaxf(fmaxf(diag + s, floor0), en);
		// F[u][j] = max(H[u][j-1] - open_t, F[u][j-1] - bt); F[u][0] = -inf
		float f = VK_NEG_INF;
		float hc = c;
		if (a_t >= 0.0f) {
			// opening costs at least an extension, so extending a gap never loses against reopening it from the same
			// cell: F[u][j] = max_k (c[j-k] - open_t - (k-1) bt), the decayed prefix maximum of c shifted by one
			// column (the border column enters at lane 0).  Last-bit differences as in dp_linear.
			f = decay_scan<LT>(dpp_f<DPP_ROW_SHR1>(bcur, c) - open_t, bt);
			hc = fmaxf(c, f);
		} else {
#pragma unroll
			for (int i = 0; i < LT; i++) {
				const float fl = fmaxf(dpp_f<DPP_ROW_SHR1>(bcur, hc) - open_t, dpp_f<DPP_ROW_SHR1>(VK_NEG_INF, f) - bt);
				f = fl;
				hc = fmaxf(c, f);
			}
		}
		if (act) { h = hc; e = en; }
		if (is_local || last_col) best = fmaxf(best, h);
	}
	float m;
	if (is_local) m = v < a.len_t ? best : 0.0f;
	else if (is_global) m = last_col ? h : VK_NEG_INF;
	else m = v < a.len_t ? fmaxf(h, last_col ? best : 0.0f) : 0.0f;
	m = row_max_to_lane15(m);
	return (is_global) ? m : fmaxf(m, 0.0f);
}

// General gap costs (Waterman-Smith-Beyer): the column history H[0..u-1][j] of each
// lane is staged in wave-private LDS (Hh[sigma][u][v]); the in-row dependency walks
// the source columns left to right with ds_bpermute broadcasts.
template <int LT>
__device__ __forceinline__ float dp_general(const float *__restrict__ S, float *__restrict__ Hh, int hstride,
	int rowbase, int len, int maxlen, int lane, const DpArgs &a) {

	const int v = lane & 15, sigma = lane >> 4;
	const bool is_local = a.locality == VK_DEV_LOCAL;
	const bool is_global = a.locality == VK_DEV_GLOBAL;
	const float floor0 = is_local ? 0.0f : VK_NEG_INF;
	const bool last_col = v == a.len_t - 1;
	float *hist = Hh + sigma * hstride * 16 + v;   // hist[u*16] = H[u][v+1]

	// wtl[p] = w_t(j - p) for source column p < j = v + 1, else +inf (no candidate)
	float wtl[LT];
#pragma unroll
	for (int p = 0; p < LT; p++) wtl[p] = (p <= v) ? a.wt[v + 1 - p] : __builtin_inff();

	float h = is_global ? -a.wt[v + 1] : 0.0f;
	hist[0] = h;
	float best = 0.0f;
	for (int u = 1; u <= maxlen; u++) {
		const bool act = u <= len;
		const float s = S[(rowbase + (act ? u - 1 : 0)) * LT + v];
		const float bprev = is_global ? -a.ws[u - 1] : 0.0f;    // ws[0] = 0
		const float bcur = is_global ? -a.ws[u] : 0.0f;
		const float diag = dpp_f<DPP_ROW_SHR1>(bprev, h);
		float c = fmaxf(diag + s, floor0);
		// gaps over s tokens: H[u-k][j] - w_s(k), k = 1..u (each lane re-reads its own column)
		for (int k = 1; k <= u; k++) c = fmaxf(c, hist[(u - k) * 16] - a.ws[k]);
		// gaps over query tokens: H[u][p] - w_t(j - p), p = 0 (border) .. j-1
		c = fmaxf(c, bcur - wtl[0]);
#pragma unroll
		for (int p = 1; p < LT; p++) {
			const float hp = __builtin_bit_cast(float,
				__builtin_amdgcn_ds_bpermute((sigma * 16 + p - 1) * 4, __builtin_bit_cast(int, c)));
			c = fmaxf(c, hp - wtl[p]);
		}
		if (act) { h = c; hist[u * 16] = c; }
		if (is_local || last_col) best = fmaxf(best, h);
	}
	float m;
	if (is_local) m = v < a.len_t ? best : 0.0f;
	else if (is_global) m = last_col ? h : VK_NEG_INF;
	else m = v < a.len_t ? fmaxf(h, last_col ? best : 0.0f) : 0.0f;
	m = row_max_to_lane15(m);
	return (is_global) ? m : fmaxf(m, 0.0f);
}

// General gap costs, fast form: sentences of at most MAXLEN tokens, the column history
// H[0..u-1][j] lives in registers (rows fully unrolled, w_s in scalar registers), and
// the in-row step takes its candidates from the row's values *before* in-row gaps,
//     H[u][j] = max(c[j], max_k c[j-k] - w_t(k)),   c = max(zero, diagonal, gaps over s tokens)
// which needs no serial chain.  This equals the sequential recurrence bit for bit when
// w_t is strictly subadditive (w(a) + w(b) > w(a+b) by a margin far above fp32 rounding):
// two consecutive in-row gaps are then always beaten by the single gap of the summed
// length, so replacing H[j-k] by c[j-k] drops only dominated candidates.  The host
// checks the margin (vk_api.cpp) and otherwise selects dp_general.
template <int LT, int MAXLEN>
__device__ __forceinline__ float dp_general_reg(const float *__restrict__ S, int rowbase, int len, int maxlen, int v,
	const DpArgs &a, const float (&wsr)[MAXLEN + 1], const float (&wtr)[LT]) {

	const bool is_local = a.locality == VK_DEV_LOCAL;
	const bool is_global = a.locality == VK_DEV_GLOBAL;
	const float floor0 = is_local ? 0.0f : VK_NEG_INF;
	const bool last_col = v == a.len_t - 1;
	const float wt_border = a.wt[v + 1];             // distance from the border column to column v + 1

	float hreg[MAXLEN + 1];
	float h = is_global ? -wt_border : 0.0f;
	hreg[0] = h;
	float best = 0.0f;
#pragma unroll
	for (int u = 1; u <= MAXLEN; u++) {
		if (u <= maxlen) {
			const bool act = u <= len;
			const float s = S[(rowbase + (act ? u - 1 : 0)) * LT + v];
			const float bprev = is_global ? -wsr[u - 1] : 0.0f;
			const float bcur = is_global ? -wsr[u] : 0.0f;
			const float diag = dpp_f<DPP_ROW_SHR1>(bprev, hreg[u - 1]);
			float c = fmaxf(diag + s, floor0);
#pragma unroll
			for (int k = 1; k <= u; k++) c = fmaxf(c, hreg[u - k] - wsr[k]);
			float hc = fmaxf(c, bcur - wt_border);
			if (LT > 1) hc = fmaxf(hc, dpp_f<0x111>(VK_NEG_INF, c) - wtr[1]);
			if (LT > 2) hc = fmaxf(hc, dpp_f<0x112>(VK_NEG_INF, c) - wtr[2]);
			if (LT > 3) hc = fmaxf(hc, dpp_f<0x113>(VK_NEG_INF, c) - wtr[3]);
			if (LT > 4) hc = fmaxf(hc, dpp_f<0x114>(VK_NEG_INF, c) - wtr[4]);
			if (LT > 5) hc = fmaxf(hc, dpp_f<0x115>(VK_NEG_INF, c) - wtr[5]);
			if (LT > 6) hc = fmaxf(hc, dpp_f<0x116>(VK_NEG_INF, c) - wtr[6]);
			if (LT > 7) hc = fmaxf(hc, dpp_f<0x117>(VK_NEG_INF, c) - wtr[7]);
			if (LT > 8) hc = fmaxf(hc, dpp_f<0x118>(VK_NEG_INF, c) - wtr[8]);
			if (LT > 9) hc = fmaxf(hc, dpp_f<0x119>(VK_NEG_INF, c) - wtr[9]);
			if (LT > 10) hc = fmaxf(hc, dpp_f<0x11a>(VK_NEG_INF, c) - wtr[10]);
			if (LT > 11) hc = fmaxf(hc, dpp_f<0x11b>(VK_NEG_INF, c) - wtr[11]);
			if (LT > 12) hc = fmaxf(hc, dpp_f<0x11c>(VK_NEG_INF, c) - wtr[12]);
			if (LT > 13) hc = fmaxf(hc, dpp_f<0x11d>(VK_NEG_INF, c) - wtr[13]);
			if (LT > 14) hc = fmaxf(hc, dpp_f<0x11e>(VK_NEG_INF, c) - wtr[14]);
			if (LT > 15) hc = fmaxf(hc, dpp_f<0x11f>(VK_NEG_INF, c) - wtr[15]);
			hreg[u] = hc;
			h = act ? hc : h;
			if (is_local || last_col) best = fmaxf(best, h);
		}
	}
	float m;
	if (is_local) m = v < a.len_t ? best : 0.0f;
	else if (is_global) m = last_col ? h : VK_NEG_INF;
	else m = v < a.len_t ? fmaxf(h, last_col ? best : 0.0f) : 0.0f;
	m = row_max_to_lane15(m);
	return (is_global) ? m : fmaxf(m, 0.0f);
}

// Relaxed Word Mover's Distance, injective form (vectorian/core/cpp/alignment/wmd.h:287-416 with
// the BOW builders of alignment/bow.h:204-333).  For the injective solver every vocabulary entry
// moves its whole mass to its nearest partner, so repeated tokens contribute the same distance
// once per occurrence and the joint-vocabulary formulation reduces to positions:
//   acc0 = sum_j w_t * min_i D[i][j]   (t -> s, wmd.h:303-306 computes this direction first)
//   acc1 = sum_i w_s * min_j D[i][j]   (s -> t)
// D = max(1 - S, 0) (wmd.h:107-135); nbow: w = 1/len, bow: w = 1 and acc /= len (wmd.h:379-381);
// cost = acc0, or max(acc0, acc1) when symmetric (wmd.h:383-390); score = (max_cost - cost)/max_cost
// with max_cost = 1 (nbow) or len_t (wmd.h:411-415).  Sums run in position order in fp32.
template <int LT>
__device__ __forceinline__ float rwmd_rows(const float *__restrict__ S, int rowbase, int len, int maxlen, int v, const DpArgs &a) {
	const int len_t = a.len_t;
	const bool nbow = a.rwmd_normalize_bow != 0;
	const float w_t = nbow ? 1.0f / (float)len_t : 1.0f;
	const float w_s = nbow ? 1.0f / (float)(len > 0 ? len : 1) : 1.0f;
	const bool col_ok = v < len_t;
	float colmin = 3.402823466e+38F;
	float acc1 = 0.0f;
	for (int u = 1; u <= maxlen; u++) {
		const bool act = u <= len;
		const float s = S[(rowbase + (act ? u - 1 : 0)) * LT + v];
		const float dist = fmaxf(1.0f - s, 0.0f);
		if (act) colmin = fminf(colmin, dist);
		// min over the query columns of this row -> lane 15
		float m = col_ok ? dist : 3.402823466e+38F;
		m = fminf(m, dpp_f<DPP_ROW_SHR1>(m, m));
		m = fminf(m, dpp_f<DPP_ROW_SHR2>(m, m));
		m = fminf(m, dpp_f<DPP_ROW_SHR4>(m, m));
		m = fminf(m, dpp_f<DPP_ROW_SHR8>(m, m));
		if (act) acc1 += w_s * m;
	}
	// acc0: sequential sum over the query columns, lane by lane
	const float x = col_ok ? w_t * colmin : 0.0f;
	float sum = x;
#pragma unroll
	for (int i = 1; i < LT; i++) {
		const float t = dpp_f<DPP_ROW_SHR1>(0.0f, sum);
		if (v == i) sum = t + x;
	}
	// lane len_t - 1 holds acc0; move it to lane 15 with a max-reduction over one valid lane
	float acc0 = (v == len_t - 1) ? sum : VK_NEG_INF;
	acc0 = row_max_to_lane15(acc0);
	if (!nbow) {
		acc0 = acc0 / (float)len_t;
		acc1 = acc1 / (float)(len > 0 ? len : 1);
	}
	if (a.wmd_bound) {
		// stage 1 of the full WMD: every unit of the side that is shipped completely travels at least to
		// its nearest partner, so 1 - (that side's relaxed cost) bounds the score from above.  nbow: both
		// sides ship everything; bow (unit masses): the shorter side does.
		float lb;
		if (a.wmd_bound == 1) lb = fmaxf(acc0, acc1);
		else lb = len_t <= len ? acc0 : acc1;
		return fminf(1.0f - lb + 3e-5f, 1.0f);
	}
	float cost = 0.0f;
	if (a.rwmd_symmetric) {
		if (acc0 > cost) cost = acc0;
		if (acc1 > cost) cost = acc1;
	} else {
		cost = acc0;
	}
	const float max_cost = nbow ? 1.0f : (float)len_t;
	return (max_cost - cost) / max_cost;
}

// Relaxed Word Mover's Distance, 1:n form (rwmd('nbow/distributed'); RelaxedSolver with injective = false,
// vectorian/core/cpp/alignment/wmd.h:339-376): every source ships its mass to the targets in order of
// ascending distance, each target taking at most its own mass; as upstream is written, whatever the last
// (partial) shipment carried is charged once more at the maximum distance 1 (wmd.h:373-375 keeps
// `remaining` after the break).  Sources and targets are vocabulary entries: q_mass / smass hold count / len
// at the first occurrence of a token and 0 at its repetitions, which then neither ship nor receive
// (contextual layout: every position is an entry of its own).
//   direction 0 (t -> s): lane = source (query token); its column is consumed in ascending (distance, row)
//     order, one selection sweep over the rows per target taken;
//   direction 1 (s -> t): row = source; the 16 lanes of the DPP row are the targets, selected by a row
//     minimum over keys (distance bits with the lane in the low 4 bits: distances closer than 2^-19
//     relative may swap, which moves the cost by less than 1e-7).
template <int LT>
__device__ __forceinline__ float rwmd_fill_rows(const float *__restrict__ S, const float *__restrict__ smass, int rowbase, int len, int maxlen,
	int lane, const DpArgs &a, float q_mass) {
	const int v = lane & 15, sigma = lane >> 4;
	const int len_t = a.len_t;
	const bool nbow = a.rwmd_normalize_bow != 0;
	const bool col_ok = v < len_t;
	const float cap_s = nbow ? 1.0f / (float)(len > 0 ? len : 1) : 1.0f;   // capacity of a slice position
	const float INF = __builtin_inff();

	// ---- direction 0
	float rem = col_ok ? q_mass : 0.0f, cost0 = 0.0f, last_d = -1.0f;
	int last_i = -1;
	bool fin = !(rem > 0.0f) || len < 1;
	for (int round = 0; round < maxlen && __any(!fin); round++) {
		float bd = INF;
		int bi = -1;
		for (int u = 1; u <= maxlen; u++) {
			const bool act = u <= len;
			const float dist = fmaxf(1.0f - S[(rowbase + (act ? u - 1 : 0)) * LT + v], 0.0f);
			const bool later = dist > last_d || (dist == last_d && u - 1 > last_i);
			if (act && later && dist < bd) { bd = dist; bi = u - 1; }
		}
		if (!fin) {
			// capacity of the target: the mass of its vocabulary entry (0 at the repetitions of a token)
			const float cap = (smass && bi >= 0) ? smass[bi] : cap_s;
			if (bi < 0) fin = true;
			else if (rem <= cap) { cost0 += rem * bd; fin = true; }
			else { rem -= cap; cost0 += cap * bd; last_d = bd; last_i = bi; }
		}
	}
	if (rem > 0.0f) cost0 += rem;
	// sum over the query columns in order, lane by lane (as rwmd_rows)
	float sum = cost0;
#pragma unroll
	for (int i = 1; i < LT; i++) {
		const float t = dpp_f<DPP_ROW_SHR1>(0.0f, sum);
		if (v == i) sum = t + cost0;
	}
	float acc0 = (v == len_t - 1) ? sum : VK_NEG_INF;
	acc0 = row_max_to_lane15(acc0);

	// ---- direction 1
	float acc1 = 0.0f;
	if (a.rwmd_symmetric) {
		for (int u = 1; u <= maxlen; u++) {
			const bool act = u <= len;
			const float dist = fmaxf(1.0f - S[(rowbase + (act ? u - 1 : 0)) * LT + v], 0.0f);
			float r1 = act ? (smass ? smass[u - 1] : cap_s) : 0.0f;
			float cost = 0.0f;
			bool used = !col_ok, done = !(r1 > 0.0f);
			for (int r = 0; r < LT && __any(!done); r++) {
				int key = used ? 0x7fffffff : ((__builtin_bit_cast(int, dist) & ~15) | v);
				key = min(key, __builtin_amdgcn_update_dpp(key, key, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
				key = min(key, __builtin_amdgcn_update_dpp(key, key, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
				key = min(key, __builtin_amdgcn_update_dpp(key, key, 0x141, 0xf, 0xf, false));   // row_half_mirror
				key = min(key, __builtin_amdgcn_update_dpp(key, key, 0x140, 0xf, 0xf, false));   // row_mirror
				const int tl = key & 15;
				const float td = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((sigma * 16 + tl) * 4, __builtin_bit_cast(int, dist)));
				const float tc = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((sigma * 16 + tl) * 4, __builtin_bit_cast(int, q_mass)));
				if (!done) {
					if (key == 0x7fffffff) done = true;
					else if (r1 <= tc) { cost += r1 * td; done = true; }
					else { r1 -= tc; cost += tc * td; }
				}
				if (v == tl) used = true;
			}
			if (r1 > 0.0f) cost += r1;
			acc1 += cost;
		}
	}
	if (!nbow) {
		acc0 = acc0 / (float)len_t;
		acc1 = acc1 / (float)(len > 0 ? len : 1);
	}
	float cost = acc0;
	if (a.rwmd_symmetric) { cost = 0.0f; if (acc0 > cost) cost = acc0; if (acc1 > cost) cost = acc1; }
	const float max_cost = nbow ? 1.0f : (float)len_t;
	return (max_cost - cost) / max_cost;
}

// Word Rotator's Distance, stage 1: an upper bound of the score for every sentence.
// WRD = 1 - EMD of the transport problem with masses |x| / sum|x| and costs max(0, 1 - S)
// (vectorian/core/cpp/alignment/wrd.h:62-146).  Every unit of mass travels at least to its nearest
// partner, so both  sum_j m_t[j] min_i C[j][i]  and  sum_i m_s[i] min_j C[j][i]  bound the EMD from
// below; the larger of the two (minus a margin for fp32 rounding) gives score <= 1 - LB.
// Stage 2 (vk_wrd_exact_kernel) solves the survivors exactly.
template <int LT>
__device__ __forceinline__ float wrd_bound_rows(const float *__restrict__ S, int rowbase, int len, int maxlen, int v,
	const DpArgs &a, const float *__restrict__ mag, const int32_t *__restrict__ ids, float q_mass) {
	const int len_t = a.len_t;
	const bool col_ok = v < len_t;
	// one sweep: sum of the slice's magnitudes and sum of magnitude x nearest distance (the quotient is the
	// bound of the s -> t direction; it is only a bound, so the order of the float operations is free)
	// t -> s direction: a query token's mass cannot all go to its nearest slice token -- that one takes at most its
	// own mass.  The lane keeps the four nearest (distance, mass) pairs of its column; filling them in order and
	// charging what is left at the fourth distance bounds the cost of any feasible plan from below (a capacity-
	// constrained relaxation, ICT / ACT of Atasu & Mittelholzer): far tighter than the nearest-neighbour bound when a
	// query token weighs several slice tokens (10 against 32 tokens), and the exact stage sees that many fewer rows.
	float sum_s = 0.0f;
	float lb1n = 0.0f;
	const float BIG = 3.402823466e+38F;
	float d1 = BIG, d2 = BIG, d3 = BIG, d4 = BIG, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f, c4 = 0.0f;
	for (int u = 1; u <= maxlen; u++) {
		const bool act = u <= len;
		const float s = S[(rowbase + (act ? u - 1 : 0)) * LT + v];
		const float dist = fmaxf(1.0f - s, 0.0f);
		float m = col_ok ? dist : 3.402823466e+38F;
		m = fminf(m, dpp_f<DPP_ROW_SHR1>(m, m));
		m = fminf(m, dpp_f<DPP_ROW_SHR2>(m, m));
		m = fminf(m, dpp_f<DPP_ROW_SHR4>(m, m));
		m = fminf(m, dpp_f<DPP_ROW_SHR8>(m, m));
		// static layout: magnitude of the vocabulary entry; no magnitudes: unit masses (bags of words, full WMD)
		const float mg = act ? (mag ? (ids ? mag[ids[u - 1]] : mag[u - 1]) : 1.0f) : 0.0f;
		sum_s += mg;
		lb1n += mg * m;
		// insert (dist, mg) into the sorted four
		float nd = act ? dist : BIG, nc = mg;
		bool sw;
		sw = nd < d4; d4 = sw ? nd : d4; c4 = sw ? nc : c4;
		sw = d4 < d3; nd = d3; nc = c3; d3 = sw ? d4 : d3; c3 = sw ? c4 : c3; d4 = sw ? nd : d4; c4 = sw ? nc : c4;
		sw = d3 < d2; nd = d2; nc = c2; d2 = sw ? d3 : d2; c2 = sw ? c3 : c2; d3 = sw ? nd : d3; c3 = sw ? nc : c3;
		sw = d2 < d1; nd = d1; nc = c1; d1 = sw ? d2 : d1; c1 = sw ? c2 : c1; d2 = sw ? nd : d2; c2 = sw ? nc : c2;
	}
	const float lb1 = lb1n / sum_s * (1.0f - 2e-6f);
	// masses on the scale of q_mass: normalised (shares of sum_s) unless the magnitudes are used as they are
	const float scale = a.wrd_raw_total > 0.0f ? 1.0f : 1.0f / sum_s;
	float rem = col_ok ? q_mass : 0.0f, x = 0.0f, last = 0.0f;
	{
		float amt;
		if (d1 < BIG) { amt = fminf(rem, c1 * scale * (1.0f + 2e-6f)); x += amt * d1; rem -= amt; last = d1; }
		if (d2 < BIG) { amt = fminf(rem, c2 * scale * (1.0f + 2e-6f)); x += amt * d2; rem -= amt; last = d2; }
		if (d3 < BIG) { amt = fminf(rem, c3 * scale * (1.0f + 2e-6f)); x += amt * d3; rem -= amt; last = d3; }
		if (d4 < BIG) { amt = fminf(rem, c4 * scale * (1.0f + 2e-6f)); x += amt * d4; rem -= amt; last = d4; }
		x += fmaxf(rem, 0.0f) * last;
	}
	// sum over lanes, any order: it is only a bound
	x += dpp_f<DPP_ROW_SHR1>(0.0f, x);
	x += dpp_f<DPP_ROW_SHR2>(0.0f, x);
	x += dpp_f<DPP_ROW_SHR4>(0.0f, x);
	x += dpp_f<DPP_ROW_SHR8>(0.0f, x);
	float lb;
	if (a.wrd_raw_total > 0.0f) {
		// magnitudes as they are (normalize_magnitudes = false): min(sum_t, sum_s) units are shipped and the score
		// is 1 - cost / shipped; only the lighter side ships everything, so only its relaxed cost is a bound
		const float sum_t = a.wrd_raw_total;
		lb = (sum_t <= sum_s ? x / sum_t : lb1) * (1.0f - 2e-6f);
		if (!(sum_s > 0.0f)) lb = 0.0f;
	} else lb = fmaxf(x, lb1);
	return fminf(1.0f - lb + 3e-5f, 1.0f);
}

// ---------------------------------------------------------------------------
// the fused scoring kernel: one wave = 4 sentences at a time, grid-stride over groups.
//   MODE 0: contextual layout, NK32 K-steps (last one half filled when TAIL), query fragments in registers
//   MODE 1: contextual layout, any d (runtime K loop)
//   MODE 2: static layout: gather rows of the per-query table by token id
//   MODE 3: contextual layout, NK32 K-steps, query tile staged in LDS (large d)
// GAP: 0 linear, 1 affine, 2 general (LDS history, serial in-row chain),
//      3 general, sentences <= 32 tokens, strictly subadditive w_t (register history),
//      4 relaxed word mover's distance (no DP: row / column minima of 1 - S),
//      5 word rotator's distance, upper bound of the score (stage 1),
//      6 as 3 for sentences <= 64 tokens,
//      7 relaxed word mover's distance, 1:n form (greedy fill by ascending distance).
// LT: padded query length (4, 8, 12, 16).
// ---------------------------------------------------------------------------

template <int MODE, int NK32, bool TAIL, int GAP, int LT>
__global__ __launch_bounds__(256) void vk_score_kernel(VkScoreParams p) {
	extern __shared__ float4 vk_smem4[];
	float *smem = reinterpret_cast<float *>(vk_smem4);
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int nwaves = blockDim.x >> 6;      // 4; 1 in the pass over long slices (group_list)
	// MODE 3: the first NK32 KiB of the dynamic LDS hold the query tile (shared by the block's waves)
	const uint8_t *qlds = reinterpret_cast<const uint8_t *>(smem);
	if constexpr (MODE == 3) {
		for (int i = threadIdx.x; i < NK32 * 64; i += blockDim.x)
			vk_smem4[i] = *reinterpret_cast<const float4 *>(p.qtile + i * 16);
		__syncthreads();
		smem += NK32 * 256;
	}
	float *S = smem + wv * p.lds_floats_per_wave;
	float *Hh = S + p.s_rows_per_wave * LT + 16;   // strip rows hold the LT query columns only; 16 floats of slack for lanes >= LT
	const int sigma = lane >> 4, v = lane & 15;

	QFrag<NK32, TAIL> qf;
	if constexpr (MODE == 0) load_qfrag<NK32, TAIL>(qf, p.qtile, lane);

	DpArgs a;
	a.locality = p.locality; a.len_t = p.len_t;
	a.gs = p.gs; a.gt = p.gt; a.a_s = p.a_s; a.a_t = p.a_t; a.open_s = p.open_s; a.open_t = p.open_t;
	a.ws = p.ws; a.wt = p.wt;
	a.rwmd_symmetric = p.rwmd_symmetric; a.rwmd_normalize_bow = p.rwmd_normalize_bow; a.wmd_bound = p.wmd_bound;
	a.wrd_raw_total = p.wrd_raw_total;
	const float inv_ref = p.ref_total;
	// tag-weighted modifier: this lane's four query columns (MFMA layout: 4*(lane>>4) + r)
	float twl[4] = {1.0f, 1.0f, 1.0f, 1.0f};
	int tposl[4] = {0, 0, 0, 0};
	if (p.pos_s) {
		const int cbase = (MODE == 2) ? (lane & 3) * 4 : (lane >> 4) * 4;
#pragma unroll
		for (int r = 0; r < 4; r++) { twl[r] = p.tw[cbase + r]; tposl[r] = p.tpos[cbase + r]; }
	}

	// general gap, fast form: gap tables in (scalar) registers for the whole kernel
	constexpr int WSN = GAP == 6 ? 65 : 33;
	float wsr[WSN], wtr[LT];
	if (GAP == 3 || GAP == 6) {
#pragma unroll
		for (int k = 0; k < WSN; k++) wsr[k] = p.ws[k];
#pragma unroll
		for (int k = 0; k < LT; k++) wtr[k] = p.wt[k];
	}

	// groups of 4 consecutive slices; slices longer than max_short_len sit alone in their group (the host
	// pads the slice table, vk_api.cpp set_slices_impl) and are left to a second launch that walks
	// group_list with one wave per workgroup and a larger LDS strip
	// A wave takes runs of VK_RUN consecutive groups: when sentences are not tile-aligned, the tile that
	// straddles two groups is then computed once and its rows are carried over in the strip (ragged corpora:
	// one tile in ten).
	const int n_groups = p.group_list ? p.n_list : (p.n_sent + 3) >> 2;
	const int run = p.group_list ? 1 : VK_RUN;
	const int n_runs = (n_groups + run - 1) / run;
	for (int ri = blockIdx.x * nwaves + wv; ri < n_runs; ri += gridDim.x * nwaves) {
	int prev_tile = -1, prev_row = 0;   // last tile of the previous group of this run, and its place in the strip
	for (int gg = 0; gg < run; gg++) {
		const int gi = ri * run + gg;
		if (gi >= n_groups) break;
		const int grp = p.group_list ? p.group_list[gi] : gi;
		const int s_idx = grp * 4 + sigma;
		const int i0 = s_idx < p.n_sent ? s_idx : p.n_sent;       // entries >= n_sent are empty slices
		const int t_a = p.sent_start[i0], t_b = p.sent_end[i0];
		const int len = t_b - t_a;
		const int g_a = __builtin_amdgcn_readlane(t_a, 0);
		const int g_b = __builtin_amdgcn_readlane(t_b, 48);
		int maxlen = __builtin_amdgcn_readlane(len, 0);
		maxlen = max(maxlen, __builtin_amdgcn_readlane(len, 16));
		maxlen = max(maxlen, __builtin_amdgcn_readlane(len, 32));
		maxlen = max(maxlen, __builtin_amdgcn_readlane(len, 48));
		if (!p.group_list && maxlen > p.max_short_len) { prev_tile = -1; continue; }

		int rowbase;
		if (MODE == 2) {
			// gather: lane handles token (lane >> 2) + 16*it, 4 query columns (lane & 3)
			// two dependent loads per token (id, then the table row): issue them four tokens deep so that the
			// L2 latencies overlap instead of adding up
			const int ntok = g_b - g_a;
			for (int it0 = 0; it0 * 16 < ntok; it0 += 4) {
				int id[4];
				float4 val[4];
#pragma unroll
				for (int q4 = 0; q4 < 4; q4++) {
					const int tk = (it0 + q4) * 16 + (lane >> 2);
					id[q4] = p.tok_id[g_a + (tk < ntok ? tk : 0)];
				}
#pragma unroll
				for (int q4 = 0; q4 < 4; q4++)
					val[q4] = *reinterpret_cast<const float4 *>(p.table + (int64_t)id[q4] * 16 + (lane & 3) * 4);
#pragma unroll
				for (int q4 = 0; q4 < 4; q4++) {
					const int tk = (it0 + q4) * 16 + (lane >> 2);
					if (tk < ntok) {
						float4 vq = val[q4];
						if (p.pos_s) {
							const int ps = p.pos_s[g_a + tk];
							vq.x = tag_weighted(vq.x, twl[0], ps, tposl[0], p.tw_keep, p.tw_threshold);
							vq.y = tag_weighted(vq.y, twl[1], ps, tposl[1], p.tw_keep, p.tw_threshold);
							vq.z = tag_weighted(vq.z, twl[2], ps, tposl[2], p.tw_keep, p.tw_threshold);
							vq.w = tag_weighted(vq.w, twl[3], ps, tposl[3], p.tw_keep, p.tw_threshold);
						}
						if ((lane & 3) * 4 < LT) *reinterpret_cast<float4 *>(S + tk * LT + (lane & 3) * 4) = vq;
					}
				}
			}
			rowbase = t_a - g_a;
		} else {
			const int tile0 = g_a >> 4;
			const int ntiles = ((g_b + 15) >> 4) - tile0;
			const uint8_t *tp = p.tiles + (int64_t)tile0 * p.tile_bytes;
			int ti0 = 0;
			if (ntiles > 0 && tile0 == prev_tile) {
				// the previous group ended inside this tile: its 16 rows are in the strip already
				if (prev_row != 0) {
					float4 keep = {0.0f, 0.0f, 0.0f, 0.0f};
					if (lane < 4 * LT) keep = reinterpret_cast<const float4 *>(S)[prev_row * 4 * LT + lane];   // 16 rows of LT floats
					wave_lds_fence();
					if (lane < 4 * LT) reinterpret_cast<float4 *>(S)[lane] = keep;
				}
				ti0 = 1;
				tp += p.tile_bytes;
			}
			if (ntiles > 0) { prev_tile = tile0 + ntiles - 1; prev_row = ntiles - 1; }
			for (int ti = ti0; ti < ntiles; ti++) {
				f32x4 acc;
				if constexpr (MODE == 0) acc = sim_tile<NK32, TAIL>(qf, tp, lane);
				else if constexpr (MODE == 3) acc = sim_tile_qlds<NK32, TAIL>(qlds, tp, lane);
				else acc = sim_tile_generic(p.qtile, tp, p.nk32, p.tail, lane, p.prec);
				if (p.pos_s) {
					const int ps = p.pos_s[(tile0 + ti) * 16 + (lane & 15)];
#pragma unroll
					for (int r = 0; r < 4; r++) acc[r] = tag_weighted(acc[r], twl[r], ps, tposl[r], p.tw_keep, p.tw_threshold);
				}
				if ((lane >> 4) * 4 < LT) *reinterpret_cast<f32x4 *>(S + (ti * 16 + (lane & 15)) * LT + (lane >> 4) * 4) = acc;
				tp += p.tile_bytes;
			}
			rowbase = t_a - tile0 * 16;
		}
		wave_lds_fence();

		float raw;
		const int lenc = len > 0 ? len : 0;
		const int rb = len > 0 ? rowbase : 0;
		if constexpr (GAP == 0) raw = dp_linear<LT>(S, rb, lenc, maxlen, v, a);
		else if constexpr (GAP == 1) raw = dp_affine<LT>(S, rb, lenc, maxlen, v, a);
		else if constexpr (GAP == 2) raw = dp_general<LT>(S, Hh, p.h_rows, rb, lenc, maxlen, lane, a);
		else if constexpr (GAP == 3) raw = dp_general_reg<LT, 32>(S, rb, lenc, maxlen, v, a, wsr, wtr);
		else if constexpr (GAP == 6) raw = dp_general_reg<LT, 64>(S, rb, lenc, maxlen, v, a, wsr, wtr);
		else if constexpr (GAP == 4) {
			// stage 1 of the full WMD over normalised bags of words = the WRD bound with unit magnitudes
			if (a.wmd_bound == 1) raw = wrd_bound_rows<LT>(S, rb, lenc, maxlen, v, a, nullptr, nullptr, 1.0f / (float)a.len_t);
			else raw = rwmd_rows<LT>(S, rb, lenc, maxlen, v, a);
		}
		else if constexpr (GAP == 7) {
			// masses of the slice's vocabulary entries (static layout: repeated token ids count once, at their
			// first position); stride 0 in the pass over long slices, where only DPP row 0 holds a slice
			float *sm = nullptr;
			if (MODE == 2) {
				sm = Hh + sigma * p.m_rows;
				const float unit = p.rwmd_normalize_bow ? 1.0f / (float)(lenc > 0 ? lenc : 1) : 1.0f;
				for (int u = v; u < lenc; u += 16) {
					const int id = p.tok_id[t_a + u];
					int cnt = 0;
					bool first = true;
					for (int i = 0; i < lenc; i++) {
						const bool same = p.tok_id[t_a + i] == id;
						cnt += same ? 1 : 0;
						first = first && !(same && i < u);
					}
					sm[u] = first ? (float)cnt * unit : 0.0f;
				}
				wave_lds_fence();
			}
			raw = rwmd_fill_rows<LT>(S, sm, rb, lenc, maxlen, lane, a, p.qmass[v]);
		}
		else if constexpr (MODE == 2) raw = wrd_bound_rows<LT>(S, rb, lenc, maxlen, v, a, p.mag, p.tok_id + (len > 0 ? t_a : 0), p.qmass[v]);
		else raw = wrd_bound_rows<LT>(S, rb, lenc, maxlen, v, a, p.mag + (len > 0 ? t_a : 0), nullptr, p.qmass[v]);

		if (v == 15 && s_idx < p.n_sent) {
			// Score::value = raw / reference_score * boost; reference_score == len_t for
			// submatch_weight == 0 (metric/alignment.h:84-106, match/match.h:302-307)
			float val = VK_NEG_INF, r = VK_NEG_INF;
			if (len >= 1) {   // document.h:160 skips empty slices
				const float boost = p.boost ? p.boost[s_idx] : 1.0f;
				r = raw;
				val = (raw / inv_ref) * boost;
			}
			p.scores[s_idx] = val;
			p.raw[s_idx] = r;
		}
		wave_lds_fence();
	}
	}
}

// ---------------------------------------------------------------------------
// static layout: per-query similarity table [V_pad x 16] (metric/static.cpp:9-78)
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void vk_table_kernel(const uint8_t *__restrict__ etiles, const uint8_t *__restrict__ qtile,
	int32_t n_tiles, int32_t nk32, int32_t tail, int32_t tile_bytes, float *__restrict__ table, int32_t prec) {
	const int lane = threadIdx.x & 63;
	const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
	if (tile >= n_tiles) return;
	const f32x4 acc = sim_tile_generic(qtile, etiles + (int64_t)tile * tile_bytes, nk32, tail, lane, prec);
	*reinterpret_cast<f32x4 *>(table + ((int64_t)tile * 16 + (lane & 15)) * 16 + (lane >> 4) * 4) = acc;
}

// sim[id(t_j)][j] = 1 (metric/static.cpp:58-67); runs after vk_table_kernel
__global__ void vk_table_fix_kernel(float *__restrict__ table, const int32_t *__restrict__ q_ids, int32_t len_t, int32_t V) {
	const int j = threadIdx.x;
	if (j < len_t) {
		const int id = q_ids[j];
		if (id >= 0 && id < V) table[(int64_t)id * 16 + j] = 1.0f;
	}
}

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

__global__ __launch_bounds__(256) void vk_topk_keys_kernel(const uint64_t *__restrict__ in, int64_t n, int32_t k,
	uint64_t *__restrict__ out) {
	__shared__ uint64_t keys[VK_TOPK_CHUNK];
	const int64_t base = (int64_t)blockIdx.x * VK_TOPK_CHUNK;
	for (int t = threadIdx.x; t < VK_TOPK_CHUNK; t += blockDim.x) {
		const int64_t g = base + t;
		keys[t] = g < n ? in[g] : 0;
	}
	bitonic_sort_desc_2048(keys);
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

// ---------------------------------------------------------------------------
// Batched relaxed Word Mover's Distance (BASELINE config 4: 256 queries x 1M sentences):
// a GEMM [T x d] . [d x (B * 16)] on MFMA with the row / column minima and the RWMD score as
// epilogue.  MFMA-bound (intensity ~2.5 kFLOP per corpus byte), so the corpus tokens stay in
// registers and the queries stream past them:
//   workgroup = 4 waves; each wave loads TPW token tiles (64 tokens for TPW = 4) ONCE into
//   registers as MFMA B operands; the B query tiles (A operands, one 16-row tile per query) are
//   staged one after the other into a double-buffered LDS slot shared by the 4 waves
//   (global -> registers -> LDS while the previous query's MFMAs run), so every query byte is
//   fetched from L2 once per 256 tokens and every corpus byte from HBM once per batch.
// Requires sentences of one length L = 16 * TPS (the config's shape); other corpora take the
// per-query path.  Scores: scores[q * n_sent + s] = Score::value as in rwmd_rows.
// ---------------------------------------------------------------------------

__device__ __forceinline__ float xor16_f(float x) {
	return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, x), (0x10 << 10) | 0x1f));
}

// DPP row broadcast used by wave reductions: CTRL 0x142 = row_bcast:15 (lane 15 of each row to the next
// row), 0x143 = row_bcast:31; ROWS = row_mask of the rows that receive.  Lanes outside get 0.
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_bcast(float x) {
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROWS, 0xf, false));
}

// partner lane l ^ 32 through ds_bpermute.  (v_permlane32_swap would be cheaper, but the builtin's
// second result did not deliver the upper halves here -- tools/probe/xlane_probe.hip -- so it is not used.)
__device__ __forceinline__ float xor32_f(float x, int lane) {
	return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, x)));
}

template <int NK, bool HALF, int TPS>
__global__ __launch_bounds__(256, 2) void vk_rwmd_batch_kernel(VkRwmdBatchParams p) {
	constexpr int TPW = TPS == 3 ? 3 : 4;          // token tiles per wave
	constexpr int SPW = TPW / TPS;                 // sentences per wave
	extern __shared__ float4 vk_smem4[];
	uint8_t *qbuf = reinterpret_cast<uint8_t *>(vk_smem4);   // 2 x tile_bytes
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int g4 = lane >> 4;
	const int n16 = p.tile_bytes >> 4;             // 16-byte pieces of one query tile
	const int64_t n_chunks = (p.n_tiles + TPW * 4 - 1) / (TPW * 4);

	for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
		const int64_t tile0 = chunk * (TPW * 4) + (int64_t)wv * TPW;
		// ---- corpus tiles of this wave -> registers (read once per batch)
		bf16x8 x[TPW][NK];
#pragma unroll
		for (int tt = 0; tt < TPW; tt++) {
			const int64_t tile = tile0 + tt < p.n_tiles ? tile0 + tt : p.n_tiles;   // one zero tile follows the corpus
			const uint8_t *tp = p.tiles + tile * p.tile_bytes;
#pragma unroll
			for (int t = 0; t < NK; t++) {
				if (HALF && t == NK - 1) x[tt][t] = load_half_block(tp + t * 1024, lane, true);
				else x[tt][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tp + t * 1024 + lane * 16));
			}
		}
		// ---- stage query 0
		__syncthreads();   // previous chunk's readers are done with the LDS slots
		for (int i = threadIdx.x; i < n16; i += 256)
			vk_smem4[i] = *reinterpret_cast<const float4 *>(p.qtiles + i * 16);
		__syncthreads();

		for (int q = 0; q < p.n_queries; q++) {
			const uint8_t *cur = qbuf + (q & 1) * p.tile_bytes;
			float4 *nxt = vk_smem4 + ((q + 1) & 1) * n16;
			// prefetch the next query tile into registers (<= 3 pieces per thread for d <= 384)
			float4 st0 = {0, 0, 0, 0}, st1 = st0, st2 = st0;
			const bool more = q + 1 < p.n_queries;
			if (more) {
				const uint8_t *src = p.qtiles + (int64_t)(q + 1) * p.tile_bytes;
				const int i0 = threadIdx.x, i1 = threadIdx.x + 256, i2 = threadIdx.x + 512;
				if (i0 < n16) st0 = *reinterpret_cast<const float4 *>(src + i0 * 16);
				if (i1 < n16) st1 = *reinterpret_cast<const float4 *>(src + i1 * 16);
				if (i2 < n16) st2 = *reinterpret_cast<const float4 *>(src + i2 * 16);
			}
			// ---- S^T = Q X^T for TPW tiles
			f32x4 acc[TPW];
#pragma unroll
			for (int tt = 0; tt < TPW; tt++) acc[tt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
			// query fragments: all NK ds_read_b128 are issued up front (40 VGPRs for d = 300) so the MFMAs
			// of step t never wait for the LDS latency of step t
			bf16x8 af[NK];
#pragma unroll
			for (int t = 0; t < NK; t++) {
				af[t] = *reinterpret_cast<const bf16x8 *>(cur + t * 1024 + ((HALF && t == NK - 1) ? (lane & 31) : lane) * 16);
				if (HALF && t == NK - 1) {
					const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
					af[t] = lane < 32 ? af[t] : z;
				}
			}
#pragma unroll
			for (int t = 0; t < NK; t++) {
#pragma unroll
				for (int tt = 0; tt < TPW; tt++) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t], x[tt][t], acc[tt], 0, 0, 0);
			}
			if (more) {
				const int i0 = threadIdx.x, i1 = threadIdx.x + 256, i2 = threadIdx.x + 512;
				if (i0 < n16) nxt[i0] = st0;
				if (i1 < n16) nxt[i1] = st1;
				if (i2 < n16) nxt[i2] = st2;
			}
			// ---- epilogue: D = 1 - clip(S); lane holds token (lane & 15) x query columns 4*g4 .. +3.
			// Lane exchanges across the four 16-lane rows go through the LDS crossbar (ds_swizzle /
			// ds_bpermute); the exchanges of all tiles are issued back to back so their latencies overlap.
			// (reciprocals instead of the oracle's divisions: a batch epilogue runs per (query, sentence) and
			// is VALU-bound; the results differ from the per-query kernel by <= 1 ulp, far inside 1e-4)
			const int len_t = p.q_len[q];
			const float inv_t = 1.0f / (float)len_t;
			const float inv_s = 1.0f / (float)(TPS * 16);
			float rm[TPW], cm[SPW][4];
#pragma unroll
			for (int sw = 0; sw < SPW; sw++)
#pragma unroll
				for (int r = 0; r < 4; r++) cm[sw][r] = 3.0f;
#pragma unroll
			for (int tt = 0; tt < TPW; tt++) {
				const f32x4 a4 = acc[tt];
				float dd[4];
#pragma unroll
				for (int r = 0; r < 4; r++) {
					dd[r] = 1.0f - clip01(a4[r]);
					cm[tt / TPS][r] = fminf(cm[tt / TPS][r], dd[r]);
				}
				// padded query columns have S = 0, D = 1: they never lower a minimum
				rm[tt] = fminf(fminf(dd[0], dd[1]), fminf(dd[2], dd[3]));
			}
			float ex[TPW];
#pragma unroll
			for (int tt = 0; tt < TPW; tt++) ex[tt] = xor16_f(rm[tt]);
#pragma unroll
			for (int tt = 0; tt < TPW; tt++) rm[tt] = fminf(rm[tt], ex[tt]);
#pragma unroll
			for (int tt = 0; tt < TPW; tt++) ex[tt] = xor32_f(rm[tt], lane);
#pragma unroll
			for (int tt = 0; tt < TPW; tt++) rm[tt] = fminf(rm[tt], ex[tt]);   // row minimum of token (lane & 15) of tile tt
#pragma unroll
			for (int sw = 0; sw < SPW; sw++) {
				float rsum = 0.0f;
#pragma unroll
				for (int ts = 0; ts < TPS; ts++) rsum += rm[sw * TPS + ts];
				// column minima over the sentence's tokens: reduce over the 16 lanes of the DPP row
#pragma unroll
				for (int r = 0; r < 4; r++) {
					float x = cm[sw][r];
					x = fminf(x, dpp_f<DPP_ROW_SHR1>(x, x));
					x = fminf(x, dpp_f<DPP_ROW_SHR2>(x, x));
					x = fminf(x, dpp_f<DPP_ROW_SHR4>(x, x));
					x = fminf(x, dpp_f<DPP_ROW_SHR8>(x, x));
					cm[sw][r] = x;
				}
				float c0 = 0.0f;
#pragma unroll
				for (int r = 0; r < 4; r++) c0 += (4 * g4 + r < len_t) ? cm[sw][r] : 0.0f;   // valid in lane 15 of each row
				// sum of the four rows' lane 15 -> lane 63 (row_bcast:15 into rows 1, 3; row_bcast:31 into rows 2, 3)
				c0 += dpp_bcast<0x142, 0xa>(c0);
				c0 += dpp_bcast<0x143, 0xc>(c0);
				rsum += dpp_f<DPP_ROW_SHR1>(0.0f, rsum);
				rsum += dpp_f<DPP_ROW_SHR2>(0.0f, rsum);
				rsum += dpp_f<DPP_ROW_SHR4>(0.0f, rsum);
				rsum += dpp_f<DPP_ROW_SHR8>(0.0f, rsum);   // lane 15 of every row: sum over the sentence's tokens
				const float acc0 = inv_t * c0, acc1 = inv_s * rsum;      // nbow and bow/len agree up to rounding
				const float cost = p.symmetric ? fmaxf(0.0f, fmaxf(acc0, acc1)) : acc0;
				const float raw = p.nbow ? 1.0f - cost : ((float)len_t - cost) * inv_t;
				const int64_t sent = (tile0 + sw * TPS) / TPS;
				if (lane == 63 && sent < p.n_sent) {
					const float boost = p.boost ? p.boost[sent] : 1.0f;
					p.scores[(int64_t)q * p.n_sent + sent] = (raw * inv_t) * boost;
				}
			}
			__syncthreads();   // next query tile is in place; this one may be overwritten
		}
	}
}

// ---------------------------------------------------------------------------
// Batched RWMD for 32-token sentences (the shape of BASELINE config 4) on v_mfma_f32_32x32x16_bf16.
// Two inefficiencies of the 16-row kernel above go away: a 10-token query no longer occupies a
// 16-row tile (QPT = 3 queries share the 32 rows of one A tile: 30 / 32 rows used, 10 / 16 before)
// and K is padded to 16, not 32 (d = 300: 19 steps of 16 = 304, not 320).
//   workgroup = 8 waves; a wave keeps its 2 sentences (4 token tiles) in registers as B operands of
//   two MFMA chains: chain 0 takes tokens 0..15 of both sentences (columns 0..15 = sentence 0,
//   16..31 = sentence 1), chain 1 tokens 16..31.  So a lane (n = lane & 31, h = lane >> 5) holds, in
//   acc0[i] and acc1[i], the similarities of ONE query row with tokens n & 15 and 16 + (n & 15) of
//   sentence n >> 4: the maximum over a sentence's tokens is one in-lane max and a reduction over the
//   16 lanes of a DPP row, never across rows.
//   A rows: M = 8 (i >> 2) + 4 h + (i & 3) for accumulator register i of half h (hardware layout of
//   the 32x32 result).  QPT = 3: half h, i < 10 = token i of query 3 qt + h; 10 <= i < 15 = token
//   5 h + i - 10 of query 3 qt + 2.  QPT = 2: half h = query 2 qt + h, i = token.  The host packs the
//   A tiles accordingly (vk_api.cpp pack_query_tiles32).
//   The query tiles stream through a double-buffered LDS slot shared by the 8 waves.
// ---------------------------------------------------------------------------

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8n;
typedef __attribute__((address_space(3))) void *vk_lds_ptr;
typedef __attribute__((address_space(1))) const void *vk_glb_ptr;

__device__ __forceinline__ float row_sum_to_lane15(float x) {
	x += dpp_f<DPP_ROW_SHR1>(0.0f, x);
	x += dpp_f<DPP_ROW_SHR2>(0.0f, x);
	x += dpp_f<DPP_ROW_SHR4>(0.0f, x);
	x += dpp_f<DPP_ROW_SHR8>(0.0f, x);
	return x;
}

// Maxima of similarities that end in clip01() may be taken on the raw bit patterns as signed integers:
// non-negative floats order like their bits, negative floats are negative integers and lose against any
// non-negative one, and a maximum that stays negative is clipped to 0 whichever negative value it is.
// Integer maxima need no canonicalisation of their inputs and fuse with DPP (v_max_i32_dpp).
__device__ __forceinline__ int fbits(float x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
template <int CTRL>
__device__ __forceinline__ int dpp_imax(int x) {
	return imax(x, __builtin_amdgcn_update_dpp((int)0x80000000, x, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ int row_imax_to_lane15(int x) {
	x = dpp_imax<DPP_ROW_SHR1>(x);
	x = dpp_imax<DPP_ROW_SHR2>(x);
	x = dpp_imax<DPP_ROW_SHR4>(x);
	x = dpp_imax<DPP_ROW_SHR8>(x);
	return x;
}
__device__ __forceinline__ float clip01_bits(int x) { return __builtin_amdgcn_fmed3f(__builtin_bit_cast(float, x), 0.0f, 1.0f); }

// Maxima of 16 registers over the 16 lanes of each DPP row, transposed: lane v of the row ends up with the row
// maximum of register v.  Halving exchange: at the step for lane bit b a lane keeps the registers whose index bit
// equals its own lane bit and hands the others to its partner (lane ^ (1 << b)), so the register count halves while
// the lane span doubles: 8 + 4 + 2 + 1 exchanges (47 instructions) instead of 16 four-step DPP reductions (64+).
__device__ __forceinline__ int row_transpose_imax16(const int (&v)[16], int lane) {
	const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
	const int NEG = (int)0x80000000;
	int w[8], x[4], y[2];
#pragma unroll
	for (int k = 0; k < 8; k++) {
		const int keep = b0 ? v[2 * k + 1] : v[2 * k], send = b0 ? v[2 * k] : v[2 * k + 1];
		w[k] = imax(keep, __builtin_amdgcn_update_dpp(NEG, send, 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]: lane ^ 1
	}
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const int keep = b1 ? w[2 * k + 1] : w[2 * k], send = b1 ? w[2 * k] : w[2 * k + 1];
		x[k] = imax(keep, __builtin_amdgcn_update_dpp(NEG, send, 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]: lane ^ 2
	}
#pragma unroll
	for (int k = 0; k < 2; k++) {
		const int keep = b2 ? x[2 * k + 1] : x[2 * k], send = b2 ? x[2 * k] : x[2 * k + 1];
		int t = __builtin_amdgcn_update_dpp(NEG, send, 0x104, 0xf, 0x5, false);                // row_shl:4 into banks 0, 2: lane + 4
		t = __builtin_amdgcn_update_dpp(t, send, 0x114, 0xf, 0xa, false);                      // row_shr:4 into banks 1, 3: lane - 4
		y[k] = imax(keep, t);
	}
	const int keep = b3 ? y[1] : y[0], send = b3 ? y[0] : y[1];
	return imax(keep, __builtin_amdgcn_update_dpp(NEG, send, 0x128, 0xf, 0xf, false));          // row_ror:8: lane ^ 8
}

// S tiles of one query tile against the wave's two sentences: 2 x NK16 MFMAs, A fragments DEPTH steps ahead
template <int NK16>
__device__ __forceinline__ void batch32_mfma(const uint8_t *cur, const bf16x8 (&x)[2][NK16], f32x16 &acc0, f32x16 &acc1) {
	constexpr int DEPTH = 4;                           // A fragments in flight (ds_read_b128 ahead of their MFMAs)
#pragma unroll
	for (int i = 0; i < 16; i++) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
	bf16x8 a[DEPTH];
#pragma unroll
	for (int t = 0; t < DEPTH && t < NK16; t++) a[t] = *reinterpret_cast<const bf16x8 *>(cur + t * 1024);
#pragma unroll
	for (int t = 0; t < NK16; t++) {
		const bf16x8 at = a[t % DEPTH];
#if defined(VK_ABL) && VK_ABL == 3
		if (t > 0) { acc0[t & 15] += (float)at[0]; acc1[t & 15] += (float)x[1][t][0] + (float)x[0][t][0]; continue; }
#endif
		acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8n, at), __builtin_bit_cast(bf16x8n, x[0][t]), acc0, 0, 0, 0);
		acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8n, at), __builtin_bit_cast(bf16x8n, x[1][t]), acc1, 0, 0, 0);
		if (t + DEPTH < NK16) a[t % DEPTH] = *reinterpret_cast<const bf16x8 *>(cur + (t + DEPTH) * 1024);
		__builtin_amdgcn_sched_barrier(0);
	}
}

// RWMD scores of query tile qt for the wave's two sentences from the accumulators.
// D = 1 - clip(S) is monotone in S: reduce S (as integers, see above), convert the reduced values only.
// Rows of absent query tokens are zero: S = 0, clip = 0, no masks.
template <int QPT>
__device__ __forceinline__ void batch32_epilogue(const VkRwmdBatchParams &p, int qt, int64_t sent, int lane, const f32x16 &acc0, const f32x16 &acc1) {
	constexpr int NMAIN = QPT == 3 ? 10 : 16;          // rows of the half's own query
	const int h = lane >> 5;
	const float inv_s = 1.0f / 32.0f;
#if defined(VK_ABL) && VK_ABL == 1
	{
		float z = 0.0f;
#pragma unroll
		for (int i = 0; i < 16; i++) z += acc0[i] + acc1[i];
		if (z == 12345.0f) p.scores[0] = z;
		return;
	}
#endif
	// (a) per token: max over the query's rows (in-lane) -> this lane's two tokens' distances -> sum
	//     over the sentence's 32 tokens = 16 lanes x 2 chains
	int ca0 = fbits(acc0[0]), ca1 = fbits(acc1[0]);
#pragma unroll
	for (int i = 1; i < NMAIN; i++) { ca0 = imax(ca0, fbits(acc0[i])); ca1 = imax(ca1, fbits(acc1[i])); }
	const float ts_main = row_sum_to_lane15((2.0f - clip01_bits(ca0)) - clip01_bits(ca1));
	float ts_third = 0.0f;
	if (QPT == 3) {
		int cb0 = fbits(acc0[10]), cb1 = fbits(acc1[10]);
#pragma unroll
		for (int i = 11; i < 15; i++) { cb0 = imax(cb0, fbits(acc0[i])); cb1 = imax(cb1, fbits(acc1[i])); }
		const int e0 = __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, cb0);
		const int e1 = __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, cb1);
		cb0 = imax(cb0, e0); cb1 = imax(cb1, e1);
		ts_third = row_sum_to_lane15((2.0f - clip01_bits(cb0)) - clip01_bits(cb1));
	}
	// (b) per query row: max over the sentence's tokens -> lane 15 of the DPP row
	const int q_main = qt * QPT + h, q_third = qt * 3 + 2;
	float s_main = 0.0f, s_third = 0.0f;
	{
		int m[16];
#pragma unroll
		for (int i = 0; i < 16; i++) m[i] = imax(fbits(acc0[i]), fbits(acc1[i]));
		const float z = clip01_bits(row_transpose_imax16(m, lane));    // lane v of the row: maximum of query row v over the sentence
		const int v = lane & 15;
		s_main = row_sum_to_lane15(v < NMAIN ? z : 0.0f);
		if (QPT == 3) {
			s_third = row_sum_to_lane15((v >= 10 && v < 15) ? z : 0.0f);
			s_third += xor32_f(s_third, lane);
		}
	}
	// (c) scores (the expressions of vk_rwmd_batch_kernel; sum (1 - x) over len rows = len - sum x)
	if ((lane & 15) == 15 && sent < p.n_sent) {
		const float boost = p.boost ? p.boost[sent] : 1.0f;
		const int len_main = q_main < p.n_queries ? p.q_len[q_main] : 0;
		if (len_main > 0) {
			const float inv_t = p.q_inv_len[q_main];
			const float a0 = inv_t * ((float)len_main - s_main), a1 = inv_s * ts_main;
			const float cost = p.symmetric ? fmaxf(0.0f, fmaxf(a0, a1)) : a0;
			const float raw = p.nbow ? 1.0f - cost : ((float)len_main - cost) * inv_t;
			p.scores[(int64_t)q_main * p.n_sent + sent] = (raw * inv_t) * boost;
		}
		const int len_third = (QPT == 3 && h == 0 && q_third < p.n_queries) ? p.q_len[q_third] : 0;
		if (len_third > 0) {
			const float inv_t = p.q_inv_len[q_third];
			const float a0 = inv_t * ((float)len_third - s_third), a1 = inv_s * ts_third;
			const float cost = p.symmetric ? fmaxf(0.0f, fmaxf(a0, a1)) : a0;
			const float raw = p.nbow ? 1.0f - cost : ((float)len_third - cost) * inv_t;
			p.scores[(int64_t)q_third * p.n_sent + sent] = (raw * inv_t) * boost;
		}
	}
}

template <int NK16, int QPT>
__global__ __launch_bounds__(512) void vk_rwmd_batch32_kernel(VkRwmdBatchParams p) {
	constexpr int QT_BYTES = NK16 * 1024;
	extern __shared__ float4 vk_smem4[];
	const uint8_t *qbuf = reinterpret_cast<const uint8_t *>(vk_smem4);
	const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const int n32 = lane & 31, h = lane >> 5;
	const int64_t n_chunks = ((int64_t)p.n_sent + 15) / 16;
	// The two waves that share a SIMD (w and w + 4 of the workgroup) run the two halves of an interval in
	// opposite order: the "late" wave first finishes the epilogue of the previous tile (VALU) while the
	// other one issues its MFMAs, then they swap.  Barriers would otherwise keep all waves in phase:
	// every matrix core idle during the epilogues, every VALU idle during the MFMAs.
	const bool late = (wv & p.late_mask) != 0;

	for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
		const int64_t sent = chunk * 16 + wv * 2 + (n32 >> 4);   // this lane's sentence
		// ---- the wave's token tiles -> registers (read once per batch)
		bf16x8 x[2][NK16];
#pragma unroll
		for (int m = 0; m < 2; m++) {
			const int64_t tile = sent < p.n_sent ? sent * 2 + m : p.n_tiles;   // one zero tile follows the corpus
			const uint8_t *tp = p.tiles + tile * p.tile_bytes + (n32 & 15) * 16;
#pragma unroll
			for (int t = 0; t < NK16; t++)
				x[m][t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tp + (t >> 1) * 1024 + (2 * (t & 1) + h) * 256));
		}
		__syncthreads();   // previous chunk's readers are done with the LDS slots
#pragma unroll
		for (int b = wv; b < NK16; b += 8)
			__builtin_amdgcn_global_load_lds((vk_glb_ptr)(p.qtiles + b * 1024 + lane * 16), (vk_lds_ptr)(qbuf + b * 1024), 16, 0, 0);
		__syncthreads();

		f32x16 acc0, acc1;
		for (int qt = 0; qt < p.n_qtiles; qt++) {
			const uint8_t *cur = qbuf + (qt & 1) * QT_BYTES + lane * 16;
			// next query tile: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no staging
			// registers -- the kernel sits at the VGPR cap), in flight during this tile's MFMAs and retired by
			// the barrier at the end of the iteration.  The tile after the last one is zero padding.
			{
				const uint8_t *src = p.qtiles + (int64_t)(qt + 1) * QT_BYTES + lane * 16;
				const uint8_t *nxt = qbuf + ((qt + 1) & 1) * QT_BYTES;
#pragma unroll
				for (int b = wv; b < NK16; b += 8)
					__builtin_amdgcn_global_load_lds((vk_glb_ptr)(src + b * 1024), (vk_lds_ptr)(nxt + b * 1024), 16, 0, 0);
			}
			if (!late) {
				batch32_mfma<NK16>(cur, x, acc0, acc1);
				batch32_epilogue<QPT>(p, qt, sent, lane, acc0, acc1);
			} else {
				if (qt > 0) batch32_epilogue<QPT>(p, qt - 1, sent, lane, acc0, acc1);
				batch32_mfma<NK16>(cur, x, acc0, acc1);
			}
			__syncthreads();   // next query tile is in place; this one may be overwritten
		}
		if (late) batch32_epilogue<QPT>(p, p.n_qtiles - 1, sent, lane, acc0, acc1);
	}
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

// ---------------------------------------------------------------------------
// flow of the winners: one wave per winner recomputes the similarity rows with the
// same MFMA sequence as the scoring kernel, then lane 0 runs the sequential DP with
// traceback exactly as the oracle states it (vko_align in oracle/vk_oracle.c):
// candidates zero (LOCAL), diagonal, gap in s (k = 1..), gap in t (k = 1..), replace
// on strictly greater; start cell = first maximum in row-major order.
// Output: mapping[j] = matched sentence token or -1 (InjectiveFlow,
// metric/alignment.h:194-196), edge_sim[j] = S[mapping[j]][j] (metric/alignment.h:335-345).
// ---------------------------------------------------------------------------

#define VK_TB_W 17

// dynamic LDS of vk_flow_kernel for slices of at most max_len tokens (bytes); the carve-up below must match
static inline size_t vk_flow_lds_bytes(int max_len, bool tagged) {
	const size_t rows = (size_t)max_len + 1, srows = (size_t)max_len + 32;
	size_t b = srows * 16 * 4 * (tagged ? 2 : 1);      // S (+ SW)
	b += rows * VK_TB_W * 4;                           // H
	b += (rows + 3) / 4 * 4 * 4 + 32 * 4;              // wsl, wtl
	b += rows * VK_TB_W * 2;                           // dk
	b += rows * VK_TB_W;                               // flags
	return (b + 15) / 16 * 16;
}

__global__ __launch_bounds__(64) void vk_flow_kernel(VkFlowParams p) {
	extern __shared__ float4 vk_smem4[];
	const int rows = p.max_len + 1, srows = p.max_len + 32;
	float *S = reinterpret_cast<float *>(vk_smem4);
	float *SW = p.pos_s ? S + srows * 16 : S;            // tag-weighted copy the DP runs on (else S itself)
	float *H = SW + srows * 16;
	float *wsl = H + rows * VK_TB_W;
	float *wtl = wsl + (rows + 3) / 4 * 4;
	int16_t *dk = reinterpret_cast<int16_t *>(wtl + 32);
	uint8_t *flags = reinterpret_cast<uint8_t *>(dk + rows * VK_TB_W);   // bits 0-1 direction, 2 E extended, 3 F extended

	const int lane = threadIdx.x;
	const int w = blockIdx.x;
	const uint64_t key = p.keys[w];
	if (key == 0) return;   // fewer than k admitted
	const int64_t g = (int64_t)(uint32_t)(key & 0xffffffffu);
	const int t_a = p.sent_start[g], t_b = p.sent_end[g];
	const int len_s = t_b - t_a, len_t = p.len_t;

	// S: unmodified similarities (reported per edge, metric/alignment.h:339); SW: what the DP runs on
	int rowbase;
	if (p.layout == VK_DEV_LAYOUT_STATIC) {
		for (int it = 0; it * 16 < len_s; it++) {
			const int tk = it * 16 + (lane >> 2);
			if (tk < len_s) {
				const int id = p.tok_id[t_a + tk];
				float4 val = *reinterpret_cast<const float4 *>(p.table + (int64_t)id * 16 + (lane & 3) * 4);
				*reinterpret_cast<float4 *>(S + tk * 16 + (lane & 3) * 4) = val;
				if (p.pos_s) {
					const int ps = p.pos_s[t_a + tk], cb = (lane & 3) * 4;
					val.x = tag_weighted(val.x, p.tw[cb + 0], ps, p.tpos[cb + 0], p.tw_keep, p.tw_threshold);
					val.y = tag_weighted(val.y, p.tw[cb + 1], ps, p.tpos[cb + 1], p.tw_keep, p.tw_threshold);
					val.z = tag_weighted(val.z, p.tw[cb + 2], ps, p.tpos[cb + 2], p.tw_keep, p.tw_threshold);
					val.w = tag_weighted(val.w, p.tw[cb + 3], ps, p.tpos[cb + 3], p.tw_keep, p.tw_threshold);
					*reinterpret_cast<float4 *>(SW + tk * 16 + (lane & 3) * 4) = val;
				}
			}
		}
		rowbase = 0;
	} else {
		const int tile0 = t_a >> 4;
		const int ntiles = ((t_b + 15) >> 4) - tile0;
		for (int ti = 0; ti < ntiles; ti++) {
			f32x4 acc = sim_tile_generic(p.qtile, p.tiles + (int64_t)(tile0 + ti) * p.tile_bytes, p.nk32, p.tail, lane, p.prec);
			*reinterpret_cast<f32x4 *>(S + (ti * 16 + (lane & 15)) * 16 + (lane >> 4) * 4) = acc;
			if (p.pos_s) {
				const int ps = p.pos_s[(tile0 + ti) * 16 + (lane & 15)], cb = (lane >> 4) * 4;
#pragma unroll
				for (int r = 0; r < 4; r++) acc[r] = tag_weighted(acc[r], p.tw[cb + r], ps, p.tpos[cb + r], p.tw_keep, p.tw_threshold);
				*reinterpret_cast<f32x4 *>(SW + (ti * 16 + (lane & 15)) * 16 + (lane >> 4) * 4) = acc;
			}
		}
		rowbase = t_a - tile0 * 16;
	}
	// gap tables into LDS (uniform broadcast reads in the candidate loops)
	for (int i = lane; i <= p.max_len; i += 64) wsl[i] = p.ws[i];
	if (lane <= VK_DEV_MAX_QUERY_LEN) wtl[lane] = p.wt[lane];
	__syncthreads();

	const float *Sm = SW + rowbase * 16;      // DP input
	const float *Su = S + rowbase * 16;       // unmodified, for the edges
	const int W = VK_TB_W;
	const bool local = p.locality == VK_DEV_LOCAL, global = p.locality == VK_DEV_GLOBAL;
	const int gap = p.gap_mode;
	const float gs = p.gs, gt = p.gt, open_s = p.open_s, open_t = p.open_t, a_s = p.a_s, a_t = p.a_t;

	// ---- fill: lane l owns query column v = l + 1 (lanes 0..len_t-1, one DPP row).  Per row the
	// zero / diagonal / gap-in-s candidates of all columns are evaluated in parallel; the gap-in-t
	// candidates need the final cells to the left, which become final one column per step and are
	// broadcast with v_readlane (the wave holds ONE sentence, so the column index is wave-uniform).
	// Candidate order and strict-greater replacement are the oracle's (vko_align): zero, diagonal,
	// gap in s with k = 1, 2, .., gap in t with k = 1, 2, ..  In-row candidates arrive with k
	// descending, so among them ">=" keeps the smallest k of the maximum, and the winner replaces
	// the earlier candidates only if strictly greater.
	const int v = lane + 1;
	const bool col = v <= len_t;
	float wrel[16];   // general: w_t(v - p) for source column p < v
#pragma unroll
	for (int pp = 0; pp < 16; pp++) wrel[pp] = (col && pp < v) ? wtl[v - pp] : __builtin_inff();

	float hprev = 0.0f, eprev = VK_NEG_INF;
	if (global && col) hprev = gap == 0 ? -(gt * (float)v) : gap == 1 ? -(a_t + gt * (float)v) : -wtl[v];
	if (col) H[v] = hprev;
	for (int u = 1; u <= len_s; u++) {
		float bprev = 0.0f, bcur = 0.0f;
		if (global) {
			bprev = u == 1 ? 0.0f : (gap == 0 ? -(gs * (float)(u - 1)) : gap == 1 ? -(a_s + gs * (float)(u - 1)) : -wsl[u - 1]);
			bcur = gap == 0 ? -(gs * (float)u) : gap == 1 ? -(a_s + gs * (float)u) : -wsl[u];
		}
		const float sv = Sm[(u - 1) * 16 + (col ? v - 1 : 0)];
		const float diag = dpp_f<DPP_ROW_SHR1>(bprev, hprev);
		float best, e = VK_NEG_INF;
		uint8_t d, ee = 0, fe = 0;
		int16_t kk = 0;
		float c = diag + sv;
		if (local) { best = 0.0f; d = 0; if (c > best) { best = c; d = 1; } }
		else { best = c; d = 1; }
		if (gap == 0) {
			c = hprev - gs;
			if (c > best) { best = c; d = 2; kk = 1; }
		} else if (gap == 1) {
			e = hprev - open_s;
			c = eprev - gs;
			if (c > e) { e = c; ee = 1; }
			if (e > best) { best = e; d = 2; }
		} else {
			for (int k = 1; k <= u; k++) {
				c = H[(u - k) * W + (col ? v : 1)] - wsl[k];
				if (c > best) { best = c; d = 2; kk = (int16_t)k; }
			}
		}
		// in-row candidates
		float left_best = VK_NEG_INF, f = VK_NEG_INF, fin = best, ffin = VK_NEG_INF;
		int16_t left_k = 0;
#pragma unroll
		for (int pp = 0; pp < 16; pp++) {
			if (pp < len_t) {
				const float sp = pp == 0 ? bcur : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fin), pp - 1));
				const float fp = pp == 0 ? VK_NEG_INF : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ffin), pp - 1));
				if (gap == 2) {
					const float cc = sp - wrel[pp];
					if (v > pp && cc >= left_best) { left_best = cc; left_k = (int16_t)(v - pp); }
				} else if (v == pp + 1) {
					if (gap == 0) { left_best = sp - gt; left_k = 1; }
					else {
						f = sp - open_t;
						const float c2 = fp - gt;
						if (c2 > f) { f = c2; fe = 1; }
					}
				}
				if (v == pp + 1) {   // all sources of this column are in: finalise it
					if (gap == 1) { if (f > best) { best = f; d = 3; } ffin = f; }
					else if (left_best > best) { best = left_best; d = 3; kk = left_k; }
					fin = best;
				}
			}
		}
		if (col) {
			H[u * W + v] = best;
			dk[u * W + v] = kk;
			flags[u * W + v] = (uint8_t)(d | (ee << 2) | (fe << 3));
		}
		hprev = best;
		eprev = e;
	}

	// ---- start cell: first maximum in row-major order (u outer, v inner), borders (0) first
	float bv = 0.0f;
	int bu = 0;
	if (col && !global) {
		for (int uu = 1; uu <= len_s; uu++) {
			if (!local && !(uu == len_s || v == len_t)) continue;
			const float hv = H[uu * W + v];
			if (hv > bv) { bv = hv; bu = uu; }
		}
	}
	wave_lds_fence();
	int u = len_s, vq = len_t;
	float raw;
	if (global) {
		raw = H[len_s * W + len_t];
	} else {
		raw = 0.0f; u = 0; vq = 0;
		for (int j = 0; j < len_t; j++) {
			const float vj = __shfl(bv, j, 64);
			const int uj = __shfl(bu, j, 64);
			if (vj > raw || (vj == raw && vj > 0.0f && uj < u)) { raw = vj; u = uj; vq = j + 1; }
		}
	}
	if (lane != 0) return;
	int v2 = vq;
	int16_t *mp = p.mapping + (int64_t)w * 16;
	float *es = p.edge_sim + (int64_t)w * 16;
	for (int j = 0; j < 16; j++) { mp[j] = -1; es[j] = 0.0f; }
	int state = 0;
	while (u > 0 && v2 > 0) {
		const int idx = u * W + v2;
		const uint8_t fl = flags[idx];
		if (gap == 1 && state == 1) { if (!(fl & 4)) state = 0; u--; continue; }
		if (gap == 1 && state == 2) { if (!(fl & 8)) state = 0; v2--; continue; }
		const uint8_t d = fl & 3;
		if (d == 0) break;
		if (d == 1) { mp[v2 - 1] = (int16_t)(u - 1); es[v2 - 1] = Su[(u - 1) * 16 + v2 - 1]; u--; v2--; }
		else if (gap == 1) state = (d == 2) ? 1 : 2;
		else if (d == 2) u -= dk[idx];
		else v2 -= dk[idx];
	}
	p.raw_out[w] = raw;
}

// ---------------------------------------------------------------------------
// Queries of 17 .. 64 tokens: one wave per slice, lane = query column (the fill of vk_flow_kernel
// widened to the whole wave).  SCORE mode walks all slices and writes Score::value / raw like
// vk_score_kernel; FLOW mode retraces the k winners.  The similarity rows are produced 16 tokens at
// a time (one MFMA tile per 16 query rows) into a small LDS strip and consumed by the row-serial DP
// at once, so LDS holds only the column history (general gaps) and, in FLOW mode, the traceback.
// Candidate order, strict-greater replacement and start-cell rule: as vk_flow_kernel / the oracle.
// Roughly 10 us of issue time per (32-token slice, 32-token query): a fallback that keeps long
// queries on the device, not a roofline kernel.
// ---------------------------------------------------------------------------

static inline size_t vk_wide_lds_bytes(int max_len, int nq, int gap_mode, bool tagged, bool flow) {
	const size_t LQ = (size_t)nq * 16, W = LQ + 1, rows = (size_t)max_len + 1;
	size_t fl = 16 * LQ * (tagged ? 2 : 1);            // Sx (+ SWx)
	fl += (rows + 3) / 4 * 4 + LQ + 4;                 // wsl, wtl
	fl += 64 + 64;                                     // twl, tposl
	if (gap_mode == 2) fl += rows * W;                 // H
	size_t b = fl * 4;
	if (flow) b += 64 * 2 + rows * W * 2 + rows * W;   // mapl, dk, flags
	return (b + 15) / 16 * 16;
}

__device__ __forceinline__ float wave_min64(float m) {
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) m = fminf(m, __shfl_xor(m, off, 64));
	return m;
}

template <bool FLOW>
__global__ __launch_bounds__(64) void vk_wide_kernel(VkWideParams p) {
	extern __shared__ float4 vk_smem4[];
	const int lane = threadIdx.x;
	const int LQ = p.nq * 16, W = LQ + 1, rows = p.max_len + 1;
	float *Sx = reinterpret_cast<float *>(vk_smem4);       // [16][LQ] similarities of the current 16 tokens
	float *SWx = p.pos_s ? Sx + 16 * LQ : Sx;              // tag-weighted copy the DP runs on
	float *wsl = SWx + 16 * LQ;
	float *wtl = wsl + (rows + 3) / 4 * 4;
	float *twl = wtl + LQ + 4;
	int *tposl = reinterpret_cast<int *>(twl + 64);
	float *H = reinterpret_cast<float *>(tposl + 64);      // general gaps: H[u][v], row stride W
	float *after = p.gap_mode == 2 ? H + rows * W : H;
	int16_t *mapl = reinterpret_cast<int16_t *>(after);    // FLOW: mapping of the winner
	int16_t *dk = mapl + 64;
	uint8_t *flags = reinterpret_cast<uint8_t *>(dk + rows * W);

	for (int i = lane; i <= p.max_len; i += 64) wsl[i] = p.ws[i];
	if (lane <= LQ) wtl[lane] = p.wt[lane];
	if (lane == 0) wtl[LQ] = p.wt[LQ <= 64 ? LQ : 64];
	twl[lane] = p.tw[lane]; tposl[lane] = p.tpos[lane];
	wave_lds_fence();

	const int len_t = p.len_t;
	const int v = lane + 1;
	const bool col = v <= len_t;
	const bool local = p.locality == VK_DEV_LOCAL, global = p.locality == VK_DEV_GLOBAL;
	const int gap = p.gap_mode;
	const float gs = p.gs, gt = p.gt, open_s = p.open_s, open_t = p.open_t, a_s = p.a_s, a_t = p.a_t;
	const bool is_static = p.layout == VK_DEV_LAYOUT_STATIC;

	const int64_t n_items = FLOW ? (int64_t)gridDim.x : (int64_t)p.n_sent;
	for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
		int64_t g = item;
		if (FLOW) {
			const uint64_t key = p.keys[item];
			if (key == 0) return;   // fewer than k admitted
			g = (int64_t)(uint32_t)(key & 0xffffffffu);
		}
		const int t_a = p.sent_start[g], t_b = p.sent_end[g];
		const int len_s = t_b - t_a;
		if (len_s < 1) {
			if (!FLOW && lane == 0) { p.scores[g] = VK_NEG_INF; p.raw[g] = VK_NEG_INF; }
			continue;
		}
		// similarities of tokens base .. base + 15 (contextual: one tile, 16-aligned; static: gather)
		auto fill = [&](int base) {
			if (is_static) {
				for (int r = 0; r < 16; r++) {
					const int tok = base + r;
					if (tok < t_b && lane < LQ) {
						const int id = p.tok_id[tok];
						const float sv = p.table[(int64_t)(lane >> 4) * p.table_stride + (int64_t)id * 16 + (lane & 15)];
						Sx[r * LQ + lane] = sv;
						if (p.pos_s) SWx[r * LQ + lane] = tag_weighted(sv, twl[lane], p.pos_s[tok], tposl[lane], p.tw_keep, p.tw_threshold);
					}
				}
			} else {
				const uint8_t *tp = p.tiles + (int64_t)(base >> 4) * p.tile_bytes;
				for (int qt = 0; qt < p.nq; qt++) {
					f32x4 acc = sim_tile_generic(p.qtile + (int64_t)qt * p.tile_bytes, tp, p.nk32, p.tail, lane, p.prec);
					const int c0 = qt * 16 + (lane >> 4) * 4;
					*reinterpret_cast<f32x4 *>(Sx + (lane & 15) * LQ + c0) = acc;
					if (p.pos_s) {
						const int ps = p.pos_s[base + (lane & 15)];
#pragma unroll
						for (int r = 0; r < 4; r++) acc[r] = tag_weighted(acc[r], twl[c0 + r], ps, tposl[c0 + r], p.tw_keep, p.tw_threshold);
						*reinterpret_cast<f32x4 *>(SWx + (lane & 15) * LQ + c0) = acc;
					}
				}
			}
		};
		const int base0 = is_static ? t_a : (t_a >> 4) * 16;

		float raw;
		int u_start = 0, v_start = 0;
		if (gap == 4) {
			// ---- relaxed word mover's distance (rwmd_rows of vk_score_kernel over <= 64 columns)
			const bool nbow = p.rwmd_normalize_bow != 0;
			const float w_t = nbow ? 1.0f / (float)len_t : 1.0f, w_s = nbow ? 1.0f / (float)len_s : 1.0f;
			float colmin = 3.402823466e+38F, acc1 = 0.0f;
			for (int base = base0; base < t_b; base += 16) {
				fill(base);
				wave_lds_fence();
				const int r0 = t_a > base ? t_a - base : 0, r1 = t_b - base < 16 ? t_b - base : 16;
				for (int r = r0; r < r1; r++) {
					const float dist = fmaxf(1.0f - Sx[r * LQ + (col ? v - 1 : 0)], 0.0f);
					colmin = fminf(colmin, dist);
					acc1 += w_s * wave_min64(col ? dist : 3.402823466e+38F);
				}
				wave_lds_fence();
			}
			const float x = col ? w_t * colmin : 0.0f;
			float acc0 = 0.0f;
			for (int j = 0; j < len_t; j++) {
				const float xj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), j));
				acc0 = j == 0 ? xj : acc0 + xj;
			}
			if (!nbow) { acc0 = acc0 / (float)len_t; acc1 = acc1 / (float)len_s; }
			float cost = 0.0f;
			if (p.rwmd_symmetric) { if (acc0 > cost) cost = acc0; if (acc1 > cost) cost = acc1; }
			else cost = acc0;
			const float max_cost = nbow ? 1.0f : (float)len_t;
			raw = (max_cost - cost) / max_cost;
		} else {
			// ---- alignment: fill, lane = column
			float hprev = 0.0f, eprev = VK_NEG_INF;
			if (global && col) hprev = gap == 0 ? -(gt * (float)v) : gap == 1 ? -(a_t + gt * (float)v) : -wtl[v];
			if (gap == 2 && col) H[v] = hprev;
			float bv = 0.0f;
			int bu = 0, u = 0;
			for (int base = base0; base < t_b; base += 16) {
				fill(base);
				wave_lds_fence();
				const int r0 = t_a > base ? t_a - base : 0, r1 = t_b - base < 16 ? t_b - base : 16;
				for (int r = r0; r < r1; r++) {
					u++;
					float bprev = 0.0f, bcur = 0.0f;
					if (global) {
						bprev = u == 1 ? 0.0f : (gap == 0 ? -(gs * (float)(u - 1)) : gap == 1 ? -(a_s + gs * (float)(u - 1)) : -wsl[u - 1]);
						bcur = gap == 0 ? -(gs * (float)u) : gap == 1 ? -(a_s + gs * (float)u) : -wsl[u];
					}
					const float sv = SWx[r * LQ + (col ? v - 1 : 0)];
					const float up = __shfl_up(hprev, 1, 64);
					const float diag = lane == 0 ? bprev : up;
					float best, e = VK_NEG_INF;
					uint8_t d, ee = 0, fe = 0;
					int16_t kk = 0;
					float c = diag + sv;
					if (local) { best = 0.0f; d = 0; if (c > best) { best = c; d = 1; } }
					else { best = c; d = 1; }
					if (gap == 0) {
						c = hprev - gs;
						if (c > best) { best = c; d = 2; kk = 1; }
					} else if (gap == 1) {
						e = hprev - open_s;
						c = eprev - gs;
						if (c > e) { e = c; ee = 1; }
						if (e > best) { best = e; d = 2; }
					} else {
						for (int k = 1; k <= u; k++) {
							c = H[(u - k) * W + (col ? v : 1)] - wsl[k];
							if (c > best) { best = c; d = 2; kk = (int16_t)k; }
						}
					}
					// in-row candidates: columns become final left to right
					float left_best = VK_NEG_INF, f = VK_NEG_INF, fin = best, ffin = VK_NEG_INF;
					int16_t left_k = 0;
					for (int pp = 0; pp < len_t; pp++) {
						const float sp = pp == 0 ? bcur : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fin), pp - 1));
						const float fp = pp == 0 ? VK_NEG_INF : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ffin), pp - 1));
						if (gap == 2) {
							if (col && v > pp) {
								const float cc = sp - wtl[v - pp];
								if (cc >= left_best) { left_best = cc; left_k = (int16_t)(v - pp); }
							}
						} else if (v == pp + 1) {
							if (gap == 0) { left_best = sp - gt; left_k = 1; }
							else {
								f = sp - open_t;
								const float c2 = fp - gt;
								if (c2 > f) { f = c2; fe = 1; }
							}
						}
						if (v == pp + 1) {
							if (gap == 1) { if (f > best) { best = f; d = 3; } ffin = f; }
							else if (left_best > best) { best = left_best; d = 3; kk = left_k; }
							fin = best;
						}
					}
					if (col) {
						if (gap == 2) H[u * W + v] = best;
						if (FLOW) {
							dk[u * W + v] = kk;
							flags[u * W + v] = (uint8_t)(d | (ee << 2) | (fe << 3));
						}
						// start cell: first maximum in row-major order; per column the first row wins (strict >)
						if (!global && (local || u == len_s || v == len_t) && best > bv) { bv = best; bu = u; }
					}
					hprev = best;
					eprev = e;
				}
				wave_lds_fence();
			}
			if (global) {
				raw = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hprev), len_t - 1));
				u_start = len_s; v_start = len_t;
			} else {
				raw = 0.0f;
				for (int j = 0; j < len_t; j++) {
					const float vj = __shfl(bv, j, 64);
					const int uj = __shfl(bu, j, 64);
					if (vj > raw || (vj == raw && vj > 0.0f && uj < u_start)) { raw = vj; u_start = uj; v_start = j + 1; }
				}
			}
		}

		if (!FLOW) {
			if (lane == 0) {
				const float boost = p.boost ? p.boost[g] : 1.0f;
				p.scores[g] = (raw / p.ref_total) * boost;
				p.raw[g] = raw;
			}
			continue;
		}
		// ---- FLOW: traceback by lane 0, then the edge similarities from a second sweep over the tiles
		mapl[lane] = -1;
		wave_lds_fence();
		if (lane == 0 && gap != 4) {
			int u = u_start, v2 = v_start, state = 0;
			while (u > 0 && v2 > 0) {
				const int idx = u * W + v2;
				const uint8_t fl = flags[idx];
				if (gap == 1 && state == 1) { if (!(fl & 4)) state = 0; u--; continue; }
				if (gap == 1 && state == 2) { if (!(fl & 8)) state = 0; v2--; continue; }
				const uint8_t d = fl & 3;
				if (d == 0) break;
				if (d == 1) { mapl[v2 - 1] = (int16_t)(u - 1); u--; v2--; }
				else if (gap == 1) state = (d == 2) ? 1 : 2;
				else if (d == 2) u -= dk[idx];
				else v2 -= dk[idx];
			}
		}
		wave_lds_fence();
		const int mine = mapl[lane];
		float es = 0.0f;
		for (int base = base0; base < t_b; base += 16) {
			fill(base);
			wave_lds_fence();
			const int row = t_a + mine - base;
			if (mine >= 0 && row >= 0 && row < 16) es = Sx[row * LQ + lane];
			wave_lds_fence();
		}
		p.mapping[item * 64 + lane] = (int16_t)mine;
		p.edge_sim[item * 64 + lane] = es;
		if (lane == 0) p.raw_out[item] = raw;
	}
}

// ---------------------------------------------------------------------------
// Word Rotator's Distance / full WMD, stage 2: exact EMD for the candidate slices.
// One wave per candidate recomputes the similarity rows (same MFMA sequence as the scoring kernel) and
// solves the transportation problem (n <= 16 query tokens = supplies, m <= 64 slice tokens = demands) by
// successive shortest paths with potentials in double precision -- the algorithm of the oracle's vko_emd
// (oracle/vk_oracle.c), which stands in for pyemd's emd_hat_gd_metric<double>
// (vectorian/core/cpp/alignment/transport.h:70,125-126) -- with the Dijkstra step spread over the wave:
//   lane i owns demand i (its distance, potential, predecessor, remaining mass; column i of the costs and
//   flows in LDS); the supplies live in small LDS arrays read uniformly.
//   All supplies with remaining mass are sources and are relaxed together.  Demands are never settled
//   one by one: the next supply to settle is the minimum over (demand i, supply b with flow b -> i) of
//   dist[i] + reduced cost(i -> b), one in-lane loop over b and ONE wave reduction; it is final because
//   any shorter path would pass through another unsettled supply first.  The search ends when the
//   nearest demand with remaining mass is at most that far.
// The optimal cost is unique, so the score equals the oracle's up to the rounding of the final sums
// (the path taken among equal-cost alternatives may differ).  A serial one-lane version of the same
// solver took 8 - 20 ms per round of candidates; this one ~0.1 ms.
// ---------------------------------------------------------------------------

#define VK_WRD_N VK_DEV_MAX_QUERY_LEN
#define VK_WRD_M VK_DEV_MAX_SENT_LEN

// minimum of x over the wave and a lane holding it (the lowest such lane)
__device__ __forceinline__ double wave_argmin_f64(double x, int lane, int &at) {
	double m = x;
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) m = fmin(m, __shfl_xor(m, off, 64));
	const unsigned long long hit = __ballot(x == m);
	at = hit ? __builtin_ctzll(hit) : 0;
	return m;
}

__global__ __launch_bounds__(64) void vk_wrd_exact_kernel(VkWrdParams p) {
	__shared__ __attribute__((aligned(16))) float S[(VK_DEV_MAX_SENT_LEN + 32) * 16];
	__shared__ double Cm[VK_WRD_N * 64];     // Cm[j * 64 + i]: cost supply j -> demand i
	__shared__ double fl[VK_WRD_N * 64];     // flow
	__shared__ double sup[VK_WRD_N], pot_s[VK_WRD_N], dist_s[VK_WRD_N];
	__shared__ int pred_s[VK_WRD_N], settled[VK_WRD_N];

	const int lane = threadIdx.x;
	const int w = blockIdx.x;
	const uint64_t key = p.keys[w];
	if (key == 0) return;
	const int64_t g = (int64_t)(uint32_t)(key & 0xffffffffu);
	const int t_a = p.sent_start[g], t_b = p.sent_end[g];
	const int m = t_b - t_a, n = p.len_t;

	int rowbase;
	if (p.layout == VK_DEV_LAYOUT_STATIC) {
		for (int it = 0; it * 16 < m; it++) {
			const int tk = it * 16 + (lane >> 2);
			if (tk < m) {
				const int id = p.tok_id[t_a + tk];
				*reinterpret_cast<float4 *>(S + tk * 16 + (lane & 3) * 4) =
					*reinterpret_cast<const float4 *>(p.table + (int64_t)id * 16 + (lane & 3) * 4);
			}
		}
		rowbase = 0;
	} else {
		const int tile0 = t_a >> 4;
		const int ntiles = ((t_b + 15) >> 4) - tile0;
		for (int ti = 0; ti < ntiles; ti++) {
			const f32x4 acc = sim_tile_generic(p.qtile, p.tiles + (int64_t)(tile0 + ti) * p.tile_bytes, p.nk32, p.tail, lane, p.prec);
			*reinterpret_cast<f32x4 *>(S + (ti * 16 + (lane & 15)) * 16 + (lane >> 4) * 4) = acc;
		}
		rowbase = t_a - tile0 * 16;
	}
	wave_lds_fence();
	const float *Sm = S + rowbase * 16;
	const bool has = lane < m;
	const double EPS = 1e-13, INF = __builtin_inf();

	// masses (wrd.h:99-102) and costs (:104-109)
	double dem = 0.0;
	if (p.mass_mode == 0) {
		const bool by_id = p.layout == VK_DEV_LAYOUT_STATIC;       // static layout: magnitudes of the vocabulary entries
		float sum_s = 0.0f;
		for (int i = 0; i < m; i++) sum_s += by_id ? p.mag[p.tok_id[t_a + i]] : p.mag[t_a + i];       // in position order, as upstream
		const float mine = has ? (by_id ? p.mag[p.tok_id[t_a + lane]] : p.mag[t_a + lane]) : 0.0f;
		if (has) dem = (double)(p.raw_masses ? mine : mine / sum_s);
		if (lane < VK_WRD_N) sup[lane] = lane < n ? (double)p.qmass[lane] : 0.0;
	} else {
		// bags of words over positions: 1 per token (bow), or 1/len (nbow, bow.h:262-270)
		const float wt = p.mass_mode == 1 ? 1.0f / (float)n : 1.0f;
		const float wsn = p.mass_mode == 1 ? 1.0f / (float)m : 1.0f;
		if (has) dem = (double)wsn;
		if (lane < VK_WRD_N) sup[lane] = lane < n ? (double)wt : 0.0;
	}
	if (lane < VK_WRD_N) pot_s[lane] = 0.0;
	for (int j = 0; j < n; j++) {
		float d = has ? 1.0f - Sm[lane * 16 + j] : 0.0f;
		if (!(d > 0.0f)) d = 0.0f;
		Cm[j * 64 + lane] = (double)d;
		fl[j * 64 + lane] = 0.0;
	}
	double pot_d = 0.0;
	wave_lds_fence();

	for (int iter = 0; iter < 4000; iter++) {
		// ---- sources: every supply with remaining mass; relax them all
		bool any_sup = false;
		double dist_d = INF;
		int pred_d = -1;
		for (int j = 0; j < n; j++) {
			const bool src = sup[j] > EPS;
			any_sup |= src;
			if (lane == 0) { settled[j] = src ? 1 : 0; dist_s[j] = src ? 0.0 : INF; pred_s[j] = -1; }
			if (src) {
				double rc = Cm[j * 64 + lane] + pot_s[j] - pot_d;
				if (rc < 0) rc = 0;
				if (rc < dist_d) { dist_d = rc; pred_d = j; }
			}
		}
		if (!has) dist_d = INF;
		const bool any_dem = __ballot(has && dem > EPS) != 0;
		if (!any_sup || !any_dem) break;
		wave_lds_fence();

		int target = -1;
		double dt = INF;
		for (int round = 0; round <= n; round++) {
			int fd_lane, c_lane;
			const double fd = wave_argmin_f64((has && dem > EPS) ? dist_d : INF, lane, fd_lane);
			double best = INF;
			int bb = -1;
			if (dist_d < INF) {
				for (int b = 0; b < n; b++) {
					if (settled[b]) continue;
					if (!(fl[b * 64 + lane] > EPS)) continue;
					double rc = pot_d - pot_s[b] - Cm[b * 64 + lane];
					if (rc < 0) rc = 0;
					const double cand = dist_d + rc;
					if (cand < best) { best = cand; bb = b; }
				}
			}
			const double cmin = wave_argmin_f64(best, lane, c_lane);
			if (fd <= cmin) {
				if (fd < INF) { target = fd_lane; dt = fd; }
				break;
			}
			const int cb = __shfl(bb, c_lane, 64);
			if (lane == 0) { settled[cb] = 1; dist_s[cb] = cmin; pred_s[cb] = c_lane; }
			wave_lds_fence();
			double rc = Cm[cb * 64 + lane] + pot_s[cb] - pot_d;
			if (rc < 0) rc = 0;
			const double nd = cmin + rc;
			if (has && nd < dist_d) { dist_d = nd; pred_d = cb; }
		}
		if (target < 0) break;

		// ---- potentials: pot += min(dist, dt)
		pot_d += dist_d < dt ? dist_d : dt;
		if (lane < n) pot_s[lane] += (settled[lane] && dist_s[lane] < dt) ? dist_s[lane] : dt;
		wave_lds_fence();

		// ---- bottleneck along target <- supply <- demand <- ... <- source, then augment
		double delta = __shfl(dem, target, 64);
		int x = target;
		for (int hop = 0; hop <= n; hop++) {
			const int a = __shfl(pred_d, x, 64);
			const int ps = pred_s[a];
			if (ps < 0) { delta = fmin(delta, sup[a]); break; }
			delta = fmin(delta, fl[a * 64 + ps]);
			x = ps;
		}
		if (lane == target) dem -= delta;
		x = target;
		for (int hop = 0; hop <= n; hop++) {
			const int a = __shfl(pred_d, x, 64);
			const int ps = pred_s[a];
			if (lane == 0) {
				fl[a * 64 + x] += delta;
				if (ps < 0) sup[a] -= delta;
				else fl[a * 64 + ps] -= delta;
			}
			if (ps < 0) break;
			x = ps;
		}
		wave_lds_fence();
	}

	// score = sum((1 - D) * G) / sum(G) (wrd.h:139), G as float
	double num = 0.0, den = 0.0;
	if (has)
		for (int j = 0; j < n; j++) {
			const float gq = (float)fl[j * 64 + lane];
			num += (double)((1.0f - (float)Cm[j * 64 + lane]) * gq);
			den += (double)gq;
		}
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) { num += __shfl_xor(num, off, 64); den += __shfl_xor(den, off, 64); }
	if (lane == 0) {
		const float raw = den > 0.0 ? (float)(num / den) : 0.0f;
		const float boost = p.boost ? p.boost[g] : 1.0f;
		p.raw_out[w] = raw;
		p.val_out[w] = (raw / (float)n) * boost;
	}
	if (p.plan_out)
		for (int j = 0; j < VK_WRD_N; j++)
			p.plan_out[((int64_t)w * VK_WRD_N + j) * 64 + lane] = (has && j < n) ? (float)fl[j * 64 + lane] : 0.0f;
}

// similarity rows of the winners of a transport query, for the host to state their flows
__global__ __launch_bounds__(64) void vk_rows_kernel(VkWrdParams p) {
	const int lane = threadIdx.x;
	const int w = blockIdx.x;
	float *out = p.rows_out + (int64_t)w * 64 * 16;
	for (int i = lane; i < 64 * 16; i += 64) out[i] = 0.0f;
	const uint64_t key = p.keys[w];
	if (key == 0) return;
	const int64_t g = (int64_t)(uint32_t)(key & 0xffffffffu);
	const int t_a = p.sent_start[g], t_b = p.sent_end[g];
	const int m = t_b - t_a;
	if (m < 1 || m > 64) return;
	__builtin_amdgcn_s_waitcnt(0);
	if (p.layout == VK_DEV_LAYOUT_STATIC) {
		for (int it = 0; it * 16 < m; it++) {
			const int tk = it * 16 + (lane >> 2);
			if (tk < m) {
				const int id = p.tok_id[t_a + tk];
				*reinterpret_cast<float4 *>(out + tk * 16 + (lane & 3) * 4) =
					*reinterpret_cast<const float4 *>(p.table + (int64_t)id * 16 + (lane & 3) * 4);
			}
		}
	} else {
		const int tile0 = t_a >> 4;
		const int ntiles = ((t_b + 15) >> 4) - tile0;
		for (int ti = 0; ti < ntiles; ti++) {
			const f32x4 acc = sim_tile_generic(p.qtile, p.tiles + (int64_t)(tile0 + ti) * p.tile_bytes, p.nk32, p.tail, lane, p.prec);
			const int row = (tile0 + ti) * 16 + (lane & 15) - t_a;      // token of this lane relative to the slice
			if (row >= 0 && row < m) *reinterpret_cast<f32x4 *>(out + row * 16 + (lane >> 4) * 4) = acc;
		}
	}
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
// over the largest ref = T.  Stage 2 (vk_api.cpp) retraces the candidates with the largest bounds.
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

// ---------------------------------------------------------------------------
// host-callable launchers (used by vk_api.cpp; keep all <<< >>> in this file)
// ---------------------------------------------------------------------------

extern "C" hipError_t vk_launch_pack(const void *in, int32_t dtype_bf16, int64_t n_rows, int32_t d, int32_t d_pad, int64_t row0,
	uint8_t *tiles, float *mag_out, int32_t normalize, int32_t prec, hipStream_t stream) {
	if (n_rows <= 0) return hipSuccess;
	const unsigned grid = (unsigned)((n_rows + 3) / 4);
	if (dtype_bf16)
		vk_pack_rows_kernel<uint16_t><<<grid, 256, 0, stream>>>((const uint16_t *)in, n_rows, d, d_pad, row0, tiles, mag_out, normalize, prec);
	else
		vk_pack_rows_kernel<float><<<grid, 256, 0, stream>>>((const float *)in, n_rows, d, d_pad, row0, tiles, mag_out, normalize, prec);
	return hipGetLastError();
}

extern "C" hipError_t vk_launch_table(const uint8_t *etiles, const uint8_t *qtile, int32_t n_tiles, int32_t nk32, int32_t tail,
	int32_t tile_bytes, float *table, const int32_t *q_ids, int32_t len_t, int32_t V, int32_t prec, hipStream_t stream) {
	vk_table_kernel<<<(n_tiles + 3) / 4, 256, 0, stream>>>(etiles, qtile, n_tiles, nk32, tail, tile_bytes, table, prec);
	if (q_ids) vk_table_fix_kernel<<<1, 64, 0, stream>>>(table, q_ids, len_t, V);
	return hipGetLastError();
}

// grid = every CU filled to the kernel's real residency (VGPR / LDS bound), not more: the waves
// walk the groups with a grid stride, so a second, partially filled round of workgroups would only
// add a tail.  VK_BLOCKS_PER_CU overrides (experiments).
template <typename K>
static hipError_t launch_sized(K kernel, const VkScoreParams &p, int want_blocks, size_t smem, hipStream_t stream) {
	int occ = 0;
	const int threads = p.group_list ? 64 : 256;
	hipError_t e;
	if (smem > 64 * 1024) {
		e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
		if (e != hipSuccess) return e;
	}
	e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, threads, smem);
	if (e != hipSuccess) return e;
	if (occ < 1) occ = 1;
	// measured on MI355X (1M x 32 x 300-d): 3 workgroups (12 waves) per CU stream HBM fastest --
	// 2.90 ms vs 3.51 ms at 5 per CU for the linear-gap kernel, 2.95 ms at 4; more concurrent streams cost bandwidth
	if (occ > 3 && p.layout != VK_DEV_LAYOUT_STATIC) occ = 3;   // the static layout is DP-bound, not a stream: keep full residency
	// 768-d rows: a wave already keeps 24 KiB of loads in flight per tile; one workgroup per CU measured fastest
	// (ragged 8..64 tokens, 400 k sentences: 3.37 ms at 1, 3.45 ms at 2 per CU)
	if (p.nk32 >= 24 && p.prec == 0 && p.layout != VK_DEV_LAYOUT_STATIC) occ = 1;
	static const char *ov = getenv("VK_BLOCKS_PER_CU");
	if (ov && atoi(ov) > 0) occ = atoi(ov);
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	const int grid = want_blocks < cus * occ ? want_blocks : cus * occ;
	kernel<<<grid, threads, smem, stream>>>(p);
	return hipGetLastError();
}

template <int MODE, int NK32, bool TAIL, int GAP>
static hipError_t launch_score_lt(const VkScoreParams &p, int grid, size_t smem, hipStream_t stream) {
	const int lt = p.len_t <= 4 ? 4 : p.len_t <= 8 ? 8 : p.len_t <= 12 ? 12 : 16;
	switch (lt) {
	case 4: return launch_sized(vk_score_kernel<MODE, NK32, TAIL, GAP, 4>, p, grid, smem, stream);
	case 8: return launch_sized(vk_score_kernel<MODE, NK32, TAIL, GAP, 8>, p, grid, smem, stream);
	case 12: return launch_sized(vk_score_kernel<MODE, NK32, TAIL, GAP, 12>, p, grid, smem, stream);
	default: return launch_sized(vk_score_kernel<MODE, NK32, TAIL, GAP, 16>, p, grid, smem, stream);
	}
}

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
	a.ws = p.ws; a.wt = p.wt;
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
		// Tiles of at most ten K-steps (d <= 320): the loads of tile ti + 1 are issued before the MFMAs of tile ti.
		// No branches around the loads (clamped addresses, results zeroed by a select): a branch would make the
		// compiler wait for all outstanding loads at its join.
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
		const bool one_chunk = p.nk32 <= 10;
		bf16x8 xa[10], xb[10];
		if (one_chunk && ntiles > 0) load10(tp, 0, xa);
		for (int ti = 0; ti < ntiles; ti++) {
			f32x4 acc[VK_QB_MAX];
#pragma unroll
			for (int qb = 0; qb < VK_QB_MAX; qb++) acc[qb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
			if (one_chunk) {
				const uint8_t *nxt = tp + (ti + 1 < ntiles ? p.tile_bytes : 0);   // last tile: a harmless reload
				load10(nxt, 0, xb);
				mfma10(0, xa, acc);
#pragma unroll
				for (int i = 0; i < 10; i++) xa[i] = xb[i];
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
				p.raw[(int64_t)qb * p.n_sent + s_idx] = r;
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

// ---------------------------------------------------------------------------
// One-token slices against a one-token query (the span-embedding index: one vector per sentence / window,
// PartitionEmbeddingSim, vectorian/index.py:679-810): local alignment of a 1 x 1 matrix is the clipped cosine
// itself, so the kernel is the similarity tile alone: one MFMA tile = 16 slices, scores written 64 bytes at a
// time.  HBM-bound: d * 2 bytes per slice.
// ---------------------------------------------------------------------------

template <int NK32, bool TAIL>
__global__ __launch_bounds__(256) void vk_span_kernel(VkScoreParams p) {
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	QFrag<NK32, TAIL> qf;
	if constexpr (NK32 > 0) load_qfrag<NK32, TAIL>(qf, p.qtile, lane);
	const int64_t n_tiles = ((int64_t)p.n_sent + 15) >> 4;
	for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
		const uint8_t *tp = p.tiles + tile * p.tile_bytes;
		f32x4 acc;
		if constexpr (NK32 > 0) acc = sim_tile<NK32, TAIL>(qf, tp, lane);
		else acc = sim_tile_generic(p.qtile, tp, p.nk32, p.tail, lane, p.prec);
		const int64_t s_idx = tile * 16 + lane;
		if (lane < 16 && s_idx < p.n_sent) {
			const float raw = acc[0];                       // query column 0, token lane
			const float boost = p.boost ? p.boost[s_idx] : 1.0f;
			p.scores[s_idx] = (raw / p.ref_total) * boost;
			p.raw[s_idx] = raw;
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
	static const char *ov = getenv("VK_BLOCKS_PER_CU");
	if (ov && atoi(ov) > 0) occ = atoi(ov);
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	const int64_t want = (((int64_t)p.n_sent + 15) / 16 + 3) / 4, cap = (int64_t)cus * occ;
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

template <int MODE, int NK32, bool TAIL>
static hipError_t launch_score_gap(const VkScoreParams &p, int grid, size_t smem, hipStream_t stream) {
	switch (p.gap_mode) {
	case 0: return launch_score_lt<MODE, NK32, TAIL, 0>(p, grid, smem, stream);
	case 1: return launch_score_lt<MODE, NK32, TAIL, 1>(p, grid, smem, stream);
	case 3: return launch_score_lt<MODE, NK32, TAIL, 3>(p, grid, smem, stream);
	case 4: return launch_score_lt<MODE, NK32, TAIL, 4>(p, grid, smem, stream);
	case 5: return launch_score_lt<MODE, NK32, TAIL, 5>(p, grid, smem, stream);
	case 6: return launch_score_lt<MODE, NK32, TAIL, 6>(p, grid, smem, stream);
	case 7: return launch_score_lt<MODE, NK32, TAIL, 7>(p, grid, smem, stream);
	default: return launch_score_lt<MODE, NK32, TAIL, 2>(p, grid, smem, stream);
	}
}

extern "C" hipError_t vk_launch_score(const VkScoreParams *pp, int32_t grid, size_t smem_bytes, hipStream_t stream) {
	const VkScoreParams &p = *pp;
	if (p.layout == VK_DEV_LAYOUT_STATIC) return launch_score_gap<2, 0, false>(p, grid, smem_bytes, stream);
	if (p.prec == 0 && p.nk32 == 10 && p.tail == 1) return launch_score_gap<0, 10, true>(p, grid, smem_bytes, stream);
	if (p.prec == 0 && p.nk32 == 24 && p.tail == 0) return launch_score_gap<3, 24, false>(p, grid, smem_bytes, stream);
	return launch_score_gap<1, 0, false>(p, grid, smem_bytes, stream);
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
	vk_topk_keys_kernel<<<nb, 256, 0, stream>>>(in, n, k, out);
	*n_blocks_out = nb;
	return hipGetLastError();
}

extern "C" hipError_t vk_launch_wrd_exact(const VkWrdParams *p, int32_t n_cand, float *scores_to_mark, hipStream_t stream) {
	vk_wrd_exact_kernel<<<n_cand, 64, 0, stream>>>(*p);
	if (scores_to_mark) vk_mark_kernel<<<(n_cand + 255) / 256, 256, 0, stream>>>(p->keys, n_cand, scores_to_mark);
	return hipGetLastError();
}

extern "C" hipError_t vk_launch_rows(const VkWrdParams *p, int32_t n_cand, hipStream_t stream) {
	vk_rows_kernel<<<n_cand, 64, 0, stream>>>(*p);
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

template <int NK, bool HALF>
static hipError_t launch_rwmd_batch_tps(const VkRwmdBatchParams &p, size_t smem, hipStream_t stream) {
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	const int tpw = p.tiles_per_sent == 3 ? 3 : 4;
	const int64_t n_chunks = (p.n_tiles + tpw * 4 - 1) / (tpw * 4);
	const int grid = (int)(n_chunks < (int64_t)cus * 2 ? n_chunks : (int64_t)cus * 2);
	switch (p.tiles_per_sent) {
	case 1: vk_rwmd_batch_kernel<NK, HALF, 1><<<grid, 256, smem, stream>>>(p); break;
	case 2: vk_rwmd_batch_kernel<NK, HALF, 2><<<grid, 256, smem, stream>>>(p); break;
	case 3: vk_rwmd_batch_kernel<NK, HALF, 3><<<grid, 256, smem, stream>>>(p); break;
	default: vk_rwmd_batch_kernel<NK, HALF, 4><<<grid, 256, smem, stream>>>(p); break;
	}
	return hipGetLastError();
}

// 32-token sentences: p->qtiles holds p->n_qtiles A tiles of 32 rows (p->qpt queries each)
extern "C" hipError_t vk_launch_rwmd_batch32(const VkRwmdBatchParams *p, hipStream_t stream) {
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	if (p->tiles_per_sent != 2 || (p->qpt != 2 && p->qpt != 3)) return hipErrorNotSupported;
	const int64_t n_chunks = ((int64_t)p->n_sent + 15) / 16;
	const int grid = (int)(n_chunks < (int64_t)cus ? n_chunks : (int64_t)cus);
	if (p->nk == 10 && p->half == 1) {
		const size_t smem = 2 * 19 * 1024;
		if (p->qpt == 3) vk_rwmd_batch32_kernel<19, 3><<<grid, 512, smem, stream>>>(*p);
		else vk_rwmd_batch32_kernel<19, 2><<<grid, 512, smem, stream>>>(*p);
	} else if (p->nk == 4 && p->half == 0) {
		const size_t smem = 2 * 8 * 1024;
		if (p->qpt == 3) vk_rwmd_batch32_kernel<8, 3><<<grid, 512, smem, stream>>>(*p);
		else vk_rwmd_batch32_kernel<8, 2><<<grid, 512, smem, stream>>>(*p);
	} else return hipErrorNotSupported;
	return hipGetLastError();
}

// returns hipErrorNotSupported when no batched kernel exists for this corpus shape
extern "C" hipError_t vk_launch_rwmd_batch(const VkRwmdBatchParams *p, hipStream_t stream) {
	const size_t smem = (size_t)p->tile_bytes * 2;
	if (p->nk == 10 && p->half == 1) return launch_rwmd_batch_tps<10, true>(*p, smem, stream);
	if (p->nk == 4 && p->half == 0) return launch_rwmd_batch_tps<4, false>(*p, smem, stream);
	return hipErrorNotSupported;
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

extern "C" hipError_t vk_launch_wide(const VkWideParams *p, int32_t flow_k, hipStream_t stream) {
	const bool flow = flow_k > 0;
	const size_t smem = vk_wide_lds_bytes(p->max_len, p->nq, p->gap_mode, p->pos_s != nullptr, flow);
	if (smem > 160 * 1024) return hipErrorInvalidValue;
	const void *fn = flow ? reinterpret_cast<const void *>(vk_wide_kernel<true>) : reinterpret_cast<const void *>(vk_wide_kernel<false>);
	if (smem > 64 * 1024) {
		hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
		if (e != hipSuccess) return e;
	}
	if (flow) {
		vk_wide_kernel<true><<<flow_k, 64, smem, stream>>>(*p);
	} else {
		int occ = 0, dev = 0, cus = 256;
		hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, vk_wide_kernel<false>, 64, smem);
		if (e != hipSuccess) return e;
		if (occ < 1) occ = 1;
		if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
		const int64_t want = p->n_sent, cap = (int64_t)cus * occ;
		vk_wide_kernel<false><<<(int)(want < cap ? want : cap), 64, smem, stream>>>(*p);
	}
	return hipGetLastError();
}

extern "C" size_t vk_wide_lds_demand(int32_t max_len, int32_t nq, int32_t gap_mode, int32_t tagged, int32_t flow) {
	return vk_wide_lds_bytes(max_len, nq, gap_mode, tagged != 0, flow != 0);
}

extern "C" hipError_t vk_launch_flow(const VkFlowParams *p, int32_t k, hipStream_t stream) {
	const size_t smem = vk_flow_lds_bytes(p->max_len, p->pos_s != nullptr);
	if (smem > 64 * 1024) {
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(vk_flow_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
		if (e != hipSuccess) return e;
	}
	vk_flow_kernel<<<k, 64, smem, stream>>>(*p);
	return hipGetLastError();
}
