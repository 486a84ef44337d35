// vk_score32.hip -- the fused scoring kernel for queries of 17..64 tokens (a whole sentence as the query) over slices
// of at most 64 tokens; contextual layout (bf16 or fp32 tiles), or the static layout (token ids + per-query tables).  Same plan as
// vk_score_kernel, NB column blocks of 16 wide:
//   NB = 2 (17..32 tokens): a wave holds TWO slices at a time, 32 lanes each (lane = query column);
//   NB = 4 (33..64 tokens): one slice per wave, 64 lanes.
// Every token tile is loaded once and multiplied with all NB query tiles (staged in LDS), the similarity strip has up to
// 16 NB columns per row, and the in-row recurrence of dp_linear / dp_affine -- a decayed prefix maximum -- is carried
// from block to block with row_bcast:15; general gaps (strictly subadditive w_t): dp32_general; injective RWMD: rwmd32.
// vk_wide_kernel (one wave per slice, serial in-row chain) remains the path for gap costs that are not subadditive,
// long slices, and the tracebacks of the winners: 29 ms per 1 M x 32-token slices there (20 tokens, linear gap),
// 3.5 ms here.
#include "vk_common.hip.h"
#include <algorithm>

// NB = column blocks of 16 per slice: 2 (two slices per wave, queries of 17..32 tokens) or 4 (one slice, 33..64).
// lane 15 of every 16-lane row, handed to all lanes of the NEXT row of the same slice (rows 1 and 3 for NB = 2,
// rows 1, 2, 3 for NB = 4); the other rows read 0.
template <int NB>
__device__ __forceinline__ float from_left_block(float x) {
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, NB == 2 ? 0xa : 0xe, 0xf, false));   // row_bcast:15
}

// value of the column to the left (col - 1) of the same slice; `edge` enters at column 0
template <int NB>
__device__ __forceinline__ float left_neighbour(float x, float edge, int v16, int blk) {
	const float in_row = dpp_f<DPP_ROW_SHR1>(edge, x);      // lanes 1..15 of each row; lane 0 keeps `edge`
	const float across = from_left_block<NB>(x);              // column 15 of the block to the left
	return (blk >= 1 && v16 == 0) ? across : in_row;
}

struct Decay32 { DecaySteps d; float far; };

__device__ __forceinline__ Decay32 decay32_steps(float g, int v16, int blk) {
	// within a block as decay_steps; `far`: distance in gap units from the last column of the block to the left
	return {decay_steps(g, v16), blk >= 1 ? (float)(v16 + 1) * g : __builtin_inff()};
}

// H[j] = max_k (x[j-k] - k g) over the 16 NB columns of a slice: the scan inside the blocks, then the carry from
// block to block (one step for NB = 2; for NB = 4 three steps, each seeing the block to its left already complete)
template <int NB>
__device__ __forceinline__ float decay_scan32(float x, const Decay32 &s, int blk) {
	x = decay_scan<16>(x, s.d);
	if (NB == 2) return fmaxf(x, from_left_block<NB>(x) - s.far);   // left blocks: far = inf, candidate -inf
#pragma unroll
	for (int r = 1; r < NB; r++) {
		const float t = from_left_block<NB>(x) - s.far;
		x = blk == r ? fmaxf(x, t) : x;
	}
	return x;
}

// maximum over the lanes of a slice -> its last lane
template <int NB>
__device__ __forceinline__ float slice_max_to_last_lane(float x, int blk) {
	x = row_max_to_lane15(x);
#pragma unroll
	for (int r = 1; r < NB; r++) {
		const float l = from_left_block<NB>(x);
		x = blk == r ? fmaxf(x, l) : x;
	}
	return x;
}

template <int GAP, int NB>
__device__ __forceinline__ float dp32(const float *__restrict__ S, int stride, int rowbase, int len, int maxlen, int col, const VkWideParams &p) {
	const int v16 = col & 15, blk = col >> 4;
	const bool is_local = p.locality == VK_DEV_LOCAL, is_global = p.locality == VK_DEV_GLOBAL;
	const float floor0 = is_local ? 0.0f : VK_NEG_INF;
	const bool last_col = col == p.len_t - 1;
	const float gs = p.gs, gt = p.gt;
	const Decay32 dt = decay32_steps(gt, v16, blk);
	float h, e = VK_NEG_INF, best = 0.0f;
	if (GAP == 0) h = is_global ? -(gt * (float)(col + 1)) : 0.0f;
	else h = is_global ? -(p.a_t + gt * (float)(col + 1)) : 0.0f;
	for (int u = 1; u <= maxlen; u++) {
		const bool act = u <= len;
		const float s = S[(rowbase + (act ? u - 1 : 0)) * stride + (col < stride ? col : 0)];   // columns >= stride hold no query token
		float bprev, bcur;   // border column H[u-1][0], H[u][0]
		if (GAP == 0) {
			const float gsb = is_global ? gs : 0.0f;
			bprev = -(gsb * (float)(u - 1)); bcur = -(gsb * (float)u);
		} else {
			bprev = (is_global && u > 1) ? -(p.a_s + gs * (float)(u - 1)) : 0.0f;
			bcur = is_global ? -(p.a_s + gs * (float)u) : 0.0f;
		}
		const float diag = left_neighbour<NB>(h, bprev, v16, blk);
		float c = fmaxf(diag + s, floor0), hn;
		if (GAP == 0) {
			c = fmaxf(c, h - gs);
			hn = decay_scan32<NB>(c, dt, blk);
			if (!is_local) hn = fmaxf(hn, bcur - gt * (float)(col + 1));
		} else {
			// Gotoh with open_t >= extend_t (checked by the host): F is the decayed prefix maximum of c shifted by one column
			const float en = fmaxf(h - p.open_s, e - gs);
			c = fmaxf(c, en);
			const float f = decay_scan32<NB>(left_neighbour<NB>(c, bcur, v16, blk) - p.open_t, dt, blk);
			hn = fmaxf(c, f);
			e = act ? en : e;
		}
		h = act ? hn : h;
		if (is_local || last_col) best = fmaxf(best, h);
	}
	float m;
	if (is_local) m = col < p.len_t ? best : 0.0f;
	else if (is_global) m = last_col ? h : VK_NEG_INF;
	else m = col < p.len_t ? fmaxf(h, last_col ? best : 0.0f) : 0.0f;
	m = slice_max_to_last_lane<NB>(m, blk);
	return is_global ? m : fmaxf(m, 0.0f);
}

// Relaxed word mover's distance, injective form (rwmd_rows of vk_score_kernel over 16 NB columns): column minima in
// the lanes, row minima folded to the slice's last lane, the sum over the query columns in column order.
template <int NB>
__device__ __forceinline__ float rwmd32(const float *__restrict__ S, int stride, int rowbase, int len, int maxlen, int col, int lane,
	const VkWideParams &p) {
	constexpr int LPS = 16 * NB;
	const int blk = col >> 4;
	const bool nbow = p.rwmd_normalize_bow != 0, col_ok = col < p.len_t;
	const float w_t = nbow ? 1.0f / (float)p.len_t : 1.0f, w_s = nbow ? 1.0f / (float)(len > 0 ? len : 1) : 1.0f;
	const float BIG = 3.402823466e+38F;
	float colmin = BIG, acc1 = 0.0f;
	for (int u = 1; u <= maxlen; u++) {
		const bool act = u <= len;
		const float dist = fmaxf(1.0f - S[(rowbase + (act ? u - 1 : 0)) * stride + (col < stride ? col : 0)], 0.0f);
		if (act) colmin = fminf(colmin, dist);
		const float m = -slice_max_to_last_lane<NB>(col_ok ? -dist : -BIG, blk);   // min = -max(-x), exact
		if (act) acc1 += w_s * m;
	}
	const float x = col_ok ? w_t * colmin : 0.0f;
	float acc0 = 0.0f;
	for (int j = 0; j < p.len_t; j++) {   // in column order, as the oracle sums (the wave holds 64 / LPS slices)
		float xj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), j));
		if (NB == 2) {
			const float xb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), LPS + j));
			xj = lane >= LPS ? xb : xj;
		}
		acc0 = j == 0 ? xj : acc0 + xj;
	}
	if (!nbow) { acc0 = acc0 / (float)p.len_t; acc1 = acc1 / (float)(len > 0 ? len : 1); }
	float cost = acc0;
	if (p.rwmd_symmetric) { cost = 0.0f; if (acc0 > cost) cost = acc0; if (acc1 > cost) cost = acc1; }
	const float max_cost = nbow ? 1.0f : (float)p.len_t;
	return (max_cost - cost) / max_cost;
}

// Relaxed word mover's distance, 1:n form (rwmd_fill_rows of vk_score_kernel over 16 NB columns; RelaxedSolver with injective =
// false, vectorian/core/cpp/alignment/wmd.h:339-376, restated as upstream is written incl. the re-charged last shipment, :373-375).
//   direction 0 (t -> s): lane = source (query token): its column is consumed in ascending (distance, row) order;
//   direction 1 (s -> t): row = source; the lanes of the slice are the targets, the nearest unused one found by a minimum over
//     keys (distance bits with the column in the low 6 bits) folded to the slice's last lane and read back with v_readlane.
// smass: masses of the slice's vocabulary entries (static layout: count / len at a token's first occurrence, 0 at its
// repetitions; null: every position an entry of its own), in wave-private LDS, one segment per slice of the wave.
template <int NB>
__device__ __forceinline__ float rwmd_fill32(const float *__restrict__ S, int stride, const float *__restrict__ smass, int rowbase, int len, int maxlen,
	int col, int lane, const VkWideParams &p) {
	constexpr int LPS = 16 * NB;
	const int blk = col >> 4;
	const int len_t = p.len_t;
	const bool nbow = p.rwmd_normalize_bow != 0, col_ok = col < len_t;
	const float cap_s = nbow ? 1.0f / (float)(len > 0 ? len : 1) : 1.0f;   // capacity of a slice position
	const float INF = __builtin_inff();
	const float q_mass = col_ok ? p.qmass[col < VK_DEV_MAX_WIDE_QUERY_LEN ? col : 0] : 0.0f;
	const int sc = col < stride ? col : 0;
	// ---- direction 0
	float rem = q_mass, cost0 = 0.0f, last_d = -1.0f;
	int last_i = -1;
	bool fin = !(rem > 0.0f) || len < 1;
	for (int round = 0; round < maxlen && __any(!fin); round++) {
		float bd = INF;
		int bi = -1;
		for (int u = 1; u <= maxlen; u++) {
			const bool act = u <= len;
			const float dist = fmaxf(1.0f - S[(rowbase + (act ? u - 1 : 0)) * stride + sc], 0.0f);
			const bool later = dist > last_d || (dist == last_d && u - 1 > last_i);
			if (act && later && dist < bd) { bd = dist; bi = u - 1; }
		}
		if (!fin) {
			const float cap = (smass && bi >= 0) ? smass[bi] : cap_s;
			if (bi < 0) fin = true;
			else if (rem <= cap) { cost0 += rem * bd; fin = true; }
			else { rem -= cap; cost0 += cap * bd; last_d = bd; last_i = bi; }
		}
	}
	if (rem > 0.0f) cost0 += rem;
	float acc0 = 0.0f;
	for (int j = 0; j < len_t; j++) {   // in column order, as the oracle sums
		float xj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cost0), j));
		if (NB == 2) {
			const float xb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cost0), LPS + j));
			xj = lane >= LPS ? xb : xj;
		}
		acc0 = j == 0 ? xj : acc0 + xj;
	}
	// ---- direction 1
	float acc1 = 0.0f;
	if (p.rwmd_symmetric) {
		for (int u = 1; u <= maxlen; u++) {
			const bool act = u <= len;
			const float dist = fmaxf(1.0f - S[(rowbase + (act ? u - 1 : 0)) * stride + sc], 0.0f);
			float r1 = act ? (smass ? smass[u - 1] : cap_s) : 0.0f;
			float cost = 0.0f;
			bool used = !col_ok, done = !(r1 > 0.0f);
			for (int r = 0; r < len_t && __any(!done); r++) {
				// nearest unused target: distances are >= 0, their bit patterns order like the values; ties go to the lower column
				const float key = used ? INF : __builtin_bit_cast(float, (__builtin_bit_cast(int, dist) & ~63) | col);
				const float best = -slice_max_to_last_lane<NB>(-key, blk);
				float kb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, best), LPS - 1));
				int tl = __builtin_bit_cast(int, kb) & 63;
				float td = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dist), tl));
				float tc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q_mass), tl));
				if (NB == 2) {
					const float kb2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, best), 63));
					const int tl2 = __builtin_bit_cast(int, kb2) & 63;
					const float td2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dist), LPS + (tl2 & (LPS - 1))));
					const float tc2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q_mass), LPS + (tl2 & (LPS - 1))));
					if (lane >= LPS) { kb = kb2; tl = tl2; td = td2; tc = tc2; }
				}
				if (!done) {
					if (!(kb < INF)) done = true;
					else if (r1 <= tc) { cost += r1 * td; done = true; }
					else { r1 -= tc; cost += tc * td; }
				}
				if (col == tl && kb < INF) used = true;
			}
			if (r1 > 0.0f) cost += r1;
			acc1 += cost;
		}
	}
	if (!nbow) { acc0 = acc0 / (float)len_t; acc1 = acc1 / (float)(len > 0 ? len : 1); }
	float cost = acc0;
	if (p.rwmd_symmetric) { cost = 0.0f; if (acc0 > cost) cost = acc0; if (acc1 > cost) cost = acc1; }
	const float max_cost = nbow ? 1.0f : (float)len_t;
	return (max_cost - cost) / max_cost;
}

// sum over the lanes of a slice -> its last lane (any order: used for bounds only)
template <int NB>
__device__ __forceinline__ float slice_sum_to_last_lane(float x, int blk) {
	x += dpp_f<DPP_ROW_SHR1>(0.0f, x);
	x += dpp_f<DPP_ROW_SHR2>(0.0f, x);
	x += dpp_f<DPP_ROW_SHR4>(0.0f, x);
	x += dpp_f<DPP_ROW_SHR8>(0.0f, x);
#pragma unroll
	for (int r = 1; r < NB; r++) {
		const float l = from_left_block<NB>(x);
		x = blk == r ? x + l : x;
	}
	return x;
}

// Exact transport (Word Rotator's Distance, full WMD) with queries of 17..64 tokens, stage 1: an upper bound of the score of every
// slice (stage 2, vk_wrd_exact_kernel, solves the survivors).  Every unit of mass travels at least to its nearest partner, so
//   sum_j m_t[j] min_i C[j][i]   and   sum_i m_s[i] min_j C[j][i],   C = max(0, 1 - S)   (wrd.h:104-109, wmd.h:107-135)
// bound the cost of any feasible plan from below whenever that side ships all its mass (the nearest-neighbour relaxation; the
// 16-column kernel adds capacity constraints, wrd_bound_rows).  Masses: magnitudes (WRD; normalised to 1 per side unless
// wrd_raw_total > 0) or the unit / 1 / len masses of bags of words (mag == null).  Only a bound: float order is free.
template <int NB>
__device__ __forceinline__ float transport_bound32(const float *__restrict__ S, int stride, int rowbase, int len, int maxlen, int col, int lane,
	const VkWideParams &p, const float *__restrict__ mag, const int32_t *__restrict__ ids) {
	constexpr int LPS = 16 * NB, REGS = 64 / LPS;   // lanes per slice; registers that hold a slice's (at most 64) masses
	const int blk = col >> 4;
	const bool col_ok = col < p.len_t;
	const float BIG = 3.402823466e+38F;
	// the slice's masses, fetched once: lane `col` of the slice holds tokens col, col + LPS
	float mreg[REGS];
#pragma unroll
	for (int c = 0; c < REGS; c++) {
		const int u0 = c * LPS + col;
		float mg = 0.0f;
		if (u0 < len) mg = mag ? (ids ? mag[ids[u0]] : mag[u0]) : 1.0f;
		mreg[c] = mg;
	}
	const int slice_lane0 = lane & ~(LPS - 1);
	float colmin = BIG, sum_s = 0.0f, lb1n = 0.0f;
#pragma unroll
	for (int c = 0; c < REGS; c++) {
		const int u_end = maxlen < (c + 1) * LPS ? maxlen : (c + 1) * LPS;
		for (int u = c * LPS + 1; u <= u_end; u++) {
			const bool act = u <= len;
			const float dist = fmaxf(1.0f - S[(rowbase + (act ? u - 1 : 0)) * stride + (col < stride ? col : 0)], 0.0f);
			if (act) colmin = fminf(colmin, dist);
			const float m = -slice_max_to_last_lane<NB>(col_ok ? -dist : -BIG, blk);   // nearest query token of this slice token
			const float mg = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((slice_lane0 | ((u - 1) & (LPS - 1))) * 4, __builtin_bit_cast(int, mreg[c])));
			sum_s += mg;            // 0 beyond the slice's length
			lb1n += mg * m;         // complete in the slice's last lane
		}
	}
	const float x = slice_sum_to_last_lane<NB>(col_ok ? p.qmass[col < VK_DEV_MAX_WIDE_QUERY_LEN ? col : 0] * colmin : 0.0f, blk);
	const float lb1 = sum_s > 0.0f ? lb1n / sum_s : 0.0f;
	float lb;
	if (p.wmd_bound == 2) lb = p.len_t <= len ? x / (float)p.len_t : lb1;            // unit masses: the shorter side ships everything
	else if (p.wrd_raw_total > 0.0f) lb = p.wrd_raw_total <= sum_s ? x / p.wrd_raw_total : lb1;   // magnitudes as they are: the lighter side
	else lb = fmaxf(x, lb1);                                                         // both sides ship 1
	if (!(sum_s > 0.0f)) lb = 0.0f;
	return fminf(1.0f - lb * (1.0f - 4e-6f) + 3e-5f, 1.0f);
}

// General gap costs (Waterman-Smith-Beyer), any table through the closure of w_t (dp_general_reg, vk_common.hip.h): the
// column history of each lane in registers; in-row candidates c[col - k] - w_t(k) from the lane's own block by
// row_shr:k, and for the blocks further right also from the columns of the blocks to their left, which pass through a
// 64-float slot of wave-private LDS (one write, broadcast b128 reads) and meet per-lane costs w_t(col - i) held in registers.
// NB = 4 (queries of 33..64 tokens, one slice per wave): the far candidates are a triangle -- block b has 16 b of them, block 3
// three times 16 while block 0 idles -- and a wave executes the longest lane's count.  They are dealt out evenly, 24 per lane:
//   block 0 lanes: sources 24..47 for column 48 + v of block 3;     block 2: its own, sources 0..23 (+ 8 from block 1's lanes)
//   block 1 lanes: their own 16, then sources 24..31 for column 32 + v;   block 3: its own, sources 0..23 (+ 24 from block 0's)
// the helpers' partial maxima cross through a second 64-float slot.  (Round 2 walked 48 candidates in every lane and fetched
// each cost from a copy of w_t in LDS: four to five operations per candidate, 216 of a row's ~300; 40 tokens: 10.4 ms = 0.23 of
// the HBM roofline.)
// x[lane - K] - w within the DPP row as ONE v_sub_f32_dpp; lanes without a source lane read 0 (bound_ctrl) and carry w = +inf
template <int K, bool FIRST = false>
__device__ __forceinline__ float shr_sub(float x, float w) {
	float r;
	if constexpr (FIRST) asm("s_nop 1\n\tv_sub_f32_dpp %0, %1, %2 row_shr:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(x), "v"(w), "n"(K));
	else asm("v_sub_f32_dpp %0, %1, %2 row_shr:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(r) : "v"(x), "v"(w), "n"(K));
	return r;
}

// B3 (NB = 4, queries of 33..48 tokens: block 3 holds no column): the far candidates are 16 for block 1's columns and 32 for block 2's,
// 48 per pair of lanes v / 16 + v / 32 + v / 48 + v -- dealt out 12 per lane instead of 24:
//   block 1 lanes: their sources 0..11;   block 3 lanes: sources 24..31 for column 32 + v, then sources 12..15 for column 16 + v
//   block 2 lanes: their sources 0..11;   block 0 lanes: sources 12..23 for column 32 + v
// the helpers' maxima cross through two 64-float slots (block 3's lanes help two columns).
// B3 with NB = 2 (queries of 17..32 tokens, two slices per wave): block 1's 16 far candidates are shared with block 0's lanes, which
// have none of their own -- 8 per lane: block 1 lanes walk sources 0..7, block 0's lane v sources 8..15 for column 16 + v of its slice;
// the helpers' maxima cross through a second 64-float slot (the launcher takes this form when the slot does not cost a workgroup per CU).
template <int MAXLEN, int NB, bool B3 = false>
__device__ __forceinline__ float dp32_general(const float *__restrict__ S, int stride, int rowbase, int len, int maxlen, int col, int lane,
	const VkWideParams &p, const float (&wsr)[MAXLEN + 1], float *__restrict__ xch) {
	constexpr bool B2 = B3 && NB == 2;     // the balanced two-block form
	constexpr bool B34 = B3 && NB == 4;    // the three-block balance of the four-block form
	const int v16 = col & 15, blk = col >> 4;
	const bool is_local = p.locality == VK_DEV_LOCAL, is_global = p.locality == VK_DEV_GLOBAL;
	const float floor0 = is_local ? 0.0f : VK_NEG_INF;
	const bool last_col = col == p.len_t - 1;
	const float inf = __builtin_inff();
	const float wt_border = p.wt[col + 1], wt_border0 = p.wt0[col + 1];   // chains of gaps from the border column / one gap (border row)
	constexpr bool BAL = NB == 4 && !B3;
	constexpr int NFAR = B2 ? 8 : B34 ? 12 : BAL ? 24 : 16 * (NB - 1);   // far candidates a lane walks
	float wtv[16], wfar[NFAR];
#pragma unroll
	for (int k = 1; k < 16; k++) wtv[k] = v16 >= k ? p.wt[k] : inf;
	// group g of four source columns this lane reads, and the column it works for there
	auto far_group = [&](int g) {
		if constexpr (B2) return blk == 0 ? g + 2 : g;
		else if constexpr (B34) return blk == 0 ? g + 3 : blk == 3 ? (g < 2 ? g + 6 : 3) : g;
		else return !BAL ? g : blk == 0 ? g + 6 : (blk == 1 && g >= 4) ? g + 2 : g;
	};
	auto far_target = [&](int g) {
		if constexpr (B2) return blk == 0 ? 16 + v16 : col;
		else if constexpr (B34) return blk == 0 ? 32 + v16 : blk == 3 ? (g < 2 ? 32 + v16 : 16 + v16) : col;
		else return !BAL ? col : blk == 0 ? 48 + v16 : (blk == 1 && g >= 4) ? 32 + v16 : col;
	};
#pragma unroll
	for (int i = 0; i < NFAR; i++) {
		const int src = 4 * far_group(i >> 2) + (i & 3), tgt = far_target(i >> 2);
		wfar[i] = (src >> 4) < (tgt >> 4) ? p.wt[tgt - src] : inf;   // a source in a block left of the target's
	}
	const f32x4 *left = reinterpret_cast<const f32x4 *>(xch + (lane & ~(16 * NB - 1)));   // c of columns 0.. of this slice
	[[maybe_unused]] float *help = xch + 64;                           // (VK_HELP_LDS) NB = 4: the helpers' partial maxima
	[[maybe_unused]] const int partner = blk == 3 ? col - 48 : col - 16;   // block 3 <- block 0's lane v, block 2 <- block 1's lane 16 + v

	float hreg[MAXLEN + 1];
	float h = is_global ? -wt_border0 : 0.0f;
	hreg[0] = h;
	float best = 0.0f;
#pragma unroll
	for (int u = 1; u <= MAXLEN; u++) {
		if (u <= maxlen) {
			const bool act = u <= len;
			const float s = S[(rowbase + (act ? u - 1 : 0)) * stride + (col < stride ? col : 0)];
			const float bprev = is_global ? -wsr[u - 1] : 0.0f;
			const float bcur = is_global ? -wsr[u] : 0.0f;
			const float diag = left_neighbour<NB>(hreg[u - 1], bprev, v16, blk);
			float c = fmaxf(diag + s, floor0);
#pragma unroll
			for (int k = 1; k <= u; k++) c = fmaxf(c, hreg[u - k] - wsr[k]);
			xch[lane] = c;
			float hc = fmaxf(c, bcur - wt_border);
			// (written as the fused instruction: beside the far candidates hipcc leaves these as v_mov_b32_dpp + v_subrev_f32, 15 more
			// VALU operations per row of ~160; the first one waits out the two states a DPP read of a fresh VALU result needs)
			hc = fmaxf(hc, shr_sub<1, true>(c, wtv[1]));
			hc = fmaxf(hc, shr_sub<2>(c, wtv[2]));
			hc = fmaxf(hc, shr_sub<3>(c, wtv[3]));
			hc = fmaxf(hc, shr_sub<4>(c, wtv[4]));
			hc = fmaxf(hc, shr_sub<5>(c, wtv[5]));
			hc = fmaxf(hc, shr_sub<6>(c, wtv[6]));
			hc = fmaxf(hc, shr_sub<7>(c, wtv[7]));
			hc = fmaxf(hc, shr_sub<8>(c, wtv[8]));
			hc = fmaxf(hc, shr_sub<9>(c, wtv[9]));
			hc = fmaxf(hc, shr_sub<10>(c, wtv[10]));
			hc = fmaxf(hc, shr_sub<11>(c, wtv[11]));
			hc = fmaxf(hc, shr_sub<12>(c, wtv[12]));
			hc = fmaxf(hc, shr_sub<13>(c, wtv[13]));
			hc = fmaxf(hc, shr_sub<14>(c, wtv[14]));
			hc = fmaxf(hc, shr_sub<15>(c, wtv[15]));
			wave_lds_fence();
			float fa = VK_NEG_INF, fb = VK_NEG_INF;   // maxima over the groups 0..3 / 4.. of the lane's list (B3: groups 0, 1 / 2)
#pragma unroll
			for (int g = 0; g < NFAR / 4; g++) {
				const f32x4 l = left[far_group(g)];
#pragma unroll
				for (int r = 0; r < 4; r++) {
					const float cand = l[r] - wfar[g * 4 + r];
					if (g < (B34 ? 2 : 4)) fa = fmaxf(fa, cand);
					else fb = fmaxf(fb, cand);
				}
			}
#ifndef VK_HELP_LDS
			// the helpers' maxima cross between the DPP rows by lane swaps (round 3: a second LDS slot, a write, a fence and a read per row)
			if constexpr (B2) {
				hc = fmaxf(hc, blk == 1 ? fa : VK_NEG_INF);
				const float hv = lane_xor16(fa, lane);              // block 1 <- block 0's lane v of its slice
				hc = fmaxf(hc, blk == 1 ? hv : VK_NEG_INF);
			} else if constexpr (B34) {
				// blocks 1, 2: everything they found is their own; block 0: for block 2's column; block 3: groups 0, 1 for block 2's, group 2 for block 1's
				hc = fmaxf(hc, (blk == 1 || blk == 2) ? fmaxf(fa, fb) : VK_NEG_INF);
				const float r32 = lane_xor32(blk == 0 ? fmaxf(fa, fb) : fb, lane);   // block 2 <- block 0, block 1 <- block 3
				const float r16 = lane_xor16(fa, lane);                               // block 2 <- block 3
				hc = fmaxf(hc, blk == 2 ? fmaxf(r32, r16) : blk == 1 ? r32 : VK_NEG_INF);
			} else if constexpr (BAL) {
				// block 0: everything it found belongs to block 3's column; block 1: groups 4, 5 belong to block 2's
				hc = fmaxf(hc, blk == 0 ? VK_NEG_INF : blk == 1 ? fa : fmaxf(fa, fb));
				const float hv = lane_xor32(lane_xor16(blk == 0 ? fmaxf(fa, fb) : fb, lane), lane);   // block 3 <- block 0, block 2 <- block 1
				hc = fmaxf(hc, blk >= 2 ? hv : VK_NEG_INF);
#else
			if constexpr (B2) {
				help[lane] = blk == 0 ? fa : VK_NEG_INF;          // for column 16 + v of this lane's slice
				hc = fmaxf(hc, blk == 1 ? fa : VK_NEG_INF);
				wave_lds_fence();
				const float hv = help[blk == 1 ? lane - 16 : lane];
				hc = fmaxf(hc, blk == 1 ? hv : VK_NEG_INF);
			} else if constexpr (B34) {
				// blocks 1, 2: everything they found is their own; block 0: for block 2's column; block 3: groups 0, 1 for block 2's, group 2 for block 1's
				help[lane] = (blk == 0 || blk == 3) ? (blk == 0 ? fmaxf(fa, fb) : fa) : VK_NEG_INF;
				help[64 + lane] = blk == 3 ? fb : VK_NEG_INF;
				hc = fmaxf(hc, (blk == 1 || blk == 2) ? fmaxf(fa, fb) : VK_NEG_INF);
				wave_lds_fence();
				const float h0 = help[blk == 2 ? v16 : lane], h3 = help[blk == 2 ? 48 + v16 : blk == 1 ? 64 + 48 + v16 : lane];
				hc = fmaxf(hc, blk == 2 ? fmaxf(h0, h3) : blk == 1 ? h3 : VK_NEG_INF);
			} else if constexpr (BAL) {
				// block 0: everything it found belongs to block 3's column; block 1: groups 4, 5 belong to block 2's
				help[lane] = blk == 0 ? fmaxf(fa, fb) : fb;
				hc = fmaxf(hc, blk == 0 ? VK_NEG_INF : blk == 1 ? fa : fmaxf(fa, fb));
				wave_lds_fence();
				const float hv = help[blk >= 2 ? partner : lane];
				hc = fmaxf(hc, blk >= 2 ? hv : VK_NEG_INF);
#endif
			} else hc = fmaxf(hc, fmaxf(fa, fb));
			hreg[u] = hc;
			h = act ? hc : h;
			if (is_local || last_col) best = fmaxf(best, h);
		}
	}
	float m;
	if (is_local) m = col < p.len_t ? best : 0.0f;
	else if (is_global) m = last_col ? h : VK_NEG_INF;
	else m = col < p.len_t ? fmaxf(h, last_col ? best : 0.0f) : 0.0f;
	m = slice_max_to_last_lane<NB>(m, blk);
	return is_global ? m : fmaxf(m, 0.0f);
}

// STATIC: token ids + the two per-query tables [V x 16] (columns 0..15 and 16..31) instead of token tiles
// (at least two waves per SIMD asked of the register allocator: the four-block general-gap form took 247 + 24 registers -- ONE wave
// per SIMD, its tile loads and its DP never overlapping with another wave's)
// NWMAX: waves per workgroup the kernel may be launched with.  12 (round 4; the three-block balance of 33..48-token queries over slices
// of at most 32 tokens takes 165 registers): ONE workgroup of twelve waves per CU shares the query tiles -- three waves per SIMD where
// two workgroups of four left two (the LDS of a CU holds the 39 KB of query tiles twice, not three times)
template <int GAP, bool STATIC, int NB, bool B3 = false, int NWMAX = 4>
__global__ __launch_bounds__(64 * NWMAX) __attribute__((amdgpu_waves_per_eu(NWMAX > 4 ? 3 : 2))) void vk_score32_kernel(VkWideParams p, int32_t rows_per_wave, int32_t stride, int32_t slack) {
	constexpr int LPS = 16 * NB, PER = 64 / LPS;   // lanes per slice, slices per wave
	extern __shared__ float4 vk_smem32[];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int nwv = blockDim.x >> 6;   // waves per workgroup: 4, or fewer when the query tiles of wide rows leave no room for four strips (vk_launch_score32)
	// the NB query tiles (tokens 0..15, 16..31, ..) in LDS, shared by the block's waves
	// (a half-filled last K-step takes 512 bytes, as in HBM: its lanes 32..63 re-read the slots of lanes 0..31 and meet zeros on the
	// token side -- 1 KiB per workgroup that decides, at 32 query tokens, whether a third workgroup fits the CU)
	const int qbytes = STATIC ? 0 : (p.tail ? p.nk32 * 1024 - 512 : p.nk32 * 1024);
	const int qpieces = qbytes / 16;
	for (int i = threadIdx.x; !STATIC && i < NB * qpieces; i += blockDim.x) {
		const int t = i / qpieces, o = i - t * qpieces;
		vk_smem32[i] = *reinterpret_cast<const float4 *>(p.qtile + (int64_t)t * p.tile_bytes + o * 16);
	}
	__syncthreads();
	const uint8_t *q0 = reinterpret_cast<const uint8_t *>(vk_smem32);
	// strip rows hold the query columns padded to a multiple of 4 (stride floats), not 32: a third workgroup per CU for 20 tokens
	float *S = reinterpret_cast<float *>(vk_smem32) + NB * (qbytes / 4) + wv * (rows_per_wave * stride + slack);   // 16-byte aligned: qbytes is a multiple of 512

	float *xch = S + rows_per_wave * stride;   // 64 floats behind the strip: the in-row exchange of dp32_general; four-block form: 64 more for the helpers' maxima
	constexpr int WSN = GAP == 6 ? 65 : 33;
	float wsr[WSN];
	if (GAP == 3 || GAP == 6) {
#pragma unroll
		for (int k = 0; k < WSN; k++) wsr[k] = p.ws[k];
	}

	const int half = lane / LPS, col = lane & (LPS - 1);
	const int n_pairs = (p.n_sent + PER - 1) / PER;
	const int nfull = p.tail ? p.nk32 - 1 : p.nk32;
	for (int pi = blockIdx.x * nwv + wv; pi < n_pairs; pi += gridDim.x * nwv) {
		const int s_idx = pi * PER + half;
		const int i0 = s_idx < p.n_sent ? s_idx : p.n_sent;      // entries >= n_sent are empty slices (padding of the table)
		const int t_a = p.sent_start[i0], t_b = p.sent_end[i0];
		const int len = t_b - t_a;
		const int g_a = __builtin_amdgcn_readlane(t_a, 0), g_b = __builtin_amdgcn_readlane(t_b, 64 - LPS);
		const int maxlen = max(__builtin_amdgcn_readlane(len, 0), __builtin_amdgcn_readlane(len, 64 - LPS));
		if (maxlen > VK_DEV_MAX_SENT_LEN) continue;   // a long slice (alone in its padded group): bounded by vk_long_bound_kernel
		const int tile0 = STATIC ? 0 : g_a >> 4;
		const int ntiles = STATIC ? 0 : ((g_b + 15) >> 4) - tile0;
		if (STATIC) {
			// gather: 4 NB lanes per token, 4 query columns each; id, then table row: four deep
			const int ntok = g_b - g_a;
			constexpr int CPL = 4 * NB, TPI = 64 / CPL;   // lanes per token, tokens per iteration
			const int cb = (lane % CPL) * 4;
			const float *tab = p.table + (cb >> 4) * p.table_stride + (cb & 15);
			for (int it0 = 0; it0 * TPI < ntok; it0 += 4) {
				int id[4];
				float4 val[4];
#pragma unroll
				for (int q4 = 0; q4 < 4; q4++) {
					const int tk = (it0 + q4) * TPI + lane / CPL;
					id[q4] = p.tok_id[g_a + (tk < ntok ? tk : 0)];
				}
#pragma unroll
				for (int q4 = 0; q4 < 4; q4++) val[q4] = *reinterpret_cast<const float4 *>(tab + (int64_t)id[q4] * 16);
#pragma unroll
				for (int q4 = 0; q4 < 4; q4++) {
					const int tk = (it0 + q4) * TPI + lane / CPL;
					if (tk < ntok && cb < stride) {
						float4 vq = val[q4];
						if (p.pos_s) {   // tag-weighted modifier (TagWeightedSlice, slice/static.h:237-264)
							const int ps = p.pos_s[g_a + tk];
							vq.x = tag_weighted(vq.x, p.tw[cb + 0], ps, p.tpos[cb + 0], p.tw_keep, p.tw_threshold);
							vq.y = tag_weighted(vq.y, p.tw[cb + 1], ps, p.tpos[cb + 1], p.tw_keep, p.tw_threshold);
							vq.z = tag_weighted(vq.z, p.tw[cb + 2], ps, p.tpos[cb + 2], p.tw_keep, p.tw_threshold);
							vq.w = tag_weighted(vq.w, p.tw[cb + 3], ps, p.tpos[cb + 3], p.tw_keep, p.tw_threshold);
						}
						*reinterpret_cast<float4 *>(S + tk * stride + cb) = vq;
					}
				}
			}
		}
		const uint8_t *tp = p.tiles + (int64_t)tile0 * p.tile_bytes;
		for (int ti = 0; ti < ntiles; ti++) {
			f32x4 acc[NB];
#pragma unroll
			for (int b = 0; b < NB; b++) acc[b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
			int t = 0;
			if (p.prec) {
				// fp32 tiles (the reference's own precision): nk32 blocks of 16 features, four v_mfma_f32_16x16x4_f32 per block and query tile
				int bk = 0;
				for (; bk + 8 <= p.nk32; bk += 8) {   // eight blocks in flight, in front of their MFMAs (20 tokens, fp32 rows: 9.6 -> 7.8 ms)
					f32x4 x[8];
#pragma unroll
					for (int i = 0; i < 8; i++) x[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(tp + (bk + i) * 1024 + lane * 16));
					__builtin_amdgcn_sched_barrier(0);
#pragma unroll
					for (int i = 0; i < 8; i++) {
#pragma unroll
						for (int b = 0; b < NB; b++) {
							const f32x4 q = *reinterpret_cast<const f32x4 *>(q0 + b * qbytes + (bk + i) * 1024 + lane * 16);
#pragma unroll
							for (int e = 0; e < 4; e++) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(q[e], x[i][e], acc[b], 0, 0, 0);
						}
					}
				}
				for (; bk + 4 <= p.nk32; bk += 4) {
					f32x4 x[4];
#pragma unroll
					for (int i = 0; i < 4; i++) x[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(tp + (bk + i) * 1024 + lane * 16));
#pragma unroll
					for (int i = 0; i < 4; i++) {
#pragma unroll
						for (int b = 0; b < NB; b++) {
							const f32x4 q = *reinterpret_cast<const f32x4 *>(q0 + b * qbytes + (bk + i) * 1024 + lane * 16);
#pragma unroll
							for (int e = 0; e < 4; e++) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(q[e], x[i][e], acc[b], 0, 0, 0);
						}
					}
				}
				for (; bk < p.nk32; bk++) {
					const f32x4 x = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(tp + bk * 1024 + lane * 16));
#pragma unroll
					for (int b = 0; b < NB; b++) {
						const f32x4 q = *reinterpret_cast<const f32x4 *>(q0 + b * qbytes + bk * 1024 + lane * 16);
#pragma unroll
						for (int e = 0; e < 4; e++) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(q[e], x[e], acc[b], 0, 0, 0);
					}
				}
				t = nfull;   // nothing left for the bf16 loops (fp32 tiles have no half block)
			}
			// nine K-steps in flight (the nine full steps of a 300-d tile), kept in front of their MFMAs: 32 tokens, linear gaps 3.15 -> 2.97 ms
			// per 1 M x 32 x 300-d (four in flight, and the scheduler sinking the loads), general gaps 4.43 -> 4.26
			constexpr int VK_S32_DEEP = (GAP == 6 && NB == 4) ? 4 : 9;   // (the four-block general-gap form sits at the register cap of two waves per SIMD: 272 registers with nine in flight, one wave per SIMD)
			for (; t + VK_S32_DEEP <= nfull; t += VK_S32_DEEP) {
				bf16x8 x[VK_S32_DEEP];
#pragma unroll
				for (int i = 0; i < VK_S32_DEEP; i++) x[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tp + (t + i) * 1024 + lane * 16));
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for (int i = 0; i < VK_S32_DEEP; i++) {
#pragma unroll
					for (int b = 0; b < NB; b++)
						acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(q0 + b * qbytes + (t + i) * 1024 + lane * 16), x[i], acc[b], 0, 0, 0);
				}
			}
			for (; t + 4 <= nfull; t += 4) {   // four K-steps of the token tile in flight, each feeding all query tiles
				bf16x8 x[4];
#pragma unroll
				for (int i = 0; i < 4; i++) x[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tp + (t + i) * 1024 + lane * 16));
#pragma unroll
				for (int i = 0; i < 4; i++) {
#pragma unroll
					for (int b = 0; b < NB; b++)
						acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(q0 + b * qbytes + (t + i) * 1024 + lane * 16), x[i], acc[b], 0, 0, 0);
				}
			}
			for (; t < nfull; t++) {
				const bf16x8 x = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tp + t * 1024 + lane * 16));
#pragma unroll
				for (int b = 0; b < NB; b++)
					acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(q0 + b * qbytes + t * 1024 + lane * 16), x, acc[b], 0, 0, 0);
			}
			if (p.tail && !p.prec) {   // half-filled last K-step: lanes 32..63 contribute zeros (their LDS slots hold zeros, the token side is masked)
				const bf16x8 x = load_half_block(tp + nfull * 1024, lane, true);
#pragma unroll
				for (int b = 0; b < NB; b++)
					acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8 *>(q0 + b * qbytes + nfull * 1024 + (lane & 31) * 16), x, acc[b], 0, 0, 0);
			}
			const int ps = p.pos_s ? p.pos_s[(tile0 + ti) * 16 + (lane & 15)] : 0, cq = (lane >> 4) * 4;
			float *row = S + (ti * 16 + (lane & 15)) * stride + cq;
#pragma unroll
			for (int b = 0; b < NB; b++) {
#pragma unroll
				for (int r = 0; r < 4; r++) {
					acc[b][r] = clip01(acc[b][r]);
					if (p.pos_s) acc[b][r] = tag_weighted(acc[b][r], p.tw[16 * b + cq + r], ps, p.tpos[16 * b + cq + r], p.tw_keep, p.tw_threshold);
				}
				if (16 * b + cq < stride) *reinterpret_cast<f32x4 *>(row + 16 * b) = acc[b];
			}
			tp += p.tile_bytes;
		}
		wave_lds_fence();
		const int lenc = len > 0 ? len : 0;
		const int rb = len > 0 ? (STATIC ? t_a - g_a : t_a - tile0 * 16) : 0;
		// tag-weighted vocabulary transport: cells upstream writes twice (static_vocab_fixup, vk_common.hip.h).  The slices of the wave
		// share one strip of rows: over sliding windows a rewritten cell may belong to a neighbour too, and the slices take turns
		int turns = 1;
		bool rewrite = false;
		if constexpr (STATIC && (GAP == 4 || GAP == 7 || GAP == 5)) {
			if (p.qid_bits) {
				rewrite = len > 0 && static_vocab_shared(len, p.len_t, p.tok_id + t_a, p.tag_s + t_a, p.qid_bits, p.qkey) != 0;
				if (PER > 1 && p.slices_overlap && __builtin_amdgcn_ballot_w64(rewrite) != 0) turns = PER;
				else if (rewrite) static_vocab_fixup<LPS>(S + rb * stride, stride, len, p.len_t, p.tok_id + t_a, p.tag_s + t_a, p.pos_s + t_a, p.table, p.table_stride,
					p.qid_bits, p.qkey, p.tw, p.tpos, p.tw_keep, p.tw_threshold, col);
				wave_lds_fence();
			}
		}
		auto evaluate = [&]() -> float {
		float raw;
		if constexpr (GAP == 3) raw = dp32_general<32, NB, B3>(S, stride, rb, lenc, maxlen, col, lane, p, wsr, xch);
		else if constexpr (GAP == 6) raw = dp32_general<64, NB, B3>(S, stride, rb, lenc, maxlen, col, lane, p, wsr, xch);
		else if constexpr (GAP == 4) raw = rwmd32<NB>(S, stride, rb, lenc, maxlen, col, lane, p);
		else if constexpr (GAP == 7) {
			// masses of the slice's vocabulary entries (static layout: repeated token ids count once, at their first position),
			// kept behind the strip: LPS lanes fill the slice's (at most 64) entries
			float *sm = nullptr;
			if (STATIC) {
				sm = xch + half * 64;
				const float wsum = p.rwmd_normalize_bow ? (float)(lenc > 0 ? lenc : 1) : 1.0f;   // bow[i] /= w_sum (bow.h:262-270): a division, as upstream -- a mass that ties with a capacity must tie here too
				for (int u = col; u < lenc; u += LPS) {
					const int id = p.tok_id[t_a + u];
					int cnt = 0;
					bool first = true;
					for (int i = 0; i < lenc; i++) {
						const bool same = p.tok_id[t_a + i] == id && (!p.tag_s || p.tag_s[t_a + i] == p.tag_s[t_a + u]);
						cnt += same ? 1 : 0;
						first = first && !(same && i < u);
					}
					sm[u] = first ? (float)cnt / wsum : 0.0f;
				}
				wave_lds_fence();
			}
			raw = rwmd_fill32<NB>(S, stride, sm, rb, lenc, maxlen, col, lane, p);
		}
		else if constexpr (GAP == 5) raw = transport_bound32<NB>(S, stride, rb, lenc, maxlen, col, lane, p,
			p.mag ? (STATIC ? p.mag : p.mag + (len > 0 ? t_a : 0)) : nullptr, (STATIC && p.mag) ? p.tok_id + (len > 0 ? t_a : 0) : nullptr);
		else raw = dp32<GAP, NB>(S, stride, rb, lenc, maxlen, col, p);
		return raw;
		};
		float raw = 0.0f;
		if constexpr (STATIC && (GAP == 4 || GAP == 7 || GAP == 5)) {
			if (turns > 1) {
				for (int turn = 0; turn < PER; turn++) {
					const bool mine = half == turn;
					if (mine && rewrite) static_vocab_fixup<LPS>(S + rb * stride, stride, len, p.len_t, p.tok_id + t_a, p.tag_s + t_a, p.pos_s + t_a, p.table, p.table_stride,
						p.qid_bits, p.qkey, p.tw, p.tpos, p.tw_keep, p.tw_threshold, col);
					wave_lds_fence();
					const float r = evaluate();
					if (mine) raw = r;
					wave_lds_fence();
					if (mine && rewrite) static_vocab_fixup<LPS, false, true>(S + rb * stride, stride, len, p.len_t, p.tok_id + t_a, p.tag_s + t_a, p.pos_s + t_a, p.table, p.table_stride,
						p.qid_bits, p.qkey, p.tw, p.tpos, p.tw_keep, p.tw_threshold, col);
					wave_lds_fence();
				}
			} else raw = evaluate();
		} else raw = evaluate();
		if (col == LPS - 1 && s_idx < p.n_sent) {
			float val = VK_NEG_INF, r = VK_NEG_INF;
			if (len >= 1) {   // document.h:160 skips empty slices
				const float boost = p.boost ? p.boost[s_idx] : 1.0f;
				r = raw;
				val = (raw / p.ref_total) * boost;
			}
			p.scores[s_idx] = val;
			if (p.raw) p.raw[s_idx] = r;
		}
		wave_lds_fence();
	}
}

static inline int strip_stride(int len_t) { return (len_t + 3) / 4 * 4; }
// floats behind a wave's strip: the 64-float exchange slot of dp32_general (general gaps), followed, in the four-block form with
// the 64-row history, by the wave's copy of w_t (65); 16 otherwise (lanes beyond the strip's columns read into it).  Exact sizes
// matter: at 32 query tokens and 300-d rows 1,280 bytes decide between two and three workgroups per CU (5.4 -> 4.3 ms).
static inline int strip_slack(int gap_mode, int len_t) {
	if (gap_mode == 7) return 144;   // 2 x 64 vocabulary masses
#ifdef VK_HELP_LDS
	if (gap_mode == 3 || gap_mode == 6) return len_t > 48 ? 128 : len_t > 32 ? 192 : 64;   // four-block form: the second slot of dp32_general (33..48 tokens: a third)
#else
	if (gap_mode == 3 || gap_mode == 6) return 64;   // the in-row exchange slot of dp32_general (the helpers' maxima cross by lane swaps since round 4)
#endif
	return 16;
}

// nk32 = 0: static layout (no query tiles in LDS); tiles: token tiles a wave's slices span (two consecutive slices for
// queries of at most 32 tokens, one slice beyond)
// waves: waves per workgroup (each with a strip of its own)
extern "C" size_t vk_score32_lds_bytes(int32_t nk32, int32_t tail, int32_t tiles, int32_t len_t, int32_t gap_mode, int32_t waves) {
	const int nb = len_t <= 32 ? 2 : 4;
	return (size_t)nb * (nk32 * 1024 - (tail && nk32 > 0 ? 512 : 0)) + (size_t)waves * ((size_t)tiles * 16 * strip_stride(len_t) + strip_slack(gap_mode, len_t)) * 4;
}

// waves per workgroup of the launch: 4 when the LDS of a CU holds the query tiles and four strips; wide rows (768-d: four query tiles of
// 24 KB for a query of 49..64 tokens) leave room for 2 or 1 -- slower, but exact transport and the 1:n RWMD have no other kernel for
// such queries (round 2 returned VK_ERR_UNSUPPORTED there).  0: not even one strip fits.
extern "C" int32_t vk_score32_waves(int32_t nk32, int32_t tail, int32_t tiles, int32_t len_t, int32_t gap_mode) {
	for (int w = 4; w >= 1; w >>= 1)
		if (vk_score32_lds_bytes(nk32, tail, tiles, len_t, gap_mode, w) <= 160 * 1024) return w;
	return 0;
}

extern "C" hipError_t vk_launch_score32(const VkWideParams *p, int32_t tiles, hipStream_t stream) {
	const bool is_static = p->layout == VK_DEV_LAYOUT_STATIC;
	const bool four = p->len_t > 32;   // 33..64 tokens: one slice per wave, four column blocks (linear / affine gaps)
	int waves = vk_score32_waves(is_static ? 0 : p->nk32, p->tail, tiles, p->len_t, p->gap_mode);
	if (waves < 1) return hipErrorInvalidValue;
	size_t smem = vk_score32_lds_bytes(is_static ? 0 : p->nk32, p->tail, tiles, p->len_t, p->gap_mode, waves);
	int slack = strip_slack(p->gap_mode, p->len_t);
#ifdef VK_HELP_LDS
	// general gaps, 17..32 tokens: the balanced form of the far candidates needs a second 64-float slot per wave; taken when that
	// does not cost a workgroup per CU (LDS is handed out in 512-byte granules; at 32 tokens and 300-d rows it would: 4.2 -> 5.7 ms)
	bool bal2 = false;
	if (!four && (p->gap_mode == 3 || p->gap_mode == 6) && !getenv("VK_NO_BAL2")) {
		auto per_cu = [](size_t bytes) { return (size_t)(156 * 1024) / ((bytes + 511) / 512 * 512); };   // (4 KB to spare: three workgroups of 54,272 bytes were not admitted)
		const size_t more = smem + (size_t)waves * 64 * 4;
		if (more <= 160 * 1024 && std::min<size_t>(per_cu(more), 3) == std::min<size_t>(per_cu(smem), 3)) { bal2 = true; smem = more; slack += 64; }
	}
#else
	// general gaps, 17..32 tokens: the balanced form of the far candidates (round 3: only where its second 64-float slot per wave did not
	// cost a workgroup per CU -- up to 28 tokens at 300-d; the lane swaps need no slot)
	const bool bal2 = !four && (p->gap_mode == 3 || p->gap_mode == 6) && !getenv("VK_NO_BAL2");
#endif
	void (*kernel)(VkWideParams, int32_t, int32_t, int32_t);
	// 33..48 tokens, general gaps over slices of at most 32 tokens, contextual rows: twelve waves in one workgroup when they fit
	const bool twelve = four && p->gap_mode == 3 && p->len_t <= 48 && !is_static && waves == 4 && !getenv("VK_NO_TWELVE") &&
		vk_score32_lds_bytes(p->nk32, p->tail, tiles, p->len_t, p->gap_mode, 12) <= 156 * 1024;
	if (twelve) { waves = 12; smem = vk_score32_lds_bytes(p->nk32, p->tail, tiles, p->len_t, p->gap_mode, 12); }
	switch (p->gap_mode) {
	case 0: kernel = four ? (is_static ? vk_score32_kernel<0, true, 4> : vk_score32_kernel<0, false, 4>)
		: (is_static ? vk_score32_kernel<0, true, 2> : vk_score32_kernel<0, false, 2>); break;
	case 1: kernel = four ? (is_static ? vk_score32_kernel<1, true, 4> : vk_score32_kernel<1, false, 4>)
		: (is_static ? vk_score32_kernel<1, true, 2> : vk_score32_kernel<1, false, 2>); break;
	case 4: kernel = four ? (is_static ? vk_score32_kernel<4, true, 4> : vk_score32_kernel<4, false, 4>)
		: (is_static ? vk_score32_kernel<4, true, 2> : vk_score32_kernel<4, false, 2>); break;
	case 7: kernel = four ? (is_static ? vk_score32_kernel<7, true, 4> : vk_score32_kernel<7, false, 4>)
		: (is_static ? vk_score32_kernel<7, true, 2> : vk_score32_kernel<7, false, 2>); break;
	case 5: kernel = four ? (is_static ? vk_score32_kernel<5, true, 4> : vk_score32_kernel<5, false, 4>)
		: (is_static ? vk_score32_kernel<5, true, 2> : vk_score32_kernel<5, false, 2>); break;
	// general gaps, 33..48 tokens: the three-block balance of the far candidates (dp32_general<.., B3>)
	case 3: kernel = twelve ? vk_score32_kernel<3, false, 4, true, 12> : four ? (p->len_t <= 48 ? (is_static ? vk_score32_kernel<3, true, 4, true> : vk_score32_kernel<3, false, 4, true>)
			: (is_static ? vk_score32_kernel<3, true, 4> : vk_score32_kernel<3, false, 4>))
		: bal2 ? (is_static ? vk_score32_kernel<3, true, 2, true> : vk_score32_kernel<3, false, 2, true>)
		: (is_static ? vk_score32_kernel<3, true, 2> : vk_score32_kernel<3, false, 2>); break;
	default: kernel = four ? (p->len_t <= 48 ? (is_static ? vk_score32_kernel<6, true, 4, true> : vk_score32_kernel<6, false, 4, true>)
			: (is_static ? vk_score32_kernel<6, true, 4> : vk_score32_kernel<6, false, 4>))
		: bal2 ? (is_static ? vk_score32_kernel<6, true, 2, true> : vk_score32_kernel<6, false, 2, true>)
		: (is_static ? vk_score32_kernel<6, true, 2> : vk_score32_kernel<6, false, 2>); break;
	}
	hipError_t e;
	if (smem > 64 * 1024) {
		e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
		if (e != hipSuccess) return e;
	}
	int occ = 0;
	e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 64 * waves, smem);
	if (e != hipSuccess) return e;
	if (occ < 1) occ = 1;
	const char *ov = getenv("VK_BLOCKS_PER_CU");   // read per launch: tools/sweep_dims.py varies it inside one process
	if (ov && atoi(ov) > 0 && atoi(ov) < occ) occ = atoi(ov);
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	const int per = four ? 1 : 2;
	const int want = (int)((((int64_t)p->n_sent + per - 1) / per + waves - 1) / waves);
	const int grid = want < cus * occ ? (want > 0 ? want : 1) : cus * occ;
	kernel<<<grid, 64 * waves, smem, stream>>>(*p, tiles * 16, strip_stride(p->len_t), slack);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------
// 1:n RWMD of a query of 17..64 tokens over the slices of more than VK_DEV_MAX_SENT_LEN tokens (the multi-block kernel above skips
// them; their groups in the slice table are padded, vk_corpus.cpp): one wave per long slice, its similarity rows [m x 64] in LDS
// (139 KB) like the rows of a short slice in the strip, then the same rwmd_fill32 four blocks wide.  Upstream sizes its problems by
// the longest sentence (metric/alignment.h:357-358); a fallback that keeps such corpora on the device, not a roofline kernel.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void vk_long_rwmd_fill_kernel(VkWideParams p, const int32_t *__restrict__ group_list, int32_t n_list, int32_t n_entries) {
	constexpr int N = 64, M = VK_DEV_MAX_LONG_LEN;
	extern __shared__ float4 vk_smem32[];
	float *S = reinterpret_cast<float *>(vk_smem32);   // [(M + 32)][N]
	float *sm = S + (M + 32) * N;                       // [M] masses of the slice's vocabulary entries (static layout)
	const int lane = threadIdx.x;
	const bool is_static = p.layout == VK_DEV_LAYOUT_STATIC;
	for (int gi = blockIdx.x; gi < n_list; gi += gridDim.x) {
		const int row = group_list[gi] * 4;
		const int t_a = p.sent_start[row], t_b = p.sent_end[row];
		const int m = t_b - t_a;
		if (lane >= 1 && lane < 4 && row + lane < n_entries) { p.scores[row + lane] = VK_NEG_INF; if (p.raw) p.raw[row + lane] = VK_NEG_INF; }
		if (m < 1 || m > M) {
			if (lane == 0) { p.scores[row] = VK_NEG_INF; if (p.raw) p.raw[row] = VK_NEG_INF; }
			continue;
		}
		int rowbase = 0;
		if (is_static) {
			for (int it = 0; it * 16 < m; it++) {
				const int tk = it * 16 + (lane >> 2);
				if (tk < m) {
					const int id = p.tok_id[t_a + tk];
					const int ps = p.pos_s ? p.pos_s[t_a + tk] : 0;
					for (int b = 0; b < p.nq; b++) {
						float4 vq = *reinterpret_cast<const float4 *>(p.table + b * p.table_stride + (int64_t)id * 16 + (lane & 3) * 4);
						if (p.pos_s) {
							const int c0 = 16 * b + (lane & 3) * 4;
							vq.x = tag_weighted(vq.x, p.tw[c0 + 0], ps, p.tpos[c0 + 0], p.tw_keep, p.tw_threshold);
							vq.y = tag_weighted(vq.y, p.tw[c0 + 1], ps, p.tpos[c0 + 1], p.tw_keep, p.tw_threshold);
							vq.z = tag_weighted(vq.z, p.tw[c0 + 2], ps, p.tpos[c0 + 2], p.tw_keep, p.tw_threshold);
							vq.w = tag_weighted(vq.w, p.tw[c0 + 3], ps, p.tpos[c0 + 3], p.tw_keep, p.tw_threshold);
						}
						*reinterpret_cast<float4 *>(S + tk * N + 16 * b + (lane & 3) * 4) = vq;
					}
				}
			}
			// masses of the vocabulary entries: repeated token ids count once, at their first position (as in the kernel above)
			const float wsum = p.rwmd_normalize_bow ? (float)m : 1.0f;
			for (int u = lane; u < m; u += 64) {
				const int id = p.tok_id[t_a + u];
				int cnt = 0;
				bool first = true;
				for (int i = 0; i < m; i++) {
					const bool same = p.tok_id[t_a + i] == id && (!p.tag_s || p.tag_s[t_a + i] == p.tag_s[t_a + u]);
					cnt += same ? 1 : 0;
					first = first && !(same && i < u);
				}
				sm[u] = first ? (float)cnt / wsum : 0.0f;
			}
			if (p.qid_bits) {
				wave_lds_fence();
				static_vocab_fixup<64>(S, N, m, p.len_t, p.tok_id + t_a, p.tag_s + t_a, p.pos_s + t_a, p.table, p.table_stride,
					p.qid_bits, p.qkey, p.tw, p.tpos, p.tw_keep, p.tw_threshold, lane);
			}
		} else {
			const int tile0 = t_a >> 4;
			const int ntiles = ((t_b + 15) >> 4) - tile0;
			for (int ti = 0; ti < ntiles; ti++) {
				const int ps = p.pos_s ? p.pos_s[(tile0 + ti) * 16 + (lane & 15)] : 0;
				for (int b = 0; b < p.nq; b++) {
					f32x4 acc = sim_tile_generic(p.qtile + (int64_t)b * p.tile_bytes, p.tiles + (int64_t)(tile0 + ti) * p.tile_bytes, p.nk32, p.tail, lane, p.prec);
					if (p.pos_s) {
						const int c0 = 16 * b + (lane >> 4) * 4;
#pragma unroll
						for (int r = 0; r < 4; r++) acc[r] = tag_weighted(acc[r], p.tw[c0 + r], ps, p.tpos[c0 + r], p.tw_keep, p.tw_threshold);
					}
					*reinterpret_cast<f32x4 *>(S + (ti * 16 + (lane & 15)) * N + 16 * b + (lane >> 4) * 4) = acc;
				}
			}
			rowbase = t_a - tile0 * 16;
		}
		wave_lds_fence();
		const float raw = rwmd_fill32<4>(S, N, is_static ? sm : nullptr, rowbase, m, m, lane, lane, p);
		if (lane == 63) {
			const float boost = p.boost ? p.boost[row] : 1.0f;
			p.scores[row] = (raw / p.ref_total) * boost;
			if (p.raw) p.raw[row] = raw;
		}
		wave_lds_fence();
	}
}

extern "C" hipError_t vk_launch_long_rwmd_fill(const VkWideParams *p, const int32_t *group_list, int32_t n_list, int32_t n_entries, hipStream_t stream) {
	if (n_list < 1) return hipSuccess;
	const size_t smem = ((size_t)(VK_DEV_MAX_LONG_LEN + 32) * 64 + VK_DEV_MAX_LONG_LEN) * 4;
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(vk_long_rwmd_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
	if (e != hipSuccess) return e;
	const int blocks = n_list < 2048 ? n_list : 2048;
	vk_long_rwmd_fill_kernel<<<blocks, 64, smem, stream>>>(*p, group_list, n_list, n_entries);
	return hipGetLastError();
}
