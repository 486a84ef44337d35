// vk_doc.hip -- long slices (65 tokens .. whole documents) under a query of at most 16 tokens (round 4): a skewed sweep without an
// in-row dependency for linear, affine and (saturating) general gaps; the relaxed 1:1 word mover's distance streamed tile by tile.
#include <type_traits>
#include "vk_common.hip.h"

// ---------------------------------------------------------------------------
// vk_wide_kernel walks a document row by row: every row waits for the row before it AND, inside the row, for the in-row gaps
// (a decayed prefix maximum over the query's columns, four dependent DPP steps) -- 0.27 us per row, and a pass over a corpus of
// documents lasts as long as its longest document's chain (5,000 tokens = 1.34 ms; the traceback of a winner, which
// re-runs the recurrence in the oracle's order column by column, 3.6 ms).  Skewed by one row per query column the in-row dependency
// disappears: on step d lane v - 1 holds cell (u = d - v, v), and
//     H[u-1][v]   is the lane's own value of the step before                     (gap over s; E of the affine solver likewise)
//     H[u][v-1]   is lane v - 2's value of the step before: one DPP row_shr:1     (gap over t; F likewise: a second shift)
//     H[u-1][v-1] is what that shift delivered one step earlier
// -- a step is ~10 vector instructions with a dependent chain of four, a document of n tokens takes n + len_t steps, and every cell
// meets its candidates in the oracle's order (zero, diagonal, gap over s, gap over t; replaced on strictly greater), so the SAME
// sweep serves the scoring pass (MFMA similarities, 16 rows per tile, two tiles ahead in an LDS ring) and the winners' tracebacks
// (rows restated canonically beforehand by vk_canon_rows_kernel; one byte per cell -- direction, E / F extended -- in a scratch
// region; start cell = first maximum in row-major order; lane 0 walks back).  Queries of at most 16 tokens.
// General gaps (GAP 2; tables constant from some k <= 126 on, others keep vk_wide_kernel): the same sweep on VALUES only -- the column
// history in a doubled LDS ring read at immediate offsets, the four DPP rows scanning a quarter of the candidates each, the gaps of
// two and more gathered one step ahead; FLOW stores H and the whole wave walks back (comments at the code).
// GAP 4: the relaxed 1:1 WMD has no recurrence; its arm consumes the tiles where the MFMA leaves them.
// One document per wave, DP in the 16 lanes of DPP row 0 (the other rows repeat it: their lanes are needed for the tiles anyway).
// A wave alone on its SIMD pays ~10 cycles per instruction it issues and a pass lasts as long as its longest document's chain: between
// two tile boundaries, while every query column is inside the document, sixteen steps run without masks, borders or bookkeeping.
// ---------------------------------------------------------------------------

#define VK_DOC_RING 64   // rows of the LDS ring: four tiles (one being consumed, the next, the one being written, slack)

// FLOW: one record per cell -- a byte under linear / affine gaps (direction, E / F extended), a dword under general gaps (+ gap length)
extern "C" size_t vk_doc_scratch_bytes(int32_t max_len, int32_t gap_mode) { return ((size_t)(max_len + 2) * 16 * (gap_mode == 2 ? 4 : 1) + 255) / 256 * 256; }

__device__ __forceinline__ uint32_t doc_orderable(float f) {
	const uint32_t u = __builtin_bit_cast(uint32_t, f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float doc_left(float x, float border) {   // value of lane v - 2 within the DPP row (lane 0 of a row: `border`)
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, border), __builtin_bit_cast(int, x), DPP_ROW_SHR1, 0xf, 0xf, false));
}

// SRC: where the similarities come from, at compile time (a runtime choice made the compiler wait for EVERY outstanding load at the
// top of every step: the registers of one path's loads were the temporaries of another's) -- 0: contextual tiles, bf16 rows of up to 12
// K-steps, loaded into registers a boundary ahead; 1: contextual tiles, any row (the MFMA sequence loads them); 2: the static layout's
// table, gathered by token id; 3: FLOW (the restated rows)
template <bool FLOW, int GAP, int SRC>
__global__ __launch_bounds__(64) void vk_doc_kernel(VkWideParams p) {
	// S[token & 63][query column]: what the DP runs on (tag weights applied).  Linear / affine gaps: every row a second time 64 rows up, so
	// that the sixteen steps between two tile boundaries read their rows at immediate offsets from one base (no wrap inside a block)
	constexpr bool RING2 = GAP == 0 || GAP == 1;
	__shared__ float ring[VK_DOC_RING * 16 * (RING2 ? 2 : 1)];
	__shared__ float twl[16];
	__shared__ int tposl[16];
	__shared__ int16_t mapl[16];
	// general gaps (GAP 2; a gap table that saturates: w_s(k) = w_s(T) for k >= T = p.ws_tail <= 126).  The sweep keeps VALUES only (a
	// max over the candidates; FLOW stores H, and the traceback finds which candidate the reference's scan would have kept):
	//   Hr2  the column history, H[row & 127][v], every row stored twice (at r and r + 128) so that the rows u - 1 .. u - 128 lie at
	//        descending addresses without a wrap: one base per step, immediate offsets (rows before the document: -inf)
	//   Ht2  the lanes' values of the last 16 steps, likewise doubled, 17 columns: column c >= 1 = lane c - 1, column 0 = the border
	//        H[step][0]; H[u][v - k] is column v - k of step d - k, H[u - 1][v - 1] column v - 1 of step d - 2
	//   wq   w_s dealt to the four DPP rows (row r scans k = r + 1 + 4 i): wq[r][i] = w_s(k) for k < T, else +inf (no candidate)
	__shared__ float Hr2[GAP == 2 ? 256 * 16 : 1];
	__shared__ float Htb[GAP == 2 ? 16 + 32 * 17 : 1];
	__shared__ float wq[GAP == 2 ? 4 * 32 : 1];
	__shared__ float wsl[GAP == 2 ? 128 : 1];
	__shared__ float wtl[GAP == 2 ? 32 : 1];
	float *Ht2 = Htb + (GAP == 2 ? 16 : 0);
	const int lane = threadIdx.x;
	if (lane < 16) { twl[lane] = p.tw[lane]; tposl[lane] = p.tpos[lane]; }
	if constexpr (GAP == 2) {
		wsl[lane] = p.ws[lane]; wsl[64 + lane] = p.ws[64 + lane];
		if (lane < 32) wtl[lane] = p.wt[lane < 17 ? lane : 16];
		for (int j = lane; j < 128; j += 64) { const int k = (j >> 5) + 1 + 4 * (j & 31); wq[j] = (k >= 2 && k < p.ws_tail) ? p.ws[k] : __builtin_inff(); }
		for (int j = lane; j < 16 + 32 * 17; j += 64) Htb[j] = 0.0f;
	}
	wave_lds_fence();
	const int v = (lane & 15) + 1, len_t = p.len_t;
	const bool col = v <= len_t;
	constexpr bool is_static = SRC == 2;
	const bool static_layout = p.layout == VK_DEV_LAYOUT_STATIC;   // (FLOW: the edges' unmodified similarities)
	const bool local = p.locality == VK_DEV_LOCAL, global = p.locality == VK_DEV_GLOBAL;
	const float gs = p.gs, gt = p.gt, a_s = p.a_s, a_t = p.a_t, open_s = p.open_s, open_t = p.open_t;
	uint8_t *D = FLOW ? p.scratch + (int64_t)blockIdx.x * p.scratch_stride : nullptr;   // FLOW: D[u * 16 + v - 1]
	float *Hs = reinterpret_cast<float *>(D);                                            // ... general gaps: H[u * 16 + v - 1]
	const int T = GAP == 2 ? p.ws_tail : 0;
	const int T2 = T > 2 ? T : 2;   // the running maximum of the tail starts two rows back at the earliest (one row back: the lane's own last value)
	const int r4 = lane >> 4;
	const int n_chunks = (((T + 2) >> 2) + 7) >> 3;   // general gaps: chunks of 8 candidates per DPP row (i < ceil((T - 1) / 4))
	float wtr[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // general gaps: this lane's four gaps over t (k = r4 + 1 + 4 i <= v)
	float wsT = 0.0f, ws1 = 0.0f, wt1 = 0.0f;
	typedef float f2w __attribute__((ext_vector_type(2)));
	f2w wr[GAP == 2 ? 16 : 1];   // ... and its 32 gaps over s (k = r4 + 1 + 4 i), in pairs
	if constexpr (GAP == 2) {
#pragma unroll
		for (int i = 0; i < 16; i++) { wr[i].x = wq[r4 * 32 + 2 * i]; wr[i].y = wq[r4 * 32 + 2 * i + 1]; }
#pragma unroll
		for (int i = 0; i < 4; i++) { const int k = r4 + 1 + 4 * i; wtr[i] = (k >= 2 && k <= v) ? wtl[k] : __builtin_inff(); }
		wsT = wsl[T];
		ws1 = wsl[1]; wt1 = wtl[1];
	}

	// contextual scoring over bf16 rows of up to 12 K-steps: the query's A fragments in registers for the whole launch
	constexpr int NKP = 12;
	const int nfull = p.tail ? p.nk32 - 1 : p.nk32;
	constexpr bool regs = SRC == 0;
	bf16x8 qf[NKP], qh = {0, 0, 0, 0, 0, 0, 0, 0};
	if constexpr (regs) {
#pragma unroll
		for (int i = 0; i < NKP; i++) qf[i] = *reinterpret_cast<const bf16x8 *>(p.qtile + (i < nfull ? i : 0) * 1024 + lane * 16);
		qh = load_half_block(p.qtile + (p.tail ? nfull : 0) * 1024, lane, false);
	}

	const int64_t n_items = FLOW ? (int64_t)gridDim.x : (int64_t)p.n_order;
	// (the work list is sorted longest first: rounds deal it back and forth over the workgroups, so that every one gets long and short)
	for (int64_t round = 0; round * gridDim.x < n_items; round++) {
		const int64_t item = round * gridDim.x + ((round & 1) ? gridDim.x - 1 - blockIdx.x : blockIdx.x);
		if (item >= n_items) continue;
		int64_t g;
		if (FLOW) {
			const uint64_t key = p.keys[item];
			if (key == 0) return;   // fewer than k admitted
			g = (int64_t)(uint32_t)(key & 0xffffffffu);
		} else g = (int64_t)p.order[item];
		const int t_a = p.sent_start[g], t_b = p.sent_end[g];
		const int len_s = t_b - t_a;
		if (len_s < 1) {
			if (!FLOW && lane == 0) { p.scores[g] = VK_NEG_INF; if (p.raw) p.raw[g] = VK_NEG_INF; }
			if (FLOW) { p.mapping[item * 64 + lane] = -1; p.edge_sim[item * 64 + lane] = 0.0f; if (lane == 0) p.raw_out[item] = 0.0f; }
			continue;
		}
		// ---- the rows of 16 tokens ("tiles" by absolute token index: the contextual layout's tiles; static / FLOW: the same grid)
		// produced into the ring ahead of their use.
		const int k_first = t_a >> 4, k_last = (t_b - 1) >> 4;
		// Pipeline of a tile k (16 tokens), in boundaries (a boundary = lane v = 1 enters a new tile, every 16 steps):
		//   static / FLOW: the lane's 16 bytes are requested three boundaries before use (two staging registers), written one before;
		//   contextual scoring: the tile's K-steps are loaded into registers two boundaries ahead (bf16 rows of up to 12 K-steps, the
		//   query's fragments stay in registers for the whole launch; wider rows and fp32 rows: loaded by the MFMA sequence itself),
		//   multiplied and written one ahead.  (First form: loads and MFMAs in one go at the boundary -- ten dependent round trips per
		//   tile, 0.36 us per row.)
		float4 st4a = {0.0f, 0.0f, 0.0f, 0.0f}, st4b = st4a;
		int st_psa = 0, st_psb = 0;
		bf16x8 xn[NKP], xh = {0, 0, 0, 0, 0, 0, 0, 0};
		auto request = [&](int k, float4 &st4, int &st_ps) {   // static / FLOW: tile k's 16 bytes of this lane (row lane >> 2, columns 4 (lane & 3) ..)
			if (k > k_last) return;
			const int tok = 16 * k + (lane >> 2), c4 = (lane & 3) * 4;
			const bool in = tok >= t_a && tok < t_b;
			if constexpr (FLOW) {
				st4 = in ? *reinterpret_cast<const float4 *>(p.dp_rows + ((int64_t)item * p.dp_rows_len + (tok - t_a)) * 16 + c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			} else {
				const int id = in ? p.tok_id[tok] : 0;
				st4 = *reinterpret_cast<const float4 *>(p.table + (int64_t)id * 16 + c4);
				st_ps = (p.pos_s && in) ? p.pos_s[tok] : 0;
			}
		};
		auto st_values = [&](float4 x, int st_ps) -> float4 {   // ... with the tag weights applied
			const int c4 = (lane & 3) * 4;
			if (!FLOW && p.pos_s) {   // (FLOW: the restated rows carry the tag weights already)
				x.x = tag_weighted(x.x, twl[c4 + 0], st_ps, tposl[c4 + 0], p.tw_keep, p.tw_threshold);
				x.y = tag_weighted(x.y, twl[c4 + 1], st_ps, tposl[c4 + 1], p.tw_keep, p.tw_threshold);
				x.z = tag_weighted(x.z, twl[c4 + 2], st_ps, tposl[c4 + 2], p.tw_keep, p.tw_threshold);
				x.w = tag_weighted(x.w, twl[c4 + 3], st_ps, tposl[c4 + 3], p.tw_keep, p.tw_threshold);
			}
			return x;
		};
		auto write_st = [&](int k, float4 x, int st_ps) {
			if (k > k_last) return;
			const int tok = 16 * k + (lane >> 2), c4 = (lane & 3) * 4;
			x = st_values(x, st_ps);
			*reinterpret_cast<float4 *>(ring + (tok & (VK_DOC_RING - 1)) * 16 + c4) = x;
			if constexpr (RING2) *reinterpret_cast<float4 *>(ring + ((tok & (VK_DOC_RING - 1)) + VK_DOC_RING) * 16 + c4) = x;
		};
		auto tile_load = [&](int k) {   // contextual, `regs`: the K-steps of tile k into registers
			if (k > k_last) return;
			const uint8_t *tp = p.tiles + (int64_t)k * p.tile_bytes;
#pragma unroll
			for (int i = 0; i < NKP; i++)   // (K-steps the row does not have re-read its first one: unconditional loads)
				xn[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tp + (i < nfull ? i : 0) * 1024 + lane * 16));
			xh = load_half_block(tp + (p.tail ? nfull : 0) * 1024, lane, true);
		};
		auto tile_values = [&](int k) -> f32x4 {   // contextual: S of tile k (from the registers, or loaded here), tag weights applied
			f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
			if constexpr (regs) {
#pragma unroll
				for (int i = 0; i < NKP; i++)
					if (i < nfull) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[i], xn[i], acc, 0, 0, 0);
				if (p.tail) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh, xh, acc, 0, 0, 0);
				acc[0] = clip01(acc[0]); acc[1] = clip01(acc[1]); acc[2] = clip01(acc[2]); acc[3] = clip01(acc[3]);
			} else acc = sim_tile_generic(p.qtile, p.tiles + (int64_t)k * p.tile_bytes, p.nk32, p.tail, lane, p.prec);   // lane: S[token lane & 15][query 4 (lane >> 4) + r]
			const int tok = 16 * k + (lane & 15), c0 = (lane >> 4) * 4;
			if (p.pos_s) {
				const int ps = (tok >= t_a && tok < t_b) ? p.pos_s[tok] : 0;
#pragma unroll
				for (int r = 0; r < 4; r++) acc[r] = tag_weighted(acc[r], twl[c0 + r], ps, tposl[c0 + r], p.tw_keep, p.tw_threshold);
			}
			return acc;
		};
		auto tile_write = [&](int k) {   // ... into the ring
			if (k > k_last) return;
			const f32x4 acc = tile_values(k);
			const int tok = 16 * k + (lane & 15), c0 = (lane >> 4) * 4;
			*reinterpret_cast<f32x4 *>(ring + (tok & (VK_DOC_RING - 1)) * 16 + c0) = acc;
			if constexpr (RING2) *reinterpret_cast<f32x4 *>(ring + ((tok & (VK_DOC_RING - 1)) + VK_DOC_RING) * 16 + c0) = acc;
		};
		if constexpr (GAP == 4) {
			// ---- relaxed 1:1 word mover's distance (rwmd_rows of vk_score_kernel; vk_wide_kernel's gap == 4 arm): no recurrence, so
			// the tiles are consumed where the MFMA (or the table gather) leaves them -- a lane holds four query columns of one token:
			// column minima stay in the lane until the document ends, a token's minimum over the columns crosses the four lanes that
			// share it (lane swaps / quad shifts), the sums are per-lane partial sums.
			const float BIG = 3.402823466e+38F;
			const bool nbow = p.rwmd_normalize_bow != 0, sym = p.rwmd_symmetric != 0;
			const float w_t = nbow ? 1.0f / (float)len_t : 1.0f, w_s = nbow ? 1.0f / (float)len_s : 1.0f;
			float cmin[4] = {BIG, BIG, BIG, BIG}, acc1 = 0.0f;
			auto consume = [&](float v0, float v1, float v2, float v3, int tok, int cbase, bool first_group, auto group_min) {
				const bool in = tok >= t_a && tok < t_b;
				const float val[4] = {v0, v1, v2, v3};
				float rmin = BIG;
#pragma unroll
				for (int r = 0; r < 4; r++) {
					float dist = fmaxf(1.0f - val[r], 0.0f);
					dist = (in && cbase + r < len_t) ? dist : BIG;
					cmin[r] = fminf(cmin[r], dist);
					rmin = fminf(rmin, dist);
				}
				if (sym) {
					rmin = group_min(rmin);
					acc1 += (in && first_group) ? w_s * rmin : 0.0f;
				}
			};
			if constexpr (is_static) {
				request(k_first, st4a, st_psa);
				request(k_first + 1, st4b, st_psb);
				for (int k = k_first; k <= k_last; k++) {
					const float4 x = st_values(st4a, st_psa);
					st4a = st4b; st_psa = st_psb;
					request(k + 2, st4b, st_psb);
					consume(x.x, x.y, x.z, x.w, 16 * k + (lane >> 2), (lane & 3) * 4, (lane & 3) == 0,
						[&](float m) { m = fminf(m, __shfl_xor(m, 1, 64)); return fminf(m, __shfl_xor(m, 2, 64)); });
				}
			} else {
				if constexpr (regs) tile_load(k_first);
				for (int k = k_first; k <= k_last; k++) {
					const f32x4 acc = tile_values(k);
					if constexpr (regs) tile_load(k + 1);
					consume(acc[0], acc[1], acc[2], acc[3], 16 * k + (lane & 15), (lane >> 4) * 4, lane < 16,
						[&](float m) { m = fminf(m, lane_xor16(m, lane)); return fminf(m, lane_xor32(m, lane)); });
				}
			}
			// the column minima over all tokens: across the lanes that hold the same columns
#pragma unroll
			for (int r = 0; r < 4; r++) {
#pragma unroll
				for (int o = is_static ? 4 : 1; o <= (is_static ? 32 : 8); o <<= 1) cmin[r] = fminf(cmin[r], __shfl_xor(cmin[r], o, 64));
			}
			float acc0 = 0.0f;
#pragma unroll
			for (int j = 0; j < 16; j++) {
				if (j < len_t) {
					const float xj = w_t * __shfl(cmin[j & 3], is_static ? (j >> 2) : 16 * (j >> 2), 64);
					acc0 = j == 0 ? xj : acc0 + xj;
				}
			}
#pragma unroll
			for (int o = 1; o <= 32; o <<= 1) acc1 += __shfl_xor(acc1, o, 64);
			if (!nbow) { acc0 = acc0 / (float)len_t; acc1 = acc1 / (float)len_s; }
			float cost = 0.0f;
			if (sym) { if (acc0 > cost) cost = acc0; if (acc1 > cost) cost = acc1; }
			else cost = acc0;
			const float max_cost = nbow ? 1.0f : (float)len_t;
			const float raw4 = (max_cost - cost) / max_cost;
			if (lane == 0) {
				const float boost = p.boost ? p.boost[g] : 1.0f;
				p.scores[g] = (raw4 / p.ref_total) * boost;
				if (p.raw) p.raw[g] = raw4;
			}
			continue;
		}
		// before the sweep: tiles k_first, k_first + 1 in the ring; k_first + 2 (and + 3: static / FLOW) requested
		if constexpr (FLOW || is_static) {
			request(k_first, st4a, st_psa); write_st(k_first, st4a, st_psa);
			request(k_first + 1, st4a, st_psa); write_st(k_first + 1, st4a, st_psa);
			request(k_first + 2, st4a, st_psa);
			request(k_first + 3, st4b, st_psb);
		} else {
			if constexpr (regs) { tile_load(k_first); tile_write(k_first); tile_load(k_first + 1); tile_write(k_first + 1); tile_load(k_first + 2); }
			else { tile_write(k_first); tile_write(k_first + 1); }
		}
		// boundary work when lane v = 1 enters tile kc: tile kc + 1 into the ring, the tiles behind it requested
		auto boundary = [&](int kc) {
			if constexpr (FLOW || is_static) {
				write_st(kc + 1, st4a, st_psa);
				st4a = st4b; st_psa = st_psb;
				request(kc + 3, st4b, st_psb);
			} else {
				tile_write(kc + 1);
				if constexpr (regs) tile_load(kc + 2);
			}
			wave_lds_fence();
		};
		if constexpr (GAP == 2) {   // the column history before the document: no candidates
			const float4 ninf = {VK_NEG_INF, VK_NEG_INF, VK_NEG_INF, VK_NEG_INF};
#pragma unroll
			for (int j = 0; j < 16; j++) reinterpret_cast<float4 *>(Hr2)[lane + 64 * j] = ninf;
		}
		wave_lds_fence();

		// ---- the sweep: a1 = H[u - 1][v] (this lane's last value; before its first row the border H[0][v])
		auto border_s = [&](int k) -> float {   // H[k][0]
			if (!global || k <= 0) return 0.0f;
			if constexpr (GAP == 2) return -(k < T ? wsl[k] : wsl[T]);
			return GAP == 0 ? -(gs * (float)k) : -(a_s + gs * (float)k);
		};
		float b_t = 0.0f;   // H[0][v]
		if (global) { if constexpr (GAP == 2) b_t = -wtl[v]; else b_t = GAP == 0 ? -(gt * (float)v) : -(a_t + gt * (float)v); }
		float a1 = b_t;
		float m_far = VK_NEG_INF;    // general gaps: the best candidate of length two and more of the coming step (the first cell has none)
		float tail_m = VK_NEG_INF;   // general gaps: max of H[u'][v] over u' <= u - T (the candidates at and beyond the table's tail: all cost w_s(T))
		if constexpr (GAP == 2) {   // what the first steps read of the steps before them: H[0][0], H[1][0]; H[0][1] (lane v = 1's border)
			if (lane == 0) {
				Ht2[0] = 0.0f; Ht2[16 * 17] = 0.0f;
				const float b1 = border_s(1);
				Ht2[17] = b1; Ht2[17 * 17] = b1;
				Ht2[17 + 1] = b_t; Ht2[17 * 17 + 1] = b_t;
				Hr2[0] = b_t; Hr2[128 * 16] = b_t;
			}
			wave_lds_fence();
		}
		float e1 = VK_NEG_INF, f1 = VK_NEG_INF;   // affine: E[u - 1][v], F[u][v] of this lane's last step
		float prev_left = 0.0f;                     // H[u - 1][v - 1]: last step's `left` (lane 0, step 2: H[0][0] = 0)
		float best_v = 0.0f;
		int best_u = 0;
		const int steps_end = len_s + len_t;
		// this lane's similarity of the NEXT step is read one step ahead (its row was written at least a tile ago): the LDS round trip
		// stays off the dependent chain
		// (any ring slot exists: rows outside the document read as whatever lies there and are never used)
		auto s_of = [&](int u) -> float { return ring[((t_a + u - 1) & (VK_DOC_RING - 1)) * 16 + (v - 1)]; };
		float s_next = s_of(2 - v);
		// General gaps: the gaps of two and more of a coming step (row u1, step d1): nothing of the current cell enters them, so their ~110
		// instructions are off the chain from cell to cell.  Gaps over s inside the table: this DPP row's share, k = r4 + 1 + 4 i (i < 32;
		// k = 1 and k >= T: +inf in wr), row u1 - k at hp[(31 - i) * 64]; one form per number of 8-candidate chunks the table needs (NC), each
		// a single block: all its loads (history, the tail's row, the gaps over t) leave together.  Then the four shares meet
		// (v_permlane{32,16}_swap: halves / DPP rows exchanged between two registers at VALU speed; one of the two outputs holds the
		// partner's value, so every DPP row ends with the maximum of all four and repeats row 0's chain).
		auto far_next = [&](int u1, int d1, auto nc) -> float {
			constexpr int NC = decltype(nc)::value;
			const float *hp = Hr2 + ((u1 & 127) + 3 - r4) * 16 + (v - 1);
			const float *tp = Hr2 + ((u1 & 127) + 128 - T2) * 16 + (v - 1);
			const float *gp = Ht2 + (((d1 & 15) + 16) * 17) + v - 18 * (r4 + 1);
			typedef float f2 __attribute__((ext_vector_type(2)));
			f2 hv[NC > 0 ? NC * 4 : 1];
#pragma unroll
			for (int i = 0; i < NC * 4; i++) { hv[i].x = hp[(31 - 2 * i) * 64]; hv[i].y = hp[(30 - 2 * i) * 64]; }
			const float xt = *tp;
			float ht[4];
#pragma unroll
			for (int i = 0; i < 4; i++) ht[i] = gp[-72 * i];
			float mm = VK_NEG_INF, mm2 = VK_NEG_INF;
			float mm3 = VK_NEG_INF, mm4 = VK_NEG_INF;   // (four chains of maxima; plain subtractions -- v_pk_add_f32 on the register pairs measured 7 % slower)
#pragma unroll
			for (int j = 0; j < NC; j++) {
				mm = fmaxf(mm, fmaxf(hv[4 * j].x - wr[4 * j].x, hv[4 * j].y - wr[4 * j].y));
				mm2 = fmaxf(mm2, fmaxf(hv[4 * j + 1].x - wr[4 * j + 1].x, hv[4 * j + 1].y - wr[4 * j + 1].y));
				mm3 = fmaxf(mm3, fmaxf(hv[4 * j + 2].x - wr[4 * j + 2].x, hv[4 * j + 2].y - wr[4 * j + 2].y));
				mm4 = fmaxf(mm4, fmaxf(hv[4 * j + 3].x - wr[4 * j + 3].x, hv[4 * j + 3].y - wr[4 * j + 3].y));
			}
			mm = fmaxf(mm, mm3); mm2 = fmaxf(mm2, mm4);
			// ... at T rows and more (at two and more under a constant table): one running maximum
			tail_m = fmaxf(tail_m, xt);
			mm = fmaxf(mm, tail_m - wsT);
			// gaps over t: k = r4 + 1 + 4 i, 2 <= k <= v (others: +inf in wtr)
#pragma unroll
			for (int i = 0; i < 4; i++) mm2 = fmaxf(mm2, ht[i] - wtr[i]);
			float mf = fmaxf(mm, mm2);
			float x0 = mf, x1 = mf;
			asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x0), "+v"(x1));
			mf = fmaxf(mf, fmaxf(x0, x1));
			x0 = mf; x1 = mf;
			asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x0), "+v"(x1));
			return fmaxf(mf, fmaxf(x0, x1));
		};
		// ... and sixteen steps of them from a tile boundary on while every query column is inside the document (as steps16 below: no
		// activity masks, no first row; local: no border column -- its slots hold zeros from the start).  The general step is 270
		// instructions with its masks, guarded stores and border look-ups, 140 of them this.
		auto steps16g = [&](int d0, auto loc, auto nc) {
			constexpr bool is_local = decltype(loc)::value == 0;
			const bool track = col && (is_local || (!global && v == len_t));
			const bool keep = col && lane < 16;
			for (int i = 0; i < 16; i++) {
				const int d = d0 + i, u = d - v;
				const float left = doc_left(a1, is_local ? 0.0f : border_s(d - 1));
				const float diag = prev_left;
				prev_left = left;
				const float sim = s_next;
				s_next = s_of(u + 1);
				float best = diag + sim;
				if constexpr (is_local) best = fmaxf(best, 0.0f);
				best = fmaxf(best, fmaxf(m_far, fmaxf(a1 - ws1, left - wt1)));
				const int B = ((d & 15) + 16) * 17;
				if (keep) {
					if constexpr (FLOW) Hs[u * 16 + (v - 1)] = best;
					Hr2[(u & 127) * 16 + (v - 1)] = best; Hr2[((u & 127) + 128) * 16 + (v - 1)] = best;
					Ht2[B - 16 * 17 + v] = best; Ht2[B + v] = best;
				}
				if constexpr (!is_local) {
					if (lane == 0) { const float x = border_s(d); Ht2[B - 16 * 17] = x; Ht2[B] = x; }
				}
				wave_lds_fence();
				m_far = far_next(u + 1, d + 1, nc);
				if constexpr (FLOW) {
					const bool nb = track && best > best_v;
					best_v = nb ? best : best_v;
					best_u = nb ? u : best_u;
				} else best_v = track ? fmaxf(best_v, best) : best_v;
				a1 = best;
			}
		};
		// Sixteen steps from a tile boundary on, while every query column is inside the document (d > len_t, d + 15 <= len_s: 1 <= u < len_s in
		// all lanes that hold a column): no activity masks, no border row, the block's similarities read up front at immediate offsets, the
		// locality at compile time.  (One loop of general steps: ~60 instructions and 450 cycles a step for a wave alone -- and a pass over
		// documents lasts as long as its longest document's chain.)
		auto steps16 = [&](int d0, auto loc) {
			constexpr int LOC = decltype(loc)::value;   // 0 local, 1 global, 2 semiglobal
			constexpr bool is_local = LOC == 0, is_global = LOC == 1;
			const int tok1 = t_a + d0 - 2;
			const float *sp = ring + ((tok1 + 1 - v) & (VK_DOC_RING - 1)) * 16 + (v - 1);
			float sv[17];
#pragma unroll
			for (int i = 0; i < 17; i++) sv[i] = sp[i * 16];
			const bool track = col && (is_local || (!is_global && v == len_t));
			const bool store = FLOW && col && lane < 16;
			uint8_t *dp = FLOW ? D + (d0 - v) * 16 + (v - 1) : nullptr;
#pragma unroll
			for (int i = 0; i < 16; i++) {
				const float left = doc_left(a1, border_s(d0 + i - 1));   // H[u][v - 1] (lane 0 of the row: the border column)
				const float diag = prev_left;                             // H[u - 1][v - 1]
				prev_left = left;
				float best, e = VK_NEG_INF, f = VK_NEG_INF;
				int dir = 1, ee = 0, fe = 0;
				{
					const float c = diag + sv[i];
					if constexpr (is_local) { const bool take = c > 0.0f; best = take ? c : 0.0f; if (FLOW) dir = take ? 1 : 0; }
					else best = c;
				}
				if constexpr (GAP == 0) {
					const float cu = a1 - gs, cl = left - gt;
					if constexpr (FLOW) {
						const bool tu = cu > best;
						best = tu ? cu : best; dir = tu ? 2 : dir;
						const bool tl = cl > best;
						best = tl ? cl : best; dir = tl ? 3 : dir;
					} else best = fmaxf(best, fmaxf(cu, cl));
				} else {
					const float left_f = doc_left(f1, VK_NEG_INF);   // F[u][v - 1]
					e = a1 - open_s;
					const float ce = e1 - gs;
					f = left - open_t;
					const float cf = left_f - gt;
					if constexpr (FLOW) {
						ee = ce > e ? 1 : 0; e = ce > e ? ce : e;
						fe = cf > f ? 1 : 0; f = cf > f ? cf : f;
						const bool tu = e > best;
						best = tu ? e : best; dir = tu ? 2 : dir;
						const bool tl = f > best;
						best = tl ? f : best; dir = tl ? 3 : dir;
					} else {
						e = fmaxf(e, ce); f = fmaxf(f, cf);
						best = fmaxf(best, fmaxf(e, f));
					}
					e1 = e; f1 = f;
				}
				if constexpr (FLOW) { if (store) dp[i * 16] = (uint8_t)(dir | (ee << 2) | (fe << 3)); }
				if constexpr (!is_global) {
					if constexpr (FLOW) {
						const bool nb = track && best > best_v;   // first maximum of this column
						best_v = nb ? best : best_v;
						best_u = nb ? d0 + i - v : best_u;
					} else best_v = track ? fmaxf(best_v, best) : best_v;
				}
				a1 = best;
			}
			s_next = sv[16];
		};
		for (int d = 2; d <= steps_end; d++) {
			const int u = d - v;
			const bool act = col && u >= 1 && u <= len_s;
			// boundary: the first lane (v = 1) is about to enter a new tile -> the tile after it is written, the one after that requested
			const int tok1 = t_a + d - 2;   // token of lane v = 1 on this step
			if ((tok1 & 15) == 0 && d > 2) {
				boundary(tok1 >> 4);
				if (d > len_t && d + 15 <= len_s) {
					if constexpr (GAP != 2) {
						if (local) steps16(d, std::integral_constant<int, 0>{});
						else if (global) steps16(d, std::integral_constant<int, 1>{});
						else steps16(d, std::integral_constant<int, 2>{});
					} else {
						auto go = [&](auto nc) { if (local) steps16g(d, std::integral_constant<int, 0>{}, nc); else steps16g(d, std::integral_constant<int, 1>{}, nc); };
						switch (n_chunks) {
							case 0: go(std::integral_constant<int, 0>{}); break;
							case 1: go(std::integral_constant<int, 1>{}); break;
							case 2: go(std::integral_constant<int, 2>{}); break;
							case 3: go(std::integral_constant<int, 3>{}); break;
							default: go(std::integral_constant<int, 4>{}); break;
						}
					}
					d += 15;
					continue;
				}
			}
			const int B = ((d & 15) + 16) * 17;   // general gaps: this step's (upper) slot of Ht2
			const float left = doc_left(a1, border_s(d - 1));   // H[u][v - 1] (lane 0 of the row: the border column)
			const float diag = prev_left;                       // H[u - 1][v - 1]
			prev_left = left;
			float left_f = VK_NEG_INF;
			if (GAP == 1) left_f = doc_left(f1, VK_NEG_INF);     // F[u][v - 1]
			const float s = act ? s_next : 0.0f;
			s_next = s_of(u + 1);
			// candidates in the oracle's order -- zero (LOCAL), diagonal, gap over s, gap over t; replaced on strictly greater -- as selects
			float best, e = VK_NEG_INF, f = VK_NEG_INF;
			int dir, ee = 0, fe = 0;
			{
				const float c = diag + s;
				const bool take = !local || c > 0.0f;
				best = take ? c : 0.0f;
				dir = take ? 1 : 0;
			}
			if constexpr (GAP == 2) {
				// the gaps of length 1 come from the step before, as under linear gaps (the lane's own value; its left neighbour's through the
				// DPP shift); all longer ones were gathered DURING the step before (m_far: they end in rows and steps that were complete then)
				const float m = fmaxf(m_far, fmaxf(a1 - ws1, left - wt1));
				dir = m > best ? 2 : dir;
				best = fmaxf(best, m);
			} else if (GAP == 0) {
				const float cu = a1 - gs, cl = left - gt;
				const bool tu = cu > best;
				best = tu ? cu : best; dir = tu ? 2 : dir;
				const bool tl = cl > best;
				best = tl ? cl : best; dir = tl ? 3 : dir;
			} else {
				// gap of length 1 (open) first, longer (extend) only if strictly greater (align_affine)
				e = a1 - open_s;
				const float ce = e1 - gs;
				ee = ce > e ? 1 : 0; e = ce > e ? ce : e;
				f = left - open_t;
				const float cf = left_f - gt;
				fe = cf > f ? 1 : 0; f = cf > f ? cf : f;
				const bool tu = e > best;
				best = tu ? e : best; dir = tu ? 2 : dir;
				const bool tl = f > best;
				best = tl ? f : best; dir = tl ? 3 : dir;
			}
			if constexpr (GAP == 2) {   // this cell (before the first row: the border H[0][v]) joins the histories, read from the next step on
				if (FLOW && act && lane < 16) Hs[u * 16 + (v - 1)] = best;
				if (lane < 16 && col && u >= 0 && u <= len_s) {
					const float x = u == 0 ? b_t : best;
					Hr2[(u & 127) * 16 + (v - 1)] = x; Hr2[((u & 127) + 128) * 16 + (v - 1)] = x;
					Ht2[B - 16 * 17 + v] = x; Ht2[B + v] = x;
				}
				if (lane == 0) { const float x = border_s(d); Ht2[B - 16 * 17] = x; Ht2[B] = x; }
				wave_lds_fence();
				// ---- the gaps of two and more of the NEXT step (row u + 1, step d + 1)
				switch (n_chunks) {
					case 0: m_far = far_next(u + 1, d + 1, std::integral_constant<int, 0>{}); break;
					case 1: m_far = far_next(u + 1, d + 1, std::integral_constant<int, 1>{}); break;
					case 2: m_far = far_next(u + 1, d + 1, std::integral_constant<int, 2>{}); break;
					case 3: m_far = far_next(u + 1, d + 1, std::integral_constant<int, 3>{}); break;
					default: m_far = far_next(u + 1, d + 1, std::integral_constant<int, 4>{}); break;
				}
			} else if (FLOW && act && lane < 16) D[u * 16 + (v - 1)] = (uint8_t)(dir | (ee << 2) | (fe << 3));
			{
				const bool nb = act && !global && (local || u == len_s || v == len_t) && best > best_v;   // first maximum of this column
				best_v = nb ? best : best_v;
				if (FLOW) best_u = nb ? u : best_u;
				a1 = act ? best : a1;
				if (GAP == 1) { e1 = act ? e : e1; f1 = act ? f : f1; }
			}
		}

		// ---- aligner score and start cell: the first maximum in row-major order (smallest u, then smallest v)
		float raw;
		int su = 0, sv = 0;
		if (global) {
			raw = __shfl(a1, len_t - 1, 64);
			su = len_s; sv = len_t;
		} else {
			float m = (col && lane < 16) ? best_v : 0.0f;
#pragma unroll
			for (int o = 8; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
			m = __shfl(m, 0, 64);
			raw = m;
			int cu = (lane < 16 && col && best_v == m && m > 0.0f) ? best_u : 0x7fffffff;
#pragma unroll
			for (int o = 8; o >= 1; o >>= 1) { const int x = __shfl_xor(cu, o, 64); cu = x < cu ? x : cu; }
			cu = __shfl(cu, 0, 64);
			if (cu != 0x7fffffff) {
				su = cu;
				const unsigned long long hit = __ballot(lane < 16 && col && best_v == m && best_u == cu);
				sv = __builtin_ctzll(hit) + 1;
			}
		}
		if constexpr (!FLOW) {
			if (lane == 0) {
				const float boost = p.boost ? p.boost[g] : 1.0f;
				p.scores[g] = (raw / p.ref_total) * boost;
				if (p.raw) p.raw[g] = raw;
			}
		} else {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			if (lane < 16) mapl[lane] = -1;
			wave_lds_fence();
			if constexpr (GAP == 2) {
				// the whole wave walks back over the stored H: at each cell the FIRST candidate in the reference's order (zero, diagonal,
				// gaps over s by length, gaps over t by length) whose value is the cell's -- what a scan replacing on strictly greater keeps
				auto h_at = [&](int uu, int vv) -> float {
					if (vv == 0) return border_s(uu);
					if (uu == 0) return global ? -wtl[vv] : 0.0f;
					return Hs[uu * 16 + (vv - 1)];
				};
				int cu = su, cv = sv;
				while (cu > 0 && cv > 0) {
					const float h = Hs[cu * 16 + (cv - 1)];
					if (local && !(h > 0.0f)) break;
					const float sdp = p.dp_rows[((int64_t)item * p.dp_rows_len + (cu - 1)) * 16 + (cv - 1)];
					if (h_at(cu - 1, cv - 1) + sdp == h) { if (lane == 0) mapl[cv - 1] = (int16_t)(cu - 1); cu--; cv--; continue; }
					int found = 0;
					for (int k0 = 0; k0 < cu && !found; k0 += 64) {
						const int k = k0 + lane + 1;
						bool hit = false;
						if (k <= cu) hit = h_at(cu - k, cv) - (k < T ? wsl[k] : wsT) == h;
						const unsigned long long bal = __ballot(hit);
						if (bal) found = k0 + __builtin_ctzll(bal) + 1;
					}
					if (found) { cu -= found; continue; }
					{
						const int k = lane + 1;
						bool hit = false;
						if (k <= cv) hit = h_at(cu, cv - k) - wtl[k] == h;
						const unsigned long long bal = __ballot(hit);
						if (!bal) break;   // (cannot happen: the cell's value is one of its candidates)
						cv -= __builtin_ctzll(bal) + 1;
					}
				}
			} else if (lane == 0) {
				int cu = su, cv = sv, state = 0;
				while (cu > 0 && cv > 0) {
					const int rec = (int)D[cu * 16 + (cv - 1)];
					if (GAP == 1 && state == 1) { if (!(rec & 4)) state = 0; cu--; continue; }
					if (GAP == 1 && state == 2) { if (!(rec & 8)) state = 0; cv--; continue; }
					const int dd = rec & 3;
					if (dd == 0) break;
					if (dd == 1) { mapl[cv - 1] = (int16_t)(cu - 1); cu--; cv--; }
					else if (GAP == 1) state = dd == 2 ? 1 : 2;
					else if (dd == 2) cu--;
					else cv--;
				}
			}
			wave_lds_fence();
			const int mine = lane < 16 ? mapl[lane] : -1;
			float es = 0.0f;
			if (mine >= 0) {
				if (!p.pos_s) es = p.dp_rows[((int64_t)item * p.dp_rows_len + mine) * 16 + lane];   // the unmodified similarity of the edge (metric/alignment.h:339)
				else {
					// with tag weights the restated rows are the modified ones: this cell's cosine once more, canonically
					float o1[1];
					const int tok = t_a + mine;
					if (static_layout) static_sim_canon<1>(p.tiles, p.tile_bytes, p.tok_id[tok], p.qtile, lane, p.d, p.prec, p.q_ids, o1);
					else { sim_canon<1>(p.tiles + (int64_t)(tok >> 4) * p.tile_bytes, tok & 15, p.qtile, lane, p.d, p.prec, o1); o1[0] = clip01(o1[0]); }
					es = o1[0];
				}
			}
			p.mapping[item * 64 + lane] = (int16_t)mine;
			p.edge_sim[item * 64 + lane] = es;
			if (lane == 0) p.raw_out[item] = raw;
		}
		wave_lds_fence();   // the next document overwrites the ring
	}
}

// flow_k == 0: scores of the p->n_order slices of p->order (longest first; grid stride); flow_k > 0: the flow_k winners of p->keys,
// their rows in p->dp_rows, one scratch region of p->scratch_stride >= vk_doc_scratch_bytes(max_len) bytes per winner
extern "C" hipError_t vk_launch_doc(const VkWideParams *p, int32_t flow_k, hipStream_t stream) {
	if (p->len_t > 16 || p->gap_mode < 0 || (p->gap_mode > 2 && !(p->gap_mode == 4 && flow_k == 0))) return hipErrorInvalidValue;
	if (p->gap_mode == 2 && (p->ws_tail < 1 || p->ws_tail > 126)) return hipErrorInvalidValue;   // (the history ring holds 128 rows)
	if (flow_k > 0) {
		if (!p->dp_rows || !p->scratch || p->scratch_stride < (int64_t)vk_doc_scratch_bytes(p->max_len, p->gap_mode)) return hipErrorInvalidValue;
		if (p->gap_mode == 0) vk_doc_kernel<true, 0, 3><<<flow_k, 64, 0, stream>>>(*p);
		else if (p->gap_mode == 1) vk_doc_kernel<true, 1, 3><<<flow_k, 64, 0, stream>>>(*p);
		else vk_doc_kernel<true, 2, 3><<<flow_k, 64, 0, stream>>>(*p);
		return hipGetLastError();
	}
	if (!p->order || p->n_order < 1) return hipSuccess;
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	// (general gaps: 23 KB of LDS a wave, six resident per CU, the others follow as those end -- measured against one wave per SIMD over
	// a dealt list, 3.9 ms against 4.5 ms on 2,000 documents: a wave issues at most one instruction every four cycles, two fill the gaps)
	const int64_t cap = (int64_t)cus * 16;
	const int grid = (int)(p->n_order < cap ? p->n_order : cap);
	const int src = p->layout == VK_DEV_LAYOUT_STATIC ? 2 : (p->prec == 0 && p->nk32 <= 12) ? 0 : 1;
	void (*kernel)(VkWideParams);
	if (p->gap_mode == 0) kernel = src == 2 ? vk_doc_kernel<false, 0, 2> : src == 0 ? vk_doc_kernel<false, 0, 0> : vk_doc_kernel<false, 0, 1>;
	else if (p->gap_mode == 1) kernel = src == 2 ? vk_doc_kernel<false, 1, 2> : src == 0 ? vk_doc_kernel<false, 1, 0> : vk_doc_kernel<false, 1, 1>;
	else if (p->gap_mode == 4) kernel = src == 2 ? vk_doc_kernel<false, 4, 2> : src == 0 ? vk_doc_kernel<false, 4, 0> : vk_doc_kernel<false, 4, 1>;   // relaxed 1:1 WMD
	else kernel = src == 2 ? vk_doc_kernel<false, 2, 2> : vk_doc_kernel<false, 2, 1>;   // (the gap scans want the registers the query's fragments would take)
	// (fewer waves per CU -- dynamic LDS nobody uses -- only slow it down: general gaps 3.8 / 4.3 / 5.9 ms at 6 / 4 / 3 waves per CU)
	kernel<<<grid, 64, 0, stream>>>(*p);
	return hipGetLastError();
}
