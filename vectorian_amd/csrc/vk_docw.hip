// vk_docw.hip -- long slices (65 tokens .. whole documents) under a query of 17 .. 64 tokens (a sentence as the query; round 4):
// vk_doc_kernel's skewed sweep across the whole wave.  Linear and affine gaps; scoring pass and the winners' tracebacks.
#include <type_traits>
#include "vk_common.hip.h"

// ---------------------------------------------------------------------------
// Lane v - 1 = query column v (1 .. 16 NQ, NQ = 2 .. 4 blocks of 16), step d <-> row u = d - v: as in vk_doc.hip a cell's three
// neighbours are the lane's own last value, its left neighbour's last value (one DPP wave_shr:1 -- the shift crosses the DPP rows) and
// what that shift delivered a step earlier; every cell meets its candidates in the oracle's order.  vk_wide_kernel, which these queries
// took over long slices until now, walks a document row by row with a serial chain through up to 64 columns inside each row: 2,000
// documents of 500 - 5,000 tokens under a 20-token query 6.8 ms (40 tokens: 12.1 ms) for the scoring pass, as much again for ten
// tracebacks.
// Similarities: a tile of 16 tokens is multiplied with all NQ query tiles when lane v = 1 is one tile away from it, and written to one
// LDS ring PER BLOCK of 16 columns -- block b consumes a token 16 b steps after block 0, so its ring holds 64 (b < 2) or 128 tokens;
// each ring repeats its first 16 rows behind its last, so that the seventeen rows a lane reads between two tile boundaries lie at
// immediate offsets from one base.  Between two boundaries, while every query column is inside the document, sixteen steps run without
// masks, border values or bookkeeping (a wave alone on its SIMD pays for every instruction it issues: DESIGN 10.13).
// FLOW: the winners' rows restated canonically beforehand (vk_canon_rows_kernel, [len][16 NQ]), one byte per cell in a scratch region,
// start cell = first maximum in row-major order, lane 0 walks back.
// ---------------------------------------------------------------------------

__device__ __forceinline__ float docw_left(float x, float border) {   // value of lane - 1 (lane 0: `border`)
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, border), __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, false));
}

template <int NQ> struct DocwRings {
	static constexpr int R0 = 64 + 16, R2 = 128 + 16;
	static constexpr int rows = NQ == 2 ? 2 * R0 : NQ == 3 ? 2 * R0 + R2 : 2 * R0 + 2 * R2;
	__device__ static constexpr int base(int b) { return b == 0 ? 0 : b == 1 ? R0 : b == 2 ? 2 * R0 : 2 * R0 + R2; }
	__device__ static constexpr int mask(int b) { return b < 2 ? 63 : 127; }
};

extern "C" size_t vk_docw_scratch_bytes(int32_t max_len, int32_t nq) { return ((size_t)(max_len + 2) * 16 * (size_t)nq + 255) / 256 * 256; }

// SRC: 0 contextual tiles, bf16 rows of up to 12 K-steps: a tile's K-steps are loaded into registers one boundary ahead and multiplied
// with the query tiles' fragments (from L2: 48 NQ registers would not fit) at the next; 1 contextual tiles, any row type (the MFMA
// sequence loads them); 2 the static layout's per-block tables gathered by token id; 3 FLOW (the restated rows)
template <bool FLOW, int GAP, int SRC, int NQ>
__global__ __launch_bounds__(64) void vk_docw_kernel(VkWideParams p) {
	using RG = DocwRings<NQ>;
	constexpr int W = 16 * NQ;
	__shared__ float ring[RG::rows * 16];
	__shared__ float twl[64];
	__shared__ int tposl[64];
	__shared__ int16_t mapl[64];
	const int lane = threadIdx.x;
	twl[lane] = p.tw[lane]; tposl[lane] = p.tpos[lane];
	wave_lds_fence();
	const int v = lane + 1, v16 = lane & 15, blk = lane >> 4, len_t = p.len_t;
	const bool col = v <= len_t;
	const bool static_layout = p.layout == VK_DEV_LAYOUT_STATIC;   // (FLOW: the edges' unmodified similarities)
	const bool local = p.locality == VK_DEV_LOCAL, global = p.locality == VK_DEV_GLOBAL;
	const float gs = p.gs, gt = p.gt, a_s = p.a_s, a_t = p.a_t, open_s = p.open_s, open_t = p.open_t;
	uint8_t *D = FLOW ? p.scratch + (int64_t)blockIdx.x * p.scratch_stride : nullptr;   // FLOW: D[u * W + v - 1]
	// this lane's ring (rows of 16 floats: the block's 16 columns of one token)
	const int rbase = blk == 0 ? RG::base(0) : blk == 1 ? RG::base(1) : blk == 2 ? RG::base(2) : RG::base(3);
	const int rmask = blk < 2 ? 63 : 127;
	const float *myring = ring + rbase * 16 + v16;

	const int64_t n_items = FLOW ? (int64_t)gridDim.x : (int64_t)p.n_order;
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
		const int k_first = t_a >> 4, k_last = (t_b - 1) >> 4;
		constexpr int NKP = 12;
		const int nfull = p.tail ? p.nk32 - 1 : p.nk32;
		bf16x8 xn[SRC == 0 ? NKP : 1], xh = {0, 0, 0, 0, 0, 0, 0, 0};
		auto tile_load = [&](int k) {   // SRC 0: the K-steps of tile k into registers
			if constexpr (SRC == 0) {
				if (k > k_last) return;
				const uint8_t *tp = p.tiles + (int64_t)k * p.tile_bytes;
#pragma unroll
				for (int i = 0; i < NKP; i++)   // (K-steps the row does not have re-read its first one: unconditional loads)
					xn[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tp + (i < nfull ? i : 0) * 1024 + lane * 16));
				xh = load_half_block(tp + (p.tail ? nfull : 0) * 1024, lane, true);
			}
		};
		// S of tile k (tokens 16 k ..) x block b of query columns, tag weights applied: SRC 0 / 1 -- this lane: token lane & 15, columns
		// 4 (lane >> 4) + r; SRC 2 / 3 -- token lane >> 2, columns 4 (lane & 3) + r
		auto tile_values = [&](int k, int b) -> f32x4 {
			f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
			if constexpr (SRC == 0 || SRC == 1) {
				if constexpr (SRC == 0) {
					const uint8_t *qb = p.qtile + (int64_t)b * p.tile_bytes;
					bf16x8 qf[NKP];
#pragma unroll
					for (int i = 0; i < NKP; i++) qf[i] = *reinterpret_cast<const bf16x8 *>(qb + (i < nfull ? i : 0) * 1024 + lane * 16);
					const bf16x8 qh = load_half_block(qb + (p.tail ? nfull : 0) * 1024, lane, false);
#pragma unroll
					for (int i = 0; i < NKP; i++)
						if (i < nfull) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[i], xn[i], acc, 0, 0, 0);
					if (p.tail) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh, xh, acc, 0, 0, 0);
					acc[0] = clip01(acc[0]); acc[1] = clip01(acc[1]); acc[2] = clip01(acc[2]); acc[3] = clip01(acc[3]);
				} else acc = sim_tile_generic(p.qtile + (int64_t)b * p.tile_bytes, p.tiles + (int64_t)k * p.tile_bytes, p.nk32, p.tail, lane, p.prec);
				const int tok = 16 * k + (lane & 15), c0 = (lane >> 4) * 4;
				if (p.pos_s) {
					const int ps = (tok >= t_a && tok < t_b) ? p.pos_s[tok] : 0;
#pragma unroll
					for (int r = 0; r < 4; r++) acc[r] = tag_weighted(acc[r], twl[16 * b + c0 + r], ps, tposl[16 * b + c0 + r], p.tw_keep, p.tw_threshold);
				}
			} else {
				const int tok = 16 * k + (lane >> 2), c4 = (lane & 3) * 4;
				const bool in = tok >= t_a && tok < t_b;
				float4 x;
				if constexpr (FLOW) {   // (the restated rows carry the tag weights already)
					x = in ? *reinterpret_cast<const float4 *>(p.dp_rows + ((int64_t)item * p.dp_rows_len + (tok - t_a)) * W + 16 * b + c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
				} else {
					const int id = in ? p.tok_id[tok] : 0;
					x = *reinterpret_cast<const float4 *>(p.table + (int64_t)b * p.table_stride + (int64_t)id * 16 + c4);
					if (p.pos_s) {
						const int ps = in ? p.pos_s[tok] : 0;
						x.x = tag_weighted(x.x, twl[16 * b + c4 + 0], ps, tposl[16 * b + c4 + 0], p.tw_keep, p.tw_threshold);
						x.y = tag_weighted(x.y, twl[16 * b + c4 + 1], ps, tposl[16 * b + c4 + 1], p.tw_keep, p.tw_threshold);
						x.z = tag_weighted(x.z, twl[16 * b + c4 + 2], ps, tposl[16 * b + c4 + 2], p.tw_keep, p.tw_threshold);
						x.w = tag_weighted(x.w, twl[16 * b + c4 + 3], ps, tposl[16 * b + c4 + 3], p.tw_keep, p.tw_threshold);
					}
				}
				acc[0] = x.x; acc[1] = x.y; acc[2] = x.z; acc[3] = x.w;
			}
			return acc;
		};
		// ... for every block, into the blocks' rings
		auto tile_write = [&](int k) {
			if (k > k_last) return;
#pragma unroll
			for (int b = 0; b < NQ; b++) {
				const int slot = (16 * k) & RG::mask(b);
				float *rb = ring + RG::base(b) * 16;
				const f32x4 acc = tile_values(k, b);
				const int row = (SRC == 0 || SRC == 1) ? (lane & 15) : (lane >> 2), c0 = (SRC == 0 || SRC == 1) ? (lane >> 4) * 4 : (lane & 3) * 4;
				*reinterpret_cast<f32x4 *>(rb + (slot + row) * 16 + c0) = acc;
				if (slot == 0) *reinterpret_cast<f32x4 *>(rb + (RG::mask(b) + 1 + row) * 16 + c0) = acc;
			}
		};
		if constexpr (GAP == 4) {
			// ---- relaxed 1:1 word mover's distance (vk_doc_kernel's GAP 4 arm over NQ blocks of columns): no recurrence, the tiles are
			// consumed where they are produced; column minima stay in the lane, a token's minimum over all columns crosses the four lanes
			// that share the token
			const float BIG = 3.402823466e+38F;
			const bool nbow = p.rwmd_normalize_bow != 0, sym = p.rwmd_symmetric != 0;
			const float w_t = nbow ? 1.0f / (float)len_t : 1.0f, w_s = nbow ? 1.0f / (float)len_s : 1.0f;
			constexpr bool quad = SRC == 2 || SRC == 3;   // (token lane >> 2, columns by lane & 3)
			float cmin[NQ][4], acc1 = 0.0f;
#pragma unroll
			for (int b = 0; b < NQ; b++) { cmin[b][0] = BIG; cmin[b][1] = BIG; cmin[b][2] = BIG; cmin[b][3] = BIG; }
			tile_load(k_first);
			for (int k = k_first; k <= k_last; k++) {
				const int tok = 16 * k + (quad ? (lane >> 2) : (lane & 15)), cbase = quad ? (lane & 3) * 4 : (lane >> 4) * 4;
				const bool in = tok >= t_a && tok < t_b;
				float rmin = BIG;
#pragma unroll
				for (int b = 0; b < NQ; b++) {
					const f32x4 acc = tile_values(k, b);
#pragma unroll
					for (int r = 0; r < 4; r++) {
						float dist = fmaxf(1.0f - acc[r], 0.0f);
						dist = (in && 16 * b + cbase + r < len_t) ? dist : BIG;
						cmin[b][r] = fminf(cmin[b][r], dist);
						rmin = fminf(rmin, dist);
					}
				}
				tile_load(k + 1);
				if (sym) {
					if constexpr (quad) { rmin = fminf(rmin, __shfl_xor(rmin, 1, 64)); rmin = fminf(rmin, __shfl_xor(rmin, 2, 64)); }
					else { rmin = fminf(rmin, lane_xor16(rmin, lane)); rmin = fminf(rmin, lane_xor32(rmin, lane)); }
					acc1 += (in && (quad ? (lane & 3) == 0 : lane < 16)) ? w_s * rmin : 0.0f;
				}
			}
			float acc0 = 0.0f;
#pragma unroll
			for (int b = 0; b < NQ; b++) {
#pragma unroll
				for (int r = 0; r < 4; r++) {
#pragma unroll
					for (int o = quad ? 4 : 1; o <= (quad ? 32 : 8); o <<= 1) cmin[b][r] = fminf(cmin[b][r], __shfl_xor(cmin[b][r], o, 64));
				}
			}
#pragma unroll
			for (int j = 0; j < W; j++) {
				if (j < len_t) {
					const float xj = w_t * __shfl(cmin[j >> 4][j & 3], quad ? ((j & 15) >> 2) : 16 * ((j & 15) >> 2), 64);
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
		tile_load(k_first); tile_write(k_first);
		tile_load(k_first + 1); tile_write(k_first + 1);
		tile_load(k_first + 2);
		wave_lds_fence();

		// ---- the sweep: a1 = H[u - 1][v] (this lane's last value; before its first row the border H[0][v])
		auto border_s = [&](int k) -> float {   // H[k][0]
			if (!global || k <= 0) return 0.0f;
			return GAP == 0 ? -(gs * (float)k) : -(a_s + gs * (float)k);
		};
		float b_t = 0.0f;   // H[0][v]
		if (global) b_t = GAP == 0 ? -(gt * (float)v) : -(a_t + gt * (float)v);
		float a1 = b_t;
		float e1 = VK_NEG_INF, f1 = VK_NEG_INF;   // affine: E[u - 1][v], F[u][v] of this lane's last step
		float prev_left = 0.0f;                     // H[u - 1][v - 1]: last step's `left` (lane 0, step 2: H[0][0] = 0)
		float best_v = 0.0f;
		int best_u = 0;
		const int steps_end = len_s + len_t;
		auto s_of = [&](int u) -> float { return myring[((t_a + u - 1) & rmask) * 16]; };
		float s_next = s_of(2 - v);
		// sixteen steps from a tile boundary on while every query column is inside the document (d > len_t, d + 15 <= len_s)
		auto steps16 = [&](int d0, auto loc) {
			constexpr int LOC = decltype(loc)::value;   // 0 local, 1 global, 2 semiglobal
			constexpr bool is_local = LOC == 0, is_global = LOC == 1;
			const int tok1 = t_a + d0 - 2;
			const float *sp = myring + ((tok1 + 1 - v) & rmask) * 16;
			float sv[17];
#pragma unroll
			for (int i = 0; i < 17; i++) sv[i] = sp[i * 16];
			const bool track = col && (is_local || (!is_global && v == len_t));
			const bool store = FLOW && col;
			uint8_t *dp = FLOW ? D + (d0 - v) * W + (v - 1) : nullptr;
#pragma unroll
			for (int i = 0; i < 16; i++) {
				const float left = docw_left(a1, border_s(d0 + i - 1));   // H[u][v - 1] (lane 0: the border column)
				const float diag = prev_left;                              // H[u - 1][v - 1]
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
					const float left_f = docw_left(f1, VK_NEG_INF);   // F[u][v - 1]
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
				if constexpr (FLOW) { if (store) dp[i * W] = (uint8_t)(dir | (ee << 2) | (fe << 3)); }
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
			// boundary: the first lane (v = 1) is about to enter a new tile -> the tile after it is written
			const int tok1 = t_a + d - 2;   // token of lane v = 1 on this step
			if ((tok1 & 15) == 0 && d > 2) {
				tile_write((tok1 >> 4) + 1);
				tile_load((tok1 >> 4) + 2);
				wave_lds_fence();
				if (d > len_t && d + 15 <= len_s) {
					if (local) steps16(d, std::integral_constant<int, 0>{});
					else if (global) steps16(d, std::integral_constant<int, 1>{});
					else steps16(d, std::integral_constant<int, 2>{});
					d += 15;
					continue;
				}
			}
			const float left = docw_left(a1, border_s(d - 1));   // H[u][v - 1] (lane 0: the border column)
			const float diag = prev_left;                         // H[u - 1][v - 1]
			prev_left = left;
			float left_f = VK_NEG_INF;
			if (GAP == 1) left_f = docw_left(f1, VK_NEG_INF);     // F[u][v - 1]
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
			if (GAP == 0) {
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
			if (FLOW && act) D[u * W + (v - 1)] = (uint8_t)(dir | (ee << 2) | (fe << 3));
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
			float m = col ? best_v : 0.0f;
#pragma unroll
			for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
			raw = m;
			int cu = (col && best_v == m && m > 0.0f) ? best_u : 0x7fffffff;
#pragma unroll
			for (int o = 32; o >= 1; o >>= 1) { const int x = __shfl_xor(cu, o, 64); cu = x < cu ? x : cu; }
			if (cu != 0x7fffffff) {
				su = cu;
				const unsigned long long hit = __ballot(col && best_v == m && best_u == cu);
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
			mapl[lane] = -1;
			wave_lds_fence();
			if (lane == 0) {
				int cu = su, cv = sv, state = 0;
				while (cu > 0 && cv > 0) {
					const int rec = (int)D[cu * W + (cv - 1)];
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
			const int mine = mapl[lane];
			float es = 0.0f;
			if (mine >= 0) {
				if (!p.pos_s) es = p.dp_rows[((int64_t)item * p.dp_rows_len + mine) * W + lane];   // the unmodified similarity of the edge (metric/alignment.h:339)
				else {
					// with tag weights the restated rows are the modified ones: this cell's cosine once more, canonically
					float o1[1];
					const int tok = t_a + mine;
					if (static_layout) static_sim_canon<1>(p.tiles, p.tile_bytes, p.tok_id[tok], p.qtile, lane, p.d, p.prec, p.q_ids, o1);
					else { sim_canon<1>(p.tiles + (int64_t)(tok >> 4) * p.tile_bytes, tok & 15, p.qtile + (int64_t)blk * p.tile_bytes, v16, p.d, p.prec, o1); o1[0] = clip01(o1[0]); }
					es = o1[0];
				}
			}
			p.mapping[item * 64 + lane] = (int16_t)mine;
			p.edge_sim[item * 64 + lane] = es;
			if (lane == 0) p.raw_out[item] = raw;
		}
		wave_lds_fence();   // the next document overwrites the rings
	}
}

// flow_k == 0: scores of the p->n_order slices of p->order (longest first); flow_k > 0: the flow_k winners of p->keys, their rows
// [dp_rows_len][16 nq] in p->dp_rows, one scratch region of p->scratch_stride >= vk_docw_scratch_bytes(max_len, nq) bytes per winner
extern "C" hipError_t vk_launch_docw(const VkWideParams *p, int32_t flow_k, hipStream_t stream) {
	if (p->len_t <= 16 || p->len_t > 64 || p->gap_mode < 0 || (p->gap_mode > 1 && !(p->gap_mode == 4 && flow_k == 0))) return hipErrorInvalidValue;
	const int nq = (p->len_t + 15) / 16;
	void (*kernel)(VkWideParams) = nullptr;
	if (flow_k > 0) {
		if (!p->dp_rows || !p->scratch || p->scratch_stride < (int64_t)vk_docw_scratch_bytes(p->max_len, nq)) return hipErrorInvalidValue;
		if (p->gap_mode == 0) kernel = nq == 2 ? vk_docw_kernel<true, 0, 3, 2> : nq == 3 ? vk_docw_kernel<true, 0, 3, 3> : vk_docw_kernel<true, 0, 3, 4>;
		else kernel = nq == 2 ? vk_docw_kernel<true, 1, 3, 2> : nq == 3 ? vk_docw_kernel<true, 1, 3, 3> : vk_docw_kernel<true, 1, 3, 4>;
		kernel<<<flow_k, 64, 0, stream>>>(*p);
		return hipGetLastError();
	}
	if (!p->order || p->n_order < 1) return hipSuccess;
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	const int64_t cap = (int64_t)cus * 8;
	const int grid = (int)(p->n_order < cap ? p->n_order : cap);
	const bool st = p->layout == VK_DEV_LAYOUT_STATIC;
	const bool regs = !st && p->prec == 0 && p->nk32 <= 12 && !getenv("VK_DOCW_GENERIC");
	if (p->gap_mode == 4) {   // the relaxed 1:1 WMD
		if (st) kernel = nq == 2 ? vk_docw_kernel<false, 4, 2, 2> : nq == 3 ? vk_docw_kernel<false, 4, 2, 3> : vk_docw_kernel<false, 4, 2, 4>;
		else if (regs) kernel = nq == 2 ? vk_docw_kernel<false, 4, 0, 2> : nq == 3 ? vk_docw_kernel<false, 4, 0, 3> : vk_docw_kernel<false, 4, 0, 4>;
		else kernel = nq == 2 ? vk_docw_kernel<false, 4, 1, 2> : nq == 3 ? vk_docw_kernel<false, 4, 1, 3> : vk_docw_kernel<false, 4, 1, 4>;
	} else if (p->gap_mode == 0) {
		if (st) kernel = nq == 2 ? vk_docw_kernel<false, 0, 2, 2> : nq == 3 ? vk_docw_kernel<false, 0, 2, 3> : vk_docw_kernel<false, 0, 2, 4>;
		else if (regs) kernel = nq == 2 ? vk_docw_kernel<false, 0, 0, 2> : nq == 3 ? vk_docw_kernel<false, 0, 0, 3> : vk_docw_kernel<false, 0, 0, 4>;
		else kernel = nq == 2 ? vk_docw_kernel<false, 0, 1, 2> : nq == 3 ? vk_docw_kernel<false, 0, 1, 3> : vk_docw_kernel<false, 0, 1, 4>;
	} else {
		if (st) kernel = nq == 2 ? vk_docw_kernel<false, 1, 2, 2> : nq == 3 ? vk_docw_kernel<false, 1, 2, 3> : vk_docw_kernel<false, 1, 2, 4>;
		else if (regs) kernel = nq == 2 ? vk_docw_kernel<false, 1, 0, 2> : nq == 3 ? vk_docw_kernel<false, 1, 0, 3> : vk_docw_kernel<false, 1, 0, 4>;
		else kernel = nq == 2 ? vk_docw_kernel<false, 1, 1, 2> : nq == 3 ? vk_docw_kernel<false, 1, 1, 3> : vk_docw_kernel<false, 1, 1, 4>;
	}
	kernel<<<grid, 64, 0, stream>>>(*p);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------
// General gaps (a table that is constant from some k <= 126 on) under a query of 17 .. 32 tokens: vk_doc_kernel's values-only sweep
// (vk_doc.hip, GAP 2) with two blocks of 16 columns.  Lanes 0 .. 31 hold the columns, lanes 32 .. 63 repeat them and scan the other
// half of the candidates (k = r + 1 + 2 i, r = lane >> 5): the column history is a ring of 128 rows x 32 columns stored twice (rows
// u - 1 .. u - 128 at descending addresses from one base), the lanes' values of the last 32 steps a second ring (gaps over t), the
// gaps of length one come from registers, those of two and more are gathered one step ahead and the halves meet through
// v_permlane32_swap.  FLOW stores H (4 bytes per cell) and the whole wave walks back: at each cell the first candidate in the
// reference's order whose value equals the cell's.  (vk_wide_kernel: 4.2 us per row of such a query; this sweep: ~0.9 us.)
// ---------------------------------------------------------------------------

extern "C" size_t vk_docg_scratch_bytes(int32_t max_len) { return ((size_t)(max_len + 2) * 32 * 4 + 255) / 256 * 256; }

template <bool FLOW, int SRC>
__global__ __launch_bounds__(64) void vk_docg_kernel(VkWideParams p) {
	using RG = DocwRings<2>;
	constexpr int W = 32, NQ = 2;
	__shared__ float ring[RG::rows * 16];
	__shared__ float Hr2[256 * 32];
	__shared__ float Htb[32 + 64 * 33];
	__shared__ float wq[2 * 64];
	__shared__ float wsl[128];
	__shared__ float wtl[64];
	__shared__ float twl[64];
	__shared__ int tposl[64];
	__shared__ int16_t mapl[64];
	float *Ht2 = Htb + 32;
	const int lane = threadIdx.x;
	twl[lane] = p.tw[lane]; tposl[lane] = p.tpos[lane];
	wsl[lane] = p.ws[lane]; wsl[64 + lane] = p.ws[64 + lane];
	wtl[lane] = p.wt[lane < 33 ? lane : 32];
	for (int j = lane; j < 128; j += 64) { const int k = (j >> 6) + 1 + 2 * (j & 63); wq[j] = (k >= 2 && k < p.ws_tail) ? p.ws[k < 128 ? k : 127] : __builtin_inff(); }
	for (int j = lane; j < 32 + 64 * 33; j += 64) Htb[j] = 0.0f;
	wave_lds_fence();
	const int cl = lane & 31, r2 = lane >> 5;
	const int v = cl + 1, v16 = cl & 15, blk = cl >> 4, len_t = p.len_t;
	const bool col = v <= len_t;
	const bool static_layout = p.layout == VK_DEV_LAYOUT_STATIC;
	const bool local = p.locality == VK_DEV_LOCAL, global = p.locality == VK_DEV_GLOBAL;
	float *Hs = FLOW ? reinterpret_cast<float *>(p.scratch + (int64_t)blockIdx.x * p.scratch_stride) : nullptr;   // FLOW: H[u * 32 + v - 1]
	const int T = p.ws_tail;
	const int T2 = T > 2 ? T : 2;
	const int n_i = T >= 2 ? T >> 1 : 0;                 // i with r + 1 + 2 i <= T - 1 for some share
	const int n_chunks = (n_i + 7) >> 3;                 // chunks of 8 candidates per share
	const float wsT = wsl[T], ws1 = wsl[1], wt1 = wtl[1];
	typedef float f2 __attribute__((ext_vector_type(2)));
	f2 wr[32];        // this lane's 64 gaps over s (k = r2 + 1 + 2 i), in pairs
	float wtr[16];    // ... and its 16 gaps over t (k = r2 + 1 + 2 i <= v, k >= 2)
#pragma unroll
	for (int i = 0; i < 32; i++) { wr[i].x = wq[r2 * 64 + 2 * i]; wr[i].y = wq[r2 * 64 + 2 * i + 1]; }
#pragma unroll
	for (int i = 0; i < 16; i++) { const int k = r2 + 1 + 2 * i; wtr[i] = (k >= 2 && k <= v) ? wtl[k] : __builtin_inff(); }
	const int rbase = blk == 0 ? RG::base(0) : RG::base(1);
	const float *myring = ring + rbase * 16 + v16;

	const int64_t n_items = FLOW ? (int64_t)gridDim.x : (int64_t)p.n_order;
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
		const int k_first = t_a >> 4, k_last = (t_b - 1) >> 4;
		constexpr int NKP = 12;
		const int nfull = p.tail ? p.nk32 - 1 : p.nk32;
		bf16x8 xn[SRC == 0 ? NKP : 1], xh = {0, 0, 0, 0, 0, 0, 0, 0};
		auto tile_load = [&](int k) {   // SRC 0: the K-steps of tile k into registers
			if constexpr (SRC == 0) {
				if (k > k_last) return;
				const uint8_t *tp = p.tiles + (int64_t)k * p.tile_bytes;
#pragma unroll
				for (int i = 0; i < NKP; i++)
					xn[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tp + (i < nfull ? i : 0) * 1024 + lane * 16));
				xh = load_half_block(tp + (p.tail ? nfull : 0) * 1024, lane, true);
			}
		};
		auto tile_write = [&](int k) {   // tile k for both blocks of query columns, into the blocks' rings (as vk_docw_kernel)
			if (k > k_last) return;
#pragma unroll
			for (int b = 0; b < NQ; b++) {
				const int slot = (16 * k) & RG::mask(b);
				float *rb = ring + RG::base(b) * 16;
				if constexpr (SRC == 0 || SRC == 1) {
					f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
					if constexpr (SRC == 0) {
						const uint8_t *qb = p.qtile + (int64_t)b * p.tile_bytes;
						bf16x8 qf[NKP];
#pragma unroll
						for (int i = 0; i < NKP; i++) qf[i] = *reinterpret_cast<const bf16x8 *>(qb + (i < nfull ? i : 0) * 1024 + lane * 16);
						const bf16x8 qh = load_half_block(qb + (p.tail ? nfull : 0) * 1024, lane, false);
#pragma unroll
						for (int i = 0; i < NKP; i++)
							if (i < nfull) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[i], xn[i], acc, 0, 0, 0);
						if (p.tail) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qh, xh, acc, 0, 0, 0);
						acc[0] = clip01(acc[0]); acc[1] = clip01(acc[1]); acc[2] = clip01(acc[2]); acc[3] = clip01(acc[3]);
					} else acc = sim_tile_generic(p.qtile + (int64_t)b * p.tile_bytes, p.tiles + (int64_t)k * p.tile_bytes, p.nk32, p.tail, lane, p.prec);
					const int tok = 16 * k + (lane & 15), c0 = (lane >> 4) * 4;
					if (p.pos_s) {
						const int ps = (tok >= t_a && tok < t_b) ? p.pos_s[tok] : 0;
#pragma unroll
						for (int r = 0; r < 4; r++) acc[r] = tag_weighted(acc[r], twl[16 * b + c0 + r], ps, tposl[16 * b + c0 + r], p.tw_keep, p.tw_threshold);
					}
					*reinterpret_cast<f32x4 *>(rb + (slot + (lane & 15)) * 16 + c0) = acc;
					if (slot == 0) *reinterpret_cast<f32x4 *>(rb + (RG::mask(b) + 1 + (lane & 15)) * 16 + c0) = acc;
				} else {
					const int tok = 16 * k + (lane >> 2), c4 = (lane & 3) * 4;
					const bool in = tok >= t_a && tok < t_b;
					float4 x;
					if constexpr (FLOW) {   // (the restated rows carry the tag weights already)
						x = in ? *reinterpret_cast<const float4 *>(p.dp_rows + ((int64_t)item * p.dp_rows_len + (tok - t_a)) * W + 16 * b + c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
					} else {
						const int id = in ? p.tok_id[tok] : 0;
						x = *reinterpret_cast<const float4 *>(p.table + (int64_t)b * p.table_stride + (int64_t)id * 16 + c4);
						if (p.pos_s) {
							const int ps = in ? p.pos_s[tok] : 0;
							x.x = tag_weighted(x.x, twl[16 * b + c4 + 0], ps, tposl[16 * b + c4 + 0], p.tw_keep, p.tw_threshold);
							x.y = tag_weighted(x.y, twl[16 * b + c4 + 1], ps, tposl[16 * b + c4 + 1], p.tw_keep, p.tw_threshold);
							x.z = tag_weighted(x.z, twl[16 * b + c4 + 2], ps, tposl[16 * b + c4 + 2], p.tw_keep, p.tw_threshold);
							x.w = tag_weighted(x.w, twl[16 * b + c4 + 3], ps, tposl[16 * b + c4 + 3], p.tw_keep, p.tw_threshold);
						}
					}
					*reinterpret_cast<float4 *>(rb + (slot + (lane >> 2)) * 16 + c4) = x;
					if (slot == 0) *reinterpret_cast<float4 *>(rb + (RG::mask(b) + 1 + (lane >> 2)) * 16 + c4) = x;
				}
			}
		};
		tile_load(k_first); tile_write(k_first);
		tile_load(k_first + 1); tile_write(k_first + 1);
		tile_load(k_first + 2);
		{   // the column history before the document: no candidates
			const float4 ninf = {VK_NEG_INF, VK_NEG_INF, VK_NEG_INF, VK_NEG_INF};
#pragma unroll
			for (int j = 0; j < 32; j++) reinterpret_cast<float4 *>(Hr2)[lane + 64 * j] = ninf;
		}
		wave_lds_fence();

		auto border_s = [&](int k) -> float {   // H[k][0]
			if (!global || k <= 0) return 0.0f;
			return -(k < T ? wsl[k] : wsl[T]);
		};
		const float b_t = global ? -wtl[v] : 0.0f;   // H[0][v]
		float a1 = b_t;
		float m_far = VK_NEG_INF;    // the best candidate of length two and more of the coming step (the first cell has none)
		float tail_m = VK_NEG_INF;   // max of H[u'][v] over u' <= u - T2 (the candidates at and beyond the table's tail)
		if (lane == 0) {   // what the first steps read of the steps before them: H[0][0], H[1][0]; H[0][1] (lane v = 1's border)
			Ht2[0] = 0.0f; Ht2[32 * 33] = 0.0f;
			const float b1 = border_s(1);
			Ht2[33] = b1; Ht2[33 * 33] = b1;
			Ht2[33 + 1] = b_t; Ht2[33 * 33 + 1] = b_t;
			Hr2[0] = b_t; Hr2[128 * 32] = b_t;
		}
		wave_lds_fence();
		float prev_left = 0.0f;
		float best_v = 0.0f;
		int best_u = 0;
		const int steps_end = len_s + len_t;
		const int rmask = 63;
		auto s_of = [&](int u) -> float { return myring[((t_a + u - 1) & rmask) * 16]; };
		float s_next = s_of(2 - v);
		// H[u][v - 1]: the left neighbour's last value (wave_shr:1; lanes 0 and 32 -- column 1 and its repeat -- take the border column)
		auto left_of = [&](float x, float border) -> float {
			const float l = docw_left(x, border);
			return lane == 32 ? border : l;
		};
		// the gaps of two and more of a coming step (row u1, step d1), NC chunks of 8 candidates per share (vk_doc.hip far_next)
		auto far_next = [&](int u1, int d1, auto nc) -> float {
			constexpr int NC = decltype(nc)::value;
			const float *hp = Hr2 + ((u1 & 127) + 1 - r2) * 32 + cl;
			const float *tp = Hr2 + ((u1 & 127) + 128 - T2) * 32 + cl;
			const float *gp = Ht2 + (((d1 & 31) + 32) * 33) + v - 34 * (r2 + 1);
			f2 hv[NC > 0 ? NC * 4 : 1];
#pragma unroll
			for (int i = 0; i < NC * 4; i++) { hv[i].x = hp[(63 - 2 * i) * 64]; hv[i].y = hp[(62 - 2 * i) * 64]; }
			const float xt = *tp;
			float ht[16];
#pragma unroll
			for (int i = 0; i < 16; i++) ht[i] = gp[-68 * i];
			float mm = VK_NEG_INF, mm2 = VK_NEG_INF, mm3 = VK_NEG_INF, mm4 = VK_NEG_INF;
#pragma unroll
			for (int j = 0; j < NC; j++) {
				mm = fmaxf(mm, fmaxf(hv[4 * j].x - wr[4 * j].x, hv[4 * j].y - wr[4 * j].y));
				mm2 = fmaxf(mm2, fmaxf(hv[4 * j + 1].x - wr[4 * j + 1].x, hv[4 * j + 1].y - wr[4 * j + 1].y));
				mm3 = fmaxf(mm3, fmaxf(hv[4 * j + 2].x - wr[4 * j + 2].x, hv[4 * j + 2].y - wr[4 * j + 2].y));
				mm4 = fmaxf(mm4, fmaxf(hv[4 * j + 3].x - wr[4 * j + 3].x, hv[4 * j + 3].y - wr[4 * j + 3].y));
			}
			tail_m = fmaxf(tail_m, xt);
			mm = fmaxf(mm, tail_m - wsT);
#pragma unroll
			for (int i = 0; i < 16; i += 4) {
				mm = fmaxf(mm, ht[i] - wtr[i]); mm2 = fmaxf(mm2, ht[i + 1] - wtr[i + 1]);
				mm3 = fmaxf(mm3, ht[i + 2] - wtr[i + 2]); mm4 = fmaxf(mm4, ht[i + 3] - wtr[i + 3]);
			}
			float mf = fmaxf(fmaxf(mm, mm2), fmaxf(mm3, mm4));
			float x0 = mf, x1 = mf;
			asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x0), "+v"(x1));
			return fmaxf(mf, fmaxf(x0, x1));
		};
		auto far_dispatch = [&](int u1, int d1) -> float {
			if (n_chunks == 0) return far_next(u1, d1, std::integral_constant<int, 0>{});
			if (n_chunks <= 1) return far_next(u1, d1, std::integral_constant<int, 1>{});
			if (n_chunks <= 2) return far_next(u1, d1, std::integral_constant<int, 2>{});
			if (n_chunks <= 4) return far_next(u1, d1, std::integral_constant<int, 4>{});
			return far_next(u1, d1, std::integral_constant<int, 8>{});
		};
		// sixteen steps from a tile boundary on while every query column is inside the document
		auto steps16g = [&](int d0, auto loc, auto nc) {
			constexpr bool is_local = decltype(loc)::value == 0;
			const bool track = col && (is_local || (!global && v == len_t));
			const bool keep = col && lane < 32;
			for (int i = 0; i < 16; i++) {
				const int d = d0 + i, u = d - v;
				const float left = left_of(a1, is_local ? 0.0f : border_s(d - 1));
				const float diag = prev_left;
				prev_left = left;
				const float sim = s_next;
				s_next = s_of(u + 1);
				float best = diag + sim;
				if constexpr (is_local) best = fmaxf(best, 0.0f);
				best = fmaxf(best, fmaxf(m_far, fmaxf(a1 - ws1, left - wt1)));
				const int B = ((d & 31) + 32) * 33;
				if (keep) {
					if constexpr (FLOW) Hs[u * W + cl] = best;
					Hr2[(u & 127) * 32 + cl] = best; Hr2[((u & 127) + 128) * 32 + cl] = best;
					Ht2[B - 32 * 33 + v] = best; Ht2[B + v] = best;
				}
				if constexpr (!is_local) {
					if (lane == 0) { const float x = border_s(d); Ht2[B - 32 * 33] = x; Ht2[B] = x; }
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
		for (int d = 2; d <= steps_end; d++) {
			const int u = d - v;
			const bool act = col && u >= 1 && u <= len_s;
			const int tok1 = t_a + d - 2;   // token of lane v = 1 on this step
			if ((tok1 & 15) == 0 && d > 2) {
				tile_write((tok1 >> 4) + 1);
				tile_load((tok1 >> 4) + 2);
				wave_lds_fence();
				if (d > len_t && d + 15 <= len_s) {
					auto go = [&](auto nc) { if (local) steps16g(d, std::integral_constant<int, 0>{}, nc); else steps16g(d, std::integral_constant<int, 1>{}, nc); };
					if (n_chunks == 0) go(std::integral_constant<int, 0>{});
					else if (n_chunks <= 1) go(std::integral_constant<int, 1>{});
					else if (n_chunks <= 2) go(std::integral_constant<int, 2>{});
					else if (n_chunks <= 4) go(std::integral_constant<int, 4>{});
					else go(std::integral_constant<int, 8>{});
					d += 15;
					continue;
				}
			}
			const int B = ((d & 31) + 32) * 33;
			const float left = left_of(a1, border_s(d - 1));
			const float diag = prev_left;
			prev_left = left;
			const float s = act ? s_next : 0.0f;
			s_next = s_of(u + 1);
			float best;
			{
				const float c = diag + s;
				best = (!local || c > 0.0f) ? c : 0.0f;
			}
			best = fmaxf(best, fmaxf(m_far, fmaxf(a1 - ws1, left - wt1)));
			if (FLOW && act && lane < 32) Hs[u * W + cl] = best;
			if (lane < 32 && col && u >= 0 && u <= len_s) {   // this cell (before the first row: the border H[0][v]) joins the histories
				const float x = u == 0 ? b_t : best;
				Hr2[(u & 127) * 32 + cl] = x; Hr2[((u & 127) + 128) * 32 + cl] = x;
				Ht2[B - 32 * 33 + v] = x; Ht2[B + v] = x;
			}
			if (lane == 0) { const float x = border_s(d); Ht2[B - 32 * 33] = x; Ht2[B] = x; }
			wave_lds_fence();
			m_far = far_dispatch(u + 1, d + 1);
			{
				const bool nb = act && !global && (local || u == len_s || v == len_t) && best > best_v;   // first maximum of this column
				best_v = nb ? best : best_v;
				if (FLOW) best_u = nb ? u : best_u;
				a1 = act ? best : a1;
			}
		}

		// ---- aligner score and start cell: the first maximum in row-major order (smallest u, then smallest v)
		float raw;
		int su = 0, sv = 0;
		if (global) {
			raw = __shfl(a1, len_t - 1, 64);
			su = len_s; sv = len_t;
		} else {
			float m = (col && lane < 32) ? best_v : 0.0f;
#pragma unroll
			for (int o = 16; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
			m = __shfl(m, 0, 64);
			raw = m;
			int cu = (lane < 32 && col && best_v == m && m > 0.0f) ? best_u : 0x7fffffff;
#pragma unroll
			for (int o = 16; o >= 1; o >>= 1) { const int x = __shfl_xor(cu, o, 64); cu = x < cu ? x : cu; }
			cu = __shfl(cu, 0, 64);
			if (cu != 0x7fffffff) {
				su = cu;
				const unsigned long long hit = __ballot(lane < 32 && col && best_v == m && best_u == cu);
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
			mapl[lane] = -1;
			wave_lds_fence();
			{
				// the whole wave walks back over the stored H: at each cell the FIRST candidate in the reference's order (zero, diagonal,
				// gaps over s by length, gaps over t by length) whose value is the cell's -- what a scan replacing on strictly greater keeps
				auto h_at = [&](int uu, int vv) -> float {
					if (vv == 0) return border_s(uu);
					if (uu == 0) return global ? -wtl[vv] : 0.0f;
					return Hs[uu * W + (vv - 1)];
				};
				int cu = su, cv = sv;
				while (cu > 0 && cv > 0) {
					const float h = Hs[cu * W + (cv - 1)];
					if (local && !(h > 0.0f)) break;
					const float sdp = p.dp_rows[((int64_t)item * p.dp_rows_len + (cu - 1)) * W + (cv - 1)];
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
			}
			wave_lds_fence();
			const int mine = mapl[lane];
			float es = 0.0f;
			if (mine >= 0) {
				if (!p.pos_s) es = p.dp_rows[((int64_t)item * p.dp_rows_len + mine) * W + lane];
				else {
					float o1[1];
					const int tok = t_a + mine;
					if (static_layout) static_sim_canon<1>(p.tiles, p.tile_bytes, p.tok_id[tok], p.qtile, lane, p.d, p.prec, p.q_ids, o1);
					else { sim_canon<1>(p.tiles + (int64_t)(tok >> 4) * p.tile_bytes, tok & 15, p.qtile + (int64_t)(lane >> 4) * p.tile_bytes, lane & 15, p.d, p.prec, o1); o1[0] = clip01(o1[0]); }
					es = o1[0];
				}
			}
			p.mapping[item * 64 + lane] = (int16_t)mine;
			p.edge_sim[item * 64 + lane] = es;
			if (lane == 0) p.raw_out[item] = raw;
		}
		wave_lds_fence();   // the next document overwrites the rings
	}
}

// general gaps (p->ws_tail in [1, 126]) under a query of 17 .. 32 tokens; flow_k as vk_launch_docw
extern "C" hipError_t vk_launch_docg(const VkWideParams *p, int32_t flow_k, hipStream_t stream) {
	if (p->len_t <= 16 || p->len_t > 32 || p->gap_mode != 2 || p->ws_tail < 1 || p->ws_tail > 126) return hipErrorInvalidValue;
	if (flow_k > 0) {
		if (!p->dp_rows || !p->scratch || p->scratch_stride < (int64_t)vk_docg_scratch_bytes(p->max_len)) return hipErrorInvalidValue;
		vk_docg_kernel<true, 3><<<flow_k, 64, 0, stream>>>(*p);
		return hipGetLastError();
	}
	if (!p->order || p->n_order < 1) return hipSuccess;
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	const int64_t cap = (int64_t)cus * 8;
	const int grid = (int)(p->n_order < cap ? p->n_order : cap);
	const bool st = p->layout == VK_DEV_LAYOUT_STATIC;
	const bool regs = !st && p->prec == 0 && p->nk32 <= 12;
	void (*kernel)(VkWideParams) = st ? vk_docg_kernel<false, 2> : regs ? vk_docg_kernel<false, 0> : vk_docg_kernel<false, 1>;
	kernel<<<grid, 64, 0, stream>>>(*p);
	return hipGetLastError();
}
