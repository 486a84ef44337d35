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
		// tile k (tokens 16 k ..) for every block of query columns, into the blocks' rings
		auto tile_write = [&](int k) {
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
					} else acc = sim_tile_generic(p.qtile + (int64_t)b * p.tile_bytes, p.tiles + (int64_t)k * p.tile_bytes, p.nk32, p.tail, lane, p.prec);   // lane: S[token lane & 15][query 16 b + 4 (lane >> 4) + r]
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
	if (p->len_t <= 16 || p->len_t > 64 || p->gap_mode < 0 || p->gap_mode > 1) return hipErrorInvalidValue;
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
	if (p->gap_mode == 0) {
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
