// vk_doc.hip -- whole documents as slices under linear / affine gaps (round 4): a skewed sweep without an in-row dependency.
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
// region; start cell = first maximum in row-major order; lane 0 walks back).  Queries of at most 16 tokens, gap modes 0 / 1; general
// gaps keep vk_wide_kernel (their candidate scans dominate either way).
// One document per wave, DP in the 16 lanes of DPP row 0 (the other rows repeat it: their lanes are needed for the tiles anyway).
// ---------------------------------------------------------------------------

#define VK_DOC_RING 64   // rows of the LDS ring: four tiles (one being consumed, the next, the one being written, slack)

extern "C" size_t vk_doc_scratch_bytes(int32_t max_len) { return ((size_t)(max_len + 2) * 16 + 255) / 256 * 256; }

__device__ __forceinline__ float doc_left(float x, float border) {   // value of lane v - 2 within the DPP row (lane 0 of a row: `border`)
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, border), __builtin_bit_cast(int, x), DPP_ROW_SHR1, 0xf, 0xf, false));
}

// SRC: where the similarities come from, at compile time (a runtime choice made the compiler wait for EVERY outstanding load at the
// top of every step: the registers of one path's loads were the temporaries of another's) -- 0: contextual tiles, bf16 rows of up to 12
// K-steps, loaded into registers a boundary ahead; 1: contextual tiles, any row (the MFMA sequence loads them); 2: the static layout's
// table, gathered by token id; 3: FLOW (the restated rows)
template <bool FLOW, int GAP, int SRC>
__global__ __launch_bounds__(64) void vk_doc_kernel(VkWideParams p) {
	__shared__ float ring[VK_DOC_RING * 16];   // S[token & 63][query column]: what the DP runs on (tag weights applied)
	__shared__ float twl[16];
	__shared__ int tposl[16];
	__shared__ int16_t mapl[16];
	const int lane = threadIdx.x;
	if (lane < 16) { twl[lane] = p.tw[lane]; tposl[lane] = p.tpos[lane]; }
	wave_lds_fence();
	const int v = (lane & 15) + 1, len_t = p.len_t;
	const bool col = v <= len_t;
	constexpr bool is_static = SRC == 2;
	const bool static_layout = p.layout == VK_DEV_LAYOUT_STATIC;   // (FLOW: the edges' unmodified similarities)
	const bool local = p.locality == VK_DEV_LOCAL, global = p.locality == VK_DEV_GLOBAL;
	const float gs = p.gs, gt = p.gt, a_s = p.a_s, a_t = p.a_t, open_s = p.open_s, open_t = p.open_t;
	uint8_t *D = FLOW ? p.scratch + (int64_t)blockIdx.x * p.scratch_stride : nullptr;   // FLOW: D[u * 16 + v - 1]

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
	for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
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
		auto write_st = [&](int k, float4 x, int st_ps) {
			if (k > k_last) return;
			const int tok = 16 * k + (lane >> 2), c4 = (lane & 3) * 4;
			if (!FLOW && p.pos_s) {   // (FLOW: the restated rows carry the tag weights already)
				x.x = tag_weighted(x.x, twl[c4 + 0], st_ps, tposl[c4 + 0], p.tw_keep, p.tw_threshold);
				x.y = tag_weighted(x.y, twl[c4 + 1], st_ps, tposl[c4 + 1], p.tw_keep, p.tw_threshold);
				x.z = tag_weighted(x.z, twl[c4 + 2], st_ps, tposl[c4 + 2], p.tw_keep, p.tw_threshold);
				x.w = tag_weighted(x.w, twl[c4 + 3], st_ps, tposl[c4 + 3], p.tw_keep, p.tw_threshold);
			}
			*reinterpret_cast<float4 *>(ring + (tok & (VK_DOC_RING - 1)) * 16 + c4) = x;
		};
		auto tile_load = [&](int k) {   // contextual, `regs`: the K-steps of tile k into registers
			if (k > k_last) return;
			const uint8_t *tp = p.tiles + (int64_t)k * p.tile_bytes;
#pragma unroll
			for (int i = 0; i < NKP; i++)   // (K-steps the row does not have re-read its first one: unconditional loads)
				xn[i] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8 *>(tp + (i < nfull ? i : 0) * 1024 + lane * 16));
			xh = load_half_block(tp + (p.tail ? nfull : 0) * 1024, lane, true);
		};
		auto tile_write = [&](int k) {   // contextual: S of tile k into the ring (from the registers, or loaded here)
			if (k > k_last) return;
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
			*reinterpret_cast<f32x4 *>(ring + (tok & (VK_DOC_RING - 1)) * 16 + c0) = acc;
		};
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
		wave_lds_fence();

		// ---- the sweep: a1 = H[u - 1][v] (this lane's last value; before its first row the border H[0][v])
		auto border_s = [&](int k) -> float { return (!global || k <= 0) ? 0.0f : (GAP == 0 ? -(gs * (float)k) : -(a_s + gs * (float)k)); };   // H[k][0]
		float a1 = (!global) ? 0.0f : (GAP == 0 ? -(gt * (float)v) : -(a_t + gt * (float)v));   // H[0][v]
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
		for (int d = 2; d <= steps_end; d++) {
			const int u = d - v;
			const bool act = col && u >= 1 && u <= len_s;
			// boundary: the first lane (v = 1) is about to enter a new tile -> the tile after it is written, the one after that requested
			const int tok1 = t_a + d - 2;   // token of lane v = 1 on this step
			if ((tok1 & 15) == 0 && d > 2) boundary(tok1 >> 4);
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
			if (FLOW && act && lane < 16) D[u * 16 + (v - 1)] = (uint8_t)(dir | (ee << 2) | (fe << 3));
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
			if (lane == 0) {
				int cu = su, cv = sv, state = 0;
				while (cu > 0 && cv > 0) {
					const int rec = D[cu * 16 + (cv - 1)];
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
	if (p->len_t > 16 || (p->gap_mode != 0 && p->gap_mode != 1)) return hipErrorInvalidValue;
	if (flow_k > 0) {
		if (!p->dp_rows || !p->scratch || p->scratch_stride < (int64_t)vk_doc_scratch_bytes(p->max_len)) return hipErrorInvalidValue;
		if (p->gap_mode == 0) vk_doc_kernel<true, 0, 3><<<flow_k, 64, 0, stream>>>(*p);
		else vk_doc_kernel<true, 1, 3><<<flow_k, 64, 0, stream>>>(*p);
		return hipGetLastError();
	}
	if (!p->order || p->n_order < 1) return hipSuccess;
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	const int64_t cap = (int64_t)cus * 16;
	const int grid = (int)(p->n_order < cap ? p->n_order : cap);
	const int src = p->layout == VK_DEV_LAYOUT_STATIC ? 2 : (p->prec == 0 && p->nk32 <= 12) ? 0 : 1;
	void (*kernel)(VkWideParams);
	if (p->gap_mode == 0) kernel = src == 2 ? vk_doc_kernel<false, 0, 2> : src == 0 ? vk_doc_kernel<false, 0, 0> : vk_doc_kernel<false, 0, 1>;
	else kernel = src == 2 ? vk_doc_kernel<false, 1, 2> : src == 0 ? vk_doc_kernel<false, 1, 0> : vk_doc_kernel<false, 1, 1>;
	kernel<<<grid, 64, 0, stream>>>(*p);
	return hipGetLastError();
}
