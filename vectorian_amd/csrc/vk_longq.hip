// vk_longq.hip -- alignments of queries of 65 .. 512 tokens (round 4; refused until then: a lane per query column bounds the other
// kernels at 64 tokens, upstream's only bound is the int16 of a mapping, vectorian/core/cpp/metric/alignment.h:357-358).
#include "vk_common.hip.h"

// ---------------------------------------------------------------------------
// Roles swapped, anti-diagonal sweep.  One wave per slice (slices of at most 64 tokens: sentences), lane = slice token u - 1,
// time = the query's tokens: on step d the lanes hold the cells of anti-diagonal u + v = d, cell (u, v = d - u) in lane u - 1.
// Every dependency of the recurrence (oracle/vk_oracle.c align_linear / align_affine / align_general; pyalign's solvers upstream,
// metric/alignment.h:247-294) lies on an earlier diagonal:
//     H[u-1][v-1]: lane u - 2 two steps ago       (diagonal)
//     H[u-1][v]:   lane u - 2 one step ago        (gap over s, length 1; E of the affine solver likewise)
//     H[u][v-1]:   this lane one step ago         (gap over t, length 1; F likewise)
//     general gaps: H[u-k][v], H[u][v-k] from the matrix kept in memory ([v][lane]: rows of a query token)
// so a step is a handful of vector instructions plus two lane shifts, whatever the query's length; len_s + len_t steps per slice.
// The similarity matrix of the slice against the whole query sits in LDS, [query token][slice token], 256 bytes per query token:
//   scoring pass (FLOW = false): MFMA, token tiles as A operand and query tiles as B (both are in the same operand order), the
//     columns shifted so that the slice's first token is column 0; static layout: gathered from the per-tile tables by token id;
//   winners (FLOW = true): the canonical arithmetic (sim_canon16: the oracle's own sums), candidates in the oracle's order
//     (zero, diagonal, gap over s by ascending length, gap over t likewise; replaced on strictly greater), direction and gap length
//     of every cell in a scratch region, start cell = first maximum in row-major order (u outer, v inner), lane 0 walks back:
//     aligner score, mapping and edge similarities equal to the oracle's bit for bit.  The values of the cells do not depend on
//     the order cells are visited in, only on the order of the candidates within a cell -- which is the oracle's.
// Cost per slice: len_t / 16 x (tiles of the slice) MFMA tiles + (len_s + len_t) steps; general gaps add the candidate scans
// (u + v loads per cell, as the reference's O(n m (n + m)) solver): a path that keeps such queries on the device, not a roofline
// kernel.
// ---------------------------------------------------------------------------

// the strip's row: the corpus's longest slice rounded up to 16 columns (a 32-token corpus takes half the LDS of a 64-token one:
// twice the waves per CU)
static inline int longq_stride(int max_len) { const int w = (max_len + 15) / 16 * 16; return w < 16 ? 16 : w > 64 ? 64 : w; }

struct VkLongqGeom { int lt_pad; size_t s_bytes, hm_bytes, dm_bytes, su_bytes; };

static inline VkLongqGeom longq_geom(int len_t, bool general, bool flow, bool tagged) {
	VkLongqGeom g;
	g.lt_pad = (len_t + 15) / 16 * 16;
	g.s_bytes = (size_t)g.lt_pad * 64 * 4;
	g.hm_bytes = general ? (size_t)(len_t + 1) * 64 * 4 : 0;            // H[v][lane]
	g.dm_bytes = flow ? (size_t)(len_t + 1) * 64 * 2 : 0;               // direction | flags | gap length per cell
	g.su_bytes = (flow && tagged) ? (size_t)g.lt_pad * 64 * 4 : 0;      // unmodified similarities (reported per edge)
	return g;
}

extern "C" int32_t vk_longq_stride(int32_t max_len) { return longq_stride(max_len); }

extern "C" size_t vk_longq_scratch_bytes(int32_t len_t, int32_t gap_mode, int32_t flow, int32_t tagged) {
	const VkLongqGeom g = longq_geom(len_t, gap_mode == 2, flow != 0, tagged != 0);
	return (g.hm_bytes + g.dm_bytes + g.su_bytes + 255) / 256 * 256 + 256;
}

extern "C" size_t vk_longq_lds_bytes(int32_t len_t, int32_t max_len, int32_t flow) {
	return (size_t)((len_t + 15) / 16 * 16) * longq_stride(max_len) * 4 + (flow ? VK_CANON_LDS : 0) + 64;
}

// value of lane - 1 (lane 0: `border`): one DPP move across the wave (wave_shr:1; the first version shifted through ds_bpermute, an LDS
// round trip per step and register, and twice per step)
__device__ __forceinline__ float lane_up(float x, float border) {
	return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, border), __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, false));
}

// 300-d bf16 rows: the slice's token tiles (up to three at a time) stay in registers as A fragments while the query's tiles stream
// past them -- each query tile is fetched from L2 once per slice (the second and third use hit the L1) instead of once per token
// tile, and the token tiles once instead of once per query tile: 407 -> 97 KB of L2 traffic per 32-token slice and 100-token query,
// which is what the first form of this pass was bound by (39 -> ms per million slices).  Plain loads for the query
// tile (it is re-read by every wave), same MFMA sequence as sim_tile.
template <int NK, bool HALF>
__device__ __forceinline__ f32x4 longq_sim_tile(const QFrag<NK, HALF> &f, const uint8_t *__restrict__ qtile, int lane) {
	bf16x8 x[NK];
#pragma unroll
	for (int t = 0; t < NK; t++) {
		if (HALF && t == NK - 1) x[t] = load_half_block(qtile + t * 1024, lane, false);
		else x[t] = *reinterpret_cast<const bf16x8 *>(qtile + t * 1024 + lane * 16);
	}
	f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
	for (int t = 0; t < NK; t++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.q[t], x[t], acc, 0, 0, 0);
	acc[0] = clip01(acc[0]); acc[1] = clip01(acc[1]); acc[2] = clip01(acc[2]); acc[3] = clip01(acc[3]);
	return acc;
}

enum { LQ_STOP = 0, LQ_DIAG = 1, LQ_UP = 2, LQ_LEFT = 3 };   // D_* of the oracle; bits 0-1 of a cell's record, bit 2: E extended, bit 3: F extended, bits 4..: gap length

// HL: general gaps, scoring pass: the matrix H[v][lane] in LDS behind the strip (queries of up to ~280 tokens over 64-token slices,
// ~570 over 32-token ones: vk_longq_hm_in_lds) instead of the workgroup's scratch region in global memory -- the candidate scans
// are loads from it, one per candidate (200,000 slices of 32 tokens, 100-token query, exp5: 396 -> ms)
template <bool FLOW, int GAP, bool HL = false>
__global__ __launch_bounds__(64) void vk_longq_kernel(VkLongqParams p) {
	extern __shared__ float4 vk_smem4[];
	uint8_t *canon = reinterpret_cast<uint8_t *>(vk_smem4);
	float *Sl = reinterpret_cast<float *>(canon + (FLOW ? VK_CANON_LDS : 0));   // [lt_pad][64]: S[v - 1][u - 1], what the DP runs on
	const int lane = threadIdx.x;
	const int LT = p.len_t, SW = p.s_stride;
	uint8_t *region = p.scratch ? p.scratch + (int64_t)blockIdx.x * p.scratch_stride : nullptr;
	const int HS = HL ? SW : 64;                                                                       // row stride of the matrix
	float *Hm = HL ? Sl + (size_t)((LT + 15) / 16 * 16) * SW : reinterpret_cast<float *>(region);      // GAP 2: H[v][lane], v = 0 .. LT
	const size_t hm_bytes = GAP == 2 ? (size_t)(LT + 1) * 64 * 4 : 0;
	int16_t *Dm = reinterpret_cast<int16_t *>(region + hm_bytes);                                      // FLOW: the cells' records
	float *Su = reinterpret_cast<float *>(region + hm_bytes + (FLOW ? (size_t)(LT + 1) * 64 * 2 : 0)); // FLOW with tag weights: unmodified S
	const bool is_static = p.layout == VK_DEV_LAYOUT_STATIC;
	const bool local = p.locality == VK_DEV_LOCAL, global = p.locality == VK_DEV_GLOBAL;
	const float gs = p.gs, gt = p.gt, a_s = p.a_s, a_t = p.a_t, open_s = p.open_s, open_t = p.open_t;
	const int u = lane + 1;

	const int64_t n_items = FLOW ? (int64_t)p.n_keys : (int64_t)p.n_sent;
	for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
		int64_t g = item;
		if (FLOW) {
			const uint64_t key = p.keys[item];
			if (key == 0) continue;   // fewer than k admitted
			g = (int64_t)(uint32_t)(key & 0xffffffffu);
		}
		const int t_a = p.sent_start[g], t_b = p.sent_end[g];
		const int len_s = t_b - t_a;
		if (len_s < 1 || len_s > 64) {   // (longer slices: refused by the host before the launch; empty ones carry no score)
			if (!FLOW && lane == 0) { p.scores[g] = VK_NEG_INF; if (p.raw) p.raw[g] = VK_NEG_INF; }
			continue;
		}
		const bool ucol = u <= len_s;

		// ---- similarities: Sl[(v - 1) * 64 + (u - 1)]
		if constexpr (FLOW) {
			for (int rb = 0; rb * 16 < len_s; rb++) {
				const int rel = rb * 16 + (lane & 15);
				const int tok = t_a + (rel < len_s ? rel : 0);   // (past the slice's end its first token again, never read)
				const int id = is_static ? p.tok_id[tok] : 0;
				const uint8_t *xrow = is_static ? canon_row_ptr_static(p.tiles, p.tile_bytes, id) : p.tiles + (int64_t)(tok >> 4) * p.tile_bytes + (tok & 15) * 16;
				const int ps = p.pos_s ? p.pos_s[tok] : 0;
#pragma unroll 1
				for (int b = 0; b < p.nq; b++) {
					float val[4];
					sim_canon16(xrow, p.qtile + (int64_t)b * p.tile_bytes, p.nk32, p.tail, p.d, p.prec, canon, lane, val);
					const int c0 = 16 * b + (lane >> 4) * 4;
#pragma unroll
					for (int r = 0; r < 4; r++) {
						float x = (is_static && p.q_ids && p.q_ids[c0 + r] == id) ? 1.0f : clip01(val[r]);   // sim[id(t_j)][j] = 1 (metric/static.cpp:58-67)
						if (p.pos_s) {
							if (rel < len_s && c0 + r < LT) Su[(c0 + r) * 64 + rel] = x;
							x = tag_weighted(x, p.tw[c0 + r], ps, p.tpos[c0 + r], p.tw_keep, p.tw_threshold);
						}
						if (rel < len_s) Sl[(c0 + r) * SW + rel] = x;
					}
				}
			}
		} else if (is_static) {
			const int id = ucol ? p.tok_id[t_a + lane] : 0;
			const int ps = (p.pos_s && ucol) ? p.pos_s[t_a + lane] : 0;
			for (int b = 0; b < p.nq; b++) {
				const float *row = p.table + (int64_t)b * p.table_stride + (int64_t)id * 16;
				float4 x[4];
#pragma unroll
				for (int j = 0; j < 4; j++) x[j] = *reinterpret_cast<const float4 *>(row + 4 * j);
#pragma unroll
				for (int j = 0; j < 4; j++) {
					const float e[4] = {x[j].x, x[j].y, x[j].z, x[j].w};
#pragma unroll
					for (int r = 0; r < 4; r++) {
						const int c = 16 * b + 4 * j + r;
						float y = e[r];
						if (p.pos_s) y = tag_weighted(y, p.tw[c], ps, p.tpos[c], p.tw_keep, p.tw_threshold);
						if (ucol) Sl[c * SW + lane] = y;
					}
				}
			}
		} else {
			const int tile0 = t_a >> 4, off = t_a - tile0 * 16, ntiles = ((t_b + 15) >> 4) - tile0;
			if (p.prec == 0 && p.nk32 == 10 && p.tail == 1) {
				for (int j0 = 0; j0 < ntiles; j0 += 3) {
					QFrag<10, true> tf[3];
#pragma unroll
					for (int jj = 0; jj < 3; jj++)   // (past the slice's last tile: that tile again, its columns fall outside the slice and are dropped)
						load_qfrag(tf[jj], p.tiles + (int64_t)(tile0 + (j0 + jj < ntiles ? j0 + jj : ntiles - 1)) * p.tile_bytes, lane);
					for (int b = 0; b < p.nq; b++) {
						const int vq = 16 * b + (lane & 15);
#pragma unroll
						for (int jj = 0; jj < 3; jj++) {
							if (j0 + jj >= ntiles) break;
							const f32x4 acc = longq_sim_tile(tf[jj], p.qtile + (int64_t)b * p.tile_bytes, lane);
#pragma unroll
							for (int r = 0; r < 4; r++) {
								const int c = 16 * (j0 + jj) + 4 * (lane >> 4) + r - off;
								if (c >= 0 && c < len_s) {
									float y = acc[r];
									if (p.pos_s) y = tag_weighted(y, p.tw[vq], p.pos_s[t_a + c], p.tpos[vq], p.tw_keep, p.tw_threshold);
									Sl[vq * SW + c] = y;
								}
							}
						}
					}
				}
			} else
			for (int b = 0; b < p.nq; b++) {
				const int vq = 16 * b + (lane & 15);   // this lane's query token (0-based)
				for (int j = 0; j < ntiles; j++) {
					// A = the token tile, B = the query tile: lane l holds S[query row l & 15][token 4 (l >> 4) + r of the tile]
					const f32x4 acc = sim_tile_generic(p.tiles + (int64_t)(tile0 + j) * p.tile_bytes, p.qtile + (int64_t)b * p.tile_bytes, p.nk32, p.tail, lane, p.prec);
#pragma unroll
					for (int r = 0; r < 4; r++) {
						const int c = 16 * j + 4 * (lane >> 4) + r - off;   // column of the slice
						if (c >= 0 && c < len_s) {
							float y = acc[r];
							if (p.pos_s) y = tag_weighted(y, p.tw[vq], p.pos_s[t_a + c], p.tpos[vq], p.tw_keep, p.tw_threshold);
							Sl[vq * SW + c] = y;
						}
					}
				}
			}
		}
		wave_lds_fence();

		// ---- the sweep.  a1 = H[u][v - 1] of this lane's row (borders included), prev_up = H[u - 1][v - 1]; e1 / f1: the affine solver's states
		auto border_s = [&](int k) -> float {   // H[k][0]
			if (!global || k == 0) return 0.0f;
			return GAP == 0 ? -(gs * (float)k) : GAP == 1 ? -(a_s + gs * (float)k) : -p.ws[k];
		};
		auto border_t = [&](int k) -> float {   // H[0][k]
			if (!global || k <= 0) return 0.0f;
			return GAP == 0 ? -(gt * (float)k) : GAP == 1 ? -(a_t + gt * (float)k) : -p.wt[k];
		};
		float a1 = border_s(u);
		float prev_up = 0.0f;         // H[u - 1][v - 1]: what `up` was one step ago (step 2: H[0][0] = 0)
		float e1 = VK_NEG_INF, f1 = VK_NEG_INF;
		float tail_max = VK_NEG_INF;   // general gaps, scoring pass: max of H[u][v'] over v' <= v - tail (see the scan below)
		float best_v = 0.0f;          // LOCAL / SEMIGLOBAL: this row's best cell, first v among equals (borders are 0: "borders first")
		int best_at = 0;
		// (general gaps: row 0 and column 0 of the matrix are not stored -- the scans below take the borders from border_t / border_s)
		const int steps_end = len_s + LT;
		for (int dg = 2; dg <= steps_end; dg++) {
			const int v = dg - u;
			const bool act = ucol && v >= 1 && v <= LT;
			const float up_h = lane_up(a1, border_t(dg - 1));       // H[u - 1][v] (lane 0: the border row)
			const float dg_h = prev_up;                              // H[u - 1][v - 1] = H[u - 1][v] of the step before
			prev_up = up_h;
			float up_e = VK_NEG_INF;
			if (GAP == 1) up_e = lane_up(e1, VK_NEG_INF);            // E[u - 1][v]
			const float s = act ? Sl[(v - 1) * SW + lane] : 0.0f;
			float best, e = VK_NEG_INF, f = VK_NEG_INF;
			int dir = LQ_DIAG, kk = 0, ee = 0, fe = 0;
			{
				const float c = dg_h + s;
				if (local) { best = 0.0f; dir = LQ_STOP; if (c > best) { best = c; dir = LQ_DIAG; } }
				else best = c;
			}
			if (GAP == 0) {
				float c = up_h - gs;
				if (c > best) { best = c; dir = LQ_UP; }
				c = a1 - gt;
				if (c > best) { best = c; dir = LQ_LEFT; }
			} else if (GAP == 1) {
				// gap of length 1 (open) first, longer (extend) only if strictly greater (align_affine)
				e = up_h - open_s;
				float c = up_e - gs;
				if (c > e) { e = c; ee = 1; }
				f = a1 - open_t;
				c = f1 - gt;
				if (c > f) { f = c; fe = 1; }
				if (e > best) { best = e; dir = LQ_UP; }
				if (f > best) { best = f; dir = LQ_LEFT; }
			} else {
				// general gaps: H[u - k][v] - w_s(k), k = 1 .. u, then H[u][v - k] - w_t(k), k = 1 .. v (align_general); the trip counts are
				// the wave's (lanes beyond their own range sit out)
				// (eight candidates at a time: their loads first -- addresses clamped into the matrix, no loads under branches --, then the
				// compares in the oracle's order; one load per trip of a branchy loop waited out an LDS round trip per candidate)
				const int ku = len_s < 64 ? len_s : 64;
				const int vc = v < 1 ? 1 : v > LT ? LT : v;   // this lane's row of the matrix, clamped for the lanes that sit out
				for (int k0 = 1; k0 <= ku; k0 += 8) {
					float src[8];
#pragma unroll
					for (int i = 0; i < 8; i++) { const int l2 = lane - (k0 + i); src[i] = Hm[vc * HS + (l2 < 0 ? 0 : l2)]; }
#pragma unroll
					for (int i = 0; i < 8; i++) {
						const int k = k0 + i;
						const float c = (k == u ? border_t(v) : src[i]) - p.ws[k < 64 ? k : 64];
						if (act && k <= u && k <= ku && c > best) { best = c; dir = LQ_UP; kk = k; }
					}
				}
				// (scoring pass under a table that saturates -- w_t(k) = w_t(tail) for every k >= tail, exp5: from 126 on --: the candidates
				// further back than `tail` columns all cost the same and are ONE running maximum per lane, tail_max; the scan stops at tail - 1)
				const int tail = (!FLOW && p.wt_tail > 0) ? p.wt_tail : LT + 1;
				int kv = dg - 1 < LT ? dg - 1 : LT;
				if (kv > tail - 1) kv = tail - 1;
				for (int k0 = 1; k0 <= kv; k0 += 8) {
					float src[8];
#pragma unroll
					for (int i = 0; i < 8; i++) { const int r2 = vc - (k0 + i); src[i] = Hm[(r2 < 1 ? 1 : r2) * HS + lane]; }
#pragma unroll
					for (int i = 0; i < 8; i++) {
						const int k = k0 + i;
						const float c = (k == v ? border_s(u) : src[i]) - p.wt[k <= LT ? k : LT];
						if (act && k <= v && k <= kv && c > best) { best = c; dir = LQ_LEFT; kk = k; }
					}
				}
				if (!FLOW && p.wt_tail > 0 && act && v >= tail) {
					const float x = v == tail ? border_s(u) : Hm[(v - tail) * HS + lane];   // H[u][v - tail] joins the candidates that far back
					tail_max = fmaxf(tail_max, x);
					const float c = tail_max - p.wt[tail];
					if (c > best) { best = c; dir = LQ_LEFT; }
				}
			}
			if (act) {
				if (GAP == 2) Hm[v * HS + lane] = best;
				if (FLOW) Dm[v * 64 + lane] = (int16_t)(dir | (ee << 2) | (fe << 3) | (kk << 4));
				// start cell: LOCAL over all cells, SEMIGLOBAL over the last row and the last column (start_cell of the oracle)
				if (!global && (local || u == len_s || v == LT) && best > best_v) { best_v = best; best_at = v; }
				a1 = best;
				if (GAP == 1) { e1 = e; f1 = f; }
			}
			if (GAP == 2) {   // the row just stored is read by other lanes on later steps: same wave, in order; pin the compiler to it
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			}
		}

		// ---- the aligner score (and where the traceback starts): first maximum in row-major order = smallest u, then smallest v
		float raw;
		int su = 0, sv = 0;
		if (global) {
			raw = __shfl(a1, len_s - 1, 64);   // H[len_s][len_t]
			su = len_s; sv = LT;
		} else {
			float m = ucol ? best_v : 0.0f;
#pragma unroll
			for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
			const unsigned long long hit = __ballot(ucol && best_v == m && m > 0.0f);
			raw = m;
			if (hit) {
				su = __builtin_ctzll(hit) + 1;
				sv = __shfl(best_at, su - 1, 64);
			}
		}
		if constexpr (!FLOW) {
			if (lane == 0) {
				const float boost = p.boost ? p.boost[g] : 1.0f;
				p.scores[g] = (raw / p.ref_total) * boost;   // Score::value with every query token's weight in the reference score (submatch_weight = 0)
				if (p.raw) p.raw[g] = raw;
			}
		} else {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			int16_t *map = p.mapping + (int64_t)item * p.out_stride;
			float *esim = p.edge_sim + (int64_t)item * p.out_stride;
			for (int j = lane; j < LT; j += 64) { map[j] = -1; esim[j] = 0.0f; }
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			if (lane == 0) {
				p.raw_out[item] = raw;
				int cu = su, cv = sv, state = 0;   // state: 0 = H, 1 = E, 2 = F (the affine solver's walk)
				while (cu > 0 && cv > 0) {
					const int rec = (int)(uint16_t)Dm[cv * 64 + (cu - 1)];
					const int d = rec & 3, k = rec >> 4;
					if (GAP == 1 && state == 1) { if (!((rec >> 2) & 1)) state = 0; cu--; continue; }
					if (GAP == 1 && state == 2) { if (!((rec >> 3) & 1)) state = 0; cv--; continue; }
					if (d == LQ_STOP) break;
					if (d == LQ_DIAG) {
						map[cv - 1] = (int16_t)(cu - 1);
						esim[cv - 1] = p.pos_s ? Su[(cv - 1) * 64 + (cu - 1)] : Sl[(cv - 1) * SW + (cu - 1)];   // the unmodified similarity of the edge (metric/alignment.h:339)
						cu--; cv--;
					} else if (d == LQ_UP) {
						if (GAP == 1) state = 1; else cu -= GAP == 2 ? k : 1;
					} else {
						if (GAP == 1) state = 2; else cv -= GAP == 2 ? k : 1;
					}
				}
			}
		}
		wave_lds_fence();   // the next slice overwrites the strip
	}
}

// general gaps, scoring pass: does the matrix fit the LDS behind the strip (leaving room for at least two workgroups per CU)?
extern "C" int32_t vk_longq_hm_in_lds(int32_t len_t, int32_t max_len) {
	const size_t hm = (size_t)(len_t + 1) * longq_stride(max_len) * 4;
	return vk_longq_lds_bytes(len_t, max_len, 0) + hm <= 78 * 1024 ? 1 : 0;
}

template <bool FLOW>
static hipError_t launch_longq(const VkLongqParams *p, int grid, size_t smem, hipStream_t stream) {
	const bool hl = !FLOW && p->gap_mode == 2 && p->scratch == nullptr;   // (the host leaves scratch null when the matrix goes to LDS)
	void (*kernel)(VkLongqParams) = p->gap_mode == 0 ? vk_longq_kernel<FLOW, 0> : p->gap_mode == 1 ? vk_longq_kernel<FLOW, 1> :
		hl ? vk_longq_kernel<FLOW, 2, true> : vk_longq_kernel<FLOW, 2>;
	if (hl) smem += (size_t)(p->len_t + 1) * p->s_stride * 4;
	if (smem > 64 * 1024) {
		const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
		if (e != hipSuccess) return e;
	}
	kernel<<<grid, 64, smem, stream>>>(*p);
	return hipGetLastError();
}

// workgroups (one wave each) of the scoring pass: as many as the CUs hold with this much LDS each, at most one per slice
extern "C" int32_t vk_longq_blocks(int32_t len_t, int32_t max_len, int64_t n_sent, int32_t hm_in_lds) {
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
	const size_t smem = vk_longq_lds_bytes(len_t, max_len, 0) + (hm_in_lds ? (size_t)(len_t + 1) * longq_stride(max_len) * 4 : 0);
	int per_cu = (int)((160 * 1024) / ((smem + 511) / 512 * 512 + 512));
	if (per_cu < 1) per_cu = 1;
	if (per_cu > 16) per_cu = 16;
	const int64_t want = (int64_t)cus * per_cu;
	return (int32_t)(n_sent < want ? (n_sent < 1 ? 1 : n_sent) : want);
}

// flow_k == 0: scores of all p->n_sent slices (grid: vk_longq_blocks; scratch regions: one per workgroup when gap_mode == 2);
// flow_k > 0: the flow_k winners of p->keys (one workgroup each, scratch regions likewise)
extern "C" hipError_t vk_launch_longq(const VkLongqParams *p, int32_t flow_k, hipStream_t stream) {
	const int max_len = p->s_stride;   // (the host sets the strip's stride from the corpus's longest slice: vk_longq_stride)
	if (flow_k > 0) return launch_longq<true>(p, flow_k, vk_longq_lds_bytes(p->len_t, max_len, 1), stream);
	const int hl = (p->gap_mode == 2 && p->scratch == nullptr) ? 1 : 0;
	return launch_longq<false>(p, vk_longq_blocks(p->len_t, max_len, p->n_sent, hl), vk_longq_lds_bytes(p->len_t, max_len, 0), stream);
}
