// vk_transport.hip -- exact transport of candidate slices (WRD, full WMD) and the similarity rows of winners.
#include "vk_common.hip.h"

// ---------------------------------------------------------------------------
// Word Rotator's Distance / full WMD, stage 2: exact EMD for the candidate slices.
// One wave per candidate restates the similarity rows (canonical arithmetic, transport_sim_rows) and
// solves the transportation problem (n <= 16 query tokens = supplies, m <= 64 slice tokens = demands) by
// successive shortest paths with potentials in double precision -- the algorithm of the oracle's vko_emd
// (oracle/vk_oracle.c), which stands in for pyemd's emd_hat_gd_metric<double>
// (vectorian/core/cpp/alignment/transport.h:70,125-126) -- with the Dijkstra step spread over the wave:
//   lane i owns demand i (its distance, potential, predecessor, remaining mass; column i of the costs and
//   flows in LDS); the supplies live in small LDS arrays read uniformly.  Queries of up to 64 tokens: 16 NQ supplies.
//   All supplies with remaining mass are sources and are relaxed together.  Demands are never settled
//   one by one: the next supply to settle is the minimum over (demand i, supply b with flow b -> i) of
//   dist[i] + reduced cost(i -> b), one in-lane loop over b and ONE wave reduction; it is final because
//   any shorter path would pass through another unsettled supply first.  The search ends when the
//   nearest demand with remaining mass is at most that far.
// The optimal cost is unique, so the score equals the oracle's up to the rounding of the final sums
// (the path taken among equal-cost alternatives may differ).  A serial one-lane version of the same
// solver took 8 - 20 ms per round of candidates; this one ~0.1 ms.
// ---------------------------------------------------------------------------


// minimum of x over the wave and a lane holding it (the lowest such lane).  The 16 lanes of a DPP row are folded with
// four cross-lane moves (quad swaps, half mirror, mirror: every lane ends with its row's minimum), the four rows through
// v_readlane -- no LDS round trips (a ds_bpermute butterfly on doubles costs twelve of them per reduction, and the solver
// below is one dependent chain of such reductions).
template <int CTRL>
__device__ __forceinline__ double dpp_min_f64(double x) {
	const long long b = __builtin_bit_cast(long long, x);
	const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
	const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
	const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
	const long long b2 = ((long long)hi2 << 32) | (long long)(unsigned)lo2;
	return fmin(x, __builtin_bit_cast(double, b2));
}

__device__ __forceinline__ double readlane_f64(double x, int src) {
	const long long b = __builtin_bit_cast(long long, x);
	const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
	const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
	return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}

__device__ __forceinline__ double wave_argmin_f64(double x, int lane, int &at) {
	double m = x;
	m = dpp_min_f64<0xB1>(m);    // quad_perm [1,0,3,2]
	m = dpp_min_f64<0x4E>(m);    // quad_perm [2,3,0,1]
	m = dpp_min_f64<0x141>(m);   // row_half_mirror
	m = dpp_min_f64<0x140>(m);   // row_mirror
	m = fmin(fmin(readlane_f64(m, 0), readlane_f64(m, 16)), fmin(readlane_f64(m, 32), readlane_f64(m, 48)));
	const unsigned long long hit = __ballot(x == m);
	at = hit ? __builtin_ctzll(hit) : 0;
	return m;
}

// similarity rows of one slice against the nq query tiles -> S[row][16 nq] (row 0 = first token of the slice's first tile or,
// static layout, of the slice); returns the row of the slice's first token.  Tag-weighted modifier applied when p.pos_s.
// (STRIDE: floats per row of S, at least 16 NQ.)
// CANON: the canonical arithmetic (sim_canon, vk_common.hip.h: the oracle's own sums) -- candidates of the exact solvers and
// winners' rows, whose costs, plans and flows then rest on the same similarities as the oracle's, bit for bit; the bound pass
// over all long slices (vk_long_bound_kernel) keeps the MFMA form.
template <int NQ, int STRIDE = 16 * NQ, bool CANON = true>
__device__ __forceinline__ int transport_sim_rows(const VkWrdParams &p, float *S, int t_a, int t_b, int lane, uint8_t *canon = nullptr) {
	constexpr int N = STRIDE;
	const int m = t_b - t_a;
	const bool is_static = p.layout == VK_DEV_LAYOUT_STATIC;
	if constexpr (CANON) {
		// 16 rows x 16 columns per step (sim_canon16, staged through the wave's `canon` bytes of LDS): lane l -> row l & 15,
		// columns 4 (l >> 4) .. + 3 of each query tile; static layout: the rows of 16 consecutive tokens gathered from the vocabulary
		// (past the slice's end its first token again, never read)
		const int tile0 = is_static ? 0 : t_a >> 4;
		const int ntiles = is_static ? (m + 15) >> 4 : ((t_b + 15) >> 4) - tile0;
		for (int ti = 0; ti < ntiles; ti++) {
			const int tok = is_static ? t_a + (ti * 16 + (lane & 15) < m ? ti * 16 + (lane & 15) : 0) : (tile0 + ti) * 16 + (lane & 15);
			const int id = is_static ? p.tok_id[tok] : 0;
			const uint8_t *xrow = is_static ? canon_row_ptr_static(p.tiles, p.tile_bytes, id) : canon_row_ptr(p.tiles + (int64_t)(tile0 + ti) * p.tile_bytes, lane);
			const int ps = p.pos_s ? p.pos_s[tok] : 0;
#pragma unroll 1
			for (int b = 0; b < NQ; b++) {
				float val[4];
				sim_canon16(xrow, p.qtile + (int64_t)b * p.tile_bytes, p.nk32, p.tail, p.d, p.prec, canon, lane, val);
				const int c0 = 16 * b + (lane >> 4) * 4;
#pragma unroll
				for (int r = 0; r < 4; r++) {
					val[r] = (is_static && p.q_ids && p.q_ids[c0 + r] == id) ? 1.0f : clip01(val[r]);   // sim[id(t_j)][j] = 1 (metric/static.cpp:58-67)
					if (p.pos_s) val[r] = tag_weighted(val[r], p.tw[c0 + r], ps, p.tpos[c0 + r], p.tw_keep, p.tw_threshold);
				}
				*reinterpret_cast<float4 *>(S + (ti * 16 + (lane & 15)) * N + c0) = make_float4(val[0], val[1], val[2], val[3]);
			}
		}
		if (is_static && p.qid_bits && p.pos_s) {   // tag-weighted vocabulary transport: cells upstream writes twice (static_vocab_fixup, vk_common.hip.h)
			wave_lds_fence();
			static_vocab_fixup<64, true>(S, N, m, p.len_t, p.tok_id + t_a, p.tag_s + t_a, p.pos_s + t_a, p.table, p.table_stride,
				p.qid_bits, p.qkey, p.tw, p.tpos, p.tw_keep, p.tw_threshold, lane, p.tiles, p.tile_bytes, p.qtile, p.d, p.prec, p.q_ids);
		}
		return is_static ? 0 : t_a - tile0 * 16;
	}
	if (is_static) {
		for (int it = 0; it * 16 < m; it++) {
			const int tk = it * 16 + (lane >> 2);
			if (tk < m) {
				const int id = p.tok_id[t_a + tk];
				const int ps = p.pos_s ? p.pos_s[t_a + tk] : 0;
#pragma unroll
				for (int b = 0; b < NQ; b++) {
					const int c0 = 16 * b + (lane & 3) * 4;
					float4 vq = *reinterpret_cast<const float4 *>(p.table + b * p.table_stride + (int64_t)id * 16 + (lane & 3) * 4);
					if (p.pos_s) {
						vq.x = tag_weighted(vq.x, p.tw[c0 + 0], ps, p.tpos[c0 + 0], p.tw_keep, p.tw_threshold);
						vq.y = tag_weighted(vq.y, p.tw[c0 + 1], ps, p.tpos[c0 + 1], p.tw_keep, p.tw_threshold);
						vq.z = tag_weighted(vq.z, p.tw[c0 + 2], ps, p.tpos[c0 + 2], p.tw_keep, p.tw_threshold);
						vq.w = tag_weighted(vq.w, p.tw[c0 + 3], ps, p.tpos[c0 + 3], p.tw_keep, p.tw_threshold);
					}
					*reinterpret_cast<float4 *>(S + tk * N + c0) = vq;
				}
			}
		}
		if (p.qid_bits && p.pos_s) {
			wave_lds_fence();
			static_vocab_fixup<64>(S, N, m, p.len_t, p.tok_id + t_a, p.tag_s + t_a, p.pos_s + t_a, p.table, p.table_stride,
				p.qid_bits, p.qkey, p.tw, p.tpos, p.tw_keep, p.tw_threshold, lane);
		}
		return 0;
	}
	const int tile0 = t_a >> 4;
	const int ntiles = ((t_b + 15) >> 4) - tile0;
	for (int ti = 0; ti < ntiles; ti++) {
		const int ps = p.pos_s ? p.pos_s[(tile0 + ti) * 16 + (lane & 15)] : 0;
#pragma unroll
		for (int b = 0; b < NQ; b++) {
			f32x4 acc = sim_tile_generic(p.qtile + (int64_t)b * p.tile_bytes, p.tiles + (int64_t)(tile0 + ti) * p.tile_bytes, p.nk32, p.tail, lane, p.prec);
			if (p.pos_s) {
				const int c0 = 16 * b + (lane >> 4) * 4;
#pragma unroll
				for (int r = 0; r < 4; r++) acc[r] = tag_weighted(acc[r], p.tw[c0 + r], ps, p.tpos[c0 + r], p.tw_keep, p.tw_threshold);
			}
			*reinterpret_cast<f32x4 *>(S + (ti * 16 + (lane & 15)) * N + 16 * b + (lane >> 4) * 4) = acc;
		}
	}
	return t_a - tile0 * 16;
}

// NQ = 16-column blocks of the query (1: at most 16 tokens .. 4: at most 64).  NQ = 1 keeps the lane's cost and flow columns and
// the supplies' potentials in registers; the wider forms read them from LDS (3 x 16 NQ doubles per lane do not fit).
template <int NQ>
__global__ __launch_bounds__(64) void vk_wrd_exact_kernel(VkWrdParams p) {
	constexpr int N = 16 * NQ;
	constexpr bool REG = NQ == 1;
	extern __shared__ double vk_smem_f64[];
	uint8_t *canon = reinterpret_cast<uint8_t *>(vk_smem_f64);   // staging of sim_canon16 (VK_CANON_LDS bytes)
	double *Cm = reinterpret_cast<double *>(canon + VK_CANON_LDS);   // Cm[j * 64 + i]: cost supply j -> demand i
	double *fl = Cm + N * 64;               // flow
	double *sup = fl + N * 64, *pot_s = sup + N, *dist_s = pot_s + N;
	int *pred_s = reinterpret_cast<int *>(dist_s + N), *settled = pred_s + N;
	float *S = reinterpret_cast<float *>(settled + N);   // [(VK_DEV_MAX_SENT_LEN + 32)][N]

	const int lane = threadIdx.x;
	const int w = blockIdx.x;
	const uint64_t key = p.keys[w];
	if (key == 0) return;
	const int64_t g = (int64_t)(uint32_t)(key & 0xffffffffu);
	const int t_a = p.sent_start[g], t_b = p.sent_end[g];
	const int m = t_b - t_a, n = p.len_t;
	if (m > VK_DEV_MAX_SENT_LEN) return;   // long slices: vk_wrd_exact_long_kernel

	const int rowbase = transport_sim_rows<NQ>(p, S, t_a, t_b, lane, canon);
	wave_lds_fence();
	const float *Sm = S + rowbase * N;
	const bool has = lane < m;
	const double EPS = 1e-13, INF = __builtin_inf();

	// masses (wrd.h:99-102) and costs (:104-109)
	double dem = 0.0;
	if (p.mass_mode == 0) {
		const bool by_id = p.layout == VK_DEV_LAYOUT_STATIC;       // static layout: magnitudes of the vocabulary entries
		// one load per lane; the sum runs in position order, as upstream (the wave holds one slice: m is uniform)
		const float mine = has ? (by_id ? p.mag[p.tok_id[t_a + lane]] : p.mag[t_a + lane]) : 0.0f;
		float sum_s = 0.0f;
		for (int i = 0; i < m; i++) sum_s += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), i));
		if (has) dem = (double)(p.raw_masses ? mine : mine / sum_s);
		if (lane < N) sup[lane] = lane < n ? (double)p.qmass[lane] : 0.0;
	} else {
		// bags of words over positions: 1 per token (bow), or 1/len (nbow, bow.h:262-270)
		const float wt = p.mass_mode == 1 ? 1.0f / (float)n : 1.0f;
		const float wsn = p.mass_mode == 1 ? 1.0f / (float)m : 1.0f;
		if (has) dem = (double)wsn;
		if (lane < N) sup[lane] = lane < n ? (double)wt : 0.0;
	}
	if (lane < N) pot_s[lane] = 0.0;
	for (int j = 0; j < n; j++) {
		float d = has ? 1.0f - Sm[lane * N + j] : 0.0f;
		if (!(d > 0.0f)) d = 0.0f;
		Cm[j * 64 + lane] = (double)d;
		fl[j * 64 + lane] = 0.0;
	}
	double pot_d = 0.0;
	wave_lds_fence();
	// REG: this lane's column of the costs stays in registers; its column of the flows and the supplies' potentials are
	// refreshed from LDS once per augmentation (LDS holds the copies that are indexed dynamically)
	double Cr[REG ? N : 1], Fr[REG ? N : 1], Ps[REG ? N : 1];
	if constexpr (REG) {
#pragma unroll
		for (int j = 0; j < N; j++) Cr[j] = j < n ? Cm[j * 64 + lane] : 0.0;
	}

	// ---- start: dual potentials pot_d[i] = min_j C[j][i] (all reduced costs stay >= 0, the nearest supply's arc is
	// tight) and as much flow on the tight arcs as the supplies allow, demands served in lane order.  Flow on tight
	// arcs keeps the optimality conditions of successive shortest paths, and most of the mass is placed here: the
	// augmentations that remain are the ones that actually reroute.
	// (balanced problems only: with unequal totals the side in excess is served selectively, which a greedy start cannot know)
	if (p.raw_masses == 0 && p.mass_mode != 2) {
		double cmin = INF;
		int jmin = -1;
		for (int j = 0; j < n; j++) {
			const double cj = Cm[j * 64 + lane];
			if (cj < cmin) { cmin = cj; jmin = j; }
		}
		if (has && jmin >= 0) pot_d = cmin;
		for (int j = 0; j < n; j++) {
			const double want = (has && jmin == j) ? dem : 0.0;
			double incl = want;
#pragma unroll
			for (int off = 1; off < 64; off <<= 1) {
				const double t = __shfl_up(incl, off, 64);
				if (lane >= off) incl += t;
			}
			const double sj = sup[j];
			const double room = sj - (incl - want);
			const double give = fmin(want, room > 0.0 ? room : 0.0);
			if (give > 0.0) { fl[j * 64 + lane] = give; dem -= give; }
			const double total = __shfl(incl, 63, 64);
			wave_lds_fence();
			if (lane == 0) sup[j] = sj > total ? sj - total : 0.0;
		}
		wave_lds_fence();
	}

	for (int iter = 0; iter < 4000 * NQ; iter++) {
		// ---- sources: every supply with remaining mass; relax them all
		bool any_sup = false;
		double dist_d = INF;
		int pred_d = -1;
		uint64_t smask = 0;   // settled supplies (uniform)
		auto relax_source = [&](int j, double cj, double psj) {
			const bool src = sup[j] > EPS;
			any_sup |= src;
			if (src) smask |= 1ull << j;
			if (lane == 0) { settled[j] = src ? 1 : 0; dist_s[j] = src ? 0.0 : INF; pred_s[j] = -1; }
			double rc = cj + psj - pot_d;
			if (rc < 0) rc = 0;
			if (src && rc < dist_d) { dist_d = rc; pred_d = j; }
		};
		if constexpr (REG) {
#pragma unroll
			for (int j = 0; j < N; j++) {
				if (j < n) {
					Fr[j] = fl[j * 64 + lane];
					Ps[j] = pot_s[j];
					relax_source(j, Cr[j], Ps[j]);
				}
			}
		} else {
			for (int j = 0; j < n; j++) relax_source(j, Cm[j * 64 + lane], pot_s[j]);
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
			auto candidate = [&](int b, double cb_, double fb, double psb) {
				double rc = pot_d - psb - cb_;
				if (rc < 0) rc = 0;
				const double cand = dist_d + rc;
				const bool ok = !((smask >> b) & 1) && fb > EPS && cand < best;   // dist_d = inf gives cand = inf: never < best
				best = ok ? cand : best;
				bb = ok ? b : bb;
			};
			if constexpr (REG) {
#pragma unroll
				for (int b = 0; b < N; b++) {
					if (b < n) candidate(b, Cr[b], Fr[b], Ps[b]);
				}
			} else {
				for (int b = 0; b < n; b++) candidate(b, Cm[b * 64 + lane], fl[b * 64 + lane], pot_s[b]);
			}
			const double cmin = wave_argmin_f64(best, lane, c_lane);
			if (fd <= cmin) {
				if (fd < INF) { target = fd_lane; dt = fd; }
				break;
			}
			const int cb = __builtin_amdgcn_readlane(bb, c_lane);
			smask |= 1ull << cb;
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
		p.val_out[w] = (raw / p.ref_total) * boost;
	}
	if (p.plan_out) {
		const int R = p.rows_len > 0 ? p.rows_len : 64;
		for (int j = 0; j < N; j++)
			p.plan_out[((int64_t)w * N + j) * R + lane] = (has && j < n) ? (float)fl[j * 64 + lane] : 0.0f;
	}
}

// ---------------------------------------------------------------------------
// The same solver for slices of 65 .. VK_DEV_MAX_LONG_LEN tokens (long sentences, sliding windows; queries of at most 16
// tokens): upstream sizes its transport problems by the document's longest sentence (metric/alignment.h:357-358,
// alignment/wrd.h:76-85).  A demand no longer has a lane of its own: lane l owns demands l, l + 64, ..; distances, potentials,
// predecessors and remaining masses live in LDS arrays, every per-demand step is a strided loop.  No greedy start (plain
// successive shortest paths from the zero flow; the optimal cost is unique).  Such slices are rare: a fallback that keeps
// corpora with a long sentence on the device, one wave per candidate, not a tuned kernel.  Costs are kept as floats (they
// are floats: 1 - S), flows and potentials in double.
// ---------------------------------------------------------------------------
constexpr int VK_WRDL_M = VK_DEV_MAX_LONG_LEN;        // demands (slice tokens)

// GLOBAL: the arrays indexed by (supply, demand) live in a scratch region of global memory (a query of up to 64 tokens against
// 512 demands: 256 KB of flows alone); lanes then exchange them through L1 / L2, so the fence is a workgroup-scope one
template <bool GLOBAL>
__device__ __forceinline__ void wrdl_fence() {
	if constexpr (GLOBAL) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	} else wave_lds_fence();
}

// NQ = 1: queries of at most 16 tokens, everything in LDS (148 KB).  NQ = 4: queries of 17..64 tokens; flows, costs and
// similarity rows in the workgroup's scratch region (p.scratch), the per-demand and per-supply vectors in LDS; a workgroup
// walks the candidates with a grid stride.
template <int NQ>
__global__ __launch_bounds__(64) void vk_wrd_exact_long_kernel(VkWrdParams p) {
	constexpr int N = 16 * NQ, M = VK_WRDL_M;
	constexpr bool GLOBAL = NQ > 1;
	extern __shared__ double vk_smem_f64[];
	uint8_t *canon = reinterpret_cast<uint8_t *>(vk_smem_f64);   // staging of sim_canon16 (VK_CANON_LDS bytes)
	double *dem = reinterpret_cast<double *>(canon + VK_CANON_LDS), *pot_d = dem + M, *dist_d = pot_d + M;   // [M] each
	double *sup = dist_d + M, *pot_s = sup + N, *dist_s = pot_s + N;    // [N] each
	int *pred_d = reinterpret_cast<int *>(dist_s + N);                  // [M]
	int *pred_s = pred_d + M, *settled = pred_s + N;                    // [N]
	double *fl;      // [N][M] flow
	float *Cm, *S;   // [N][M] cost supply j -> demand i; [(M + 32)][N] similarity rows
	if constexpr (GLOBAL) {
		uint8_t *base = p.scratch + (int64_t)blockIdx.x * p.scratch_stride;
		fl = reinterpret_cast<double *>(base);
		Cm = reinterpret_cast<float *>(fl + N * M);
		S = Cm + N * M;
	} else {
		fl = reinterpret_cast<double *>(settled + N);   // 8-byte aligned: the ints before it come in even numbers
		Cm = reinterpret_cast<float *>(fl + N * M);
		S = Cm + N * M;
	}

	const int lane = threadIdx.x;
	const double EPS = 1e-13, INF = __builtin_inf();
	for (int w = blockIdx.x; w < p.n_cand; w += gridDim.x) {
	const uint64_t key = p.keys[w];
	if (key == 0) continue;
	const int64_t g = (int64_t)(uint32_t)(key & 0xffffffffu);
	const int t_a = p.sent_start[g], t_b = p.sent_end[g];
	const int m = t_b - t_a, n = p.len_t;
	if (m <= VK_DEV_MAX_SENT_LEN || m > M) continue;   // short slices are vk_wrd_exact_kernel's

	int rowbase;
	if constexpr (NQ == 1) rowbase = transport_sim_rows<1>(p, S, t_a, t_b, lane, canon);
	else rowbase = p.nq == 2 ? transport_sim_rows<2, N>(p, S, t_a, t_b, lane, canon) : p.nq == 3 ? transport_sim_rows<3, N>(p, S, t_a, t_b, lane, canon) : transport_sim_rows<4, N>(p, S, t_a, t_b, lane, canon);
	wrdl_fence<GLOBAL>();
	const float *Sm = S + rowbase * N;

	// masses (wrd.h:99-102) and costs (:104-109)
	if (p.mass_mode == 0) {
		const bool by_id = p.layout == VK_DEV_LAYOUT_STATIC;
		float sum_s = 0.0f;
		for (int i = 0; i < m; i++) sum_s += by_id ? p.mag[p.tok_id[t_a + i]] : p.mag[t_a + i];   // position order, as upstream (uniform loop)
		for (int i = lane; i < m; i += 64) {
			const float mine = by_id ? p.mag[p.tok_id[t_a + i]] : p.mag[t_a + i];
			dem[i] = (double)(p.raw_masses ? mine : mine / sum_s);
		}
		if (lane < N) sup[lane] = lane < n ? (double)p.qmass[lane] : 0.0;
	} else {
		const float wt = p.mass_mode == 1 ? 1.0f / (float)n : 1.0f;
		const float wsn = p.mass_mode == 1 ? 1.0f / (float)m : 1.0f;
		for (int i = lane; i < m; i += 64) dem[i] = (double)wsn;
		if (lane < N) sup[lane] = lane < n ? (double)wt : 0.0;
	}
	if (lane < N) pot_s[lane] = 0.0;
	for (int i = lane; i < m; i += 64) {
		pot_d[i] = 0.0;
		for (int j = 0; j < n; j++) {
			float d = 1.0f - Sm[i * N + j];
			if (!(d > 0.0f)) d = 0.0f;
			Cm[j * M + i] = d;
			fl[j * M + i] = 0.0;
		}
	}
	wrdl_fence<GLOBAL>();

	for (int iter = 0; iter < 40000 * NQ; iter++) {
		// ---- sources: every supply with remaining mass; relax them all
		bool any_sup = false;
		uint64_t smask = 0;   // settled supplies (uniform)
		for (int j = 0; j < n; j++) {
			const bool src = sup[j] > EPS;
			any_sup |= src;
			if (src) smask |= 1ull << j;
			if (lane == 0) { settled[j] = src ? 1 : 0; dist_s[j] = src ? 0.0 : INF; pred_s[j] = -1; }
		}
		bool any_dem = false;
		for (int i = lane; i < m; i += 64) {
			double dd = INF;
			int pd = -1;
			for (int j = 0; j < n; j++) {
				if (!((smask >> j) & 1)) continue;
				double rc = (double)Cm[j * M + i] + pot_s[j] - pot_d[i];
				if (rc < 0) rc = 0;
				if (rc < dd) { dd = rc; pd = j; }
			}
			dist_d[i] = dd; pred_d[i] = pd;
			any_dem |= dem[i] > EPS;
		}
		if (!any_sup || !__any(any_dem)) break;
		wrdl_fence<GLOBAL>();

		int target = -1;
		double dt = INF;
		for (int round = 0; round <= n; round++) {
			// nearest demand with remaining mass, and the cheapest way back into an unsettled supply
			double fd_l = INF, best = INF;
			int fd_i = -1, bb = -1, bi = -1;
			for (int i = lane; i < m; i += 64) {
				const double di = dist_d[i];
				if (dem[i] > EPS && di < fd_l) { fd_l = di; fd_i = i; }
				for (int b = 0; b < n; b++) {
					if ((smask >> b) & 1) continue;
					if (!(fl[b * M + i] > EPS)) continue;
					double rc = pot_d[i] - pot_s[b] - (double)Cm[b * M + i];
					if (rc < 0) rc = 0;
					const double cand = di + rc;
					if (cand < best) { best = cand; bb = b; bi = i; }
				}
			}
			int fd_lane, c_lane;
			const double fd = wave_argmin_f64(fd_l, lane, fd_lane);
			const double cmin = wave_argmin_f64(best, lane, c_lane);
			if (fd <= cmin) {
				if (fd < INF) { target = __builtin_amdgcn_readlane(fd_i, fd_lane); dt = fd; }
				break;
			}
			const int cb = __builtin_amdgcn_readlane(bb, c_lane);
			const int ci = __builtin_amdgcn_readlane(bi, c_lane);
			smask |= 1ull << cb;
			if (lane == 0) { settled[cb] = 1; dist_s[cb] = cmin; pred_s[cb] = ci; }
			wrdl_fence<GLOBAL>();
			for (int i = lane; i < m; i += 64) {
				double rc = (double)Cm[cb * M + i] + pot_s[cb] - pot_d[i];
				if (rc < 0) rc = 0;
				const double nd = cmin + rc;
				if (nd < dist_d[i]) { dist_d[i] = nd; pred_d[i] = cb; }
			}
			wrdl_fence<GLOBAL>();
		}
		if (target < 0) break;

		// ---- potentials: pot += min(dist, dt)
		for (int i = lane; i < m; i += 64) { const double di = dist_d[i]; pot_d[i] += di < dt ? di : dt; }
		if (lane < n) pot_s[lane] += (settled[lane] && dist_s[lane] < dt) ? dist_s[lane] : dt;
		wrdl_fence<GLOBAL>();

		// ---- bottleneck along target <- supply <- demand <- ... <- source, then augment (uniform walk)
		double delta = dem[target];
		int x = target;
		for (int hop = 0; hop <= n; hop++) {
			const int a = pred_d[x];
			const int ps = pred_s[a];
			if (ps < 0) { delta = fmin(delta, sup[a]); break; }
			delta = fmin(delta, fl[a * M + ps]);
			x = ps;
		}
		wrdl_fence<GLOBAL>();
		if (lane == 0) {
			dem[target] -= delta;
			x = target;
			for (int hop = 0; hop <= n; hop++) {
				const int a = pred_d[x];
				const int ps = pred_s[a];
				fl[a * M + x] += delta;
				if (ps < 0) { sup[a] -= delta; break; }
				fl[a * M + ps] -= delta;
				x = ps;
			}
		}
		wrdl_fence<GLOBAL>();
	}

	// score = sum((1 - D) * G) / sum(G) (wrd.h:139), G as float
	double num = 0.0, den = 0.0;
	for (int i = lane; i < m; i += 64)
		for (int j = 0; j < n; j++) {
			const float gq = (float)fl[j * M + i];
			num += (double)((1.0f - Cm[j * M + i]) * gq);
			den += (double)gq;
		}
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) { num += __shfl_xor(num, off, 64); den += __shfl_xor(den, off, 64); }
	if (lane == 0) {
		const float raw = den > 0.0 ? (float)(num / den) : 0.0f;
		const float boost = p.boost ? p.boost[g] : 1.0f;
		p.raw_out[w] = raw;
		p.val_out[w] = (raw / p.ref_total) * boost;
	}
	if (p.plan_out && p.rows_len >= m) {   // the plan of a winner, for the host to state its flow: [16 nq x rows_len]
		const int Wp = 16 * p.nq;
		for (int i = lane; i < m; i += 64)
			for (int j = 0; j < n; j++) p.plan_out[((int64_t)w * Wp + j) * p.rows_len + i] = (float)fl[j * M + i];
	}
	wrdl_fence<GLOBAL>();   // the next candidate reuses the arrays
	}
}

static size_t transport_long_lds_bytes(int nq) {
	const size_t N = 16 * (size_t)nq, M = VK_WRDL_M;
	const size_t vectors = 3 * M * 8 + 3 * N * 8 + M * 4 + 2 * N * 4;
	return VK_CANON_LDS + (nq > 1 ? vectors : vectors + N * M * 8 + N * M * 4 + (M + 32) * N * 4);
}

extern "C" int vk_wrd_long_blocks(void) { return 256; }

extern "C" size_t vk_wrd_long_scratch_bytes(void) {
	const size_t N = 64, M = VK_WRDL_M;
	return (N * M * 8 + N * M * 4 + (M + 32) * N * 4 + 255) / 256 * 256;
}

// candidates whose slice has more than VK_DEV_MAX_SENT_LEN tokens (the others are skipped at once)
extern "C" hipError_t vk_launch_wrd_exact_long(const VkWrdParams *pp, int32_t n_cand, hipStream_t stream) {
	VkWrdParams p = *pp;
	p.n_cand = n_cand;
	if (n_cand < 1) return hipSuccess;
	if (p.nq <= 1) {
		const size_t smem = transport_long_lds_bytes(1);
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(vk_wrd_exact_long_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
		if (e != hipSuccess) return e;
		vk_wrd_exact_long_kernel<1><<<n_cand, 64, smem, stream>>>(p);
		return hipGetLastError();
	}
	if (!p.scratch || p.scratch_stride < (int64_t)vk_wrd_long_scratch_bytes()) return hipErrorInvalidValue;
	const int blocks = n_cand < vk_wrd_long_blocks() ? n_cand : vk_wrd_long_blocks();
	vk_wrd_exact_long_kernel<4><<<blocks, 64, transport_long_lds_bytes(4), stream>>>(p);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Exact transport with a query of 17..64 tokens, stage 1 for the slices of more than VK_DEV_MAX_SENT_LEN tokens (the short
// ones: transport_bound32 in vk_score32_kernel): the nearest-neighbour relaxation in both directions, the same bound with the
// same combination of the two sides.  One wave per long slice, its similarity rows [m x 64] in LDS (139 KB).  A fallback that
// keeps such corpora on the device, not a roofline kernel.  The three padding rows of the slice's group get the "empty" score.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void vk_long_bound_kernel(VkWrdParams p) {
	constexpr int N = 64, M = VK_WRDL_M;
	extern __shared__ double vk_smem_f64[];
	float *S = reinterpret_cast<float *>(vk_smem_f64);   // [(M + 32)][N]
	const int lane = threadIdx.x;
	const float BIG = 3.402823466e+38F;
	for (int gi = blockIdx.x; gi < p.n_list; gi += gridDim.x) {
		const int row = p.group_list[gi] * 4;
		const int t_a = p.sent_start[row], t_b = p.sent_end[row];
		const int m = t_b - t_a, n = p.len_t;
		if (lane >= 1 && lane < 4 && row + lane < p.n_entries) { p.scores[row + lane] = VK_NEG_INF; p.raw[row + lane] = VK_NEG_INF; }
		if (m < 1 || m > M) {
			if (lane == 0) { p.scores[row] = VK_NEG_INF; p.raw[row] = VK_NEG_INF; }
			continue;
		}
		const int rowbase = p.nq == 2 ? transport_sim_rows<2, N, false>(p, S, t_a, t_b, lane) : p.nq == 3 ? transport_sim_rows<3, N, false>(p, S, t_a, t_b, lane) : transport_sim_rows<4, N, false>(p, S, t_a, t_b, lane);
		wave_lds_fence();
		const float *Sm = S + rowbase * N;
		const bool by_id = p.layout == VK_DEV_LAYOUT_STATIC;
		// slice side: every slice token to its nearest query token
		float sum_s = 0.0f, lb1n = 0.0f;
		for (int i = lane; i < m; i += 64) {
			const float mg = p.mag ? (by_id ? p.mag[p.tok_id[t_a + i]] : p.mag[t_a + i]) : 1.0f;
			float rmin = BIG;
			for (int j = 0; j < n; j++) rmin = fminf(rmin, fmaxf(1.0f - Sm[i * N + j], 0.0f));
			sum_s += mg;
			lb1n += mg * rmin;
		}
		// query side: every query token to its nearest slice token
		float x = 0.0f;
		if (lane < n) {
			float cmin = BIG;
			for (int i = 0; i < m; i++) cmin = fminf(cmin, fmaxf(1.0f - Sm[i * N + lane], 0.0f));
			x = p.qmass[lane] * cmin;
		}
#pragma unroll
		for (int off = 32; off >= 1; off >>= 1) { sum_s += __shfl_xor(sum_s, off, 64); lb1n += __shfl_xor(lb1n, off, 64); x += __shfl_xor(x, off, 64); }
		const float lb1 = sum_s > 0.0f ? lb1n / sum_s : 0.0f;
		float lb;
		if (p.wmd_bound == 2) lb = n <= m ? x / (float)n : lb1;                                       // unit masses: the shorter side ships everything
		else if (p.wrd_raw_total > 0.0f) lb = p.wrd_raw_total <= sum_s ? x / p.wrd_raw_total : lb1;  // magnitudes as they are: the lighter side
		else lb = fmaxf(x, lb1);                                                                      // both sides ship 1
		if (!(sum_s > 0.0f)) lb = 0.0f;
		const float raw = fminf(1.0f - lb * (1.0f - 4e-6f) + 3e-5f, 1.0f);
		if (lane == 0) {
			const float boost = p.boost ? p.boost[row] : 1.0f;
			p.scores[row] = (raw / p.ref_total) * boost;
			p.raw[row] = raw;
		}
		wave_lds_fence();
	}
}

extern "C" hipError_t vk_launch_long_bound(const VkWrdParams *p, hipStream_t stream) {
	if (p->n_list < 1) return hipSuccess;
	const size_t smem = (size_t)(VK_WRDL_M + 32) * 64 * 4;
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(vk_long_bound_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
	if (e != hipSuccess) return e;
	const int blocks = p->n_list < 2048 ? p->n_list : 2048;
	vk_long_bound_kernel<<<blocks, 64, smem, stream>>>(*p);
	return hipGetLastError();
}

// similarity rows of the winners of a transport query, for the host to state their flows: [64][16 nq] per winner
template <int NQ>
__global__ __launch_bounds__(64) void vk_rows_kernel(VkWrdParams p) {
	constexpr int N = 16 * NQ;
	extern __shared__ double vk_smem_f64[];
	uint8_t *canon = reinterpret_cast<uint8_t *>(vk_smem_f64);   // staging of sim_canon16 (VK_CANON_LDS bytes)
	float *S = reinterpret_cast<float *>(canon + VK_CANON_LDS);   // [(R + 32)][N]
	const int lane = threadIdx.x;
	const int w = blockIdx.x;
	const int R = p.rows_len > 0 ? p.rows_len : 64;
	float *out = p.rows_out + (int64_t)w * R * N;
	const uint64_t key = p.keys[w];
	int m = 0, rowbase = 0;
	if (p.cand_query) {   // a batch: every candidate against its own query (static layout: its own token ids)
		p.qtile += (int64_t)p.cand_query[w] * p.qtile_stride;
		if (p.q_ids) p.q_ids += (int64_t)p.cand_query[w] * p.q_ids_stride;
	}
	if (key != 0) {
		const int64_t g = (int64_t)(uint32_t)(key & 0xffffffffu);
		const int t_a = p.sent_start[g], t_b = p.sent_end[g];
		m = t_b - t_a;
		if (m < 1 || m > R) m = 0;
		else rowbase = transport_sim_rows<NQ>(p, S, t_a, t_b, lane, canon);
	}
	wave_lds_fence();
	for (int i = lane; i < R * N; i += 64) out[i] = i / N < m ? S[rowbase * N + i] : 0.0f;
}

// The same rows for winners of any length (whole documents as slices): one wave per (winner, 16 tokens) instead of one per winner
// -- a 5,000-token winner is 313 independent tiles, not a serial sweep -- each writing its tokens' rows where they belong (the host
// zeroes rows_out first: rows past a winner's end stay 0).  The canonical arithmetic, sim[id(t_j)][j] = 1 and the tag-weighted
// modifier as transport_sim_rows; no vocabulary fixup in this form (vk_validate_query refuses such queries over long slices).
// Also the similarities the traceback kernel of long winners runs its recurrence on (vk_wide_kernel FLOW, dp_rows).
template <int NQ>
__global__ __launch_bounds__(64) void vk_canon_rows_kernel(VkWrdParams p) {
	constexpr int N = 16 * NQ;
	extern __shared__ double vk_smem_f64[];
	uint8_t *canon = reinterpret_cast<uint8_t *>(vk_smem_f64);
	const int lane = threadIdx.x;
	const int w = blockIdx.y, ti = blockIdx.x;
	const int R = p.rows_len;
	const uint64_t key = p.keys[w];
	if (key == 0) return;
	const int64_t g = (int64_t)(uint32_t)(key & 0xffffffffu);
	const int t_a = p.sent_start[g], t_b = p.sent_end[g];
	const int m = t_b - t_a;
	if (m < 1 || m > R) return;
	const bool is_static = p.layout == VK_DEV_LAYOUT_STATIC;
	const int tile0 = is_static ? 0 : t_a >> 4;
	const int ntiles = is_static ? (m + 15) >> 4 : ((t_b + 15) >> 4) - tile0;
	if (ti >= ntiles) return;
	// this lane's token: row lane & 15 of the tile (static layout: the slice's own 16-token steps; past its end its first token, not written)
	const int rel = is_static ? ti * 16 + (lane & 15) : (tile0 + ti) * 16 + (lane & 15) - t_a;   // row of the slice
	const bool live = rel >= 0 && rel < m;
	const int tok = is_static ? t_a + (live ? rel : 0) : (tile0 + ti) * 16 + (lane & 15);
	const int id = is_static ? p.tok_id[tok] : 0;
	const uint8_t *xrow = is_static ? canon_row_ptr_static(p.tiles, p.tile_bytes, id) : canon_row_ptr(p.tiles + (int64_t)(tile0 + ti) * p.tile_bytes, lane);
	const int ps = (p.pos_s && live) ? p.pos_s[tok] : 0;
	float *out = p.rows_out + ((int64_t)w * R + (live ? rel : 0)) * N;
#pragma unroll 1
	for (int b = 0; b < NQ; b++) {
		float val[4];
		sim_canon16(xrow, p.qtile + (int64_t)b * p.tile_bytes, p.nk32, p.tail, p.d, p.prec, canon, lane, val);
		const int c0 = 16 * b + (lane >> 4) * 4;
#pragma unroll
		for (int r = 0; r < 4; r++) {
			val[r] = (is_static && p.q_ids && p.q_ids[c0 + r] == id) ? 1.0f : clip01(val[r]);
			if (p.pos_s) val[r] = tag_weighted(val[r], p.tw[c0 + r], ps, p.tpos[c0 + r], p.tw_keep, p.tw_threshold);
		}
		if (live) *reinterpret_cast<float4 *>(out + c0) = make_float4(val[0], val[1], val[2], val[3]);
	}
}

static size_t transport_lds_bytes(int nq, bool solver) {
	const size_t n = 16 * (size_t)nq;
	return VK_CANON_LDS + (solver ? n * 64 * 8 * 2 + n * 8 * 3 + n * 4 * 2 : 0) + (size_t)(VK_DEV_MAX_SENT_LEN + 32) * n * 4;
}

extern "C" hipError_t vk_launch_wrd_exact(const VkWrdParams *p, int32_t n_cand, float *scores_to_mark, hipStream_t stream) {
	const size_t smem = transport_lds_bytes(p->nq, true);
	void (*kernel)(VkWrdParams) = p->nq <= 1 ? vk_wrd_exact_kernel<1> : p->nq == 2 ? vk_wrd_exact_kernel<2> : p->nq == 3 ? vk_wrd_exact_kernel<3> : vk_wrd_exact_kernel<4>;
	if (smem > 64 * 1024) {
		const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
		if (e != hipSuccess) return e;
	}
	kernel<<<n_cand, 64, smem, stream>>>(*p);
	if (scores_to_mark) return vk_launch_mark(p->keys, n_cand, scores_to_mark, stream);   // vk_select.hip
	return hipGetLastError();
}

extern "C" hipError_t vk_launch_rows(const VkWrdParams *p, int32_t n_cand, hipStream_t stream) {
	const int R = p->rows_len > 0 ? p->rows_len : 64;
	const size_t smem = VK_CANON_LDS + (size_t)(R + 32) * 16 * (size_t)(p->nq < 1 ? 1 : p->nq) * 4;
	void (*kernel)(VkWrdParams) = p->nq <= 1 ? vk_rows_kernel<1> : p->nq == 2 ? vk_rows_kernel<2> : p->nq == 3 ? vk_rows_kernel<3> : vk_rows_kernel<4>;
	if (smem > 64 * 1024) {
		const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
		if (e != hipSuccess) return e;
	}
	kernel<<<n_cand, 64, smem, stream>>>(*p);
	return hipGetLastError();
}

// rows of winners of up to rows_len tokens, tile-parallel; rows_out must be zeroed by the caller (max_tiles: 16-token tiles the
// longest slice of the corpus can touch)
extern "C" hipError_t vk_launch_canon_rows(const VkWrdParams *p, int32_t n_cand, int32_t max_tiles, hipStream_t stream) {
	if (n_cand < 1 || max_tiles < 1 || p->rows_len < 1) return hipErrorInvalidValue;
	void (*kernel)(VkWrdParams) = p->nq <= 1 ? vk_canon_rows_kernel<1> : p->nq == 2 ? vk_canon_rows_kernel<2> : p->nq == 3 ? vk_canon_rows_kernel<3> : vk_canon_rows_kernel<4>;
	kernel<<<dim3((unsigned)max_tiles, (unsigned)n_cand), 64, VK_CANON_LDS, stream>>>(*p);
	return hipGetLastError();
}
