// vk_score.hip.h -- the fused scoring kernel and its launcher templates.  The instantiations are spread over
// vk_score_m0..m3.hip (one per similarity MODE) so that they compile in parallel.
#ifndef VK_SCORE_HIP_H
#define VK_SCORE_HIP_H

#include "vk_common.hip.h"

// ---------------------------------------------------------------------------
// the fused scoring kernel: one wave = 4 sentences at a time, grid-stride over groups.
//   MODE 0: contextual layout, NK32 K-steps (last one half filled when TAIL), query fragments in registers
//   MODE 1: contextual layout, bf16 rows of any d (runtime K loop)
//   MODE 2: static layout: gather rows of the per-query table by token id
//   MODE 3: contextual layout, NK32 K-steps, query tile staged in LDS (large d)
//   MODE 4: contextual layout, fp32 tiles of NK32 blocks of 16 features (compile time), query tile staged in LDS
//   MODE 5: MODE 1 for bf16 rows of 256 features and more: eight K-steps in flight instead of four
//   MODE 6: MODE 1 for fp32 rows (any d but the 300 of MODE 4)
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
	if constexpr (MODE == 3 || MODE == 4) {
		for (int i = threadIdx.x; i < NK32 * 64; i += blockDim.x)
			vk_smem4[i] = *reinterpret_cast<const float4 *>(p.qtile + i * 16);
		__syncthreads();
		smem += NK32 * 256;
	}
	// MODE 1 (any d, fp32 tiles): the same staging with a runtime size; the K loop then reads the query with ds_read instead of
	// going through L1 for every token tile
	if constexpr (MODE == 1 || MODE == 5 || MODE == 6) {
		if (p.q_lds > 0) {
			for (int i = threadIdx.x; i * 16 < p.q_lds; i += blockDim.x)
				vk_smem4[i] = i * 16 < p.tile_bytes ? *reinterpret_cast<const float4 *>(p.qtile + i * 16) : float4{0.0f, 0.0f, 0.0f, 0.0f};
			__syncthreads();
			smem += p.q_lds / 4;
		}
	}
	float *S = smem + wv * p.lds_floats_per_wave;
	float *Hh = S + p.s_rows_per_wave * LT + 16;   // strip rows hold the LT query columns only; 16 floats of slack for lanes >= LT
	const int sigma = lane >> 4, v = lane & 15;

	QFrag<NK32, TAIL> qf;
	if constexpr (MODE == 0) load_qfrag<NK32, TAIL>(qf, p.qtile, lane);

	DpArgs a;
	a.locality = p.locality; a.len_t = p.len_t;
	a.gs = p.gs; a.gt = p.gt; a.a_s = p.a_s; a.a_t = p.a_t; a.open_s = p.open_s; a.open_t = p.open_t;
	a.ws = p.ws; a.wt = p.wt; a.wt0 = p.wt0;
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
	// pads the slice table, vk_corpus.cpp set_slices_impl) and are left to a second launch that walks
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
				else if constexpr (MODE == 4) acc = sim_tile_f32_qlds<NK32>(qlds, tp, lane);
				else {
					// two calls, not one with a selected pointer: an LDS-or-global pointer is a flat pointer, and flat loads count
					// on both wait counters -- every batch of tile loads would be waited for in full
					constexpr bool DEEP = MODE == 5 || MODE == 6;
					constexpr int PREC = MODE == 6 ? 1 : 0;
					if (p.q_lds > 0) acc = sim_tile_generic<DEEP, PREC>(qlds, tp, p.nk32, p.tail, lane);
					else acc = sim_tile_generic<DEEP, PREC>(p.qtile, tp, p.nk32, p.tail, lane);
				}
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
		// tag-weighted vocabulary transport: cells upstream writes twice (static_vocab_fixup).  The four slices of the wave share one
		// strip of rows: over sliding windows a rewritten cell may belong to a neighbour too, and the slices take turns
		int turns = 1;
		bool rewrite = false;
		if constexpr (MODE == 2 && (GAP == 4 || GAP == 7)) {
			if (p.qid_bits) {
				rewrite = len > 0 && static_vocab_shared(len, p.len_t, p.tok_id + t_a, p.tag_s + t_a, p.qid_bits, p.qkey) != 0;
				if (p.slices_overlap && __builtin_amdgcn_ballot_w64(rewrite) != 0) turns = 4;
				else if (rewrite) static_vocab_fixup<16>(S + rowbase * LT, LT, len, p.len_t, p.tok_id + t_a, p.tag_s + t_a, p.pos_s + t_a, p.table, 0,
					p.qid_bits, p.qkey, p.tw, p.tpos, p.tw_keep, p.tw_threshold, v);
				wave_lds_fence();
			}
		}

		const int lenc = len > 0 ? len : 0;
		const int rb = len > 0 ? rowbase : 0;
		auto evaluate = [&]() -> float {
		float raw;
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
				const float wsum = p.rwmd_normalize_bow ? (float)(lenc > 0 ? lenc : 1) : 1.0f;   // bow[i] /= w_sum (bow.h:262-270): a division, as upstream -- a mass that ties with a capacity must tie here too
				for (int u = v; u < lenc; u += 16) {
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
			raw = rwmd_fill_rows<LT>(S, sm, rb, lenc, maxlen, lane, a, p.qmass[v]);
		}
		else if constexpr (MODE == 2) raw = wrd_bound_rows<LT>(S, rb, lenc, maxlen, v, a, p.mag, p.tok_id + (len > 0 ? t_a : 0), p.qmass[v]);
		else raw = wrd_bound_rows<LT>(S, rb, lenc, maxlen, v, a, p.mag + (len > 0 ? t_a : 0), nullptr, p.qmass[v]);
		return raw;
		};
		float raw = 0.0f;
		if constexpr (MODE == 2 && (GAP == 4 || GAP == 7)) {
			if (turns > 1) {
				for (int turn = 0; turn < 4; turn++) {
					const bool mine = sigma == turn;
					if (mine && rewrite) static_vocab_fixup<16>(S + rowbase * LT, LT, len, p.len_t, p.tok_id + t_a, p.tag_s + t_a, p.pos_s + t_a, p.table, 0,
						p.qid_bits, p.qkey, p.tw, p.tpos, p.tw_keep, p.tw_threshold, v);
					wave_lds_fence();
					const float r = evaluate();
					if (mine) raw = r;
					wave_lds_fence();
					if (mine && rewrite) static_vocab_fixup<16, false, true>(S + rowbase * LT, LT, len, p.len_t, p.tok_id + t_a, p.tag_s + t_a, p.pos_s + t_a, p.table, 0,
						p.qid_bits, p.qkey, p.tw, p.tpos, p.tw_keep, p.tw_threshold, v);
					wave_lds_fence();
				}
			} else raw = evaluate();
		} else raw = evaluate();

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
			if (p.raw) p.raw[s_idx] = r;
		}
		wave_lds_fence();
	}
	}
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
	const int real_occ = occ;
	// measured on MI355X (1M x 32 x 300-d): 3 workgroups (12 waves) per CU stream HBM fastest --
	// 2.90 ms vs 3.51 ms at 5 per CU for the linear-gap kernel, 2.95 ms at 4; more concurrent streams cost bandwidth
	// the transport bound (WRD / full WMD stage 1) has the longest epilogue per group and wants every wave it can get to
	// hide it: 1 M x 32 x 300-d 3.54 ms at 3 per CU, 3.08 at 4; 768-d 4.31 ms at 1, 3.37 at 2
	const bool bound_pass = p.gap_mode == 5 || (p.gap_mode == 4 && p.wmd_bound == 1);
	// narrow rows (d <= 96) are not a stream either: a slice is 6 KB or less and the DP dominates -- 64-d, 32 tokens: 3.4 TB/s at
	// 3 per CU, 4.8 TB/s at 5; 96-d: 3.7 against 5.4 TB/s.  129..224-d: 4 per CU stream 4-8 % faster than 3 (6.4 against
	// 6.0 TB/s); 128-d and 256..1536-d: 3 per CU within 2 % of the best setting (tools/sweep_dims.py, profiles/r02_sweep_dims*.jsonl)
	const bool dp_bound = p.layout == VK_DEV_LAYOUT_STATIC || p.nk32 <= 3;
	const int stream_cap = (p.nk32 >= 5 && p.nk32 <= 7) ? 4 : 3;
	if (occ > stream_cap && !dp_bound && !bound_pass) occ = stream_cap;   // the static layout is DP-bound, not a stream: keep full residency
	// 768-d rows: a wave already keeps 24 KiB of loads in flight per tile; one workgroup per CU measured fastest
	// (ragged 8..64 tokens, 400 k sentences: 3.37 ms at 1, 3.45 ms at 2 per CU)
	// -- the 768-d specialisation only: the generic kernel at 1024-d loses 20 % (32 tokens) to 38 % (ragged) with one group
	// per CU (tools/sweep_dims.py)
	if (p.nk32 == 24 && p.tail == 0 && p.prec == 0 && p.layout != VK_DEV_LAYOUT_STATIC && !bound_pass) occ = 1;
	const char *ov = getenv("VK_BLOCKS_PER_CU");   // read per launch: tools/sweep_dims.py varies it inside one process
	if (ov && atoi(ov) > 0) occ = atoi(ov) < real_occ ? atoi(ov) : real_occ;
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

#endif
