// vk_batch.cpp -- C-ABI: several queries per call (the RWMD GEMM pass, queries sharing a pass over the tiles,
// or one by one).
// No CPU compute fallback exists: without a HIP device every entry point that
// would compute returns VK_ERR_NO_DEVICE / VK_ERR_HIP.

#include "vk_internal.h"
#include "vk_guard.h"
#include "vk_transport_host.h"

#include <chrono>
#include <thread>

static int build_static_buckets(vk_corpus *c);

static bool same_gap(const vk_gap &a, const vk_gap &b, int upto) {
	if (a.kind != b.kind) return false;
	if (a.kind != VK_GAP_TABLE) return a.u == b.u && (a.kind == VK_GAP_LINEAR || a.v == b.v);
	if (a.table == b.table && a.n_table == b.n_table) return true;
	for (int k = 0; k <= upto; k++) if (gap_cost(a, k) != gap_cost(b, k)) return false;
	return true;
}

// The result sets of a batch of relaxed-WMD queries from the kk = k + margin slices selected per query: the similarity rows of
// every candidate in the canonical arithmetic (one launch for all queries, every candidate against its own query's tile), its score
// restated from them on the host as the reference computes it (vk_transport_host.h), the k best kept -- scores, slices, aligner
// scores and the rows the host states the flows from (SparseFlow, alignment/wmd.h:392-408) written to outs[i].
// keys: [n_queries x kk] as selected (0 = empty slot).
// packed16: the queries' 16-row tiles as vk_pack_query lays them out, tile_bytes apart, when the caller has packed them already
// (the GEMM path: packing 256 queries a second time, on one thread, cost 12 ms per batch); null: packed here
static int batch_winner_rows(vk_corpus *c, const vk_query_desc *qs, int n_queries, vk_topk_out *outs, const uint64_t *keys, int kk, int k, hipStream_t st,
	vk_host_keep &keep, const uint8_t *packed16 = nullptr) {
	int rc;
	// which of a query's kk candidates are restated: its k best, and of the runners-up those the k-th could lose its place to -- a
	// score of the scoring pass within rounding (2e-5, ten times what MFMA accumulation was seen to differ by) of the k-th's.
	// Typically none: 10 rows per query travel instead of 18.  first[i]: where query i's candidates start in the compact list.
	auto key_score = [](uint64_t key) {
		const uint32_t ob = (uint32_t)(key >> 32);
		const uint32_t bits = (ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob;
		float s;
		memcpy(&s, &bits, 4);
		return s;
	};
	std::vector<int32_t> first((size_t)n_queries + 1, 0);
	for (int i = 0; i < n_queries; i++) {
		int cnt = 0;
		while (cnt < kk && keys[(size_t)i * kk + cnt] != 0) cnt++;
		int need = std::min(cnt, k);
		if (cnt > k) {
			const float sk = key_score(keys[(size_t)i * kk + k - 1]);
			// (the static layout's pass ranks on 16-bit cells: its scores sit within 1e-5 of the fp32 ones, the slack is doubled)
			const float slack = (c->desc.layout == VK_LAYOUT_STATIC ? 4e-5f : 2e-5f) * std::max(1.0f, std::fabs(sk));
			while (need < cnt && key_score(keys[(size_t)i * kk + need]) >= sk - slack) need++;
		}
		first[(size_t)i + 1] = first[(size_t)i] + need;
	}
	const size_t n_cand = (size_t)first[(size_t)n_queries];
	if (n_cand == 0) {
		for (int i = 0; i < n_queries; i++) outs[i].n_out = 0;
		return VK_OK;
	}
	const bool trace = getenv("VK_TRACE_BATCH") != nullptr;
	const auto t_begin = std::chrono::steady_clock::now();
	auto stamp = [&](const char *what) {
		if (trace) fprintf(stderr, "[vk] batch_winner_rows %s: %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
	};
	if (c->bqt_cap < (size_t)n_queries) {
		if (c->d_bqt) { VK_HIP(hipFree(c->d_bqt)); c->d_bqt = nullptr; }
		if ((rc = alloc_t(c, &c->d_bqt, (size_t)n_queries * c->tile_bytes))) return rc;
		c->bqt_cap = (size_t)n_queries;
	}
	const size_t cap_cand = (size_t)n_queries * (size_t)kk;   // buffers for the most a batch of this shape can ask for: no regrowth from batch to batch
	if (c->bcand_cap < cap_cand) {
		for (void *ptr : {(void *)c->d_bcand, (void *)c->d_bcandq, (void *)c->d_brows}) if (ptr) VK_HIP(hipFree(ptr));
		c->d_bcand = nullptr; c->d_bcandq = nullptr; c->d_brows = nullptr;
		if ((rc = alloc_t(c, &c->d_bcand, cap_cand))) return rc;
		if ((rc = alloc_t(c, &c->d_bcandq, cap_cand))) return rc;
		if ((rc = alloc_t(c, &c->d_brows, cap_cand * 64 * 16))) return rc;
		c->bcand_cap = cap_cand;
	}
	std::vector<uint8_t> &qt = keep.vec<uint8_t>(), one;
	std::vector<uint64_t> &hk = keep.vec<uint64_t>(n_cand, 0);
	std::vector<int32_t> &hq = keep.vec<int32_t>(n_cand, 0);
	// static layout: the token ids of every query, 16 per query (-1: none) -- the rows kernel sets sim[id(t_j)][j] = 1 from them
	const bool is_static = c->desc.layout == VK_LAYOUT_STATIC;
	std::vector<int32_t> &hids = keep.vec<int32_t>(is_static ? (size_t)n_queries * 16 : 0, -1);
	if (is_static) {
		for (int i = 0; i < n_queries; i++)
			for (int j = 0; j < qs[i].len_t && j < 16 && qs[i].q_token_ids; j++) hids[(size_t)i * 16 + j] = qs[i].q_token_ids[j];
		if (c->bqids_cap < hids.size()) {
			if (c->d_bqids) { VK_HIP(hipFree(c->d_bqids)); c->d_bqids = nullptr; c->bqids_cap = 0; }
			if ((rc = alloc_t(c, &c->d_bqids, hids.size()))) return rc;
			c->bqids_cap = hids.size();
		}
		VK_HIP(hipMemcpyAsync(c->d_bqids, hids.data(), hids.size() * 4, hipMemcpyHostToDevice, st));
	}
	float mags[VK_MAX_QUERY_LEN];
	if (!packed16) qt.assign((size_t)n_queries * c->tile_bytes, 0);
	for (int i = 0; i < n_queries; i++) {
		if (!packed16) {
			vk_pack_query(c, &qs[i], one, mags);
			memcpy(qt.data() + (size_t)i * c->tile_bytes, one.data(), std::min(one.size(), (size_t)c->tile_bytes));
		}
		for (int j = 0; j < first[(size_t)i + 1] - first[(size_t)i]; j++) {
			hq[(size_t)first[(size_t)i] + j] = i;
			hk[(size_t)first[(size_t)i] + j] = (1ull << 32) | (uint64_t)(uint32_t)(keys[(size_t)i * kk + j] & 0xffffffffu);
		}
	}
	VK_HIP(hipMemcpyAsync(c->d_bqt, packed16 ? packed16 : qt.data(), (size_t)n_queries * c->tile_bytes, hipMemcpyHostToDevice, st));
	VK_HIP(hipMemcpyAsync(c->d_bcand, hk.data(), n_cand * 8, hipMemcpyHostToDevice, st));
	VK_HIP(hipMemcpyAsync(c->d_bcandq, hq.data(), n_cand * 4, hipMemcpyHostToDevice, st));
	stamp("uploads issued");
	VkWrdParams w{};
	w.tiles = c->d_tiles; w.sent_start = c->d_sent_start; w.sent_end = c->d_sent_end;
	w.layout = is_static ? VK_DEV_LAYOUT_STATIC : VK_DEV_LAYOUT_CONTEXTUAL;
	if (is_static) { w.tok_id = c->d_tok_id; w.q_ids = c->d_bqids; w.q_ids_stride = 16; }
	w.nk32 = c->nk32; w.tail = c->tail; w.tile_bytes = c->tile_bytes; w.prec = c->prec;
	w.qtile = c->d_bqt; w.qtile_stride = c->tile_bytes; w.cand_query = c->d_bcandq; w.nq = 1; w.len_t = qs[0].len_t;
	w.d = c->desc.d;   // canonical similarity rows (sim_canon)
	w.keys = c->d_bcand; w.rows_out = c->d_brows;
	VK_HIP(vk_launch_rows(&w, (int32_t)n_cand, st));
	// one copy into pinned host memory, then into the callers' arrays (copies into pageable memory go through the runtime's
	// staging: 256 of them, one per query, took 5 - 30 ms; a zero-initialised std::vector as the bounce buffer 3 ms)
	const size_t bytes = n_cand * 64 * 16 * 4, cap_bytes = cap_cand * 64 * 16 * 4;
	if (c->h_brows_cap < cap_bytes) {
		if (c->h_brows) { VK_HIP(hipHostFree(c->h_brows)); c->h_brows = nullptr; c->h_brows_cap = 0; }
		VK_HIP(hipHostMalloc((void **)&c->h_brows, cap_bytes, hipHostMallocDefault));
		c->h_brows_cap = cap_bytes;
	}
	stamp("kernel issued");
	VK_HIP(hipMemcpyAsync(c->h_brows, c->d_brows, bytes, hipMemcpyDeviceToHost, st));
	VK_HIP(hipStreamSynchronize(st));
	stamp("rows on the host");
	// the score of every candidate from its rows, in the reference's order of operations (vk_transport_host.h; contextual layout:
	// every position is a vocabulary entry); the k best of a query's kk candidates, in the order of the result set
	auto rank_range = [&](int i0, int i1) {
		std::vector<int> order;
		std::vector<float> val((size_t)kk), raw((size_t)kk);
		std::vector<int32_t> key_s, key_t;   // static layout: vocabulary keys (token ids) of both sides (alignment/bow.h:204-275)
		for (int i = i0; i < i1; i++) {
			const vk_query_desc &q = qs[i];
			vk_topk_out *out = &outs[i];
			order.clear();
			const size_t at = (size_t)first[(size_t)i];
			for (int j = 0; j < first[(size_t)i + 1] - first[(size_t)i]; j++) {
				const uint64_t key = keys[(size_t)i * kk + j];
				const int64_t g = (int64_t)(uint32_t)(key & 0xffffffffu);
				const int t_a = (*c->h_start)[(size_t)g], len_s = (*c->h_end)[(size_t)g] - t_a;
				const bool vocab = is_static && q.q_token_ids && c->h_tok;
				if (vocab) {
					key_t.assign(q.q_token_ids, q.q_token_ids + q.len_t);
					key_s.assign(c->h_tok->begin() + t_a, c->h_tok->begin() + t_a + len_s);
				}
				raw[(size_t)j] = vk_host::rwmd_from_rows(c->h_brows + (at + j) * 64 * 16, 16, len_s, q.len_t, vocab ? key_s.data() : nullptr, vocab ? key_t.data() : nullptr,
					q.rwmd_injective != 0, q.rwmd_symmetric != 0, q.rwmd_normalize_bow != 0);
				val[(size_t)j] = (raw[(size_t)j] / (float)q.len_t) * (q.boost ? q.boost[g] : 1.0f);
				if (val[(size_t)j] > q.min_score) order.push_back(j);
			}
			std::sort(order.begin(), order.end(), [&](int a, int b) {
				if (val[(size_t)a] != val[(size_t)b]) return val[(size_t)a] > val[(size_t)b];
				return (uint32_t)(keys[(size_t)i * kk + a] & 0xffffffffu) > (uint32_t)(keys[(size_t)i * kk + b] & 0xffffffffu);
			});
			const int n_out = std::min((int)order.size(), k);
			for (int r = 0; r < n_out; r++) {
				const int j = order[(size_t)r];
				out->score[r] = val[(size_t)j];
				out->sentence[r] = (int64_t)(uint32_t)(keys[(size_t)i * kk + j] & 0xffffffffu);
				if (out->raw_score) out->raw_score[r] = raw[(size_t)j];
				if (out->mapping && out->edge_sim)
					for (int t = 0; t < q.len_t; t++) {
						out->mapping[(size_t)r * q.len_t + t] = -1;
						out->edge_sim[(size_t)r * q.len_t + t] = 0.0f;
					}
				const size_t room = out->rows_per_winner > 0 ? (size_t)out->rows_per_winner : (size_t)VK_FAST_SENT_LEN;   // rows per winner of the caller's array (>= 64)
				memcpy(out->sim_rows + (size_t)r * room * 16, c->h_brows + (at + j) * 64 * 16, (size_t)64 * 16 * 4);
			}
			out->n_out = n_out;
		}
	};
	const int n_threads = n_queries >= 64 ? 4 : 1;
	if (n_threads == 1) rank_range(0, n_queries);
	else {
		std::vector<std::thread> pool;
		for (int t = 0; t < n_threads; t++) pool.emplace_back(rank_range, (int)((int64_t)n_queries * t / n_threads), (int)((int64_t)n_queries * (t + 1) / n_threads));
		for (auto &th : pool) th.join();
	}
	stamp("ranked and copied out");
	return VK_OK;
}

// Queries with common options over a contextual corpus: up to 4 queries share one pass over the token tiles
// (vk_score_batch_kernel).  Returns VK_ERR_UNSUPPORTED (without setting an error) when the batch does not qualify;
// the caller then runs the queries one by one.
static int query_batch_shared_pass(vk_corpus_t *c, const vk_query_desc *qs, int32_t n_queries, vk_topk_out *outs, vk_host_keep &keep) {
	if (n_queries < 2 || !c->finalized || c->prec != 0 || c->desc.layout != VK_LAYOUT_CONTEXTUAL || c->n_long_groups > 0 || c->max_len > VK_FAST_SENT_LEN || c->desc.n_sentences < 1) return VK_ERR_UNSUPPORTED;
	if (c->nk32 > 10 && !getenv("VK_BATCH_QB")) return VK_ERR_UNSUPPORTED;   // measured: no gain over single queries for 768-d rows (the kernel pipelines tiles of <= 10 K-steps)
	const vk_query_desc &q0 = qs[0];
	{
		// The wave-streaming selection of a batch holds at most 64 keys per query, and queries whose winners are restated (alignments
		// with traceback; relaxed WMD with similarity rows) select k + 8 candidates as vk_query does: beyond k = 56 such a batch is
		// answered query by query, with the full margin (round 3 shrank the margin to 64 - k instead: at k = 64 no runner-up was restated
		// and a near tie could differ from vk_query).  A batch in which only SOME result sets carry sim_rows would be restated for none:
		// per query as well.
		bool all_rows = true, any_rows = false;
		for (int i = 0; i < n_queries; i++) { all_rows = all_rows && outs[i].sim_rows != nullptr; any_rows = any_rows || outs[i].sim_rows != nullptr; }
		const bool align0 = q0.algorithm == VK_ALG_ALIGN;
		if (!align0 && q0.want_flow && any_rows && !all_rows) return VK_ERR_UNSUPPORTED;
		const bool restated = q0.want_flow && (align0 || all_rows);
		if (q0.max_matches + (restated ? 8 : 0) > 64) return VK_ERR_UNSUPPORTED;
	}
	int max_len_t = 0;
	for (int i = 0; i < n_queries; i++) {
		const vk_query_desc &q = qs[i];
		const bool alg_ok = q.algorithm == VK_ALG_ALIGN || (q.algorithm == VK_ALG_RWMD && q.rwmd_injective && !q.wmd_full);
		if (!alg_ok || q.algorithm != q0.algorithm || q.len_t > VK_FAST_QUERY_LEN || q.len_t < 1 || q.tag_weights || q.submatch_weight != 0.0f ||
			q.locality != q0.locality || q.max_matches != q0.max_matches || q.min_score != q0.min_score || q.boost != q0.boost ||
			q.want_flow != q0.want_flow || q.rwmd_symmetric != q0.rwmd_symmetric || q.rwmd_normalize_bow != q0.rwmd_normalize_bow)
			return VK_ERR_UNSUPPORTED;
		if (q.algorithm == VK_ALG_ALIGN && (!same_gap(q.gap_s, q0.gap_s, c->max_len) || !same_gap(q.gap_t, q0.gap_t, VK_FAST_QUERY_LEN)))
			return VK_ERR_UNSUPPORTED;
		max_len_t = std::max(max_len_t, (int)q.len_t);
	}
	int rc;
	for (int i = 0; i < n_queries; i++)
		if ((rc = vk_validate_query(c, &qs[i], &outs[i]))) return rc;
	VK_HIP(hipSetDevice(c->device));
	hipStream_t st = c->stream;
	const int64_t n = c->n_entries;
	const int k = q0.max_matches;
	const bool is_align = q0.algorithm == VK_ALG_ALIGN;
	// alignments with traceback: k + 8 slices are selected and restated in the canonical arithmetic, the k best of them returned
	// (vk_query: the result set is then the oracle's, not only its members' numbers); kk slots per query below
	// relaxed WMD with flows: likewise, restated on the host from the candidates' canonical similarity rows (batch_winner_rows)
	bool canon_tr = !is_align && q0.want_flow;
	for (int i = 0; i < n_queries; i++) canon_tr = canon_tr && outs[i].sim_rows != nullptr;
	const bool margin = (q0.want_flow && is_align) || canon_tr;
	const int kk = margin ? k + 8 : k;   // (<= 64: checked above)
	const float sel_floor = margin ? q0.min_score - 1e-5f * std::max(1.0f, std::fabs(q0.min_score)) : q0.min_score;

	// ---- common options: gap tables, DP form
	VkScoreBatchParams p{};
	const int ks = q0.gap_s.kind, kt = q0.gap_t.kind;
	float *ws = keep.array<float>(kGapTable), *wt = keep.array<float>(160);   // wt[80..159]: the subadditive closure of w_t
	if (!is_align) {
		p.gap_mode = 4; p.rwmd_symmetric = q0.rwmd_symmetric; p.rwmd_normalize_bow = q0.rwmd_normalize_bow;
	} else if (ks == VK_GAP_LINEAR && kt == VK_GAP_LINEAR) {
		p.gap_mode = 0; p.gs = q0.gap_s.u; p.gt = q0.gap_t.u;
	} else if ((ks == VK_GAP_LINEAR || ks == VK_GAP_AFFINE) && (kt == VK_GAP_LINEAR || kt == VK_GAP_AFFINE)) {
		p.gap_mode = 1;
		p.a_s = ks == VK_GAP_AFFINE ? q0.gap_s.u : 0.0f; p.gs = ks == VK_GAP_AFFINE ? q0.gap_s.v : q0.gap_s.u;
		p.a_t = kt == VK_GAP_AFFINE ? q0.gap_t.u : 0.0f; p.gt = kt == VK_GAP_AFFINE ? q0.gap_t.v : q0.gap_t.u;
		p.open_s = p.a_s + p.gs; p.open_t = p.a_t + p.gt;
	} else p.gap_mode = 2;
	for (int i = 0; i < kGapTable; i++) ws[i] = (is_align && i <= c->max_len) ? gap_cost(q0.gap_s, i) : 0.0f;
	for (int i = 0; i < 80; i++) wt[i] = (is_align && i <= max_len_t) ? gap_cost(q0.gap_t, i) : 0.0f;
	for (int k = 0; k < 80; k++) wt[80 + k] = wt[k];   // the subadditive closure of w_t for the in-row candidates (vk_query.cpp)
	for (int k = 2; k <= max_len_t && k < 80; k++)
		for (int a = 1; a < k; a++) wt[80 + k] = std::min(wt[80 + k], wt[80 + a] + wt[80 + k - a]);
	if (p.gap_mode == 2) p.gap_mode = c->max_short_len <= 32 ? 3 : 6;
	const int lt = max_len_t <= 4 ? 4 : max_len_t <= 8 ? 8 : max_len_t <= 12 ? 12 : 16;
	p.s_rows_per_wave = c->max_group_tiles * 16;
	p.h_rows = c->max_short_len + 1;
	const int strip = p.s_rows_per_wave * lt + 16;
	const int hist = p.gap_mode == 2 ? 4 * p.h_rows * 16 : 0;
	auto smem_of = [&](int qb) { return (size_t)qb * c->tile_bytes + ((size_t)qb * strip + hist) * 4 * 4; };   // query tiles + 4 waves x (qb strips + history)
	// measured (12 queries x 1 M x 32 x 300-d): 2 queries per pass with two workgroups per CU and 4 per pass with one
	// take the same time (22 ms linear, 30 ms WSB); 2 leaves LDS for ragged corpora
	int qb_max = 2;
	if (const char *e = getenv("VK_BATCH_QB")) qb_max = std::max(2, std::min(4, atoi(e)));   // tuning aid
	while (qb_max > 1 && smem_of(qb_max) > 160 * 1024) qb_max--;
	if (qb_max < 2) return VK_ERR_UNSUPPORTED;
	p.n_strips = qb_max;
	p.lds_floats_per_wave = qb_max * strip + hist;
	auto smem_for = [&](int qb) { return (size_t)qb * c->tile_bytes + (size_t)p.lds_floats_per_wave * 4 * 4; };

	// ---- buffers
	const size_t need_q = (size_t)4 * c->tile_bytes;
	if (c->bq_cap < need_q) {
		if (c->d_bq) VK_HIP(hipFree(c->d_bq));
		if ((rc = alloc_t(c, &c->d_bq, need_q))) return rc;
		c->bq_cap = need_q;
	}
	if (c->bqlen_cap < 16) {
		if (c->d_bqlen) VK_HIP(hipFree(c->d_bqlen));
		if ((rc = alloc_t(c, &c->d_bqlen, 16))) return rc;
		c->bqlen_cap = 16;
	}
	const size_t need_s = (size_t)4 * (size_t)n;
	if (c->bscores_cap < need_s) {
		if (c->d_bscores) VK_HIP(hipFree(c->d_bscores));
		if ((rc = alloc_t(c, &c->d_bscores, need_s))) return rc;
		c->bscores_cap = need_s;
	}
	if (c->braw_cap < need_s) {
		if (c->d_braw) VK_HIP(hipFree(c->d_braw));
		if ((rc = alloc_t(c, &c->d_braw, need_s))) return rc;
		c->braw_cap = need_s;
	}
	const int64_t nw1 = (n + 4095) / 4096;
	const size_t need_k = (size_t)4 * (size_t)nw1 * (size_t)kk;
	if (c->bkeys_cap < need_k) {
		for (auto &b : c->d_bkeys) if (b) VK_HIP(hipFree(b));
		if ((rc = alloc_t(c, &c->d_bkeys[0], need_k))) return rc;
		if ((rc = alloc_t(c, &c->d_bkeys[1], need_k))) return rc;
		c->bkeys_cap = need_k;
	}
	VK_HIP(hipMemcpyAsync(c->d_ws, ws, kGapTable * sizeof(float), hipMemcpyHostToDevice, st));
	VK_HIP(hipMemcpyAsync(c->d_wt, wt, 160 * sizeof(float), hipMemcpyHostToDevice, st));
	if (q0.boost) {
		if (!c->d_boost) { rc = alloc_t(c, &c->d_boost, (size_t)n + 8); if (rc) return rc; }
		VK_HIP(hipMemcpyAsync(c->d_boost, q0.boost, (size_t)n * 4, hipMemcpyHostToDevice, st));   // no long slices: rows == slices
	}
	p.tiles = c->d_tiles; p.sent_start = c->d_sent_start; p.sent_end = c->d_sent_end; p.n_sent = (int32_t)n;
	p.nk32 = c->nk32; p.tail = c->tail; p.tile_bytes = c->tile_bytes;
	p.qtiles = c->d_bq; p.locality = q0.locality; p.ws = c->d_ws; p.wt = c->d_wt + 80; p.wt0 = c->d_wt;
	p.boost = q0.boost ? c->d_boost : nullptr; p.scores = c->d_bscores; p.raw = c->d_braw;
	if (q0.want_flow && q0.algorithm == VK_ALG_ALIGN) p.raw = nullptr;   // the flow kernel restates the winners' aligner scores; nothing else reads the array (no submatch weights here)

	float score_ms_total = 0.0f, total_ms = 0.0f;
	std::vector<uint8_t> &all = keep.vec<uint8_t>(need_q), one;
	float mags[VK_MAX_QUERY_LEN];
	// (host ends of the copies below: one set for all passes -- every pass ends synchronised)
	std::vector<uint64_t> &keys = keep.vec<uint64_t>((size_t)qb_max * kk);
	std::vector<float> &raw = keep.vec<float>((size_t)qb_max * kk), &sim = keep.vec<float>((size_t)qb_max * kk * 16);
	std::vector<int16_t> &map = keep.vec<int16_t>((size_t)qb_max * kk * 16);
	for (int base = 0; base < n_queries; base += qb_max) {
		const int qb = std::min(qb_max, n_queries - base);
		// Query::abort (query.h:183-189): polled between the passes; the queries before `base` are complete
		if (qs[base].abort && *qs[base].abort) {
			for (int i = base; i < n_queries; i++) outs[i].n_out = 0;
			return fail(VK_ERR_ABORTED, "batch aborted by the caller");
		}
		VK_HIP(hipEventRecord(c->ev[0], st));
		std::fill(all.begin(), all.end(), 0);
		for (int i = 0; i < qb; i++) {
			vk_pack_query(c, &qs[base + i], one, mags);
			memcpy(all.data() + (size_t)i * c->tile_bytes, one.data(), (size_t)c->tile_bytes);
			p.len_t[i] = qs[base + i].len_t;
		}
		for (int i = qb; i < 4; i++) p.len_t[i] = 1;
		p.n_queries = qb;
		VK_HIP(hipMemcpyAsync(c->d_bq, all.data(), (size_t)qb * c->tile_bytes, hipMemcpyHostToDevice, st));
		VK_HIP(hipEventRecord(c->ev[1], st));
		VK_HIP(vk_launch_score_batch(&p, lt, smem_for(qb), st));
		VK_HIP(hipEventRecord(c->ev[2], st));
		int64_t nw = 0;
		int cur = 0;
		VK_HIP(vk_launch_topk_wave_batch(c->d_bscores, nullptr, n, sel_floor, kk, 4096, qb, n, nw1 * kk, c->d_bkeys[0], &nw, st));
		const int64_t stride = nw1 * kk;
		while (nw > 1) {
			const int64_t nkeys = nw * kk;
			const int64_t per_wave = nkeys <= 16384 ? nkeys : 4096;
			VK_HIP(vk_launch_topk_wave_batch(nullptr, c->d_bkeys[cur], nkeys, 0.0f, kk, per_wave, qb, stride, stride, c->d_bkeys[1 - cur], &nw, st));
			cur = 1 - cur;
		}
		VK_HIP(hipEventRecord(c->ev[3], st));
		const bool do_flow = q0.want_flow && is_align;
		if (do_flow) {
			for (int i = 0; i < qb; i++) {
				VkFlowParams f{};
				f.tiles = c->d_tiles; f.sent_start = c->d_sent_start; f.sent_end = c->d_sent_end;
				f.layout = VK_DEV_LAYOUT_CONTEXTUAL; f.nk32 = c->nk32; f.tail = c->tail; f.tile_bytes = c->tile_bytes;
				f.qtile = c->d_bq + (size_t)i * c->tile_bytes; f.len_t = qs[base + i].len_t; f.locality = q0.locality;
				f.gap_mode = (p.gap_mode == 3 || p.gap_mode == 6) ? 2 : p.gap_mode; f.max_len = c->max_len;
				f.gs = p.gs; f.gt = p.gt; f.a_s = p.a_s; f.a_t = p.a_t; f.open_s = p.open_s; f.open_t = p.open_t;
				f.ws = c->d_ws; f.wt = c->d_wt;
				f.d = c->desc.d;   // canonical similarity rows (sim_canon)
				f.keys = c->d_bkeys[cur] + (size_t)i * stride;
				f.raw_out = c->d_out_raw + (size_t)i * kk; f.mapping = c->d_out_map + (size_t)i * kk * 16; f.edge_sim = c->d_out_sim + (size_t)i * kk * 16;
				VK_HIP(vk_launch_flow(&f, kk, st));
			}
		}
		VK_HIP(hipEventRecord(c->ev[4], st));
		VK_HIP(hipMemcpy2DAsync(keys.data(), (size_t)kk * 8, c->d_bkeys[cur], (size_t)stride * 8, (size_t)kk * 8, (size_t)qb, hipMemcpyDeviceToHost, st));
		if (do_flow) {
			VK_HIP(hipMemcpyAsync(raw.data(), c->d_out_raw, (size_t)qb * kk * 4, hipMemcpyDeviceToHost, st));
			VK_HIP(hipMemcpyAsync(map.data(), c->d_out_map, (size_t)qb * kk * 16 * 2, hipMemcpyDeviceToHost, st));
			VK_HIP(hipMemcpyAsync(sim.data(), c->d_out_sim, (size_t)qb * kk * 16 * 4, hipMemcpyDeviceToHost, st));
		}
		VK_HIP(hipStreamSynchronize(st));
		for (int i = 0; i < qb && !canon_tr; i++) {   // (relaxed WMD with flows: batch_winner_rows below writes the result sets)
			const vk_query_desc &q = qs[base + i];
			vk_topk_out *out = &outs[base + i];
			// winners of this query: position in the selection, score.  With traceback the score is restated from the canonical
			// aligner score of the flow kernel (the oracle's, bit for bit) and the winners are put in that order (vk_query).
			std::vector<int> order;
			std::vector<float> val((size_t)kk, 0.0f);
			for (int j = 0; j < kk; j++) {
				const uint64_t key = keys[(size_t)i * kk + j];
				if (key == 0) break;
				const uint32_t ob = (uint32_t)(key >> 32);
				const uint32_t bits = (ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob;
				memcpy(&val[(size_t)j], &bits, 4);
				if (do_flow) {
					float matched = 0.0f;
					for (int t = 0; t < q.len_t; t++) matched += map[((size_t)i * kk + j) * 16 + t] >= 0 ? 1.0f : 0.0f;
					const float total = (float)q.len_t, uw = powf((total - matched) / total, 0.0f);
					const float ref = matched + uw * (total - matched);
					const float boost = q.boost ? q.boost[(int64_t)(uint32_t)(key & 0xffffffffu)] : 1.0f;
					val[(size_t)j] = (raw[(size_t)i * kk + j] / ref) * boost;
					if (!(val[(size_t)j] > q.min_score)) continue;
				}
				order.push_back(j);
			}
			if (do_flow) std::sort(order.begin(), order.end(), [&](int a, int b2) {
				if (val[(size_t)a] != val[(size_t)b2]) return val[(size_t)a] > val[(size_t)b2];
				return (uint32_t)(keys[(size_t)i * kk + a] & 0xffffffffu) > (uint32_t)(keys[(size_t)i * kk + b2] & 0xffffffffu);
			});
			int n_out = 0;
			for (const int j : order) {
				if (n_out >= k) break;
				const uint64_t key = keys[(size_t)i * kk + j];
				const int64_t g = (int64_t)(uint32_t)(key & 0xffffffffu);
				out->score[n_out] = val[(size_t)j];
				out->sentence[n_out] = g;
				if (out->raw_score) {
					if (do_flow) out->raw_score[n_out] = raw[(size_t)i * kk + j];
					else VK_HIP(hipMemcpy(&out->raw_score[n_out], c->d_braw + (size_t)i * n + g, 4, hipMemcpyDeviceToHost));
				}
				if (q.want_flow && out->mapping && out->edge_sim)
					for (int t = 0; t < q.len_t; t++) {
						out->mapping[(size_t)n_out * q.len_t + t] = do_flow ? map[((size_t)i * kk + j) * 16 + t] : (int16_t)-1;
						out->edge_sim[(size_t)n_out * q.len_t + t] = do_flow ? sim[((size_t)i * kk + j) * 16 + t] : 0.0f;
					}
				n_out++;
			}
			out->n_out = n_out;
		}
		if (canon_tr && (rc = batch_winner_rows(c, qs + base, qb, outs + base, keys.data(), kk, k, st, keep))) return rc;
		float ms = 0;
		if (hipEventElapsedTime(&ms, c->ev[1], c->ev[2]) == hipSuccess) score_ms_total += ms;
		if (hipEventElapsedTime(&ms, c->ev[0], c->ev[4]) == hipSuccess) total_ms += ms;
	}
	c->have_scores = false;
	vk_timings t{};
	t.score_ms = score_ms_total; t.total_ms = total_ms;
	c->last = t;
	return VK_OK;
}

// The padded copy of a ragged corpus the batched GEMM runs on: sentences sorted into buckets by their length rounded up to
// 16, 32, 48 or 64 tokens, every sentence padded with zero rows to its bucket's length (tile aligned: a wave's token tiles
// then belong to whole sentences, as in a corpus of one sentence length), with the real length and the original index of
// each.  Built once per handle, on the device from the resident tiles.
// gran: tiles per bucket step -- 2 (buckets of 32 and 64 padded tokens: the 32x32x16 kernels, 300-d and 128-d rows) or 1 (16, 32,
// 48, 64: the 16-row kernel, 768-d rows)
static int build_batch_layout(vk_corpus *c, int gran) {
	if (c->bl_built) return VK_OK;
	const int64_t n = c->desc.n_sentences;
	std::vector<int32_t> start((size_t)n), end((size_t)n);
	VK_HIP(hipMemcpy(start.data(), c->d_sent_start, (size_t)n * 4, hipMemcpyDeviceToHost));
	VK_HIP(hipMemcpy(end.data(), c->d_sent_end, (size_t)n * 4, hipMemcpyDeviceToHost));
	std::vector<int32_t> ids[4], lens[4];
	c->bl_empty = 0;
	for (int64_t s = 0; s < n; s++) {
		const int len = end[(size_t)s] - start[(size_t)s];
		if (len < 1) { c->bl_empty++; continue; }
		const int b = (len + 16 * gran - 1) / (16 * gran) - 1;
		ids[b].push_back((int32_t)s);
		lens[b].push_back(len);
	}
	for (int b = 0; b < 4; b++) {
		auto &B = c->bl[b];
		B.n = (int64_t)ids[b].size();
		if (B.n == 0) continue;
		const int tps = (b + 1) * gran;
		int rc;
		const size_t bytes = ((size_t)B.n * tps + 1) * (size_t)c->tile_bytes;   // one zero tile follows (the kernel reads it for the last, partly filled chunk)
		if ((rc = alloc_t(c, &B.tiles, bytes))) return rc;
		if ((rc = alloc_t(c, &B.len, (size_t)B.n))) return rc;
		if ((rc = alloc_t(c, &B.id, (size_t)B.n))) return rc;
		VK_HIP(hipMemsetAsync(B.tiles + (size_t)B.n * tps * c->tile_bytes, 0, (size_t)c->tile_bytes, c->stream));
		VK_HIP(hipMemcpy(B.len, lens[b].data(), (size_t)B.n * 4, hipMemcpyHostToDevice));   // (blocking: the host vectors are this function's)
		VK_HIP(hipMemcpy(B.id, ids[b].data(), (size_t)B.n * 4, hipMemcpyHostToDevice));
		VK_HIP(vk_launch_batch_pack(c->d_tiles, B.tiles, B.id, c->d_sent_start, c->d_sent_end, B.n, tps, c->tile_bytes, c->stream));
		VK_HIP(hipStreamSynchronize(c->stream));   // the host vectors go out of scope
	}
	c->bl_built = true;
	return VK_OK;
}

// Every host buffer that is the source or the destination of an asynchronous copy lives in `keep` (owned by vk_query_batch below,
// which drains the stream before the buffers die when this body fails or is aborted: vk_guard.h).
static int query_batch_body(vk_corpus_t *c, const vk_query_desc *qs, int32_t n_queries, vk_topk_out *outs, vk_host_keep &keep) {
	// the GEMM path: injective RWMD, contextual layout, one sentence length (multiple of 16), common options
	// (uniform corpora of 16 / 32 / 48 / 64-token sentences run on the resident tiles; any other corpus of slices of at most 64
	// tokens on a padded copy, bucket by bucket)
	const bool uniform16 = c->contiguous && c->uniform_len > 0 && c->uniform_len % 16 == 0 && c->uniform_len <= 64;
	// (the kernels address a score row by a 32-bit sentence offset: 9 x n_sentences must stay below 2^31)
	// The static layout (round 4): the same epilogues over a per-batch similarity table gathered by token id instead of MFMA tiles
	// (vk_rwmd_static32_kernel): slices of at most 64 tokens, any row width; the table -- 64 bytes per (word, query tile) -- within 4 GiB
	const bool stat = c->desc.layout == VK_LAYOUT_STATIC;
	bool gemm = c->finalized && c->prec == 0 && c->desc.n_sentences > 0 && c->desc.n_sentences < (1ll << 27) && qs[0].max_matches <= 64 &&
		(stat ? (n_queries >= 2 && c->max_len <= VK_FAST_SENT_LEN && c->entry_sent.empty() && !getenv("VK_BATCH_NO_STATIC") &&
				(size_t)((c->n_tiles + 1) / 2) * 32 * (size_t)((n_queries + 1) / 2 + 5) * 64 <= ((size_t)4 << 30))
			: (vk_rwmd_batch_supported(c->nk32, c->tail) && (uniform16 || (c->max_len <= VK_FAST_SENT_LEN && c->entry_sent.empty() && !getenv("VK_BATCH_NO_RAGGED")))));
	{   // as in the shared pass: restated winners need k + 8 <= 64 keys per query, and every result set or none carries sim_rows
		bool all_rows = true, any_rows = false;
		for (int i = 0; i < n_queries; i++) { all_rows = all_rows && outs[i].sim_rows != nullptr; any_rows = any_rows || outs[i].sim_rows != nullptr; }
		if (qs[0].want_flow && any_rows && !all_rows) gemm = false;
		if (qs[0].want_flow && all_rows && qs[0].max_matches + 8 > 64) gemm = false;
	}
	for (int i = 0; i < n_queries && gemm; i++) {
		const vk_query_desc &q = qs[i];
		gemm = q.len_t <= VK_FAST_QUERY_LEN && q.algorithm == VK_ALG_RWMD && q.rwmd_injective && !q.wmd_full && q.rwmd_symmetric == qs[0].rwmd_symmetric &&
			q.rwmd_normalize_bow == qs[0].rwmd_normalize_bow && q.max_matches == qs[0].max_matches &&
			q.min_score == qs[0].min_score && q.boost == qs[0].boost && !q.tag_weights;
	}
	if (!gemm) {
		const int rcb = query_batch_shared_pass(c, qs, n_queries, outs, keep);
		if (rcb != VK_ERR_UNSUPPORTED) return rcb;
		for (int i = 0; i < n_queries; i++) {
			const int rc = vk_query(c, &qs[i], &outs[i]);   // polls the query's abort flag
			if (rc) {
				for (int r = i; r < n_queries; r++) outs[r].n_out = 0;
				return rc;
			}
		}
		return VK_OK;
	}
	for (int i = 0; i < n_queries; i++) {
		const int rc = vk_validate_query(c, &qs[i], &outs[i]);
		if (rc) return rc;
	}
	VK_HIP(hipSetDevice(c->device));
	hipStream_t st = c->stream;
	const int64_t n = c->desc.n_sentences;
	const int k = qs[0].max_matches;
	int rc;

	// 32-token sentences take the 32x32x16 kernel: 3 queries of <= 10 tokens (else 2 of <= 16) share one 32-row A tile
	// (768-d rows, round 3: the same kernels with four waves per workgroup, one per SIMD -- a wave's 64 token columns take 384
	// registers; the 16-row kernel used 10 of its 16 A rows and ran at 0.64 G pairs/s.  VK_BATCH_WIDE16=1 keeps the old path for A/B runs)
	const bool wide32 = c->nk32 == 24 && c->tail == 0 && !getenv("VK_BATCH_WIDE16");
	bool b32 = uniform16 && (c->uniform_len == 32 || c->uniform_len == 64) && (c->nk32 <= 10 || wide32);   // 64 tokens: one sentence per wave (W64)
	const int gran = (c->nk32 <= 10 || wide32) ? 2 : 1;   // ragged corpora: bucket step in tiles
	if (!stat && !uniform16 && (rc = build_batch_layout(c, gran))) return rc;
	const bool r32 = !stat && !uniform16 && gran == 2;   // ragged, on the 32x32x16 kernels
	b32 = b32 || r32 || stat;                            // query tiles packed for them (the static layout's table is built from the same tiles)
	const bool stat_uniform32 = stat && c->contiguous && c->uniform_len == 32;
	if (stat && !stat_uniform32 && (rc = build_static_buckets(c))) return rc;
	int qpt = 3;
	for (int i = 0; i < n_queries; i++) if (qs[i].len_t > 10) qpt = 2;
	const int nk16 = c->d_pad / 16;
	// batches of ten-token queries: 16 queries fill five A tiles exactly (vk_rwmd_batch32d_kernel); small batches keep 3 per tile
	const bool dense = b32 && qpt == 3 && n_queries >= 32 && !getenv("VK_BATCH32_NO_DENSE");
	const int n_super = (n_queries + 15) / 16;
	const int n_qtiles = dense ? 5 * n_super : (n_queries + qpt - 1) / qpt;
	const int score_rows = !b32 ? n_queries : dense ? 16 * n_super : n_qtiles * qpt;   // the 32-token kernels write whole tiles of queries

	// ---- device buffers (kept for the next batch)
	// (+1: the kernel prefetches one tile past the last)
	const size_t need_q = std::max((size_t)n_queries * c->tile_bytes, b32 ? (size_t)(n_qtiles + 1) * nk16 * 1024 : (size_t)0);
	if (c->bq_cap < need_q) {
		if (c->d_bq) VK_HIP(hipFree(c->d_bq));
		if ((rc = alloc_t(c, &c->d_bq, need_q))) return rc;
		c->bq_cap = need_q;
	}
	const size_t need_len = 2 * ((size_t)n_queries + 4) + std::max(8 * ((size_t)n_qtiles + 2), 32 * ((size_t)n_super + 1));   // lengths, their reciprocals, kernel parameters
	if (c->bqlen_cap < need_len) {
		if (c->d_bqlen) VK_HIP(hipFree(c->d_bqlen));
		if ((rc = alloc_t(c, &c->d_bqlen, need_len))) return rc;
		c->bqlen_cap = need_len;
	}
	const size_t need_s = (size_t)score_rows * (size_t)n;
	if (c->bscores_cap < need_s) {
		if (c->d_bscores) VK_HIP(hipFree(c->d_bscores));
		if ((rc = alloc_t(c, &c->d_bscores, need_s))) return rc;
		c->bscores_cap = need_s;
	}
	// with flows: k + 8 candidates per query, restated on the host from their canonical rows; the k best are kept (batch_winner_rows)
	bool canon_tr = qs[0].want_flow != 0;
	for (int i = 0; i < n_queries; i++) canon_tr = canon_tr && outs[i].sim_rows != nullptr;
	const int kk = canon_tr ? k + 8 : k;   // (<= 64: checked above)
	const float sel_floor = canon_tr ? qs[0].min_score - 1e-5f * std::max(1.0f, std::fabs(qs[0].min_score)) : qs[0].min_score;
	const int64_t nw1 = (n + 4095) / 4096;
	const size_t need_k = (size_t)n_queries * (size_t)nw1 * (size_t)kk;
	if (c->bkeys_cap < need_k) {
		for (auto &b : c->d_bkeys) if (b) VK_HIP(hipFree(b));
		if ((rc = alloc_t(c, &c->d_bkeys[0], need_k))) return rc;
		if ((rc = alloc_t(c, &c->d_bkeys[1], need_k))) return rc;
		c->bkeys_cap = need_k;
	}

	VK_HIP(hipEventRecord(c->ev[0], st));
	std::vector<uint8_t> &all = keep.vec<uint8_t>((size_t)need_q, 0);
	std::vector<int32_t> &qlen = keep.vec<int32_t>((size_t)n_queries);
	// the queries' 16-row tiles are kept when the winners' similarity rows will be asked for (batch_winner_rows)
	bool rows_wanted = false;
	for (int i = 0; i < n_queries; i++) rows_wanted |= qs[i].want_flow && outs[i].sim_rows;
	std::vector<uint8_t> &tiles16 = keep.vec<uint8_t>(rows_wanted ? (size_t)n_queries * c->tile_bytes : 0, 0);
	// where token j of query i sits in the 32-row A tiles: tile, lane half hd, accumulator register acc of the MFMA result
	auto slot_of = [&](int i, int j, int &tile, int &hd, int &acc) {
		const int slot = i % qpt;
		tile = i / qpt;
		if (dense) {
			// super tile i / 16; lane half hd serves its queries 8 hd .. 8 hd + 7: five whole ones (registers 0..9 of tile k) and
			// three in the registers 10..15 of the five tiles, in slot order (vk_rwmd_batch32d_kernel)
			const int idx = i % 16, r = idx % 8;
			int k;
			hd = idx / 8;
			if (r < 5) { k = r; acc = j; }
			else {
				const int lin = (r - 5) * 10 + j;   // 0..29 over the 30 spare slots of the half
				k = lin / 6; acc = 10 + lin % 6;
			}
			tile = (i / 16) * 5 + k;
		}
		else if (qpt == 2) { hd = slot; acc = j; }
		else if (slot < 2) { hd = slot; acc = j; }
		else { hd = j / 5; acc = 10 + j % 5; }
	};
	// normalise, round and lay out the queries: a few host threads over disjoint query ranges (256 queries: 1.6 ms on one)
	auto pack_range = [&](int i0, int i1) {
	std::vector<uint8_t> one;
	float mags[VK_MAX_QUERY_LEN];
	for (int i = i0; i < i1; i++) {
		vk_pack_query(c, &qs[i], one, mags);
		qlen[(size_t)i] = qs[i].len_t;
		if (!tiles16.empty()) memcpy(tiles16.data() + (size_t)i * c->tile_bytes, one.data(), std::min(one.size(), (size_t)c->tile_bytes));   // for the winners' rows
		if (!b32) {
			memcpy(all.data() + (size_t)i * c->tile_bytes, one.data(), one.size());
			continue;
		}
		// A tile of v_mfma_f32_32x32x16_bf16: K-step t = 1 KiB, lane l = 32 (k >> 3 & 1) + M owns row M, 8 features.
		// Row M of the result lands in accumulator register acc = 4 (M >> 3) + (M & 3) of lane half hd = M >> 2 & 1;
		// the kernel (vk_rwmd_batch32_kernel) expects query tokens at (hd, acc) as laid out below.
		for (int j = 0; j < qs[i].len_t; j++) {
			int hd, acc, tile_ij;
			slot_of(i, j, tile_ij, hd, acc);
			uint8_t *dst = all.data() + (size_t)tile_ij * nk16 * 1024;
			const int M = 8 * (acc >> 2) + 4 * hd + (acc & 3);
			for (int k = 0; k < c->d_pad; k += 8) {   // 8 features = one 16-byte piece in both layouts
				const size_t src = (size_t)(k >> 5) * 1024 + (size_t)(((k & 31) >> 3) * 16 + j) * 16;
				const size_t off = (size_t)(k >> 4) * 1024 + (size_t)(((k >> 3) & 1) * 32 + M) * 16;
				memcpy(dst + off, one.data() + src, 16);
			}
		}
	}
	};
	{
		const int n_thr = n_queries >= 32 ? std::min<int>(8, std::max(1u, std::thread::hardware_concurrency())) : 1;
		std::vector<std::thread> pool;
		const int per = (n_queries + n_thr - 1) / n_thr;
		for (int t = 1; t < n_thr; t++)
			if (t * per < n_queries) pool.emplace_back(pack_range, t * per, std::min(n_queries, (t + 1) * per));
		pack_range(0, std::min(n_queries, per));
		for (auto &th : pool) th.join();
	}
	VK_HIP(hipMemcpyAsync(c->d_bq, all.data(), all.size(), hipMemcpyHostToDevice, st));
	VK_HIP(hipMemcpyAsync(c->d_bqlen, qlen.data(), qlen.size() * 4, hipMemcpyHostToDevice, st));
	std::vector<float> &qinv = keep.vec<float>((size_t)n_queries);
	for (int i = 0; i < n_queries; i++) qinv[(size_t)i] = 1.0f / (float)qs[i].len_t;
	float *d_qinv = reinterpret_cast<float *>(c->d_bqlen + n_queries + 4);
	VK_HIP(hipMemcpyAsync(d_qinv, qinv.data(), qinv.size() * 4, hipMemcpyHostToDevice, st));
	// vk_rwmd_batch32_kernel: per A tile the lengths of its queries (as floats) and their reciprocals, 0 for an absent query
	float *d_qparam = d_qinv + n_queries + 4;
	std::vector<float> &qparam = keep.vec<float>((size_t)(n_qtiles + 1) * 8, 0.0f);
	if (dense) qparam.assign((size_t)(n_super + 1) * 32, 0.0f);
	for (int i = 0; i < n_queries; i++) {
		if (dense) {   // [query][2]
			qparam[(size_t)i * 2] = (float)qs[i].len_t;
			qparam[(size_t)i * 2 + 1] = qinv[(size_t)i];
			continue;
		}
		qparam[(size_t)(i / qpt) * 8 + (size_t)(i % qpt)] = (float)qs[i].len_t;
		qparam[(size_t)(i / qpt) * 8 + 4 + (size_t)(i % qpt)] = qinv[(size_t)i];
	}
	if (b32) VK_HIP(hipMemcpyAsync(d_qparam, qparam.data(), qparam.size() * 4, hipMemcpyHostToDevice, st));
	if (qs[0].boost) {
		if (!c->d_boost) { rc = alloc_t(c, &c->d_boost, (size_t)n + 8); if (rc) return rc; }
		VK_HIP(hipMemcpyAsync(c->d_boost, qs[0].boost, (size_t)n * 4, hipMemcpyHostToDevice, st));
	}

	// handles on one corpus take turns, as in vk_query: this batch's GEMM starts when the peer's has finished, so that the
	// selection, the copies and the host part of one batch run beside the GEMM of the next (the wait is on the device)
	VK_HIP(hipEventRecord(c->ev[5], st));
	if ((rc = vk_wait_peer_turn(c, st))) return rc;
	VK_HIP(hipEventRecord(c->ev[1], st));
	VkRwmdBatchParams p{};
	p.tiles = c->d_tiles; p.n_tiles = (c->desc.n_tokens + 15) / 16;
	p.tile_bytes = c->tile_bytes; p.nk = c->nk32; p.half = c->tail;
	p.qtiles = c->d_bq; p.q_len = c->d_bqlen; p.n_queries = n_queries; p.n_sent = (int32_t)n;
	p.tiles_per_sent = c->uniform_len / 16;
	p.symmetric = qs[0].rwmd_symmetric; p.nbow = qs[0].rwmd_normalize_bow;
	p.boost = qs[0].boost ? c->d_boost : nullptr;
	p.scores = c->d_bscores;
	p.n_qtiles = n_qtiles; p.qpt = qpt; p.q_inv_len = d_qinv; p.q_param = d_qparam; p.dense = dense ? 1 : 0;
	p.late_mask = wide32 ? 0 : 4;   // waves w and w + 4 of a workgroup share a SIMD (768-d rows: one wave per SIMD, nobody to alternate with)
	if (const char *e = getenv("VK_BATCH32_LATE_MASK")) p.late_mask = atoi(e);   // tuning aid
	if (stat) {
		// ---- the static layout: table of the batch over the vocabulary, its diagonal cells, then the gather pass per length bucket
		const int64_t v_rows = (int64_t)((c->n_tiles + 1) / 2) * 32;
		const int64_t table_row = (int64_t)n_qtiles * 32;
		const size_t need_t = (size_t)v_rows * (size_t)table_row;
		if (c->btable_cap < need_t) {
			if (c->d_btable) { VK_HIP(hipFree(c->d_btable)); c->d_btable = nullptr; c->btable_cap = 0; }
			if ((rc = alloc_t(c, &c->d_btable, need_t))) return rc;
			c->btable_cap = need_t;
		}
		// sim[id(t_j)][j] = 1 for every query token the vocabulary holds (metric/static.cpp:58-67), after the clip as there
		std::vector<int64_t> &fix = keep.vec<int64_t>();
		for (int i = 0; i < n_queries; i++)
			for (int j = 0; j < qs[i].len_t && qs[i].q_token_ids; j++) {
				const int32_t id = qs[i].q_token_ids[j];
				if (id < 0 || id >= c->desc.vocab_size) continue;
				int tile, hd, acc;
				slot_of(i, j, tile, hd, acc);
				fix.push_back((int64_t)id * table_row + (int64_t)tile * 32 + hd * 16 + acc);
			}
		if (c->bfix_cap < fix.size()) {
			if (c->d_bfix) { VK_HIP(hipFree(c->d_bfix)); c->d_bfix = nullptr; c->bfix_cap = 0; }
			if ((rc = alloc_t(c, &c->d_bfix, fix.size() + 16))) return rc;
			c->bfix_cap = fix.size() + 16;
		}
		if (!fix.empty()) VK_HIP(hipMemcpyAsync(c->d_bfix, fix.data(), fix.size() * 8, hipMemcpyHostToDevice, st));
		VK_HIP(vk_launch_table_batch(c->d_tiles, c->n_tiles, c->tile_bytes, nk16, c->d_bq, n_qtiles, c->d_btable, st));
		VK_HIP(vk_launch_table_batch_fix(c->d_btable, c->d_bfix, (int32_t)fix.size(), st));
		p.tok_id = c->d_tok_id; p.sent_start = c->d_sent_start; p.sent_end = c->d_sent_end;
		p.table = c->d_btable; p.table_row = table_row; p.zero_row = (int32_t)((c->n_tiles - 1) * 16);   // the zero tile behind the vocabulary
		if (stat_uniform32) VK_HIP(vk_launch_rwmd_static32(&p, 0, st));
		else {
			if (c->sb_empty > 0) VK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->d_bscores), (int)0xff800000u, (size_t)score_rows * (size_t)n, st));   // -inf: empty slices
			for (int b = 0; b < 2; b++) {
				if (c->sb_n[b] == 0) continue;
				VkRwmdBatchParams pb = p;
				pb.sent_id = c->d_sb_id[b]; pb.n_sent = (int32_t)c->sb_n[b]; pb.score_stride = n;
				VK_HIP(vk_launch_rwmd_static32(&pb, b, st));
			}
		}
	}
	else if (b32 && !r32) VK_HIP(vk_launch_rwmd_batch32(&p, st));
	else if (uniform16) VK_HIP(vk_launch_rwmd_batch(&p, st));
	else {
		if (c->bl_empty > 0) VK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->d_bscores), (int)0xff800000u, (size_t)n_queries * (size_t)n, st));   // -inf: empty slices
		for (int b = 0; b < 4; b++) {
			const auto &B = c->bl[b];
			if (B.n == 0) continue;
			VkRwmdBatchParams pb = p;
			pb.tiles = B.tiles; pb.n_tiles = B.n * (b + 1) * gran; pb.n_sent = (int32_t)B.n; pb.tiles_per_sent = (b + 1) * gran;
			pb.sent_len = B.len; pb.sent_id = B.id; pb.score_stride = n;
			if (r32) VK_HIP(vk_launch_rwmd_batch32(&pb, st));
			else VK_HIP(vk_launch_rwmd_batch(&pb, st));
		}
	}

	VK_HIP(hipEventRecord(c->ev[6], st));
	int64_t nw = 0;
	int cur = 0;
	VK_HIP(vk_launch_topk_wave_batch(c->d_bscores, nullptr, n, sel_floor, kk, 4096, n_queries, n, nw1 * kk, c->d_bkeys[0], &nw, st));
	int64_t stride = nw1 * kk;
	while (nw > 1) {
		const int64_t nkeys = nw * kk;
		const int64_t per_wave = nkeys <= 16384 ? nkeys : 4096;
		VK_HIP(vk_launch_topk_wave_batch(nullptr, c->d_bkeys[cur], nkeys, 0.0f, kk, per_wave, n_queries, stride, stride, c->d_bkeys[1 - cur], &nw, st));
		cur = 1 - cur;
	}
	// the turn passes AFTER the selection: a GEMM fills every CU for 40 ms, and a selection that starts beside the peer's GEMM waits
	// for it to end (0.5 ms of work returned 20 ms late, the handle's next batch submitted late: 1.6 - 1.9 ms of idle GPU per pair of
	// batches); behind the selection the peer's GEMM starts 0.5 ms later and this handle's host part runs beside it
	VK_HIP(hipEventRecord(c->ev[2], st));
	c->ev2_recorded = true;
	VK_HIP(hipEventRecord(c->ev[3], st));
	VK_HIP(hipEventRecord(c->ev[4], st));
	std::vector<uint64_t> &keys = keep.vec<uint64_t>((size_t)n_queries * (size_t)kk);
	VK_HIP(hipMemcpy2DAsync(keys.data(), (size_t)kk * 8, c->d_bkeys[cur], (size_t)stride * 8, (size_t)kk * 8, (size_t)n_queries, hipMemcpyDeviceToHost, st));
	VK_HIP(hipStreamSynchronize(st));
	for (int i = 0; i < n_queries && !canon_tr; i++) {   // without flows: the scores of the GEMM pass (kk == k)
		vk_topk_out *out = &outs[i];
		int n_out = 0;
		for (int j = 0; j < k; j++) {
			const uint64_t key = keys[(size_t)i * k + j];
			if (key == 0) break;
			const uint32_t ob = (uint32_t)(key >> 32);
			const uint32_t bits = (ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob;
			float s;
			memcpy(&s, &bits, 4);
			const int64_t g = (int64_t)(uint32_t)(key & 0xffffffffu);
			out->score[j] = s;
			out->sentence[j] = g;
			if (out->raw_score) out->raw_score[j] = (qs[i].boost ? s / qs[i].boost[g] : s) * (float)qs[i].len_t;
			if (qs[i].want_flow && out->mapping && out->edge_sim)
				for (int t = 0; t < qs[i].len_t; t++) {
					out->mapping[(size_t)j * qs[i].len_t + t] = -1;
					out->edge_sim[(size_t)j * qs[i].len_t + t] = 0.0f;
				}
			n_out++;
		}
		out->n_out = n_out;
	}
	if (canon_tr && (rc = batch_winner_rows(c, qs, n_queries, outs, keys.data(), kk, k, st, keep, tiles16.empty() ? nullptr : tiles16.data()))) return rc;
	c->have_scores = false;
	float ms = 0;
	vk_timings t{};
	if (hipEventElapsedTime(&ms, c->ev[0], c->ev[5]) == hipSuccess) t.prepare_ms = ms;
	if (hipEventElapsedTime(&ms, c->ev[5], c->ev[1]) == hipSuccess) t.queue_ms = ms;
	if (hipEventElapsedTime(&ms, c->ev[1], c->ev[6]) == hipSuccess) t.score_ms = ms;
	if (hipEventElapsedTime(&ms, c->ev[6], c->ev[3]) == hipSuccess) t.topk_ms = ms;
	if (hipEventElapsedTime(&ms, c->ev[0], c->ev[4]) == hipSuccess) t.total_ms = ms - t.queue_ms;
	c->last = t;
	return VK_OK;
}

// The rows of the slice table by length bucket, for the batched relaxed WMD over the static layout (vk_rwmd_static32_kernel takes two
// slices of at most 32 tokens per wave, or one of 33..64): built once per handle from the host mirror of the table.
static int build_static_buckets(vk_corpus *c) {
	if (c->sb_built) return VK_OK;
	const int64_t n = c->desc.n_sentences;
	std::vector<int32_t> ids[2];
	c->sb_empty = 0;
	for (int64_t s = 0; s < n; s++) {
		const int len = (*c->h_end)[(size_t)s] - (*c->h_start)[(size_t)s];
		if (len < 1) { c->sb_empty++; continue; }
		ids[len <= 32 ? 0 : 1].push_back((int32_t)s);
	}
	for (int b = 0; b < 2; b++) {
		c->sb_n[b] = (int64_t)ids[b].size();
		if (c->sb_n[b] == 0) continue;
		int rc;
		if ((rc = alloc_t(c, &c->d_sb_id[b], ids[b].size()))) return rc;
		VK_HIP(hipMemcpy(c->d_sb_id[b], ids[b].data(), ids[b].size() * 4, hipMemcpyHostToDevice));   // (blocking: the host vector is this function's)
	}
	c->sb_built = true;
	return VK_OK;
}

extern "C" {

int vk_query_batch(vk_corpus_t *c, const vk_query_desc *qs, int32_t n_queries, vk_topk_out *outs) {
	if (!c || !qs || !outs || n_queries < 0) return fail(VK_ERR_INVALID, "null argument");
	if (n_queries == 0) return VK_OK;
	if (qs[0].abort && *qs[0].abort) {
		for (int i = 0; i < n_queries; i++) outs[i].n_out = 0;
		return fail(VK_ERR_ABORTED, "batch aborted by the caller");
	}
	// an error or an abort between the passes leaves no copy in flight behind (vk_guard.h)
	return vk_run_guarded([&](vk_host_keep &keep) { return query_batch_body(c, qs, n_queries, outs, keep); },
		[&]() { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); },
		[](const char *what) { return fail(VK_ERR_INVALID, std::string("vk_query_batch: ") + what); });
}

} // extern "C"
