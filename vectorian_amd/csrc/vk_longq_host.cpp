// vk_longq_host.cpp -- C-ABI: alignments of queries of 65 .. VK_MAX_LONG_QUERY_LEN tokens (vk_longq_kernel: roles swapped, one wave
// per slice, anti-diagonal sweep).  Called by vk_query's body for such queries; the same stages as there -- scoring pass over all
// slices (MFMA similarities), selection of k + 8, canonical retrace of those, the k best returned -- with buffers of their own.
// No CPU compute fallback exists.

#include "vk_internal.h"
#include "vk_guard.h"

namespace {

template <typename T> int grow(vk_corpus *c, T **p, size_t *cap, size_t need) {
	if (*cap >= need && *p) return VK_OK;
	if (*p) { VK_HIP(hipFree(*p)); *p = nullptr; *cap = 0; }
	const int rc = alloc_t(c, p, need);
	if (rc == VK_OK) *cap = need;
	return rc;
}

} // namespace

int vk_longq_query(vk_corpus_t *c, const vk_query_desc *q, vk_topk_out *out, vk_host_keep &keep) {
	int rc = VK_OK;
	hipStream_t st = c->stream;
	const int64_t n = c->n_entries;
	const int k = q->max_matches, LT = q->len_t, nq = (LT + 15) / 16, LTP = 16 * nq;
	const bool only = q->only_slices != nullptr;
	const bool is_static = c->desc.layout == VK_LAYOUT_STATIC;
	auto &lq = c->lq;
	out->n_out = 0;
	c->have_scores = false;
	if (n == 0) return VK_OK;

	// ---- prepare: query tiles, gap tables, tag weights, token ids, static tables
	VK_HIP(hipEventRecord(c->ev[0], st));
	std::vector<uint8_t> &qtile = keep.vec<uint8_t>();
	std::vector<float> &qmags = keep.vec<float>((size_t)LTP);
	vk_pack_query(c, q, qtile, qmags.data());
	if ((rc = grow(c, &lq.qt, &lq.qt_cap, (size_t)nq * c->tile_bytes))) return rc;
	VK_HIP(hipMemcpyAsync(lq.qt, qtile.data(), qtile.size(), hipMemcpyHostToDevice, st));

	VkLongqParams p{};
	const int ks = q->gap_s.kind, kt = q->gap_t.kind;
	if (ks == VK_GAP_LINEAR && kt == VK_GAP_LINEAR) {
		p.gap_mode = 0; p.gs = q->gap_s.u; p.gt = q->gap_t.u;
	} else if ((ks == VK_GAP_LINEAR || ks == VK_GAP_AFFINE) && (kt == VK_GAP_LINEAR || kt == VK_GAP_AFFINE)) {
		p.gap_mode = 1;
		p.a_s = ks == VK_GAP_AFFINE ? q->gap_s.u : 0.0f; p.gs = ks == VK_GAP_AFFINE ? q->gap_s.v : q->gap_s.u;
		p.a_t = kt == VK_GAP_AFFINE ? q->gap_t.u : 0.0f; p.gt = kt == VK_GAP_AFFINE ? q->gap_t.v : q->gap_t.u;
		p.open_s = p.a_s + p.gs; p.open_t = p.a_t + p.gt;
	} else p.gap_mode = 2;
	// one host block for the small per-query arrays: w_s[0 .. 64 + pad], w_t[0 .. LT], tag weights, POS codes, token ids
	const size_t n_ws = 80, n_wt = (size_t)LT + 16;
	std::vector<float> &fl = keep.vec<float>(n_ws + n_wt + (size_t)LTP);
	std::vector<int32_t> &il = keep.vec<int32_t>(2 * (size_t)LTP);
	for (size_t i = 0; i < n_ws; i++) fl[i] = (int)i <= c->max_len ? gap_cost(q->gap_s, (int)i) : 0.0f;
	for (size_t i = 0; i < n_wt; i++) fl[n_ws + i] = (int)i <= LT ? gap_cost(q->gap_t, (int)i) : 0.0f;
	float total = (float)LT;
	if (q->tag_weights) {
		total = 0.0f;
		for (int j = 0; j < LT; j++) total += q->tag_weights[j];
	}
	for (int j = 0; j < LTP; j++) {
		fl[n_ws + n_wt + j] = (q->tag_weights && j < LT) ? q->tag_weights[j] : 0.0f;
		il[(size_t)j] = (q->tag_weights && j < LT) ? (int32_t)q->q_pos[j] : -1;
		il[(size_t)LTP + j] = (q->q_token_ids && j < LT) ? q->q_token_ids[j] : -1;
	}
	if ((rc = grow(c, &lq.fl, &lq.fl_cap, fl.size()))) return rc;
	if ((rc = grow(c, &lq.il, &lq.il_cap, il.size()))) return rc;
	VK_HIP(hipMemcpyAsync(lq.fl, fl.data(), fl.size() * 4, hipMemcpyHostToDevice, st));
	VK_HIP(hipMemcpyAsync(lq.il, il.data(), il.size() * 4, hipMemcpyHostToDevice, st));

	std::vector<float> &boost_rows = keep.vec<float>();
	if (q->boost) {
		if (!c->d_boost) { rc = alloc_t(c, &c->d_boost, (size_t)n + 8); if (rc) return rc; }
		const float *src = q->boost;
		if (!c->entry_sent.empty()) {
			boost_rows.resize((size_t)n);
			for (int64_t e = 0; e < n; e++) boost_rows[(size_t)e] = c->entry_sent[(size_t)e] >= 0 ? q->boost[c->entry_sent[(size_t)e]] : 1.0f;
			src = boost_rows.data();
		}
		VK_HIP(hipMemcpyAsync(c->d_boost, src, (size_t)n * 4, hipMemcpyHostToDevice, st));
	}
	const int64_t table_stride = (int64_t)c->n_tiles * 16 * 16;
	if (is_static && !only) {
		if ((rc = grow(c, &lq.table, &lq.table_cap, (size_t)nq * (size_t)table_stride))) return rc;
		for (int t = 0; t < nq; t++)   // one [V_pad x 16] table per 16 query tokens: cosine, sim[id(t_j)][j] = 1, clip (metric/static.cpp:9-78)
			VK_HIP(vk_launch_table(c->d_tiles, lq.qt + (size_t)t * c->tile_bytes, (int32_t)c->n_tiles, c->nk32, c->tail, c->tile_bytes,
				lq.table + (size_t)t * table_stride, q->q_token_ids ? lq.il + LTP + t * 16 : nullptr, std::min(16, LT - t * 16), c->desc.vocab_size, c->prec, st));
	}

	p.tiles = c->d_tiles; p.tok_id = c->d_tok_id; p.table = lq.table; p.table_stride = table_stride;
	p.sent_start = c->d_sent_start; p.sent_end = c->d_sent_end; p.n_sent = (int32_t)n;
	p.layout = is_static ? VK_DEV_LAYOUT_STATIC : VK_DEV_LAYOUT_CONTEXTUAL;
	p.nk32 = c->nk32; p.tail = c->tail; p.tile_bytes = c->tile_bytes; p.prec = c->prec;
	p.qtile = lq.qt; p.nq = nq; p.len_t = LT; p.locality = q->locality; p.s_stride = vk_longq_stride(c->max_len);
	p.ws = lq.fl; p.wt = lq.fl + n_ws;
	if (q->tag_weights) {
		p.pos_s = c->d_pos; p.tw = lq.fl + n_ws + n_wt; p.tpos = lq.il;
		p.tw_keep = 1.0f - q->pos_mismatch_penalty; p.tw_threshold = q->similarity_threshold;
	}
	p.ref_total = total;
	p.boost = q->boost ? c->d_boost : nullptr;
	p.scores = c->d_scores; p.raw = (q->want_flow ? nullptr : c->d_raw);   // with traceback the retrace restates the winners' aligner scores
	p.d = c->desc.d; p.q_ids = is_static ? lq.il + LTP : nullptr;

	// ---- the scoring pass (handles on one corpus take turns, as in vk_query)
	VK_HIP(hipEventRecord(c->ev[5], st));
	if ((rc = vk_wait_peer_turn(c, st))) return rc;
	VK_HIP(hipEventRecord(c->ev[1], st));
	if (!only) {
		if (p.gap_mode == 2) {
			// the constant tail of w_t (a saturated table): from which k on
			int kt_ = LT;
			while (kt_ > 1 && fl[n_ws + (size_t)kt_ - 1] == fl[n_ws + (size_t)LT]) kt_--;
			if (kt_ < LT) p.wt_tail = kt_;
			if (!vk_longq_hm_in_lds(LT, c->max_len)) {   // the matrix of the scans: in LDS behind the strip where it fits, else a region per workgroup
				const size_t per = vk_longq_scratch_bytes(LT, 2, 0, 0);
				if ((rc = grow(c, &lq.scratch, &lq.scratch_cap, per * (size_t)vk_longq_blocks(LT, c->max_len, n, 0)))) return rc;
				p.scratch = lq.scratch; p.scratch_stride = (int64_t)per;
			}
		}
		VK_HIP(vk_launch_longq(&p, 0, st));
	}
	VK_HIP(hipEventRecord(c->ev[2], st));
	c->ev2_recorded = true;

	// ---- bounded result set: k + 8 slices selected, all of them retraced canonically, the k best kept (vk_query)
	const bool do_flow = q->want_flow != 0;
	constexpr int kCanonMargin = 8;
	const int kk = only ? q->n_only : (int)std::min<int64_t>(do_flow ? std::min(k + kCanonMargin, VK_MAX_MATCHES) : k, n);
	const float sel_floor = do_flow ? q->min_score - 1e-5f * std::max(1.0f, std::fabs(q->min_score)) : q->min_score;
	int cur = 0;
	if (only) {
		std::vector<uint64_t> &hk = keep.vec<uint64_t>((size_t)q->n_only);
		for (int i = 0; i < q->n_only; i++) hk[(size_t)i] = (1ull << 32) | (uint64_t)(uint32_t)q->only_slices[i];   // (no long slices: rows == slices)
		VK_HIP(hipMemcpyAsync(c->d_keys[0], hk.data(), hk.size() * 8, hipMemcpyHostToDevice, st));
	} else if (kk <= 64) {
		int64_t nw = 0;
		VK_HIP(vk_launch_topk_wave(c->d_scores, nullptr, n, sel_floor, kk, 4096, c->d_keys[0], &nw, st));
		while (nw > 1) {
			const int64_t nkeys = nw * kk;
			const int64_t per_wave = nkeys <= 16384 ? nkeys : 4096;
			VK_HIP(vk_launch_topk_wave(nullptr, c->d_keys[cur], nkeys, 0.0f, kk, per_wave, c->d_keys[1 - cur], &nw, st));
			cur = 1 - cur;
		}
	} else {
		int nb = 0;
		VK_HIP(vk_launch_topk_scores(c->d_scores, n, sel_floor, kk, c->d_keys[0], &nb, st));
		while (nb > 1) {
			const int64_t nkeys = (int64_t)nb * kk;
			VK_HIP(vk_launch_topk_keys(c->d_keys[cur], nkeys, kk, c->d_keys[1 - cur], &nb, st));
			cur = 1 - cur;
		}
	}
	VK_HIP(hipEventRecord(c->ev[3], st));

	// ---- the winners' tracebacks
	if (do_flow) {
		const size_t per = vk_longq_scratch_bytes(LT, p.gap_mode, 1, q->tag_weights != nullptr);
		if ((rc = grow(c, &lq.fscratch, &lq.fscratch_cap, per * (size_t)kk))) return rc;
		if ((rc = grow(c, &lq.raw, &lq.raw_cap, (size_t)kk))) return rc;
		if ((rc = grow(c, &lq.map, &lq.map_cap, (size_t)kk * LTP))) return rc;
		if ((rc = grow(c, &lq.sim, &lq.sim_cap, (size_t)kk * LTP))) return rc;
		VkLongqParams f = p;
		f.wt_tail = 0;   // (the tracebacks meet every candidate, in the oracle's order)
		f.keys = c->d_keys[cur]; f.n_keys = kk; f.raw_out = lq.raw; f.mapping = lq.map; f.edge_sim = lq.sim; f.out_stride = LTP;
		f.scratch = lq.fscratch; f.scratch_stride = (int64_t)per;
		VK_HIP(vk_launch_longq(&f, kk, st));
	}
	VK_HIP(hipEventRecord(c->ev[4], st));

	// ---- results to the host
	std::vector<uint64_t> &keys = keep.vec<uint64_t>((size_t)kk);
	std::vector<float> &raw = keep.vec<float>((size_t)kk), &sim = keep.vec<float>(do_flow ? (size_t)kk * LTP : 0);
	std::vector<int16_t> &map = keep.vec<int16_t>(do_flow ? (size_t)kk * LTP : 0);
	VK_HIP(hipMemcpyAsync(keys.data(), c->d_keys[cur], (size_t)kk * 8, hipMemcpyDeviceToHost, st));
	if (do_flow) {
		VK_HIP(hipMemcpyAsync(raw.data(), lq.raw, (size_t)kk * 4, hipMemcpyDeviceToHost, st));
		VK_HIP(hipMemcpyAsync(map.data(), lq.map, map.size() * 2, hipMemcpyDeviceToHost, st));
		VK_HIP(hipMemcpyAsync(sim.data(), lq.sim, sim.size() * 4, hipMemcpyDeviceToHost, st));
	}
	VK_HIP(hipStreamSynchronize(st));
	int n_sel = 0;
	for (int i = 0; i < kk; i++) {
		if (keys[(size_t)i] == 0) break;
		n_sel++;
	}
	std::vector<int> order((size_t)n_sel);
	std::vector<float> val((size_t)std::max(n_sel, 1));
	for (int i = 0; i < n_sel; i++) {
		order[(size_t)i] = i;
		const uint32_t ob = (uint32_t)(keys[(size_t)i] >> 32);
		const uint32_t bits = (ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob;
		memcpy(&val[(size_t)i], &bits, 4);
	}
	auto sentence_of = [c](int64_t row) { return c->entry_sent.empty() ? row : (int64_t)c->entry_sent[(size_t)row]; };
	int n_out = n_sel;
	if (do_flow) {
		// Score of a winner from its canonical aligner score, operation by operation as the oracle's vko_score (match/match.h:295-307;
		// reference_score, metric/alignment.h:84-106: matched weight of this traceback, pow(., submatch_weight = 0) = 1)
		for (int i = 0; i < n_sel; i++) {
			float matched = 0.0f;
			for (int j = 0; j < LT; j++)
				if (map[(size_t)i * LTP + j] >= 0) matched += q->tag_weights ? q->tag_weights[j] : 1.0f;
			const float uw = powf((total - matched) / total, 0.0f);
			const float ref = matched + uw * (total - matched);
			const int64_t row = (int64_t)(uint32_t)(keys[(size_t)i] & 0xffffffffu);
			const float boost = q->boost ? q->boost[sentence_of(row)] : 1.0f;
			val[(size_t)i] = (raw[(size_t)i] / ref) * boost;
		}
		if (!only) {
			order.erase(std::remove_if(order.begin(), order.end(), [&](int i) { return !(val[(size_t)i] > q->min_score); }), order.end());
			std::sort(order.begin(), order.end(), [&](int a, int b) {   // the total order of the result set: score, then slice, descending
				if (val[(size_t)a] != val[(size_t)b]) return val[(size_t)a] > val[(size_t)b];
				return (uint32_t)(keys[(size_t)a] & 0xffffffffu) > (uint32_t)(keys[(size_t)b] & 0xffffffffu);
			});
		}
		n_out = std::min((int)order.size(), only ? q->n_only : k);
	} else if (out->raw_score && n_out > 0) {
		std::vector<float> &all_raw = keep.vec<float>((size_t)n);
		VK_HIP(hipMemcpyAsync(all_raw.data(), c->d_raw, (size_t)n * 4, hipMemcpyDeviceToHost, st));
		VK_HIP(hipStreamSynchronize(st));
		for (int i = 0; i < n_out; i++) raw[(size_t)i] = all_raw[(size_t)(uint32_t)(keys[(size_t)i] & 0xffffffffu)];
	}
	for (int i = 0; i < n_out; i++) {
		const int src = order[(size_t)i];
		out->score[i] = val[(size_t)src];
		out->sentence[i] = sentence_of((int64_t)(uint32_t)(keys[(size_t)src] & 0xffffffffu));
		if (out->raw_score) out->raw_score[i] = raw[(size_t)src];
		if (do_flow)
			for (int j = 0; j < LT; j++) {
				out->mapping[(size_t)i * LT + j] = map[(size_t)src * LTP + j];
				out->edge_sim[(size_t)i * LT + j] = sim[(size_t)src * LTP + j];
			}
	}
	out->n_out = n_out;
	c->have_scores = !only;

	float ms = 0;
	vk_timings t{};
	if (hipEventElapsedTime(&ms, c->ev[0], c->ev[5]) == hipSuccess) t.prepare_ms = ms;
	if (hipEventElapsedTime(&ms, c->ev[5], c->ev[1]) == hipSuccess) t.queue_ms = ms;
	if (hipEventElapsedTime(&ms, c->ev[1], c->ev[2]) == hipSuccess) t.score_ms = ms;
	if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) t.topk_ms = ms;
	if (hipEventElapsedTime(&ms, c->ev[3], c->ev[4]) == hipSuccess) t.flow_ms = ms;
	if (hipEventElapsedTime(&ms, c->ev[0], c->ev[4]) == hipSuccess) t.total_ms = ms - t.queue_ms;
	c->last = t;
	return VK_OK;
}
