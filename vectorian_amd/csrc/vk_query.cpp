// vk_query.cpp -- C-ABI: one query against a resident corpus (validation, launches, result assembly) and the
// merge of result sets.
// No CPU compute fallback exists: without a HIP device every entry point that
// would compute returns VK_ERR_NO_DEVICE / VK_ERR_HIP.

#include "vk_internal.h"
#include "vk_guard.h"
#include "vk_transport_host.h"

// The multi-block kernel (vk_score32_kernel) for a query of 17..64 tokens: the gap mode it is launched with, the token tiles a
// wave's strip spans, and whether the query tiles and at least one wave's strip fit the LDS of a CU (vk_score32_waves).  One
// place for vk_validate_query and vk_query: exact transport and the 1:n RWMD have no other kernel for such queries, so a shape that
// does not fit is refused before anything is enqueued -- with the gap mode of the launch, not a stand-in (round 2 tested mode 6 and
// launched 5 or 7, whose strips are up to 1,280 bytes larger: a borderline shape failed in hipFuncSetAttribute instead).
struct Score32Plan { int gap_mode, wave_tiles; bool fits, only_kernel; };
static Score32Plan score32_plan(const vk_corpus *c, const vk_query_desc *q) {
	const bool is_static = c->desc.layout == VK_LAYOUT_STATIC;
	const bool bound_pass = q->algorithm == VK_ALG_WRD || (q->algorithm == VK_ALG_RWMD && q->wmd_full);
	const bool fill = q->algorithm == VK_ALG_RWMD && !q->wmd_full && !q->rwmd_injective;
	const int ks = q->gap_s.kind, kt = q->gap_t.kind;
	int gm;
	if (bound_pass) gm = 5;
	else if (fill) gm = 7;
	else if (q->algorithm == VK_ALG_RWMD) gm = 4;
	else if (ks == VK_GAP_LINEAR && kt == VK_GAP_LINEAR) gm = 0;
	else if (ks != VK_GAP_TABLE && kt != VK_GAP_TABLE) gm = 1;
	else gm = -1;   // general gaps: by the longest slice the kernel sees (below)
	// alignments and the relaxed 1:1 WMD over a corpus that holds slices of more than 64 tokens (round 4): the multi-block kernel takes the
	// others, the slices apart a pass of their own (vk_docw_kernel under linear / affine gaps, else vk_wide_kernel over their list)
	const bool apart = (q->algorithm == VK_ALG_ALIGN || gm == 4) && c->h_apart && !c->h_apart->empty() && !getenv("VK_NO_APART");
	if (gm < 0) gm = (apart ? c->max_short_len : c->max_len) <= 32 ? 3 : 6;
	const bool long_apart = ((bound_pass || fill) && c->n_long_groups > 0) || apart;   // the long slices have kernels of their own
	const int wave_tiles = long_apart ? (q->len_t <= 32 ? c->max_short_pair_tiles : (c->max_short_len + 15) / 16 + 1)
		: (q->len_t <= 32 ? c->max_pair_tiles : (c->max_len + 15) / 16 + 1);
	const bool fits = vk_score32_waves(is_static ? 0 : c->nk32, c->tail, wave_tiles, q->len_t, gm) >= 1;
	return {gm, wave_tiles, fits, bound_pass || fill};
}

int vk_validate_query(const vk_corpus *c, const vk_query_desc *q, const vk_topk_out *out) {
	if (!c || !q || !out) return fail(VK_ERR_INVALID, "null argument");
	if (!c->finalized) return fail(VK_ERR_STATE, "corpus not finalized");
	if (q->len_t < 1) return fail(VK_ERR_INVALID, "empty query");
	if (q->len_t > VK_MAX_LONG_QUERY_LEN) return fail(VK_ERR_UNSUPPORTED, "query longer than VK_MAX_LONG_QUERY_LEN (512) tokens");
	if (q->len_t > VK_MAX_QUERY_LEN) {
		// 65 .. 512 tokens: the role-swapped one-wave-per-slice kernel (vk_longq_kernel) -- alignments over slices of at most 64 tokens
		if (q->algorithm != VK_ALG_ALIGN) return fail(VK_ERR_UNSUPPORTED, "queries of more than VK_MAX_QUERY_LEN (64) tokens: alignments only (the transports keep a lane per query token)");
		if (c->max_len > VK_FAST_SENT_LEN) return fail(VK_ERR_UNSUPPORTED, "queries of more than 64 tokens over a corpus that holds a slice of more than 64 tokens: a lane per slice token");
		if (q->submatch_weight != 0.0f) return fail(VK_ERR_UNSUPPORTED, "queries of more than 64 tokens with a submatch weight");
		if (out->sim_rows) return fail(VK_ERR_UNSUPPORTED, "queries of more than 64 tokens: similarity rows of the winners are not returned");
		if (q->max_matches > VK_MAX_MATCHES) return fail(VK_ERR_UNSUPPORTED, "queries of more than 64 tokens: max_matches <= VK_MAX_MATCHES");
	}
	const bool exact_tr = q->algorithm == VK_ALG_WRD || (q->algorithm == VK_ALG_RWMD && (q->wmd_full || !q->rwmd_injective));   // multi-block kernel + kernels of their own for the long slices: no wide kernel
	if (q->len_t > VK_FAST_QUERY_LEN && exact_tr && !score32_plan(c, q).fits)
		return fail(VK_ERR_UNSUPPORTED, "exact transport / 1:n RWMD with a query of more than 16 tokens: the query tiles of rows this wide and one wave's similarity strip exceed the LDS of a workgroup (160 KiB)");
	// Slices of more than VK_MAX_SENT_LEN tokens (whole documents as slices, up to VK_MAX_DOC_LEN): alignments -- the
	// one-wave-per-slice kernel with the slice's state in global memory (vk_wide_kernel, global-state form; the same form takes a
	// query of more than 16 tokens whose state over the corpus's longest slice exceeds the LDS: VK_ERR_UNSUPPORTED in round 2) --
	if (c->max_len > VK_MAX_SENT_LEN && q->algorithm != VK_ALG_ALIGN) {
		// ... and the relaxed word mover's distance in its 1:1 form (a stream of row / column minima; the winners' rows restated
		// tile by tile, vk_canon_rows_kernel); the 1:n form and the exact transports keep a slice's bag of words in LDS
		const bool relaxed_11 = q->algorithm == VK_ALG_RWMD && !q->wmd_full && q->rwmd_injective;
		if (!relaxed_11)
			return fail(VK_ERR_UNSUPPORTED, "slices of more than VK_MAX_SENT_LEN (512) tokens: alignments and the relaxed 1:1 word mover's distance only (the 1:n form and the exact transports keep a slice's bag of words in LDS)");
		if (c->desc.layout == VK_LAYOUT_STATIC && q->tag_weights && q->q_tags && q->q_token_ids && c->d_tag)
			return fail(VK_ERR_UNSUPPORTED, "slices of more than VK_MAX_SENT_LEN (512) tokens: tag-weighted vocabulary transports keyed by (id, tag) rewrite cells of a slice's rows in LDS");
	}
	if (!q->q_vectors) return fail(VK_ERR_INVALID, "q_vectors is null");
	if (q->q_dtype != VK_F32 && q->q_dtype != VK_BF16) return fail(VK_ERR_INVALID, "bad q_dtype");
	if (q->max_matches < 1 || q->max_matches > VK_MAX_MATCHES_SORTED) return fail(VK_ERR_INVALID, "max_matches out of range");
	// beyond VK_MAX_MATCHES: every score sorted on the device -- alignments (with or without their tracebacks), no submatch weight;
	// the transports' candidate rounds and row buffers are sized for VK_MAX_MATCHES
	if (q->max_matches > VK_MAX_MATCHES && !q->only_slices && (q->algorithm != VK_ALG_ALIGN || q->submatch_weight != 0.0f || out->sim_rows))
		return fail(VK_ERR_UNSUPPORTED, "max_matches beyond VK_MAX_MATCHES (1024): alignments without a submatch weight only");
	if (out->capacity < q->max_matches) return fail(VK_ERR_INVALID, "output capacity smaller than max_matches");
	if (!out->score || !out->sentence) return fail(VK_ERR_INVALID, "output arrays missing");
	// (the batched paths copy 64 rows per winner into sim_rows at a stride of rows_per_winner: never below 64)
	if (out->sim_rows && out->rows_per_winner != 0 && (out->rows_per_winner < VK_FAST_SENT_LEN || out->rows_per_winner % 64 != 0 || out->rows_per_winner > VK_MAX_DOC_LEN + 1))
		return fail(VK_ERR_INVALID, "rows_per_winner must be 0 or a multiple of 64 (64 .. VK_MAX_DOC_LEN + 1)");
	if (!(q->submatch_weight >= 0.0f)) return fail(VK_ERR_INVALID, "submatch_weight must be >= 0 (pow of a zero base, metric/alignment.h:97-99)");
	if (q->only_slices) {
		if (q->n_only < 1 || q->n_only > VK_MAX_MATCHES || q->n_only > out->capacity) return fail(VK_ERR_INVALID, "only_slices: n_only out of range (1 .. min(VK_MAX_MATCHES, capacity))");
		const bool relaxed = q->algorithm == VK_ALG_RWMD && !q->wmd_full && out->sim_rows != nullptr;   // restated on the host from the rows
		const bool exact = q->algorithm == VK_ALG_WRD || (q->algorithm == VK_ALG_RWMD && q->wmd_full);  // every listed slice solved
		// (a submatch weight: alignments only -- the score of a listed slice is its aligner score over the reference score of its own
		// traceback, metric/alignment.h:84-106, no candidate rounds; the transports' reference score does not depend on it)
		if (!(q->algorithm == VK_ALG_ALIGN || relaxed || exact) || !q->want_flow || (q->submatch_weight != 0.0f && q->algorithm != VK_ALG_ALIGN))
			return fail(VK_ERR_UNSUPPORTED, "only_slices states alignments, relaxed WMD with sim_rows, or exact transports, with want_flow (and submatch_weight = 0 for the transports)");
		for (int i = 0; i < q->n_only; i++)
			if (q->only_slices[i] < 0 || q->only_slices[i] >= c->desc.n_sentences) return fail(VK_ERR_INVALID, "only_slices: slice index out of range");
	}
	if (q->bidirectional) return fail(VK_ERR_UNSUPPORTED, "bidirectional is not implemented (unused upstream, query.cpp:81-83)");
	if (q->algorithm == VK_ALG_ALIGN) {
		if (q->locality < VK_LOCAL || q->locality > VK_SEMIGLOBAL) return fail(VK_ERR_INVALID, "bad locality");
		for (const vk_gap *g : {&q->gap_s, &q->gap_t}) {
			if (g->kind < VK_GAP_LINEAR || g->kind > VK_GAP_TABLE) return fail(VK_ERR_INVALID, "bad gap kind");
			if (g->kind == VK_GAP_TABLE && (!g->table || g->n_table < 1)) return fail(VK_ERR_INVALID, "gap table missing");
		}
		if (q->gap_s.kind == VK_GAP_TABLE && q->gap_s.n_table <= c->max_len) return fail(VK_ERR_INVALID, "gap_s table shorter than the longest sentence");
		if (q->gap_t.kind == VK_GAP_TABLE && q->gap_t.n_table <= q->len_t) return fail(VK_ERR_INVALID, "gap_t table shorter than the query");
		if (q->want_flow && (!out->mapping || !out->edge_sim)) return fail(VK_ERR_INVALID, "want_flow needs mapping and edge_sim arrays");
		if (q->tag_weights) {
			if (!q->q_pos) return fail(VK_ERR_INVALID, "tag-weighted query without q_pos");
			if (!c->d_pos) return fail(VK_ERR_STATE, "tag-weighted query needs vk_corpus_set_token_pos");
			if (q->similarity_threshold < 0.0f) return fail(VK_ERR_INVALID, "similarity_threshold must be >= 0 (slice/static.h:209)");
		}
	} else if (q->algorithm == VK_ALG_RWMD) {
		if (q->tag_weights) {   // TagWeightedSlice wraps any slice, whatever the matcher (match/instantiate.cpp:173-189)
			if (!q->q_pos) return fail(VK_ERR_INVALID, "tag-weighted query without q_pos");
			if (!c->d_pos) return fail(VK_ERR_STATE, "tag-weighted query needs vk_corpus_set_token_pos");
			if (q->similarity_threshold < 0.0f) return fail(VK_ERR_INVALID, "similarity_threshold must be >= 0 (slice/static.h:209)");
			if (q->q_tags && c->desc.layout == VK_LAYOUT_STATIC) {
				// vocabulary keys (token id, tag) as id * 256 + tag in 32 bits, ordered as upstream's signed pairs: ids below 2^23, tags 0 .. 127
				if (c->desc.vocab_size > (1 << 23)) return fail(VK_ERR_UNSUPPORTED, "tag-weighted transport with q_tags over the static layout: vocabularies of more than 2^23 entries overflow the (id, tag) keys");
				for (int j = 0; j < q->len_t; j++) {
					if (q->q_tags[j] < 0) return fail(VK_ERR_INVALID, "q_tags: tag codes must be 0 .. 127");
					// (words the corpus does not hold carry ids of their own above the vocabulary, QueryVocabulary: they key entries too)
					if (q->q_token_ids && q->q_token_ids[j] >= (1 << 23)) return fail(VK_ERR_UNSUPPORTED, "q_token_ids: ids of 2^23 and more overflow the (id, tag) keys");
				}
			}
			if (q->q_tags && !q->rwmd_injective && !q->wmd_full && c->desc.layout == VK_LAYOUT_STATIC && !c->d_tag)
				return fail(VK_ERR_STATE, "tag-weighted 1:n RWMD over the static layout with q_tags needs vk_corpus_set_token_tags (its vocabulary is keyed by (token, tag), bow.h:150-176)");
		}
		if (q->rwmd_symmetric && !q->rwmd_normalize_bow)
			return fail(VK_ERR_INVALID, "cannot run symmetric mode WMD with bow (needs nbow)");   // wmd.h:441-449
		if (q->wmd_full) {
			if (q->rwmd_injective) return fail(VK_ERR_INVALID, "non-relaxed WMD with injective mapping is not supported");      // wmd.h:201-204
			if (q->rwmd_symmetric) return fail(VK_ERR_INVALID, "non-relaxed WMD with symmetric computation is not supported");  // wmd.h:206-209
		}
		if (q->want_flow && (!out->mapping || !out->edge_sim)) return fail(VK_ERR_INVALID, "want_flow needs mapping and edge_sim arrays");
	} else if (q->algorithm == VK_ALG_WRD) {
		if (q->tag_weights) {
			if (!q->q_pos) return fail(VK_ERR_INVALID, "tag-weighted query without q_pos");
			if (!c->d_pos) return fail(VK_ERR_STATE, "tag-weighted query needs vk_corpus_set_token_pos");
			if (q->similarity_threshold < 0.0f) return fail(VK_ERR_INVALID, "similarity_threshold must be >= 0 (slice/static.h:209)");
		}
		if (!c->d_mag) return fail(VK_ERR_STATE, "VK_ALG_WRD needs a corpus created with keep_magnitudes = 1");
		if (q->want_flow && (!out->mapping || !out->edge_sim)) return fail(VK_ERR_INVALID, "want_flow needs mapping and edge_sim arrays");
	} else {
		return fail(VK_ERR_INVALID, "bad algorithm");
	}
	return VK_OK;
}

// Vectors.normalized for the query rows, then bf16 (RNE), then tile order (16 rows,
// rows >= len_t zero).  Same arithmetic as oracle/vk_oracle.c vko_normalize_rows_bf16.
void vk_pack_query(const vk_corpus *c, const vk_query_desc *q, std::vector<uint8_t> &tile, float *mags) {
	const int d = c->desc.d;
	tile.assign((size_t)c->tile_bytes * (size_t)((q->len_t + 15) / 16), 0);   // tile i / 16 holds row i % 16
	std::vector<float> row((size_t)d);
	for (int i = 0; i < q->len_t; i++) {
		for (int k = 0; k < d; k++)
			row[(size_t)k] = q->q_dtype == VK_F32 ? ((const float *)q->q_vectors)[(size_t)i * d + k]
			                                       : bf16_to_f32(((const uint16_t *)q->q_vectors)[(size_t)i * d + k]);
		double acc = 0.0;
		for (int k = 0; k < d; k++) acc += (double)row[(size_t)k] * (double)row[(size_t)k];
		float m = (float)std::sqrt(acc);
		if (m != m) m = 0.0f;
		mags[i] = m;
		if (q->q_normalize) {
			for (int k = 0; k < d; k++) {
				float v = row[(size_t)k] / m;
				if (v != v) v = 0.0f;
				row[(size_t)k] = v;
			}
		}
		for (int k = 0; k < d; k++) {
			if (c->prec) {   // fp32 tile: block k >> 4, lane 16 (k & 3) + row, element (k & 15) >> 2
				const size_t off = (size_t)(i >> 4) * c->tile_bytes + (size_t)(k >> 4) * 1024 + (size_t)((k & 3) * 16 + (i & 15)) * 16 + (size_t)((k & 15) >> 2) * 4;
				memcpy(&tile[off], &row[(size_t)k], 4);
				continue;
			}
			const uint16_t b = f32_to_bf16(row[(size_t)k]);
			const int t = k >> 5, g = (k & 31) >> 3, j = k & 7;
			const size_t off = (size_t)(i >> 4) * c->tile_bytes + (size_t)t * 1024 + (size_t)(g * 16 + (i & 15)) * 16 + (size_t)j * 2;
			memcpy(&tile[off], &b, 2);
		}
	}
}

// One query.  Every host buffer that is the source or the destination of an asynchronous copy lives in `keep`, which the entry point
// (vk_query below) owns: when this body returns an error with copies still in flight, the entry point synchronises the stream before
// the buffers die (vk_guard.h).
static int query_body(vk_corpus_t *c, const vk_query_desc *q, vk_topk_out *out, vk_host_keep &keep) {
	int rc = VK_OK;
	VK_HIP(hipSetDevice(c->device));
	if (q->len_t > VK_MAX_QUERY_LEN) return vk_longq_query(c, q, out, keep);   // 65 .. 512 tokens (vk_longq_host.cpp)
	hipStream_t st = c->stream;
	const int64_t n = c->n_entries;           // rows of the slice table (== n_sentences unless long slices were padded)
	const int k = q->max_matches;
	const bool only = q->only_slices != nullptr;   // state the listed slices: no scoring pass, no selection
	out->n_out = 0;
	c->have_scores = false;
	if (n == 0) return VK_OK;
	auto sentence_of = [c](int64_t row) { return c->entry_sent.empty() ? row : (int64_t)c->entry_sent[(size_t)row]; };
	const bool is_static_l = c->desc.layout == VK_LAYOUT_STATIC;
	// transport algorithms: similarity rows (and, for exact transport, the optimal plan) of the winners, from which
	// the host states their SparseFlow / DenseFlow.  rows_idx: rows of the slice table, best first.
	// what vk_wrd_exact_kernel / vk_rows_kernel need to restate the similarity rows of a slice (tag weights included)
	auto fill_transport = [&](VkWrdParams &w) {
		w.tiles = c->d_tiles; w.tok_id = c->d_tok_id; w.table = c->d_table; w.table_stride = (int64_t)c->n_tiles * 16 * 16;
		w.sent_start = c->d_sent_start; w.sent_end = c->d_sent_end;
		w.layout = is_static_l ? VK_DEV_LAYOUT_STATIC : VK_DEV_LAYOUT_CONTEXTUAL; w.nk32 = c->nk32; w.tail = c->tail; w.tile_bytes = c->tile_bytes; w.prec = c->prec;
		w.qtile = c->d_qtile; w.nq = (q->len_t + 15) / 16; w.len_t = q->len_t; w.mag = c->d_mag;
		w.d = c->desc.d; w.q_ids = is_static_l ? c->d_qids : nullptr;   // canonical similarity rows (sim_canon)
		w.ref_total = (float)q->len_t;
		if (q->tag_weights) {
			float total = 0.0f;
			for (int j = 0; j < q->len_t; j++) total += q->tag_weights[j];
			for (int j = 0; j < VK_MAX_QUERY_LEN; j++) {
				w.tw[j] = j < q->len_t ? q->tag_weights[j] : 0.0f;
				w.tpos[j] = j < q->len_t ? (int32_t)q->q_pos[j] : -1;
			}
			w.pos_s = c->d_pos; w.tw_keep = 1.0f - q->pos_mismatch_penalty; w.tw_threshold = q->similarity_threshold;
			w.ref_total = total;   // reference_score with max_sum_of_similarities = sum of t_pos_weights (slice/static.h:280-286)
			if (is_static_l && q->algorithm == VK_ALG_RWMD && q->q_tags && q->q_token_ids && c->d_tag && c->d_qbits) {
				// the cells upstream writes twice (static_vocab_fixup); d_qbits holds this query's bitmap (set before the scoring launch)
				w.tag_s = c->d_tag; w.qid_bits = c->d_qbits;
				for (int j = 0; j < VK_MAX_QUERY_LEN; j++)
					w.qkey[j] = (j < q->len_t && q->q_token_ids[j] >= 0 && q->q_token_ids[j] < c->desc.vocab_size) ? q->q_token_ids[j] * 256 + ((int32_t)q->q_tags[j] & 255) : -1;
			}
		}
	};
	auto transport_flows = [&](const std::vector<int64_t> &rows_idx, bool exact, const float *qmass, int mass_mode, int raw_masses, float *rows_dst = nullptr) -> int {
		if (!q->want_flow || !out->sim_rows || rows_idx.empty()) return VK_OK;
		if (!rows_dst) rows_dst = out->sim_rows;
		const int nqw = (q->len_t + 15) / 16, W = 16 * nqw;   // columns of a similarity row: the query length padded to 16
		const int R = out->rows_per_winner > 0 ? out->rows_per_winner : VK_FAST_SENT_LEN;   // rows per winner (longer winners: zero rows)
		if (R % 64 != 0 || R > VK_MAX_DOC_LEN + 1 || (exact && R > VK_MAX_SENT_LEN))
			return fail(VK_ERR_INVALID, "rows_per_winner must be a multiple of 64, at most VK_MAX_SENT_LEN (rows without plans: VK_MAX_DOC_LEN + 1)");
		int rc2;
		const int cnt = (int)rows_idx.size();
		const size_t need = (size_t)cnt * R * W;
		if (c->rows_cap < need) {
			if (c->d_rows_out) { VK_HIP(hipFree(c->d_rows_out)); VK_HIP(hipFree(c->d_plan_out)); c->d_rows_out = c->d_plan_out = nullptr; }
			if ((rc2 = alloc_t(c, &c->d_rows_out, need))) return rc2;
			if ((rc2 = alloc_t(c, &c->d_plan_out, need))) return rc2;
			c->rows_cap = need;
		}
		if (!c->d_wrd_raw) {
			if ((rc2 = alloc_t(c, &c->d_wrd_raw, (size_t)VK_MAX_MATCHES))) return rc2;
			if ((rc2 = alloc_t(c, &c->d_wrd_val, (size_t)VK_MAX_MATCHES))) return rc2;
			c->wrd_cap = VK_MAX_MATCHES;
		}
		std::vector<uint64_t> &hk = keep.vec<uint64_t>((size_t)cnt);
		for (int i = 0; i < cnt; i++) hk[(size_t)i] = (1ull << 32) | (uint64_t)(uint32_t)rows_idx[(size_t)i];
		VK_HIP(hipMemcpyAsync(c->d_keys[1], hk.data(), hk.size() * 8, hipMemcpyHostToDevice, c->stream));
		VkWrdParams w{};
		fill_transport(w);
		w.keys = c->d_keys[1]; w.rows_out = c->d_rows_out; w.rows_len = R;
		if (R > VK_MAX_SENT_LEN) {   // rows of whole documents: one wave per 16 tokens of a winner (vk_canon_rows_kernel)
			VK_HIP(hipMemsetAsync(c->d_rows_out, 0, need * 4, c->stream));
			VK_HIP(vk_launch_canon_rows(&w, cnt, (c->max_len + 15) / 16 + 1, c->stream));
		} else VK_HIP(vk_launch_rows(&w, cnt, c->stream));
		VK_HIP(hipMemcpyAsync(rows_dst, c->d_rows_out, need * 4, hipMemcpyDeviceToHost, c->stream));
		if (exact && out->plan) {
			w.mass_mode = mass_mode; w.raw_masses = raw_masses;
			memcpy(w.qmass, qmass, sizeof w.qmass);
			w.raw_out = c->d_wrd_raw; w.val_out = c->d_wrd_val; w.plan_out = c->d_plan_out;
			VK_HIP(hipMemsetAsync(c->d_plan_out, 0, need * 4, c->stream));   // a solver writes the columns of its winner's tokens only
			VK_HIP(vk_launch_wrd_exact(&w, cnt, nullptr, c->stream));
			if (R > VK_FAST_SENT_LEN && c->max_len > VK_FAST_SENT_LEN) {   // winners of 65 .. R tokens: the long solver restates their plans
				if (w.nq > 1 && !c->d_wrdl_scratch) {
					if ((rc2 = alloc_t(c, &c->d_wrdl_scratch, (size_t)vk_wrd_long_blocks() * vk_wrd_long_scratch_bytes()))) return rc2;
				}
				w.scratch = c->d_wrdl_scratch; w.scratch_stride = (int64_t)vk_wrd_long_scratch_bytes();
				VK_HIP(vk_launch_wrd_exact_long(&w, cnt, c->stream));
			}
			VK_HIP(hipMemcpyAsync(out->plan, c->d_plan_out, need * 4, hipMemcpyDeviceToHost, c->stream));
		}
		VK_HIP(hipStreamSynchronize(c->stream));
		return VK_OK;
	};

	// ---- prepare: query tile, gap tables, boost, static table -------------
	VK_HIP(hipEventRecord(c->ev[0], st));
	std::vector<uint8_t> &qtile = keep.vec<uint8_t>();
	float qmags[VK_MAX_QUERY_LEN] = {0};
	// whole documents as slices (beyond VK_MAX_SENT_LEN tokens): scored by the one-wave-per-slice kernel, every winner retraced by it;
	// the other slices of such a corpus keep their fused kernels when the query has at most 16 tokens (wide_score false)
	const bool xlong = c->max_len > VK_MAX_SENT_LEN;
	const bool wide_score = q->len_t > VK_FAST_QUERY_LEN;
	// General gaps over slices of 65 .. 512 tokens: the one-wave-per-slice kernel (candidate scan dealt out over the idle lanes, eight
	// loads deep, a saturated table as a running maximum) is several times faster than the fused kernel's pass over long slices with
	// its serial LDS-history scan (8,000 slices of 300 .. 512 tokens at 300-d: 101 ms) -- they take the documents' pass as well,
	// and every winner is retraced by the same kernel
	// (round 4: linear / affine gaps too -- the skewed sweep of vk_doc_kernel is faster than the fused kernel's long pass; VK_LONG_LINEAR=1: that pass)
	const bool long_via_wide = q->algorithm == VK_ALG_ALIGN && !wide_score && c->n_long_groups > 0 &&
		(q->gap_s.kind == VK_GAP_TABLE || q->gap_t.kind == VK_GAP_TABLE || !getenv("VK_LONG_LINEAR")) && !getenv("VK_LONG_PASS");
	const bool wide = wide_score || xlong || long_via_wide;
	// the relaxed 1:1 WMD over slices of 65 .. 512 tokens: scored by vk_doc_kernel's streaming arm instead of the fused kernel's long pass
	// (the winners keep their path: rows from vk_rows_kernel, scores restated on the host)
	const bool rwmd_long_doc = q->algorithm == VK_ALG_RWMD && q->rwmd_injective && !q->wmd_full && !wide_score && !xlong && c->n_long_groups > 0 && !getenv("VK_LONG_PASS");
	// whole documents under linear / affine gaps with a query of at most 16 tokens: vk_doc_kernel scores them and retraces the winners
	// (general gaps: under a table that saturates within 126 tokens -- wp.ws_tail, set below; doc_ok() asks once wp is filled)
	// (... and the slices of 65 .. 512 tokens that general gaps send through the one-wave-per-slice pass)
	const bool doc_fast = (xlong || rwmd_long_doc || (long_via_wide && !getenv("VK_NO_DOC_MID"))) && !wide_score && (q->algorithm == VK_ALG_ALIGN || q->algorithm == VK_ALG_RWMD) && !getenv("VK_NO_DOC_KERNEL");
	const int nq = (q->len_t + 15) / 16;
	vk_pack_query(c, q, qtile, qmags);
	VK_HIP(hipMemcpyAsync(c->d_qtile, qtile.data(), qtile.size(), hipMemcpyHostToDevice, st));

	VkScoreParams p{};
	bool span_skip_raw = false;   // span-embedding path without its second output array
	float qmass_all[VK_MAX_QUERY_LEN] = {0};   // masses of the query tokens (transport algorithms), all 64 columns
	const int ks = q->gap_s.kind, kt = q->gap_t.kind;
	const size_t n_ws = std::max<size_t>((size_t)kGapTable, (size_t)c->max_len + 2);   // w_s up to the longest slice
	std::vector<float> &ws = keep.vec<float>(n_ws);
	float *wt = keep.array<float>(160);   // wt[0..79]: w_t as given; wt[80..159]: its subadditive closure
	const bool is_align = q->algorithm == VK_ALG_ALIGN;
	if (q->algorithm == VK_ALG_WRD) {
		p.gap_mode = 5;
		float sum_t = 0.0f;
		for (int j = 0; j < q->len_t; j++) sum_t += qmags[j];           // wrd.h:99-102, float sum in order
		const bool rawm = !q->wrd_normalize_magnitudes;   // wrd.h:99-102: masses stay the magnitudes
		for (int j = 0; j < VK_MAX_QUERY_LEN; j++) qmass_all[j] = j < q->len_t ? (rawm ? qmags[j] : qmags[j] / sum_t) : 0.0f;
		memcpy(p.qmass, qmass_all, sizeof p.qmass);
		p.wrd_raw_total = rawm ? sum_t : 0.0f;
		p.mag = c->d_mag;
	} else if (q->algorithm == VK_ALG_RWMD) {
		p.gap_mode = 4;
		p.rwmd_symmetric = q->rwmd_symmetric;
		p.rwmd_normalize_bow = q->rwmd_normalize_bow;
		if (q->wmd_full) {
			p.wmd_bound = q->rwmd_normalize_bow ? 1 : 2;
			for (int j = 0; j < q->len_t; j++) qmass_all[j] = q->rwmd_normalize_bow ? 1.0f / (float)q->len_t : 1.0f;
		}
		else if (!q->rwmd_injective) {
			// 1:n form: masses of the query's vocabulary entries (count / len at the first occurrence of a token id)
			p.gap_mode = 7;
			const bool ids = c->desc.layout == VK_LAYOUT_STATIC && q->q_token_ids;
			for (int j = 0; j < VK_MAX_QUERY_LEN; j++) {
				float mass = 0.0f;
				if (j < q->len_t) {
					int cnt = 1;
					bool first = true;
					// (tag-weighted with q_tags: the entries are (token id, tag) pairs, TaggedTokenFactory, bow.h:150-176)
					const bool tagged = q->tag_weights && q->q_tags;
					if (ids && q->q_token_ids[j] >= 0)
						for (int i = 0; i < q->len_t; i++)
							if (i != j && q->q_token_ids[i] == q->q_token_ids[j] && (!tagged || q->q_tags[i] == q->q_tags[j])) { cnt++; if (i < j) first = false; }
					mass = first ? (q->rwmd_normalize_bow ? (float)cnt / (float)q->len_t : (float)cnt) : 0.0f;
				}
				qmass_all[j] = mass;
			}
			memcpy(p.qmass, qmass_all, sizeof p.qmass);
			if (ids && q->tag_weights && q->q_tags) p.tag_s = c->d_tag;
		}
	} else if (ks == VK_GAP_LINEAR && kt == VK_GAP_LINEAR) {
		p.gap_mode = 0;
		p.gs = q->gap_s.u; p.gt = q->gap_t.u;
	} else if ((ks == VK_GAP_LINEAR || ks == VK_GAP_AFFINE) && (kt == VK_GAP_LINEAR || kt == VK_GAP_AFFINE)) {
		p.gap_mode = 1;
		p.a_s = ks == VK_GAP_AFFINE ? q->gap_s.u : 0.0f;
		p.gs = ks == VK_GAP_AFFINE ? q->gap_s.v : q->gap_s.u;
		p.a_t = kt == VK_GAP_AFFINE ? q->gap_t.u : 0.0f;
		p.gt = kt == VK_GAP_AFFINE ? q->gap_t.v : q->gap_t.u;
		p.open_s = p.a_s + p.gs;
		p.open_t = p.a_t + p.gt;
	} else {
		p.gap_mode = 2;
	}
	for (size_t i = 0; i < n_ws; i++) ws[i] = (is_align && (int64_t)i <= c->max_len) ? gap_cost(q->gap_s, (int)i) : 0.0f;
	if (c->ws_cap < n_ws) {
		if (c->d_ws) { VK_HIP(hipFree(c->d_ws)); c->d_ws = nullptr; c->ws_cap = 0; }
		if ((rc = alloc_t(c, &c->d_ws, n_ws))) return rc;
		c->ws_cap = n_ws;
	}
	for (int i = 0; i < 80; i++) wt[i] = (is_align && i <= q->len_t) ? gap_cost(q->gap_t, i) : 0.0f;
	for (int i = 80; i < 160; i++) wt[i] = 0.0f;
	// The register-history kernels take their in-row candidates from the row's values before in-row gaps, which is the
	// sequential recurrence with w_t replaced by its subadditive closure w* (dp_general_reg in vk_common.hip.h): wt[80..159].
	// (Round 1 sent every table that was not strictly subadditive -- a linear cost handed over as a table, a convex one -- to the
	// LDS-history kernel with its serial in-row chain: 12.7 ms per 1 M x 32 tokens against 2.9 ms.)
	bool wide_sub = false;   // a long query with general gaps: the multi-block kernel takes it
	for (int k = 0; k < 80; k++) wt[80 + k] = wt[k];
	for (int k = 2; k <= q->len_t && k < 80; k++)
		for (int a = 1; a < k; a++) wt[80 + k] = std::min(wt[80 + k], wt[80 + a] + wt[80 + k - a]);
	if (p.gap_mode == 2) {
		if (!wide_score) p.gap_mode = c->max_short_len <= 32 ? 3 : 6;
		wide_sub = wide;
	}
	VK_HIP(hipMemcpyAsync(c->d_ws, ws.data(), n_ws * sizeof(float), hipMemcpyHostToDevice, st));
	VK_HIP(hipMemcpyAsync(c->d_wt, wt, 160 * sizeof(float), hipMemcpyHostToDevice, st));

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

	const bool is_static = c->desc.layout == VK_LAYOUT_STATIC;
	const int64_t table_stride = (int64_t)c->n_tiles * 16 * 16;
	if (is_static) {
		int32_t *ids = keep.array<int32_t>(80);
		for (int j = 0; j < 80; j++) ids[j] = (q->q_token_ids && j < q->len_t) ? q->q_token_ids[j] : -1;
		VK_HIP(hipMemcpyAsync(c->d_qids, ids, 80 * sizeof(int32_t), hipMemcpyHostToDevice, st));
		for (int t = 0; t < nq && !only; t++)   // one [V_pad x 16] table per 16 query tokens (the traceback kernels restate their cells themselves)
			VK_HIP(vk_launch_table(c->d_tiles, c->d_qtile + (size_t)t * c->tile_bytes, (int32_t)c->n_tiles, c->nk32, c->tail, c->tile_bytes,
				c->d_table + t * table_stride, q->q_token_ids ? c->d_qids + t * 16 : nullptr, std::min(16, q->len_t - t * 16), c->desc.vocab_size, c->prec, st));
	}

	// ---- the fused scoring kernel ------------------------------------------
	// handles on one corpus take turns: this scoring kernel starts when the peer's has finished (its selection and
	// traceback then run beside this kernel); the wait is on the device, the host does not block
	VK_HIP(hipEventRecord(c->ev[5], st));
	if ((rc = vk_wait_peer_turn(c, st))) return rc;
	VK_HIP(hipEventRecord(c->ev[1], st));
	p.tiles = c->d_tiles; p.tok_id = c->d_tok_id; p.table = c->d_table; p.sent_start = c->d_sent_start; p.sent_end = c->d_sent_end;
	p.n_sent = (int32_t)n; p.layout = is_static ? VK_DEV_LAYOUT_STATIC : VK_DEV_LAYOUT_CONTEXTUAL;
	p.nk32 = c->nk32; p.tail = c->tail; p.tile_bytes = c->tile_bytes; p.prec = c->prec;
	p.qtile = c->d_qtile; p.len_t = q->len_t; p.locality = q->locality;
	p.ws = c->d_ws; p.wt = c->d_wt + 80; p.wt0 = c->d_wt;
	p.boost = q->boost ? c->d_boost : nullptr;
	p.scores = c->d_scores; p.raw = c->d_raw;
	p.ref_total = (float)q->len_t;
	if (q->tag_weights) {
		float total = 0.0f;
		for (int j = 0; j < q->len_t; j++) total += q->tag_weights[j];
		for (int j = 0; j < VK_FAST_QUERY_LEN; j++) {
			p.tw[j] = j < q->len_t ? q->tag_weights[j] : 0.0f;
			p.tpos[j] = j < q->len_t ? (int32_t)q->q_pos[j] : -1;
		}
		p.pos_s = c->d_pos;
		p.tw_keep = 1.0f - q->pos_mismatch_penalty;
		p.tw_threshold = q->similarity_threshold;
		p.ref_total = total;   // reference_score with max_similarity_for_t = t_pos_weights (slice/static.h:280-286)
	}
	// vocabulary transports with tag weights over the static layout: the cells upstream writes twice (static_vocab_fixup,
	// vk_common.hip.h) -- needs the (id, tag) keys of both sides: q_tags and vk_corpus_set_token_tags
	int32_t qkey_all[VK_MAX_QUERY_LEN];
	for (int j = 0; j < VK_MAX_QUERY_LEN; j++) qkey_all[j] = -1;
	const bool vocab_fix = is_static && q->algorithm == VK_ALG_RWMD && q->tag_weights && q->q_tags && q->q_token_ids && c->d_tag && c->d_pos;
	if (vocab_fix) {
		const size_t words = ((size_t)c->desc.vocab_size + 31) / 32 + 1;
		if (!c->d_qbits && (rc = alloc_t(c, &c->d_qbits, words))) return rc;
		std::vector<uint32_t> &bits = keep.vec<uint32_t>(words, 0u);
		for (int j = 0; j < q->len_t; j++) {
			const int32_t id = q->q_token_ids[j];
			if (id < 0 || id >= c->desc.vocab_size) continue;
			bits[(size_t)id >> 5] |= 1u << (id & 31);
			qkey_all[j] = id * 256 + ((int32_t)q->q_tags[j] & 255);
		}
		VK_HIP(hipMemcpyAsync(c->d_qbits, bits.data(), words * 4, hipMemcpyHostToDevice, st));
		VK_HIP(hipStreamSynchronize(st));   // `bits` leaves scope
		p.qid_bits = c->d_qbits; p.tag_s = c->d_tag; p.slices_overlap = c->overlapping ? 1 : 0;
		for (int j = 0; j < VK_FAST_QUERY_LEN; j++) p.qkey[j] = qkey_all[j];
	}
	VkWideParams wp{};
	// Winners of 65 .. 512 tokens under linear / affine gaps (scored by the fused kernel's long pass): their tracebacks on vk_doc_kernel's
	// sweep as well -- vk_flow_kernel fills such a matrix row by row in LDS (8,000 slices of 300 .. 512 tokens: 1.6 ms of a 2.8 ms query)
	const bool flow_doc = is_align && !wide && c->max_len > VK_FAST_SENT_LEN && q->len_t <= 16 && (p.gap_mode == 0 || p.gap_mode == 1) && !getenv("VK_NO_DOC_FLOW");
	// a query of 17 .. 64 tokens under linear / affine gaps: the long slices (65 tokens .. whole documents) and every winner's traceback on
	// the wave-wide skewed sweep (vk_docw_kernel); the slices of at most 64 tokens keep the multi-block kernel
	const bool docw = is_align && wide_score && q->len_t <= VK_MAX_QUERY_LEN && (p.gap_mode == 0 || p.gap_mode == 1) && !getenv("VK_NO_DOCW");
	// ... of 17 .. 32 tokens under general gaps whose table saturates within 126 tokens (wp.ws_tail): vk_docg_kernel
	auto docg_ok = [&]() { return is_align && wide_score && q->len_t <= 32 && c->max_len > VK_FAST_SENT_LEN && p.gap_mode == 2 && wp.ws_tail >= 1 && wp.ws_tail <= 126 && !getenv("VK_NO_DOCG"); };
	auto doc_ok = [&]() { return (doc_fast || flow_doc) && (q->algorithm == VK_ALG_RWMD ? wp.gap_mode == 4 && !getenv("VK_NO_DOC_RWMD")   // (the relaxed 1:1 form)
		: (wp.gap_mode == 0 || wp.gap_mode == 1 || (wp.gap_mode == 2 && wp.ws_tail >= 1 && wp.ws_tail <= 126 && !getenv("VK_NO_DOC_GENERAL")))); };
	// vk_wide_kernel: the state of a slice in LDS where that fits, else in global memory (one region per workgroup)
	// The scoring pass of vk_wide_kernel takes one wave per slice: its work list, longest first (wp.order).  A query of more than 16
	// tokens: every non-empty row of the slice table (the others carry no score: preset); at most 16 tokens: only the slices the fused
	// kernels leave to it -- those beyond VK_MAX_SENT_LEN, or, under general gaps, every slice of more than 64 tokens.
	auto wide_order = [&](int force = -1) -> int {
		const int which = force >= 0 ? force : wide_score ? 0 : (long_via_wide || rwmd_long_doc) ? 1 : 2;
		int32_t *&d_ord = which == 0 ? c->d_wide_order : which == 1 ? c->d_apart_order : c->d_xlong_order;
		int32_t &n_ord = which == 0 ? c->n_wide_order : which == 1 ? c->n_apart_order : c->n_xlong_order;
		if (n_ord < 0) {
			std::vector<int32_t> ord;
			if (which == 0) { for (int64_t e = 0; e < n; e++) if ((*c->h_end)[(size_t)e] > (*c->h_start)[(size_t)e]) ord.push_back((int32_t)e); }
			else if (which == 1 && c->h_apart) ord = *c->h_apart;
			else if (which == 2 && c->h_xlong) ord = *c->h_xlong;
			std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) {
				return (*c->h_end)[(size_t)a] - (*c->h_start)[(size_t)a] > (*c->h_end)[(size_t)b] - (*c->h_start)[(size_t)b]; });
			int rcw;
			if ((rcw = alloc_t(c, &d_ord, ord.size() + 1))) return rcw;
			VK_HIP(hipMemcpy(d_ord, ord.data(), ord.size() * 4, hipMemcpyHostToDevice));
			n_ord = (int32_t)ord.size();
		}
		wp.order = d_ord; wp.n_order = n_ord;
		if (which == 0) {
			VK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c->d_scores), (int)0xff800000u, (size_t)n, st));
			if (wp.raw) VK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(wp.raw), (int)0xff800000u, (size_t)n, st));
		}
		return VK_OK;
	};
	auto wide_state = [&](int flow_k, int force = -1) -> int {   // force: the work list (wide_order), whatever the query's own choice
		const bool flow = flow_k > 0;
		wp.scratch = nullptr; wp.scratch_stride = 0; wp.h_ring = 0; wp.order = nullptr; wp.n_order = 0;
		// (the pass over the long slices of a corpus: the ring form where the gap table saturates -- 9 KB of LDS per wave, not 35)
		const bool part = !flow && ((!wide_score && (xlong || long_via_wide || rwmd_long_doc)) || force >= 0);
		const bool want_ring = part && vk_wide_ring_rows(nq, wp.gap_mode, wp.ws_tail) > 0;
		if (!xlong && !want_ring && !(flow && (doc_ok() || docw || docg_ok())) && vk_wide_lds_demand(c->max_len, nq, wp.gap_mode, q->tag_weights != nullptr, flow) <= 160 * 1024) {
			if (part) {   // state in LDS, but still only the long slices
				int rcw = VK_OK;
				if ((rcw = wide_order(force))) return rcw;
			}
			return VK_OK;
		}
		wp.h_ring = vk_wide_ring_rows(nq, wp.gap_mode, wp.ws_tail);   // a saturated gap table: the column history is a ring in LDS
		size_t per = vk_wide_scratch_bytes(c->max_len, nq, wp.gap_mode, flow, wp.h_ring);
		if (flow && doc_ok()) per = std::max(per, vk_doc_scratch_bytes(c->max_len, wp.gap_mode));   // (vk_doc_kernel's records of a winner)
		if (flow && docw) per = std::max(per, vk_docw_scratch_bytes(c->max_len, nq));
		if (flow && docg_ok()) per = std::max(per, vk_docg_scratch_bytes(c->max_len));
		const size_t blocks = (size_t)vk_wide_gs_blocks(c->max_len, nq, wp.gap_mode, flow_k, n, wp.h_ring);
		const size_t need = per * blocks;
		if (need > ((size_t)16 << 30)) return fail(VK_ERR_UNSUPPORTED, "traceback state of this many slices this long exceeds 16 GiB of scratch");
		if (c->wide_scratch_cap < need) {
			if (c->d_wide_scratch) { VK_HIP(hipFree(c->d_wide_scratch)); c->d_wide_scratch = nullptr; c->wide_scratch_cap = 0; }
			int rcw;
			if ((rcw = alloc_t(c, &c->d_wide_scratch, need))) return rcw;
			c->wide_scratch_cap = need;
		}
		wp.scratch = c->d_wide_scratch; wp.scratch_stride = (int64_t)per;
		if (!flow && (xlong || part)) {
			int rcw = VK_OK;
			if ((rcw = wide_order(force))) return rcw;
		}
		return VK_OK;
	};
	if (wide || flow_doc || rwmd_long_doc) {
		wp.tiles = c->d_tiles; wp.tok_id = c->d_tok_id; wp.table = c->d_table; wp.table_stride = table_stride;
		wp.sent_start = c->d_sent_start; wp.sent_end = c->d_sent_end; wp.n_sent = (int32_t)n; wp.layout = p.layout;
		wp.nk32 = c->nk32; wp.tail = c->tail; wp.tile_bytes = c->tile_bytes; wp.prec = c->prec;
		wp.qtile = c->d_qtile; wp.nq = nq; wp.len_t = q->len_t; wp.locality = q->locality; wp.max_len = c->max_len;
		wp.gap_mode = (p.gap_mode == 3 || p.gap_mode == 6) ? 2 : p.gap_mode;   // (the fused kernels' register-history forms of general gaps)
		wp.rwmd_symmetric = p.rwmd_symmetric; wp.rwmd_normalize_bow = p.rwmd_normalize_bow;
		wp.gs = p.gs; wp.gt = p.gt; wp.a_s = p.a_s; wp.a_t = p.a_t; wp.open_s = p.open_s; wp.open_t = p.open_t;
		wp.ws = c->d_ws; wp.wt = c->d_wt; wp.wt0 = c->d_wt;
		if (wp.gap_mode == 2 && c->max_len >= 2) {   // the constant tail of w_s (a saturated table): from which k on
			int kt = c->max_len;
			while (kt > 1 && ws[(size_t)kt - 1] == ws[(size_t)c->max_len]) kt--;
			if (kt < c->max_len) wp.ws_tail = kt;
		}
		wp.pos_s = p.pos_s; wp.tag_s = p.tag_s; wp.qid_bits = p.qid_bits; wp.slices_overlap = p.slices_overlap; memcpy(wp.qkey, qkey_all, sizeof wp.qkey);
		wp.tw_keep = p.tw_keep; wp.tw_threshold = p.tw_threshold; wp.ref_total = p.ref_total;
		for (int j = 0; j < VK_MAX_QUERY_LEN; j++) {
			wp.tw[j] = (p.pos_s && j < q->len_t) ? q->tag_weights[j] : 0.0f;
			wp.tpos[j] = (p.pos_s && j < q->len_t) ? (int32_t)q->q_pos[j] : -1;
		}
		wp.boost = p.boost; wp.scores = c->d_scores; wp.raw = c->d_raw;
		wp.d = c->desc.d; wp.q_ids = is_static ? c->d_qids : nullptr;   // FLOW: canonical similarity rows (sim_canon)
		{   // as for the 16-column kernel: the aligner scores of all slices only if something reads them
			const bool exact_tr2 = q->algorithm == VK_ALG_WRD || (q->algorithm == VK_ALG_RWMD && q->wmd_full);
			if (((is_align && q->want_flow) || exact_tr2) && !(q->submatch_weight > 0.0f) && !getenv("VK_KEEP_RAW")) wp.raw = nullptr;
		}
		if (wide_score) {
		// 17..32 tokens with linear / affine gaps over a bf16 contextual corpus of short slices: the fused two-block kernel
		// (affine: the prefix-scan form of F needs open_t >= extend_t, as dp_affine)
		// (33..64 tokens: one slice per wave and four column blocks)
		const bool rwmd_inj = q->algorithm == VK_ALG_RWMD && (p.gap_mode == 4 || p.gap_mode == 7) && !q->wmd_full;
		const bool bound_pass = q->algorithm == VK_ALG_WRD || (q->algorithm == VK_ALG_RWMD && q->wmd_full);   // exact transport: stage 1
		// exact transport over a corpus with long slices: the multi-block kernel skips them (their groups are padded, vk_corpus.cpp),
		// vk_long_bound_kernel bounds them
		const bool long_apart = (bound_pass || p.gap_mode == 7) && c->n_long_groups > 0;
		const Score32Plan plan32 = score32_plan(c, q);
		const int wave_tiles = plan32.wave_tiles;
		// the multi-block kernel skips the slices of more than 64 tokens; a pass of their own scores them (vk_docw_kernel, else vk_wide_kernel)
		const bool docw_rwmd = rwmd_inj && p.gap_mode == 4 && c->max_len > VK_FAST_SENT_LEN && !getenv("VK_NO_DOCW");   // the relaxed 1:1 WMD: vk_docw_kernel's streaming arm
		const bool apart_route = (is_align || (rwmd_inj && p.gap_mode == 4)) && c->h_apart && !c->h_apart->empty() && !getenv("VK_NO_APART");
		const bool two_blocks = (is_align || rwmd_inj || bound_pass) && (long_apart || apart_route || (c->n_long_groups == 0 &&
			c->max_len <= VK_FAST_SENT_LEN)) && (apart_route || c->max_len <= VK_MAX_SENT_LEN) && (rwmd_inj || bound_pass || p.gap_mode == 0 || (p.gap_mode == 1 && p.a_t >= 0.0f) || (p.gap_mode == 2 && wide_sub)) &&
			plan32.fits && (bound_pass || p.gap_mode == 7 || !getenv("VK_NO_SCORE32"));
		if ((bound_pass || p.gap_mode == 7) && !two_blocks)
			return fail(VK_ERR_UNSUPPORTED, "exact transport / 1:n RWMD with a query of more than 16 tokens: the multi-block kernel does not fit this corpus (LDS)");
		if (p.gap_mode == 7) memcpy(wp.qmass, qmass_all, sizeof wp.qmass);
		if (only) {
		} else if (two_blocks) {
			if (bound_pass) {
				wp.gap_mode = 5;
				wp.mag = q->algorithm == VK_ALG_WRD ? c->d_mag : nullptr;
				memcpy(wp.qmass, qmass_all, sizeof wp.qmass);
				wp.wrd_raw_total = p.wrd_raw_total; wp.wmd_bound = q->algorithm == VK_ALG_WRD ? 0 : p.wmd_bound;
			}
			if (p.gap_mode == 2) { wp.gap_mode = (apart_route ? c->max_short_len : c->max_len) <= 32 ? 3 : 6; wp.wt = c->d_wt + 80; }   // register history of 32 / 64 rows, closure of w_t
			VK_HIP(vk_launch_score32(&wp, wave_tiles, st));
			if (long_apart && p.gap_mode == 7) VK_HIP(vk_launch_long_rwmd_fill(&wp, c->d_long_groups, c->n_long_groups, (int32_t)n, st));
			else if (long_apart) {
				VkWrdParams lw{};
				fill_transport(lw);
				lw.mag = q->algorithm == VK_ALG_WRD ? c->d_mag : nullptr;
				memcpy(lw.qmass, qmass_all, sizeof lw.qmass);
				lw.wrd_raw_total = p.wrd_raw_total; lw.wmd_bound = wp.wmd_bound;
				lw.group_list = c->d_long_groups; lw.n_list = c->n_long_groups; lw.n_entries = (int32_t)n;
				lw.scores = c->d_scores; lw.raw = c->d_raw; lw.boost = p.boost;
				VK_HIP(vk_launch_long_bound(&lw, st));
			}
			wp.gap_mode = p.gap_mode;                                        // the traceback kernel knows 0 / 1 / 2
			wp.wt = c->d_wt;                                                 // ... and walks the caller's table
			if (apart_route && (docw || docw_rwmd)) {
				if ((rc = wide_order(1))) return rc;
				if (wp.n_order > 0) VK_HIP(vk_launch_docw(&wp, 0, st));
			} else if (apart_route && docg_ok()) {
				if ((rc = wide_order(1))) return rc;
				if (wp.n_order > 0) VK_HIP(vk_launch_docg(&wp, 0, st));
			} else if (apart_route) {   // general gaps, relaxed 1:1 WMD: row by row, but only the slices apart
				if ((rc = wide_state(0, 1))) return rc;
				if (wp.n_order > 0) VK_HIP(vk_launch_wide(&wp, 0, st));
			}
		}
		else if (docw || docw_rwmd) {   // (the multi-block kernel does not fit this corpus: every slice on the sweep)
			if ((rc = wide_order(0))) return rc;
			if (wp.n_order > 0) VK_HIP(vk_launch_docw(&wp, 0, st));
		}
		else if (docg_ok()) {
			if ((rc = wide_order(0))) return rc;
			if (wp.n_order > 0) VK_HIP(vk_launch_docg(&wp, 0, st));
		}
		else {
			if ((rc = wide_state(0))) return rc;
			VK_HIP(vk_launch_wide(&wp, 0, st));
		}
		}
	}
	if (wide_score) {
	} else if (is_align && !is_static && q->len_t == 1 && c->uniform_len == 1 && q->locality == VK_LOCAL && !p.pos_s) {
		// span-embedding index: one vector per slice, one query vector -> the clipped cosine is the local alignment score.
		// The aligner scores are written only if something reads them: the traceback kernel restates those of the winners, and
		// without a booster the score IS the aligner score ((raw / 1) * 1)
		span_skip_raw = !(q->submatch_weight > 0.0f) && ((q->want_flow && is_align) || !p.boost || !out->raw_score);
		VkScoreParams ps = p;
		if (span_skip_raw) ps.raw = nullptr;
		if (!only) VK_HIP(vk_launch_span(&ps, st));
	} else {
	p.max_short_len = VK_FAST_SENT_LEN;
	// the aligner scores of all slices: read by the submatch bound and, without traceback, for the winners; with traceback the
	// flow kernel restates those of the winners
	// (exact transport: the solver states them); a second output array costs the stream 1 % (2.90 -> 2.87 ms per 1 M x 32 x 300-d)
	{
		const bool exact_tr2 = q->algorithm == VK_ALG_WRD || (q->algorithm == VK_ALG_RWMD && q->wmd_full);
		if (((is_align && q->want_flow) || exact_tr2) && !(q->submatch_weight > 0.0f) && !getenv("VK_KEEP_RAW")) p.raw = nullptr;
	}
	p.s_rows_per_wave = is_static ? (c->max_group_tokens + 15) / 16 * 16 : c->max_group_tiles * 16;
	p.h_rows = c->max_short_len + 1;
	const int lt = q->len_t <= 4 ? 4 : q->len_t <= 8 ? 8 : q->len_t <= 12 ? 12 : 16;   // strip rows hold the padded query columns (launch_score_lt)
	int lds_floats = p.s_rows_per_wave * lt + 16;
	if (p.gap_mode == 2) lds_floats += 4 * p.h_rows * 16;   // column history of dp_general
	p.m_rows = (c->max_short_len + 4) / 4 * 4;
	if (p.gap_mode == 7) lds_floats += 4 * p.m_rows;       // vocabulary masses of the 4 slices (static layout)
	p.lds_floats_per_wave = lds_floats;
	size_t smem = (size_t)lds_floats * 4 * 4;   // 4 waves per block
	// 300-d rows with general gaps (register history): the register form of the kernel takes 160 VGPRs, three waves per SIMD
	// leave 32, and the traceback kernel of the previous query has to wait until this kernel has drained (2.3 - 2.8 ms); with
	// the query tile in LDS the kernel takes 136 and runs at the same speed, the neighbours beside it.
	// Linear / affine gaps take 112 registers in the register form.  VK_QREG=1 / VK_QLDS=1 force one or the other.
	const bool reg_history = p.gap_mode == 3 || p.gap_mode == 6;
	if (!is_static && c->prec == 0 && c->nk32 == 10 && c->tail == 1)
		p.q_mode3 = getenv("VK_QLDS") ? 1 : getenv("VK_QREG") ? 0 : (reg_history ? 1 : 0);
	if (!is_static && c->prec == 1 && c->nk32 == 19 && !getenv("VK_NO_F32_SPECIAL")) p.q_mode3 = 1;   // fp32 rows at 300-d: MODE 4 (all 19 blocks of a tile in flight)
	const size_t qlds = (!is_static && ((c->prec == 0 && c->nk32 == 24 && c->tail == 0) || p.q_mode3)) ? (size_t)c->nk32 * 1024 : 0;   // MODE 3 / 4: query tile in LDS
	smem += qlds;
	// the generic contextual kernel (MODE 1: fp32 tiles, or a d without a specialised form) stages the query tile in LDS when it fits
	// beside the strips of at least two workgroups per CU
	if (!is_static && qlds == 0 && !(c->prec == 0 && c->nk32 == 10 && c->tail == 1)) {
		const size_t qb = ((size_t)c->tile_bytes + 1023) / 1024 * 1024;
		if (2 * (smem + qb) <= 160 * 1024 && !getenv("VK_NO_QLDS1")) { p.q_lds = (int32_t)qb; smem += qb; }
	}
	if (smem > 160 * 1024) return fail(VK_ERR_UNSUPPORTED, "LDS demand exceeds 160 KiB per workgroup");
	const int64_t n_groups = (n + 3) / 4;
	const int grid = (int)std::min<int64_t>((n_groups + 3) / 4, (int64_t)1 << 20);   // capped to residency by the launcher
	if (!only) VK_HIP(vk_launch_score(&p, grid, smem, st));
	if (c->n_long_groups > 0 && !only && !long_via_wide && !rwmd_long_doc) {
		// slices longer than VK_FAST_SENT_LEN: one per wave, one wave per workgroup, LDS strip for the longest;
		// general gaps take the LDS-history form (the four DPP rows share one history: only row 0 is active)
		VkScoreParams pl = p;
		pl.group_list = c->d_long_groups; pl.n_list = c->n_long_groups;
		if (pl.gap_mode == 3 || pl.gap_mode == 6) pl.gap_mode = 2;
		pl.s_rows_per_wave = is_static ? (c->long_group_tokens + 15) / 16 * 16 : c->long_group_tiles * 16;
		pl.h_rows = 0;
		int lf = pl.s_rows_per_wave * lt + 16;
		if (pl.gap_mode == 2) lf += (c->max_long_len + 1) * 16;
		pl.m_rows = 0;
		if (pl.gap_mode == 7) lf += (c->max_long_len + 4) / 4 * 4;
		pl.lds_floats_per_wave = lf;
		const size_t smem_l = (size_t)lf * 4 + qlds + (size_t)pl.q_lds;
		if (smem_l > 160 * 1024) return fail(VK_ERR_UNSUPPORTED, "LDS demand of the long-slice pass exceeds 160 KiB");
		VK_HIP(vk_launch_score(&pl, c->n_long_groups, smem_l, st));
	}
	if ((xlong || long_via_wide || rwmd_long_doc) && !only) {
		// slices beyond VK_MAX_SENT_LEN (whole documents; general gaps: beyond 64 tokens): one wave per slice, longest first
		if ((rc = wide_state(0))) return rc;
		// (linear / affine gaps: the skewed sweep of vk_doc_kernel -- no in-row dependency, a fifth of the time per row; round 4)
		if (wp.n_order > 0) VK_HIP(doc_ok() ? vk_launch_doc(&wp, 0, st) : vk_launch_wide(&wp, 0, st));
	}
	}

	const bool exact_transport = q->algorithm == VK_ALG_WRD || (q->algorithm == VK_ALG_RWMD && q->wmd_full);
	if (exact_transport && only) {
		// ---- only_slices: the listed slices solved exactly, in the caller's order (no bound pass ran, nothing is pruned)
		if (!c->entry_sent.empty() && c->sent_entry.empty()) {
			c->sent_entry.assign((size_t)c->desc.n_sentences, -1);
			for (int64_t e = 0; e < n; e++) if (c->entry_sent[(size_t)e] >= 0) c->sent_entry[(size_t)c->entry_sent[(size_t)e]] = (int32_t)e;
		}
		const int cnt = q->n_only;
		std::vector<uint64_t> &hk = keep.vec<uint64_t>((size_t)cnt);
		std::vector<int64_t> rows_idx((size_t)cnt);
		for (int i = 0; i < cnt; i++) {
			rows_idx[(size_t)i] = c->sent_entry.empty() ? q->only_slices[i] : (int64_t)c->sent_entry[(size_t)q->only_slices[i]];
			hk[(size_t)i] = (1ull << 32) | (uint64_t)(uint32_t)rows_idx[(size_t)i];
		}
		if (c->wrd_cap < (size_t)VK_MAX_MATCHES) {
			if (c->d_wrd_raw) { VK_HIP(hipFree(c->d_wrd_raw)); VK_HIP(hipFree(c->d_wrd_val)); c->d_wrd_raw = c->d_wrd_val = nullptr; }
			rc = alloc_t(c, &c->d_wrd_raw, (size_t)VK_MAX_MATCHES); if (rc) return rc;
			rc = alloc_t(c, &c->d_wrd_val, (size_t)VK_MAX_MATCHES); if (rc) return rc;
			c->wrd_cap = VK_MAX_MATCHES;
		}
		VK_HIP(hipMemcpyAsync(c->d_keys[0], hk.data(), hk.size() * 8, hipMemcpyHostToDevice, st));
		VkWrdParams w{};
		fill_transport(w);
		w.mass_mode = q->algorithm == VK_ALG_WRD ? 0 : (q->rwmd_normalize_bow ? 1 : 2);
		memcpy(w.qmass, qmass_all, sizeof w.qmass);
		w.raw_masses = (q->algorithm == VK_ALG_WRD && !q->wrd_normalize_magnitudes) ? 1 : 0;
		w.boost = p.boost; w.raw_out = c->d_wrd_raw; w.val_out = c->d_wrd_val; w.keys = c->d_keys[0];
		VK_HIP(hipMemsetAsync(c->d_wrd_raw, 0xff, (size_t)cnt * 4, st));   // NaN: a slice no solver takes (empty) stays marked
		VK_HIP(hipMemsetAsync(c->d_wrd_val, 0xff, (size_t)cnt * 4, st));
		VK_HIP(vk_launch_wrd_exact(&w, cnt, nullptr, st));
		if (c->max_len > VK_FAST_SENT_LEN) {
			if (w.nq > 1 && !c->d_wrdl_scratch) {
				if ((rc = alloc_t(c, &c->d_wrdl_scratch, (size_t)vk_wrd_long_blocks() * vk_wrd_long_scratch_bytes()))) return rc;
			}
			w.scratch = c->d_wrdl_scratch; w.scratch_stride = (int64_t)vk_wrd_long_scratch_bytes();
			VK_HIP(vk_launch_wrd_exact_long(&w, cnt, st));
		}
		std::vector<float> &vals = keep.vec<float>((size_t)cnt), &raws = keep.vec<float>((size_t)cnt);
		VK_HIP(hipMemcpyAsync(vals.data(), c->d_wrd_val, (size_t)cnt * 4, hipMemcpyDeviceToHost, st));
		VK_HIP(hipMemcpyAsync(raws.data(), c->d_wrd_raw, (size_t)cnt * 4, hipMemcpyDeviceToHost, st));
		VK_HIP(hipStreamSynchronize(st));
		if ((rc = transport_flows(rows_idx, true, w.qmass, w.mass_mode, w.raw_masses))) return rc;
		for (int i = 0; i < cnt; i++) {
			const bool empty = (*c->h_end)[(size_t)rows_idx[(size_t)i]] - (*c->h_start)[(size_t)rows_idx[(size_t)i]] < 1;
			out->score[i] = empty ? -INFINITY : vals[(size_t)i];
			out->sentence[i] = q->only_slices[i];
			if (out->raw_score) out->raw_score[i] = empty ? -INFINITY : raws[(size_t)i];
			if (out->mapping && out->edge_sim)
				for (int j = 0; j < q->len_t; j++) {
					out->mapping[i * (size_t)q->len_t + j] = -1;
					out->edge_sim[i * (size_t)q->len_t + j] = 0.0f;
				}
		}
		out->n_out = cnt;
		return VK_OK;
	}
	if (exact_transport) {
		// ---- stage 2: exact EMD on the candidates with the largest bounds, until the k-th best
		// exact score is above every remaining bound (then no unsolved sentence can enter)
		VK_HIP(hipEventRecord(c->ev[2], st));
		if (getenv("VK_WRD_TURNS")) c->ev2_recorded = true;   // experiment: bound passes take turns like the alignment kernels
		// (no turn-taking between handles here: ev2_recorded stays unset.  Two bound passes sharing the chip, each with its
		// long epilogue, fill each other's gaps: 336 M pairs/s with three handles against 302 M/s when they queue)
		// Round 1: the M largest bounds.  Its k-th best exact score theta prunes: every row whose bound is below
		// theta is out; all others are solved in one launch (round 2), which then fills the GPU instead of a
		// trickle of M-candidate rounds.
		const int M = 512;
		const size_t cap = ((size_t)((n + kTopkChunk - 1) / kTopkChunk) + 1) * VK_MAX_MATCHES;   // keys d_keys[0] holds
		if (c->wrd_cap < cap) {
			if (c->d_wrd_raw) { VK_HIP(hipFree(c->d_wrd_raw)); VK_HIP(hipFree(c->d_wrd_val)); c->d_wrd_raw = c->d_wrd_val = nullptr; }
			rc = alloc_t(c, &c->d_wrd_raw, cap); if (rc) return rc;
			rc = alloc_t(c, &c->d_wrd_val, cap); if (rc) return rc;
			c->wrd_cap = cap;
		}
		if (!c->d_counter) { rc = alloc_t(c, &c->d_counter, 4); if (rc) return rc; }
		struct Cand { float val, raw; int64_t g; };
		std::vector<Cand> best;
		std::vector<uint64_t> &keys = keep.vec<uint64_t>();
		std::vector<float> &vals = keep.vec<float>(), &raws = keep.vec<float>();
		VkWrdParams w{};
		fill_transport(w);
		w.mass_mode = q->algorithm == VK_ALG_WRD ? 0 : (q->rwmd_normalize_bow ? 1 : 2);
		memcpy(w.qmass, qmass_all, sizeof w.qmass);
		w.raw_masses = (q->algorithm == VK_ALG_WRD && !q->wrd_normalize_magnitudes) ? 1 : 0;
		w.boost = p.boost; w.raw_out = c->d_wrd_raw; w.val_out = c->d_wrd_val;
		// solves the `count` candidates whose keys sit at d_keys, merges them into `best`; returns the smallest bound among them
		auto solve = [&](const uint64_t *d_keys, int count, float *ub_min, int *n_cand_out) -> int {
			w.keys = d_keys;
			VK_HIP(vk_launch_wrd_exact(&w, count, c->d_scores, st));
			if (c->max_len > VK_FAST_SENT_LEN) {   // candidates of 65 .. 512 tokens
				if (w.nq > 1 && !c->d_wrdl_scratch) {
					int rc3 = alloc_t(c, &c->d_wrdl_scratch, (size_t)vk_wrd_long_blocks() * vk_wrd_long_scratch_bytes());
					if (rc3) return rc3;
				}
				w.scratch = c->d_wrdl_scratch; w.scratch_stride = (int64_t)vk_wrd_long_scratch_bytes();
				VK_HIP(vk_launch_wrd_exact_long(&w, count, st));
			}
			keys.resize((size_t)count); vals.resize((size_t)count); raws.resize((size_t)count);
			VK_HIP(hipMemcpyAsync(keys.data(), d_keys, (size_t)count * 8, hipMemcpyDeviceToHost, st));
			VK_HIP(hipMemcpyAsync(vals.data(), c->d_wrd_val, (size_t)count * 4, hipMemcpyDeviceToHost, st));
			VK_HIP(hipMemcpyAsync(raws.data(), c->d_wrd_raw, (size_t)count * 4, hipMemcpyDeviceToHost, st));
			VK_HIP(hipStreamSynchronize(st));
			int n_cand = 0;
			float ub = INFINITY;
			for (int i = 0; i < count; i++) {
				if (keys[(size_t)i] == 0) break;
				n_cand++;
				const uint32_t ob = (uint32_t)(keys[(size_t)i] >> 32);
				const uint32_t bits = (ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob;
				float u;
				memcpy(&u, &bits, 4);
				ub = std::min(ub, u);
				if (vals[(size_t)i] > q->min_score)
					best.push_back({vals[(size_t)i], raws[(size_t)i], (int64_t)(uint32_t)(keys[(size_t)i] & 0xffffffffu)});
			}
			const auto better = [](const Cand &a, const Cand &b) {
				if (a.val != b.val) return a.val > b.val;
				return a.g > b.g;
			};
			if ((int)best.size() > k) {
				std::partial_sort(best.begin(), best.begin() + k, best.end(), better);
				best.resize((size_t)k);
			} else std::sort(best.begin(), best.end(), better);
			*ub_min = ub;
			*n_cand_out = n_cand;
			return VK_OK;
		};
		{
			int nb = 0, cur = 0;
			VK_HIP(vk_launch_topk_scores(c->d_scores, n, q->min_score, M, c->d_keys[0], &nb, st));
			while (nb > 1) {
				const int64_t nkeys = (int64_t)nb * M;
				VK_HIP(vk_launch_topk_keys(c->d_keys[cur], nkeys, M, c->d_keys[1 - cur], &nb, st));
				cur = 1 - cur;
			}
			float ub_last = INFINITY;
			int n_cand = 0;
			if ((rc = solve(c->d_keys[cur], M, &ub_last, &n_cand))) return rc;
			bool done = n_cand < M || ((int)best.size() == k && best.back().val > ub_last);
			while (!done) {
				const float theta = (int)best.size() == k ? best.back().val : -INFINITY;
				VK_HIP(vk_launch_select_ge(c->d_scores, n, theta, q->min_score, c->d_keys[0], c->d_counter, (uint32_t)cap, st));
				uint32_t &count = *keep.array<uint32_t>(1);
				VK_HIP(hipMemcpyAsync(&count, c->d_counter, 4, hipMemcpyDeviceToHost, st));
				VK_HIP(hipStreamSynchronize(st));
				if (count == 0) break;
				if (getenv("VK_DEBUG_CANDIDATES")) fprintf(stderr, "[vk] exact transport: round 2 solves %u candidates (theta %.6f, n %lld)\n", count, theta, (long long)n);
				const int take = (int)std::min<size_t>(count, cap);
				if ((rc = solve(c->d_keys[0], take, &ub_last, &n_cand))) return rc;
				done = (size_t)count <= cap;   // every row that could still enter has been solved
			}
		}
		VK_HIP(hipEventRecord(c->ev[3], st));
		{
			std::vector<int64_t> rows_idx;
			for (const Cand &b : best) rows_idx.push_back(b.g);
			if ((rc = transport_flows(rows_idx, true, w.qmass, w.mass_mode, w.raw_masses))) return rc;
		}
		VK_HIP(hipEventRecord(c->ev[4], st));
		VK_HIP(hipStreamSynchronize(st));
		for (size_t i = 0; i < best.size(); i++) {
			out->score[i] = best[i].val;
			out->sentence[i] = sentence_of(best[i].g);
			if (out->raw_score) out->raw_score[i] = best[i].raw;
			if (q->want_flow && out->mapping && out->edge_sim)
				for (int j = 0; j < q->len_t; j++) {
					out->mapping[i * (size_t)q->len_t + j] = -1;
					out->edge_sim[i * (size_t)q->len_t + j] = 0.0f;
				}
		}
		out->n_out = (int)best.size();
		float ms = 0;
		vk_timings t{};
		if (hipEventElapsedTime(&ms, c->ev[0], c->ev[5]) == hipSuccess) t.prepare_ms = ms;
		if (hipEventElapsedTime(&ms, c->ev[5], c->ev[1]) == hipSuccess) t.queue_ms = ms;
		if (hipEventElapsedTime(&ms, c->ev[1], c->ev[2]) == hipSuccess) t.score_ms = ms;
		if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) t.topk_ms = ms;
		if (hipEventElapsedTime(&ms, c->ev[0], c->ev[4]) == hipSuccess) t.total_ms = ms - t.queue_ms;
		c->last = t;
		return VK_OK;
	}

	// ---- flow (traceback) of `count` slices named by device keys: narrow or wide kernel
	const int ostride = (wide || flow_doc) ? 64 : 16;   // row stride of the mapping / edge_sim device arrays
	auto launch_flow = [&](const uint64_t *d_keys, int count) -> int {
		if (wide || flow_doc) {
			wp.keys = d_keys; wp.raw_out = c->d_out_raw; wp.mapping = c->d_out_map; wp.edge_sim = c->d_out_sim;
			int rcw = wide_state(count);
			if (rcw) return rcw;
			wp.dp_rows = nullptr; wp.dp_rows_len = 0;
			if (xlong || doc_ok() || docw || docg_ok()) {
				// Long winners: their similarities (canonical arithmetic, tag weights applied) restated beforehand by one wave per 16
				// tokens, so that the serial sweep of a winner is its recurrence alone (5,000 tokens: 8.4 ms of a 12 ms query were the
				// sweep restating 313 tiles one after the other; 3.6 ms since).  Within 2 GiB; else the sweep restates them itself.
				const int R = (c->max_len + 63) / 64 * 64, Wq = 16 * nq;
				const size_t need = (size_t)count * R * Wq;
				if (need * 4 <= ((size_t)2 << 30)) {
					if (c->rows_cap < need) {
						if (c->d_rows_out) { VK_HIP(hipFree(c->d_rows_out)); VK_HIP(hipFree(c->d_plan_out)); c->d_rows_out = c->d_plan_out = nullptr; c->rows_cap = 0; }
						if ((rcw = alloc_t(c, &c->d_rows_out, need))) return rcw;
						if ((rcw = alloc_t(c, &c->d_plan_out, need))) return rcw;
						c->rows_cap = need;
					}
					VkWrdParams w{};
					fill_transport(w);
					w.keys = d_keys; w.rows_out = c->d_rows_out; w.rows_len = R;
					VK_HIP(hipMemsetAsync(c->d_rows_out, 0, need * 4, st));
					VK_HIP(vk_launch_canon_rows(&w, count, (c->max_len + 15) / 16 + 1, st));
					wp.dp_rows = c->d_rows_out; wp.dp_rows_len = R;
				}
			}
			if (docw && wp.dp_rows && wp.scratch && wp.scratch_stride >= (int64_t)vk_docw_scratch_bytes(c->max_len, nq)) VK_HIP(vk_launch_docw(&wp, count, st));
			else if (docg_ok() && wp.dp_rows && wp.scratch && wp.scratch_stride >= (int64_t)vk_docg_scratch_bytes(c->max_len)) VK_HIP(vk_launch_docg(&wp, count, st));
			else if (doc_ok() && wp.dp_rows && wp.scratch && wp.scratch_stride >= (int64_t)vk_doc_scratch_bytes(c->max_len, wp.gap_mode)) VK_HIP(vk_launch_doc(&wp, count, st));
			else VK_HIP(vk_launch_wide(&wp, count, st));
			return VK_OK;
		}
		VkFlowParams f{};
		f.tiles = c->d_tiles; f.tok_id = c->d_tok_id; f.table = c->d_table; f.sent_start = c->d_sent_start; f.sent_end = c->d_sent_end;
		f.layout = p.layout; f.nk32 = c->nk32; f.tail = c->tail; f.tile_bytes = c->tile_bytes; f.prec = c->prec;
		f.qtile = c->d_qtile; f.len_t = q->len_t; f.locality = q->locality; f.gap_mode = (p.gap_mode == 3 || p.gap_mode == 6) ? 2 : p.gap_mode;
		f.max_len = c->max_len;
		f.gs = p.gs; f.gt = p.gt; f.a_s = p.a_s; f.a_t = p.a_t; f.open_s = p.open_s; f.open_t = p.open_t;
		f.ws = c->d_ws; f.wt = c->d_wt;
		f.pos_s = p.pos_s; f.tw_keep = p.tw_keep; f.tw_threshold = p.tw_threshold;
		memcpy(f.tw, p.tw, sizeof f.tw);
		memcpy(f.tpos, p.tpos, sizeof f.tpos);
		f.d = c->desc.d; f.q_ids = is_static ? c->d_qids : nullptr;   // canonical similarity rows (sim_canon)
		f.keys = d_keys; f.raw_out = c->d_out_raw; f.mapping = c->d_out_map; f.edge_sim = c->d_out_sim;
		VK_HIP(vk_launch_flow(&f, count, st));
		return VK_OK;
	};

	if (is_align && q->submatch_weight != 0.0f && !only) {
		// ---- submatch_weight: bound from raw, then exact scores of the candidates from their tracebacks, until the
		// k-th best exact score is above every remaining bound (vk_submatch_bound_kernel)
		VK_HIP(hipEventRecord(c->ev[2], st));
		c->ev2_recorded = true;
		const float wsub = q->submatch_weight, total = p.ref_total;
		const float m_star = total * (1.0f - powf(1.0f / (wsub + 1.0f), 1.0f / wsub));
		VK_HIP(vk_launch_submatch_bound(c->d_raw, p.boost, n, total, wsub, m_star, c->d_scores, st));
		const int M = 512;
		struct Cand { float val, raw; int64_t row; std::vector<int16_t> map; std::vector<float> sim; };
		std::vector<Cand> best;
		std::vector<uint64_t> &keys = keep.vec<uint64_t>((size_t)M);
		std::vector<float> &raws = keep.vec<float>((size_t)M), &sims = keep.vec<float>((size_t)M * ostride);
		std::vector<int16_t> &maps = keep.vec<int16_t>((size_t)M * ostride);
		for (;;) {
			int nb = 0, cur = 0;
			VK_HIP(vk_launch_topk_scores(c->d_scores, n, q->min_score, M, c->d_keys[0], &nb, st));
			while (nb > 1) {
				const int64_t nkeys = (int64_t)nb * M;
				VK_HIP(vk_launch_topk_keys(c->d_keys[cur], nkeys, M, c->d_keys[1 - cur], &nb, st));
				cur = 1 - cur;
			}
			if ((rc = launch_flow(c->d_keys[cur], M))) return rc;
			VK_HIP(vk_launch_mark(c->d_keys[cur], M, c->d_scores, st));
			VK_HIP(hipMemcpyAsync(keys.data(), c->d_keys[cur], (size_t)M * 8, hipMemcpyDeviceToHost, st));
			VK_HIP(hipMemcpyAsync(raws.data(), c->d_out_raw, (size_t)M * 4, hipMemcpyDeviceToHost, st));
			VK_HIP(hipMemcpyAsync(maps.data(), c->d_out_map, maps.size() * 2, hipMemcpyDeviceToHost, st));
			VK_HIP(hipMemcpyAsync(sims.data(), c->d_out_sim, sims.size() * 4, hipMemcpyDeviceToHost, st));
			VK_HIP(hipStreamSynchronize(st));
			int n_cand = 0;
			float ub_last = INFINITY;
			for (int i = 0; i < M; i++) {
				if (keys[(size_t)i] == 0) break;
				n_cand++;
				const uint32_t ob = (uint32_t)(keys[(size_t)i] >> 32);
				const uint32_t bits = (ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob;
				memcpy(&ub_last, &bits, 4);
				const int64_t row = (int64_t)(uint32_t)(keys[(size_t)i] & 0xffffffffu);
				// reference_score (metric/alignment.h:84-106) with the matched weight of this traceback, in float as upstream
				float matched = 0.0f;
				for (int j = 0; j < q->len_t; j++)
					if (maps[(size_t)i * ostride + j] >= 0) matched += (q->tag_weights && is_align) ? q->tag_weights[j] : 1.0f;
				const float uw = powf((total - matched) / total, wsub);
				const float ref = matched + uw * (total - matched);
				const float boost = q->boost ? q->boost[sentence_of(row)] : 1.0f;
				const float val = (raws[(size_t)i] / ref) * boost;
				if (val > q->min_score) {
					Cand cd{val, raws[(size_t)i], row, {}, {}};
					cd.map.assign(maps.begin() + (size_t)i * ostride, maps.begin() + (size_t)i * ostride + q->len_t);
					cd.sim.assign(sims.begin() + (size_t)i * ostride, sims.begin() + (size_t)i * ostride + q->len_t);
					best.push_back(std::move(cd));
				}
			}
			std::sort(best.begin(), best.end(), [](const Cand &a, const Cand &b) {
				if (a.val != b.val) return a.val > b.val;
				return a.row > b.row;
			});
			if ((int)best.size() > k) best.resize((size_t)k);
			if (n_cand < M) break;
			if ((int)best.size() == k && best.back().val > ub_last) break;
		}
		VK_HIP(hipEventRecord(c->ev[3], st));
		VK_HIP(hipEventRecord(c->ev[4], st));
		VK_HIP(hipStreamSynchronize(st));
		for (size_t i = 0; i < best.size(); i++) {
			out->score[i] = best[i].val;
			out->sentence[i] = sentence_of(best[i].row);
			if (out->raw_score) out->raw_score[i] = best[i].raw;
			if (q->want_flow && out->mapping && out->edge_sim)
				for (int j = 0; j < q->len_t; j++) {
					out->mapping[i * (size_t)q->len_t + j] = best[i].map[(size_t)j];
					out->edge_sim[i * (size_t)q->len_t + j] = best[i].sim[(size_t)j];
				}
		}
		out->n_out = (int)best.size();
		if (out->sim_rows && !best.empty()) {   // similarity rows of the winners on request (debug hook)
			std::vector<int64_t> rows_idx;
			for (const Cand &b : best) rows_idx.push_back(b.row);
			float no_mass[VK_MAX_QUERY_LEN] = {0};
			if ((rc = transport_flows(rows_idx, false, no_mass, 0, 0))) return rc;
		}
		float ms = 0;
		vk_timings t{};
		if (hipEventElapsedTime(&ms, c->ev[0], c->ev[5]) == hipSuccess) t.prepare_ms = ms;
		if (hipEventElapsedTime(&ms, c->ev[5], c->ev[1]) == hipSuccess) t.queue_ms = ms;
		if (hipEventElapsedTime(&ms, c->ev[1], c->ev[2]) == hipSuccess) t.score_ms = ms;
		if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) t.topk_ms = ms;
		if (hipEventElapsedTime(&ms, c->ev[0], c->ev[4]) == hipSuccess) t.total_ms = ms - t.queue_ms;
		c->last = t;
		return VK_OK;
	}

	// ---- bounded result set -------------------------------------------------
	VK_HIP(hipEventRecord(c->ev[2], st));
	c->ev2_recorded = true;
	const bool rows_on_request = out->sim_rows != nullptr;   // alignments: similarity rows of the winners only on request (debug hook)
	// Alignments with traceback: the scores of the scoring pass rest on MFMA cosines and differ from the oracle's in the last
	// bits; the flow kernel restates every winner in the canonical arithmetic (aligner score, mapping, edge similarities: the
	// oracle's, bit for bit).  So that the result SET is the oracle's too, a few runners-up are retraced with the winners
	// (kCanonMargin more slices; the floor of the selection is lowered by the rounding slack likewise) and the k best canonical
	// scores are kept: exact unless more than kCanonMargin slices sit within rounding (~2e-6) of the k-th score.
	const bool do_flow = q->want_flow && is_align;
	// Relaxed word mover's distance: likewise -- the rows of the winners come back in the canonical arithmetic (vk_rows_kernel) and
	// the host restates each winner's score from them in the reference's order of operations (vk_transport_host.h): the scores of the
	// result set are the oracle's floats, whichever kernel ranked the slices (per query, batched GEMM, a shard of the corpus).
	const bool canon_tr = q->algorithm == VK_ALG_RWMD && !q->wmd_full && q->want_flow && out->sim_rows != nullptr;
	constexpr int kCanonMargin = 8;
	// (57 .. 64 matches: the margin takes the selection to the k > 64 path; beyond VK_MAX_MATCHES: every score is sorted, never more winners than rows)
	const int kk = only ? q->n_only : (int)std::min<int64_t>(!(do_flow || canon_tr) ? k : (k <= VK_MAX_MATCHES ? std::min(k + kCanonMargin, VK_MAX_MATCHES) : k + kCanonMargin), std::max<int64_t>(n, 1));
	const uint64_t *d_sel = nullptr;   // the selected keys on the device, best first
	if ((size_t)kk > c->out_cap) {   // the winners' device arrays: grown to this result set
		for (void *ptr : {(void *)c->d_out_raw, (void *)c->d_out_sim, (void *)c->d_out_map}) if (ptr) VK_HIP(hipFree(ptr));
		c->d_out_raw = nullptr; c->d_out_sim = nullptr; c->d_out_map = nullptr; c->out_cap = 0;
		if ((rc = alloc_t(c, &c->d_out_raw, (size_t)kk))) return rc;
		if ((rc = alloc_t(c, &c->d_out_sim, (size_t)kk * 64))) return rc;
		if ((rc = alloc_t(c, &c->d_out_map, (size_t)kk * 64))) return rc;
		c->out_cap = (size_t)kk;
	}
	const float sel_floor = (do_flow || canon_tr) ? q->min_score - 1e-5f * std::max(1.0f, std::fabs(q->min_score)) : q->min_score;
	int cur = 0;
	if (only) {
		// keys of the listed slices, in the caller's order (rows of the slice table: long slices sit in padded groups)
		if (!c->entry_sent.empty() && c->sent_entry.empty()) {
			c->sent_entry.assign((size_t)c->desc.n_sentences, -1);
			for (int64_t e = 0; e < n; e++) if (c->entry_sent[(size_t)e] >= 0) c->sent_entry[(size_t)c->entry_sent[(size_t)e]] = (int32_t)e;
		}
		std::vector<uint64_t> &hk = keep.vec<uint64_t>((size_t)q->n_only);
		for (int i = 0; i < q->n_only; i++) {
			const int64_t row = c->sent_entry.empty() ? q->only_slices[i] : (int64_t)c->sent_entry[(size_t)q->only_slices[i]];
			hk[(size_t)i] = (1ull << 32) | (uint64_t)(uint32_t)row;
		}
		VK_HIP(hipMemcpyAsync(c->d_keys[0], hk.data(), hk.size() * 8, hipMemcpyHostToDevice, st));
		VK_HIP(hipStreamSynchronize(st));   // `hk` leaves scope
	} else if (kk > VK_MAX_MATCHES) {
		// more matches than the block selection keeps per 2,048 keys: the keys of all n rows, sorted
		if (c->sort_cap < (size_t)n) {
			for (auto &b : c->d_sort) if (b) { VK_HIP(hipFree(b)); b = nullptr; }
			c->sort_cap = 0;
			if ((rc = alloc_t(c, &c->d_sort[0], (size_t)n + 64))) return rc;
			if ((rc = alloc_t(c, &c->d_sort[1], (size_t)n + 64))) return rc;
			c->sort_cap = (size_t)n;
		}
		size_t temp_bytes = 0;
		uint64_t *sorted = nullptr;
		VK_HIP(vk_launch_sort_all(nullptr, n, sel_floor, c->d_sort[0], c->d_sort[1], nullptr, &temp_bytes, &sorted, st));
		if (c->sort_temp_cap < temp_bytes) {
			if (c->d_sort_temp) { VK_HIP(hipFree(c->d_sort_temp)); c->d_sort_temp = nullptr; c->sort_temp_cap = 0; }
			if ((rc = alloc(c, &c->d_sort_temp, temp_bytes))) return rc;
			c->sort_temp_cap = temp_bytes;
		}
		VK_HIP(vk_launch_sort_all(c->d_scores, n, sel_floor, c->d_sort[0], c->d_sort[1], c->d_sort_temp, &temp_bytes, &sorted, st));
		d_sel = sorted;
	} else if (kk <= 64) {
		// wave-streaming selection: n -> ceil(n/4096) * k keys -> ... -> k keys
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

	// ---- flow of the winners ------------------------------------------------
	if (!d_sel) d_sel = c->d_keys[cur];
	VK_HIP(hipEventRecord(c->ev[3], st));
	if (do_flow && (rc = launch_flow(d_sel, kk))) return rc;
	VK_HIP(hipEventRecord(c->ev[4], st));

	// ---- results to host ------------------------------------------------------
	std::vector<uint64_t> &keys = keep.vec<uint64_t>((size_t)kk);
	std::vector<float> &raw = keep.vec<float>((size_t)kk), &sim = keep.vec<float>((size_t)kk * ostride);
	std::vector<int16_t> &map = keep.vec<int16_t>((size_t)kk * ostride);
	VK_HIP(hipMemcpyAsync(keys.data(), d_sel, (size_t)kk * 8, hipMemcpyDeviceToHost, st));
	if (do_flow) {
		VK_HIP(hipMemcpyAsync(raw.data(), c->d_out_raw, (size_t)kk * 4, hipMemcpyDeviceToHost, st));
		VK_HIP(hipMemcpyAsync(map.data(), c->d_out_map, map.size() * 2, hipMemcpyDeviceToHost, st));
		VK_HIP(hipMemcpyAsync(sim.data(), c->d_out_sim, sim.size() * 4, hipMemcpyDeviceToHost, st));
	}
	VK_HIP(hipStreamSynchronize(st));

	int n_sel = 0;
	for (int i = 0; i < kk; i++) {
		if (keys[(size_t)i] == 0) break;
		n_sel++;
	}
	// order[i]: position among the selected slices of the i-th result
	std::vector<int> order((size_t)n_sel);
	std::vector<float> val((size_t)std::max(n_sel, 1));
	for (int i = 0; i < n_sel; i++) {
		order[(size_t)i] = i;
		const uint32_t ob = (uint32_t)(keys[(size_t)i] >> 32);
		const uint32_t bits = (ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob;
		memcpy(&val[(size_t)i], &bits, 4);
	}
	int n_out = n_sel;
	if (do_flow) {
		// Score of a winner from its canonical aligner score (match/match.h:295-307, reference_score metric/alignment.h:84-106),
		// operation by operation as the oracle's vko_score: matched weight of this traceback, pow(., submatch_weight = 0) = 1
		const float total = p.ref_total;
		for (int i = 0; i < n_sel; i++) {
			float matched = 0.0f;
			for (int j = 0; j < q->len_t; j++)
				if (map[(size_t)i * ostride + j] >= 0) matched += q->tag_weights ? q->tag_weights[j] : 1.0f;
			const float uw = powf((total - matched) / total, only ? q->submatch_weight : 0.0f);   // (searches with a submatch weight take the candidate rounds above)
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
	}
	// canon_tr: similarity rows of every selected slice -- in the handle's pinned staging (a std::vector made the copy of a document
	// corpus's winners, 12 MB for 18 x 5,056 rows x 32 columns, go through the runtime's bounce buffers: 2 - 4 ms of a 5.7 ms query)
	float *rows_all = nullptr;
	const int rows_R = out->rows_per_winner > 0 ? out->rows_per_winner : VK_FAST_SENT_LEN, rows_W = 16 * ((q->len_t + 15) / 16);
	if (canon_tr && n_sel > 0) {
		const size_t rows_bytes = (size_t)n_sel * rows_R * rows_W * 4;
		if (c->h_brows_cap < rows_bytes) {
			if (c->h_brows) { VK_HIP(hipHostFree(c->h_brows)); c->h_brows = nullptr; c->h_brows_cap = 0; }
			VK_HIP(hipHostMalloc((void **)&c->h_brows, rows_bytes, hipHostMallocDefault));
			c->h_brows_cap = rows_bytes;
		}
		rows_all = c->h_brows;
		std::vector<int64_t> rows_idx;
		for (int i = 0; i < n_sel; i++) rows_idx.push_back((int64_t)(uint32_t)(keys[(size_t)i] & 0xffffffffu));
		float no_mass[VK_MAX_QUERY_LEN] = {0};
		if ((rc = transport_flows(rows_idx, false, no_mass, 0, 0, rows_all))) return rc;
		// vocabulary keys (static layout): token ids, or (id, tag) pairs when the similarity is tag-weighted (alignment/bow.h:106-127, 150-176)
		const bool vocab = is_static && q->q_token_ids && c->h_tok;
		const bool tagged = vocab && q->tag_weights && q->q_tags && c->h_tag;
		std::vector<int32_t> key_t((size_t)q->len_t), key_s;
		for (int j = 0; vocab && j < q->len_t; j++) key_t[(size_t)j] = tagged ? q->q_token_ids[j] * 256 + (int32_t)(uint8_t)q->q_tags[j] : q->q_token_ids[j];
		float total = (float)q->len_t;
		if (q->tag_weights) {
			total = 0.0f;
			for (int j = 0; j < q->len_t; j++) total += q->tag_weights[j];
		}
		for (int i = 0; i < n_sel; i++) {
			const int64_t row = rows_idx[(size_t)i];
			const int32_t t_a = (*c->h_start)[(size_t)row], len_s = (*c->h_end)[(size_t)row] - t_a;
			const float boost = q->boost ? q->boost[sentence_of(row)] : 1.0f;
			if (len_s < 1 || len_s > rows_R) {   // no rows for this one (longer than the caller's room): it keeps the scoring pass's value
				if (only) { raw[(size_t)i] = val[(size_t)i] = len_s < 1 ? -INFINITY : NAN; continue; }   // (no scoring pass ran: an empty slice has no score, a longer one cannot be stated)
				VK_HIP(hipMemcpy(&raw[(size_t)i], c->d_raw + row, 4, hipMemcpyDeviceToHost));
				continue;
			}
			if (vocab) {
				key_s.resize((size_t)len_s);
				for (int u = 0; u < len_s; u++)
					key_s[(size_t)u] = tagged ? (*c->h_tok)[(size_t)(t_a + u)] * 256 + (int32_t)(uint8_t)(*c->h_tag)[(size_t)(t_a + u)] : (*c->h_tok)[(size_t)(t_a + u)];
			}
			raw[(size_t)i] = vk_host::rwmd_from_rows(rows_all + (size_t)i * rows_R * rows_W, rows_W, len_s, q->len_t,
				vocab ? key_s.data() : nullptr, vocab ? key_t.data() : nullptr, q->rwmd_injective != 0, q->rwmd_symmetric != 0, q->rwmd_normalize_bow != 0);
			val[(size_t)i] = (raw[(size_t)i] / total) * boost;   // reference_score with every query token matched: the sum of the weights (match.h:165-176)
		}
		if (!only) {
			order.erase(std::remove_if(order.begin(), order.end(), [&](int i) { return !(val[(size_t)i] > q->min_score); }), order.end());
			std::sort(order.begin(), order.end(), [&](int a, int b) {
				if (val[(size_t)a] != val[(size_t)b]) return val[(size_t)a] > val[(size_t)b];
				return (uint32_t)(keys[(size_t)a] & 0xffffffffu) > (uint32_t)(keys[(size_t)b] & 0xffffffffu);
			});
		}
		n_out = std::min((int)order.size(), only ? q->n_only : k);
	}
	std::vector<float> &raw_sel = keep.vec<float>((size_t)std::max(n_out, 1));
	if (!do_flow && !(canon_tr && n_sel > 0) && out->raw_score && n_out > 0) {
		// gather the aligner scores of the winners (a large result set: the whole array in one copy, gathered here)
		if (n_out > 256 && !span_skip_raw) {
			std::vector<float> &all_raw = keep.vec<float>((size_t)n);
			VK_HIP(hipMemcpyAsync(all_raw.data(), c->d_raw, (size_t)n * 4, hipMemcpyDeviceToHost, st));
			VK_HIP(hipStreamSynchronize(st));
			for (int i = 0; i < n_out; i++) raw_sel[(size_t)i] = all_raw[(size_t)(uint32_t)(keys[(size_t)i] & 0xffffffffu)];
		} else for (int i = 0; i < n_out; i++) {
			const int64_t g = (int64_t)(uint32_t)(keys[(size_t)i] & 0xffffffffu);
			if (!span_skip_raw) VK_HIP(hipMemcpyAsync(&raw_sel[(size_t)i], c->d_raw + g, 4, hipMemcpyDeviceToHost, st));
		}
		VK_HIP(hipStreamSynchronize(st));
	}
	for (int i = 0; i < n_out; i++) {
		const int src = order[(size_t)i];
		const uint64_t key = keys[(size_t)src];
		const float s = val[(size_t)src];
		out->score[i] = s;
		out->sentence[i] = sentence_of((int64_t)(uint32_t)(key & 0xffffffffu));
		if (out->raw_score) out->raw_score[i] = (do_flow || canon_tr) ? raw[(size_t)src] : span_skip_raw ? s : raw_sel[(size_t)i];
		if (do_flow) {
			for (int j = 0; j < q->len_t; j++) {
				out->mapping[(size_t)i * q->len_t + j] = map[(size_t)src * ostride + j];
				out->edge_sim[(size_t)i * q->len_t + j] = sim[(size_t)src * ostride + j];
			}
		} else if (q->want_flow && out->mapping && out->edge_sim) {
			// transport flows of the winners (SparseFlow / DenseFlow) are not produced yet
			for (int j = 0; j < q->len_t; j++) {
				out->mapping[(size_t)i * q->len_t + j] = -1;
				out->edge_sim[(size_t)i * q->len_t + j] = 0.0f;
			}
		}
	}
	out->n_out = n_out;
	c->have_scores = !only;
	if (canon_tr && n_out > 0) {
		for (int i = 0; i < n_out; i++)   // the rows of the winners, in their final order
			memcpy(out->sim_rows + (size_t)i * rows_R * rows_W, rows_all + (size_t)order[(size_t)i] * rows_R * rows_W, (size_t)rows_R * rows_W * 4);
	} else if ((q->algorithm == VK_ALG_RWMD || (is_align && rows_on_request)) && n_out > 0) {
		std::vector<int64_t> rows_idx;
		for (int i = 0; i < n_out; i++) rows_idx.push_back((int64_t)(uint32_t)(keys[(size_t)order[(size_t)i]] & 0xffffffffu));
		float no_mass[VK_MAX_QUERY_LEN] = {0};
		if ((rc = transport_flows(rows_idx, false, no_mass, 0, 0))) return rc;
	}

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

extern "C" {

int vk_query(vk_corpus_t *c, const vk_query_desc *q, vk_topk_out *out) {
	const int rc = vk_validate_query(c, q, out);
	if (rc) return rc;
	if (q->abort && *q->abort) { out->n_out = 0; return fail(VK_ERR_ABORTED, "query aborted by the caller"); }
	// an error inside the body leaves no copy in flight behind: the stream is drained before the body's host buffers die
	return vk_run_guarded([&](vk_host_keep &keep) { return query_body(c, q, out, keep); },
		[&]() { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); },
		[](const char *what) { return fail(VK_ERR_INVALID, std::string("vk_query: ") + what); });
}

int vk_merge_topk(const vk_topk_out *sets, int32_t n_sets, int32_t len_t, int32_t max_matches, vk_topk_out *out) {
	if (!sets || !out || n_sets < 0) return fail(VK_ERR_INVALID, "null argument");
	if (max_matches < 1 || out->capacity < max_matches) return fail(VK_ERR_INVALID, "output capacity smaller than max_matches");
	struct Ref { float score; int64_t sent; int set, idx; };
	std::vector<Ref> all;
	// (a NaN score -- records arrive from other ranks unchecked -- would break the strict weak order std::sort needs: it ranks last,
	// as -inf; the kernels never produce one, degenerate vectors score 0)
	for (int s = 0; s < n_sets; s++)
		for (int i = 0; i < sets[s].n_out; i++) all.push_back({sets[s].score[i] == sets[s].score[i] ? sets[s].score[i] : -INFINITY, sets[s].sentence[i], s, i});
	std::sort(all.begin(), all.end(), [](const Ref &a, const Ref &b) {
		if (a.score != b.score) return a.score > b.score;
		return a.sent > b.sent;
	});
	const int n_out = (int)std::min<size_t>(all.size(), (size_t)max_matches);
	for (int i = 0; i < n_out; i++) {
		const Ref &r = all[(size_t)i];
		const vk_topk_out &src = sets[r.set];
		out->score[i] = r.score;
		out->sentence[i] = r.sent;
		if (out->raw_score) out->raw_score[i] = src.raw_score ? src.raw_score[r.idx] : 0.0f;
		if (out->mapping && src.mapping)
			memcpy(out->mapping + (size_t)i * len_t, src.mapping + (size_t)r.idx * len_t, (size_t)len_t * 2);
		if (out->edge_sim && src.edge_sim)
			memcpy(out->edge_sim + (size_t)i * len_t, src.edge_sim + (size_t)r.idx * len_t, (size_t)len_t * 4);
	}
	out->n_out = n_out;
	return VK_OK;
}

int vk_rwmd_from_rows(const float *S, int32_t ld, int32_t len_s, int32_t len_t, const int32_t *key_s, const int32_t *key_t,
	int32_t injective, int32_t symmetric, int32_t normalize_bow, float *score_out) {
	if (!S || !score_out) return fail(VK_ERR_INVALID, "null argument");
	if (len_s < 0 || len_t < 0 || ld < len_t) return fail(VK_ERR_INVALID, "rows narrower than the query");
	if ((key_s == nullptr) != (key_t == nullptr)) return fail(VK_ERR_INVALID, "vocabulary keys for both sides or for neither");
	if (symmetric && !normalize_bow) return fail(VK_ERR_INVALID, "the symmetric relaxed WMD needs normalised bags of words (alignment/wmd.h:441-449)");
	*score_out = vk_host::rwmd_from_rows(S, ld, len_s, len_t, key_s, key_t, injective != 0, symmetric != 0, normalize_bow != 0);
	return VK_OK;
}

static inline int record_w(int len_t) { return (std::max(1, len_t) + 15) / 16 * 16; }

int32_t vk_record_words(int32_t len_t) {
	const int w = record_w(len_t);
	return (5 + w / 2 + w + 3) / 4 * 4;
}

int vk_pack_records(const vk_topk_out *set, int32_t len_t, int32_t k, int64_t sentence_offset, int32_t *records) {
	if (!set || !records || k < 0 || len_t < 1) return fail(VK_ERR_INVALID, "null argument");
	if (set->n_out > k) return fail(VK_ERR_INVALID, "result set larger than k records");
	const int w = record_w(len_t), words = vk_record_words(len_t);
	memset(records, 0, (size_t)k * words * 4);
	for (int i = 0; i < set->n_out; i++) {
		int32_t *r = records + (size_t)i * words;
		r[0] = 1;
		memcpy(r + 1, set->score + i, 4);
		if (set->raw_score) memcpy(r + 2, set->raw_score + i, 4);
		const int64_t g = set->sentence[i] + sentence_offset;
		memcpy(r + 3, &g, 8);
		int16_t *m = (int16_t *)(r + 5);
		for (int j = 0; j < w; j++) m[j] = -1;
		if (set->mapping) memcpy(m, set->mapping + (size_t)i * len_t, (size_t)len_t * 2);
		if (set->edge_sim) memcpy(r + 5 + w / 2, set->edge_sim + (size_t)i * len_t, (size_t)len_t * 4);
	}
	return VK_OK;
}

int vk_merge_records(const int32_t *records, int32_t n_sets, int32_t len_t, int32_t k, vk_topk_out *out) {
	if (!records || !out || n_sets < 0 || k < 1 || len_t < 1) return fail(VK_ERR_INVALID, "null argument");
	if (out->capacity < k) return fail(VK_ERR_INVALID, "output capacity smaller than k");
	const int w = record_w(len_t), words = vk_record_words(len_t);
	struct Ref { float score; int64_t sent; const int32_t *rec; };
	std::vector<Ref> all;
	all.reserve((size_t)n_sets * k);
	for (size_t i = 0; i < (size_t)n_sets * k; i++) {
		const int32_t *r = records + i * words;
		if (!r[0]) continue;
		Ref e;
		memcpy(&e.score, r + 1, 4);
		if (e.score != e.score) e.score = -INFINITY;   // NaN ranks last (see vk_merge_topk): a strict weak order whatever arrives
		memcpy(&e.sent, r + 3, 8);
		e.rec = r;
		all.push_back(e);
	}
	// the order of vk_merge_topk (and of the selection on one GPU): score descending, ties by slice index descending
	std::sort(all.begin(), all.end(), [](const Ref &a, const Ref &b) {
		if (a.score != b.score) return a.score > b.score;
		return a.sent > b.sent;
	});
	const int n_out = (int)std::min<size_t>(all.size(), (size_t)k);
	for (int i = 0; i < n_out; i++) {
		const int32_t *r = all[(size_t)i].rec;
		out->score[i] = all[(size_t)i].score;
		out->sentence[i] = all[(size_t)i].sent;
		if (out->raw_score) memcpy(out->raw_score + i, r + 2, 4);
		if (out->mapping) memcpy(out->mapping + (size_t)i * len_t, r + 5, (size_t)len_t * 2);
		if (out->edge_sim) memcpy(out->edge_sim + (size_t)i * len_t, r + 5 + w / 2, (size_t)len_t * 4);
	}
	out->n_out = n_out;
	return VK_OK;
}

} // extern "C"
