// vk_corpus.cpp -- C-ABI (include/vectorian_hip.h): library state, corpus handles and their residency in HBM.
// No CPU compute fallback exists: without a HIP device every entry point that
// would compute returns VK_ERR_NO_DEVICE / VK_ERR_HIP.

#include "vk_internal.h"

std::string &vk_error_slot() {
	static thread_local std::string err;
	return err;
}

std::mutex &vk_ring_mutex() {
	static std::mutex mu;
	return mu;
}

void vk_devblock::release(void *p) {
	if (!p) return;
	{
		std::lock_guard<std::mutex> g(mu);
		ptrs.erase(std::remove(ptrs.begin(), ptrs.end(), p), ptrs.end());
	}
	(void)hipFree(p);
}

vk_devblock::~vk_devblock() {
	(void)hipSetDevice(device);
	for (void *p : ptrs) if (p) (void)hipFree(p);
}

int vk_wait_peer_turn(vk_corpus *c, hipStream_t st) {
	std::lock_guard<std::mutex> g(vk_ring_mutex());
	vk_corpus *p = c->peer;
	if (p && p->ev2_recorded.load()) VK_HIP(hipStreamWaitEvent(st, p->ev[2], 0));
	return VK_OK;
}

extern "C" {

int vk_abi_version(void) { return VK_ABI_VERSION; }

const char *vk_last_error(void) { return vk_error_slot().c_str(); }

int vk_device_count(int *count) {
	if (!count) return fail(VK_ERR_INVALID, "count is null");
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess) { *count = 0; return fail(VK_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
	*count = n;
	return VK_OK;
}

int vk_init(int device) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(VK_ERR_NO_DEVICE, "no HIP device available");
	if (device < 0 || device >= n) return fail(VK_ERR_INVALID, "device index out of range");
	VK_HIP(hipSetDevice(device));
	hipDeviceProp_t prop;
	VK_HIP(hipGetDeviceProperties(&prop, device));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return fail(VK_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
	return VK_OK;
}

int vk_corpus_create(const vk_corpus_desc *desc, vk_corpus_t **out) {
	if (!desc || !out) return fail(VK_ERR_INVALID, "null argument");
	if (desc->layout != VK_LAYOUT_CONTEXTUAL && desc->layout != VK_LAYOUT_STATIC) return fail(VK_ERR_INVALID, "bad layout");
	if (desc->d < 1 || desc->d > 8192) return fail(VK_ERR_INVALID, "embedding dimension out of range");
	if (desc->n_tokens < 0 || desc->n_tokens >= (1ll << 31) - 64) return fail(VK_ERR_INVALID, "n_tokens must be < 2^31 per shard");
	if (desc->n_sentences < 0 || desc->n_sentences >= (1ll << 31) - 8) return fail(VK_ERR_INVALID, "n_sentences out of range");
	if (desc->layout == VK_LAYOUT_STATIC && desc->vocab_size < 1) return fail(VK_ERR_INVALID, "static layout needs vocab_size >= 1");
	if (desc->precision != VK_PREC_BF16 && desc->precision != VK_PREC_F32) return fail(VK_ERR_INVALID, "bad precision");

	int dev = 0;
	VK_HIP(hipGetDevice(&dev));
	vk_corpus *c = new vk_corpus();
	c->desc = *desc;
	c->device = dev;
	c->shared = std::make_shared<vk_devblock>();
	c->shared->device = dev;
	c->d_pad = (desc->d + 15) / 16 * 16;
	c->prec = desc->precision == VK_PREC_F32 ? 1 : 0;
	if (c->prec) {
		c->nk32 = c->d_pad / 16;          // fp32 tiles: blocks of 16 features, 1 KiB each
		c->tail = 0;
		c->tile_bytes = c->d_pad * 64;
	} else {
		c->nk32 = (c->d_pad + 31) / 32;   // K=32 steps; the last one is half filled when tail
		c->tail = (c->d_pad % 32) ? 1 : 0;
		c->tile_bytes = c->d_pad * 32;
	}
	c->rows_total = desc->layout == VK_LAYOUT_STATIC ? desc->vocab_size : desc->n_tokens;
	c->n_tiles = (c->rows_total + 15) / 16 + 1;   // + one zero tile: waves may read one tile past the end

	int rc = VK_OK;
	do {
		if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rc = fail(VK_ERR_HIP, "hipStreamCreate failed"); break; }
		for (auto &e : c->ev) if (hipEventCreate(&e) != hipSuccess) { rc = fail(VK_ERR_HIP, "hipEventCreate failed"); break; }
		if (rc) break;
		if ((rc = alloc_shared(c, &c->d_tiles, (size_t)c->n_tiles * c->tile_bytes))) break;
		if (hipMemsetAsync(c->d_tiles, 0, (size_t)c->n_tiles * c->tile_bytes, c->stream) != hipSuccess) { rc = fail(VK_ERR_HIP, "memset failed"); break; }
		if (desc->keep_magnitudes && (rc = alloc_shared(c, &c->d_mag, (size_t)c->rows_total + 16))) break;
		if (desc->layout == VK_LAYOUT_STATIC) {
			if ((rc = alloc_shared(c, &c->d_tok_id, (size_t)desc->n_tokens + 64))) break;
			if ((rc = alloc_t(c, &c->d_table, (size_t)c->n_tiles * 16 * 16 * 4))) break;   // one [V_pad x 16] table per query tile
		}
		if ((rc = alloc_t(c, &c->d_qtile, (size_t)c->tile_bytes * 4))) break;   // up to 4 tiles of 16 query rows
		if ((rc = alloc_t(c, &c->d_ws, kGapTable))) break;
		if ((rc = alloc_t(c, &c->d_wt, 160))) break;
		if ((rc = alloc_t(c, &c->d_qids, 80))) break;
		if ((rc = alloc_t(c, &c->d_out_raw, VK_MAX_MATCHES))) break;
		if ((rc = alloc_t(c, &c->d_out_sim, (size_t)VK_MAX_MATCHES * 64))) break;
		if ((rc = alloc_t(c, &c->d_out_map, (size_t)VK_MAX_MATCHES * 64))) break;
	} while (0);
	if (rc) { vk_corpus_free(c); return rc; }
	*out = c;
	return VK_OK;
}

// A second handle on the same resident corpus: shares the read-only arrays (tiles, magnitudes, token ids, POS codes,
// slice table) and owns a stream, events and workspaces.  Two handles serve two queries at a time from two host
// threads: the selection, traceback and host part of one query overlap the scoring kernel of the next.
int vk_corpus_view(vk_corpus_t *src, vk_corpus_t **out) {
	if (!src || !out) return fail(VK_ERR_INVALID, "null argument");
	if (!src->finalized) return fail(VK_ERR_STATE, "corpus not finalized");
	if (src->is_view) return fail(VK_ERR_INVALID, "views are taken from the owning handle");
	VK_HIP(hipSetDevice(src->device));
	vk_corpus *c = new vk_corpus();
	c->desc = src->desc; c->device = src->device;
	c->d_pad = src->d_pad; c->nk32 = src->nk32; c->tail = src->tail; c->tile_bytes = src->tile_bytes; c->prec = src->prec;
	c->rows_total = src->rows_total; c->rows_appended = src->rows_appended; c->n_tiles = src->n_tiles;
	c->d_tiles = src->d_tiles; c->d_mag = src->d_mag; c->d_tok_id = src->d_tok_id; c->d_pos = src->d_pos; c->d_tag = src->d_tag;
	c->d_sent_start = src->d_sent_start; c->d_sent_end = src->d_sent_end; c->d_long_groups = src->d_long_groups;
	c->contiguous = src->contiguous; c->overlapping = src->overlapping; c->have_ids = src->have_ids; c->have_sent = src->have_sent; c->finalized = true;
	c->max_len = src->max_len; c->max_group_tiles = src->max_group_tiles; c->max_group_tokens = src->max_group_tokens; c->max_pair_tiles = src->max_pair_tiles; c->max_short_pair_tiles = src->max_short_pair_tiles;
	c->n_entries = src->n_entries; c->entry_sent = src->entry_sent;
	c->h_start = src->h_start; c->h_end = src->h_end; c->h_tok = src->h_tok; c->h_tag = src->h_tag;
	c->n_long_groups = src->n_long_groups; c->max_short_len = src->max_short_len; c->max_long_len = src->max_long_len; c->h_xlong = src->h_xlong; c->h_apart = src->h_apart;
	c->long_group_tiles = src->long_group_tiles; c->long_group_tokens = src->long_group_tokens;
	c->uniform_len = src->uniform_len;
	c->is_view = true;
	c->shared = src->shared;           // the arrays live as long as any handle names them
	c->vectors_of = src->vectors_of;
	c->shares_vectors = src->shares_vectors;
	int rc = VK_OK;
	do {
		if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rc = fail(VK_ERR_HIP, "hipStreamCreate failed"); break; }
		for (auto &e : c->ev) if (hipEventCreate(&e) != hipSuccess) { rc = fail(VK_ERR_HIP, "hipEventCreate failed"); break; }
		if (rc) break;
		if (c->desc.layout == VK_LAYOUT_STATIC && (rc = alloc_t(c, &c->d_table, (size_t)c->n_tiles * 16 * 16 * 4))) break;
		if ((rc = alloc_t(c, &c->d_qtile, (size_t)c->tile_bytes * 4))) break;
		if ((rc = alloc_t(c, &c->d_ws, kGapTable))) break;
		if ((rc = alloc_t(c, &c->d_wt, 160))) break;
		if ((rc = alloc_t(c, &c->d_qids, 80))) break;
		if ((rc = alloc_t(c, &c->d_out_raw, VK_MAX_MATCHES))) break;
		if ((rc = alloc_t(c, &c->d_out_sim, (size_t)VK_MAX_MATCHES * 64))) break;
		if ((rc = alloc_t(c, &c->d_out_map, (size_t)VK_MAX_MATCHES * 64))) break;
		if ((rc = alloc_t(c, &c->d_scores, (size_t)c->n_entries + 8))) break;
		if ((rc = alloc_t(c, &c->d_raw, (size_t)c->n_entries + 8))) break;
		const size_t nblk = (size_t)((c->n_entries + kTopkChunk - 1) / kTopkChunk) + 1;
		if ((rc = alloc_t(c, &c->d_keys[0], nblk * VK_MAX_MATCHES + kTopkChunk))) break;
		if ((rc = alloc_t(c, &c->d_keys[1], (nblk * VK_MAX_MATCHES) / 2 + 2 * kTopkChunk))) break;
	} while (0);
	if (rc) { vk_corpus_free(c); return rc; }
	{   // into the ring of the handles on this corpus, once the handle is complete
		std::lock_guard<std::mutex> g(vk_ring_mutex());
		c->peer = src->peer ? src->peer : src;
		src->peer = c;
	}
	*out = c;
	return VK_OK;
}

// Frees this handle's own stream, events and workspaces.  The arrays it shares with other handles (a corpus and its views, the
// vocabulary of a filtered static corpus) belong to refcounted blocks and go with the last handle: any order of frees is fine, and
// a peer that is inside vk_query on another thread keeps everything it reads.  (Calls on ONE handle are the caller's to serialise.)
int vk_corpus_free(vk_corpus_t *c) {
	if (!c) return VK_OK;
	(void)hipSetDevice(c->device);
	{   // out of the ring first: from here on no peer reaches this handle's events
		std::lock_guard<std::mutex> g(vk_ring_mutex());
		if (c->peer) {
			vk_corpus *p = c->peer;
			while (p->peer != c) p = p->peer;
			p->peer = c->peer == p ? nullptr : c->peer;
			c->peer = nullptr;
		}
	}
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	void *ptrs[] = {c->d_stage, c->d_qtile, c->d_ws, c->d_wt, c->d_qids,
		c->d_table, c->d_scores, c->d_raw, c->d_boost, c->d_keys[0], c->d_keys[1], c->d_out_raw, c->d_out_sim, c->d_out_map, c->d_wrd_raw, c->d_wrd_val, c->d_bq, c->d_bqlen, c->d_bscores, c->d_bkeys[0], c->d_bkeys[1], c->d_counter, c->d_rows_out, c->d_plan_out, c->d_braw, c->d_wrdl_scratch, c->d_wide_scratch, c->d_wide_order, c->d_xlong_order, c->d_apart_order, c->d_bqt, c->d_bcand, c->d_bcandq, c->d_brows, c->d_qbits,
		c->d_sb_id[0], c->d_sb_id[1], c->d_btable, c->d_bfix, c->d_bqids, c->d_sort[0], c->d_sort[1], c->d_sort_temp,
		c->lq.qt, c->lq.fl, c->lq.il, c->lq.table, c->lq.scratch, c->lq.fscratch, c->lq.raw, c->lq.sim, c->lq.map};
	for (void *p : ptrs) if (p) (void)hipFree(p);
	if (c->h_brows) (void)hipHostFree(c->h_brows);
	for (auto &b : c->bl) for (void *p : {(void *)b.tiles, (void *)b.len, (void *)b.id}) if (p) (void)hipFree(p);
	for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
	return VK_OK;
}

int vk_corpus_device_bytes(const vk_corpus_t *c, int64_t *bytes) {
	if (!c || !bytes) return fail(VK_ERR_INVALID, "null argument");
	*bytes = c->device_bytes;
	return VK_OK;
}

int vk_corpus_append_vectors(vk_corpus_t *c, const void *rows, int64_t n_rows, int32_t dtype, int32_t mem, int32_t normalize) {
	if (!c || (!rows && n_rows > 0)) return fail(VK_ERR_INVALID, "null argument");
	if (c->finalized) return fail(VK_ERR_STATE, "corpus already finalized");
	if (dtype != VK_F32 && dtype != VK_BF16) return fail(VK_ERR_INVALID, "bad dtype");
	if (n_rows < 0 || c->rows_appended + n_rows > c->rows_total) return fail(VK_ERR_INVALID, "more rows appended than declared");
	VK_HIP(hipSetDevice(c->device));
	const size_t esz = dtype == VK_F32 ? 4 : 2;
	const size_t row_bytes = esz * (size_t)c->desc.d;
	if (mem == VK_MEM_DEVICE) {
		VK_HIP(vk_launch_pack(rows, dtype == VK_BF16, n_rows, c->desc.d, c->d_pad, c->rows_appended, c->d_tiles, c->d_mag,
			normalize, c->prec, c->stream));
		VK_HIP(hipStreamSynchronize(c->stream));
		c->rows_appended += n_rows;
		return VK_OK;
	}
	if (mem != VK_MEM_HOST) return fail(VK_ERR_INVALID, "bad memory kind");
	if (!c->d_stage) {
		int rc = alloc(c, &c->d_stage, (size_t)kStageBytes);
		if (rc) return rc;
	}
	const int64_t rows_per_chunk = std::max<int64_t>(1, kStageBytes / (int64_t)row_bytes);
	for (int64_t r = 0; r < n_rows; r += rows_per_chunk) {
		const int64_t nr = std::min(rows_per_chunk, n_rows - r);
		VK_HIP(hipMemcpyAsync(c->d_stage, (const uint8_t *)rows + (size_t)r * row_bytes, (size_t)nr * row_bytes, hipMemcpyHostToDevice, c->stream));
		VK_HIP(vk_launch_pack(c->d_stage, dtype == VK_BF16, nr, c->desc.d, c->d_pad, c->rows_appended + r, c->d_tiles, c->d_mag,
			normalize, c->prec, c->stream));
		VK_HIP(hipStreamSynchronize(c->stream));
	}
	c->rows_appended += n_rows;
	return VK_OK;
}

int vk_corpus_set_token_ids(vk_corpus_t *c, const int32_t *ids, int64_t n, int32_t mem) {
	if (!c || !ids) return fail(VK_ERR_INVALID, "null argument");
	if (c->desc.layout != VK_LAYOUT_STATIC) return fail(VK_ERR_STATE, "token ids belong to the static layout");
	if (c->finalized) return fail(VK_ERR_STATE, "corpus already finalized");
	if (n != c->desc.n_tokens) return fail(VK_ERR_INVALID, "token id count differs from n_tokens");
	VK_HIP(hipSetDevice(c->device));
	if (mem == VK_MEM_HOST) {
		for (int64_t i = 0; i < n; i++)
			if (ids[i] < 0 || ids[i] >= c->desc.vocab_size) return fail(VK_ERR_INVALID, "token id outside the vocabulary");
	}
	VK_HIP(hipMemcpyAsync(c->d_tok_id, ids, (size_t)n * 4, mem == VK_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
	VK_HIP(hipStreamSynchronize(c->stream));
	c->h_tok = std::make_shared<std::vector<int32_t>>((size_t)n);
	if (mem == VK_MEM_DEVICE) VK_HIP(hipMemcpy(c->h_tok->data(), c->d_tok_id, (size_t)n * 4, hipMemcpyDeviceToHost));
	else memcpy(c->h_tok->data(), ids, (size_t)n * 4);
	c->have_ids = true;
	return VK_OK;
}

static int set_token_codes(vk_corpus_t *c, int8_t **slot, const int8_t *codes, int64_t n, int32_t mem, const char *what) {
	if (!c || !codes) return fail(VK_ERR_INVALID, "null argument");
	if (n != c->desc.n_tokens) return fail(VK_ERR_INVALID, std::string(what) + " count differs from n_tokens");
	if (c->is_view) return fail(VK_ERR_STATE, std::string("set ") + what + " codes on the owning handle, before taking views");
	VK_HIP(hipSetDevice(c->device));
	if (!*slot) {
		int rc = alloc_shared(c, slot, (size_t)n + 64);
		if (rc) return rc;
		VK_HIP(hipMemsetAsync(*slot, 0, (size_t)n + 64, c->stream));
	}
	VK_HIP(hipMemcpyAsync(*slot, codes, (size_t)n, mem == VK_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
	VK_HIP(hipStreamSynchronize(c->stream));
	return VK_OK;
}

int vk_corpus_set_token_pos(vk_corpus_t *c, const int8_t *pos, int64_t n, int32_t mem) {
	return set_token_codes(c, c ? &c->d_pos : nullptr, pos, n, mem, "POS");
}

int vk_corpus_set_token_tags(vk_corpus_t *c, const int8_t *tags, int64_t n, int32_t mem) {
	// The (token id, tag) vocabulary keys of tag-weighted transports are built as id * 256 + tag and must order like upstream's
	// signed pairs (TaggedTokenFactory, alignment/bow.h:150-176): tag codes have to be 0 .. 127.  Host arrays are checked here.
	if (c && tags && mem != VK_MEM_DEVICE && n == c->desc.n_tokens)
		for (int64_t i = 0; i < n; i++)
			if (tags[i] < 0) return fail(VK_ERR_INVALID, "tag codes must be 0 .. 127 (they key the vocabulary of tag-weighted transports as id * 256 + tag)");
	const int rc = set_token_codes(c, c ? &c->d_tag : nullptr, tags, n, mem, "tag");
	if (rc) return rc;
	c->h_tag = std::make_shared<std::vector<int8_t>>((size_t)n);
	if (mem == VK_MEM_DEVICE) VK_HIP(hipMemcpy(c->h_tag->data(), c->d_tag, (size_t)n, hipMemcpyDeviceToHost));
	else memcpy(c->h_tag->data(), tags, (size_t)n);
	return VK_OK;
}

static int set_slices_impl(vk_corpus_t *c, const int64_t *start, const int64_t *end, int64_t n_sentences, bool contiguous) {
	int max_len = 0, max_short = 0, max_long = 0;
	int64_t n_long = 0;
	for (int64_t s = 0; s < n_sentences; s++) {
		const int64_t len = end[s] - start[s];
		if (start[s] < 0 || end[s] > c->desc.n_tokens || len < 0) return fail(VK_ERR_INVALID, "slice outside the token stream");
		if (s > 0 && (start[s] < start[s - 1] || end[s] < end[s - 1])) return fail(VK_ERR_INVALID, "slice starts and ends must be non-decreasing");
		if (len > VK_MAX_DOC_LEN) {   // (65 .. VK_MAX_SENT_LEN: every algorithm; beyond: alignments, vk_validate_query)
			char buf[128];
			snprintf(buf, sizeof buf, "slice %lld has %lld tokens; the HIP path handles at most %d", (long long)s, (long long)len, VK_MAX_DOC_LEN);
			return fail(VK_ERR_UNSUPPORTED, buf);
		}
		max_len = std::max(max_len, (int)len);
		if (len > VK_FAST_SENT_LEN) n_long++;
		else max_short = std::max(max_short, (int)len);
		if (len > VK_FAST_SENT_LEN && len <= VK_MAX_SENT_LEN) max_long = std::max(max_long, (int)len);
	}
	VK_HIP(hipSetDevice(c->device));

	// ---- the slice table.  Without long slices it is the caller's table.  A long slice gets a group of 4 rows
	// of its own ([L, empty, empty, empty]); the group before it is closed with empty rows, so that no group of
	// the main launch spans the tokens of a long slice.  Rows stay in slice order (ties are broken by row).
	// apart_groups: every group that holds a slice of more than 64 tokens; long_groups: those of them the long pass takes (65 .. 512
	// tokens, LDS strips); the slices beyond (whole documents) are left to the one-wave-per-slice kernel (xlong_entries: their rows)
	std::vector<int32_t> st32, en32, long_groups, apart_groups;
	auto xlong_entries = std::make_shared<std::vector<int32_t>>();
	auto apart_entries = std::make_shared<std::vector<int32_t>>();   // the rows of every group apart (general gaps: all of them take the one-wave-per-slice pass)
	c->entry_sent.clear(); c->sent_entry.clear();
	if (n_long == 0) {
		st32.resize((size_t)n_sentences); en32.resize((size_t)n_sentences);
		for (int64_t s = 0; s < n_sentences; s++) { st32[(size_t)s] = (int32_t)start[s]; en32[(size_t)s] = (int32_t)end[s]; }
	} else {
		if (n_sentences + 3 * n_long + 3 >= (1ll << 31) - 16) return fail(VK_ERR_INVALID, "slice table too large");
		auto push = [&](int32_t a, int32_t b, int32_t sent) { st32.push_back(a); en32.push_back(b); c->entry_sent.push_back(sent); };
		for (int64_t s = 0; s < n_sentences; s++) {
			if (end[s] - start[s] > VK_FAST_SENT_LEN) {
				while (st32.size() % 4) push(en32.back(), en32.back(), -1);
				apart_groups.push_back((int32_t)(st32.size() / 4));
				for (int i = 0; i < 4; i++) apart_entries->push_back((int32_t)st32.size() + i);
				if (end[s] - start[s] > VK_MAX_SENT_LEN) {
					// (the four rows of its group: no other pass writes the scores of the three empty ones)
					for (int i = 0; i < 4; i++) xlong_entries->push_back((int32_t)st32.size() + i);
				} else long_groups.push_back((int32_t)(st32.size() / 4));
				push((int32_t)start[s], (int32_t)end[s], (int32_t)s);
				for (int i = 0; i < 3; i++) push((int32_t)end[s], (int32_t)end[s], -1);
			} else {
				push((int32_t)start[s], (int32_t)end[s], (int32_t)s);
			}
		}
	}
	const int64_t n_entries = (int64_t)st32.size();
	const int32_t tail = n_entries > 0 ? en32.back() : 0;
	for (int i = 0; i < 8; i++) { st32.push_back(tail); en32.push_back(tail); }   // padding: empty slices at the end

	// ---- device arrays sized by the table (re-created when the slices are set again)
	for (void *p : {(void *)c->d_sent_start, (void *)c->d_sent_end, (void *)c->d_long_groups}) c->shared->release(p);
	for (void *p : {(void *)c->d_scores, (void *)c->d_raw, (void *)c->d_keys[0], (void *)c->d_keys[1], (void *)c->d_boost})
		if (p) VK_HIP(hipFree(p));
	c->d_sent_start = c->d_sent_end = nullptr; c->d_scores = c->d_raw = c->d_boost = nullptr; c->d_keys[0] = c->d_keys[1] = nullptr; c->d_long_groups = nullptr;
	int rc;
	if ((rc = alloc_shared(c, &c->d_sent_start, st32.size()))) return rc;
	if ((rc = alloc_shared(c, &c->d_sent_end, en32.size()))) return rc;
	if ((rc = alloc_t(c, &c->d_scores, (size_t)n_entries + 8))) return rc;
	if ((rc = alloc_t(c, &c->d_raw, (size_t)n_entries + 8))) return rc;
	const size_t nblk = (size_t)((n_entries + kTopkChunk - 1) / kTopkChunk) + 1;
	if ((rc = alloc_t(c, &c->d_keys[0], nblk * VK_MAX_MATCHES + kTopkChunk))) return rc;
	if ((rc = alloc_t(c, &c->d_keys[1], (nblk * VK_MAX_MATCHES) / 2 + 2 * kTopkChunk))) return rc;
	if (!long_groups.empty()) {
		if ((rc = alloc_shared(c, &c->d_long_groups, long_groups.size()))) return rc;
		VK_HIP(hipMemcpy(c->d_long_groups, long_groups.data(), long_groups.size() * 4, hipMemcpyHostToDevice));
	}
	VK_HIP(hipMemcpy(c->d_sent_start, st32.data(), st32.size() * 4, hipMemcpyHostToDevice));
	VK_HIP(hipMemcpy(c->d_sent_end, en32.data(), en32.size() * 4, hipMemcpyHostToDevice));
	c->h_start = std::make_shared<std::vector<int32_t>>(st32);
	c->h_end = std::make_shared<std::vector<int32_t>>(en32);
	c->n_entries = n_entries;
	if (c->d_wide_order) { VK_HIP(hipFree(c->d_wide_order)); c->d_wide_order = nullptr; }
	if (c->d_xlong_order) { VK_HIP(hipFree(c->d_xlong_order)); c->d_xlong_order = nullptr; }
	if (c->d_apart_order) { VK_HIP(hipFree(c->d_apart_order)); c->d_apart_order = nullptr; }
	c->n_wide_order = c->n_xlong_order = c->n_apart_order = -1;   // the work list of the one-wave-per-slice pass follows the table
	c->n_long_groups = (int)long_groups.size();
	c->max_len = max_len;
	c->max_short_len = max_short;
	c->max_long_len = max_long;
	c->h_xlong = xlong_entries;
	c->h_apart = apart_entries;
	c->contiguous = contiguous;
	c->overlapping = false;
	for (int64_t s = 1; s < n_sentences && !c->overlapping; s++) c->overlapping = start[s] < end[s - 1] && end[s] > start[s];
	c->uniform_len = 0;
	if (n_sentences > 0 && contiguous) {
		const int64_t l0 = end[0] - start[0];
		bool uni = l0 > 0;
		for (int64_t s = 1; s < n_sentences && uni; s++) uni = (end[s] - start[s]) == l0;
		if (uni) c->uniform_len = (int)l0;
	}
	// per wave: groups of 4 consecutive rows; LDS strips are sized for the main launch and the long one apart
	int mt = 1, mtok = 1, lt = 1, ltok = 1;
	size_t li = 0;
	for (int64_t g = 0; g * 4 < n_entries; g++) {
		const int64_t a = st32[(size_t)(g * 4)], b = en32[(size_t)std::min<int64_t>(g * 4 + 3, n_entries - 1)];
		const int tiles = (int)(((b + 15) >> 4) - (a >> 4));
		if (li < apart_groups.size() && apart_groups[li] == g) {
			li++;
			if (b - a <= VK_MAX_SENT_LEN) {   // (a group apart holds one slice: its tokens are the slice's)
				lt = std::max(lt, tiles);
				ltok = std::max(ltok, (int)(b - a));
			}
		} else {
			mt = std::max(mt, tiles);
			mtok = std::max(mtok, (int)(b - a));
		}
	}
	int pt = 1, spt = 1;
	li = 0;
	for (int64_t g = 0; g * 2 < n_entries; g++) {
		const int64_t a = st32[(size_t)(g * 2)], b = en32[(size_t)std::min<int64_t>(g * 2 + 1, n_entries - 1)];
		const int tiles = (int)(((b + 15) >> 4) - (a >> 4));
		pt = std::max(pt, tiles);
		while (li < apart_groups.size() && apart_groups[li] < g / 2) li++;
		if (!(li < apart_groups.size() && apart_groups[li] == g / 2)) spt = std::max(spt, tiles);
	}
	c->max_pair_tiles = pt;
	c->max_short_pair_tiles = spt;
	c->max_group_tiles = mt;
	c->max_group_tokens = mtok;
	c->long_group_tiles = lt;
	c->long_group_tokens = ltok;
	c->have_sent = true;
	return VK_OK;
}

int vk_corpus_set_sentences(vk_corpus_t *c, const int64_t *sent_off, int64_t n_sentences) {
	if (!c || !sent_off) return fail(VK_ERR_INVALID, "null argument");
	if (c->finalized) return fail(VK_ERR_STATE, "corpus already finalized");
	if (n_sentences != c->desc.n_sentences) return fail(VK_ERR_INVALID, "sentence count differs from n_sentences");
	if (sent_off[0] != 0) return fail(VK_ERR_INVALID, "sentence spans must start at token 0 (document.h:151-168)");
	if (sent_off[n_sentences] != c->desc.n_tokens) return fail(VK_ERR_INVALID, "sentence spans must cover exactly n_tokens");
	for (int64_t s = 0; s < n_sentences; s++)
		if (sent_off[s + 1] < sent_off[s]) return fail(VK_ERR_INVALID, "sentence offsets must be non-decreasing");
	return set_slices_impl(c, sent_off, sent_off + 1, n_sentences, true);
}

int vk_corpus_set_slices(vk_corpus_t *c, const int64_t *start, const int64_t *end, int64_t n_sentences) {
	if (!c || !start || !end) return fail(VK_ERR_INVALID, "null argument");
	if (c->finalized) return fail(VK_ERR_STATE, "corpus already finalized");
	if (n_sentences != c->desc.n_sentences) return fail(VK_ERR_INVALID, "slice count differs from n_sentences");
	return set_slices_impl(c, start, end, n_sentences, false);
}

int vk_corpus_finalize(vk_corpus_t *c) {
	if (!c) return fail(VK_ERR_INVALID, "null argument");
	if (c->rows_appended != c->rows_total) return fail(VK_ERR_STATE, "not all vectors were appended");
	if (!c->have_sent) return fail(VK_ERR_STATE, "sentence spans were not set");
	if (c->desc.layout == VK_LAYOUT_STATIC && !c->have_ids) return fail(VK_ERR_STATE, "token ids were not set");
	VK_HIP(hipSetDevice(c->device));
	if (c->d_stage) { VK_HIP(hipFree(c->d_stage)); c->d_stage = nullptr; c->device_bytes -= kStageBytes; }
	VK_HIP(hipStreamSynchronize(c->stream));
	c->finalized = true;
	return VK_OK;
}

// The filtered corpus: keep flags and their scan on the device, slice table re-indexed, token rows / ids / codes /
// magnitudes gathered.  A handle of its own (stream, workspaces); the static layout shares the vocabulary arrays.
int vk_corpus_filter(vk_corpus_t *src, uint64_t pos_mask, uint64_t tag_mask, vk_corpus_t **out) {
	if (!src || !out) return fail(VK_ERR_INVALID, "null argument");
	if (!src->finalized) return fail(VK_ERR_STATE, "corpus not finalized");
	if (pos_mask && !src->d_pos) return fail(VK_ERR_STATE, "pos filter needs vk_corpus_set_token_pos");
	if (tag_mask && !src->d_tag) return fail(VK_ERR_STATE, "tag filter needs vk_corpus_set_token_tags");
	VK_HIP(hipSetDevice(src->device));
	hipStream_t st = src->stream;
	const int64_t n = src->desc.n_tokens, ne = src->n_entries, ns = src->desc.n_sentences;
	const bool is_static = src->desc.layout == VK_LAYOUT_STATIC;

	int32_t *keep = nullptr, *new_index = nullptr, *src_of = nullptr, *fs = nullptr, *fe = nullptr;
	void *temp = nullptr;
	vk_corpus *c = nullptr;
	int rc = VK_OK;
	auto body = [&]() -> int {
		VK_HIP(hipMalloc((void **)&keep, ((size_t)n + 1) * 4));
		VK_HIP(hipMalloc((void **)&new_index, ((size_t)n + 1) * 4));
		VK_HIP(hipMalloc((void **)&src_of, ((size_t)n + 1) * 4));
		VK_HIP(hipMalloc((void **)&fs, ((size_t)ne + 1) * 4));
		VK_HIP(hipMalloc((void **)&fe, ((size_t)ne + 1) * 4));
		size_t temp_bytes = 0;
		VK_HIP(vk_launch_filter_scan(nullptr, nullptr, 0, 0, n, keep, new_index, nullptr, &temp_bytes, st));
		VK_HIP(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
		VK_HIP(vk_launch_filter_scan(src->d_pos, src->d_tag, pos_mask, tag_mask, n, keep, new_index, temp, &temp_bytes, st));
		VK_HIP(vk_launch_filter_maps(keep, new_index, n, src_of, src->d_sent_start, src->d_sent_end, ne, fs, fe, st));
		int32_t n_kept32 = 0;
		std::vector<int32_t> hs((size_t)ne), he((size_t)ne);
		VK_HIP(hipMemcpyAsync(&n_kept32, new_index + n, 4, hipMemcpyDeviceToHost, st));
		if (ne > 0) {
			VK_HIP(hipMemcpyAsync(hs.data(), fs, (size_t)ne * 4, hipMemcpyDeviceToHost, st));
			VK_HIP(hipMemcpyAsync(he.data(), fe, (size_t)ne * 4, hipMemcpyDeviceToHost, st));
		}
		VK_HIP(hipStreamSynchronize(st));
		const int64_t n_kept = n_kept32;

		// slices of the filtered stream, per sentence (rows of the source's table that are padding carry no sentence)
		std::vector<int64_t> start((size_t)ns), end((size_t)ns);
		for (int64_t e = 0; e < ne; e++) {
			const int64_t sent = src->entry_sent.empty() ? e : src->entry_sent[(size_t)e];
			if (sent >= 0 && sent < ns) { start[(size_t)sent] = hs[(size_t)e]; end[(size_t)sent] = he[(size_t)e]; }
		}

		vk_corpus_desc desc = src->desc;
		desc.n_tokens = n_kept;
		if (is_static) {
			// token ids and slices are this handle's, the vocabulary (tiles, magnitudes) stays the source's
			c = new vk_corpus();
			c->desc = desc; c->device = src->device;
			c->shared = std::make_shared<vk_devblock>();
			c->shared->device = src->device;
			c->d_pad = src->d_pad; c->nk32 = src->nk32; c->tail = src->tail; c->tile_bytes = src->tile_bytes; c->prec = src->prec;
			c->rows_total = c->rows_appended = src->rows_total; c->n_tiles = src->n_tiles;
			c->d_tiles = src->d_tiles; c->d_mag = src->d_mag; c->shares_vectors = true;
			c->vectors_of = src->shares_vectors ? src->vectors_of : src->shared;   // the vocabulary stays alive with this handle
			if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(VK_ERR_HIP, "hipStreamCreate failed");
			for (auto &e : c->ev) if (hipEventCreate(&e) != hipSuccess) return fail(VK_ERR_HIP, "hipEventCreate failed");
			int r;
			if ((r = alloc_shared(c, &c->d_tok_id, (size_t)n_kept + 64))) return r;
			if ((r = alloc_t(c, &c->d_table, (size_t)c->n_tiles * 16 * 16 * 4))) return r;
			if ((r = alloc_t(c, &c->d_qtile, (size_t)c->tile_bytes * 4))) return r;
			if ((r = alloc_t(c, &c->d_ws, kGapTable))) return r;
			if ((r = alloc_t(c, &c->d_wt, 160))) return r;
			if ((r = alloc_t(c, &c->d_qids, 80))) return r;
			if ((r = alloc_t(c, &c->d_out_raw, VK_MAX_MATCHES))) return r;
			if ((r = alloc_t(c, &c->d_out_sim, (size_t)VK_MAX_MATCHES * 64))) return r;
			if ((r = alloc_t(c, &c->d_out_map, (size_t)VK_MAX_MATCHES * 64))) return r;
			VK_HIP(hipMemsetAsync(c->d_tok_id, 0, ((size_t)n_kept + 64) * 4, st));
			VK_HIP(vk_launch_filter_gather(src->d_tok_id, c->d_tok_id, 4, src_of, n_kept, st));
			c->have_ids = true;
		} else {
			int r = vk_corpus_create(&desc, &c);
			if (r) return r;
			VK_HIP(hipStreamSynchronize(c->stream));   // the zero fill of the new tiles
			VK_HIP(vk_launch_filter_rows(src->d_tiles, c->d_tiles, src_of, n_kept, src->tile_bytes, st));
			if (src->d_mag && c->d_mag) VK_HIP(vk_launch_filter_gather(src->d_mag, c->d_mag, 4, src_of, n_kept, st));
			c->rows_appended = c->rows_total;
		}
		for (int which = 0; which < 2; which++) {
			int8_t *from = which ? src->d_tag : src->d_pos;
			int8_t **to = which ? &c->d_tag : &c->d_pos;
			if (!from) continue;
			int r = alloc_shared(c, to, (size_t)n_kept + 64);
			if (r) return r;
			VK_HIP(hipMemsetAsync(*to, 0, (size_t)n_kept + 64, st));
			VK_HIP(vk_launch_filter_gather(from, *to, 1, src_of, n_kept, st));
		}
		VK_HIP(hipStreamSynchronize(st));
		if (is_static) {
			c->h_tok = std::make_shared<std::vector<int32_t>>((size_t)n_kept);
			if (n_kept > 0) VK_HIP(hipMemcpy(c->h_tok->data(), c->d_tok_id, (size_t)n_kept * 4, hipMemcpyDeviceToHost));
		}
		if (c->d_tag) {
			c->h_tag = std::make_shared<std::vector<int8_t>>((size_t)n_kept);
			if (n_kept > 0) VK_HIP(hipMemcpy(c->h_tag->data(), c->d_tag, (size_t)n_kept, hipMemcpyDeviceToHost));
		}
		int r = set_slices_impl(c, start.data(), end.data(), ns, src->contiguous);
		if (r) return r;
		c->finalized = true;
		return VK_OK;
	};
	rc = body();
	for (void *p : {(void *)keep, (void *)new_index, (void *)src_of, (void *)fs, (void *)fe, temp}) if (p) (void)hipFree(p);
	if (rc) { if (c) vk_corpus_free(c); return rc; }
	*out = c;
	return VK_OK;
}

int vk_last_scores(vk_corpus_t *c, float *scores, int64_t n) {
	if (!c || !scores) return fail(VK_ERR_INVALID, "null argument");
	if (!c->have_scores) return fail(VK_ERR_STATE, "no query has run on this corpus");
	if (n != c->desc.n_sentences) return fail(VK_ERR_INVALID, "n differs from n_sentences");
	VK_HIP(hipSetDevice(c->device));
	if (c->entry_sent.empty()) {
		VK_HIP(hipMemcpy(scores, c->d_scores, (size_t)n * 4, hipMemcpyDeviceToHost));
		return VK_OK;
	}
	std::vector<float> rows((size_t)c->n_entries);
	VK_HIP(hipMemcpy(rows.data(), c->d_scores, rows.size() * 4, hipMemcpyDeviceToHost));
	for (int64_t e = 0; e < c->n_entries; e++)
		if (c->entry_sent[(size_t)e] >= 0) scores[c->entry_sent[(size_t)e]] = rows[(size_t)e];
	return VK_OK;
}

int vk_last_timings(const vk_corpus_t *c, vk_timings *t) {
	if (!c || !t) return fail(VK_ERR_INVALID, "null argument");
	*t = c->last;
	return VK_OK;
}

} // extern "C"
