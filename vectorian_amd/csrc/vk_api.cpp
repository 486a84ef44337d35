// vk_api.cpp -- implementation of the C-ABI declared in include/vectorian_hip.h.
// Host side only: validation, device residency, launches, result assembly.
// No CPU compute fallback exists: without a HIP device every entry point that
// would compute returns VK_ERR_NO_DEVICE / VK_ERR_HIP.

#include "../../include/vectorian_hip.h"
#include "vk_device.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
	g_err = msg;
	return code;
}

#define VK_HIP(call) \
	do { \
		hipError_t e_ = (call); \
		if (e_ != hipSuccess) { \
			char buf_[512]; \
			snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
			return fail(VK_ERR_HIP, buf_); \
		} \
	} while (0)

uint16_t f32_to_bf16(float x) {
	uint32_t u;
	memcpy(&u, &x, 4);
	if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
	u += 0x7fffu + ((u >> 16) & 1u);
	return (uint16_t)(u >> 16);
}

float bf16_to_f32(uint16_t b) {
	uint32_t u = ((uint32_t)b) << 16;
	float f;
	memcpy(&f, &u, 4);
	return f;
}

float gap_cost(const vk_gap &g, int k) {
	if (k <= 0) return 0.0f;
	switch (g.kind) {
	case VK_GAP_LINEAR: return g.u * (float)k;
	case VK_GAP_AFFINE: return g.u + g.v * (float)k;
	default: return (g.table && k < g.n_table) ? g.table[k] : INFINITY;
	}
}

constexpr int kTopkChunk = 2048;
constexpr int kGapTable = 640;   // entries of the gap tables sent to the device (> VK_MAX_SENT_LEN)
constexpr int64_t kStageBytes = 64ll << 20;

} // namespace

struct vk_corpus {
	vk_corpus_desc desc{};
	int device = 0;
	hipStream_t stream = nullptr;
	int d_pad = 0, nk32 = 0, tail = 0, tile_bytes = 0;
	int prec = 0;                // vk_precision: 1 = fp32 tiles (nk32 then counts blocks of 16 features, tail = 0)
	int64_t rows_total = 0, rows_appended = 0, n_tiles = 0;
	uint8_t *d_tiles = nullptr;
	float *d_mag = nullptr;
	int32_t *d_tok_id = nullptr;
	int8_t *d_pos = nullptr;   // POS code per token (tag-weighted queries)
	int32_t *d_sent_start = nullptr, *d_sent_end = nullptr;
	bool contiguous = false;   // slices are the CSR partition of the token stream
	bool have_ids = false, have_sent = false, finalized = false;
	int max_len = 0, max_group_tiles = 0, max_group_tokens = 0;
	// slice table on the device: n_entries >= n_sentences rows.  Slices longer than VK_FAST_SENT_LEN sit alone in
	// their group of 4 (padded with empty rows) and are scored by a second launch over d_long_groups.
	int64_t n_entries = 0;
	std::vector<int32_t> entry_sent;   // [n_entries] sentence of a row, -1 = padding; empty when the table is the identity
	int32_t *d_long_groups = nullptr;
	int n_long_groups = 0, max_short_len = 0, long_group_tiles = 0, long_group_tokens = 0;
	int uniform_len = 0;       // > 0: every sentence has exactly this many tokens
	uint8_t *d_bq = nullptr; int32_t *d_bqlen = nullptr; float *d_bscores = nullptr; uint64_t *d_bkeys[2] = {nullptr, nullptr};
	float *d_braw = nullptr; size_t braw_cap = 0;   // aligner scores of a batch of alignment queries
	size_t bq_cap = 0, bscores_cap = 0, bkeys_cap = 0;
	int64_t device_bytes = 0;
	// workspaces
	void *d_stage = nullptr;
	uint8_t *d_qtile = nullptr;
	float *d_ws = nullptr, *d_wt = nullptr;
	int32_t *d_qids = nullptr;
	float *d_table = nullptr;
	float *d_scores = nullptr, *d_raw = nullptr, *d_boost = nullptr;
	uint64_t *d_keys[2] = {nullptr, nullptr};
	float *d_out_raw = nullptr, *d_out_sim = nullptr;
	float *d_wrd_raw = nullptr, *d_wrd_val = nullptr;
	uint32_t *d_counter = nullptr;
	float *d_rows_out = nullptr, *d_plan_out = nullptr;   // transport flows of the winners
	size_t wrd_cap = 0;          // candidates d_wrd_raw / d_wrd_val (and d_keys[0]) can hold
	int16_t *d_out_map = nullptr;
	hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
	vk_timings last{};
	bool have_scores = false;
	bool is_view = false;        // shares the corpus arrays of another handle (vk_corpus_view): does not free them
	vk_corpus *peer = nullptr;   // ring of the handles on one corpus: a handle's scoring kernel starts after its peer's
	bool ev2_recorded = false;   // (device-side wait on ev[2]), so that scoring kernels run back to back, never queued inside each other
};

namespace {

int alloc(vk_corpus *c, void **p, size_t bytes) {
	VK_HIP(hipMalloc(p, bytes ? bytes : 16));
	c->device_bytes += (int64_t)bytes;
	return VK_OK;
}

template <typename T> int alloc_t(vk_corpus *c, T **p, size_t n) { return alloc(c, (void **)p, n * sizeof(T)); }

} // namespace

extern "C" {

int vk_abi_version(void) { return VK_ABI_VERSION; }

const char *vk_last_error(void) { return g_err.c_str(); }

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
		if ((rc = alloc_t(c, &c->d_tiles, (size_t)c->n_tiles * c->tile_bytes))) break;
		if (hipMemsetAsync(c->d_tiles, 0, (size_t)c->n_tiles * c->tile_bytes, c->stream) != hipSuccess) { rc = fail(VK_ERR_HIP, "memset failed"); break; }
		if (desc->keep_magnitudes && (rc = alloc_t(c, &c->d_mag, (size_t)c->rows_total + 16))) break;
		if (desc->layout == VK_LAYOUT_STATIC) {
			if ((rc = alloc_t(c, &c->d_tok_id, (size_t)desc->n_tokens + 64))) break;
			if ((rc = alloc_t(c, &c->d_table, (size_t)c->n_tiles * 16 * 16 * 4))) break;   // one [V_pad x 16] table per query tile
		}
		if ((rc = alloc_t(c, &c->d_qtile, (size_t)c->tile_bytes * 4))) break;   // up to 4 tiles of 16 query rows
		if ((rc = alloc_t(c, &c->d_ws, kGapTable))) break;
		if ((rc = alloc_t(c, &c->d_wt, 80))) break;
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
	c->d_tiles = src->d_tiles; c->d_mag = src->d_mag; c->d_tok_id = src->d_tok_id; c->d_pos = src->d_pos;
	c->d_sent_start = src->d_sent_start; c->d_sent_end = src->d_sent_end; c->d_long_groups = src->d_long_groups;
	c->contiguous = src->contiguous; c->have_ids = src->have_ids; c->have_sent = src->have_sent; c->finalized = true;
	c->max_len = src->max_len; c->max_group_tiles = src->max_group_tiles; c->max_group_tokens = src->max_group_tokens;
	c->n_entries = src->n_entries; c->entry_sent = src->entry_sent;
	c->n_long_groups = src->n_long_groups; c->max_short_len = src->max_short_len;
	c->long_group_tiles = src->long_group_tiles; c->long_group_tokens = src->long_group_tokens;
	c->uniform_len = src->uniform_len;
	c->is_view = true;
	c->peer = src->peer ? src->peer : src;
	src->peer = c;
	int rc = VK_OK;
	do {
		if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rc = fail(VK_ERR_HIP, "hipStreamCreate failed"); break; }
		for (auto &e : c->ev) if (hipEventCreate(&e) != hipSuccess) { rc = fail(VK_ERR_HIP, "hipEventCreate failed"); break; }
		if (rc) break;
		if (c->desc.layout == VK_LAYOUT_STATIC && (rc = alloc_t(c, &c->d_table, (size_t)c->n_tiles * 16 * 16 * 4))) break;
		if ((rc = alloc_t(c, &c->d_qtile, (size_t)c->tile_bytes * 4))) break;
		if ((rc = alloc_t(c, &c->d_ws, kGapTable))) break;
		if ((rc = alloc_t(c, &c->d_wt, 80))) break;
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
	*out = c;
	return VK_OK;
}

int vk_corpus_free(vk_corpus_t *c) {
	if (!c) return VK_OK;
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	if (c->peer) {   // unlink from the ring of handles
		vk_corpus *p = c->peer;
		while (p->peer != c) p = p->peer;
		p->peer = c->peer == p ? nullptr : c->peer;
		c->peer = nullptr;
	}
	if (c->is_view) c->d_tiles = nullptr, c->d_mag = nullptr, c->d_tok_id = nullptr, c->d_pos = nullptr,
		c->d_sent_start = c->d_sent_end = nullptr, c->d_long_groups = nullptr;
	void *ptrs[] = {c->d_tiles, c->d_mag, c->d_tok_id, c->d_pos, c->d_sent_start, c->d_sent_end, c->d_stage, c->d_qtile, c->d_ws, c->d_wt, c->d_qids,
		c->d_table, c->d_scores, c->d_raw, c->d_boost, c->d_keys[0], c->d_keys[1], c->d_out_raw, c->d_out_sim, c->d_out_map, c->d_wrd_raw, c->d_wrd_val, c->d_bq, c->d_bqlen, c->d_bscores, c->d_bkeys[0], c->d_bkeys[1], c->d_long_groups, c->d_counter, c->d_rows_out, c->d_plan_out, c->d_braw};
	for (void *p : ptrs) if (p) (void)hipFree(p);
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
	c->have_ids = true;
	return VK_OK;
}

int vk_corpus_set_token_pos(vk_corpus_t *c, const int8_t *pos, int64_t n, int32_t mem) {
	if (!c || !pos) return fail(VK_ERR_INVALID, "null argument");
	if (n != c->desc.n_tokens) return fail(VK_ERR_INVALID, "POS count differs from n_tokens");
	if (c->is_view) return fail(VK_ERR_STATE, "set POS codes on the owning handle, before taking views");
	VK_HIP(hipSetDevice(c->device));
	if (!c->d_pos) {
		int rc = alloc_t(c, &c->d_pos, (size_t)n + 64);
		if (rc) return rc;
		VK_HIP(hipMemsetAsync(c->d_pos, 0, (size_t)n + 64, c->stream));
	}
	VK_HIP(hipMemcpyAsync(c->d_pos, pos, (size_t)n, mem == VK_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
	VK_HIP(hipStreamSynchronize(c->stream));
	return VK_OK;
}

static int set_slices_impl(vk_corpus_t *c, const int64_t *start, const int64_t *end, int64_t n_sentences, bool contiguous) {
	int max_len = 0, max_short = 0;
	int64_t n_long = 0;
	for (int64_t s = 0; s < n_sentences; s++) {
		const int64_t len = end[s] - start[s];
		if (start[s] < 0 || end[s] > c->desc.n_tokens || len < 0) return fail(VK_ERR_INVALID, "slice outside the token stream");
		if (s > 0 && (start[s] < start[s - 1] || end[s] < end[s - 1])) return fail(VK_ERR_INVALID, "slice starts and ends must be non-decreasing");
		if (len > VK_MAX_SENT_LEN) {
			char buf[128];
			snprintf(buf, sizeof buf, "slice %lld has %lld tokens; the HIP path handles at most %d", (long long)s, (long long)len, VK_MAX_SENT_LEN);
			return fail(VK_ERR_UNSUPPORTED, buf);
		}
		max_len = std::max(max_len, (int)len);
		if (len > VK_FAST_SENT_LEN) n_long++;
		else max_short = std::max(max_short, (int)len);
	}
	VK_HIP(hipSetDevice(c->device));

	// ---- the slice table.  Without long slices it is the caller's table.  A long slice gets a group of 4 rows
	// of its own ([L, empty, empty, empty]); the group before it is closed with empty rows, so that no group of
	// the main launch spans the tokens of a long slice.  Rows stay in slice order (ties are broken by row).
	std::vector<int32_t> st32, en32, long_groups;
	c->entry_sent.clear();
	if (n_long == 0) {
		st32.resize((size_t)n_sentences); en32.resize((size_t)n_sentences);
		for (int64_t s = 0; s < n_sentences; s++) { st32[(size_t)s] = (int32_t)start[s]; en32[(size_t)s] = (int32_t)end[s]; }
	} else {
		if (n_sentences + 3 * n_long + 3 >= (1ll << 31) - 16) return fail(VK_ERR_INVALID, "slice table too large");
		auto push = [&](int32_t a, int32_t b, int32_t sent) { st32.push_back(a); en32.push_back(b); c->entry_sent.push_back(sent); };
		for (int64_t s = 0; s < n_sentences; s++) {
			if (end[s] - start[s] > VK_FAST_SENT_LEN) {
				while (st32.size() % 4) push(en32.back(), en32.back(), -1);
				long_groups.push_back((int32_t)(st32.size() / 4));
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
	for (void *p : {(void *)c->d_sent_start, (void *)c->d_sent_end, (void *)c->d_scores, (void *)c->d_raw, (void *)c->d_keys[0], (void *)c->d_keys[1],
			(void *)c->d_boost, (void *)c->d_long_groups})
		if (p) VK_HIP(hipFree(p));
	c->d_sent_start = c->d_sent_end = nullptr; c->d_scores = c->d_raw = c->d_boost = nullptr; c->d_keys[0] = c->d_keys[1] = nullptr; c->d_long_groups = nullptr;
	int rc;
	if ((rc = alloc_t(c, &c->d_sent_start, st32.size()))) return rc;
	if ((rc = alloc_t(c, &c->d_sent_end, en32.size()))) return rc;
	if ((rc = alloc_t(c, &c->d_scores, (size_t)n_entries + 8))) return rc;
	if ((rc = alloc_t(c, &c->d_raw, (size_t)n_entries + 8))) return rc;
	const size_t nblk = (size_t)((n_entries + kTopkChunk - 1) / kTopkChunk) + 1;
	if ((rc = alloc_t(c, &c->d_keys[0], nblk * VK_MAX_MATCHES + kTopkChunk))) return rc;
	if ((rc = alloc_t(c, &c->d_keys[1], (nblk * VK_MAX_MATCHES) / 2 + 2 * kTopkChunk))) return rc;
	if (!long_groups.empty()) {
		if ((rc = alloc_t(c, &c->d_long_groups, long_groups.size()))) return rc;
		VK_HIP(hipMemcpy(c->d_long_groups, long_groups.data(), long_groups.size() * 4, hipMemcpyHostToDevice));
	}
	VK_HIP(hipMemcpy(c->d_sent_start, st32.data(), st32.size() * 4, hipMemcpyHostToDevice));
	VK_HIP(hipMemcpy(c->d_sent_end, en32.data(), en32.size() * 4, hipMemcpyHostToDevice));
	c->n_entries = n_entries;
	c->n_long_groups = (int)long_groups.size();
	c->max_len = max_len;
	c->max_short_len = max_short;
	c->contiguous = contiguous;
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
		if (li < long_groups.size() && long_groups[li] == g) {
			li++;
			lt = std::max(lt, tiles);
			ltok = std::max(ltok, (int)(b - a));
		} else {
			mt = std::max(mt, tiles);
			mtok = std::max(mtok, (int)(b - a));
		}
	}
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

static int validate_query(const vk_corpus *c, const vk_query_desc *q, const vk_topk_out *out) {
	if (!c || !q || !out) return fail(VK_ERR_INVALID, "null argument");
	if (!c->finalized) return fail(VK_ERR_STATE, "corpus not finalized");
	if (q->len_t < 1) return fail(VK_ERR_INVALID, "empty query");
	if (q->len_t > VK_MAX_QUERY_LEN) return fail(VK_ERR_UNSUPPORTED, "query longer than VK_MAX_QUERY_LEN (64) tokens");
	if (q->len_t > VK_FAST_QUERY_LEN) {
		if (q->algorithm == VK_ALG_WRD || (q->algorithm == VK_ALG_RWMD && q->wmd_full))
			return fail(VK_ERR_UNSUPPORTED, "exact transport (WRD, full WMD) is implemented for queries of at most 16 tokens");
		const int gm = q->algorithm == VK_ALG_RWMD ? 4 : (q->gap_s.kind == VK_GAP_TABLE || q->gap_t.kind == VK_GAP_TABLE) ? 2 : 1;
		if (vk_wide_lds_demand(c->max_len, (q->len_t + 15) / 16, gm, q->tag_weights != nullptr, q->want_flow) > 160 * 1024)
			return fail(VK_ERR_UNSUPPORTED, "query of more than 16 tokens over slices this long exceeds the LDS of a workgroup");
	}
	if (!q->q_vectors) return fail(VK_ERR_INVALID, "q_vectors is null");
	if (q->q_dtype != VK_F32 && q->q_dtype != VK_BF16) return fail(VK_ERR_INVALID, "bad q_dtype");
	if (q->max_matches < 1 || q->max_matches > VK_MAX_MATCHES) return fail(VK_ERR_INVALID, "max_matches out of range");
	if (out->capacity < q->max_matches) return fail(VK_ERR_INVALID, "output capacity smaller than max_matches");
	if (!out->score || !out->sentence) return fail(VK_ERR_INVALID, "output arrays missing");
	if (!(q->submatch_weight >= 0.0f)) return fail(VK_ERR_INVALID, "submatch_weight must be >= 0 (pow of a zero base, metric/alignment.h:97-99)");
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
		if (q->tag_weights) return fail(VK_ERR_UNSUPPORTED, "tag-weighted similarity is implemented for alignments only");
		if (q->rwmd_symmetric && !q->rwmd_normalize_bow)
			return fail(VK_ERR_INVALID, "cannot run symmetric mode WMD with bow (needs nbow)");   // wmd.h:441-449
		if (q->wmd_full) {
			if (c->max_len > VK_FAST_SENT_LEN) return fail(VK_ERR_UNSUPPORTED, "full WMD needs every slice <= VK_FAST_SENT_LEN (64) tokens");
			if (q->rwmd_injective) return fail(VK_ERR_INVALID, "non-relaxed WMD with injective mapping is not supported");      // wmd.h:201-204
			if (q->rwmd_symmetric) return fail(VK_ERR_INVALID, "non-relaxed WMD with symmetric computation is not supported");  // wmd.h:206-209
		} else if (!q->rwmd_injective && q->len_t > VK_FAST_QUERY_LEN)
			return fail(VK_ERR_UNSUPPORTED, "non-injective RWMD (rwmd('nbow/distributed')) is implemented for queries of at most 16 tokens");
		if (q->want_flow && (!out->mapping || !out->edge_sim)) return fail(VK_ERR_INVALID, "want_flow needs mapping and edge_sim arrays");
	} else if (q->algorithm == VK_ALG_WRD) {
		if (c->max_len > VK_FAST_SENT_LEN) return fail(VK_ERR_UNSUPPORTED, "VK_ALG_WRD needs every slice <= VK_FAST_SENT_LEN (64) tokens");
		if (q->tag_weights) return fail(VK_ERR_UNSUPPORTED, "tag-weighted similarity is implemented for alignments only");
		if (!c->d_mag) return fail(VK_ERR_STATE, "VK_ALG_WRD needs a corpus created with keep_magnitudes = 1");
		if (q->want_flow && (!out->mapping || !out->edge_sim)) return fail(VK_ERR_INVALID, "want_flow needs mapping and edge_sim arrays");
	} else {
		return fail(VK_ERR_INVALID, "bad algorithm");
	}
	return VK_OK;
}

// Vectors.normalized for the query rows, then bf16 (RNE), then tile order (16 rows,
// rows >= len_t zero).  Same arithmetic as oracle/vk_oracle.c vko_normalize_rows_bf16.
static void pack_query(const vk_corpus *c, const vk_query_desc *q, std::vector<uint8_t> &tile, float *mags) {
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

int vk_query(vk_corpus_t *c, const vk_query_desc *q, vk_topk_out *out) {
	int rc = validate_query(c, q, out);
	if (rc) return rc;
	VK_HIP(hipSetDevice(c->device));
	hipStream_t st = c->stream;
	const int64_t n = c->n_entries;           // rows of the slice table (== n_sentences unless long slices were padded)
	const int k = q->max_matches;
	out->n_out = 0;
	c->have_scores = false;
	if (n == 0) return VK_OK;
	auto sentence_of = [c](int64_t row) { return c->entry_sent.empty() ? row : (int64_t)c->entry_sent[(size_t)row]; };
	const bool is_static_l = c->desc.layout == VK_LAYOUT_STATIC;
	// transport algorithms: similarity rows (and, for exact transport, the optimal plan) of the winners, from which
	// the host states their SparseFlow / DenseFlow.  rows_idx: rows of the slice table, best first.
	auto transport_flows = [&](const std::vector<int64_t> &rows_idx, bool exact, const float *qmass, int mass_mode, int raw_masses) -> int {
		if (!q->want_flow || !out->sim_rows || rows_idx.empty()) return VK_OK;
		if (c->max_len > VK_FAST_SENT_LEN || q->len_t > VK_FAST_QUERY_LEN) return VK_OK;
		int rc2;
		if (!c->d_rows_out) {
			if ((rc2 = alloc_t(c, &c->d_rows_out, (size_t)VK_MAX_MATCHES * 64 * 16))) return rc2;
			if ((rc2 = alloc_t(c, &c->d_plan_out, (size_t)VK_MAX_MATCHES * 16 * 64))) return rc2;
		}
		if (!c->d_wrd_raw) {
			if ((rc2 = alloc_t(c, &c->d_wrd_raw, (size_t)VK_MAX_MATCHES))) return rc2;
			if ((rc2 = alloc_t(c, &c->d_wrd_val, (size_t)VK_MAX_MATCHES))) return rc2;
			c->wrd_cap = VK_MAX_MATCHES;
		}
		const int cnt = (int)rows_idx.size();
		std::vector<uint64_t> hk((size_t)cnt);
		for (int i = 0; i < cnt; i++) hk[(size_t)i] = (1ull << 32) | (uint64_t)(uint32_t)rows_idx[(size_t)i];
		VK_HIP(hipMemcpyAsync(c->d_keys[1], hk.data(), hk.size() * 8, hipMemcpyHostToDevice, c->stream));
		VkWrdParams w{};
		w.tiles = c->d_tiles; w.tok_id = c->d_tok_id; w.table = c->d_table; w.sent_start = c->d_sent_start; w.sent_end = c->d_sent_end;
		w.layout = is_static_l ? VK_DEV_LAYOUT_STATIC : VK_DEV_LAYOUT_CONTEXTUAL; w.nk32 = c->nk32; w.tail = c->tail; w.tile_bytes = c->tile_bytes; w.prec = c->prec;
		w.qtile = c->d_qtile; w.len_t = q->len_t; w.mag = c->d_mag;
		w.keys = c->d_keys[1]; w.rows_out = c->d_rows_out;
		VK_HIP(vk_launch_rows(&w, cnt, c->stream));
		VK_HIP(hipMemcpyAsync(out->sim_rows, c->d_rows_out, (size_t)cnt * 64 * 16 * 4, hipMemcpyDeviceToHost, c->stream));
		if (exact && out->plan) {
			w.mass_mode = mass_mode; w.raw_masses = raw_masses;
			memcpy(w.qmass, qmass, sizeof w.qmass);
			w.raw_out = c->d_wrd_raw; w.val_out = c->d_wrd_val; w.plan_out = c->d_plan_out;
			VK_HIP(vk_launch_wrd_exact(&w, cnt, nullptr, c->stream));
			VK_HIP(hipMemcpyAsync(out->plan, c->d_plan_out, (size_t)cnt * 16 * 64 * 4, hipMemcpyDeviceToHost, c->stream));
		}
		VK_HIP(hipStreamSynchronize(c->stream));
		return VK_OK;
	};

	// ---- prepare: query tile, gap tables, boost, static table -------------
	VK_HIP(hipEventRecord(c->ev[0], st));
	std::vector<uint8_t> qtile;
	float qmags[VK_MAX_QUERY_LEN] = {0};
	const bool wide = q->len_t > VK_FAST_QUERY_LEN;
	const int nq = (q->len_t + 15) / 16;
	pack_query(c, q, qtile, qmags);
	VK_HIP(hipMemcpyAsync(c->d_qtile, qtile.data(), qtile.size(), hipMemcpyHostToDevice, st));

	VkScoreParams p{};
	const int ks = q->gap_s.kind, kt = q->gap_t.kind;
	float ws[kGapTable], wt[80];
	const bool is_align = q->algorithm == VK_ALG_ALIGN;
	if (q->algorithm == VK_ALG_WRD) {
		p.gap_mode = 5;
		float sum_t = 0.0f;
		for (int j = 0; j < q->len_t; j++) sum_t += qmags[j];           // wrd.h:99-102, float sum in order
		const bool rawm = !q->wrd_normalize_magnitudes;   // wrd.h:99-102: masses stay the magnitudes
		for (int j = 0; j < VK_FAST_QUERY_LEN; j++) p.qmass[j] = j < q->len_t ? (rawm ? qmags[j] : qmags[j] / sum_t) : 0.0f;
		p.wrd_raw_total = rawm ? sum_t : 0.0f;
		p.mag = c->d_mag;
	} else if (q->algorithm == VK_ALG_RWMD) {
		p.gap_mode = 4;
		p.rwmd_symmetric = q->rwmd_symmetric;
		p.rwmd_normalize_bow = q->rwmd_normalize_bow;
		if (q->wmd_full) p.wmd_bound = q->rwmd_normalize_bow ? 1 : 2;
		else if (!q->rwmd_injective) {
			// 1:n form: masses of the query's vocabulary entries (count / len at the first occurrence of a token id)
			p.gap_mode = 7;
			const bool ids = c->desc.layout == VK_LAYOUT_STATIC && q->q_token_ids;
			for (int j = 0; j < VK_FAST_QUERY_LEN; j++) {
				float mass = 0.0f;
				if (j < q->len_t) {
					int cnt = 1;
					bool first = true;
					if (ids && q->q_token_ids[j] >= 0)
						for (int i = 0; i < q->len_t; i++)
							if (i != j && q->q_token_ids[i] == q->q_token_ids[j]) { cnt++; if (i < j) first = false; }
					mass = first ? (q->rwmd_normalize_bow ? (float)cnt / (float)q->len_t : (float)cnt) : 0.0f;
				}
				p.qmass[j] = mass;
			}
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
	for (int i = 0; i < kGapTable; i++) ws[i] = (is_align && i <= c->max_len) ? gap_cost(q->gap_s, i) : 0.0f;
	for (int i = 0; i < 80; i++) wt[i] = (is_align && i <= q->len_t) ? gap_cost(q->gap_t, i) : 0.0f;
	if (p.gap_mode == 2) {
		// register-history kernel: needs w_t strictly subadditive over the query length
		// (see dp_general_reg in vk_common.cuh); margin far above fp32 rounding of the DP values
		bool sub = true;
		for (int x = 1; x < q->len_t && sub; x++)
			for (int y = 1; x + y <= q->len_t; y++)
				if (!(wt[x] + wt[y] > wt[x + y] + 1e-4f)) { sub = false; break; }
		if (sub && !wide) p.gap_mode = c->max_short_len <= 32 ? 3 : 6;
	}
	VK_HIP(hipMemcpyAsync(c->d_ws, ws, sizeof ws, hipMemcpyHostToDevice, st));
	VK_HIP(hipMemcpyAsync(c->d_wt, wt, sizeof wt, hipMemcpyHostToDevice, st));

	std::vector<float> boost_rows;
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
		int32_t ids[80];
		for (int j = 0; j < 80; j++) ids[j] = (q->q_token_ids && j < q->len_t) ? q->q_token_ids[j] : -1;
		VK_HIP(hipMemcpyAsync(c->d_qids, ids, sizeof ids, hipMemcpyHostToDevice, st));
		for (int t = 0; t < nq; t++)   // one [V_pad x 16] table per 16 query tokens
			VK_HIP(vk_launch_table(c->d_tiles, c->d_qtile + (size_t)t * c->tile_bytes, (int32_t)c->n_tiles, c->nk32, c->tail, c->tile_bytes,
				c->d_table + t * table_stride, q->q_token_ids ? c->d_qids + t * 16 : nullptr, std::min(16, q->len_t - t * 16), c->desc.vocab_size, c->prec, st));
	}

	// ---- the fused scoring kernel ------------------------------------------
	// handles on one corpus take turns: this scoring kernel starts when the peer's has finished (its selection and
	// traceback then run beside this kernel); the wait is on the device, the host does not block
	if (c->peer && c->peer->ev2_recorded) VK_HIP(hipStreamWaitEvent(st, c->peer->ev[2], 0));
	VK_HIP(hipEventRecord(c->ev[1], st));
	p.tiles = c->d_tiles; p.tok_id = c->d_tok_id; p.table = c->d_table; p.sent_start = c->d_sent_start; p.sent_end = c->d_sent_end;
	p.n_sent = (int32_t)n; p.layout = is_static ? VK_DEV_LAYOUT_STATIC : VK_DEV_LAYOUT_CONTEXTUAL;
	p.nk32 = c->nk32; p.tail = c->tail; p.tile_bytes = c->tile_bytes; p.prec = c->prec;
	p.qtile = c->d_qtile; p.len_t = q->len_t; p.locality = q->locality;
	p.ws = c->d_ws; p.wt = c->d_wt;
	p.boost = q->boost ? c->d_boost : nullptr;
	p.scores = c->d_scores; p.raw = c->d_raw;
	p.ref_total = (float)q->len_t;
	if (q->tag_weights && is_align) {
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
	VkWideParams wp{};
	if (wide) {
		wp.tiles = c->d_tiles; wp.tok_id = c->d_tok_id; wp.table = c->d_table; wp.table_stride = table_stride;
		wp.sent_start = c->d_sent_start; wp.sent_end = c->d_sent_end; wp.n_sent = (int32_t)n; wp.layout = p.layout;
		wp.nk32 = c->nk32; wp.tail = c->tail; wp.tile_bytes = c->tile_bytes; wp.prec = c->prec;
		wp.qtile = c->d_qtile; wp.nq = nq; wp.len_t = q->len_t; wp.locality = q->locality; wp.gap_mode = p.gap_mode; wp.max_len = c->max_len;
		wp.rwmd_symmetric = p.rwmd_symmetric; wp.rwmd_normalize_bow = p.rwmd_normalize_bow;
		wp.gs = p.gs; wp.gt = p.gt; wp.a_s = p.a_s; wp.a_t = p.a_t; wp.open_s = p.open_s; wp.open_t = p.open_t;
		wp.ws = c->d_ws; wp.wt = c->d_wt;
		wp.pos_s = p.pos_s; wp.tw_keep = p.tw_keep; wp.tw_threshold = p.tw_threshold; wp.ref_total = p.ref_total;
		for (int j = 0; j < VK_MAX_QUERY_LEN; j++) {
			wp.tw[j] = (p.pos_s && j < q->len_t) ? q->tag_weights[j] : 0.0f;
			wp.tpos[j] = (p.pos_s && j < q->len_t) ? (int32_t)q->q_pos[j] : -1;
		}
		wp.boost = p.boost; wp.scores = c->d_scores; wp.raw = c->d_raw;
		VK_HIP(vk_launch_wide(&wp, 0, st));
	} else if (is_align && !is_static && q->len_t == 1 && c->uniform_len == 1 && q->locality == VK_LOCAL && !p.pos_s) {
		// span-embedding index: one vector per slice, one query vector -> the clipped cosine is the local alignment score
		VK_HIP(vk_launch_span(&p, st));
	} else {
	p.max_short_len = VK_FAST_SENT_LEN;
	p.s_rows_per_wave = is_static ? (c->max_group_tokens + 15) / 16 * 16 : c->max_group_tiles * 16;
	p.h_rows = c->max_short_len + 1;
	const int lt = q->len_t <= 4 ? 4 : q->len_t <= 8 ? 8 : q->len_t <= 12 ? 12 : 16;   // strip rows hold the padded query columns (launch_score_lt)
	int lds_floats = p.s_rows_per_wave * lt + 16;
	if (p.gap_mode == 2) lds_floats += 4 * p.h_rows * 16;   // column history of dp_general
	p.m_rows = (c->max_short_len + 4) / 4 * 4;
	if (p.gap_mode == 7) lds_floats += 4 * p.m_rows;       // vocabulary masses of the 4 slices (static layout)
	p.lds_floats_per_wave = lds_floats;
	size_t smem = (size_t)lds_floats * 4 * 4;   // 4 waves per block
	const size_t qlds = (!is_static && c->prec == 0 && c->nk32 == 24 && c->tail == 0) ? (size_t)c->nk32 * 1024 : 0;   // MODE 3: query tile in LDS
	smem += qlds;
	if (smem > 160 * 1024) return fail(VK_ERR_UNSUPPORTED, "LDS demand exceeds 160 KiB per workgroup");
	const int64_t n_groups = (n + 3) / 4;
	const int grid = (int)std::min<int64_t>((n_groups + 3) / 4, (int64_t)1 << 20);   // capped to residency by the launcher
	VK_HIP(vk_launch_score(&p, grid, smem, st));
	if (c->n_long_groups > 0) {
		// slices longer than VK_FAST_SENT_LEN: one per wave, one wave per workgroup, LDS strip for the longest;
		// general gaps take the LDS-history form (the four DPP rows share one history: only row 0 is active)
		VkScoreParams pl = p;
		pl.group_list = c->d_long_groups; pl.n_list = c->n_long_groups;
		if (pl.gap_mode == 3 || pl.gap_mode == 6) pl.gap_mode = 2;
		pl.s_rows_per_wave = is_static ? (c->long_group_tokens + 15) / 16 * 16 : c->long_group_tiles * 16;
		pl.h_rows = 0;
		int lf = pl.s_rows_per_wave * lt + 16;
		if (pl.gap_mode == 2) lf += (c->max_len + 1) * 16;
		pl.m_rows = 0;
		if (pl.gap_mode == 7) lf += (c->max_len + 4) / 4 * 4;
		pl.lds_floats_per_wave = lf;
		const size_t smem_l = (size_t)lf * 4 + qlds;
		if (smem_l > 160 * 1024) return fail(VK_ERR_UNSUPPORTED, "LDS demand of the long-slice pass exceeds 160 KiB");
		VK_HIP(vk_launch_score(&pl, c->n_long_groups, smem_l, st));
	}
	}

	const bool exact_transport = q->algorithm == VK_ALG_WRD || (q->algorithm == VK_ALG_RWMD && q->wmd_full);
	if (exact_transport) {
		// ---- stage 2: exact EMD on the candidates with the largest bounds, until the k-th best
		// exact score is above every remaining bound (then no unsolved sentence can enter)
		VK_HIP(hipEventRecord(c->ev[2], st));
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
		std::vector<uint64_t> keys;
		std::vector<float> vals, raws;
		VkWrdParams w{};
		w.tiles = c->d_tiles; w.tok_id = c->d_tok_id; w.table = c->d_table; w.sent_start = c->d_sent_start; w.sent_end = c->d_sent_end;
		w.layout = p.layout; w.nk32 = c->nk32; w.tail = c->tail; w.tile_bytes = c->tile_bytes; w.prec = c->prec;
		w.qtile = c->d_qtile; w.len_t = q->len_t; w.mag = c->d_mag;
		w.mass_mode = q->algorithm == VK_ALG_WRD ? 0 : (q->rwmd_normalize_bow ? 1 : 2);
		memcpy(w.qmass, p.qmass, sizeof w.qmass);
		w.raw_masses = (q->algorithm == VK_ALG_WRD && !q->wrd_normalize_magnitudes) ? 1 : 0;
		w.boost = p.boost; w.raw_out = c->d_wrd_raw; w.val_out = c->d_wrd_val;
		// solves the `count` candidates whose keys sit at d_keys, merges them into `best`; returns the smallest bound among them
		auto solve = [&](const uint64_t *d_keys, int count, float *ub_min, int *n_cand_out) -> int {
			w.keys = d_keys;
			VK_HIP(vk_launch_wrd_exact(&w, count, c->d_scores, st));
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
				uint32_t count = 0;
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
		if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) t.prepare_ms = ms;
		if (hipEventElapsedTime(&ms, c->ev[1], c->ev[2]) == hipSuccess) t.score_ms = ms;
		if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) t.topk_ms = ms;
		if (hipEventElapsedTime(&ms, c->ev[0], c->ev[4]) == hipSuccess) t.total_ms = ms;
		c->last = t;
		return VK_OK;
	}

	// ---- flow (traceback) of `count` slices named by device keys: narrow or wide kernel
	const int ostride = wide ? 64 : 16;   // row stride of the mapping / edge_sim device arrays
	auto launch_flow = [&](const uint64_t *d_keys, int count) -> int {
		if (wide) {
			wp.keys = d_keys; wp.raw_out = c->d_out_raw; wp.mapping = c->d_out_map; wp.edge_sim = c->d_out_sim;
			VK_HIP(vk_launch_wide(&wp, count, st));
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
		f.keys = d_keys; f.raw_out = c->d_out_raw; f.mapping = c->d_out_map; f.edge_sim = c->d_out_sim;
		VK_HIP(vk_launch_flow(&f, count, st));
		return VK_OK;
	};

	if (is_align && q->submatch_weight != 0.0f) {
		// ---- submatch_weight: bound from raw, then exact scores of the candidates from their tracebacks, until the
		// k-th best exact score is above every remaining bound (vk_submatch_bound_kernel)
		VK_HIP(hipEventRecord(c->ev[2], st));
		const float wsub = q->submatch_weight, total = p.ref_total;
		const float m_star = total * (1.0f - powf(1.0f / (wsub + 1.0f), 1.0f / wsub));
		VK_HIP(vk_launch_submatch_bound(c->d_raw, p.boost, n, total, wsub, m_star, c->d_scores, st));
		const int M = 512;
		struct Cand { float val, raw; int64_t row; std::vector<int16_t> map; std::vector<float> sim; };
		std::vector<Cand> best;
		std::vector<uint64_t> keys((size_t)M);
		std::vector<float> raws((size_t)M), sims((size_t)M * ostride);
		std::vector<int16_t> maps((size_t)M * ostride);
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
		float ms = 0;
		vk_timings t{};
		if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) t.prepare_ms = ms;
		if (hipEventElapsedTime(&ms, c->ev[1], c->ev[2]) == hipSuccess) t.score_ms = ms;
		if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) t.topk_ms = ms;
		if (hipEventElapsedTime(&ms, c->ev[0], c->ev[4]) == hipSuccess) t.total_ms = ms;
		c->last = t;
		return VK_OK;
	}

	// ---- bounded result set -------------------------------------------------
	VK_HIP(hipEventRecord(c->ev[2], st));
	c->ev2_recorded = true;
	int cur = 0;
	if (k <= 64) {
		// wave-streaming selection: n -> ceil(n/4096) * k keys -> ... -> k keys
		int64_t nw = 0;
		VK_HIP(vk_launch_topk_wave(c->d_scores, nullptr, n, q->min_score, k, 4096, c->d_keys[0], &nw, st));
		while (nw > 1) {
			const int64_t nkeys = nw * k;
			const int64_t per_wave = nkeys <= 16384 ? nkeys : 4096;
			VK_HIP(vk_launch_topk_wave(nullptr, c->d_keys[cur], nkeys, 0.0f, k, per_wave, c->d_keys[1 - cur], &nw, st));
			cur = 1 - cur;
		}
	} else {
		int nb = 0;
		VK_HIP(vk_launch_topk_scores(c->d_scores, n, q->min_score, k, c->d_keys[0], &nb, st));
		while (nb > 1) {
			const int64_t nkeys = (int64_t)nb * k;
			VK_HIP(vk_launch_topk_keys(c->d_keys[cur], nkeys, k, c->d_keys[1 - cur], &nb, st));
			cur = 1 - cur;
		}
	}

	// ---- flow of the winners ------------------------------------------------
	VK_HIP(hipEventRecord(c->ev[3], st));
	const bool do_flow = q->want_flow && is_align;
	if (do_flow && (rc = launch_flow(c->d_keys[cur], k))) return rc;
	VK_HIP(hipEventRecord(c->ev[4], st));

	// ---- results to host ------------------------------------------------------
	std::vector<uint64_t> keys((size_t)k);
	std::vector<float> raw((size_t)k), sim((size_t)k * ostride);
	std::vector<int16_t> map((size_t)k * ostride);
	VK_HIP(hipMemcpyAsync(keys.data(), c->d_keys[cur], (size_t)k * 8, hipMemcpyDeviceToHost, st));
	if (do_flow) {
		VK_HIP(hipMemcpyAsync(raw.data(), c->d_out_raw, (size_t)k * 4, hipMemcpyDeviceToHost, st));
		VK_HIP(hipMemcpyAsync(map.data(), c->d_out_map, map.size() * 2, hipMemcpyDeviceToHost, st));
		VK_HIP(hipMemcpyAsync(sim.data(), c->d_out_sim, sim.size() * 4, hipMemcpyDeviceToHost, st));
	}
	VK_HIP(hipStreamSynchronize(st));

	int n_out = 0;
	for (int i = 0; i < k; i++) {
		if (keys[(size_t)i] == 0) break;
		n_out++;
	}
	std::vector<float> raw_sel((size_t)std::max(n_out, 1));
	if (!do_flow && out->raw_score && n_out > 0) {
		// gather the aligner scores of the winners
		for (int i = 0; i < n_out; i++) {
			const int64_t g = (int64_t)(uint32_t)(keys[(size_t)i] & 0xffffffffu);
			VK_HIP(hipMemcpyAsync(&raw_sel[(size_t)i], c->d_raw + g, 4, hipMemcpyDeviceToHost, st));
		}
		VK_HIP(hipStreamSynchronize(st));
	}
	for (int i = 0; i < n_out; i++) {
		const uint64_t key = keys[(size_t)i];
		const uint32_t ob = (uint32_t)(key >> 32);
		const uint32_t bits = (ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob;
		float s;
		memcpy(&s, &bits, 4);
		out->score[i] = s;
		out->sentence[i] = sentence_of((int64_t)(uint32_t)(key & 0xffffffffu));
		if (out->raw_score) out->raw_score[i] = do_flow ? raw[(size_t)i] : raw_sel[(size_t)i];
		if (do_flow) {
			for (int j = 0; j < q->len_t; j++) {
				out->mapping[(size_t)i * q->len_t + j] = map[(size_t)i * ostride + j];
				out->edge_sim[(size_t)i * q->len_t + j] = sim[(size_t)i * ostride + j];
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
	c->have_scores = true;
	if (q->algorithm == VK_ALG_RWMD && n_out > 0) {
		std::vector<int64_t> rows_idx;
		for (int i = 0; i < n_out; i++) rows_idx.push_back((int64_t)(uint32_t)(keys[(size_t)i] & 0xffffffffu));
		float no_mass[VK_FAST_QUERY_LEN] = {0};
		if ((rc = transport_flows(rows_idx, false, no_mass, 0, 0))) return rc;
	}

	float ms = 0;
	vk_timings t{};
	if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) t.prepare_ms = ms;
	if (hipEventElapsedTime(&ms, c->ev[1], c->ev[2]) == hipSuccess) t.score_ms = ms;
	if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) t.topk_ms = ms;
	if (hipEventElapsedTime(&ms, c->ev[3], c->ev[4]) == hipSuccess) t.flow_ms = ms;
	if (hipEventElapsedTime(&ms, c->ev[0], c->ev[4]) == hipSuccess) t.total_ms = ms;
	c->last = t;
	return VK_OK;
}

static bool same_gap(const vk_gap &a, const vk_gap &b, int upto) {
	if (a.kind != b.kind) return false;
	if (a.kind != VK_GAP_TABLE) return a.u == b.u && (a.kind == VK_GAP_LINEAR || a.v == b.v);
	if (a.table == b.table && a.n_table == b.n_table) return true;
	for (int k = 0; k <= upto; k++) if (gap_cost(a, k) != gap_cost(b, k)) return false;
	return true;
}

// Queries with common options over a contextual corpus: up to 4 queries share one pass over the token tiles
// (vk_score_batch_kernel).  Returns VK_ERR_UNSUPPORTED (without setting an error) when the batch does not qualify;
// the caller then runs the queries one by one.
static int query_batch_shared_pass(vk_corpus_t *c, const vk_query_desc *qs, int32_t n_queries, vk_topk_out *outs) {
	if (n_queries < 2 || !c->finalized || c->prec != 0 || c->desc.layout != VK_LAYOUT_CONTEXTUAL || c->n_long_groups > 0 || c->desc.n_sentences < 1) return VK_ERR_UNSUPPORTED;
	if (c->nk32 > 10 && !getenv("VK_BATCH_QB")) return VK_ERR_UNSUPPORTED;   // measured: no gain over single queries for 768-d rows (the kernel pipelines tiles of <= 10 K-steps)
	const vk_query_desc &q0 = qs[0];
	if (q0.max_matches > 64) return VK_ERR_UNSUPPORTED;
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
		if ((rc = validate_query(c, &qs[i], &outs[i]))) return rc;
	VK_HIP(hipSetDevice(c->device));
	hipStream_t st = c->stream;
	const int64_t n = c->n_entries;
	const int k = q0.max_matches;
	const bool is_align = q0.algorithm == VK_ALG_ALIGN;

	// ---- common options: gap tables, DP form
	VkScoreBatchParams p{};
	const int ks = q0.gap_s.kind, kt = q0.gap_t.kind;
	float ws[kGapTable], wt[80];
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
	if (p.gap_mode == 2) {
		bool sub = true;   // see vk_query: register-history DP needs w_t strictly subadditive
		for (int x = 1; x < max_len_t && sub; x++)
			for (int y = 1; x + y <= max_len_t; y++)
				if (!(wt[x] + wt[y] > wt[x + y] + 1e-4f)) { sub = false; break; }
		if (sub) p.gap_mode = c->max_short_len <= 32 ? 3 : 6;
	}
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
		if (c->d_bq) { VK_HIP(hipFree(c->d_bq)); VK_HIP(hipFree(c->d_bqlen)); }
		if ((rc = alloc_t(c, &c->d_bq, need_q))) return rc;
		if ((rc = alloc_t(c, &c->d_bqlen, 2 * ((size_t)4 + 4)))) return rc;
		c->bq_cap = need_q;
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
	const size_t need_k = (size_t)4 * (size_t)nw1 * (size_t)k;
	if (c->bkeys_cap < need_k) {
		for (auto &b : c->d_bkeys) if (b) VK_HIP(hipFree(b));
		if ((rc = alloc_t(c, &c->d_bkeys[0], need_k))) return rc;
		if ((rc = alloc_t(c, &c->d_bkeys[1], need_k))) return rc;
		c->bkeys_cap = need_k;
	}
	VK_HIP(hipMemcpyAsync(c->d_ws, ws, sizeof ws, hipMemcpyHostToDevice, st));
	VK_HIP(hipMemcpyAsync(c->d_wt, wt, sizeof wt, hipMemcpyHostToDevice, st));
	std::vector<float> boost_rows;
	if (q0.boost) {
		if (!c->d_boost) { rc = alloc_t(c, &c->d_boost, (size_t)n + 8); if (rc) return rc; }
		VK_HIP(hipMemcpyAsync(c->d_boost, q0.boost, (size_t)n * 4, hipMemcpyHostToDevice, st));   // no long slices: rows == slices
	}
	p.tiles = c->d_tiles; p.sent_start = c->d_sent_start; p.sent_end = c->d_sent_end; p.n_sent = (int32_t)n;
	p.nk32 = c->nk32; p.tail = c->tail; p.tile_bytes = c->tile_bytes;
	p.qtiles = c->d_bq; p.locality = q0.locality; p.ws = c->d_ws; p.wt = c->d_wt;
	p.boost = q0.boost ? c->d_boost : nullptr; p.scores = c->d_bscores; p.raw = c->d_braw;

	float score_ms_total = 0.0f, total_ms = 0.0f;
	std::vector<uint8_t> all(need_q), one;
	float mags[VK_MAX_QUERY_LEN];
	for (int base = 0; base < n_queries; base += qb_max) {
		const int qb = std::min(qb_max, n_queries - base);
		VK_HIP(hipEventRecord(c->ev[0], st));
		std::fill(all.begin(), all.end(), 0);
		for (int i = 0; i < qb; i++) {
			pack_query(c, &qs[base + i], one, mags);
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
		VK_HIP(vk_launch_topk_wave_batch(c->d_bscores, nullptr, n, q0.min_score, k, 4096, qb, n, nw1 * k, c->d_bkeys[0], &nw, st));
		const int64_t stride = nw1 * k;
		while (nw > 1) {
			const int64_t nkeys = nw * k;
			const int64_t per_wave = nkeys <= 16384 ? nkeys : 4096;
			VK_HIP(vk_launch_topk_wave_batch(nullptr, c->d_bkeys[cur], nkeys, 0.0f, k, per_wave, qb, stride, stride, c->d_bkeys[1 - cur], &nw, st));
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
				f.keys = c->d_bkeys[cur] + (size_t)i * stride;
				f.raw_out = c->d_out_raw + (size_t)i * k; f.mapping = c->d_out_map + (size_t)i * k * 16; f.edge_sim = c->d_out_sim + (size_t)i * k * 16;
				VK_HIP(vk_launch_flow(&f, k, st));
			}
		}
		VK_HIP(hipEventRecord(c->ev[4], st));
		std::vector<uint64_t> keys((size_t)qb * k);
		std::vector<float> raw((size_t)qb * k), sim((size_t)qb * k * 16);
		std::vector<int16_t> map((size_t)qb * k * 16);
		VK_HIP(hipMemcpy2DAsync(keys.data(), (size_t)k * 8, c->d_bkeys[cur], (size_t)stride * 8, (size_t)k * 8, (size_t)qb, hipMemcpyDeviceToHost, st));
		if (do_flow) {
			VK_HIP(hipMemcpyAsync(raw.data(), c->d_out_raw, raw.size() * 4, hipMemcpyDeviceToHost, st));
			VK_HIP(hipMemcpyAsync(map.data(), c->d_out_map, map.size() * 2, hipMemcpyDeviceToHost, st));
			VK_HIP(hipMemcpyAsync(sim.data(), c->d_out_sim, sim.size() * 4, hipMemcpyDeviceToHost, st));
		}
		VK_HIP(hipStreamSynchronize(st));
		for (int i = 0; i < qb; i++) {
			const vk_query_desc &q = qs[base + i];
			vk_topk_out *out = &outs[base + i];
			int n_out = 0;
			for (int j = 0; j < k; j++) {
				const uint64_t key = keys[(size_t)i * k + j];
				if (key == 0) break;
				const uint32_t ob = (uint32_t)(key >> 32);
				const uint32_t bits = (ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob;
				float sc;
				memcpy(&sc, &bits, 4);
				const int64_t g = (int64_t)(uint32_t)(key & 0xffffffffu);
				out->score[j] = sc;
				out->sentence[j] = g;
				if (out->raw_score) {
					if (do_flow) out->raw_score[j] = raw[(size_t)i * k + j];
					else VK_HIP(hipMemcpy(&out->raw_score[j], c->d_braw + (size_t)i * n + g, 4, hipMemcpyDeviceToHost));
				}
				if (q.want_flow && out->mapping && out->edge_sim)
					for (int t = 0; t < q.len_t; t++) {
						out->mapping[(size_t)j * q.len_t + t] = do_flow ? map[((size_t)i * k + j) * 16 + t] : (int16_t)-1;
						out->edge_sim[(size_t)j * q.len_t + t] = do_flow ? sim[((size_t)i * k + j) * 16 + t] : 0.0f;
					}
				n_out++;
			}
			out->n_out = n_out;
		}
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

int vk_query_batch(vk_corpus_t *c, const vk_query_desc *qs, int32_t n_queries, vk_topk_out *outs) {
	if (!c || !qs || !outs || n_queries < 0) return fail(VK_ERR_INVALID, "null argument");
	if (n_queries == 0) return VK_OK;
	// the GEMM path: injective RWMD, contextual layout, one sentence length (multiple of 16), common options
	bool gemm = c->finalized && c->prec == 0 && c->contiguous && c->desc.layout == VK_LAYOUT_CONTEXTUAL && c->uniform_len > 0 && c->uniform_len % 16 == 0 &&
		c->uniform_len <= 64 && c->desc.n_sentences > 0 && qs[0].max_matches <= 64 &&
		((c->nk32 == 10 && c->tail == 1) || (c->nk32 == 4 && c->tail == 0));
	for (int i = 0; i < n_queries && gemm; i++) {
		const vk_query_desc &q = qs[i];
		gemm = q.len_t <= VK_FAST_QUERY_LEN && q.algorithm == VK_ALG_RWMD && q.rwmd_injective && !q.wmd_full && q.rwmd_symmetric == qs[0].rwmd_symmetric &&
			q.rwmd_normalize_bow == qs[0].rwmd_normalize_bow && q.max_matches == qs[0].max_matches &&
			q.min_score == qs[0].min_score && q.boost == qs[0].boost && !q.tag_weights;
	}
	if (!gemm) {
		const int rcb = query_batch_shared_pass(c, qs, n_queries, outs);
		if (rcb != VK_ERR_UNSUPPORTED) return rcb;
		for (int i = 0; i < n_queries; i++) {
			const int rc = vk_query(c, &qs[i], &outs[i]);
			if (rc) return rc;
		}
		return VK_OK;
	}
	for (int i = 0; i < n_queries; i++) {
		const int rc = validate_query(c, &qs[i], &outs[i]);
		if (rc) return rc;
	}
	VK_HIP(hipSetDevice(c->device));
	hipStream_t st = c->stream;
	const int64_t n = c->desc.n_sentences;
	const int k = qs[0].max_matches;
	int rc;

	// 32-token sentences take the 32x32x16 kernel: 3 queries of <= 10 tokens (else 2 of <= 16) share one 32-row A tile
	const bool b32 = c->uniform_len == 32;
	int qpt = 3;
	for (int i = 0; i < n_queries; i++) if (qs[i].len_t > 10) qpt = 2;
	const int nk16 = c->d_pad / 16;
	const int n_qtiles = (n_queries + qpt - 1) / qpt;

	// ---- device buffers (kept for the next batch)
	// (+1: the kernel prefetches one tile past the last)
	const size_t need_q = std::max((size_t)n_queries * c->tile_bytes, b32 ? (size_t)(n_qtiles + 1) * nk16 * 1024 : (size_t)0);
	if (c->bq_cap < need_q) {
		if (c->d_bq) { VK_HIP(hipFree(c->d_bq)); VK_HIP(hipFree(c->d_bqlen)); }
		if ((rc = alloc_t(c, &c->d_bq, need_q))) return rc;
		if ((rc = alloc_t(c, &c->d_bqlen, 2 * ((size_t)n_queries + 4)))) return rc;   // lengths, then their reciprocals
		c->bq_cap = need_q;
	}
	const size_t need_s = (size_t)n_queries * (size_t)n;
	if (c->bscores_cap < need_s) {
		if (c->d_bscores) VK_HIP(hipFree(c->d_bscores));
		if ((rc = alloc_t(c, &c->d_bscores, need_s))) return rc;
		c->bscores_cap = need_s;
	}
	const int64_t nw1 = (n + 4095) / 4096;
	const size_t need_k = (size_t)n_queries * (size_t)nw1 * (size_t)k;
	if (c->bkeys_cap < need_k) {
		for (auto &b : c->d_bkeys) if (b) VK_HIP(hipFree(b));
		if ((rc = alloc_t(c, &c->d_bkeys[0], need_k))) return rc;
		if ((rc = alloc_t(c, &c->d_bkeys[1], need_k))) return rc;
		c->bkeys_cap = need_k;
	}

	VK_HIP(hipEventRecord(c->ev[0], st));
	std::vector<uint8_t> all((size_t)need_q, 0), one;
	std::vector<int32_t> qlen((size_t)n_queries);
	float mags[VK_MAX_QUERY_LEN];
	for (int i = 0; i < n_queries; i++) {
		pack_query(c, &qs[i], one, mags);
		qlen[(size_t)i] = qs[i].len_t;
		if (!b32) {
			memcpy(all.data() + (size_t)i * c->tile_bytes, one.data(), one.size());
			continue;
		}
		// A tile of v_mfma_f32_32x32x16_bf16: K-step t = 1 KiB, lane l = 32 (k >> 3 & 1) + M owns row M, 8 features.
		// Row M of the result lands in accumulator register acc = 4 (M >> 3) + (M & 3) of lane half hd = M >> 2 & 1;
		// the kernel (vk_rwmd_batch32_kernel) expects query tokens at (hd, acc) as laid out below.
		uint8_t *dst = all.data() + (size_t)(i / qpt) * nk16 * 1024;
		const int slot = i % qpt;
		for (int j = 0; j < qs[i].len_t; j++) {
			int hd, acc;
			if (qpt == 2) { hd = slot; acc = j; }
			else if (slot < 2) { hd = slot; acc = j; }
			else { hd = j / 5; acc = 10 + j % 5; }
			const int M = 8 * (acc >> 2) + 4 * hd + (acc & 3);
			for (int k = 0; k < c->d_pad; k += 8) {   // 8 features = one 16-byte piece in both layouts
				const size_t src = (size_t)(k >> 5) * 1024 + (size_t)(((k & 31) >> 3) * 16 + j) * 16;
				const size_t off = (size_t)(k >> 4) * 1024 + (size_t)(((k >> 3) & 1) * 32 + M) * 16;
				memcpy(dst + off, one.data() + src, 16);
			}
		}
	}
	VK_HIP(hipMemcpyAsync(c->d_bq, all.data(), all.size(), hipMemcpyHostToDevice, st));
	VK_HIP(hipMemcpyAsync(c->d_bqlen, qlen.data(), qlen.size() * 4, hipMemcpyHostToDevice, st));
	std::vector<float> qinv((size_t)n_queries);
	for (int i = 0; i < n_queries; i++) qinv[(size_t)i] = 1.0f / (float)qs[i].len_t;
	float *d_qinv = reinterpret_cast<float *>(c->d_bqlen + n_queries + 4);
	VK_HIP(hipMemcpyAsync(d_qinv, qinv.data(), qinv.size() * 4, hipMemcpyHostToDevice, st));
	if (qs[0].boost) {
		if (!c->d_boost) { rc = alloc_t(c, &c->d_boost, (size_t)n + 8); if (rc) return rc; }
		VK_HIP(hipMemcpyAsync(c->d_boost, qs[0].boost, (size_t)n * 4, hipMemcpyHostToDevice, st));
	}

	VK_HIP(hipEventRecord(c->ev[1], st));
	VkRwmdBatchParams p{};
	p.tiles = c->d_tiles; p.n_tiles = (c->desc.n_tokens + 15) / 16;
	p.tile_bytes = c->tile_bytes; p.nk = c->nk32; p.half = c->tail;
	p.qtiles = c->d_bq; p.q_len = c->d_bqlen; p.n_queries = n_queries; p.n_sent = (int32_t)n;
	p.tiles_per_sent = c->uniform_len / 16;
	p.symmetric = qs[0].rwmd_symmetric; p.nbow = qs[0].rwmd_normalize_bow;
	p.boost = qs[0].boost ? c->d_boost : nullptr;
	p.scores = c->d_bscores;
	p.n_qtiles = n_qtiles; p.qpt = qpt; p.q_inv_len = d_qinv;
	p.late_mask = 4;   // waves w and w + 4 of a workgroup share a SIMD
	if (const char *e = getenv("VK_BATCH32_LATE_MASK")) p.late_mask = atoi(e);   // tuning aid
	if (b32) VK_HIP(vk_launch_rwmd_batch32(&p, st));
	else VK_HIP(vk_launch_rwmd_batch(&p, st));

	VK_HIP(hipEventRecord(c->ev[2], st));
	int64_t nw = 0;
	int cur = 0;
	VK_HIP(vk_launch_topk_wave_batch(c->d_bscores, nullptr, n, qs[0].min_score, k, 4096, n_queries, n, nw1 * k, c->d_bkeys[0], &nw, st));
	int64_t stride = nw1 * k;
	while (nw > 1) {
		const int64_t nkeys = nw * k;
		const int64_t per_wave = nkeys <= 16384 ? nkeys : 4096;
		VK_HIP(vk_launch_topk_wave_batch(nullptr, c->d_bkeys[cur], nkeys, 0.0f, k, per_wave, n_queries, stride, stride, c->d_bkeys[1 - cur], &nw, st));
		cur = 1 - cur;
	}
	VK_HIP(hipEventRecord(c->ev[3], st));
	VK_HIP(hipEventRecord(c->ev[4], st));
	std::vector<uint64_t> keys((size_t)n_queries * (size_t)k);
	VK_HIP(hipMemcpy2DAsync(keys.data(), (size_t)k * 8, c->d_bkeys[cur], (size_t)stride * 8, (size_t)k * 8, (size_t)n_queries, hipMemcpyDeviceToHost, st));
	VK_HIP(hipStreamSynchronize(st));
	for (int i = 0; i < n_queries; i++) {
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
	c->have_scores = false;
	float ms = 0;
	vk_timings t{};
	if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) t.prepare_ms = ms;
	if (hipEventElapsedTime(&ms, c->ev[1], c->ev[2]) == hipSuccess) t.score_ms = ms;
	if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) t.topk_ms = ms;
	if (hipEventElapsedTime(&ms, c->ev[0], c->ev[4]) == hipSuccess) t.total_ms = ms;
	c->last = t;
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

int vk_merge_topk(const vk_topk_out *sets, int32_t n_sets, int32_t len_t, int32_t max_matches, vk_topk_out *out) {
	if (!sets || !out || n_sets < 0) return fail(VK_ERR_INVALID, "null argument");
	if (max_matches < 1 || out->capacity < max_matches) return fail(VK_ERR_INVALID, "output capacity smaller than max_matches");
	struct Ref { float score; int64_t sent; int set, idx; };
	std::vector<Ref> all;
	for (int s = 0; s < n_sets; s++)
		for (int i = 0; i < sets[s].n_out; i++) all.push_back({sets[s].score[i], sets[s].sentence[i], s, i});
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

} // extern "C"
