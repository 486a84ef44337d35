// vk_internal.h -- shared by the host-side units of the C-ABI (vk_corpus.cpp, vk_query.cpp, vk_batch.cpp):
// error reporting, small conversions, the corpus handle.  Internal; the public interface is include/vectorian_hip.h.
#ifndef VK_INTERNAL_H
#define VK_INTERNAL_H

#include "../../include/vectorian_hip.h"
#include "vk_device.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

std::string &vk_error_slot();   // the calling thread's last error message (vk_corpus.cpp)

namespace {

int fail(int code, const std::string &msg) {
	vk_error_slot() = msg;
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

// Device arrays that several handles read -- a corpus, its views (vk_corpus_view), the filtered corpora of a static layout
// (vk_corpus_filter shares the vocabulary vectors): owned by a refcounted block, freed when the LAST handle that names them is
// freed.  The order in which a caller (or a garbage collector) frees the handles of one corpus therefore cannot matter; the reference
// keeps its results alive the same way (shared_ptr graph, vectorian/core/cpp/result_set.h:17-30).
struct vk_devblock {
	int device = 0;
	std::mutex mu;
	std::vector<void *> ptrs;
	void add(void *p) { std::lock_guard<std::mutex> g(mu); ptrs.push_back(p); }
	void release(void *p);   // free one array now (the slice table is re-created when the slices are set again before finalize)
	~vk_devblock();
};

// The ring of the handles on one corpus (vk_corpus::peer) is read by every query of every handle and written by vk_corpus_view /
// vk_corpus_free, from whichever threads the caller uses: one mutex around insert, unlink and the peer's turn-taking event.
std::mutex &vk_ring_mutex();

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
	int8_t *d_pos = nullptr;   // POS code per token (tag-weighted queries, token filters)
	int8_t *d_tag = nullptr;   // tag code per token (token filters)
	int32_t *d_sent_start = nullptr, *d_sent_end = nullptr;
	// host mirrors, shared by the views of a corpus: rows of the slice table (as on the device, padding included), token ids and tag
	// codes of the static layout -- what restating a transport winner on the host needs (vk_transport_host.h)
	std::shared_ptr<std::vector<int32_t>> h_start, h_end, h_tok;
	std::shared_ptr<std::vector<int8_t>> h_tag;
	bool contiguous = false;   // slices are the CSR partition of the token stream
	bool overlapping = false;  // some token belongs to more than one slice (sliding windows)
	bool have_ids = false, have_sent = false, finalized = false;
	int max_len = 0, max_group_tiles = 0, max_group_tokens = 0;
	int max_pair_tiles = 0;    // tiles spanned by two consecutive rows of the slice table (vk_score32_kernel)
	int max_short_pair_tiles = 0;   // ... leaving out the groups that hold a long slice
	// slice table on the device: n_entries >= n_sentences rows.  Slices longer than VK_FAST_SENT_LEN sit alone in
	// their group of 4 (padded with empty rows) and are scored by a second launch over d_long_groups.
	int64_t n_entries = 0;
	std::vector<int32_t> entry_sent;   // [n_entries] sentence of a row, -1 = padding; empty when the table is the identity
	std::vector<int32_t> sent_entry;   // its inverse (row of a sentence), built when vk_query_desc.only_slices first needs it
	int32_t *d_long_groups = nullptr;
	int n_long_groups = 0, max_short_len = 0, long_group_tiles = 0, long_group_tokens = 0;   // the long pass: slices of 65 .. VK_MAX_SENT_LEN tokens
	int max_long_len = 0;      // ... the longest of them
	std::shared_ptr<std::vector<int32_t>> h_apart;   // rows of the slice table of every group that holds a slice of more than 64 tokens
	std::shared_ptr<std::vector<int32_t>> h_xlong;   // rows of the slice table of slices beyond VK_MAX_SENT_LEN (whole documents)
	int uniform_len = 0;       // > 0: every sentence has exactly this many tokens
	uint8_t *d_bq = nullptr; int32_t *d_bqlen = nullptr; float *d_bscores = nullptr; uint64_t *d_bkeys[2] = {nullptr, nullptr};
	float *d_braw = nullptr; size_t braw_cap = 0;   // aligner scores of a batch of alignment queries
	size_t bq_cap = 0, bqlen_cap = 0, bscores_cap = 0, bkeys_cap = 0;
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
	size_t rows_cap = 0;   // floats each of them holds
	uint8_t *d_bqt = nullptr; uint64_t *d_bcand = nullptr; int32_t *d_bcandq = nullptr; float *d_brows = nullptr;   // similarity rows of a batch's winners
	size_t bqt_cap = 0, bcand_cap = 0;
	float *h_brows = nullptr; size_t h_brows_cap = 0;   // pinned host staging of the similarity rows of a batch's winners
	uint32_t *d_qbits = nullptr;   // tag-weighted vocabulary transports over the static layout: bitmap of the query's token ids
	uint8_t *d_wrdl_scratch = nullptr;   // exact transport, queries of 17..64 tokens over long slices: per-workgroup state
	uint8_t *d_wide_scratch = nullptr; size_t wide_scratch_cap = 0;   // vk_wide_kernel, global-state form: per-workgroup state of a slice
	int32_t *d_apart_order = nullptr; int32_t n_apart_order = -1;   // ... over the slices of more than 64 tokens (general gaps: the one-wave-per-slice pass is their fastest kernel)
	int32_t *d_xlong_order = nullptr; int32_t n_xlong_order = -1;   // the same list over the slices beyond VK_MAX_SENT_LEN only (queries of at most 16 tokens: the other slices keep their fused kernels)
	int32_t *d_wide_order = nullptr; int32_t n_wide_order = -1;   // ... its work list: the non-empty rows of the slice table, longest first
	size_t ws_cap = kGapTable;   // floats d_ws holds (grown by a query over a corpus with longer slices)
	int rows_w = 0;              // columns per similarity row they are sized for (16, 32, 48 or 64)
	// batched GEMM over a ragged corpus (vk_query_batch): a padded copy of the sentences, one bucket per padded length
	// 16 / 32 / 48 / 64 tokens, built on the first such batch (this handle's; about 1.2 x the corpus for lengths 8..64)
	struct batch_bucket { uint8_t *tiles = nullptr; int32_t *len = nullptr, *id = nullptr; int64_t n = 0; };
	batch_bucket bl[4];
	bool bl_built = false;
	int64_t bl_empty = 0;        // slices without tokens (in no bucket: their scores stay -inf)
	// batched relaxed WMD over the static layout (vk_rwmd_static32_kernel): the rows of the slice table by length bucket (1..32 /
	// 33..64 tokens; null lists when every slice has exactly 32 tokens), the batch's similarity table and its diagonal cells
	int32_t *d_sb_id[2] = {nullptr, nullptr}; int64_t sb_n[2] = {0, 0}; bool sb_built = false; int64_t sb_empty = 0;
	uint16_t *d_btable = nullptr; size_t btable_cap = 0;
	int64_t *d_bfix = nullptr; size_t bfix_cap = 0;
	int32_t *d_bqids = nullptr; size_t bqids_cap = 0;   // token ids of a batch's queries, 16 per query (the winners' rows: sim[id(t_j)][j] = 1)
	size_t wrd_cap = 0;          // candidates d_wrd_raw / d_wrd_val (and d_keys[0]) can hold
	int16_t *d_out_map = nullptr;
	size_t out_cap = VK_MAX_MATCHES;   // winners d_out_raw / d_out_sim / d_out_map hold (grown by a query that asks for more matches)
	uint64_t *d_sort[2] = {nullptr, nullptr}; void *d_sort_temp = nullptr; size_t sort_cap = 0, sort_temp_cap = 0;   // result sets beyond VK_MAX_MATCHES: all keys, sorted
	hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // 0 start, 5 before / 1 after the wait for the peer's kernel, 2 scored (the peer's turn), 3 selected, 4 done; 6: the batched GEMM has ended (its turn ends after the selection)
	vk_timings last{};
	bool have_scores = false;
	bool is_view = false;        // shares the corpus arrays of another handle (vk_corpus_view)
	bool shares_vectors = false; // a filtered corpus of the static layout: vocabulary tiles and magnitudes belong to its source
	// who owns the arrays several handles read (d_tiles, d_mag, d_tok_id, d_pos, d_tag, d_sent_start, d_sent_end, d_long_groups):
	// `shared` the block this handle allocates into (a view: its source's), `vectors_of` the block of the source of a filtered
	// static corpus (its vocabulary tiles and magnitudes).  The raw pointers above are aliases into these blocks.
	std::shared_ptr<vk_devblock> shared, vectors_of;
	// queries of 65 .. VK_MAX_LONG_QUERY_LEN tokens (vk_longq_host.cpp): query tiles, small per-query arrays (floats / ints), the static
	// layout's tables (one per 16 query tokens), scratch of the scoring pass and of the tracebacks, the winners' outputs
	struct longq_bufs {
		uint8_t *qt = nullptr; size_t qt_cap = 0;
		float *fl = nullptr; size_t fl_cap = 0;
		int32_t *il = nullptr; size_t il_cap = 0;
		float *table = nullptr; size_t table_cap = 0;
		uint8_t *scratch = nullptr; size_t scratch_cap = 0;
		uint8_t *fscratch = nullptr; size_t fscratch_cap = 0;
		float *raw = nullptr; size_t raw_cap = 0;
		float *sim = nullptr; size_t sim_cap = 0;
		int16_t *map = nullptr; size_t map_cap = 0;
	} lq;
	vk_corpus *peer = nullptr;   // ring of the handles on one corpus: a handle's scoring kernel starts after its peer's (under vk_ring_mutex)
	std::atomic<bool> ev2_recorded{false};   // (device-side wait on ev[2]), so that scoring kernels run back to back, never queued inside each other
};

namespace {

int alloc(vk_corpus *c, void **p, size_t bytes) {
	VK_HIP(hipMalloc(p, bytes ? bytes : 16));
	c->device_bytes += (int64_t)bytes;
	return VK_OK;
}

template <typename T> int alloc_t(vk_corpus *c, T **p, size_t n) { return alloc(c, (void **)p, n * sizeof(T)); }

// an array the views of this corpus read too: owned by the handle's refcounted block
template <typename T> int alloc_shared(vk_corpus *c, T **p, size_t n) {
	const int rc = alloc(c, (void **)p, n * sizeof(T));
	if (rc == VK_OK) c->shared->add((void *)*p);
	return rc;
}

} // namespace

// units
// this handle's scoring kernel starts when its peer's has finished: a device-side wait on the peer's event, taken under the
// ring's mutex so that the peer cannot be unlinked and destroyed in between (vk_corpus.cpp)
int vk_wait_peer_turn(vk_corpus *c, hipStream_t st);
int vk_validate_query(const vk_corpus *c, const vk_query_desc *q, const vk_topk_out *out);
class vk_host_keep;
int vk_longq_query(vk_corpus *c, const vk_query_desc *q, vk_topk_out *out, vk_host_keep &keep);   // vk_longq_host.cpp
void vk_pack_query(const vk_corpus *c, const vk_query_desc *q, std::vector<uint8_t> &tile, float *mags);

#endif
