// vk_device.h -- structures shared between the kernels (vk_*.hip) and the
// C-ABI implementation (vk_corpus.cpp, vk_query.cpp, vk_batch.cpp).  Internal; the public interface is
// include/vectorian_hip.h.
#ifndef VK_DEVICE_H
#define VK_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define VK_DEV_LOCAL 0
#define VK_DEV_GLOBAL 1
#define VK_DEV_SEMIGLOBAL 2

#define VK_DEV_LAYOUT_CONTEXTUAL 0
#define VK_DEV_LAYOUT_STATIC 1

#define VK_DEV_MAX_SENT_LEN 64        // fused fast path: 4 slices per wave
#define VK_DEV_MAX_LONG_LEN 512       // longer slices: one per wave, second launch
#define VK_DEV_MAX_QUERY_LEN 16        // fused kernels: one 16-column block
#define VK_DEV_MAX_WIDE_QUERY_LEN 64   // vk_wide_kernel: lane = query column

struct VkScoreParams {
	// corpus
	const uint8_t *tiles;      // contextual: token tiles
	const int32_t *tok_id;     // static: token ids
	const float *table;        // static: per-query similarity table [V_pad x 16]
	const int32_t *sent_start; // [n_sent + 8] first token of each slice (entries >= n_sent: empty)
	const int32_t *sent_end;   // [n_sent + 8] one past the last token
	int32_t n_sent;
	int32_t layout;
	int32_t nk32, tail, tile_bytes;
	int32_t prec;              // 0: bf16 tiles (nk32 K-steps of 32); 1: fp32 tiles (nk32 blocks of 16 features)
	int32_t q_lds;             // MODE 1: bytes of the query tile staged at the start of the dynamic LDS (0: read through L1 / L2)
	int32_t q_mode3;           // 300-d bf16 rows: 1 = query tile in LDS (MODE 3, 136 VGPRs with general gaps), 0 = in registers (MODE 0, 160);
	                           // 300-d fp32 rows: 1 = the specialised form (MODE 4), 0 = the generic loop (MODE 1)
	const int32_t *group_list; // null: all groups of 4 slices; else the groups holding one long slice each (64-thread blocks)
	int32_t n_list;
	int32_t max_short_len;     // main launch: groups with a longer slice are skipped
	// query
	const uint8_t *qtile;      // query in tile order (16 rows, rows >= len_t are zero)
	int32_t len_t;
	int32_t locality;
	int32_t gap_mode;          // 0 linear, 1 affine, 2 general, 3 general (register history), 4 RWMD
	int32_t rwmd_symmetric, rwmd_normalize_bow;
	int32_t wmd_bound;         // 0: RWMD score; 1: upper bound of the full WMD score, nbow; 2: same, bow
	float gs, gt;              // linear: w(k) = g*k; affine: b (extension)
	float a_s, a_t;            // affine: a
	float open_s, open_t;      // affine: a + b
	const float *ws;           // general: [max_len + 1]
	const float *wt;           // general: [17]; for the in-row candidates: the subadditive closure of the caller's table (vk_query.cpp)
	const float *wt0;          // general: the caller's table itself: the border row H[0][j] = -w_t(j)
	const float *boost;        // [n_sent] or null
	// tag-weighted similarity modifier (pos_s == nullptr: off)
	const int8_t *pos_s;       // [n_tokens + pad] POS code per token
	const int8_t *tag_s;       // tagged 1:n RWMD over the static layout: tag code per token (vocabulary entries are (id, tag) pairs); else null
	// tag-weighted vocabulary transports over the static layout (static_vocab_fixup, vk_common.hip.h): bitmap of the query's token
	// ids over the vocabulary (null: off) and the (id, tag) key of every query token
	const uint32_t *qid_bits;
	int32_t slices_overlap;    // slices share tokens (sliding windows): rewritten cells are per slice, the wave takes its slices in turns
	int32_t qkey[VK_DEV_MAX_QUERY_LEN];
	float tw[VK_DEV_MAX_QUERY_LEN];      // t_pos_weights
	int32_t tpos[VK_DEV_MAX_QUERY_LEN];  // POS code per query token
	float tw_keep;             // 1 - pos_mismatch_penalty
	float tw_threshold;
	float ref_total;           // reference_score: len_t, or sum(tw)
	const float *mag;          // WRD: token magnitudes (contextual) / vocabulary magnitudes (static)
	float qmass[VK_DEV_MAX_QUERY_LEN];   // WRD: query masses |q_j| / sum |q| (or |q_j| when wrd_raw_total > 0)
	float wrd_raw_total;       // WRD with normalize_magnitudes = false: sum |q_j|; 0 otherwise
	// outputs
	float *scores;             // Score::value per sentence (-inf: empty slice)
	float *raw;                // aligner score per sentence
	// LDS carve-up (floats)
	int32_t lds_floats_per_wave;
	int32_t s_rows_per_wave;   // rows of the similarity staging area
	int32_t h_rows;            // general gap: history rows per sentence (max_len + 1)
	int32_t m_rows;            // GAP 7, static layout: floats per sentence of the mass scratch (after the S strip)
};

struct VkRwmdBatchParams {
	const uint8_t *tiles;      // corpus token tiles
	int64_t n_tiles;
	int32_t tile_bytes, nk, half;
	const uint8_t *qtiles;     // [n_queries] query tiles, back to back (batch32: [n_qtiles] 32-row tiles)
	int32_t n_qtiles, qpt;     // batch32: tiles and queries per tile (2 or 3)
	int32_t dense;             // batch32: 1 = 16 ten-token queries per 5 tiles (vk_rwmd_batch32d_kernel); q_param then [queries][2]
	int32_t late_mask;         // batch32: waves with (wave & late_mask) != 0 run epilogue-then-MFMA (0: none)
	const int32_t *q_len;      // [n_queries]
	const float *q_inv_len;    // [n_queries] 1 / q_len
	const float *q_param;      // batch32: [n_qtiles + 1][8] per tile: length (as float) of its 3 queries, pad, reciprocals, pad; 0 = absent
	int32_t n_queries;
	int32_t n_sent;
	int32_t tiles_per_sent;    // every sentence has 16 * tiles_per_sent tokens
	int32_t symmetric, nbow;
	const float *boost;
	float *scores;             // [n_queries x n_sent]
	// ragged corpora (vk_rwmd_batch_kernel over one length bucket of the padded batch layout): real length and original index of
	// each of the bucket's n_sent sentences; null for uniform corpora.  score_stride: sentences of the whole corpus (row stride)
	const int32_t *sent_len;
	const int32_t *sent_id;
	int64_t score_stride;
	// the static layout (vk_rwmd_static32_kernel): the similarities come out of a per-batch table instead of MFMAs --
	// table[vocabulary row][query tile][lane half][16 accumulator slots], 16-bit fixed point (65535 = 1), i.e. for every word what the
	// 32x32 MFMA result holds for a token column of that word (vk_table_batch_kernel); tokens are gathered by id, a sentence's padding points at zero_row
	const int32_t *tok_id;
	const int32_t *sent_start, *sent_end;   // the slice table (p.sent_id: rows of it in this launch's bucket; null: rows 0 .. n_sent)
	const uint16_t *table;
	int64_t table_row;         // cells per vocabulary row of the table: 32 x (query tiles of the table)
	int32_t zero_row;          // a vocabulary row of zeros (the zero tile behind the vocabulary)
};

struct VkWrdParams {
	const uint8_t *tiles;
	const int32_t *tok_id;
	const float *table;        // static: nq tables [V_pad x 16], table_stride floats apart
	int64_t table_stride;
	const int32_t *sent_start;
	const int32_t *sent_end;
	int32_t layout;
	int32_t nk32, tail, tile_bytes;
	int32_t prec;
	const uint8_t *qtile;      // nq query tiles of 16 rows, tile_bytes apart
	int32_t nq;                // 16-column blocks of the query: 1 (<= 16 tokens) .. 4 (<= 64)
	int32_t len_t;
	const float *mag;
	float qmass[VK_DEV_MAX_WIDE_QUERY_LEN];
	int32_t mass_mode;         // 0: magnitudes (WRD); 1: 1/len per token (nbow); 2: 1 per token (bow)
	int32_t raw_masses;        // mass_mode 0: magnitudes as they are (normalize_magnitudes = false)
	// tag-weighted similarity modifier (pos_s == nullptr: off), as in VkScoreParams
	const int8_t *pos_s;
	float tw[VK_DEV_MAX_WIDE_QUERY_LEN];
	int32_t tpos[VK_DEV_MAX_WIDE_QUERY_LEN];
	float tw_keep, tw_threshold;
	float ref_total;           // reference_score: len_t, or sum(tw)
	const float *boost;
	const uint64_t *keys;      // candidates (0 = empty slot)
	float *raw_out;            // [n_cand]
	float *val_out;            // [n_cand]
	float *plan_out;           // optional [n_cand x 16 nq x rows_len]: the optimal plan G[j][i]
	float *rows_out;           // vk_rows_kernel: [n_cand x rows_len x 16 nq] similarity rows
	int32_t rows_len;          // slice tokens per candidate in rows_out / plan_out (0 = 64)
	const int8_t *tag_s;       // static layout, tag-weighted vocabulary transports: tag code per token, bitmap of the query's ids, keys of
	const uint32_t *qid_bits;  // the query tokens (static_vocab_fixup; qid_bits null: off)
	int32_t qkey[VK_DEV_MAX_WIDE_QUERY_LEN];
	const int32_t *cand_query; // vk_rows_kernel, batches: candidate w belongs to query cand_query[w], whose tile sits at qtile + that * qtile_stride
	int64_t qtile_stride;      // (static layout: its token ids at q_ids + that * q_ids_stride)
	int32_t q_ids_stride;
	// queries of more than 16 tokens over slices of more than VK_DEV_MAX_SENT_LEN tokens
	const int32_t *group_list; // vk_long_bound_kernel: groups of the slice table that hold one long slice (row 4 g)
	int32_t n_list;
	int32_t n_entries;         // rows of the slice table
	float *scores, *raw;       // vk_long_bound_kernel: per row of the slice table
	float wrd_raw_total;       // > 0: magnitudes as they are, the query's total
	int32_t wmd_bound;         // 0: WRD; 1: nbow; 2: bow (as VkWideParams)
	int32_t d;                 // features per row (unpadded) and, static layout, the query's token ids: the similarity rows of candidates
	const int32_t *q_ids;      // and winners are taken in the canonical arithmetic (sim_canon, vk_common.hip.h)
	uint8_t *scratch;          // vk_wrd_exact_long_kernel, nq > 1: flows, costs and similarity rows of one candidate per workgroup
	int64_t scratch_stride;
	int32_t n_cand;
};

struct VkFlowParams {
	const uint8_t *tiles;
	const int32_t *tok_id;
	const float *table;
	const int32_t *sent_start;
	const int32_t *sent_end;
	int32_t layout;
	int32_t nk32, tail, tile_bytes;
	int32_t prec;
	const uint8_t *qtile;
	int32_t len_t;
	int32_t locality;
	int32_t gap_mode;
	int32_t max_len;           // longest slice of the corpus: sizes the LDS carve-up
	float gs, gt, a_s, a_t, open_s, open_t;
	const float *ws;
	const float *wt;
	const int8_t *pos_s;
	float tw[VK_DEV_MAX_QUERY_LEN];
	int32_t tpos[VK_DEV_MAX_QUERY_LEN];
	float tw_keep, tw_threshold;
	int32_t d;                 // features per row (unpadded): the canonical sums follow the oracle's k order (sim_canon)
	const int32_t *q_ids;      // static layout: token ids of the query (-1: none), for sim[id(t_j)][j] = 1; null: no ids
	const uint64_t *keys;      // winners, best first; 0 = empty slot
	float *raw_out;            // [k]
	int16_t *mapping;          // [k x 16]
	float *edge_sim;           // [k x 16]
};

// a batch of at most 4 queries with common options over one pass of a contextual corpus (vk_score_batch_kernel)
struct VkScoreBatchParams {
	const uint8_t *tiles;
	const int32_t *sent_start;
	const int32_t *sent_end;
	int32_t n_sent;
	int32_t nk32, tail, tile_bytes;
	const uint8_t *qtiles;     // n_queries query tiles (16 rows each), tile_bytes apart
	int32_t n_queries;
	int32_t n_strips;          // strips per wave in the LDS carve-up (>= n_queries)
	int32_t len_t[4];
	int32_t locality;
	int32_t gap_mode;          // 0 linear, 1 affine, 2 general, 3 / 6 general (register history), 4 RWMD (injective)
	int32_t rwmd_symmetric, rwmd_normalize_bow;
	float gs, gt, a_s, a_t, open_s, open_t;
	const float *ws;
	const float *wt;           // closure of w_t (see VkScoreParams)
	const float *wt0;          // w_t as given
	const float *boost;
	float *scores;             // [n_queries x n_sent]
	float *raw;                // [n_queries x n_sent]
	int32_t lds_floats_per_wave;
	int32_t s_rows_per_wave;
	int32_t h_rows;
};

// queries of 17 .. 64 tokens (vk_wide_kernel): one wave per slice
struct VkWideParams {
	const uint8_t *tiles;
	const int32_t *tok_id;
	const float *table;        // static: nq tables [V_pad x 16], table_stride floats apart
	int64_t table_stride;
	const int32_t *sent_start;
	const int32_t *sent_end;
	int32_t n_sent;
	int32_t layout;
	int32_t nk32, tail, tile_bytes;
	int32_t prec;
	const uint8_t *qtile;      // nq query tiles of 16 rows, tile_bytes apart
	int32_t nq, len_t;
	int32_t locality;
	int32_t gap_mode;          // 0 linear, 1 affine, 2 general, 4 RWMD
	int32_t max_len;
	int32_t rwmd_symmetric, rwmd_normalize_bow;
	float gs, gt, a_s, a_t, open_s, open_t;
	const float *ws;
	const float *wt;           // [65]; vk_score32_kernel: the closure of w_t (see VkScoreParams); vk_wide_kernel: w_t as given
	const float *wt0;          // vk_score32_kernel: w_t as given (border row)
	const int8_t *pos_s;
	const int8_t *tag_s;       // tagged 1:n RWMD over the static layout: tag code per token (vocabulary entries are (id, tag) pairs); else null
	// tag-weighted vocabulary transports over the static layout (static_vocab_fixup, vk_common.hip.h): bitmap of the query's token
	// ids over the vocabulary (null: off) and the (id, tag) key of every query token
	const uint32_t *qid_bits;
	int32_t slices_overlap;    // as VkScoreParams
	int32_t qkey[VK_DEV_MAX_WIDE_QUERY_LEN];
	float tw[VK_DEV_MAX_WIDE_QUERY_LEN];
	int32_t tpos[VK_DEV_MAX_WIDE_QUERY_LEN];
	float tw_keep, tw_threshold;
	float ref_total;
	// transport bound (vk_score32_kernel, GAP 5): upper bound of the WRD / full WMD score of every slice
	const float *mag;          // token magnitudes (contextual) / vocabulary magnitudes (static); null: unit masses (bags of words)
	float qmass[VK_DEV_MAX_WIDE_QUERY_LEN];   // query masses: |q_j| / sum |q|, |q_j| (wrd_raw_total > 0), 1 / len_t (nbow) or 1 (bow)
	float wrd_raw_total;       // WRD on the magnitudes as they are: sum |q_j|; 0 otherwise
	int32_t wmd_bound;         // 0: WRD; 1: full WMD over normalised bags; 2: full WMD over unit masses
	const float *boost;
	float *scores;
	float *raw;
	int32_t d;                 // FLOW: features per row (unpadded) and, static layout, the query's token ids (sim_canon)
	const int32_t *q_ids;
	const uint64_t *keys;      // FLOW: winners
	float *raw_out;            // [k]
	int16_t *mapping;          // [k x 64]
	float *edge_sim;           // [k x 64]
	// vk_wide_kernel, global-state form (non-null: the state of a slice that grows with its length lives here, not in LDS):
	// one region of scratch_stride bytes per workgroup (vk_wide_scratch_bytes; vk_wide_gs_blocks regions)
	uint8_t *scratch;
	int64_t scratch_stride;
	const int32_t *order;      // vk_wide_kernel SCORE: rows of the slice table to walk, longest first (null: all of them, in order)
	int32_t n_order;
	const float *dp_rows;      // vk_wide_kernel FLOW: the winners' similarities, restated tile-parallel beforehand (vk_canon_rows_kernel):
	int32_t dp_rows_len;       //   [k][dp_rows_len][16 nq], row 0 = a winner's first token; null: the kernel restates them itself
	int32_t h_ring;            // vk_wide_kernel, global-state form with ws_tail: rows of the LDS ring of the column history (vk_wide_ring_rows; 0: history in the scratch)
	int32_t ws_tail;           // vk_wide_kernel, general gaps: w_s[k] == w_s[ws_tail] for every k >= ws_tail up to max_len (0: no such tail)
};

// queries of 65 .. VK_MAX_QUERY_LEN tokens (vk_longq_kernel): one wave per slice, lane = slice token, anti-diagonal sweep
struct VkLongqParams {
	const uint8_t *tiles;
	const int32_t *tok_id;
	const float *table;        // static: nq tables [V_pad x 16] (one per 16 query tokens), table_stride floats apart
	int64_t table_stride;
	const int32_t *sent_start;
	const int32_t *sent_end;
	int32_t n_sent;
	int32_t layout;
	int32_t nk32, tail, tile_bytes;
	int32_t prec;
	const uint8_t *qtile;      // nq query tiles of 16 rows, tile_bytes apart
	int32_t nq, len_t;
	int32_t s_stride;          // floats per row of the LDS strip: the corpus's longest slice rounded up to 16 (vk_longq_stride)
	int32_t locality;
	int32_t gap_mode;          // 0 linear, 1 affine, 2 general
	float gs, gt, a_s, a_t, open_s, open_t;
	const float *ws;           // general: w_s[0 .. 64]
	const float *wt;           // general: w_t[0 .. len_t]
	int32_t wt_tail;           // general, scoring pass: w_t[k] == w_t[wt_tail] for every k >= wt_tail up to len_t (0: no such tail)
	const int8_t *pos_s;       // tag-weighted similarity modifier (null: off)
	const float *tw;           // [16 nq] t_pos_weights (device)
	const int32_t *tpos;       // [16 nq] POS code per query token (device)
	float tw_keep, tw_threshold;
	float ref_total;
	const float *boost;
	float *scores;
	float *raw;
	int32_t d;                 // FLOW: features per row (unpadded) and, static layout, the query's token ids [16 nq] (sim_canon)
	const int32_t *q_ids;
	const uint64_t *keys;      // FLOW: winners
	int32_t n_keys;
	float *raw_out;            // [k]
	int16_t *mapping;          // [k x out_stride]
	float *edge_sim;           // [k x out_stride]
	int32_t out_stride;
	uint8_t *scratch;          // one region of scratch_stride bytes per workgroup (vk_longq_scratch_bytes): the matrix of general gaps, the cells' records
	int64_t scratch_stride;
};

#ifdef __cplusplus
extern "C" {
#endif
size_t vk_longq_scratch_bytes(int32_t len_t, int32_t gap_mode, int32_t flow, int32_t tagged);
size_t vk_longq_lds_bytes(int32_t len_t, int32_t max_len, int32_t flow);
int32_t vk_longq_blocks(int32_t len_t, int32_t max_len, int64_t n_sent, int32_t hm_in_lds);
int32_t vk_longq_hm_in_lds(int32_t len_t, int32_t max_len);
int32_t vk_longq_stride(int32_t max_len);
hipError_t vk_launch_longq(const VkLongqParams *p, int32_t flow_k, hipStream_t stream);
hipError_t vk_launch_wide(const VkWideParams *p, int32_t flow_k, hipStream_t stream);
// whole documents under linear / affine gaps, queries of at most 16 tokens: the skewed sweep (vk_doc.hip); flow_k > 0: p->dp_rows required
hipError_t vk_launch_doc(const VkWideParams *p, int32_t flow_k, hipStream_t stream);
size_t vk_doc_scratch_bytes(int32_t max_len, int32_t gap_mode);
// queries of 17 .. 64 tokens over long slices, linear / affine gaps (vk_docw.hip)
hipError_t vk_launch_docw(const VkWideParams *p, int32_t flow_k, hipStream_t stream);
size_t vk_docw_scratch_bytes(int32_t max_len, int32_t nq);
hipError_t vk_launch_docg(const VkWideParams *p, int32_t flow_k, hipStream_t stream);   // ... general gaps, 17 .. 32 tokens
size_t vk_docg_scratch_bytes(int32_t max_len);
hipError_t vk_launch_submatch_bound(const float *raw, const float *boost, int64_t n, float total, float w, float m_star,
	float *scores, hipStream_t stream);
hipError_t vk_launch_mark(const uint64_t *keys, int32_t n, float *scores, hipStream_t stream);
hipError_t vk_launch_select_ge(const float *scores, int64_t n, float theta, float floor_excl, uint64_t *keys_out,
	uint32_t *counter, uint32_t cap, hipStream_t stream);
size_t vk_wide_lds_demand(int32_t max_len, int32_t nq, int32_t gap_mode, int32_t tagged, int32_t flow);
size_t vk_wide_scratch_bytes(int32_t max_len, int32_t nq, int32_t gap_mode, int32_t flow, int32_t ring);
int32_t vk_wide_gs_blocks(int32_t max_len, int32_t nq, int32_t gap_mode, int32_t flow_k, int64_t n_sent, int32_t ring);
int32_t vk_wide_ring_rows(int32_t nq, int32_t gap_mode, int32_t ws_tail);
hipError_t vk_launch_pack(const void *in, int32_t dtype_bf16, int64_t n_rows, int32_t d, int32_t d_pad, int64_t row0,
	uint8_t *tiles, float *mag_out, int32_t normalize, int32_t prec, hipStream_t stream);
// queries of 17..32 tokens, linear / affine gaps (vk_score32.hip)
size_t vk_score32_lds_bytes(int32_t nk32, int32_t tail, int32_t max_pair_tiles, int32_t len_t, int32_t gap_mode, int32_t waves);
int32_t vk_score32_waves(int32_t nk32, int32_t tail, int32_t max_pair_tiles, int32_t len_t, int32_t gap_mode);   // waves per workgroup that fit the LDS: 4, 2, 1 or 0 (none)
hipError_t vk_launch_score32(const VkWideParams *p, int32_t max_pair_tiles, hipStream_t stream);
// token filters (vk_filter.hip): temp == nullptr only sizes the scan's workspace
hipError_t vk_launch_filter_scan(const int8_t *pos, const int8_t *tag, uint64_t pos_mask, uint64_t tag_mask, int64_t n,
	int32_t *keep, int32_t *new_index, void *temp, size_t *temp_bytes, hipStream_t stream);
hipError_t vk_launch_filter_maps(const int32_t *keep, const int32_t *new_index, int64_t n, int32_t *src_of,
	const int32_t *start, const int32_t *end, int64_t n_entries, int32_t *out_start, int32_t *out_end, hipStream_t stream);
hipError_t vk_launch_filter_rows(const uint8_t *src, uint8_t *dst, const int32_t *src_of, int64_t n_kept, int32_t tile_bytes,
	hipStream_t stream);
hipError_t vk_launch_filter_gather(const void *src, void *dst, int32_t elem_bytes, const int32_t *src_of, int64_t n_kept,
	hipStream_t stream);
hipError_t vk_launch_table(const uint8_t *etiles, const uint8_t *qtile, int32_t n_tiles, int32_t nk32, int32_t tail,
	int32_t tile_bytes, float *table, const int32_t *q_ids, int32_t len_t, int32_t V, int32_t prec, hipStream_t stream);
hipError_t vk_launch_score(const VkScoreParams *p, int32_t grid, size_t smem_bytes, hipStream_t stream);
hipError_t vk_launch_span(const VkScoreParams *p, hipStream_t stream);
hipError_t vk_launch_score_batch(const VkScoreBatchParams *p, int32_t lt, size_t smem, hipStream_t stream);
hipError_t vk_launch_topk_scores(const float *scores, int64_t n, float min_score, int32_t k, uint64_t *out,
	int32_t *n_blocks_out, hipStream_t stream);
hipError_t vk_launch_topk_keys(const uint64_t *in, int64_t n, int32_t k, uint64_t *out, int32_t *n_blocks_out,
	hipStream_t stream);
hipError_t vk_launch_topk_wave(const float *scores, const uint64_t *keys_in, int64_t n, float min_score, int32_t k,
	int64_t per_wave, uint64_t *out, int64_t *n_waves_out, hipStream_t stream);
hipError_t vk_launch_flow(const VkFlowParams *p, int32_t k, hipStream_t stream);
// result sets of more than VK_MAX_MATCHES entries: keys of ALL n scores (0 where score <= min_score), sorted descending (hipcub radix
// sort); keys_a / keys_b: n entries each; temp == nullptr only sizes the workspace; *sorted_out: whichever buffer holds the result
hipError_t vk_launch_sort_all(const float *scores, int64_t n, float min_score, uint64_t *keys_a, uint64_t *keys_b, void *temp, size_t *temp_bytes,
	uint64_t **sorted_out, hipStream_t stream);
hipError_t vk_launch_rwmd_batch(const VkRwmdBatchParams *p, hipStream_t stream);
hipError_t vk_launch_rwmd_batch32(const VkRwmdBatchParams *p, hipStream_t stream);
// padded copy of the sentences `ids` (n of them, each `tps` tiles long) for the batched GEMM over a ragged corpus
hipError_t vk_launch_batch_pack(const uint8_t *src_tiles, uint8_t *dst_tiles, const int32_t *ids, const int32_t *sent_start, const int32_t *sent_end,
	int64_t n, int32_t tps, int32_t tile_bytes, hipStream_t stream);
int vk_rwmd_batch_supported(int32_t nk, int32_t half);
// the static layout's batch: table[V rows][n_qtiles][2][16] from the packed 32-row query tiles (any row width), its diagonal cells
// (sim[id(t_j)][j] = 1, metric/static.cpp:58-67: `offsets` floats of the table set to 1), and the gather + epilogue pass
hipError_t vk_launch_table_batch(const uint8_t *etiles, int64_t n_vtiles, int32_t tile_bytes, int32_t nk16, const uint8_t *qtiles, int32_t n_qtiles,
	uint16_t *table, hipStream_t stream);
hipError_t vk_launch_table_batch_fix(uint16_t *table, const int64_t *offsets, int32_t n, hipStream_t stream);
hipError_t vk_launch_rwmd_static32(const VkRwmdBatchParams *p, int32_t w64, hipStream_t stream);
hipError_t vk_launch_topk_wave_batch(const float *scores, const uint64_t *keys_in, int64_t n, float min_score, int32_t k,
	int64_t per_wave, int32_t n_queries, int64_t in_stride, int64_t out_stride, uint64_t *out, int64_t *n_waves_out, hipStream_t stream);
hipError_t vk_launch_wrd_exact(const VkWrdParams *p, int32_t n_cand, float *scores_to_mark, hipStream_t stream);
hipError_t vk_launch_rows(const VkWrdParams *p, int32_t n_cand, hipStream_t stream);
hipError_t vk_launch_canon_rows(const VkWrdParams *p, int32_t n_cand, int32_t max_tiles, hipStream_t stream);
hipError_t vk_launch_wrd_exact_long(const VkWrdParams *p, int32_t n_cand, hipStream_t stream);
// queries of 17..64 tokens over long slices: workgroups of the exact solver (each with its own scratch) and bytes per workgroup
int vk_wrd_long_blocks(void);
size_t vk_wrd_long_scratch_bytes(void);
// 1:n RWMD of the long slices (group_list) for queries of 17..64 tokens
hipError_t vk_launch_long_rwmd_fill(const VkWideParams *p, const int32_t *group_list, int32_t n_list, int32_t n_entries, hipStream_t stream);
// upper bound of the transport score of every long slice (p->group_list), queries of 17..64 tokens
hipError_t vk_launch_long_bound(const VkWrdParams *p, hipStream_t stream);
#ifdef __cplusplus
}
#endif

#endif
