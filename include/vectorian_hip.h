/*
 * vectorian_hip.h -- C-ABI of the MI355X (gfx950) brute-force alignment search.
 *
 * This is the drop-in boundary for ONE path of poke1024/vectorian: what
 * BruteForceIndex._find (vectorian/index.py:530-560) obtains today from the
 * pybind11 module `vectorian_core` (vectorian/core/cpp/module.cpp:36-151) by calling
 *     core.Query(...).initialize(tokens, **options)      module.cpp:115-120, query.cpp:32-154
 *     core.Document(...).find(query, booster)            module.cpp:124-127, document.cpp:60-69
 *     ResultSet.extend / ResultSet.best_n                module.cpp:142-145, result_set.h:70-93
 * The reference's native seam is welded to Python objects (string vocabularies,
 * numpy callbacks), so it is not re-exported 1:1; instead the same
 * data contracts -- token vectors / token ids, sentence spans, option values,
 * bounded result set with scores and the injective flow mapping -- cross this
 * plain-pointer interface.  Each entry point names the reference interface it
 * replaces.  No torch / numpy / pybind types appear here; INTEGRATION.md shows the
 * ctypes binding the reference's maintainers would add.
 *
 * Conventions: every function returns a vk_status (0 = ok); on error
 * vk_last_error() returns a thread-local message.  Host buffers are borrowed for
 * the duration of the call; the library owns device copies behind the opaque
 * handle; outputs go to caller-allocated buffers.  Calls on different handles
 * may run concurrently from different threads; calls on one handle must be
 * serialised by the caller.
 */
#ifndef VECTORIAN_HIP_H
#define VECTORIAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VK_ABI_VERSION 12
#define VK_MAX_QUERY_LEN 64   /* query tokens, every strategy */
#define VK_MAX_LONG_QUERY_LEN 512 /* alignments (VK_ALG_ALIGN, submatch_weight = 0) over slices of at most VK_FAST_SENT_LEN tokens: queries of up to
                                 this many tokens (a whole paragraph as the query) -- one wave per slice with the roles swapped, lanes = the
                                 slice's tokens, the query's tokens streamed (vk_longq_kernel); upstream's own bound is the int16 of a
                                 mapping (metric/alignment.h:357-358) */
#define VK_FAST_QUERY_LEN 16  /* queries up to this length run in the fused kernels (one 16-wide MFMA column block); longer ones
                                 take the multi-block kernel (2 or 4 column blocks) or, where that does not apply, a
                                 one-wave-per-slice kernel, ~10x slower */
#define VK_MAX_SENT_LEN 512   /* tokens per sentence (slice), every algorithm; see VK_MAX_DOC_LEN */
#define VK_FAST_SENT_LEN 64   /* slices up to this length run 4 per wave in the fused kernel (|s| <= 64); longer
                                 ones take a second launch, one slice per wave (exact transport: a slower solver; with a query of
                                 more than VK_FAST_QUERY_LEN tokens its state lives in global memory) */
#define VK_MAX_DOC_LEN 32767  /* alignments (VK_ALG_ALIGN): tokens per slice when slices are whole documents -- what the int16 of a
                                 mapping can name, upstream's own bound (metric/alignment.h:357-358).  A corpus that holds a slice
                                 of more than VK_MAX_SENT_LEN tokens is scored by one wave per slice with the slice's state (column
                                 history of general gaps, traceback) in global memory; the transports return VK_ERR_UNSUPPORTED on it */
#define VK_MAX_MATCHES 1024   /* result sets up to this size come out of the streaming / block selection kernels; every strategy */
#define VK_MAX_MATCHES_SORTED 1048576   /* alignments (VK_ALG_ALIGN, submatch_weight = 0): result sets of up to this many matches -- a
                                 query that asks for more than VK_MAX_MATCHES has every score sorted on the device (upstream's
                                 ResultSet is bounded by max_matches alone, vectorian/core/cpp/result_set.h:32-68) */

typedef enum {
	VK_OK = 0,
	VK_ERR_INVALID = 1,      /* bad argument / option (query.cpp:60-63 throws std::runtime_error) */
	VK_ERR_UNSUPPORTED = 2,  /* option the HIP path does not implement: explicit, never a fallback */
	VK_ERR_HIP = 3,          /* HIP runtime error */
	VK_ERR_NO_DEVICE = 4,
	VK_ERR_STATE = 5,        /* call order violated (e.g. query before finalize) */
	VK_ERR_ABORTED = 6       /* the caller raised the query's abort flag (Query::abort, query.h:183-189): nothing, or only the
	                            queries of a batch before the poll, was computed */
} vk_status;

/* pyalign::enums::Locality (metric/alignment.h:363-364; vectorian/alignment.py:97,130,187) */
typedef enum { VK_LOCAL = 0, VK_GLOBAL = 1, VK_SEMIGLOBAL = 2 } vk_locality;

/* pyalign.gaps families: linear w(k)=u*k, affine w(k)=u+v*k,
 * table w(k)=table[k] (ConstantGapCost / ExponentialGapCost / any GapCost.costs(n)) */
typedef enum { VK_GAP_LINEAR = 0, VK_GAP_AFFINE = 1, VK_GAP_TABLE = 2 } vk_gap_kind;

/* 'algorithm' of the alignment option dict (vectorian/alignment.py:33,277,310) */
typedef enum { VK_ALG_ALIGN = 0, VK_ALG_RWMD = 1, VK_ALG_WRD = 2 } vk_algorithm;

typedef enum { VK_F32 = 0, VK_BF16 = 1 } vk_dtype;
typedef enum { VK_MEM_HOST = 0, VK_MEM_DEVICE = 1 } vk_mem;

/* VK_LAYOUT_CONTEXTUAL: one vector per token occurrence (ContextualEmbeddingSlice,
 *   slice/contextual.h:65-67; metric/contextual.cpp:26-63).
 * VK_LAYOUT_STATIC: token ids + vocabulary table (StaticEmbeddingSlice,
 *   slice/static.h:71-75; metric/static.cpp:9-78). */
typedef enum { VK_LAYOUT_CONTEXTUAL = 0, VK_LAYOUT_STATIC = 1 } vk_layout;

/* VK_PREC_BF16 (default): unit rows rounded to bf16, cosines on v_mfma_f32_16x16x32_bf16 -- the fast path
 * (north star); scores follow the reference's algorithm on the ROUNDED vectors and can differ from its fp32
 * results by a few 1e-4.  VK_PREC_F32: unit rows in fp32, cosines on v_mfma_f32_16x16x4_f32 (exact fp32
 * products, fp32 accumulation) -- the reference's own precision (CosineSim = fp32 sgemm, vectorian/sim/vector.py:
 * 66-78) at twice the bytes per token; scores within 1e-6 of the fp32 CPU path. */
typedef enum { VK_PREC_BF16 = 0, VK_PREC_F32 = 1 } vk_precision;

typedef struct vk_corpus vk_corpus_t;

typedef struct {
	int32_t layout;        /* vk_layout */
	int32_t d;             /* embedding dimension */
	int64_t n_tokens;      /* total tokens of this shard (< 2^31) */
	int64_t n_sentences;   /* slices of this shard */
	int32_t vocab_size;    /* VK_LAYOUT_STATIC: rows of the vocabulary table */
	int32_t keep_magnitudes; /* keep |x| per appended row (needed by VK_ALG_WRD; metric/contextual.cpp:49-54, metric/static.cpp:69-73) */
	int32_t precision;       /* vk_precision: how the unit rows are kept in HBM */
} vk_corpus_desc;

typedef struct {
	int32_t kind;          /* vk_gap_kind */
	float u, v;
	const float *table;    /* host, VK_GAP_TABLE: table[0..n_table), table[0] = 0 */
	int32_t n_table;       /* must cover the longest gap: > max(len_s, len_t) */
} vk_gap;

typedef struct {
	int32_t algorithm;       /* vk_algorithm */
	int32_t len_t;           /* query tokens, 1..VK_MAX_QUERY_LEN (alignments: ..VK_MAX_LONG_QUERY_LEN) */
	const void *q_vectors;   /* host [len_t x d] row-major */
	int32_t q_dtype;         /* vk_dtype */
	int32_t q_normalize;     /* 1: rows are L2-normalised by the library (Vectors.normalized) */
	const int32_t *q_token_ids; /* host [len_t], VK_LAYOUT_STATIC: vocabulary id or -1;
	                               sets sim[id(t_j)][j] = 1 (metric/static.cpp:58-67) */
	int32_t locality;        /* vk_locality (alignment only) */
	vk_gap gap_s;            /* cost of skipping document tokens */
	vk_gap gap_t;            /* cost of skipping query tokens   */
	float submatch_weight;   /* query.cpp:77-79, metric/alignment.h:84-106; >= 0 */
	int32_t bidirectional;   /* query.cpp:81-83 (parsed, unused upstream); must be 0 */
	int32_t max_matches;     /* query.cpp:87-89 */
	float min_score;         /* query.cpp:91-93; admission is score > min_score (metric/alignment.h:284) */
	const float *boost;      /* host [n_sentences] or NULL (Booster, matcher_impl.h:99) */
	int32_t want_flow;       /* 1: also produce mapping + edge similarities for the winners */
	/* VK_ALG_RWMD (vectorian/alignment.py:275-283) */
	int32_t rwmd_injective, rwmd_symmetric, rwmd_normalize_bow;
	/* VK_ALG_WRD (vectorian/alignment.py:308-313) */
	int32_t wrd_normalize_magnitudes;
	/* 'alignment-tag-weighted' (vectorian/sim/span.py:63-71; TagWeightedSlice, slice/static.h:186-288):
	 * S'[i][j] = S[i][j] * tag_weights[j] * (pos_s[i] != q_pos[j] ? 1 - pos_mismatch_penalty : 1), set to 0
	 * when <= similarity_threshold; the score is divided by sum(tag_weights) instead of len_t.
	 * tag_weights NULL = 'alignment-isolated'.  Any algorithm (TagWeightedSlice wraps every slice, match/instantiate.cpp:
	 * 173-189); needs vk_corpus_set_token_pos. */
	const float *tag_weights;  /* host [len_t] t_pos_weights (match/instantiate.cpp:10-38) */
	const int8_t *q_pos;       /* host [len_t] universal POS code per query token */
	const int8_t *q_tags;      /* host [len_t] tag code per query token, or NULL: with tag weights the bags of words are keyed by
	                              (token id, tag) (TaggedTokenFactory, alignment/bow.h:150-176).  Over the static layout the
	                              device needs the keys (with vk_corpus_set_token_tags) for the masses of the 1:n RWMD (repeated ids
	                              are one entry only if their tags agree) and for the cells upstream's distance matrix writes
	                              twice (entries in both documents, wmd.h:121-133); without them those cells keep the positions'
	                              own similarity (scores of such slices within 5e-4) */
	float pos_mismatch_penalty;
	float similarity_threshold;
	/* VK_ALG_RWMD with 'relaxed': False (vectorian/alignment.py:206-218): the full Word Mover's Distance,
	 * exact EMD between the two bags of words (FullSolver, alignment/wmd.h:194-270); needs
	 * rwmd_injective = rwmd_symmetric = 0 as upstream (wmd.h:201-209) */
	int32_t wmd_full;
	/* Query::abort (module.cpp:120, query.h:183-189; polled per slice at match/matcher_impl.h:105): optional host flag the caller
	 * may raise from another thread.  Polled when vk_query starts and between the queries / launches of vk_query_batch (a query
	 * is a few milliseconds of device time and is not interrupted); a raised flag ends the call with VK_ERR_ABORTED. */
	const volatile int32_t *abort;
	/* The debug hook of the reference is called for EVERY slice a matcher scores (call_debug_hook, metric/alignment.h:145-173:
	 * slice, similarity, flow, score), not only for the winners.  only_slices (host, n_only entries, each a slice index of this
	 * handle, n_only <= min(VK_MAX_MATCHES, capacity); NULL: off) makes vk_query skip the scoring pass and the selection and
	 * state exactly these slices instead, in the given order: score, raw_score, mapping and edge_sim from the traceback kernel
	 * (canonical arithmetic), sim_rows when the array is given; no min_score admission, n_out = n_only.  VK_ALG_ALIGN with
	 * want_flow (ABI 11: any submatch_weight -- the score of a listed slice rests on its own traceback); since ABI 10 also the relaxed WMD (VK_ALG_RWMD without wmd_full) with sim_rows: score /
	 * raw_score restated on the host from the rows (NaN for a slice longer than rows_per_winner, -inf for an empty one); and the
	 * exact transports (VK_ALG_WRD, wmd_full): every listed slice solved, no bound pass, sim_rows and plan filled when given.
	 * A caller walks the corpus in chunks with it (Index: debug = AllSlices(hook)). */
	const int64_t *only_slices;
	int32_t n_only;
} vk_query_desc;

/* Bounded result set, best first.  Order: score descending, then sentence index
 * descending (the deterministic part of Match::compare_by_score,
 * match/match_impl.h:8-42). */
typedef struct {
	int32_t capacity;        /* entries the arrays below can hold (>= max_matches) */
	int32_t n_out;           /* written by the library */
	float *score;            /* [capacity] Score::value = raw / reference_score * boost (match.h:302-307) */
	float *raw_score;        /* [capacity] aligner score / cost_to_score */
	int64_t *sentence;       /* [capacity] slice index within this corpus handle */
	int16_t *mapping;        /* [capacity x len_t] InjectiveFlow target per query token or -1
	                            (metric/alignment.h:194-196); NULL if !want_flow */
	float *edge_sim;         /* [capacity x len_t] S[mapping[j]][j] (ScoreComputer, metric/alignment.h:335-345
	                            reports distance = 1 - this); NULL if !want_flow */
	/* transport algorithms (VK_ALG_RWMD, VK_ALG_WRD) with want_flow: what the host needs to state the flow of a
	 * winner as SparseFlow / DenseFlow (match/match.h:140-260; alignment/wmd.h:392-408, 228-248; wrd.h:120-135).
	 * Alignments fill sim_rows too when the array is given: the 'similarity' matrix of the debug hook
	 * (call_debug_hook, metric/alignment.h:145-173), for the winners.
	 * Optional (NULL: not produced).
	 * Relaxed WMD (VK_ALG_RWMD without wmd_full) with want_flow and sim_rows given: score / raw_score of the result set are
	 * restated on the host from these rows in the reference's order of operations (RelaxedSolver, alignment/wmd.h:287-416) --
	 * bit-identical to a scalar fp32 run of the reference, whichever kernel ranked the slices; a winner of more than R tokens
	 * keeps the value of the scoring pass.
	 * W = the query length rounded up to a multiple of 16 (16 for queries of at most VK_FAST_QUERY_LEN tokens). */
	float *sim_rows;         /* [capacity x R x W] similarity S[i][j] of slice token i and query token j
	                            (clipped, tag weights applied, static layout: sim[id(t_j)][j] = 1); rows >= the slice's length are zero */
	float *plan;             /* [capacity x W x R] exact transport only (VK_ALG_WRD, wmd_full): the optimal
	                            plan G[j][i], mass moved from query token j to slice token i (positions) */
	int32_t rows_per_winner; /* R: slice tokens per winner the two arrays hold; 0 = VK_FAST_SENT_LEN.  Winners of more than R tokens
	                            get zero rows (their flows are not stated); up to VK_MAX_SENT_LEN, a multiple of 64 */
} vk_topk_out;

/* kernel timings of the last vk_query on a handle, milliseconds, from HIP events
 * recorded on the handle's stream */
typedef struct {
	float prepare_ms;   /* query upload + static table (if any) */
	float score_ms;     /* the fused similarity + DP kernel (dominant) */
	float topk_ms;      /* bounded result set selection */
	float flow_ms;      /* traceback of the winners */
	float total_ms;     /* first to last event, without queue_ms: the device time this query took once it had its turn */
	float queue_ms;     /* several handles on one corpus (vk_corpus_view): time this query's scoring kernel waited on the
	                       device for the kernel of the handle before it (the kernels take turns; 0 with one handle) */
} vk_timings;

int vk_abi_version(void);
const char *vk_last_error(void);

/* selects the HIP device for the calling thread's subsequent vk_corpus_create calls */
int vk_init(int device);
int vk_device_count(int *count);

/* --- corpus residency: replaces core.Document(...) construction (document.cpp:30-58)
 *     and the per-document vectors handed to the contextual metric ------------- */
int vk_corpus_create(const vk_corpus_desc *desc, vk_corpus_t **out);
/* Appends rows in order: token vectors (contextual) or vocabulary vectors (static).
 * rows: [n_rows x d] row-major, dtype f32/bf16, in host or device memory.
 * normalize=1 L2-normalises each row (Vectors.normalized, vectorian/embedding/vectors.py:71-80). */
int vk_corpus_append_vectors(vk_corpus_t *c, const void *rows, int64_t n_rows, int32_t dtype, int32_t mem, int32_t normalize);
/* VK_LAYOUT_STATIC: token ids of the whole shard (Token.id, common.h:34-42) */
int vk_corpus_set_token_ids(vk_corpus_t *c, const int32_t *ids, int64_t n, int32_t mem);
/* universal POS code per token occurrence (Token.pos, common.h:34-42); only needed by tag-weighted queries */
int vk_corpus_set_token_pos(vk_corpus_t *c, const int8_t *pos, int64_t n, int32_t mem);
/* fine-grained tag code per token occurrence (Token.tag, common.h:34-42); only needed by tag filters */
int vk_corpus_set_token_tags(vk_corpus_t *c, const int8_t *tags, int64_t n, int32_t mem);
/* sentence spans as CSR offsets in token units, contiguous from 0
 * (Spans::iterate, document.h:147-169).  sent_off: host [n_sentences + 1]. */
int vk_corpus_set_sentences(vk_corpus_t *c, const int64_t *sent_off, int64_t n_sentences);
/* general slices, e.g. sliding windows with window_step != window_size (Spans::iterate with
 * bounded_len, document.h:147-169): slice i = tokens [start[i], end[i]); starts and ends non-decreasing,
 * at most VK_MAX_SENT_LEN tokens each (alignments: VK_MAX_DOC_LEN); slices may overlap or leave gaps.  host arrays [n_sentences]. */
int vk_corpus_set_slices(vk_corpus_t *c, const int64_t *start, const int64_t *end, int64_t n_sentences);
int vk_corpus_finalize(vk_corpus_t *c);
/* A second handle on the same resident corpus (read-only arrays shared, own stream and workspaces): two handles serve
 * two queries at a time from two host threads, so that selection, traceback and the host part of one query overlap the
 * scoring kernel of the next.  The shared arrays are reference-counted: handles may be freed in any order, and freeing one while
 * ANOTHER handle of the corpus is inside a call is safe (calls on one handle stay the caller's to serialise).  (The reference runs one ThreadPool task per
 * document, vectorian/index.py:544-558; here the unit of concurrency is the query.) */
int vk_corpus_view(vk_corpus_t *src, vk_corpus_t **out);
/* Token filter (TokenFilter::pass, vectorian/core/cpp/query.h:8-28; options pos_filter / tag_filter of
 * Query::initialize, query.cpp:220-228): a token is dropped when bit Token.pos of pos_mask or bit Token.tag of tag_mask
 * is set (codes 0..63).  The reference compacts every slice per query and document (FilteredSliceFactory,
 * slice/static.h:366-416); here the filtered corpus is built once on the device -- rows of the dropped tokens removed,
 * slices re-indexed, same slice ids -- and queried like any other corpus; traceback positions count the tokens that
 * passed, as upstream's do (flow.cpp:49-60 maps them back).  The static layout shares the vocabulary vectors with
 * `src` (reference-counted: either may be freed first). */
int vk_corpus_filter(vk_corpus_t *src, uint64_t pos_mask, uint64_t tag_mask, vk_corpus_t **out);
int vk_corpus_free(vk_corpus_t *c);
int vk_corpus_device_bytes(const vk_corpus_t *c, int64_t *bytes);

/* --- one query against the shard: replaces Query.initialize + Document.find for
 *     every document + ResultSet.extend/best_n (vectorian/index.py:530-560) ------ */
int vk_query(vk_corpus_t *c, const vk_query_desc *q, vk_topk_out *out);

/* A batch of queries against the shard (BASELINE config 4: 256 queries per call).  Semantically
 * n_queries calls of vk_query; for injective RWMD over a contextual corpus whose sentences all have
 * the same length 16 / 32 / 48 / 64 it runs as one MFMA-bound GEMM with the row / column minima as
 * epilogue (every corpus byte read once per batch).  outs: [n_queries]; with want_flow, relaxed-WMD queries get the
 * similarity rows of their winners in outs[i].sim_rows when that array is given (one launch for the whole batch). */
int vk_query_batch(vk_corpus_t *c, const vk_query_desc *qs, int32_t n_queries, vk_topk_out *outs);

/* every sentence's Score::value of the last query, for the debug hook
 * ('alignment' callback, metric/alignment.h:145-173).  host [n_sentences]. */
int vk_last_scores(vk_corpus_t *c, float *scores, int64_t n);
int vk_last_timings(const vk_corpus_t *c, vk_timings *t);

/* ResultSet::extend (result_set.h:70-93) over plain arrays: merges `n_sets` result
 * sets (e.g. one per GPU rank after the all-gather) into one bounded set.
 * sentence indices must already be global. Pure host code. */
int vk_merge_topk(const vk_topk_out *sets, int32_t n_sets, int32_t len_t, int32_t max_matches, vk_topk_out *out);

/* The exchange records of the sharded path (one process per GPU; the only traffic per query is an all-gather of every
 * rank's k records, vectorian_amd/shards.py): a result set as k fixed-size records of int32 words, so that the host
 * part of the exchange -- packing before the collective, ResultSet::extend over all ranks' records after it -- is two
 * calls whatever the number of ranks.  One record: valid, score, raw score, global slice index (2 words), mapping
 * i16[W], edge_sim f32[W], padded to a multiple of 4 words; W = len_t rounded up to a multiple of 16. */
int32_t vk_record_words(int32_t len_t);
/* records: [k x vk_record_words(len_t)], rows >= set->n_out zeroed; slice indices become sentence_offset + local index */
int vk_pack_records(const vk_topk_out *set, int32_t len_t, int32_t k, int64_t sentence_offset, int32_t *records);
/* ResultSet::extend (vectorian/core/cpp/result_set.h:70-93) + best_n (result_set.cpp:3-21) over the records of n_sets
 * result sets.  records: [n_sets x k x words] as the all-gather delivers them; out as for vk_merge_topk (capacity >= k);
 * order: score descending, ties by slice index descending (Match::compare_by_score, match/match_impl.h:8-42). */
int vk_merge_records(const int32_t *records, int32_t n_sets, int32_t len_t, int32_t k, vk_topk_out *out);

/* The relaxed word mover's distance of ONE slice from its similarity rows, on the host, in the reference's order of operations
 * (BOWBuilder / UniqueTokensBOWBuilder, vectorian/core/cpp/alignment/bow.h:204-333; distance matrix, alignment/wmd.h:107-135;
 * RelaxedSolver, :287-416; cost_to_score, :138-140) -- what vk_query / vk_query_batch return for the winners of a relaxed-WMD query
 * (vk_topk_out.sim_rows), exposed so that a caller holding rows can restate a score (no GPU needed).
 * S[i * ld + j]: modified similarity of slice token i and query token j.  key_s / key_t: vocabulary keys of the tokens -- token ids,
 * or (id, tag) pairs folded into one int for the tag-weighted similarity -- or both NULL (every position its own entry: contextual
 * embeddings).  score_out: (max_cost - cost) / max_cost, the aligner score of RelaxedWordMoversDistance (metric/alignment.h:579-607). */
int vk_rwmd_from_rows(const float *S, int32_t ld, int32_t len_s, int32_t len_t, const int32_t *key_s, const int32_t *key_t,
	int32_t injective, int32_t symmetric, int32_t normalize_bow, float *score_out);

#ifdef __cplusplus
}
#endif
#endif
