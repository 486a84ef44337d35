/*
 * vk_oracle.h -- CPU restatement of Vectorian's brute-force alignment search path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the timed CPU baseline.  The product path
 * (vectorian_amd/ + libvectorian_hip.so) never links or calls it.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - similarity (a1-a5): pinned by golden vectors generated from the reference's
 *     own vectorian/sim/vector.py + vectorian/embedding/vectors.py
 *     (tests/golden/cosine_*.npz, generator tools/make_goldens.py);
 *   - alignment score arithmetic: pinned by the one known answer the reference
 *     ships, mkdocs/docs/introduction.md:150-184 (score 0.8001667857170105
 *     reproduces bit-exactly);
 *   - traceback tie-breaking, EMD quantisation: PARITY UNPINNED -- the DP lives in
 *     the third-party pyalign (submodule pinned at branch 0.3.6, absent from
 *     /root/reference) and the EMD in pyemd (absent); the published algorithms
 *     are restated and the tie rules are defined here (see vko_align).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).
 */
#ifndef VK_ORACLE_H
#define VK_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VKO_MAX_LEN_S 32767 /* int16 of a mapping (metric/alignment.h:357-358) */
#define VKO_MAX_LEN_T 512 /* (64 until round 4: queries of up to 512 tokens, alignments) */

/* pyalign::enums::Locality as used at vectorian/core/cpp/metric/alignment.h:363-364
 * and vectorian/alignment.py:97,130,187 */
enum { VKO_LOCAL = 0, VKO_GLOBAL = 1, VKO_SEMIGLOBAL = 2 };

/* gap cost families, pyalign.gaps */
enum { VKO_GAP_LINEAR = 0, VKO_GAP_AFFINE = 1, VKO_GAP_TABLE = 2 };

typedef struct {
	int32_t kind;       /* VKO_GAP_* */
	float u;            /* linear: w(k)=u*k ; affine: w(k)=u+v*k */
	float v;
	const float *table; /* table: w(k)=table[k], k < n_table (table[0] = 0) */
	int32_t n_table;
} vko_gap;

/* ---- bf16 ------------------------------------------------------------- */
uint16_t vko_f32_to_bf16(float x);
float vko_bf16_to_f32(uint16_t x);
void vko_round_bf16(const float *in, uint16_t *out, int64_t n);

/* ---- a1: Vectors.magnitudes / Vectors.normalized ---------------------- */
void vko_magnitudes(const float *x, int64_t n, int32_t d, float *mag);
void vko_normalize_rows(const float *x, int64_t n, int32_t d, float *out);
/* library-side normalisation used by vk_corpus_append_vectors(normalize=1):
 * norm in double, one fp32 division, RNE to bf16 */
void vko_normalize_rows_bf16(const float *x, int64_t n, int32_t d, uint16_t *out, float *mag);

/* ---- a2/a4/a5/a7: contextual similarity ------------------------------- */
void vko_sim_bf16(const uint16_t *X, int64_t n_rows, int32_t d,
	const uint16_t *Q, int32_t len_t, float *S /* [n_rows x len_t] */);
void vko_sim_f32(const float *X, int64_t n_rows, int32_t d,
	const float *Q, int32_t len_t, float *S);

/* ---- a3/a6: static similarity table ----------------------------------- */
void vko_sim_table_static_bf16(const uint16_t *E, int32_t V, int32_t d,
	const uint16_t *Q, int32_t len_t, const int32_t *q_ids, float *table /* [V x len_t] */);

/* ---- a10: alignment DP + traceback ------------------------------------ */
/* S is row-major with leading dimension ld (S[i*ld + j], i<len_s, j<len_t).
 * mapping (len_t entries) may be NULL.  Returns 0 on success. */
int vko_align(const float *S, int32_t ld, int32_t len_s, int32_t len_t,
	int32_t locality, const vko_gap *gap_s, const vko_gap *gap_t,
	float *raw_score, int16_t *mapping);

/* forces the general (Waterman-Smith-Beyer) solver whatever the gap kinds */
int vko_align_general(const float *S, int32_t ld, int32_t len_s, int32_t len_t,
	int32_t locality, const vko_gap *gap_s, const vko_gap *gap_t,
	float *raw_score, int16_t *mapping);

float vko_gap_cost(const vko_gap *g, int32_t k);

/* ---- a11: Score -------------------------------------------------------- */
float vko_score(float raw, int32_t len_t, int32_t n_matched, float submatch_weight, float boost);

/* ---- a14-a17: (R)WMD ---------------------------------------------------- */
/* ids_s / ids_t: token ids for the static BOW builder, or NULL for the
 * contextual one (every position unique, bow.h:281-333).  Returns raw score
 * (cost_to_score); the ranking value is raw/len_t*boost (F8). */
float vko_rwmd(const float *S, int32_t ld, int32_t len_s, int32_t len_t,
	const int32_t *ids_s, const int32_t *ids_t,
	int32_t injective, int32_t symmetric, int32_t normalize_bow);

/* relaxed = 0: the full WMD (FullSolver, wmd.h:194-270), exact EMD over the joint vocabulary.
 * Returns NaN for the option combinations the reference rejects (wmd.h:201-209, 441-449). */
float vko_wmd(const float *S, int32_t ld, int32_t len_s, int32_t len_t,
	const int32_t *ids_s, const int32_t *ids_t,
	int32_t relaxed, int32_t injective, int32_t symmetric, int32_t normalize_bow);

/* ---- a18: WRD ----------------------------------------------------------- */
float vko_wrd(const float *S, int32_t ld, int32_t len_s, int32_t len_t,
	const float *mag_s, const float *mag_t, int32_t normalize_magnitudes);

/* exact EMD (transportation problem) in double; a[n], b[m] masses, C[n x m] */
double vko_emd(const double *a, int32_t n, const double *b, int32_t m, const double *C, double *flow /* may be NULL */);

/* ---- whole-corpus search (a8, a9, a12) -------------------------------- */
enum { VKO_LAYOUT_CONTEXTUAL = 0, VKO_LAYOUT_STATIC = 1 };
enum { VKO_ALG_ALIGN = 0, VKO_ALG_RWMD = 1, VKO_ALG_WRD = 2 };

typedef struct {
	int32_t layout;
	int32_t d;
	int64_t n_tokens;
	int64_t n_sentences;
	const uint16_t *X;        /* contextual: bf16 [n_tokens x d] unit rows */
	const float *X_mag;       /* contextual: optional magnitudes [n_tokens] (WRD) */
	const int32_t *tok_id;    /* static: [n_tokens] */
	const uint16_t *E;        /* static: bf16 [V x d] unit rows */
	int32_t V;
	const int64_t *sent_off;  /* [n_sentences+1], token units, contiguous (document.h:147-169, B8) */
	const int64_t *sent_end;  /* optional [n_sentences]: slice s = [sent_off[s], sent_end[s]) (sliding windows) */
	const int8_t *pos_s;      /* optional [n_tokens]: universal POS code per token (TagWeightedSlice) */
	/* fp32 unit rows instead of the bf16 ones (the reference's own precision, vectorian/sim/vector.py:66-78) */
	const float *X_f32;       /* contextual: [n_tokens x d], used when non-NULL */
	const float *E_f32;       /* static: [V x d], used when non-NULL */
	const int8_t *tag_s;      /* optional [n_tokens]: fine-grained tag code per token (Token.tag): with tag-weighted similarity the
	                             vocabulary of the bags of words is keyed by (token id, tag) (TaggedTokenFactory, alignment/bow.h:150-176) */
} vko_corpus;

typedef struct {
	int32_t algorithm;
	int32_t len_t;
	const uint16_t *Q;        /* bf16 [len_t x d] unit rows */
	const float *Q_mag;       /* optional [len_t] (WRD) */
	const int32_t *q_ids;     /* static: vocabulary id per query token or -1 */
	int32_t locality;
	vko_gap gap_s, gap_t;
	float submatch_weight;
	int32_t max_matches;
	float min_score;
	const float *boost;       /* optional [n_sentences] */
	int32_t rwmd_injective, rwmd_symmetric, rwmd_normalize_bow;
	int32_t wrd_normalize_magnitudes;
	/* alignment-tag-weighted (slice/static.h:186-288; match/instantiate.cpp:10-38,173-189) */
	const float *tag_weights; /* optional [len_t]: t_pos_weights; NULL = alignment-isolated */
	const int8_t *q_pos;      /* [len_t] universal POS code per query token */
	float pos_mismatch_penalty;
	float similarity_threshold;
	int32_t wmd_full;         /* VKO_ALG_RWMD with relaxed = False: full WMD */
	const float *Q_f32;       /* fp32 unit rows [len_t x d], used when non-NULL */
	/* contextual layout: the per-document similarity matrix already computed by the caller, clipped,
	 * [n_tokens x len_t] -- how the reference itself proceeds (metric/contextual.cpp:26-63 fills the matrix with ONE
	 * sgemm per document through vectorian/sim/vector.py:66-78, slice/contextual.h:65-67 then reads S[offset + i][j]);
	 * used when non-NULL (bench.py's BLAS leg of the CPU baseline) */
	const float *S_rows;
	const int8_t *q_tag;      /* optional [len_t]: fine-grained tag code per query token (see vko_corpus.tag_s) */
} vko_query;

typedef struct {
	int32_t n_out;
	float *score;             /* [max_matches] value = raw/ref*boost */
	float *raw;               /* [max_matches] */
	int64_t *sentence;        /* [max_matches] */
	int16_t *mapping;         /* [max_matches x len_t] (alignment only) */
	float *all_scores;        /* optional [n_sentences]: every sentence's value */
} vko_result;

int vko_find(const vko_corpus *c, const vko_query *q, vko_result *out, int32_t n_threads);
/* a batch of queries with the worker threads started once (each thread keeps its static
 * sentence range and walks the queries in turn) */
int vko_find_many(const vko_corpus *c, const vko_query *qs, int32_t n_queries, vko_result *outs, int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif
