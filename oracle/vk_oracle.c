/*
 * vk_oracle.c -- CPU restatement of Vectorian's brute-force alignment search path.
 * TEST INFRASTRUCTURE ONLY (see vk_oracle.h).  Plain C, no dependencies.
 *
 * Build: make -C oracle      (gcc -O3 -ffp-contract=off: no FMA contraction, so the
 *                             fp32 DP arithmetic is the literal sequence written here)
 */
#include "vk_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* bf16                                                                      */
/* ------------------------------------------------------------------------- */

uint16_t vko_f32_to_bf16(float x) {
	uint32_t u;
	memcpy(&u, &x, 4);
	if ((u & 0x7fffffffu) > 0x7f800000u) {
		return (uint16_t)((u >> 16) | 0x0040u); /* keep NaN a NaN */
	}
	u += 0x7fffu + ((u >> 16) & 1u); /* round to nearest even */
	return (uint16_t)(u >> 16);
}

float vko_bf16_to_f32(uint16_t x) {
	uint32_t u = ((uint32_t)x) << 16;
	float f;
	memcpy(&f, &u, 4);
	return f;
}

void vko_round_bf16(const float *in, uint16_t *out, int64_t n) {
	for (int64_t i = 0; i < n; i++) out[i] = vko_f32_to_bf16(in[i]);
}

/* ------------------------------------------------------------------------- */
/* a1: Vectors.magnitudes / Vectors.normalized                               */
/* vectorian/embedding/vectors.py:82-86 (magnitudes), :71-80 (normalized)    */
/* ------------------------------------------------------------------------- */

void vko_magnitudes(const float *x, int64_t n, int32_t d, float *mag) {
	/* np.linalg.norm(axis=1) + nan_to_num(nan=0) */
	for (int64_t i = 0; i < n; i++) {
		double acc = 0.0;
		const float *r = x + i * (int64_t)d;
		for (int32_t k = 0; k < d; k++) acc += (double)r[k] * (double)r[k];
		float m = (float)sqrt(acc);
		if (m != m) m = 0.0f;
		mag[i] = m;
	}
}

void vko_normalize_rows(const float *x, int64_t n, int32_t d, float *out) {
	/* data = unmodified / magnitudes[:, None]; nan_to_num(nan=0).
	 * The "vanishing" rows are NOT zeroed: the reference's fill(0) acts on a copy
	 * (vectors.py:78) -- only 0/0 = NaN becomes 0. */
	for (int64_t i = 0; i < n; i++) {
		const float *r = x + i * (int64_t)d;
		float m;
		vko_magnitudes(r, 1, d, &m);
		for (int32_t k = 0; k < d; k++) {
			float v = r[k] / m;
			if (v != v) v = 0.0f;
			out[i * (int64_t)d + k] = v;
		}
	}
}

void vko_normalize_rows_bf16(const float *x, int64_t n, int32_t d, uint16_t *out, float *mag) {
	for (int64_t i = 0; i < n; i++) {
		const float *r = x + i * (int64_t)d;
		float m;
		vko_magnitudes(r, 1, d, &m);
		if (mag) mag[i] = m;
		for (int32_t k = 0; k < d; k++) {
			float v = r[k] / m;
			if (v != v) v = 0.0f;
			out[i * (int64_t)d + k] = vko_f32_to_bf16(v);
		}
	}
}

/* ------------------------------------------------------------------------- */
/* a2, a4, a5, a7: cosine of unit rows + clip                                */
/* vectorian/sim/vector.py:66-78 (normalized @ normalized.T),                */
/* vectorian/core/cpp/metric/metric.h:28-30 (clip to [0,1]),                 */
/* vectorian/core/cpp/metric/contextual.cpp:46-56                            */
/* ------------------------------------------------------------------------- */

static inline float clip01(float x) {
	/* xt::clip(sim, 0, 1) */
	if (!(x > 0.0f)) return 0.0f; /* also maps NaN to 0 */
	if (x > 1.0f) return 1.0f;
	return x;
}

static inline float dot_f32(const float *a, const float *b, int32_t d) {
	/* products of bf16-valued floats are exact; fixed 4-way double accumulation */
	double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
	int32_t k = 0;
	for (; k + 4 <= d; k += 4) {
		s0 += (double)a[k] * (double)b[k];
		s1 += (double)a[k + 1] * (double)b[k + 1];
		s2 += (double)a[k + 2] * (double)b[k + 2];
		s3 += (double)a[k + 3] * (double)b[k + 3];
	}
	for (; k < d; k++) s0 += (double)a[k] * (double)b[k];
	return (float)((s0 + s1) + (s2 + s3));
}

void vko_sim_f32(const float *X, int64_t n_rows, int32_t d, const float *Q, int32_t len_t, float *S) {
	for (int64_t i = 0; i < n_rows; i++)
		for (int32_t j = 0; j < len_t; j++)
			S[i * len_t + j] = clip01(dot_f32(X + i * (int64_t)d, Q + (int64_t)j * d, d));
}

static void widen_rows(const uint16_t *in, int64_t n, float *out) {
	for (int64_t i = 0; i < n; i++) out[i] = vko_bf16_to_f32(in[i]);
}

void vko_sim_bf16(const uint16_t *X, int64_t n_rows, int32_t d, const uint16_t *Q, int32_t len_t, float *S) {
	float *q = (float *)malloc(sizeof(float) * (size_t)len_t * d);
	float *x = (float *)malloc(sizeof(float) * (size_t)d);
	widen_rows(Q, (int64_t)len_t * d, q);
	for (int64_t i = 0; i < n_rows; i++) {
		widen_rows(X + i * (int64_t)d, d, x);
		for (int32_t j = 0; j < len_t; j++)
			S[i * len_t + j] = clip01(dot_f32(x, q + (int64_t)j * d, d));
	}
	free(q);
	free(x);
}

/* ------------------------------------------------------------------------- */
/* a3: static similarity table                                               */
/* vectorian/core/cpp/metric/static.cpp:9-78: sim = VectorSim(E, Q); then    */
/* sim[id(t_j), j] = 1 (:58-67); then clip (:75)                             */
/* ------------------------------------------------------------------------- */

void vko_sim_table_static_bf16(const uint16_t *E, int32_t V, int32_t d, const uint16_t *Q, int32_t len_t,
	const int32_t *q_ids, float *table) {

	float *q = (float *)malloc(sizeof(float) * (size_t)len_t * d);
	float *x = (float *)malloc(sizeof(float) * (size_t)d);
	widen_rows(Q, (int64_t)len_t * d, q);
	for (int32_t i = 0; i < V; i++) {
		widen_rows(E + (int64_t)i * d, d, x);
		for (int32_t j = 0; j < len_t; j++)
			table[(int64_t)i * len_t + j] = dot_f32(x, q + (int64_t)j * d, d);
	}
	if (q_ids) {
		for (int32_t j = 0; j < len_t; j++)
			if (q_ids[j] >= 0 && q_ids[j] < V) table[(int64_t)q_ids[j] * len_t + j] = 1.0f;
	}
	for (int64_t i = 0; i < (int64_t)V * len_t; i++) table[i] = clip01(table[i]);
	free(q);
	free(x);
}

/* ------------------------------------------------------------------------- */
/* gap costs: pyalign.gaps (third party, absent), restated per SURVEY A.4.   */
/* Call sites: vectorian/core/cpp/metric/alignment.h:296-304, :365-370       */
/* ------------------------------------------------------------------------- */

float vko_gap_cost(const vko_gap *g, int32_t k) {
	if (k <= 0) return 0.0f;
	switch (g->kind) {
	case VKO_GAP_LINEAR:
		return g->u * (float)k;
	case VKO_GAP_AFFINE:
		return g->u + g->v * (float)k;
	default:
		if (g->table && k < g->n_table) return g->table[k];
		return INFINITY;
	}
}

/* ------------------------------------------------------------------------- */
/* a10: alignment DP + traceback                                             */
/*                                                                           */
/* pyalign (github.com/poke1024/pyalign, submodule pinned at branch 0.3.6;   */
/* absent from /root/reference) is called at                                 */
/* vectorian/core/cpp/metric/alignment.h:254-269: solve(pairwise, len_s,     */
/* len_t) then alignment<Factory>() which reports edges (u, v) via           */
/* add_edge -> flow.set(v, u)  (:194-196), i.e. mapping[v] = u.              */
/* Options: MAXIMIZE, one optimal alignment, float values, int16 indices     */
/* (:357-358, :388-406); locality default LOCAL (:363-364).                  */
/*                                                                           */
/* Restated algorithm (docstrings vectorian/alignment.py:50-76,  */
/* 100-109, 133-165):                                                        */
/*   H[u][v] = max( [0 if LOCAL], H[u-1][v-1] + S[u-1][v-1],                 */
/*                  max_k H[u-k][v] - w_s(k), max_k H[u][v-k] - w_t(k) )     */
/* Solver choice as pyalign does it: both gaps linear -> O(nm) 3-candidate   */
/* recurrence; both linear/affine -> Gotoh E/F; otherwise the general        */
/* (Waterman-Smith-Beyer) recurrence over gap tables.                        */
/*                                                                           */
/* Tie rules DEFINED here (upstream unverifiable -> "parity unpinned"):      */
/* candidates are tried in the order zero (LOCAL only), diagonal, gap in s   */
/* (k = 1, 2, ..), gap in t (k = 1, 2, ..) and replace the incumbent only    */
/* when strictly greater.  LOCAL / SEMIGLOBAL start cell = first maximum in  */
/* row-major order (u outer, v inner), borders first.  All arithmetic fp32,  */
/* each candidate is exactly one add or one subtract of an fp32 H value.     */
/* ------------------------------------------------------------------------- */

enum { D_STOP = 0, D_DIAG = 1, D_UP = 2, D_LEFT = 3 };

typedef struct {
	float *H, *E, *F;
	uint8_t *dir;   /* D_* of H */
	int16_t *dk;    /* gap length for D_UP / D_LEFT (general solver) */
	uint8_t *eext;  /* affine: E extended (1) or opened (0) */
	uint8_t *fext;
	float *ws, *wt;
	size_t cap;
} dp_buf;

static __thread dp_buf tl_buf;

static dp_buf *get_buf(int32_t len_s, int32_t len_t) {
	size_t need = (size_t)(len_s + 1) * (size_t)(len_t + 1);
	dp_buf *b = &tl_buf;
	if (b->cap < need) {
		free(b->H); free(b->E); free(b->F); free(b->dir); free(b->dk); free(b->eext); free(b->fext);
		free(b->ws); free(b->wt);
		size_t cap = need * 2 + 64;
		b->H = (float *)malloc(cap * sizeof(float));
		b->E = (float *)malloc(cap * sizeof(float));
		b->F = (float *)malloc(cap * sizeof(float));
		b->dir = (uint8_t *)malloc(cap);
		b->dk = (int16_t *)malloc(cap * sizeof(int16_t));
		b->eext = (uint8_t *)malloc(cap);
		b->fext = (uint8_t *)malloc(cap);
		b->ws = (float *)malloc(cap * sizeof(float));
		b->wt = (float *)malloc(cap * sizeof(float));
		b->cap = cap;
	}
	return b;
}

/* pick the cell the traceback starts from; returns its value */
static float start_cell(const float *H, int32_t W, int32_t len_s, int32_t len_t, int32_t locality,
	int32_t *pu, int32_t *pv) {

	if (locality == VKO_GLOBAL) {
		*pu = len_s;
		*pv = len_t;
		return H[(size_t)len_s * W + len_t];
	}
	/* LOCAL: max over all cells; SEMIGLOBAL: max over last row and last column
	 * ("end gaps free", vectorian/alignment.py:100-109).  Border cells are 0. */
	float best = 0.0f;
	int32_t bu = 0, bv = 0;
	for (int32_t u = 1; u <= len_s; u++) {
		for (int32_t v = 1; v <= len_t; v++) {
			if (locality == VKO_SEMIGLOBAL && !(u == len_s || v == len_t)) continue;
			float h = H[(size_t)u * W + v];
			if (h > best) {
				best = h;
				bu = u;
				bv = v;
			}
		}
	}
	*pu = bu;
	*pv = bv;
	return best;
}

static void clear_mapping(int16_t *mapping, int32_t len_t) {
	if (mapping)
		for (int32_t j = 0; j < len_t; j++) mapping[j] = -1;
}

static int align_linear(const float *S, int32_t ld, int32_t len_s, int32_t len_t, int32_t locality,
	float gs, float gt, float *raw, int16_t *mapping) {

	dp_buf *b = get_buf(len_s, len_t);
	const int32_t W = len_t + 1;
	float *H = b->H;
	uint8_t *dir = b->dir;
	const int local = (locality == VKO_LOCAL);
	const int global = (locality == VKO_GLOBAL);

	H[0] = 0.0f;
	for (int32_t v = 1; v <= len_t; v++) H[v] = global ? -(gt * (float)v) : 0.0f;
	for (int32_t u = 1; u <= len_s; u++) {
		float *Hu = H + (size_t)u * W;
		const float *Hp = Hu - W;
		const float *Su = S + (size_t)(u - 1) * ld;
		uint8_t *du = dir + (size_t)u * W;
		Hu[0] = global ? -(gs * (float)u) : 0.0f;
		for (int32_t v = 1; v <= len_t; v++) {
			float best;
			uint8_t d;
			float c = Hp[v - 1] + Su[v - 1];
			if (local) {
				best = 0.0f;
				d = D_STOP;
				if (c > best) { best = c; d = D_DIAG; }
			} else {
				best = c;
				d = D_DIAG;
			}
			c = Hp[v] - gs;
			if (c > best) { best = c; d = D_UP; }
			c = Hu[v - 1] - gt;
			if (c > best) { best = c; d = D_LEFT; }
			Hu[v] = best;
			du[v] = d;
		}
	}

	int32_t u, v;
	*raw = start_cell(H, W, len_s, len_t, locality, &u, &v);
	clear_mapping(mapping, len_t);
	if (mapping) {
		while (u > 0 && v > 0) {
			uint8_t d = dir[(size_t)u * W + v];
			if (d == D_STOP) break;
			if (d == D_DIAG) { mapping[v - 1] = (int16_t)(u - 1); u--; v--; }
			else if (d == D_UP) u--;
			else v--;
		}
	}
	return 0;
}

static int align_general(const float *S, int32_t ld, int32_t len_s, int32_t len_t, int32_t locality,
	const vko_gap *gs, const vko_gap *gt, float *raw, int16_t *mapping) {

	dp_buf *b = get_buf(len_s, len_t);
	const int32_t W = len_t + 1;
	float *H = b->H;
	uint8_t *dir = b->dir;
	int16_t *dk = b->dk;
	float *ws = b->ws, *wt = b->wt;
	const int local = (locality == VKO_LOCAL);
	const int global = (locality == VKO_GLOBAL);

	for (int32_t k = 0; k <= len_s; k++) ws[k] = vko_gap_cost(gs, k);
	for (int32_t k = 0; k <= len_t; k++) wt[k] = vko_gap_cost(gt, k);

	H[0] = 0.0f;
	for (int32_t v = 1; v <= len_t; v++) H[v] = global ? -wt[v] : 0.0f;
	for (int32_t u = 1; u <= len_s; u++) {
		float *Hu = H + (size_t)u * W;
		const float *Su = S + (size_t)(u - 1) * ld;
		Hu[0] = global ? -ws[u] : 0.0f;
		for (int32_t v = 1; v <= len_t; v++) {
			float best;
			uint8_t d;
			int16_t kk = 0;
			float c = Hu[v - 1 - W] + Su[v - 1];
			if (local) {
				best = 0.0f;
				d = D_STOP;
				if (c > best) { best = c; d = D_DIAG; }
			} else {
				best = c;
				d = D_DIAG;
			}
			for (int32_t k = 1; k <= u; k++) {
				c = H[(size_t)(u - k) * W + v] - ws[k];
				if (c > best) { best = c; d = D_UP; kk = (int16_t)k; }
			}
			for (int32_t k = 1; k <= v; k++) {
				c = Hu[v - k] - wt[k];
				if (c > best) { best = c; d = D_LEFT; kk = (int16_t)k; }
			}
			Hu[v] = best;
			dir[(size_t)u * W + v] = d;
			dk[(size_t)u * W + v] = kk;
		}
	}

	int32_t u, v;
	*raw = start_cell(H, W, len_s, len_t, locality, &u, &v);
	clear_mapping(mapping, len_t);
	if (mapping) {
		while (u > 0 && v > 0) {
			uint8_t d = dir[(size_t)u * W + v];
			if (d == D_STOP) break;
			if (d == D_DIAG) { mapping[v - 1] = (int16_t)(u - 1); u--; v--; }
			else if (d == D_UP) u -= dk[(size_t)u * W + v];
			else v -= dk[(size_t)u * W + v];
		}
	}
	return 0;
}

/* Gotoh: w(k) = a + b*k.  E = best value of a path ending in a gap that skips
 * s tokens (vertical), F = ... skips t tokens (horizontal). */
static int align_affine(const float *S, int32_t ld, int32_t len_s, int32_t len_t, int32_t locality,
	float as, float bs, float at, float bt, float *raw, int16_t *mapping) {

	dp_buf *b = get_buf(len_s, len_t);
	const int32_t W = len_t + 1;
	float *H = b->H, *E = b->E, *F = b->F;
	uint8_t *dir = b->dir, *eext = b->eext, *fext = b->fext;
	const int local = (locality == VKO_LOCAL);
	const int global = (locality == VKO_GLOBAL);
	const float open_s = as + bs, open_t = at + bt;

	H[0] = 0.0f; E[0] = -INFINITY; F[0] = -INFINITY;
	for (int32_t v = 1; v <= len_t; v++) {
		H[v] = global ? -(at + bt * (float)v) : 0.0f;
		E[v] = -INFINITY;
		F[v] = global ? H[v] : -INFINITY;
	}
	for (int32_t u = 1; u <= len_s; u++) {
		float *Hu = H + (size_t)u * W, *Eu = E + (size_t)u * W, *Fu = F + (size_t)u * W;
		const float *Hp = Hu - W, *Ep = Eu - W;
		const float *Su = S + (size_t)(u - 1) * ld;
		Hu[0] = global ? -(as + bs * (float)u) : 0.0f;
		Eu[0] = global ? Hu[0] : -INFINITY;
		Fu[0] = -INFINITY;
		for (int32_t v = 1; v <= len_t; v++) {
			/* gap of length 1 (open) first, longer (extend) only if strictly greater */
			float e = Hp[v] - open_s;
			uint8_t ee = 0;
			float c = Ep[v] - bs;
			if (c > e) { e = c; ee = 1; }
			float f = Hu[v - 1] - open_t;
			uint8_t fe = 0;
			c = Fu[v - 1] - bt;
			if (c > f) { f = c; fe = 1; }

			float best;
			uint8_t d;
			c = Hp[v - 1] + Su[v - 1];
			if (local) {
				best = 0.0f;
				d = D_STOP;
				if (c > best) { best = c; d = D_DIAG; }
			} else {
				best = c;
				d = D_DIAG;
			}
			if (e > best) { best = e; d = D_UP; }
			if (f > best) { best = f; d = D_LEFT; }
			Hu[v] = best; Eu[v] = e; Fu[v] = f;
			size_t idx = (size_t)u * W + v;
			dir[idx] = d; eext[idx] = ee; fext[idx] = fe;
		}
	}

	int32_t u, v;
	*raw = start_cell(H, W, len_s, len_t, locality, &u, &v);
	clear_mapping(mapping, len_t);
	if (mapping) {
		int state = 0; /* 0 = H, 1 = E, 2 = F */
		while (u > 0 && v > 0) {
			size_t idx = (size_t)u * W + v;
			if (state == 0) {
				uint8_t d = dir[idx];
				if (d == D_STOP) break;
				if (d == D_DIAG) { mapping[v - 1] = (int16_t)(u - 1); u--; v--; }
				else if (d == D_UP) state = 1;
				else state = 2;
			} else if (state == 1) {
				if (!eext[idx]) state = 0;
				u--;
			} else {
				if (!fext[idx]) state = 0;
				v--;
			}
		}
	}
	return 0;
}

int vko_align_general(const float *S, int32_t ld, int32_t len_s, int32_t len_t, int32_t locality,
	const vko_gap *gap_s, const vko_gap *gap_t, float *raw_score, int16_t *mapping) {
	if (len_s < 1 || len_t < 1 || len_s > VKO_MAX_LEN_S || len_t > VKO_MAX_LEN_T) return 1;
	return align_general(S, ld, len_s, len_t, locality, gap_s, gap_t, raw_score, mapping);
}

int vko_align(const float *S, int32_t ld, int32_t len_s, int32_t len_t, int32_t locality,
	const vko_gap *gap_s, const vko_gap *gap_t, float *raw_score, int16_t *mapping) {

	if (len_s < 1 || len_t < 1 || len_s > VKO_MAX_LEN_S || len_t > VKO_MAX_LEN_T) return 1;
	const int ks = gap_s->kind, kt = gap_t->kind;
	if (ks == VKO_GAP_LINEAR && kt == VKO_GAP_LINEAR)
		return align_linear(S, ld, len_s, len_t, locality, gap_s->u, gap_t->u, raw_score, mapping);
	if ((ks == VKO_GAP_LINEAR || ks == VKO_GAP_AFFINE) && (kt == VKO_GAP_LINEAR || kt == VKO_GAP_AFFINE)) {
		const float as = ks == VKO_GAP_AFFINE ? gap_s->u : 0.0f;
		const float bs = ks == VKO_GAP_AFFINE ? gap_s->v : gap_s->u;
		const float at = kt == VKO_GAP_AFFINE ? gap_t->u : 0.0f;
		const float bt = kt == VKO_GAP_AFFINE ? gap_t->v : gap_t->u;
		return align_affine(S, ld, len_s, len_t, locality, as, bs, at, bt, raw_score, mapping);
	}
	return align_general(S, ld, len_s, len_t, locality, gap_s, gap_t, raw_score, mapping);
}

/* ------------------------------------------------------------------------- */
/* a11: reference_score + Score                                              */
/* vectorian/core/cpp/metric/alignment.h:84-106; match/match.h:112-132       */
/* (max_similarity_for_t == 1 for static/contextual slices, slice/static.h:  */
/* 94-100), match/match.h:295-307                                            */
/* ------------------------------------------------------------------------- */

float vko_score(float raw, int32_t len_t, int32_t n_matched, float submatch_weight, float boost) {
	const float total = (float)len_t;
	const float matched = (float)n_matched;
	const float unmatched_weight = powf((total - matched) / total, submatch_weight);
	const float ref = matched + unmatched_weight * (total - matched);
	return (raw / ref) * boost;
}

/* ------------------------------------------------------------------------- */
/* a14-a17: relaxed word mover's distance                                    */
/* vectorian/core/cpp/alignment/bow.h:204-275 (BOWBuilder),                  */
/* :281-333 (UniqueTokensBOWBuilder); alignment/wmd.h:107-135 (distance      */
/* matrix), :287-416 (RelaxedSolver)                                         */
/* ------------------------------------------------------------------------- */

typedef struct { int32_t id; int16_t pos; int8_t doc; } ref_token;

static int cmp_ref_token(const void *a, const void *b) {
	const ref_token *x = (const ref_token *)a, *y = (const ref_token *)b;
	if (x->id != y->id) return x->id < y->id ? -1 : 1;
	/* std::sort is unstable; make it deterministic: s before t, then position */
	if (x->doc != y->doc) return x->doc < y->doc ? -1 : 1;
	return (x->pos > y->pos) - (x->pos < y->pos);
}

typedef struct { float d; int32_t j; int32_t pos; } dist_ref;

static int cmp_dist_ref(const void *a, const void *b) {
	const dist_ref *x = (const dist_ref *)a, *y = (const dist_ref *)b;
	if (x->d != y->d) return x->d < y->d ? -1 : 1;
	return (x->pos > y->pos) - (x->pos < y->pos);
}

double vko_emd(const double *a, int32_t n, const double *b, int32_t m, const double *C, double *flow);

float vko_wmd(const float *S, int32_t ld, int32_t len_s, int32_t len_t,
	const int32_t *ids_s, const int32_t *ids_t,
	int32_t relaxed, int32_t injective, int32_t symmetric, int32_t normalize_bow) {

	if (len_s <= 0 || len_t <= 0) return 0.0f;
	/* WMD::operator() (wmd.h:441-449) and FullSolver (wmd.h:201-209) reject these combinations */
	if (symmetric && !normalize_bow) return NAN;
	if (!relaxed && (injective || symmetric)) return NAN;
	const int32_t K = len_s + len_t;
	/* per doc (0 = s, 1 = t): bow over joint vocabulary, vocab list, first position */
	float *bow[2];
	int32_t *vocab[2], nvocab[2] = {0, 0}, *first_pos[2];
	int32_t w_sum[2] = {0, 0};
	for (int c = 0; c < 2; c++) {
		bow[c] = (float *)calloc((size_t)K, sizeof(float));
		vocab[c] = (int32_t *)malloc(sizeof(int32_t) * (size_t)K);
		first_pos[c] = (int32_t *)malloc(sizeof(int32_t) * (size_t)K);
		for (int32_t i = 0; i < K; i++) first_pos[c][i] = -1;
	}
	int32_t vocab_size;
	if (ids_s && ids_t) {
		ref_token *z = (ref_token *)malloc(sizeof(ref_token) * (size_t)K);
		int32_t k = 0;
		for (int32_t i = 0; i < len_s; i++) { z[k].id = ids_s[i]; z[k].pos = (int16_t)i; z[k].doc = 0; k++; }
		for (int32_t i = 0; i < len_t; i++) { z[k].id = ids_t[i]; z[k].pos = (int16_t)i; z[k].doc = 1; k++; }
		qsort(z, (size_t)K, sizeof(ref_token), cmp_ref_token);
		int32_t cur = z[0].id, vi = 0;
		for (int32_t i = 0; i < K; i++) {
			if (z[i].id != cur) { cur = z[i].id; vi++; }
			const int c = z[i].doc;
			bow[c][vi] += 1.0f;
			w_sum[c] += 1;
			if (first_pos[c][vi] < 0) {
				first_pos[c][vi] = z[i].pos;
				vocab[c][nvocab[c]++] = vi;
			}
		}
		vocab_size = vi + 1;
		free(z);
	} else {
		int32_t off = 0;
		const int32_t lens[2] = {len_s, len_t};
		for (int c = 0; c < 2; c++) {
			for (int32_t j = 0; j < lens[c]; j++) {
				bow[c][off + j] = 1.0f;
				vocab[c][j] = off + j;
				first_pos[c][off + j] = j;
			}
			nvocab[c] = lens[c];
			w_sum[c] = lens[c];
			off += lens[c];
		}
		vocab_size = K;
	}
	if (normalize_bow) {
		for (int c = 0; c < 2; c++) {
			const float s = (float)w_sum[c];
			for (int32_t i = 0; i < nvocab[c]; i++) bow[c][vocab[c][i]] /= s;
		}
	}

	/* distance matrix over vocabulary pairs, written symmetrically in the
	 * reference's loop order (wmd.h:121-133; later writes win) */
	float *D = (float *)malloc(sizeof(float) * (size_t)vocab_size * vocab_size);
	for (int64_t i = 0; i < (int64_t)vocab_size * vocab_size; i++) D[i] = 1.0f;
	for (int32_t a = 0; a < nvocab[0]; a++) {
		const int32_t u = vocab[0][a], i = first_pos[0][u];
		for (int32_t bidx = 0; bidx < nvocab[1]; bidx++) {
			const int32_t v = vocab[1][bidx], j = first_pos[1][v];
			float d = 1.0f - S[(size_t)i * ld + j];
			if (!(d > 0.0f)) d = 0.0f;
			D[(size_t)u * vocab_size + v] = d;
			D[(size_t)v * vocab_size + u] = d;
		}
	}

	if (!relaxed) {
		/* FullSolver (wmd.h:194-270): exact EMD between the two histograms over the joint vocabulary,
		 * transport.h:91-145 -> pyemd emd_hat_gd_metric<double>; score = sum((1-D)*G)/sum(G) (:247) */
		double *P = (double *)calloc((size_t)vocab_size, sizeof(double));
		double *Qm = (double *)calloc((size_t)vocab_size, sizeof(double));
		double *Cd = (double *)malloc(sizeof(double) * (size_t)vocab_size * vocab_size);
		double *G = (double *)malloc(sizeof(double) * (size_t)vocab_size * vocab_size);
		for (int32_t i = 0; i < vocab_size; i++) { P[i] = (double)bow[1][i]; Qm[i] = (double)bow[0][i]; }
		for (int64_t i = 0; i < (int64_t)vocab_size * vocab_size; i++) Cd[i] = (double)D[i];
		vko_emd(P, vocab_size, Qm, vocab_size, Cd, G);
		double num = 0.0, den = 0.0;
		for (int64_t i = 0; i < (int64_t)vocab_size * vocab_size; i++) {
			const float gq = (float)G[i];
			num += (double)((1.0f - D[i]) * gq);
			den += (double)gq;
		}
		free(P); free(Qm); free(Cd); free(G);
		for (int c = 0; c < 2; c++) { free(bow[c]); free(vocab[c]); free(first_pos[c]); }
		free(D);
		return den > 0.0 ? (float)(num / den) : 0.0f;
	}

	/* RelaxedSolver: c = 0 moves t -> s, c = 1 moves s -> t (wmd.h:303-306) */
	const int order[2] = {1, 0};
	dist_ref *cand = (dist_ref *)malloc(sizeof(dist_ref) * (size_t)K);
	float cost = 0.0f;
	for (int c = 0; c < 2; c++) {
		const int d1 = order[c], d2 = order[1 - c];
		const float *w1 = bow[d1], *w2 = bow[d2];
		float acc = 0.0f;
		for (int32_t a = 0; a < nvocab[d1]; a++) {
			const int32_t i = vocab[d1][a];
			if (injective) {
				float best = 3.402823466e+38F;
				int32_t best_j = -1;
				for (int32_t bidx = 0; bidx < nvocab[d2]; bidx++) {
					const int32_t j = vocab[d2][bidx];
					const float d = D[(size_t)i * vocab_size + j];
					if (d < best) { best = d; best_j = j; }
				}
				const float d = best_j >= 0 ? best : 1.0f;
				acc += w1[i] * d;
			} else {
				float remaining = w1[i];
				int32_t nc = 0;
				for (int32_t bidx = 0; bidx < nvocab[d2]; bidx++) {
					const int32_t j = vocab[d2][bidx];
					cand[nc].d = D[(size_t)i * vocab_size + j];
					cand[nc].j = j;
					cand[nc].pos = first_pos[d2][j];
					nc++;
				}
				/* upstream pops a binary heap keyed on the distance alone (wmd.h:29-37,352-358): the order among
				 * equal distances is whatever std::pop_heap yields.  Defined here: by first position in the document. */
				qsort(cand, (size_t)nc, sizeof(dist_ref), cmp_dist_ref);
				for (int32_t r = 0; r < nc; r++) {
					const int32_t target = cand[r].j;
					if (remaining <= w2[target]) {
						acc += remaining * cand[r].d;
						break;
					} else {
						remaining -= w2[target];
						acc += w2[target] * cand[r].d;
					}
				}
				/* wmd.h:373-375 as written: `remaining` keeps its value when the loop above breaks, so the
				 * last partial shipment is charged a second time at the maximum distance.  Restated as is. */
				if (remaining > 0.0f) acc += remaining * 1.0f;
			}
		}
		if (!normalize_bow) acc /= (float)w_sum[d1];
		if (!symmetric) { cost = acc; break; }
		else if (acc > cost) cost = acc;
	}

	const float max_cost = normalize_bow ? 1.0f : (float)len_t;
	for (int c = 0; c < 2; c++) { free(bow[c]); free(vocab[c]); free(first_pos[c]); }
	free(D);
	free(cand);
	return (max_cost - cost) / max_cost; /* cost_to_score, wmd.h:138-140 */
}

float vko_rwmd(const float *S, int32_t ld, int32_t len_s, int32_t len_t,
	const int32_t *ids_s, const int32_t *ids_t,
	int32_t injective, int32_t symmetric, int32_t normalize_bow) {
	return vko_wmd(S, ld, len_s, len_t, ids_s, ids_t, 1, injective, symmetric, normalize_bow);
}

/* ------------------------------------------------------------------------- */
/* exact EMD: successive shortest paths on the bipartite transportation      */
/* problem, double precision.  Stands in for pyemd's                         */
/* emd_hat_gd_metric<double> (vectorian/core/cpp/alignment/transport.h:70,   */
/* 125-126; pyemd absent -> its internal quantisation is unpinned).          */
/* Moves min(sum a, sum b) units of mass at minimum cost.                    */
/* ------------------------------------------------------------------------- */

double vko_emd(const double *a, int32_t n, const double *b, int32_t m, const double *C, double *flow) {
	const int32_t N = n + m;
	const double EPS = 1e-13;
	double *f = (double *)calloc((size_t)n * m, sizeof(double));
	double *sup = (double *)malloc(sizeof(double) * (size_t)n);
	double *dem = (double *)malloc(sizeof(double) * (size_t)m);
	double *pot = (double *)calloc((size_t)N, sizeof(double));
	double *dist = (double *)malloc(sizeof(double) * (size_t)N);
	int32_t *prev = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
	uint8_t *done = (uint8_t *)malloc((size_t)N);
	memcpy(sup, a, sizeof(double) * (size_t)n);
	memcpy(dem, b, sizeof(double) * (size_t)m);

	for (int iter = 0; iter < 100000; iter++) {
		int any = 0;
		for (int32_t i = 0; i < N; i++) { dist[i] = INFINITY; prev[i] = -1; done[i] = 0; }
		for (int32_t i = 0; i < n; i++)
			if (sup[i] > EPS) { dist[i] = 0.0; any = 1; }
		if (!any) break;
		int any_dem = 0;
		for (int32_t j = 0; j < m; j++) if (dem[j] > EPS) any_dem = 1;
		if (!any_dem) break;

		int32_t target = -1;
		for (;;) { /* dense Dijkstra on reduced costs */
			int32_t x = -1;
			double bd = INFINITY;
			for (int32_t i = 0; i < N; i++)
				if (!done[i] && dist[i] < bd) { bd = dist[i]; x = i; }
			if (x < 0) break;
			done[x] = 1;
			if (x >= n && dem[x - n] > EPS) { target = x; break; }
			if (x < n) {
				for (int32_t j = 0; j < m; j++) {
					if (done[n + j]) continue;
					double rc = C[(size_t)x * m + j] + pot[x] - pot[n + j];
					if (rc < 0) rc = 0;
					if (dist[x] + rc < dist[n + j]) { dist[n + j] = dist[x] + rc; prev[n + j] = x; }
				}
			} else {
				const int32_t j = x - n;
				for (int32_t i = 0; i < n; i++) {
					if (done[i] || !(f[(size_t)i * m + j] > EPS)) continue;
					double rc = -C[(size_t)i * m + j] + pot[x] - pot[i];
					if (rc < 0) rc = 0;
					if (dist[x] + rc < dist[i]) { dist[i] = dist[x] + rc; prev[i] = x; }
				}
			}
		}
		if (target < 0) break;
		const double dt = dist[target];
		for (int32_t i = 0; i < N; i++) pot[i] += (done[i] && dist[i] < dt) ? dist[i] : dt;

		/* bottleneck */
		double delta = dem[target - n];
		int32_t x = target;
		while (prev[x] >= 0) {
			const int32_t p = prev[x];
			if (p >= n) { /* backward edge sink p -> source x */
				const double cap = f[(size_t)x * m + (p - n)];
				if (cap < delta) delta = cap;
			}
			x = p;
		}
		if (sup[x] < delta) delta = sup[x];
		/* augment */
		sup[x] -= delta;
		dem[target - n] -= delta;
		x = target;
		while (prev[x] >= 0) {
			const int32_t p = prev[x];
			if (p < n) f[(size_t)p * m + (x - n)] += delta;
			else f[(size_t)x * m + (p - n)] -= delta;
			x = p;
		}
	}
	double cost = 0.0;
	for (int64_t i = 0; i < (int64_t)n * m; i++) cost += f[i] * C[i];
	if (flow) memcpy(flow, f, sizeof(double) * (size_t)n * m);
	free(f); free(sup); free(dem); free(pot); free(dist); free(prev); free(done);
	return cost;
}

/* ------------------------------------------------------------------------- */
/* a18: Word Rotator's Distance                                              */
/* vectorian/core/cpp/alignment/wrd.h:62-146: masses = magnitudes, each side */
/* normalised to 1 (:99-102); D[t][s] = max(0, 1 - S[s][t]) (:104-109);      */
/* score = sum((1-D)*G) / sum(G) (:139)                                      */
/* ------------------------------------------------------------------------- */

float vko_wrd(const float *S, int32_t ld, int32_t len_s, int32_t len_t,
	const float *mag_s, const float *mag_t, int32_t normalize_magnitudes) {

	if (len_s <= 0 || len_t <= 0) return 0.0f;
	double *a = (double *)malloc(sizeof(double) * (size_t)len_t);
	double *b = (double *)malloc(sizeof(double) * (size_t)len_s);
	double *C = (double *)malloc(sizeof(double) * (size_t)len_t * len_s);
	double *G = (double *)malloc(sizeof(double) * (size_t)len_t * len_s);
	float sum_t = 0.0f, sum_s = 0.0f;
	for (int32_t j = 0; j < len_t; j++) sum_t += mag_t[j];
	for (int32_t i = 0; i < len_s; i++) sum_s += mag_s[i];
	for (int32_t j = 0; j < len_t; j++) a[j] = normalize_magnitudes ? (double)(mag_t[j] / sum_t) : (double)mag_t[j];
	for (int32_t i = 0; i < len_s; i++) b[i] = normalize_magnitudes ? (double)(mag_s[i] / sum_s) : (double)mag_s[i];
	for (int32_t j = 0; j < len_t; j++)
		for (int32_t i = 0; i < len_s; i++) {
			float d = 1.0f - S[(size_t)i * ld + j];
			if (!(d > 0.0f)) d = 0.0f;
			C[(size_t)j * len_s + i] = (double)d;
		}
	vko_emd(a, len_t, b, len_s, C, G);
	double num = 0.0, den = 0.0;
	for (int64_t i = 0; i < (int64_t)len_t * len_s; i++) {
		const float g = (float)G[i];
		num += (double)((1.0f - (float)C[i]) * g);
		den += (double)g;
	}
	free(a); free(b); free(C); free(G);
	if (!(den > 0.0)) return 0.0f;
	return (float)(num / den);
}

/* ------------------------------------------------------------------------- */
/* whole-corpus search                                                       */
/* a8: Spans::iterate (vectorian/core/cpp/document.h:147-169), window size   */
/*     and step 1 over contiguous sentence spans;                            */
/* a9: MatcherImpl::run_matches (match/matcher_impl.h:71-109);               */
/* a12: ResultSet (result_set.h:32-60, 70-93; result_set.cpp:3-21) with the  */
/*     total order (score desc, sentence index desc) -- the deterministic    */
/*     part of Match::compare_by_score (match/match_impl.h:8-42); */
/*     admission score > min_score (metric/alignment.h:284).                 */
/* ------------------------------------------------------------------------- */

typedef struct { float score; float raw; int64_t g; } hit;

static inline int hit_better(const hit *a, const hit *b) {
	if (a->score != b->score) return a->score > b->score;
	return a->g > b->g;
}

typedef struct { hit *h; int32_t n, k; } topk;

static void topk_push(topk *t, hit x) {
	if (t->n == t->k && !hit_better(&x, &t->h[t->n - 1])) return;
	int32_t i = t->n < t->k ? t->n++ : t->k - 1;
	while (i > 0 && hit_better(&x, &t->h[i - 1])) { t->h[i] = t->h[i - 1]; i--; }
	t->h[i] = x;
}

typedef struct {
	const vko_corpus *c;
	const vko_query *q;
	const float *qf;        /* widened query [len_t x d] */
	const float *table;     /* static table [V x len_t] */
	int64_t s0, s1;
	topk tk;
	float *all_scores;
	int status;
} work;

static int score_sentence(const work *w, int64_t s, float *Sbuf, float *xbuf, float *raw_out, float *value_out,
	int16_t *mapping) {

	const vko_corpus *c = w->c;
	const vko_query *q = w->q;
	const int64_t t0 = c->sent_off[s], t1 = c->sent_end ? c->sent_end[s] : c->sent_off[s + 1];
	const int32_t len_s = (int32_t)(t1 - t0), len_t = q->len_t;
	if (len_s < 1) return 1; /* document.h:160 */
	if (len_s > VKO_MAX_LEN_S) return 2;

	if (c->layout == VKO_LAYOUT_CONTEXTUAL && q->S_rows) {
		/* slice/contextual.h:65-67: rows of the matrix the caller's sgemm produced (metric/contextual.cpp:26-63) */
		memcpy(Sbuf, q->S_rows + t0 * (int64_t)len_t, sizeof(float) * (size_t)len_s * len_t);
	} else if (c->layout == VKO_LAYOUT_CONTEXTUAL) {
		for (int32_t i = 0; i < len_s; i++) {
			const float *xr = xbuf;
			if (c->X_f32) xr = c->X_f32 + (t0 + i) * (int64_t)c->d;
			else widen_rows(c->X + (t0 + i) * (int64_t)c->d, c->d, xbuf);
			for (int32_t j = 0; j < len_t; j++)
				Sbuf[i * len_t + j] = clip01(dot_f32(xr, w->qf + (int64_t)j * c->d, c->d));
		}
	} else {
		for (int32_t i = 0; i < len_s; i++) {
			const float *row = w->table + (int64_t)c->tok_id[t0 + i] * len_t; /* slice/static.h:71-75 */
			for (int32_t j = 0; j < len_t; j++) Sbuf[i * len_t + j] = row[j];
		}
	}

	if (q->tag_weights) {
		/* TagWeightedSlice::similarity (slice/static.h:237-264): S * weight(i, j), zero at or below the threshold */
		for (int32_t i = 0; i < len_s; i++)
			for (int32_t j = 0; j < len_t; j++) {
				float wgt = q->tag_weights[j];
				if (c->pos_s && q->q_pos && c->pos_s[t0 + i] != q->q_pos[j]) wgt *= 1.0f - q->pos_mismatch_penalty;
				const float sc = Sbuf[i * len_t + j] * wgt;
				Sbuf[i * len_t + j] = sc <= q->similarity_threshold ? 0.0f : sc;
			}
	}

	const float boost = q->boost ? q->boost[s] : 1.0f; /* matcher_impl.h:99 */
	float raw = 0.0f;
	if (q->algorithm == VKO_ALG_ALIGN) {
		int16_t local_map[VKO_MAX_LEN_T];
		int16_t *m = mapping ? mapping : local_map;
		/* reference_score takes the matched weight of THIS slice's flow (metric/alignment.h:84-106; InjectiveFlow::max_score,
		 * match/match.h:112-132) -- upstream always has the flow.  With unit weights matched + (total - matched) is exact whatever
		 * the flow, so the traceback may be skipped; with tag weights it is not (an ulp), so it is taken. */
		const int need_map = mapping != NULL || q->submatch_weight != 0.0f || q->tag_weights != NULL;
		if (vko_align(Sbuf, len_t, len_s, len_t, q->locality, &q->gap_s, &q->gap_t, &raw, need_map ? m : NULL)) return 2;
		if (q->tag_weights) {
			/* reference_score with max_similarity_for_t = t_pos_weights (slice/static.h:280-286) */
			float total = 0.0f, matched_w = 0.0f;
			for (int32_t j = 0; j < len_t; j++) total += q->tag_weights[j];
			if (need_map) for (int32_t j = 0; j < len_t; j++) if (m[j] >= 0) matched_w += q->tag_weights[j];
			const float uw = powf((total - matched_w) / total, q->submatch_weight);
			const float ref = matched_w + uw * (total - matched_w);
			*value_out = (raw / ref) * boost;
			*raw_out = raw;
			return 0;
		}
		int32_t matched = 0;
		if (need_map) for (int32_t j = 0; j < len_t; j++) matched += m[j] >= 0;
		*value_out = vko_score(raw, len_t, matched, q->submatch_weight, boost);
	} else if (q->algorithm == VKO_ALG_RWMD) {
		const int32_t *ids_s = c->layout == VKO_LAYOUT_STATIC ? c->tok_id + t0 : NULL;
		const int32_t *ids_t = c->layout == VKO_LAYOUT_STATIC ? q->q_ids : NULL;
		int32_t key_s[VKO_MAX_LEN_S], key_t[VKO_MAX_LEN_T];
		if (q->tag_weights && ids_s && ids_t && c->tag_s && q->q_tag) {
			/* TagWeightedSlice::similarity_dependency() == TAGS (slice/static.h:233-235) selects the tagged BOW builder
			 * (metric/alignment.h:551-576): vocabulary entries are (token id, tag) pairs (bow.h:106-127, 150-176) */
			for (int32_t i = 0; i < len_s; i++) key_s[i] = ids_s[i] * 256 + (int32_t)(uint8_t)c->tag_s[t0 + i];
			for (int32_t j = 0; j < len_t; j++) key_t[j] = ids_t[j] * 256 + (int32_t)(uint8_t)q->q_tag[j];
			ids_s = key_s; ids_t = key_t;
		}
		raw = vko_wmd(Sbuf, len_t, len_s, len_t, ids_s, ids_t, !q->wmd_full, q->rwmd_injective, q->rwmd_symmetric, q->rwmd_normalize_bow);
		if (raw != raw) return 2;
		/* SparseFlow::max_score -> matched = sum of max_similarity_for_t = len_t (match.h:165-176) => ref = len_t;
		 * tag-weighted: max_similarity_for_t = t_pos_weights (slice/static.h:280-286) => matched = total = their sum */
		if (q->tag_weights) {
			float total = 0.0f;
			for (int32_t j = 0; j < len_t; j++) total += q->tag_weights[j];
			*value_out = (raw / total) * boost;
		} else *value_out = vko_score(raw, len_t, len_t, q->submatch_weight, boost);
	} else {
		float mag_s[VKO_MAX_LEN_S];
		for (int32_t i = 0; i < len_s; i++) mag_s[i] = c->X_mag ? c->X_mag[t0 + i] : 1.0f;
		float mag_t[VKO_MAX_LEN_T];
		for (int32_t j = 0; j < len_t; j++) mag_t[j] = q->Q_mag ? q->Q_mag[j] : 1.0f;
		raw = vko_wrd(Sbuf, len_t, len_s, len_t, mag_s, mag_t, q->wrd_normalize_magnitudes);
		if (q->tag_weights) {   /* DenseFlow::max_score (match.h:227-237): matched = total = sum of t_pos_weights */
			float total = 0.0f;
			for (int32_t j = 0; j < len_t; j++) total += q->tag_weights[j];
			*value_out = (raw / total) * boost;
		} else *value_out = vko_score(raw, len_t, len_t, q->submatch_weight, boost);
	}
	*raw_out = raw;
	return 0;
}

static void run_range(work *w) {
	const vko_query *q = w->q;
	float *Sbuf = (float *)malloc(sizeof(float) * (size_t)VKO_MAX_LEN_S * q->len_t);
	float *xbuf = (float *)malloc(sizeof(float) * (size_t)(w->c->d > 0 ? w->c->d : 1));
	for (int64_t s = w->s0; s < w->s1; s++) {
		float raw, value;
		const int r = score_sentence(w, s, Sbuf, xbuf, &raw, &value, NULL);
		if (r == 2) { w->status = 2; break; }
		if (r == 1) { if (w->all_scores) w->all_scores[s] = 0.0f; continue; }
		if (w->all_scores) w->all_scores[s] = value;
		if (value > q->min_score) {
			hit h = {value, raw, s};
			topk_push(&w->tk, h);
		}
	}
	free(Sbuf);
	free(xbuf);
}

/* one thread = one static sentence range, for every query of the batch in turn (the analogue
 * of the reference's thread-per-document pool, vectorian/index.py:544-558) */
typedef struct { work *items; int32_t n; } thread_arg;

static void *worker(void *arg) {
	thread_arg *ta = (thread_arg *)arg;
	for (int32_t i = 0; i < ta->n; i++) run_range(&ta->items[i]);
	return NULL;
}

int vko_find_many(const vko_corpus *c, const vko_query *qs, int32_t n_queries, vko_result *outs, int32_t n_threads) {
	if (n_queries < 1) return 0;
	for (int32_t i = 0; i < n_queries; i++)
		if (qs[i].len_t < 1 || qs[i].len_t > VKO_MAX_LEN_T || qs[i].max_matches < 1) return 1;
	if (n_threads < 1) n_threads = 1;
	if ((int64_t)n_threads > c->n_sentences) n_threads = c->n_sentences > 0 ? (int32_t)c->n_sentences : 1;

	float **qf = (float **)calloc((size_t)n_queries, sizeof(float *));
	float **table = (float **)calloc((size_t)n_queries, sizeof(float *));
	for (int32_t i = 0; i < n_queries; i++) {
		qf[i] = (float *)malloc(sizeof(float) * (size_t)qs[i].len_t * c->d);
		if (qs[i].Q_f32) memcpy(qf[i], qs[i].Q_f32, sizeof(float) * (size_t)qs[i].len_t * c->d);
		else widen_rows(qs[i].Q, (int64_t)qs[i].len_t * c->d, qf[i]);
		if (c->layout == VKO_LAYOUT_STATIC) {
			table[i] = (float *)malloc(sizeof(float) * (size_t)c->V * qs[i].len_t);
			if (c->E_f32) {
				/* static.cpp:9-78 on fp32 rows: sim = E . Q^T, sim[id(t_j)][j] = 1, clip */
				for (int32_t v = 0; v < c->V; v++)
					for (int32_t j = 0; j < qs[i].len_t; j++)
						table[i][(int64_t)v * qs[i].len_t + j] = dot_f32(c->E_f32 + (int64_t)v * c->d, qf[i] + (int64_t)j * c->d, c->d);
				if (qs[i].q_ids)
					for (int32_t j = 0; j < qs[i].len_t; j++)
						if (qs[i].q_ids[j] >= 0 && qs[i].q_ids[j] < c->V) table[i][(int64_t)qs[i].q_ids[j] * qs[i].len_t + j] = 1.0f;
				for (int64_t e = 0; e < (int64_t)c->V * qs[i].len_t; e++) table[i][e] = clip01(table[i][e]);
			} else vko_sim_table_static_bf16(c->E, c->V, c->d, qs[i].Q, qs[i].len_t, qs[i].q_ids, table[i]);
		}
	}

	work *ws = (work *)calloc((size_t)n_threads * n_queries, sizeof(work));
	thread_arg *ta = (thread_arg *)calloc((size_t)n_threads, sizeof(thread_arg));
	pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
	for (int32_t t = 0; t < n_threads; t++) {
		ta[t].items = ws + (size_t)t * n_queries;
		ta[t].n = n_queries;
		for (int32_t i = 0; i < n_queries; i++) {
			work *w = &ta[t].items[i];
			w->c = c; w->q = &qs[i]; w->qf = qf[i]; w->table = table[i];
			w->s0 = c->n_sentences * t / n_threads;
			w->s1 = c->n_sentences * (t + 1) / n_threads;
			w->tk.h = (hit *)malloc(sizeof(hit) * (size_t)qs[i].max_matches);
			w->tk.n = 0; w->tk.k = qs[i].max_matches;
			w->all_scores = outs[i].all_scores;
		}
	}
	if (n_threads == 1) worker(&ta[0]);
	else {
		for (int32_t t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, worker, &ta[t]);
		for (int32_t t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
	}

	int status = 0;
	for (int32_t i = 0; i < n_queries; i++) {
		const vko_query *q = &qs[i];
		const int32_t k = q->max_matches;
		/* ResultSet::extend (result_set.h:70-93): merge the bounded sets */
		topk all;
		all.h = (hit *)malloc(sizeof(hit) * (size_t)k);
		all.n = 0; all.k = k;
		for (int32_t t = 0; t < n_threads; t++) {
			work *w = &ta[t].items[i];
			if (w->status) status = w->status;
			for (int32_t j = 0; j < w->tk.n; j++) topk_push(&all, w->tk.h[j]);
			free(w->tk.h);
		}
		vko_result *out = &outs[i];
		out->n_out = all.n;
		if (status == 0) {
			float *Sbuf = (float *)malloc(sizeof(float) * (size_t)VKO_MAX_LEN_S * q->len_t);
			float *xbuf = (float *)malloc(sizeof(float) * (size_t)(c->d > 0 ? c->d : 1));
			work w0 = ta[0].items[i];
			for (int32_t j = 0; j < all.n; j++) {
				out->score[j] = all.h[j].score;
				if (out->raw) out->raw[j] = all.h[j].raw;
				out->sentence[j] = all.h[j].g;
				if (out->mapping && q->algorithm == VKO_ALG_ALIGN) {
					float raw, value;
					score_sentence(&w0, all.h[j].g, Sbuf, xbuf, &raw, &value, out->mapping + (size_t)j * q->len_t);
				}
			}
			free(Sbuf);
			free(xbuf);
		}
		free(all.h);
		free(qf[i]);
		free(table[i]);
	}
	free(ws); free(ta); free(th); free(qf); free(table);
	return status;
}

int vko_find(const vko_corpus *c, const vko_query *q, vko_result *out, int32_t n_threads) {
	return vko_find_many(c, q, 1, out, n_threads);
}
