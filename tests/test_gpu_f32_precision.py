"""-m gpu: precision = "f32" -- unit rows kept in fp32, cosines on v_mfma_f32_16x16x4_f32: the reference's own precision
(CosineSim is an fp32 sgemm, vectorian/sim/vector.py:66-78).  HIP against the oracle fed the same UNROUNDED fp32 rows:
scores within 1e-5 (fp32 against double accumulation), ids and tracebacks exact.  The default bf16 path, on the same
vectors, is the same algorithm on rounded vectors and sits a few 1e-4 away."""

import numpy as np
import pytest

from vectorian_amd import synth

from helpers import assert_same_results

pytestmark = pytest.mark.gpu

EXP5L = ("table", (1 - 2.0 ** (-np.arange(0, 513) / 5)).astype(np.float32))
AFF = ("affine", 0.2, 0.05)


def contextual(hip, corpus, d, precision, keep_magnitudes=False):
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=corpus["X"].shape[0], n_sentences=len(corpus["sent_off"]) - 1,
		keep_magnitudes=keep_magnitudes, precision=precision)
	c.append_vectors(corpus["X"], normalize=True)
	c.set_sentences(corpus["sent_off"])
	c.finalize()
	return c


@pytest.mark.parametrize("d", [50, 300, 768])
def test_f32_contextual(hip, oracle, d):
	corpus = synth.make_contextual_corpus(600, 1, 64, 1500, d)
	lens = np.diff(corpus["sent_off"])
	X = oracle.normalize_rows(corpus["X"])
	c = contextual(hip, corpus, d, "f32")
	cb = contextual(hip, corpus, d, "bf16")
	gap = 0.0
	for q in synth.make_queries(corpus, 3, 9):
		qv = q["vectors"]
		Q = oracle.normalize_rows(qv)
		for loc, ms, gaps in ((0, 0.0, (0.1, 0.1)), (0, 0.0, (EXP5L, EXP5L)), (1, -1e9, (EXP5L, EXP5L)), (2, -1e9, (AFF, AFF))):
			kw = dict(locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=10, min_score=ms)
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=corpus["sent_off"], X=X, Q=Q, want_all_scores=True, **kw)
			got = c.query(qv, q_normalize=True, **kw)
			assert_same_results(got.trimmed(), ref, score_tol=1e-5)
			np.testing.assert_allclose(c.last_scores()[lens > 0], ref["all_scores"][lens > 0], atol=1e-5)
			cb.query(qv, q_normalize=True, **kw)
			gap = max(gap, float(np.abs(cb.last_scores()[lens > 0] - ref["all_scores"][lens > 0]).max()))
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=corpus["sent_off"], X=X, Q=Q, algorithm=oracle.ALG_RWMD, max_matches=10)
		got = c.query(qv, q_normalize=True, algorithm=hip.VK_ALG_RWMD, max_matches=10)
		assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=1e-5, tie_tol=1e-5)
	assert 2e-5 < gap < 5e-3, gap   # what rounding the vectors to bf16 costs against the fp32 arithmetic
	# long query, long slices: the same tiles feed the wide kernel and the second launch
	qv = synth.make_queries(corpus, 1, 24)[0]["vectors"]
	ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=corpus["sent_off"], X=X, Q=oracle.normalize_rows(qv), gap_s=0.1, gap_t=0.1, max_matches=8)
	assert_same_results(c.query(qv, q_normalize=True, gap_s=0.1, gap_t=0.1, max_matches=8).trimmed(), ref, score_tol=1e-5)
	c.close()
	cb.close()


def test_f32_static_wrd_and_span(hip, oracle):
	corpus = synth.make_static_corpus(800, 1, 40, 500, 100, seed=4)
	rng = np.random.default_rng(5)
	E = (corpus["E"] * rng.lognormal(0, 0.3, size=(500, 1))).astype(np.float32)
	En, emag = oracle.normalize_rows(E), oracle.magnitudes(E)
	off, ids = corpus["sent_off"], corpus["tok_id"]
	c = hip.Corpus(layout=hip.VK_LAYOUT_STATIC, d=100, n_tokens=len(ids), n_sentences=800, vocab_size=500, keep_magnitudes=True, precision="f32")
	c.append_vectors(E, normalize=True)
	c.set_token_ids(ids)
	c.set_sentences(off)
	c.finalize()
	for _ in range(2):
		q_ids = rng.integers(0, 60, size=8).astype(np.int32)
		for loc, ms, gaps in ((0, 0.0, (EXP5L, EXP5L)), (1, -1e9, (0.1, 0.1))):
			kw = dict(locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=10, min_score=ms)
			ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=100, sent_off=off, tok_id=ids, E=En, Q=En[q_ids], q_ids=q_ids, **kw)
			got = c.query(E[q_ids], q_token_ids=q_ids, q_normalize=True, **kw)
			assert_same_results(got.trimmed(), ref, score_tol=1e-5, check_mapping=False)
			# tracebacks: identical up to the choice among repetitions of one word inside the slice (equal similarities;
			# the fp32 cosine of the device and the double-accumulated one of the oracle differ in the last bit, and
			# which of two equal-valued paths sums to the larger float flips with it)
			for i in range(got.n):
				if got.sentence[i] != ref["sentence"][i]:
					continue
				a = int(off[got.sentence[i]])
				for mg, mr in zip(got.mapping[i], ref["mapping"][i]):
					assert (mg < 0) == (mr < 0) and (mg < 0 or ids[a + mg] == ids[a + mr]), (got.mapping[i], ref["mapping"][i])
		ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=100, sent_off=off, tok_id=ids, E=En, X_mag=emag[ids], Q=En[q_ids], q_ids=q_ids,
			Q_mag=emag[q_ids], algorithm=oracle.ALG_WRD, max_matches=10, min_score=0.0)
		got = c.query(E[q_ids], q_token_ids=q_ids, algorithm=hip.VK_ALG_WRD, q_normalize=True, max_matches=10, min_score=0.0)
		assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=1e-5)
	c.close()
	# span-embedding shape: one vector per slice, one query vector
	V = rng.standard_normal((3000, 96)).astype(np.float32)
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=96, n_tokens=3000, n_sentences=3000, precision="f32")
	c.append_vectors(V, normalize=True)
	c.set_sentences(np.arange(3001, dtype=np.int64))
	c.finalize()
	q = V[17:18] + 0.05 * rng.standard_normal((1, 96)).astype(np.float32)
	got = c.query(q, q_normalize=True, max_matches=5)
	cos = np.clip(oracle.normalize_rows(V) @ oracle.normalize_rows(q)[0], 0, 1)
	order = np.argsort(-cos, kind="stable")[:5]
	assert list(got.sentence[:5]) == list(order)
	np.testing.assert_allclose(got.score[:5], cos[order], atol=2e-6)
	c.close()
