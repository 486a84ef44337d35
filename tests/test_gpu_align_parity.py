"""-m gpu: the HIP path (through the C-ABI) against the CPU oracle on the same
seeded inputs.  Alignments with traceback: slice ids, scores and traceback mappings
bit-identical to the oracle's (helpers.assert_same_results, exact); the score vector of
all slices within 1e-4 (north_star tolerance)."""

import numpy as np
import pytest

from vectorian_amd import synth

from helpers import assert_same_results, hip_contextual_corpus, hip_static_corpus, prep_contextual, prep_query

pytestmark = pytest.mark.gpu

EXP5 = ("table", (1 - 2.0 ** (-np.arange(0, 65) / 5)).astype(np.float32))  # smooth_gap_cost(5)

GAPS = {
	"linear0.1": (0.1, 0.1),
	"linear0": (0.0, 0.0),
	"linear_asym": (0.25, 0.05),
	"affine": (("affine", 0.2, 0.05), ("affine", 0.1, 0.1)),
	"exp5": (EXP5, EXP5),
	"mixed": (0.1, EXP5),
}


@pytest.mark.parametrize("locality", [0, 1, 2])
@pytest.mark.parametrize("gap", list(GAPS))
@pytest.mark.parametrize("shape", ["fixed32_q10", "ragged_q5", "ragged64_q16", "tiny_q1", "ragged_q7_d1040", "ragged_q12_d272", "ragged_q9_d160"])
def test_contextual_parity(hip, oracle, locality, gap, shape):
	n, lo, hi, len_t, d = {
		"fixed32_q10": (600, 32, 32, 10, 300),
		"ragged_q5": (500, 1, 40, 5, 300),
		"ragged64_q16": (300, 8, 64, 16, 768),
		"tiny_q1": (37, 1, 3, 1, 50),
		# widths without a specialised kernel: the runtime K loop, eight K-steps deep from 256 features (32.5 / 8.5 K-steps: a
		# full eight, the four-step remainder, single steps and the half block), four deep below
		"ragged_q7_d1040": (250, 4, 64, 7, 1040),
		"ragged_q12_d272": (300, 1, 50, 12, 272),
		"ragged_q9_d160": (300, 1, 50, 9, 160),
	}[shape]
	corpus = synth.make_contextual_corpus(n, lo, hi, 2000, d)
	Xb = prep_contextual(corpus)
	c = hip_contextual_corpus(hip, corpus, Xb)
	gs, gt = GAPS[gap]
	for q in synth.make_queries(corpus, 2, len_t):
		Qb = prep_query(q)
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=corpus["sent_off"], X=Xb, Q=Qb,
			locality=locality, gap_s=gs, gap_t=gt, max_matches=10, min_score=0.0 if locality != 1 else -100.0)
		got = c.query(Qb, locality=locality, gap_s=gs, gap_t=gt, q_normalize=False, max_matches=10,
			min_score=0.0 if locality != 1 else -100.0).trimmed()
		assert_same_results(got, ref)
		np.testing.assert_allclose(got["raw_score"], ref["raw"], atol=2e-4, rtol=0)
	c.close()


@pytest.mark.parametrize("locality", [0, 1, 2])
@pytest.mark.parametrize("gap", ["linear0.1", "affine", "exp5"])
def test_static_parity(hip, oracle, locality, gap):
	corpus = synth.make_static_corpus(700, 1, 40, 3000, 300)
	c, Eb = hip_static_corpus(hip, corpus)
	gs, gt = GAPS[gap]
	for q in synth.make_queries(corpus, 2, 7):
		Qb = prep_query(q)
		ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=300, sent_off=corpus["sent_off"], tok_id=corpus["tok_id"], E=Eb,
			Q=Qb, q_ids=q["ids"], locality=locality, gap_s=gs, gap_t=gt, max_matches=20,
			min_score=0.0 if locality != 1 else -100.0)
		got = c.query(Qb, q_token_ids=q["ids"], locality=locality, gap_s=gs, gap_t=gt, q_normalize=False,
			max_matches=20, min_score=0.0 if locality != 1 else -100.0).trimmed()
		assert_same_results(got, ref)
	c.close()


@pytest.mark.parametrize("locality", [0, 1, 2])
@pytest.mark.parametrize("gap", ["linear0.1", "linear0", "linear_asym"])
def test_static_linear_gaps_repeated_words(hip, oracle, locality, gap):
	"""The case round 2 exempted: static layout, LINEAR gaps, words that repeat inside the slices and inside the query (a small
	vocabulary: equal similarity rows, co-optimal alignments that tie exactly in real arithmetic, the traceback then decided by
	the last bit of S).  The winners are restated in the oracle's arithmetic (sim_canon): ids, scores, tracebacks bit for bit,
	with the query's own token ids (sim[id(t_j)][j] = 1) and without them."""
	rng = np.random.default_rng(77 + locality)
	V, d, n = 12, 300, 800
	E = rng.standard_normal((V, d)).astype(np.float32)
	lens = rng.integers(1, 41, size=n)
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	ids = rng.integers(0, V, size=int(off[-1])).astype(np.int32)
	c = hip.Corpus(layout=hip.VK_LAYOUT_STATIC, d=d, n_tokens=len(ids), n_sentences=n, vocab_size=V)
	c.append_vectors(E, normalize=True)
	c.set_token_ids(ids)
	c.set_sentences(off)
	c.finalize()
	En = oracle.normalize_rows_bf16(E)[0]
	gs, gt = GAPS[gap]
	ms = 0.0 if locality != 1 else -1e9
	repeated = 0
	for len_t in (4, 9, 16):
		q_ids = rng.integers(0, V, size=len_t).astype(np.int32)
		q_ids[-1] = q_ids[0]   # the query repeats a word too
		for with_ids in (True, False):
			kw = dict(locality=locality, gap_s=gs, gap_t=gt, max_matches=25, min_score=ms)
			ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=d, sent_off=off, tok_id=ids, E=En, Q=En[q_ids], q_ids=q_ids if with_ids else None, **kw)
			got = c.query(E[q_ids], q_token_ids=q_ids if with_ids else None, q_normalize=True, **kw)
			assert_same_results(got.trimmed(), ref)
			for s in ref["sentence"]:
				repeated += len(set(ids[off[s]:off[s + 1]].tolist())) < int(off[s + 1] - off[s])
	assert repeated > 50   # the winners did hold repeated words
	c.close()


def test_all_scores_and_boost(hip, oracle):
	corpus = synth.make_contextual_corpus(1000, 4, 40, 2000, 300)
	Xb = prep_contextual(corpus)
	c = hip_contextual_corpus(hip, corpus, Xb)
	q = synth.make_queries(corpus, 1, 6)[0]
	Qb = prep_query(q)
	boost = np.random.default_rng(5).uniform(0.5, 1.5, size=1000).astype(np.float32)
	ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=300, sent_off=corpus["sent_off"], X=Xb, Q=Qb,
		gap_s=0.1, gap_t=0.1, max_matches=50, boost=boost, want_all_scores=True)
	got = c.query(Qb, gap_s=0.1, gap_t=0.1, q_normalize=False, max_matches=50, boost=boost)
	assert_same_results(got.trimmed(), ref)
	np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-4, rtol=0)
	c.close()


def test_errors_are_loud(hip):
	corpus = synth.make_contextual_corpus(10, 4, 8, 100, 32)
	c = hip_contextual_corpus(hip, corpus)
	q = np.ones((513, 32), dtype=np.float32)   # (65 .. 512 tokens run since round 4: vk_longq_kernel)
	with pytest.raises(hip.VkError):
		c.query(q)
	with pytest.raises(hip.VkError):
		c.query(q[:65], algorithm=hip.VK_ALG_RWMD)
	with pytest.raises(hip.VkError):
		c.query(q[:3], submatch_weight=-0.5)
	with pytest.raises(hip.VkError):
		c.query(q[:3], algorithm=hip.VK_ALG_WRD)
	c.close()


@pytest.mark.parametrize("len_t", [3, 10, 16, 20, 32, 45])
@pytest.mark.parametrize("table", ["linear_as_table", "convex", "steps"])
def test_general_gaps_that_are_not_subadditive(hip, oracle, len_t, table):
	"""gap tables for which two gaps in a row are cheaper than (or as cheap as) one of the summed length: the sequential
	recurrence composes them, the register-history kernels take the subadditive closure of the table (vk_query.cpp)"""
	k = np.arange(0, 513, dtype=np.float64)
	w = {"linear_as_table": 0.1 * k, "convex": np.minimum(0.02 * k ** 2, 1.5), "steps": 0.15 * np.ceil(k / 3.0) ** 1.5}[table].astype(np.float32)
	corpus = synth.make_contextual_corpus(600, 1, 40, 1500, 64)
	Xb = prep_contextual(corpus)
	c = hip_contextual_corpus(hip, corpus, Xb)
	for q in synth.make_queries(corpus, 2, len_t):
		Qb = prep_query(q)
		for loc, ms in ((0, 0.0), (1, -1e9), (2, -1e9)):
			kw = dict(locality=loc, gap_s=("table", w), gap_t=("table", w), max_matches=10, min_score=ms)
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=64, sent_off=corpus["sent_off"], X=Xb, Q=Qb, want_all_scores=True, **kw)
			got = c.query(Qb, q_normalize=False, **kw)
			assert_same_results(got.trimmed(), ref, score_tol=2e-5, tie_tol=2e-6)
			np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=2e-5, rtol=0)
	c.close()
