"""-m gpu: relaxed Word Mover's Distance on the HIP path (through the C-ABI) against the oracle's
restatement of the reference's BOW + RelaxedSolver code (alignment/bow.h, alignment/wmd.h)."""

import numpy as np
import pytest

from vectorian_amd import synth

from helpers import assert_same_results, hip_contextual_corpus, hip_static_corpus, prep_contextual, prep_query

pytestmark = pytest.mark.gpu

VARIANTS = {
	"nbow": (True, True, True),        # rwmd('nbow'): injective, symmetric, normalised (vectorian/alignment.py:232-233)
	"bow/fast": (True, False, False),  # rwmd('bow/fast') (:236-237)
	"nbow-onesided": (True, False, True),
	# 1:n forms (RelaxedSolver with injective = false, alignment/wmd.h:339-376)
	"nbow/distributed": (False, True, True),   # rwmd('nbow/distributed') (:234-235)
	"distributed-onesided": (False, False, True),
	"distributed-bow": (False, False, False),
}


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("shape", ["fixed32_q10", "ragged_q5", "ragged64_q16"])
def test_contextual_rwmd(hip, oracle, variant, shape):
	n, lo, hi, len_t, d = {"fixed32_q10": (600, 32, 32, 10, 300), "ragged_q5": (500, 1, 40, 5, 300),
		"ragged64_q16": (300, 8, 64, 16, 768)}[shape]
	corpus = synth.make_contextual_corpus(n, lo, hi, 2000, d)
	Xb = prep_contextual(corpus)
	c = hip_contextual_corpus(hip, corpus, Xb)
	flags = VARIANTS[variant]
	for q in synth.make_queries(corpus, 2, len_t):
		Qb = prep_query(q)
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=corpus["sent_off"], X=Xb, Q=Qb,
			algorithm=oracle.ALG_RWMD, rwmd=flags, max_matches=15, min_score=0.0, want_all_scores=True)
		got = c.query(Qb, algorithm=hip.VK_ALG_RWMD, rwmd=flags, q_normalize=False, max_matches=15, min_score=0.0)
		assert_same_results(got.trimmed(), ref, check_mapping=False, exact=True)   # relaxed WMD: the winners are restated on the host from canonical rows (vk_transport_host.h)
		np.testing.assert_allclose(got.raw_score[:got.n], ref["raw"], atol=1e-4, rtol=0)
		np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-5, rtol=0)
	c.close()


@pytest.mark.parametrize("variant", ["nbow", "bow/fast", "nbow/distributed", "distributed-bow"])
def test_static_rwmd(hip, oracle, variant):
	# query vectors are the vocabulary's own vectors (the regime of Index.find); repeated tokens in
	# sentences and tokens shared between query and sentence exercise the joint-vocabulary BOW builder
	corpus = synth.make_static_corpus(700, 1, 40, 300, 300)
	c, Eb = hip_static_corpus(hip, corpus)
	E = synth.bf16_bits_to_f32(Eb)
	rng = np.random.default_rng(11)
	flags = VARIANTS[variant]
	for _ in range(3):
		qids = rng.integers(0, 40, size=6).astype(np.int32)      # frequent ids: overlap with sentences is common
		qids[4] = qids[1]                                         # a repeated query token: one vocabulary entry of mass 2
		Qb = Eb[qids]
		ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=300, sent_off=corpus["sent_off"], tok_id=corpus["tok_id"], E=Eb,
			Q=Qb, q_ids=qids, algorithm=oracle.ALG_RWMD, rwmd=flags, max_matches=20, want_all_scores=True)
		got = c.query(Qb, q_token_ids=qids, algorithm=hip.VK_ALG_RWMD, rwmd=flags, q_normalize=False, max_matches=20)
		np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-5, rtol=0)
		assert_same_results(got.trimmed(), ref, check_mapping=False, exact=True)
	c.close()


@pytest.mark.parametrize("layout", ["contextual", "static"])
def test_distributed_rwmd_long_slices(hip, oracle, layout):
	# one query token must be spread over every token of the slice; slices of more than 64 tokens take the second launch
	# (17 .. 64 query tokens: the multi-block kernel for the short slices, vk_long_rwmd_fill_kernel for the long ones)
	lens = np.array([3, 70, 12, 1, 33, 200, 64, 5, 9, 130])
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	rng = np.random.default_rng(5)
	if layout == "static":
		V = 60     # few words: repeated token ids inside the long slices (vocabulary entries with counts)
		Eb = synth.to_bf16_bits(synth.normalize_rows(rng.standard_normal((V, 64)).astype(np.float32)))
		ids = rng.integers(0, V, size=int(off[-1])).astype(np.int32)
		c = hip.Corpus(layout=hip.VK_LAYOUT_STATIC, d=64, n_tokens=len(ids), n_sentences=len(lens), vocab_size=V)
		c.append_vectors(Eb, normalize=False)
		c.set_token_ids(ids)
	else:
		Xb = synth.to_bf16_bits(synth.normalize_rows(rng.standard_normal((int(off[-1]), 64)).astype(np.float32)))
		c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=64, n_tokens=Xb.shape[0], n_sentences=len(lens))
		c.append_vectors(Xb, normalize=False)
	c.set_sentences(off)
	c.finalize()
	for len_t in (1, 2, 16, 17, 40, 64):
		if layout == "static":
			qids = rng.integers(0, V, size=len_t).astype(np.int32)
			Qb = Eb[qids]
			base = dict(layout=oracle.LAYOUT_STATIC, d=64, sent_off=off, tok_id=ids, E=Eb, Q=Qb, q_ids=qids)
			qargs = dict(q_token_ids=qids)
		else:
			Qb = synth.to_bf16_bits(synth.normalize_rows(rng.standard_normal((len_t, 64)).astype(np.float32)))
			base = dict(layout=oracle.LAYOUT_CONTEXTUAL, d=64, sent_off=off, X=Xb, Q=Qb)
			qargs = {}
		for flags in ((False, True, True), (False, False, True), (False, False, False)):
			ref = oracle.find(algorithm=oracle.ALG_RWMD, rwmd=flags, max_matches=9, min_score=-1.0, want_all_scores=True, **base)
			got = c.query(Qb, algorithm=hip.VK_ALG_RWMD, rwmd=flags, q_normalize=False, max_matches=9, min_score=-1.0, **qargs)
			np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-5, rtol=0)
			assert_same_results(got.trimmed(), ref, check_mapping=False, exact=True)
	c.close()


@pytest.mark.parametrize("layout", ["contextual", "static"])
@pytest.mark.parametrize("len_t", [17, 24, 32, 33, 50, 64])
def test_distributed_rwmd_long_queries(hip, oracle, layout, len_t):
	"""the 1:n form (rwmd('nbow/distributed'), wmd.h:339-376) with queries of 17..64 tokens: rwmd_fill32 of the multi-block kernel"""
	rng = np.random.default_rng(7 + len_t)
	if layout == "static":
		V, d = 60, 64                                   # a small vocabulary: words repeat inside slices and queries
		corpus = synth.make_static_corpus(300, 1, 64, V, d, seed=5)
		c, Eb = hip_static_corpus(hip, corpus)
		q_ids = rng.integers(0, V, size=len_t).astype(np.int32)
		base = dict(layout=oracle.LAYOUT_STATIC, d=d, sent_off=corpus["sent_off"], tok_id=corpus["tok_id"], E=Eb, Q=Eb[q_ids], q_ids=q_ids)
		qargs = dict(q_token_ids=q_ids)
		Qb = Eb[q_ids]
	else:
		corpus = synth.make_contextual_corpus(300, 1, 64, 500, 96)
		Xb = prep_contextual(corpus)
		c = hip_contextual_corpus(hip, corpus, Xb)
		Qb = prep_query(synth.make_queries(corpus, 1, len_t)[0])
		base = dict(layout=oracle.LAYOUT_CONTEXTUAL, d=96, sent_off=corpus["sent_off"], X=Xb, Q=Qb)
		qargs = {}
	for flags in ((False, True, True), (False, False, True), (False, False, False)):
		ref = oracle.find(algorithm=oracle.ALG_RWMD, rwmd=flags, max_matches=9, min_score=-1.0, want_all_scores=True, **base)
		got = c.query(Qb, algorithm=hip.VK_ALG_RWMD, rwmd=flags, q_normalize=False, max_matches=9, min_score=-1.0, **qargs)
		np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-5, rtol=0)
		assert_same_results(got.trimmed(), ref, check_mapping=False, exact=True)
	c.close()


def test_symmetric_bow_is_rejected_like_upstream(hip):
	# WMD::operator() throws "cannot run symmetric mode WMD with bow (needs nbow)" (alignment/wmd.h:441-449)
	corpus = synth.make_contextual_corpus(10, 4, 8, 100, 32)
	c = hip_contextual_corpus(hip, corpus)
	with pytest.raises(hip.VkError):
		c.query(np.ones((3, 32), np.float32), algorithm=hip.VK_ALG_RWMD, rwmd=(True, True, False))
	c.close()


@pytest.mark.parametrize("variant", ["wmd/nbow", "wmd/bow"])
@pytest.mark.parametrize("shape", ["fixed32_q10", "ragged_q5", "long_query"])
def test_full_wmd(hip, oracle, variant, shape):
	# WordMoversDistance.wmd (vectorian/alignment.py:206-218; labels as upstream):
	# 'nbow' -> normalize_bow False (unit masses: an assignment problem), 'bow' -> normalize_bow True
	n, lo, hi, len_t, d = {"fixed32_q10": (800, 32, 32, 10, 300), "ragged_q5": (600, 1, 40, 5, 128),
		"long_query": (300, 2, 12, 16, 64)}[shape]
	nbow = variant == "wmd/bow"
	corpus = synth.make_contextual_corpus(n, lo, hi, 2000, d)
	Xb = prep_contextual(corpus)
	c = hip_contextual_corpus(hip, corpus, Xb)
	for q in synth.make_queries(corpus, 2, len_t):
		Qb = prep_query(q)
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=corpus["sent_off"], X=Xb, Q=Qb, algorithm=oracle.ALG_RWMD,
			rwmd=(False, False, nbow), wmd_full=True, max_matches=10, min_score=0.0, n_threads=8)
		got = c.query(Qb, algorithm=hip.VK_ALG_RWMD, rwmd=(False, False, nbow), wmd_full=True, q_normalize=False, max_matches=10, min_score=0.0)
		assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=1e-5, tie_tol=1e-5)
	c.close()


def test_full_wmd_static_layout(hip, oracle):
	corpus = synth.make_static_corpus(500, 1, 30, 300, 300)
	c, Eb = hip_static_corpus(hip, corpus)
	rng = np.random.default_rng(5)
	for _ in range(2):
		qids = rng.integers(0, 40, size=6).astype(np.int32)
		Qb = Eb[qids]
		ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=300, sent_off=corpus["sent_off"], tok_id=corpus["tok_id"], E=Eb, Q=Qb, q_ids=qids,
			algorithm=oracle.ALG_RWMD, rwmd=(False, False, False), wmd_full=True, max_matches=12)
		got = c.query(Qb, q_token_ids=qids, algorithm=hip.VK_ALG_RWMD, rwmd=(False, False, False), wmd_full=True, q_normalize=False, max_matches=12)
		assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=1e-5, tie_tol=1e-5)
	c.close()
