"""-m gpu: 'alignment-tag-weighted' on the HIP path against the oracle's restatement of
TagWeightedSlice (vectorian/core/cpp/slice/static.h:186-288): modified similarities drive the DP,
scores are divided by sum(tag_weights), reported edge similarities stay unmodified."""

import numpy as np
import pytest

from vectorian_amd import synth

from helpers import assert_same_results, hip_contextual_corpus, hip_static_corpus, prep_contextual, prep_query

pytestmark = pytest.mark.gpu

EXP5 = ("table", (1 - 2.0 ** (-np.arange(0, 65) / 5)).astype(np.float32))


@pytest.mark.parametrize("d,lo,hi,len_t", [(300, 32, 32, 10), (300, 1, 40, 5), (768, 8, 64, 16), (64, 2, 20, 3), (300, 1, 40, 20), (64, 2, 64, 32), (96, 2, 50, 45)])
@pytest.mark.parametrize("gap", [(0.1, 0.1), (EXP5, EXP5), (("affine", 0.2, 0.05), 0.1)])
def test_contextual_tag_weighted(hip, oracle, d, lo, hi, len_t, gap):
	n = 400
	corpus = synth.make_contextual_corpus(n, lo, hi, 1500, d)
	Xb = prep_contextual(corpus)
	rng = np.random.default_rng(21)
	pos_s = rng.integers(1, 5, size=Xb.shape[0]).astype(np.int8)
	c = hip_contextual_corpus(hip, corpus, Xb)
	c.set_token_pos(pos_s)
	for qi, q in enumerate(synth.make_queries(corpus, 2, len_t)):
		Qb = prep_query(q)
		tw = rng.choice([0.5, 1.0, 2.0], size=len_t).astype(np.float32)
		q_pos = rng.integers(1, 5, size=len_t).astype(np.int8)
		kw = dict(tag_weights=tw, q_pos=q_pos, pos_mismatch_penalty=0.4, similarity_threshold=0.15)
		for loc, ms in ((0, 0.0), (1, -100.0)):
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=corpus["sent_off"], X=Xb, Q=Qb, pos_s=pos_s,
				locality=loc, gap_s=gap[0], gap_t=gap[1], max_matches=10, min_score=ms, want_all_scores=True, **kw)
			got = c.query(Qb, q_normalize=False, locality=loc, gap_s=gap[0], gap_t=gap[1], max_matches=10, min_score=ms, **kw)
			assert_same_results(got.trimmed(), ref)
			np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-4, rtol=0)
			# edges report the unmodified similarity
			S_top = oracle.sim_bf16(Xb[corpus["sent_off"][ref["sentence"][0]]:corpus["sent_off"][ref["sentence"][0] + 1]], Qb)
			for j in range(len_t):
				if got.mapping[0][j] >= 0:
					assert abs(got.edge_sim[0][j] - S_top[got.mapping[0][j], j]) < 1e-5
	c.close()


@pytest.mark.parametrize("len_t", [7, 24, 40])
def test_static_tag_weighted(hip, oracle, len_t):
	corpus = synth.make_static_corpus(500, 1, 40, 2000, 300)
	c, Eb = hip_static_corpus(hip, corpus)
	rng = np.random.default_rng(22)
	pos_s = rng.integers(1, 4, size=len(corpus["tok_id"])).astype(np.int8)
	c.set_token_pos(pos_s)
	for q in synth.make_queries(corpus, 2, len_t):
		Qb = prep_query(q)
		tw = rng.choice([0.5, 1.0, 3.0], size=len_t).astype(np.float32)
		q_pos = rng.integers(1, 4, size=len_t).astype(np.int8)
		kw = dict(tag_weights=tw, q_pos=q_pos, pos_mismatch_penalty=0.25, similarity_threshold=0.05)
		ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=300, sent_off=corpus["sent_off"], tok_id=corpus["tok_id"], E=Eb, Q=Qb,
			q_ids=q["ids"], pos_s=pos_s, gap_s=EXP5, gap_t=EXP5, max_matches=15, **kw)
		got = c.query(Qb, q_token_ids=q["ids"], q_normalize=False, gap_s=EXP5, gap_t=EXP5, max_matches=15, **kw).trimmed()
		assert_same_results(got, ref)
	c.close()


def test_tag_weighted_needs_pos_and_alignment(hip):
	corpus = synth.make_contextual_corpus(10, 4, 8, 100, 32)
	c = hip_contextual_corpus(hip, corpus)
	q = np.ones((3, 32), np.float32)
	with pytest.raises(hip.VkError):
		c.query(q, tag_weights=[1, 1, 1], q_pos=[1, 1, 1])           # no POS uploaded
	c.set_token_pos(np.ones(corpus["X"].shape[0], np.int8))
	with pytest.raises(hip.VkError):
		c.query(q, algorithm=hip.VK_ALG_RWMD, tag_weights=[1, 1, 1], q_pos=[1, 1, 1])
	c.query(q, tag_weights=[1, 1, 1], q_pos=[1, 1, 1])
	c.close()
