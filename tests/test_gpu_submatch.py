"""-m gpu: submatch_weight != 0 (reference_score, vectorian/core/cpp/metric/alignment.h:84-106): the score divides by
a reference that depends on how many query tokens the optimal alignment matched.  The HIP path bounds the score
from the aligner score, retraces the candidates and stops when no remaining bound can enter the result set; the
result must be the oracle's (which retraces every slice)."""

import numpy as np
import pytest

from vectorian_amd import synth

from helpers import assert_same_results, hip_contextual_corpus, hip_static_corpus, prep_contextual, prep_query

pytestmark = pytest.mark.gpu

EXP5 = ("table", (1 - 2.0 ** (-np.arange(0, 65) / 5)).astype(np.float32))
AFF = ("affine", 0.2, 0.05)


@pytest.mark.parametrize("w", [0.5, 1.0, 3.0])
@pytest.mark.parametrize("len_t", [6, 12, 24])
def test_submatch_weight_contextual(hip, oracle, w, len_t):
	corpus = synth.make_contextual_corpus(3000, 2, 40, 1200, 64)
	Xb = prep_contextual(corpus)
	c = hip_contextual_corpus(hip, corpus, Xb)
	boost = np.random.default_rng(2).uniform(0.7, 1.3, size=3000).astype(np.float32)
	for qi, q in enumerate(synth.make_queries(corpus, 2, len_t)):
		Qb = prep_query(q)
		for loc, ms, gaps, bst in ((0, 0.0, (0.1, 0.1), None), (0, 0.05, (EXP5, EXP5), boost), (2, 0.0, (AFF, AFF), None), (1, -1e9, (0.05, 0.05), None)):
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=64, sent_off=corpus["sent_off"], X=Xb, Q=Qb, locality=loc,
				gap_s=gaps[0], gap_t=gaps[1], max_matches=10, min_score=ms, boost=bst, submatch_weight=w)
			got = c.query(Qb, q_normalize=False, locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=10, min_score=ms, boost=bst,
				submatch_weight=w)
			assert_same_results(got.trimmed(), ref)
	with pytest.raises(hip.VkError):
		c.query(Qb, q_normalize=False, submatch_weight=-1.0)
	c.close()


def test_submatch_weight_static_tag_weighted(hip, oracle):
	corpus = synth.make_static_corpus(2000, 3, 30, 600, 50, seed=3)
	rng = np.random.default_rng(4)
	pos = rng.integers(0, 6, size=len(corpus["tok_id"])).astype(np.int8)
	c, Eb = hip_static_corpus(hip, corpus)
	c.set_token_pos(pos)
	off = corpus["sent_off"]
	for s in (10, 500):
		q_ids = corpus["tok_id"][off[s]:off[s] + 7].astype(np.int32)
		Qb = Eb[q_ids]
		tw = rng.uniform(0.3, 1.0, size=len(q_ids)).astype(np.float32)
		qp = pos[off[s]:off[s] + len(q_ids)].copy()
		qp[1] = (qp[1] + 1) % 6
		for w in (1.0, 2.0):
			kw = dict(locality=0, gap_s=0.1, gap_t=0.1, max_matches=8, min_score=0.0, submatch_weight=w, tag_weights=tw, q_pos=qp,
				pos_mismatch_penalty=0.3, similarity_threshold=0.1)
			ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=50, sent_off=off, tok_id=corpus["tok_id"], E=Eb, Q=Qb, q_ids=q_ids, pos_s=pos, **kw)
			got = c.query(Qb, q_token_ids=q_ids, q_normalize=False, **kw)
			assert_same_results(got.trimmed(), ref)
	# transport metrics: every query token is matched, the reference is len_t whatever the weight (match.h:165-176)
	ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=50, sent_off=off, tok_id=corpus["tok_id"], E=Eb, Q=Qb, q_ids=q_ids,
		algorithm=oracle.ALG_RWMD, max_matches=8, submatch_weight=1.0)
	got = c.query(Qb, q_token_ids=q_ids, q_normalize=False, algorithm=hip.VK_ALG_RWMD, max_matches=8, submatch_weight=1.0)
	assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=2e-5, tie_tol=2e-5)
	c.close()
