"""-m gpu: exact transport (Word Rotator's Distance, full Word Mover's Distance) with queries of 17..64 tokens: the bound pass of
the multi-block kernel (vk_score32_kernel, GAP 5) + the exact stage with 16 NQ supplies (vk_wrd_exact_kernel<NQ>) must return the
result set of the oracle, which solves every sentence exactly (vectorian/core/cpp/alignment/wrd.h:62-146, wmd.h:194-270)."""

import numpy as np
import pytest

from vectorian_amd import synth

from helpers import assert_same_results

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("opts", [
	("wrd", dict(wrd_normalize=True)), ("wrd", dict(wrd_normalize=False)),
	("wmd", dict(rwmd=(False, False, True), wmd_full=True)), ("wmd", dict(rwmd=(False, False, False), wmd_full=True)),
])
# (768-d rows: the four query tiles of a 49..64-token query take 96 KB of LDS -- the multi-block kernel then runs two waves or one
# wave per workgroup instead of four, vk_score32_waves; round 2 returned VK_ERR_UNSUPPORTED for these)
@pytest.mark.parametrize("shape", [(128, 32, 32, 20), (96, 1, 40, 17), (64, 8, 64, 32), (300, 3, 50, 40), (64, 2, 64, 64), (768, 8, 64, 64), (768, 4, 40, 50)])
def test_contextual_wide_transport(hip, oracle, shape, opts):
	d, lo, hi, len_t = shape
	alg, kw = opts
	n = 700
	corpus = synth.make_contextual_corpus(n, lo, hi, 2000, d, noise=0.3, norm_sigma=0.25)
	X = corpus["X"]
	Xb, mag = oracle.normalize_rows_bf16(X)
	off = corpus["sent_off"]
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=X.shape[0], n_sentences=n, keep_magnitudes=True)
	c.append_vectors(X, normalize=True)
	c.set_sentences(off)
	c.finalize()
	rng = np.random.default_rng(3)
	o_alg = oracle.ALG_WRD if alg == "wrd" else oracle.ALG_RWMD
	h_alg = hip.VK_ALG_WRD if alg == "wrd" else hip.VK_ALG_RWMD
	for q in synth.make_queries(corpus, 2, len_t):
		qv = (q["vectors"] * rng.lognormal(0, 0.25, size=(len_t, 1))).astype(np.float32)
		Qb, qmag = oracle.normalize_rows_bf16(qv)
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=off, X=Xb, X_mag=mag, Q=Qb, Q_mag=qmag,
			algorithm=o_alg, max_matches=8, min_score=0.0, n_threads=8, **kw)
		got = c.query(qv, algorithm=h_alg, q_normalize=True, max_matches=8, min_score=0.0, want_flow=True, **kw)
		assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=2e-5, tie_tol=2e-5)
		# the flows of the winners: similarity rows [64 x W] and the optimal plan [W x 64], W = len_t rounded up to 16
		W = (len_t + 15) // 16 * 16
		assert got.sim_rows.shape[1:] == (64, W) and got.plan.shape[1:] == (W, 64)
		for i in range(got.n):
			s = int(got.sentence[i])
			a, b = int(off[s]), int(off[s + 1])
			S = oracle.sim_bf16(Xb[a:b], Qb)
			assert (got.sim_rows[i, :b - a, :len_t] == S).all()   # the winners' rows are restated in the oracle's arithmetic (sim_canon): bit for bit
			G = got.plan[i, :len_t, :b - a].astype(np.float64)
			assert not got.plan[i, len_t:].any() and not got.plan[i, :, b - a:].any()
			raw = ((1.0 - np.maximum(1.0 - S.T, 0.0)) * G).sum() / G.sum()
			assert abs(raw - got.raw_score[i]) < 1e-5
	c.close()


@pytest.mark.parametrize("len_t", [19, 33])
def test_static_wide_transport(hip, oracle, len_t):
	V, d = 400, 64
	corpus = synth.make_static_corpus(800, 1, 40, V, d, seed=23)
	rng = np.random.default_rng(24)
	E = (corpus["E"] * rng.lognormal(0, 0.3, size=(V, 1))).astype(np.float32)
	Eb, emag = oracle.normalize_rows_bf16(E)
	off, ids = corpus["sent_off"], corpus["tok_id"]
	c = hip.Corpus(layout=hip.VK_LAYOUT_STATIC, d=d, n_tokens=len(ids), n_sentences=len(off) - 1, vocab_size=V, keep_magnitudes=True)
	c.append_vectors(E, normalize=True)
	c.set_token_ids(ids)
	c.set_sentences(off)
	c.finalize()
	for alg, kw in (("wrd", dict(wrd_normalize=True)), ("wrd", dict(wrd_normalize=False)), ("wmd", dict(rwmd=(False, False, True), wmd_full=True))):
		q_ids = rng.integers(0, 80, size=len_t).astype(np.int32)
		o_alg = oracle.ALG_WRD if alg == "wrd" else oracle.ALG_RWMD
		h_alg = hip.VK_ALG_WRD if alg == "wrd" else hip.VK_ALG_RWMD
		ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=d, sent_off=off, tok_id=ids, E=Eb, X_mag=emag[ids], Q=Eb[q_ids], q_ids=q_ids,
			Q_mag=emag[q_ids], algorithm=o_alg, max_matches=10, min_score=0.0, **kw)
		got = c.query(E[q_ids], q_token_ids=q_ids, algorithm=h_alg, q_normalize=True, max_matches=10, min_score=0.0, **kw)
		assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=2e-5, tie_tol=2e-5)
	c.close()


def test_wide_rows_wide_query_fill_rwmd_and_the_limit(hip, oracle):
	"""1:n RWMD (rwmd('nbow/distributed')) of a 64-token query over 768-d rows: no other kernel exists for this form, the
	multi-block kernel runs with fewer waves per workgroup.  Rows too wide for even one wave's strip beside the query tiles are
	refused by vk_validate_query (VK_ERR_UNSUPPORTED), before anything is enqueued."""
	d, len_t, n = 768, 64, 400
	corpus = synth.make_contextual_corpus(n, 8, 64, 1500, d, noise=0.3)
	X, off = corpus["X"], corpus["sent_off"]
	Xb, _ = oracle.normalize_rows_bf16(X)
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=X.shape[0], n_sentences=n, keep_magnitudes=True)
	c.append_vectors(X, normalize=True)
	c.set_sentences(off)
	c.finalize()
	for q in synth.make_queries(corpus, 2, len_t):
		Qb, _ = oracle.normalize_rows_bf16(q["vectors"])
		for flags in ((False, True, True), (False, False, True)):
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=off, X=Xb, Q=Qb, algorithm=oracle.ALG_RWMD, rwmd=flags,
				max_matches=8, min_score=-10.0, n_threads=8)
			got = c.query(q["vectors"], algorithm=hip.VK_ALG_RWMD, rwmd=flags, q_normalize=True, max_matches=8, min_score=-10.0)
			assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=2e-5, tie_tol=2e-5)
	c.close()
	# 2048-d fp32 rows: a query tile alone is 128 KB
	d2 = 2048
	rng = np.random.default_rng(2)
	X2 = rng.standard_normal((600, d2)).astype(np.float32)
	c2 = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d2, n_tokens=600, n_sentences=20, keep_magnitudes=True, precision="f32")
	c2.append_vectors(X2, normalize=True)
	c2.set_sentences(np.arange(21, dtype=np.int64) * 30)
	c2.finalize()
	qv = rng.standard_normal((40, d2)).astype(np.float32)
	with pytest.raises(hip.VkError) as e:
		c2.query(qv, algorithm=hip.VK_ALG_WRD, q_normalize=True, max_matches=5)
	assert e.value.status == hip.VK_ERR_UNSUPPORTED
	assert c2.query(qv[:12], algorithm=hip.VK_ALG_WRD, q_normalize=True, max_matches=5).n == 5   # at most 16 tokens: fine
	c2.close()

