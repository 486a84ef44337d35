"""-m gpu: Word Rotator's Distance on the HIP path: bound pass + exact EMD of the survivors must
return exactly the result set of the oracle, which solves every sentence exactly."""

import numpy as np
import pytest

from vectorian_amd import synth

from helpers import assert_same_results

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("normalize", [True, False])
@pytest.mark.parametrize("shape", ["q10_len32", "q5_ragged", "q16_ragged64"])
def test_contextual_wrd(hip, oracle, shape, normalize):
	n, lo, hi, len_t, d = {"q10_len32": (1500, 32, 32, 10, 128), "q5_ragged": (1200, 1, 40, 5, 96),
		"q16_ragged64": (600, 8, 64, 16, 768)}[shape]
	corpus = synth.make_contextual_corpus(n, lo, hi, 2000, d, noise=0.3, norm_sigma=0.25)
	X = corpus["X"]
	Xb, mag = oracle.normalize_rows_bf16(X)
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=X.shape[0], n_sentences=n, keep_magnitudes=True)
	c.append_vectors(X, normalize=True)
	c.set_sentences(corpus["sent_off"])
	c.finalize()
	rng = np.random.default_rng(2)
	for q in synth.make_queries(corpus, 2, len_t):
		qv = (q["vectors"] * rng.lognormal(0, 0.25, size=(len_t, 1))).astype(np.float32)
		Qb, qmag = oracle.normalize_rows_bf16(qv)
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=corpus["sent_off"], X=Xb, X_mag=mag, Q=Qb, Q_mag=qmag,
			algorithm=oracle.ALG_WRD, max_matches=10, min_score=0.0, n_threads=8, wrd_normalize=normalize)
		got = c.query(qv, algorithm=hip.VK_ALG_WRD, q_normalize=True, max_matches=10, min_score=0.0, wrd_normalize=normalize)
		assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=1e-5)
	c.close()


def test_static_layout_wrd(hip, oracle):
	# static embeddings carry magnitudes too (metric/static.cpp:69-73, 80-120): mass of a token = magnitude of its vocabulary vector
	corpus = synth.make_static_corpus(900, 1, 40, 400, 64, seed=21)
	rng = np.random.default_rng(22)
	E = (corpus["E"] * rng.lognormal(0, 0.3, size=(400, 1))).astype(np.float32)
	Eb, emag = oracle.normalize_rows_bf16(E)
	off, ids = corpus["sent_off"], corpus["tok_id"]
	c = hip.Corpus(layout=hip.VK_LAYOUT_STATIC, d=64, n_tokens=len(ids), n_sentences=len(off) - 1, vocab_size=400, keep_magnitudes=True)
	c.append_vectors(E, normalize=True)
	c.set_token_ids(ids)
	c.set_sentences(off)
	c.finalize()
	for normalize in (True, False):
		for _ in range(2):
			q_ids = rng.integers(0, 60, size=7).astype(np.int32)
			ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=64, sent_off=off, tok_id=ids, E=Eb, X_mag=emag[ids], Q=Eb[q_ids], q_ids=q_ids,
				Q_mag=emag[q_ids], algorithm=oracle.ALG_WRD, max_matches=10, min_score=0.0, wrd_normalize=normalize)
			got = c.query(E[q_ids], q_token_ids=q_ids, algorithm=hip.VK_ALG_WRD, q_normalize=True, max_matches=10, min_score=0.0, wrd_normalize=normalize)
			assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=1e-5)
	c.close()


def test_wrd_needs_magnitudes(hip):
	corpus = synth.make_contextual_corpus(10, 4, 8, 100, 32)
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=32, n_tokens=corpus["X"].shape[0], n_sentences=10)
	c.append_vectors(corpus["X"], normalize=True)
	c.set_sentences(corpus["sent_off"])
	c.finalize()
	with pytest.raises(hip.VkError):
		c.query(np.ones((3, 32), np.float32), algorithm=hip.VK_ALG_WRD)
	c.close()


def test_transport_flow_outputs(hip, oracle):
	"""want_flow on transport metrics: similarity rows of the winners, and for exact transport the optimal plan
	(its marginals are the masses, its value the score)"""
	corpus = synth.make_contextual_corpus(400, 3, 50, 800, 96, noise=0.3, norm_sigma=0.25)
	X = corpus["X"]
	Xb, mag = oracle.normalize_rows_bf16(X)
	off = corpus["sent_off"]
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=96, n_tokens=X.shape[0], n_sentences=400, keep_magnitudes=True)
	c.append_vectors(X, normalize=True)
	c.set_sentences(off)
	c.finalize()
	qv = synth.make_queries(corpus, 1, 7)[0]["vectors"]
	Qb, qmag = oracle.normalize_rows_bf16(qv)
	for alg, kw in ((hip.VK_ALG_WRD, {}), (hip.VK_ALG_RWMD, dict(rwmd=(False, False, True), wmd_full=True)), (hip.VK_ALG_RWMD, {})):
		got = c.query(qv, algorithm=alg, q_normalize=True, max_matches=6, min_score=0.0, want_flow=True, **kw)
		assert got.n == 6 and got.sim_rows is not None
		for i in range(got.n):
			s = int(got.sentence[i])
			a, b = int(off[s]), int(off[s + 1])
			S = oracle.sim_bf16(Xb[a:b], Qb)
			np.testing.assert_allclose(got.sim_rows[i, :b - a, :7], S, atol=2e-6)
			assert not got.sim_rows[i, b - a:].any()
			if alg == hip.VK_ALG_RWMD and not kw:
				continue
			G = got.plan[i, :7, :b - a].astype(np.float64)
			if alg == hip.VK_ALG_WRD:
				mt, ms = qmag / qmag.sum(), mag[a:b] / mag[a:b].sum()
			else:
				mt, ms = np.full(7, 1 / 7), np.full(b - a, 1 / (b - a))
			np.testing.assert_allclose(G.sum(axis=1), mt, atol=1e-6)
			np.testing.assert_allclose(G.sum(axis=0), ms, atol=1e-6)
			raw = ((1.0 - np.maximum(1.0 - S.T, 0.0)) * G).sum() / G.sum()
			assert abs(raw - got.raw_score[i]) < 1e-5
	c.close()
