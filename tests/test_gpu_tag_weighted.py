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


def test_tag_weighted_needs_pos(hip):
	corpus = synth.make_contextual_corpus(10, 4, 8, 100, 32)
	c = hip_contextual_corpus(hip, corpus)
	q = np.ones((3, 32), np.float32)
	with pytest.raises(hip.VkError):
		c.query(q, tag_weights=[1, 1, 1], q_pos=[1, 1, 1])           # no POS uploaded
	c.set_token_pos(np.ones(corpus["X"].shape[0], np.int8))
	c.query(q, algorithm=hip.VK_ALG_RWMD, tag_weights=[1, 1, 1], q_pos=[1, 1, 1])   # any matcher takes the modifier
	c.query(q, tag_weights=[1, 1, 1], q_pos=[1, 1, 1])
	c.close()


TRANSPORTS = [
	("rwmd", dict(rwmd=(True, True, True))), ("rwmd", dict(rwmd=(True, False, False))),
	("rwmd", dict(rwmd=(False, True, True))), ("rwmd", dict(rwmd=(False, False, False))),      # 1:n form
	("wmd", dict(rwmd=(False, False, True), wmd_full=True)), ("wmd", dict(rwmd=(False, False, False), wmd_full=True)),
	("wrd", dict(wrd_normalize=True)), ("wrd", dict(wrd_normalize=False)),
]


@pytest.mark.parametrize("shape", [(96, 1, 40, 6), (300, 32, 32, 10), (64, 2, 64, 16), (96, 3, 50, 23), (64, 2, 64, 50)])
@pytest.mark.parametrize("alg,opts", TRANSPORTS)
def test_contextual_tag_weighted_transport(hip, oracle, shape, alg, opts):
	"""the tag-weighted modifier with the transport metrics (TagWeightedSlice wraps any slice, match/instantiate.cpp:173-189):
	modified similarities drive the solver, the score is divided by sum(tag_weights) (slice/static.h:280-286)"""
	d, lo, hi, len_t = shape
	n = 300
	corpus = synth.make_contextual_corpus(n, lo, hi, 1500, d, noise=0.3, norm_sigma=0.25)
	X = corpus["X"]
	Xb, mag = oracle.normalize_rows_bf16(X)
	rng = np.random.default_rng(31)
	pos_s = rng.integers(1, 5, size=X.shape[0]).astype(np.int8)
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=X.shape[0], n_sentences=n, keep_magnitudes=True)
	c.append_vectors(X, normalize=True)
	c.set_sentences(corpus["sent_off"])
	c.set_token_pos(pos_s)
	c.finalize()
	for q in synth.make_queries(corpus, 2, len_t):
		qv = (q["vectors"] * rng.lognormal(0, 0.25, size=(len_t, 1))).astype(np.float32)
		Qb, qmag = oracle.normalize_rows_bf16(qv)
		tw = rng.choice([0.5, 1.0, 2.0], size=len_t).astype(np.float32)
		q_pos = rng.integers(1, 5, size=len_t).astype(np.int8)
		kw = dict(tag_weights=tw, q_pos=q_pos, pos_mismatch_penalty=0.4, similarity_threshold=0.15, max_matches=10, min_score=0.0, **opts)
		o_alg = oracle.ALG_WRD if alg == "wrd" else oracle.ALG_RWMD
		h_alg = hip.VK_ALG_WRD if alg == "wrd" else hip.VK_ALG_RWMD
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=corpus["sent_off"], X=Xb, X_mag=mag, Q=Qb, Q_mag=qmag, pos_s=pos_s,
			algorithm=o_alg, n_threads=8, **kw)
		got = c.query(qv, q_normalize=True, algorithm=h_alg, **kw)
		assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=2e-5, tie_tol=2e-5, exact=(alg == "rwmd"))   # relaxed forms: the oracle's floats
	c.close()


@pytest.mark.parametrize("len_t", [7, 20, 40])
@pytest.mark.parametrize("alg,opts", TRANSPORTS)
def test_static_tag_weighted_transport(hip, oracle, len_t, alg, opts):
	"""static layout: the bags of words are keyed by (token id, tag) (TaggedTokenFactory, alignment/bow.h:150-176); the universal POS
	is a function of the tag, as in spaCy's tag map, so equal keys have equal similarity rows"""
	V, d = 300, 64
	corpus = synth.make_static_corpus(400, 1, 40, V, d, seed=41)
	rng = np.random.default_rng(42)
	E = (corpus["E"] * rng.lognormal(0, 0.3, size=(V, 1))).astype(np.float32)
	Eb, emag = oracle.normalize_rows_bf16(E)
	off, ids = corpus["sent_off"], corpus["tok_id"]
	tag_s = rng.integers(1, 9, size=len(ids)).astype(np.int8)
	pos_s = (tag_s % 3 + 1).astype(np.int8)
	c = hip.Corpus(layout=hip.VK_LAYOUT_STATIC, d=d, n_tokens=len(ids), n_sentences=len(off) - 1, vocab_size=V, keep_magnitudes=True)
	c.append_vectors(E, normalize=True)
	c.set_token_ids(ids)
	c.set_sentences(off)
	c.set_token_pos(pos_s)
	c.set_token_tags(tag_s)
	c.finalize()
	for _ in range(2):
		q_ids = rng.integers(0, 40, size=len_t).astype(np.int32)
		q_tag = rng.integers(1, 9, size=len_t).astype(np.int8)
		q_pos = (q_tag % 3 + 1).astype(np.int8)
		tw = np.array([0.5, 1.0, 3.0, 1.5, 0.75, 2.0, 1.0, 0.25, 1.25], dtype=np.float32)[q_tag]   # weights go by tag (parse_tag_weights, instantiate.cpp:10-38)
		kw = dict(tag_weights=tw, q_pos=q_pos, pos_mismatch_penalty=0.25, similarity_threshold=0.05, max_matches=10, min_score=0.0, **opts)
		o_alg = oracle.ALG_WRD if alg == "wrd" else oracle.ALG_RWMD
		h_alg = hip.VK_ALG_WRD if alg == "wrd" else hip.VK_ALG_RWMD
		ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=d, sent_off=off, tok_id=ids, E=Eb, X_mag=emag[ids], Q=Eb[q_ids], q_ids=q_ids, Q_mag=emag[q_ids],
			pos_s=pos_s, tag_s=tag_s, q_tag=q_tag, algorithm=o_alg, **kw)
		# q_tags: the (id, tag) keys on the device -- the masses of the 1:n form, and for every vocabulary transport the cells upstream's
		# distance matrix writes twice (static_vocab_fixup)
		got = c.query(E[q_ids], q_token_ids=q_ids, q_normalize=True, algorithm=h_alg, q_tags=q_tag, **kw)
		assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=2e-5, tie_tol=2e-5, exact=(alg == "rwmd"))
	c.close()


def test_static_tag_weighted_transport_shared_entries(hip, oracle):
	"""Upstream fills its vocabulary distance matrix with dist(u, v) = dist(v, u) = d in loop order (alignment/wmd.h:121-133; SURVEY
	B9): when TWO vocabulary entries occur in both the slice and the query, their cell is written twice, and with the tag-weighted
	(asymmetric) similarity the two values differ -- the later write wins.  The oracle restates that.  With the tags of the query
	tokens (q_tags) and of the corpus the device reproduces it (static_vocab_fixup): every slice exact, for a 40-token query over 40
	word ids, where 5 % of the slices have such cells.  Without q_tags the device takes every distance from the positions' own
	similarity: those slices then differ, by less than 5e-4 in the score -- measured here so that it cannot grow unnoticed."""
	V, d, len_t = 300, 64, 40
	corpus = synth.make_static_corpus(400, 1, 40, V, d, seed=41)
	rng = np.random.default_rng(42)
	E = (corpus["E"] * rng.lognormal(0, 0.3, size=(V, 1))).astype(np.float32)
	Eb, emag = oracle.normalize_rows_bf16(E)
	off, ids = corpus["sent_off"], corpus["tok_id"]
	tag_s = rng.integers(1, 9, size=len(ids)).astype(np.int8)
	pos_s = (tag_s % 3 + 1).astype(np.int8)
	c = hip.Corpus(layout=hip.VK_LAYOUT_STATIC, d=d, n_tokens=len(ids), n_sentences=len(off) - 1, vocab_size=V, keep_magnitudes=True)
	c.append_vectors(E, normalize=True)
	c.set_token_ids(ids)
	c.set_sentences(off)
	c.set_token_pos(pos_s)
	c.set_token_tags(tag_s)
	c.finalize()
	worst, affected, total = 0.0, 0, 0
	for _ in range(2):
		q_ids = rng.integers(0, 40, size=len_t).astype(np.int32)
		q_tag = rng.integers(1, 9, size=len_t).astype(np.int8)
		q_pos = (q_tag % 3 + 1).astype(np.int8)
		tw = np.array([0.5, 1.0, 3.0, 1.5, 0.75, 2.0, 1.0, 0.25, 1.25], dtype=np.float32)[q_tag]
		for flags in ((True, True, True), (True, False, True)):
			kw = dict(tag_weights=tw, q_pos=q_pos, pos_mismatch_penalty=0.25, similarity_threshold=0.05, max_matches=10, min_score=-1.0, rwmd=flags)
			ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=d, sent_off=off, tok_id=ids, E=Eb, X_mag=emag[ids], Q=Eb[q_ids], q_ids=q_ids, Q_mag=emag[q_ids],
				pos_s=pos_s, tag_s=tag_s, q_tag=q_tag, algorithm=oracle.ALG_RWMD, want_all_scores=True, **kw)
			c.query(E[q_ids], q_token_ids=q_ids, q_normalize=True, algorithm=hip.VK_ALG_RWMD, q_tags=q_tag, **kw)
			np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=2e-5, rtol=0)      # with the keys: exact
			c.query(E[q_ids], q_token_ids=q_ids, q_normalize=True, algorithm=hip.VK_ALG_RWMD, **kw)
			diff = np.abs(c.last_scores() - ref["all_scores"])
			# without them only slices that share at least two (id, tag) entries with the query can differ
			keys_q = set(zip(q_ids.tolist(), q_tag.tolist()))
			for s_ in np.nonzero(diff > 2e-5)[0]:
				a, b = int(off[s_]), int(off[s_ + 1])
				assert len(set(zip(ids[a:b].tolist(), tag_s[a:b].tolist())) & keys_q) >= 2
			worst, affected, total = max(worst, float(diff.max())), affected + int((diff > 2e-5).sum()), total + len(diff)
	assert 0 < affected < 0.1 * total and worst < 5e-4
	c.close()


@pytest.mark.parametrize("len_t", [20, 40])
@pytest.mark.parametrize("flags", [(True, True, True), (True, False, False)])
def test_tag_weighted_injective_rwmd_wide_query_over_a_corpus_with_long_slices(hip, oracle, len_t, flags):
	"""Queries of more than 16 tokens over a corpus that holds slices of more than 64: the injective RWMD takes the one-wave-per-slice
	fallback (vk_wide_kernel), which must minimise over the TAG-WEIGHTED similarities (round 2 read the unweighted ones there:
	found by the Index-level sweep of round 3, tests/test_gpu_index_sweep.py)"""
	rng = np.random.default_rng(77)
	d, n = 64, 120
	lens = rng.integers(2, 41, size=n)
	lens[[5, 60]] = [90, 130]
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	X = rng.standard_normal((int(off[-1]), d)).astype(np.float32)
	Xb, _ = oracle.normalize_rows_bf16(X)
	pos_s = rng.integers(1, 5, size=X.shape[0]).astype(np.int8)
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=X.shape[0], n_sentences=n)
	c.append_vectors(X, normalize=True)
	c.set_sentences(off)
	c.set_token_pos(pos_s)
	c.finalize()
	for rep in range(2):
		s = int(rng.integers(0, n))
		qv = (X[rng.integers(off[s], off[s + 1], size=len_t)] + 0.3 * rng.standard_normal((len_t, d))).astype(np.float32)
		Qb, _ = oracle.normalize_rows_bf16(qv)
		tw = rng.choice([0.25, 1.0, 2.5], size=len_t).astype(np.float32)
		q_pos = rng.integers(1, 5, size=len_t).astype(np.int8)
		kw = dict(tag_weights=tw, q_pos=q_pos, pos_mismatch_penalty=0.4, similarity_threshold=0.1, max_matches=10, min_score=-10.0, rwmd=flags)
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=off, X=Xb, Q=Qb, pos_s=pos_s, algorithm=oracle.ALG_RWMD, want_all_scores=True, **kw)
		got = c.query(qv, q_normalize=True, algorithm=hip.VK_ALG_RWMD, **kw)
		assert_same_results(got.trimmed(), ref, check_mapping=False, exact=True)
		np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=2e-5, rtol=0)
	c.close()


@pytest.mark.parametrize("len_t,window,step", [(12, 11, 4), (9, 16, 1), (24, 30, 7), (40, 20, 5), (12, 64, 16)])
@pytest.mark.parametrize("flags,full", [((True, True, True), False), ((True, False, False), False), ((True, False, True), False), ((False, False, True), True)])
def test_static_tag_weighted_transport_over_sliding_windows(hip, oracle, len_t, window, step, flags, full):
	"""The rewritten cells (static_vocab_fixup) are per slice, but the scoring kernels keep the rows of the slices of a wave in one
	strip: over sliding windows a row belongs to several slices.  Round 2 rewrote it in place and the neighbours read the rewritten
	value (found by the Index-level sweep, seed 4480: token windows of 11 every 4 tokens, the query a passage of the document).
	Queries cut from the token stream, so that the windows around them share many (id, tag) entries with the query."""
	rng = np.random.default_rng(1000 * len_t + window)
	V, d, n_tok = 120, 64, 900
	E = rng.standard_normal((V, d)).astype(np.float32)
	Eb, emag = oracle.normalize_rows_bf16(E)
	ids = rng.integers(0, V, size=n_tok).astype(np.int32)
	tag_of = rng.integers(1, 9, size=V).astype(np.int8)
	tag_s = tag_of[ids]
	pos_s = (tag_s % 3 + 1).astype(np.int8)
	start = np.arange(0, n_tok - window + 1, step, dtype=np.int64)
	end = start + window
	c = hip.Corpus(layout=hip.VK_LAYOUT_STATIC, d=d, n_tokens=n_tok, n_sentences=len(start), vocab_size=V, keep_magnitudes=True)
	c.append_vectors(E, normalize=True)
	c.set_token_ids(ids)
	c.set_slices(start, end)
	c.set_token_pos(pos_s)
	c.set_token_tags(tag_s)
	c.finalize()
	weights = np.array([0.5, 1.0, 3.0, 1.5, 0.75, 2.0, 1.0, 0.25, 1.25], dtype=np.float32)
	for rep in range(3):
		a = int(rng.integers(0, n_tok - len_t))
		q_ids = ids[a:a + len_t].copy()
		q_tag = tag_of[q_ids]
		q_pos = (q_tag % 3 + 1).astype(np.int8)
		kw = dict(tag_weights=weights[q_tag], q_pos=q_pos, pos_mismatch_penalty=0.25, similarity_threshold=0.05, max_matches=10, min_score=-10.0, rwmd=flags, wmd_full=full)
		ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=d, sent_off=start, sent_end=end, tok_id=ids, E=Eb, X_mag=emag[ids], Q=Eb[q_ids], q_ids=q_ids, Q_mag=emag[q_ids],
			pos_s=pos_s, tag_s=tag_s, q_tag=q_tag, algorithm=oracle.ALG_RWMD, want_all_scores=True, **kw)
		got = c.query(E[q_ids], q_token_ids=q_ids, q_normalize=True, algorithm=hip.VK_ALG_RWMD, q_tags=q_tag, **kw)
		if not full:
			np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=2e-5, rtol=0)
		assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=2e-5, tie_tol=2e-5, exact=not full)
	c.close()
