"""-m gpu: vk_query_batch.  The GEMM-shaped RWMD path (uniform sentence length, contextual layout) and
the generic per-query fallback must both return what single vk_query calls / the oracle return."""

import numpy as np
import pytest

from vectorian_amd import synth

from helpers import assert_same_results, hip_contextual_corpus, prep_contextual, prep_query

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("length,d,len_t,flags", [
	(32, 300, 10, (True, True, True)),
	(16, 300, 5, (True, False, False)),
	(48, 128, 16, (True, False, True)),
	(64, 300, 3, (True, True, True)),
])
def test_rwmd_gemm_batch(hip, oracle, length, d, len_t, flags):
	n = 1037   # not a multiple of the 4 x TPW tiles of a workgroup: exercises the tail
	corpus = synth.make_contextual_corpus(n, length, length, 2000, d)
	Xb = prep_contextual(corpus)
	c = hip_contextual_corpus(hip, corpus, Xb)
	qs = [prep_query(q) for q in synth.make_queries(corpus, 9, len_t)]
	qs[3] = qs[3][:max(1, len_t - 2)]          # queries of different lengths in one batch
	boost = np.random.default_rng(1).uniform(0.5, 1.5, size=n).astype(np.float32) if length == 32 else None
	outs = c.query_batch(qs, algorithm=hip.VK_ALG_RWMD, rwmd=flags, q_normalize=False, max_matches=12, min_score=0.0, boost=boost)
	assert len(outs) == len(qs)
	for Qb, got in zip(qs, outs):
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=corpus["sent_off"], X=Xb, Q=Qb,
			algorithm=oracle.ALG_RWMD, rwmd=flags, max_matches=12, min_score=0.0, boost=boost)
		assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=2e-5, tie_tol=2e-5)
		single = c.query(Qb, algorithm=hip.VK_ALG_RWMD, rwmd=flags, q_normalize=False, max_matches=12, min_score=0.0, boost=boost)
		np.testing.assert_allclose(got.score[:got.n], single.score[:single.n], atol=2e-6)
	c.close()


@pytest.mark.parametrize("d,len_t,n_q,flags", [
	(300, 10, 10, (True, True, True)),      # 3 queries per 32-row tile, last tile holds one
	(300, 10, 8, (True, False, True)),      # last tile holds two
	(300, 7, 6, (True, False, False)),
	(300, 16, 5, (True, True, True)),       # longer than 10 tokens: 2 queries per tile
	(128, 10, 7, (True, True, True)),
	(128, 13, 4, (True, False, False)),
	(300, 10, 53, (True, True, True)),      # >= 32 queries of <= 10 tokens: 16 queries per five tiles (vk_rwmd_batch32d_kernel), last super tile partly empty
	(300, 9, 32, (True, False, True)),
	(128, 10, 40, (True, True, False)),
])
def test_rwmd_batch_32_token_sentences(hip, oracle, d, len_t, n_q, flags):
	"""32-token sentences run on the 32x32x16 MFMA kernels (queries share A tiles)."""
	n = 1000 + 13
	corpus = synth.make_contextual_corpus(n, 32, 32, 2000, d)
	Xb = prep_contextual(corpus)
	c = hip_contextual_corpus(hip, corpus, Xb)
	qs = [prep_query(q) for q in synth.make_queries(corpus, n_q, len_t)]
	qs[1] = qs[1][:max(1, len_t - 3)]
	qs[-1] = qs[-1][:1]
	for i in range(5, n_q, 7):              # queries of every length in one batch: whole and split rows of the dense layout
		qs[i] = qs[i][:1 + (i * 3) % len_t]
	if flags[1] and not flags[2]:
		flags = (flags[0], True, True)       # symmetric needs nbow (wmd.h:441-449)
	outs = c.query_batch(qs, algorithm=hip.VK_ALG_RWMD, rwmd=flags, q_normalize=False, max_matches=12, min_score=0.0)
	for Qb, got in zip(qs, outs):
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=corpus["sent_off"], X=Xb, Q=Qb,
			algorithm=oracle.ALG_RWMD, rwmd=flags, max_matches=12, min_score=0.0)
		assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=2e-5, tie_tol=2e-5)
	c.close()


EXP5 = ("table", (1 - 2.0 ** (-np.arange(0, 65) / 5)).astype(np.float32))


@pytest.mark.parametrize("d,lo,hi", [(300, 32, 32), (64, 1, 64), (768, 8, 40)])
def test_shared_pass_alignment_batch(hip, oracle, d, lo, hi):
	"""queries with common options share one pass over the token tiles (vk_score_batch_kernel): results as from
	single queries and as the oracle's"""
	n = 901
	corpus = synth.make_contextual_corpus(n, lo, hi, 1500, d)
	Xb = prep_contextual(corpus)
	c = hip_contextual_corpus(hip, corpus, Xb)
	qs = [prep_query(q) for q in synth.make_queries(corpus, 7, 11)]
	qs[1], qs[4], qs[6] = qs[1][:3], qs[4][:8], qs[6][:1]     # different lengths in one batch
	boost = np.random.default_rng(1).uniform(0.5, 1.5, size=n).astype(np.float32)
	for loc, ms, gaps, bst in ((0, 0.0, (0.1, 0.1), None), (0, 0.0, (EXP5, EXP5), boost), (1, -1e9, (EXP5, EXP5), None),
			(2, -1e9, (("affine", 0.2, 0.05),) * 2, None), (1, -1e9, (0.05, 0.2), boost)):
		kw = dict(locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=9, min_score=ms, boost=bst)
		outs = c.query_batch(qs, q_normalize=False, **kw)
		assert c.last_timings()["score_ms"] > 0
		for Qb, got in zip(qs, outs):
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=corpus["sent_off"], X=Xb, Q=Qb, **kw)
			assert_same_results(got.trimmed(), ref)
			single = c.query(Qb, q_normalize=False, **kw)
			assert (got.sentence[:got.n] == single.sentence[:single.n]).all()
			np.testing.assert_array_equal(got.score[:got.n], single.score[:single.n])
			np.testing.assert_array_equal(got.mapping[:got.n], single.mapping[:single.n])
	# relaxed WMD over ragged sentences takes the same shared pass
	for flags in ((True, True, True), (True, False, False)):
		outs = c.query_batch(qs, algorithm=hip.VK_ALG_RWMD, rwmd=flags, q_normalize=False, max_matches=9)
		for Qb, got in zip(qs, outs):
			single = c.query(Qb, algorithm=hip.VK_ALG_RWMD, rwmd=flags, q_normalize=False, max_matches=9)
			# the GEMM kernels (uniform 32-token sentences; since round 2 ragged corpora and 768-d rows too): other summation order
			np.testing.assert_allclose(got.score[:got.n], single.score[:single.n], atol=2e-6)
	c.close()


def test_batch_fallback_for_alignment_and_ragged(hip, oracle):
	corpus = synth.make_contextual_corpus(400, 2, 30, 800, 64)
	Xb = prep_contextual(corpus)
	c = hip_contextual_corpus(hip, corpus, Xb)
	qs = [prep_query(q) for q in synth.make_queries(corpus, 4, 6)]
	outs = c.query_batch(qs, locality=0, gap_s=0.1, gap_t=0.1, q_normalize=False, max_matches=7)
	for Qb, got in zip(qs, outs):
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=64, sent_off=corpus["sent_off"], X=Xb, Q=Qb, gap_s=0.1, gap_t=0.1, max_matches=7)
		assert_same_results(got.trimmed(), ref)
	outs = c.query_batch(qs, algorithm=hip.VK_ALG_RWMD, q_normalize=False, max_matches=7)   # ragged: per-query path
	for Qb, got in zip(qs, outs):
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=64, sent_off=corpus["sent_off"], X=Xb, Q=Qb, algorithm=oracle.ALG_RWMD, max_matches=7)
		assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=1e-5)
	c.close()


def test_batch_with_long_queries(hip, oracle):
	"""a batch that mixes short and long queries: the long ones fall out of the shared pass onto the multi-block kernel,
	each result set as from a single query and as the oracle's"""
	n, d = 777, 300
	corpus = synth.make_contextual_corpus(n, 1, 64, 1500, d)
	Xb = prep_contextual(corpus)
	c = hip_contextual_corpus(hip, corpus, Xb)
	qs = [prep_query(q) for q in synth.make_queries(corpus, 10, 40)]
	qs = [q[:m] for q, m in zip(qs, (7, 24, 40, 16, 17, 33, 3, 32, 12, 20))]
	for loc, ms, gaps in ((0, 0.0, (0.1, 0.1)), (1, -1e9, (EXP5, EXP5))):
		kw = dict(locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=8, min_score=ms)
		outs = c.query_batch(qs, q_normalize=False, **kw)
		for Qb, got in zip(qs, outs):
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=corpus["sent_off"], X=Xb, Q=Qb, **kw)
			assert_same_results(got.trimmed(), ref)
			single = c.query(Qb, q_normalize=False, **kw)
			np.testing.assert_array_equal(got.score[:got.n], single.score[:single.n])
			np.testing.assert_array_equal(got.mapping[:got.n], single.mapping[:single.n])
	c.close()


@pytest.mark.parametrize("d,lo,hi,len_t,n_q,flags", [
	(300, 1, 64, 10, 19, (True, True, True)),       # every bucket (16 / 32 / 48 / 64 padded tokens), lengths of all residues
	(300, 8, 64, 7, 12, (True, False, False)),
	(128, 3, 40, 16, 9, (True, True, True)),
	(768, 8, 64, 10, 11, (True, True, True)),       # config 5's rows: 24 K-steps, one wave per SIMD
	(768, 32, 32, 10, 6, (True, False, True)),      # uniform 32-token sentences at 768-d: the 16-row kernel on the resident tiles
	(768, 48, 48, 12, 5, (True, True, True)),
])
def test_rwmd_gemm_batch_ragged_and_768(hip, oracle, d, lo, hi, len_t, n_q, flags):
	"""the GEMM-shaped RWMD over corpora the round-1 kernels refused: ragged sentence lengths (padded length buckets of a copy
	of the corpus, scores back at the original sentence indices, empty slices left out) and 768-d rows (wmd.h:287-416 has no
	shape restriction)"""
	n = 700
	corpus = synth.make_contextual_corpus(n, lo, hi, 2000, d)
	off = corpus["sent_off"].copy()
	if lo < hi:                       # a few empty slices too: sliding windows over them would be skipped upstream (document.h:160)
		start, end = off[:-1].copy(), off[1:].copy()
		end[[5, 77, n - 1]] = start[[5, 77, n - 1]]
	else:
		start, end = off[:-1], off[1:]
	Xb = prep_contextual(corpus)
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=Xb.shape[0], n_sentences=n)
	c.append_vectors(Xb, normalize=False)
	c.set_slices(start, end)
	c.finalize()
	qs = [prep_query(q) for q in synth.make_queries(corpus, n_q, len_t)]
	qs[1] = qs[1][:max(1, len_t - 3)]
	qs[-1] = qs[-1][:1]
	boost = np.random.default_rng(2).uniform(0.5, 1.5, size=n).astype(np.float32) if d == 300 else None
	for rep in range(2):              # the second call runs on the layout the first one built
		outs = c.query_batch(qs, algorithm=hip.VK_ALG_RWMD, rwmd=flags, q_normalize=False, max_matches=9, min_score=0.0, boost=boost)
		for Qb, got in zip(qs, outs):
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=start, sent_end=end, X=Xb, Q=Qb,
				algorithm=oracle.ALG_RWMD, rwmd=flags, max_matches=9, min_score=0.0, boost=boost)
			assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=2e-5, tie_tol=2e-5)
	c.close()


def test_batch_rows_of_relaxed_wmd_winners(hip):
	"""want_flow on a batch of relaxed-WMD queries: the similarity rows of every query's winners (one launch for the batch), as the
	single-query path returns them -- on the GEMM kernels (300-d, ragged), on the shared pass (96-d), with queries of different
	lengths, queries without winners and fewer winners than k"""
	for d in (300, 96):
		corpus = synth.make_contextual_corpus(700, 3, 40, 900, d)
		X = corpus["X"]
		c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=X.shape[0], n_sentences=700)
		c.append_vectors(X, normalize=True)
		c.set_sentences(corpus["sent_off"])
		c.finalize()
		rng = np.random.default_rng(d)
		qs = [q["vectors"] for q in synth.make_queries(corpus, 10, 4)] + [q["vectors"] for q in synth.make_queries(corpus, 11, 9, seed=5)]
		qs.append(-qs[0])     # nothing above min_score
		kw = dict(algorithm=hip.VK_ALG_RWMD, rwmd=(True, True, True), q_normalize=True, max_matches=5, min_score=0.35, want_flow=True)
		outs = c.query_batch(qs, **kw)
		assert outs[-1].n == 0
		for q, b in zip(qs, outs):
			one = c.query(q, **kw)
			assert b.n == one.n and (b.sentence[:b.n] == one.sentence[:one.n]).all()
			np.testing.assert_allclose(b.score[:b.n], one.score[:one.n], atol=2e-6)
			np.testing.assert_array_equal(b.sim_rows[:b.n, :, :one.sim_rows.shape[2]], one.sim_rows[:one.n])
			assert not b.sim_rows[b.n:].any()
		c.close()


@pytest.mark.parametrize("shape,n_q,len_t,flags", [
	("ragged40", 9, 6, (True, True, True)),        # 3 queries per tile, slices of 1..40 tokens: both length buckets, empty slices
	("ragged40", 53, 10, (True, True, True)),      # >= 32 ten-token queries: 16 per five tiles (the dense epilogues)
	("uniform32", 40, 9, (True, False, True)),     # every slice 32 tokens: no bucket lists
	("ragged64", 7, 16, (True, True, True)),       # queries of up to 16 tokens: 2 per tile; slices of up to 64 tokens
	("ragged64", 34, 10, (True, False, False)),    # bow
	("wide", 12, 8, (True, True, True)),           # 96-d rows: no GEMM kernel exists for this width, the table kernel takes any
])
def test_static_rwmd_batch(hip, oracle, shape, n_q, len_t, flags):
	"""vk_query_batch over the static layout (round 4): one similarity table over the vocabulary for the whole batch
	(vk_table_batch_kernel), one gather pass over the token ids (vk_rwmd_static32_kernel), winners restated from canonical rows with
	the vocabulary keys -- slice ids and scores equal to vk_query's and to the oracle's bit for bit.
	Reference: metric/static.cpp:9-78 (table, sim[id(t_j)][j] = 1, clip), slice/static.h:71-75 (gather by token id)."""
	n, lo, hi, V, d = {"ragged40": (1500, 0, 40, 400, 300), "uniform32": (1003, 32, 32, 300, 300), "ragged64": (900, 1, 64, 500, 128),
		"wide": (700, 2, 50, 350, 96)}[shape]
	corpus = synth.make_static_corpus(n, lo, hi, V, d)
	from helpers import hip_static_corpus
	c, Eb = hip_static_corpus(hip, corpus)
	rng = np.random.default_rng(7)
	qids, qs = [], []
	for i in range(n_q):
		ids = rng.integers(0, 60 if i % 2 else V, size=1 + (i * 3) % len_t if i % 5 == 3 else len_t).astype(np.int32)   # frequent words: the query's own ids occur in slices
		if len(ids) > 3:
			ids[3] = ids[0]            # a repeated query token: one vocabulary entry of mass 2
		if i % 7 == 2:
			ids[-1] = -1               # a word the vocabulary does not hold: no diagonal cell (its vector: some other word's)
		qids.append(ids)
		qs.append(Eb[np.where(ids >= 0, ids, 5)])
	boost = rng.uniform(0.5, 1.5, size=n).astype(np.float32) if shape == "ragged40" else None
	kw = dict(algorithm=hip.VK_ALG_RWMD, rwmd=flags, q_normalize=False, max_matches=12, min_score=0.0, boost=boost)
	outs = c.query_batch(qs, token_ids=qids, **kw)
	assert len(outs) == n_q
	for Qb, ids, got in zip(qs, qids, outs):
		one = c.query(Qb, q_token_ids=ids, **kw)
		assert got.n == one.n and (got.sentence[:got.n] == one.sentence[:one.n]).all()
		assert (got.score[:got.n].view(np.uint32) == one.score[:one.n].view(np.uint32)).all(), (got.score[:got.n], one.score[:one.n])
		assert (got.sim_rows[:got.n, :64] == one.sim_rows[:one.n, :64]).all()   # the rows the flows are stated from
		ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=d, sent_off=corpus["sent_off"], tok_id=corpus["tok_id"], E=Eb, Q=Qb, q_ids=ids,
			algorithm=oracle.ALG_RWMD, rwmd=flags, max_matches=12, min_score=0.0, boost=boost)
		assert_same_results(got.trimmed(), ref, check_mapping=False, exact=True)
	# without flows: the scores of the gather pass itself (MFMA table cells, fp32 sums in another order) within 2e-5 of the oracle's
	outs = c.query_batch(qs, token_ids=qids, want_flow=False, **kw)
	for Qb, ids, got in zip(qs, qids, outs):
		ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=d, sent_off=corpus["sent_off"], tok_id=corpus["tok_id"], E=Eb, Q=Qb, q_ids=ids,
			algorithm=oracle.ALG_RWMD, rwmd=flags, max_matches=12, min_score=0.0, boost=boost)
		assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=2e-5, tie_tol=2e-5)
	c.close()
