"""-m gpu: slices longer than the 64 tokens of the fused fast path (up to VK_MAX_SENT_LEN = 512) take a
second launch, one slice per wave; the slice table is padded so that no group of the main launch spans
a long slice.  HIP (through the C-ABI) against the oracle: scores, order, mappings."""

import numpy as np
import pytest

from vectorian_amd import synth

from helpers import assert_same_results, hip_static_corpus, prep_query

pytestmark = pytest.mark.gpu

EXP5L = ("table", (1 - 2.0 ** (-np.arange(0, 513) / 5)).astype(np.float32))
AFF = ("affine", 0.2, 0.05)


def vectors(n, d, seed):
	return synth.to_bf16_bits(synth.normalize_rows(np.random.default_rng(seed).standard_normal((n, d)).astype(np.float32)))


def mixed_lengths(seed, n=90):
	rng = np.random.default_rng(seed)
	lens = rng.integers(1, 41, size=n)
	# long slices first, last, alone, in a run, at every position of a group of 4
	for pos, ln in ((0, 65), (5, 100), (6, 257), (7, 70), (18, 512), (23, 131), (40, 66), (41, 64), (n - 1, 300)):
		lens[pos] = ln
	return np.concatenate(([0], np.cumsum(lens))).astype(np.int64)


def planted_query(Xb_f32, off, sent, len_t, seed):
	"""noisy copy of tokens of one (long) slice, spread out so that gaps matter"""
	rng = np.random.default_rng(seed)
	a, b = int(off[sent]), int(off[sent + 1])
	idx = np.sort(rng.choice(np.arange(a, b), size=min(len_t, b - a), replace=False))
	q = Xb_f32[idx] + 0.05 * rng.standard_normal((len(idx), Xb_f32.shape[1])).astype(np.float32)
	return synth.to_bf16_bits(synth.normalize_rows(q))


@pytest.mark.parametrize("d", [64, 300, 768])
def test_long_slices_contextual(hip, oracle, d):
	off = mixed_lengths(21)
	X = np.random.default_rng(22).standard_normal((int(off[-1]), d)).astype(np.float32)
	Xb = synth.to_bf16_bits(synth.normalize_rows(X))
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=Xb.shape[0], n_sentences=len(off) - 1)
	c.append_vectors(Xb, normalize=False)
	c.set_sentences(off)
	c.finalize()
	boost = np.random.default_rng(23).uniform(0.5, 1.5, size=len(off) - 1).astype(np.float32)
	for qi, (sent, len_t) in enumerate(((18, 9), (6, 16), (3, 5), (89, 12))):
		Qb = planted_query(X, off, sent, len_t, 30 + qi)
		for loc, ms, gaps, bst in ((0, 0.0, (0.1, 0.1), None), (0, 0.0, (EXP5L, EXP5L), boost), (1, -1e9, (EXP5L, EXP5L), None),
				(2, -1e9, (AFF, AFF), None), (1, -1e9, (0.05, 0.2), boost), (0, 0.0, (AFF, 0.1), None)):
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=off, X=Xb, Q=Qb, locality=loc,
				gap_s=gaps[0], gap_t=gaps[1], max_matches=15, min_score=ms, boost=bst, want_all_scores=True)
			got = c.query(Qb, q_normalize=False, locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=15, min_score=ms, boost=bst)
			assert_same_results(got.trimmed(), ref)
			if loc == 0:
				assert sent in got.sentence[:got.n]
			sc = c.last_scores()
			np.testing.assert_allclose(sc, ref["all_scores"], atol=1e-4)
	# transport: the relaxed distance and (since round 2, for queries of at most 16 tokens) the exact ones have no length limit
	ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=off, X=Xb, Q=Qb, algorithm=oracle.ALG_RWMD, max_matches=15)
	got = c.query(Qb, q_normalize=False, algorithm=hip.VK_ALG_RWMD, max_matches=15).trimmed()
	assert_same_results(got, ref, check_mapping=False, exact=True)
	ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=off, X=Xb, Q=Qb, algorithm=oracle.ALG_RWMD, rwmd=(False, False, True), wmd_full=True, max_matches=5)
	got = c.query(Qb, q_normalize=False, algorithm=hip.VK_ALG_RWMD, rwmd=(False, False, True), wmd_full=True, max_matches=5).trimmed()
	assert_same_results(got, ref, check_mapping=False, score_tol=2e-5, tie_tol=2e-5)
	c.close()


def test_long_slices_static_layout(hip, oracle):
	corpus = synth.make_static_corpus(120, 1, 40, 900, 100, seed=5)
	lens = np.diff(corpus["sent_off"]).copy()
	for pos, ln in ((2, 90), (3, 200), (64, 65), (119, 400)):
		lens[pos] = ln
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	rng = np.random.default_rng(6)
	corpus["sent_off"] = off
	corpus["tok_id"] = rng.integers(0, 900, size=int(off[-1])).astype(np.int32)
	c, Eb = hip_static_corpus(hip, corpus)
	for qi in range(3):
		s = (3, 119, 50)[qi]
		q_ids = corpus["tok_id"][off[s]:off[s] + 8 + qi].astype(np.int32)
		Qb = Eb[q_ids]
		for loc, ms, gaps in ((0, 0.0, (0.1, 0.1)), (1, -1e9, (EXP5L, EXP5L)), (2, -1e9, (AFF, AFF))):
			ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=100, sent_off=off, tok_id=corpus["tok_id"], E=Eb, Q=Qb, q_ids=q_ids,
				locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=10, min_score=ms)
			got = c.query(Qb, q_token_ids=q_ids, q_normalize=False, locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=10, min_score=ms)
			assert_same_results(got.trimmed(), ref)
	c.close()


def test_long_sliding_windows(hip, oracle):
	# window_size 4, window_step 2 over sentences of up to 40 tokens: slices of up to 160 tokens, overlapping
	rng = np.random.default_rng(40)
	lens = rng.integers(3, 41, size=60)
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	ids = np.arange(0, 60, 2)
	start = off[ids]
	end = off[np.minimum(ids + 4, 60)]
	assert (end - start).max() > 64
	Xb, Qb = vectors(int(off[-1]), 64, 41), vectors(7, 64, 42)
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=64, n_tokens=Xb.shape[0], n_sentences=len(ids))
	c.append_vectors(Xb, normalize=False)
	c.set_slices(start, end)
	c.finalize()
	for loc, ms, gaps in ((0, 0.0, (0.1, 0.1)), (1, -1e9, (EXP5L, EXP5L)), (2, -1e9, (AFF, AFF))):
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=64, sent_off=start, sent_end=end, X=Xb, Q=Qb, locality=loc,
			gap_s=gaps[0], gap_t=gaps[1], max_matches=12, min_score=ms)
		got = c.query(Qb, q_normalize=False, locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=12, min_score=ms).trimmed()
		assert_same_results(got, ref)
	c.close()


def test_only_long_slices_and_limit(hip, oracle):
	off = np.array([0, 512, 512 + 65, 512 + 65 + 200], dtype=np.int64)
	Xb, Qb = vectors(int(off[-1]), 48, 50), vectors(4, 48, 51)
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=48, n_tokens=Xb.shape[0], n_sentences=3)
	c.append_vectors(Xb, normalize=False)
	c.set_sentences(off)
	c.finalize()
	ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=48, sent_off=off, X=Xb, Q=Qb, gap_s=EXP5L, gap_t=EXP5L, max_matches=5)
	got = c.query(Qb, q_normalize=False, gap_s=EXP5L, gap_t=EXP5L, max_matches=5).trimmed()
	assert_same_results(got, ref)
	c.close()
	# (slices of more than 512 tokens: alignments only, tests/test_gpu_document_slices.py)


@pytest.mark.parametrize("alg,opts", [
	("wrd", dict(wrd_normalize=True)), ("wrd", dict(wrd_normalize=False)),
	("wmd", dict(rwmd=(False, False, True), wmd_full=True)), ("wmd", dict(rwmd=(False, False, False), wmd_full=True)),
])
@pytest.mark.parametrize("layout", ["contextual", "static"])
@pytest.mark.parametrize("len_t", [9, 20, 40, 64])
def test_exact_transport_over_long_slices(hip, oracle, layout, alg, opts, len_t):
	"""Word Rotator's Distance and the full WMD over a corpus with sentences of 65..400 tokens (upstream sizes its transport problems
	by the longest sentence, metric/alignment.h:357-358).  Queries of at most 16 tokens: bound pass of the one-slice-per-wave launch
	+ vk_wrd_exact_long_kernel<1> (state in LDS); 17..64 tokens: the multi-block kernel bounds the short slices, vk_long_bound_kernel
	the long ones, vk_wrd_exact_long_kernel<4> solves long candidates with its (supply, demand) arrays in global scratch"""
	rng = np.random.default_rng(12)
	n = 260
	lens = rng.integers(1, 50, size=n)
	lens[rng.integers(0, n, size=40)] = rng.integers(65, 400, size=40)
	lens[7] = 512
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	T, d = int(off[-1]), 64
	o_alg = oracle.ALG_WRD if alg == "wrd" else oracle.ALG_RWMD
	h_alg = hip.VK_ALG_WRD if alg == "wrd" else hip.VK_ALG_RWMD
	if layout == "static":
		V = 300
		E = (synth.make_vocab(V, d) * rng.lognormal(0, 0.3, size=(V, 1))).astype(np.float32)
		Eb, emag = oracle.normalize_rows_bf16(E)
		ids = synth.zipf_ids(T, V, rng)
		c = hip.Corpus(layout=hip.VK_LAYOUT_STATIC, d=d, n_tokens=T, n_sentences=n, vocab_size=V, keep_magnitudes=True)
		c.append_vectors(E, normalize=True)
		c.set_token_ids(ids)
	else:
		X = (rng.standard_normal((T, d)) * rng.lognormal(0, 0.3, size=(T, 1))).astype(np.float32)
		Xb, mag = oracle.normalize_rows_bf16(X)
		c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=T, n_sentences=n, keep_magnitudes=True)
		c.append_vectors(X, normalize=True)
	c.set_sentences(off)
	c.finalize()
	for qi in range(3):
		s = int(np.argsort(-lens)[qi * 5])                 # plant part of a long sentence
		if layout == "static":
			q_ids = ids[off[s] + 3:off[s] + 3 + len_t].astype(np.int32)
			ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=d, sent_off=off, tok_id=ids, E=Eb, X_mag=emag[ids], Q=Eb[q_ids], q_ids=q_ids, Q_mag=emag[q_ids],
				algorithm=o_alg, max_matches=n, min_score=-1.0, **opts)
			got = c.query(E[q_ids], q_token_ids=q_ids, q_normalize=True, algorithm=h_alg, max_matches=n, min_score=-1.0, **opts)
		else:
			qv = (X[off[s] + 3:off[s] + 3 + len_t] + 0.3 * rng.standard_normal((len_t, d))).astype(np.float32)
			Qb, qmag = oracle.normalize_rows_bf16(qv)
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=off, X=Xb, X_mag=mag, Q=Qb, Q_mag=qmag,
				algorithm=o_alg, max_matches=n, min_score=-1.0, n_threads=8, **opts)
			got = c.query(qv, q_normalize=True, algorithm=h_alg, max_matches=n, min_score=-1.0, **opts)
		assert len(ref["sentence"]) == n   # every sentence is ranked: all the long ones are solved exactly
		assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=2e-5, tie_tol=2e-5)
	c.close()


@pytest.mark.parametrize("len_t", [7, 24])
def test_transport_flow_outputs_of_long_winners(hip, oracle, len_t):
	"""want_flow on transport metrics over a corpus with slices of 65..300 tokens: the similarity rows of the winners and, for exact
	transport, the optimal plan (its marginals are the masses, its value the score) -- vk_topk_out.rows_per_winner sized by the
	corpus's longest slice, the long solver restating the plans of winners of more than 64 tokens"""
	rng = np.random.default_rng(8 + len_t)
	lens = rng.integers(3, 50, size=120)
	lens[rng.integers(0, 120, size=25)] = rng.integers(65, 300, size=25)
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	d = 64
	X = (rng.standard_normal((int(off[-1]), d)) * rng.lognormal(0, 0.3, size=(int(off[-1]), 1))).astype(np.float32)
	Xb, mag = oracle.normalize_rows_bf16(X)
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=X.shape[0], n_sentences=len(lens), keep_magnitudes=True)
	c.append_vectors(X, normalize=True)
	c.set_sentences(off)
	c.finalize()
	s_long = int(np.argmax(lens))
	qv = (X[off[s_long] + 5:off[s_long] + 5 + len_t] + 0.2 * rng.standard_normal((len_t, d))).astype(np.float32)   # planted in the longest slice
	Qb, qmag = oracle.normalize_rows_bf16(qv)
	n_long = 0
	for alg, kw in ((hip.VK_ALG_WRD, {}), (hip.VK_ALG_RWMD, dict(rwmd=(False, False, True), wmd_full=True)), (hip.VK_ALG_RWMD, {})):
		got = c.query(qv, algorithm=alg, q_normalize=True, max_matches=len(lens), min_score=-1.0, want_flow=True, **kw)
		assert got.n == len(lens) and got.sim_rows.shape[1] >= int(lens.max())
		for i in range(got.n):
			s = int(got.sentence[i])
			a, b = int(off[s]), int(off[s + 1])
			if b - a <= 64 and i > 12:
				continue
			n_long += b - a > 64
			S = oracle.sim_bf16(Xb[a:b], Qb)
			np.testing.assert_allclose(got.sim_rows[i, :b - a, :len_t], S, atol=2e-6)
			assert not got.sim_rows[i, b - a:].any()
			if alg == hip.VK_ALG_RWMD and not kw:
				continue
			G = got.plan[i, :len_t, :b - a].astype(np.float64)
			if alg == hip.VK_ALG_WRD:
				mt, ms = qmag / qmag.sum(), mag[a:b] / mag[a:b].sum()
			else:
				mt, ms = np.full(len_t, 1 / len_t), np.full(b - a, 1 / (b - a))
			np.testing.assert_allclose(G.sum(axis=1), mt, atol=2e-6)
			np.testing.assert_allclose(G.sum(axis=0), ms, atol=2e-6)
			raw = ((1.0 - np.maximum(1.0 - S.T, 0.0)) * G).sum() / G.sum()
			assert abs(raw - got.raw_score[i]) < 2e-5
	assert n_long >= 25
	c.close()


@pytest.mark.parametrize("len_t", [9, 24])
def test_only_slices_states_every_slice_long_ones_included(hip, oracle, len_t):
	"""vk_query_desc.only_slices (the debug hook's walk over every slice): no scoring pass, no selection -- the listed slices in the
	caller's order, whatever their score, with the aligner score, the score, the traceback and the similarity rows of the canonical
	arithmetic.  Over a slice table padded for long slices (sentence -> row of the table), for short and long queries, with a boost."""
	d = 64
	off = mixed_lengths(41)
	n = len(off) - 1
	X = np.random.default_rng(42).standard_normal((int(off[-1]), d)).astype(np.float32)
	Xb = synth.to_bf16_bits(synth.normalize_rows(X))
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=Xb.shape[0], n_sentences=n)
	c.append_vectors(Xb, normalize=False)
	c.set_sentences(off)
	c.finalize()
	Qb = planted_query(X, off, 18, len_t, 43)
	boost = np.random.default_rng(44).uniform(0.5, 1.5, size=n).astype(np.float32)
	kw = dict(locality=0, gap_s=EXP5L, gap_t=EXP5L, boost=boost)
	ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=off, X=Xb, Q=Qb, max_matches=n, min_score=-1.0, **kw)
	by_sent = {int(s): i for i, s in enumerate(ref["sentence"])}
	assert len(by_sent) == n                                   # the oracle states every slice once min_score is out of the way
	order = np.random.default_rng(45).permutation(n).astype(np.int64)   # any order, any subset
	for ids in (order[:37], order[37:], np.array([18, 18, 6], dtype=np.int64)):
		got = c.query(Qb, q_normalize=False, only_slices=ids, want_rows=True, **kw)
		assert got.n == len(ids) and list(got.sentence[:got.n]) == list(ids)
		for i, s in enumerate(ids):
			j = by_sent[int(s)]
			assert np.float32(got.score[i]).view(np.uint32) == np.float32(ref["score"][j]).view(np.uint32)
			assert np.float32(got.raw_score[i]).view(np.uint32) == np.float32(ref["raw"][j]).view(np.uint32)
			assert list(got.mapping[i]) == list(ref["mapping"][j])
			a, b = int(off[s]), int(off[s + 1])
			S = oracle.sim_bf16(Xb[a:b], Qb)
			assert (got.sim_rows[i, :b - a, :len_t] == S).all()
	with pytest.raises(hip.VkError):
		c.query(Qb, q_normalize=False, only_slices=np.array([n], dtype=np.int64), **kw)              # out of range
	# the transports take the list too (ABI 10): relaxed WMD restated on the host from the rows -- the oracle's floats for every listed
	# slice, the long ones included --, exact transports solved slice by slice
	ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=off, X=Xb, Q=Qb, algorithm=oracle.ALG_RWMD, max_matches=n, min_score=-10.0)
	by_sent = {int(s): j for j, s in enumerate(ref["sentence"])}
	got = c.query(Qb, q_normalize=False, only_slices=ids, algorithm=hip.VK_ALG_RWMD)
	assert got.n == len(ids) and list(got.sentence[:got.n]) == list(ids)
	for i, s in enumerate(ids):
		assert np.float32(got.score[i]).view(np.uint32) == np.float32(ref["score"][by_sent[int(s)]]).view(np.uint32)
	# a submatch weight (ABI 11): the score of a listed slice over the reference score of its own traceback (metric/alignment.h:84-106)
	ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=off, X=Xb, Q=Qb, max_matches=n, min_score=-1.0, submatch_weight=1.0, **kw)
	by_sent = {int(s): j for j, s in enumerate(ref["sentence"])}
	got = c.query(Qb, q_normalize=False, only_slices=ids, submatch_weight=1.0, **kw)
	assert got.n == len(ids) and list(got.sentence[:got.n]) == list(ids)
	for i, s in enumerate(ids):
		j = by_sent[int(s)]
		assert np.float32(got.score[i]).view(np.uint32) == np.float32(ref["score"][j]).view(np.uint32)
		assert list(got.mapping[i]) == list(ref["mapping"][j])
	c.close()
