"""-m gpu: whole documents as slices.  A corpus that holds a slice of more than VK_MAX_SENT_LEN (512) tokens -- up to VK_MAX_DOC_LEN
= 32767, what the int16 of a mapping can name, upstream's own bound (metric/alignment.h:357-358) -- is scored by alignments through
vk_wide_kernel's global-state form: one wave per slice, the column history of general gaps and the traceback in global memory.
The same form takes queries of more than 16 tokens over slices whose state exceeds the LDS (VK_ERR_UNSUPPORTED until round 3).
HIP (through the C-ABI) against the oracle: slice ids, scores and tracebacks bit for bit, the scores of ALL slices within 1e-4."""

import numpy as np
import pytest

from vectorian_amd import synth

from helpers import assert_same_results, hip_static_corpus

pytestmark = pytest.mark.gpu


def exp5(n):
	return ("table", (1 - 2.0 ** (-np.arange(0, n + 1) / 5)).astype(np.float32))


AFF = ("affine", 0.2, 0.05)


def document_lengths(seed, n, docs):
	rng = np.random.default_rng(seed)
	lens = rng.integers(1, 60, size=n)
	for pos, ln in docs:
		lens[pos] = ln
	return np.concatenate(([0], np.cumsum(lens))).astype(np.int64)


def planted_query(X, off, sent, len_t, seed, spread=True):
	"""noisy copy of tokens of one document, spread out over it so that long gaps matter"""
	rng = np.random.default_rng(seed)
	a, b = int(off[sent]), int(off[sent + 1])
	if spread:
		idx = np.sort(rng.choice(np.arange(a, b), size=min(len_t, b - a), replace=False))
	else:
		s0 = int(rng.integers(a, max(a + 1, b - 2 * len_t)))
		idx = np.sort(rng.choice(np.arange(s0, min(b, s0 + 2 * len_t)), size=min(len_t, b - s0), replace=False))
	q = X[idx] + 0.05 * rng.standard_normal((len(idx), X.shape[1])).astype(np.float32)
	return synth.to_bf16_bits(synth.normalize_rows(q))


def contextual(hip, off, d, seed):
	X = np.random.default_rng(seed).standard_normal((int(off[-1]), d)).astype(np.float32)
	Xb = synth.to_bf16_bits(synth.normalize_rows(X))
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=Xb.shape[0], n_sentences=len(off) - 1)
	c.append_vectors(Xb, normalize=False)
	c.set_sentences(off)
	c.finalize()
	return c, X, Xb


@pytest.mark.parametrize("d,len_t", [(64, 5), (64, 12), (300, 16), (64, 20), (96, 40), (64, 64)])
def test_documents_contextual(hip, oracle, d, len_t):
	"""documents of 513 .. 3,000 tokens between short slices, every locality and gap family, queries of 5 .. 64 tokens (1 .. 4 column
	blocks of the one-wave-per-slice kernel)"""
	docs = ((0, 513), (5, 1200), (6, 700), (7, 65), (18, 3000), (23, 131), (40, 2049), (41, 512), (69, 900))
	off = document_lengths(21, 70, docs)
	c, X, Xb = contextual(hip, off, d, 22)
	w = exp5(int(np.diff(off).max()))
	boost = np.random.default_rng(23).uniform(0.5, 1.5, size=len(off) - 1).astype(np.float32)
	for qi, (sent, spread) in enumerate(((18, True), (40, False), (3, True))):
		Qb = planted_query(X, off, sent, len_t, 30 + qi, spread)
		for loc, ms, gaps, bst in ((0, 0.0, (0.1, 0.1), None), (0, 0.0, (w, w), boost), (1, -1e9, (w, w), None),
				(2, -1e9, (AFF, AFF), None), (1, -1e9, (0.05, 0.2), boost), (2, -1e9, (w, 0.1), None)):
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=off, X=Xb, Q=Qb, locality=loc,
				gap_s=gaps[0], gap_t=gaps[1], max_matches=12, min_score=ms, boost=bst, want_all_scores=True, n_threads=8)
			got = c.query(Qb, q_normalize=False, locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=12, min_score=ms, boost=bst)
			assert_same_results(got.trimmed(), ref)
			np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-4)
	c.close()


def test_longest_document(hip, oracle):
	"""one document of VK_MAX_DOC_LEN tokens (linear and affine gaps: the oracle's general-gap recurrence is cubic) and the limit"""
	n_max = hip.VK_MAX_DOC_LEN
	off = np.array([0, 40, 40 + n_max, 40 + n_max + 700, 40 + n_max + 700 + 9], dtype=np.int64)
	c, X, Xb = contextual(hip, off, 32, 5)
	for len_t in (8, 24):
		Qb = planted_query(X, off, 1, len_t, 6 + len_t)
		for loc, ms, gaps in ((0, 0.0, (0.1, 0.1)), (1, -1e9, (0.02, 0.1)), (2, -1e9, (AFF, AFF))):
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=32, sent_off=off, X=Xb, Q=Qb, locality=loc,
				gap_s=gaps[0], gap_t=gaps[1], max_matches=4, min_score=ms, want_all_scores=True)
			got = c.query(Qb, q_normalize=False, locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=4, min_score=ms)
			assert_same_results(got.trimmed(), ref)
			if loc == 0:
				assert got.sentence[0] == 1
			np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-4)
	# the relaxed 1:1 word mover's distance streams a document too, its rows come back through global memory
	ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=32, sent_off=off, X=Xb, Q=Qb, algorithm=oracle.ALG_RWMD, max_matches=4, min_score=-1.0, want_all_scores=True)
	got = c.query(Qb, q_normalize=False, algorithm=hip.VK_ALG_RWMD, max_matches=4, min_score=-1.0)
	assert_same_results(got.trimmed(), ref, check_mapping=False, exact=True)
	np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-5)
	# the 1:n form and the exact transports keep a slice's bag of words in LDS: refused before anything is enqueued
	for kw in (dict(algorithm=hip.VK_ALG_RWMD, rwmd=(False, True, True)), dict(algorithm=hip.VK_ALG_RWMD, rwmd=(False, False, True), wmd_full=True),
			dict(algorithm=hip.VK_ALG_WRD)):
		with pytest.raises(hip.VkError) as e:
			c.query(Qb, q_normalize=False, max_matches=3, **kw)
		assert e.value.status in (hip.VK_ERR_UNSUPPORTED, hip.VK_ERR_STATE)   # (WRD: this corpus keeps no magnitudes either)
	c.close()
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=16, n_tokens=n_max + 1, n_sentences=1)
	c.append_vectors(np.ones((n_max + 1, 16), np.float32))
	with pytest.raises(hip.VkError):
		c.set_sentences(np.array([0, n_max + 1], dtype=np.int64))
	c.close()


@pytest.mark.parametrize("len_t", [1, 2, 3, 15, 16])
def test_documents_shortest_and_widest_queries(hip, oracle, len_t):
	"""the edges of vk_doc_kernel's lane layout: one query column (no left neighbour at all), two, three, and all sixteen; documents
	whose length leaves no room for a sixteen-step block (66, 79), exactly one (81 with the right alignment), many; every gap family"""
	docs = ((0, 600), (1, 66), (2, 79), (3, 81), (4, 97), (9, 1100), (10, 513), (20, 2000))
	off = document_lengths(51, 30, docs)
	c, X, Xb = contextual(hip, off, 32, 52)
	w = exp5(int(np.diff(off).max()))
	for qi, sent in enumerate((20, 9, 4)):
		Qb = planted_query(X, off, sent, len_t, 90 + qi)
		for loc, ms, gaps in ((0, 0.0, (0.1, 0.1)), (1, -1e9, (0.05, 0.2)), (2, -1e9, (AFF, AFF)), (0, 0.0, (w, w)), (1, -1e9, (w, w)), (2, -1e9, (w, 0.1))):
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=32, sent_off=off, X=Xb, Q=Qb, locality=loc,
				gap_s=gaps[0], gap_t=gaps[1], max_matches=10, min_score=ms, want_all_scores=True, n_threads=8)
			got = c.query(Qb, q_normalize=False, locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=10, min_score=ms)
			assert_same_results(got.trimmed(), ref)
			np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-4)
	c.close()


@pytest.mark.parametrize("d,precision", [(300, "bf16"), (64, "bf16"), (40, "f32")])
def test_sentence_queries_over_sentences_paragraphs_and_documents(hip, oracle, monkeypatch, d, precision):
	"""a query of 17 .. 64 tokens over a corpus that holds all three kinds of slices: at most 64 tokens (the multi-block kernel), 65 .. 512
	and beyond (vk_docw_kernel: the skewed sweep across the wave; 300-d bf16 rows with a half K-step and 64-d rows in registers, fp32 rows
	through the generic tile routine), linear and affine gaps, every locality; then every slice on the sweep (VK_NO_SCORE32)"""
	docs = ((0, 700), (4, 65), (5, 512), (6, 513), (17, 129), (30, 2100), (31, 64), (49, 300))
	off = document_lengths(61, 50, docs)
	X = np.random.default_rng(62).standard_normal((int(off[-1]), d)).astype(np.float32)
	Xn = synth.normalize_rows(X)
	Xb = Xn if precision == "f32" else synth.to_bf16_bits(Xn)
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=Xb.shape[0], n_sentences=len(off) - 1, precision=precision)
	c.append_vectors(Xb, normalize=False)
	c.set_sentences(off)
	c.finalize()
	boost = np.random.default_rng(63).uniform(0.5, 1.5, size=len(off) - 1).astype(np.float32)
	for round_ in range(2):
		if round_ == 1:
			monkeypatch.setenv("VK_NO_SCORE32", "1")
		for qi, (sent, len_t) in enumerate(((30, 17), (5, 33), (0, 48), (12, 64), (6, 25))):
			rng = np.random.default_rng(100 + qi)
			a, b = int(off[sent]), int(off[sent + 1])
			idx = np.sort(rng.choice(np.arange(a, b), size=min(len_t, b - a), replace=False))
			q = synth.normalize_rows(X[idx] + 0.05 * rng.standard_normal((len(idx), d)).astype(np.float32))
			if len(q) < len_t:
				q = np.concatenate((q, synth.normalize_rows(rng.standard_normal((len_t - len(q), d)).astype(np.float32))))
			Qb = q if precision == "f32" else synth.to_bf16_bits(q)
			for loc, ms, gaps, bst in ((0, 0.0, (0.1, 0.1), None), (1, -1e9, (0.05, 0.2), boost), (2, -1e9, (AFF, AFF), None), (0, 0.0, (AFF, 0.1), boost)):
				ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=off, X=Xb, Q=Qb, locality=loc,
					gap_s=gaps[0], gap_t=gaps[1], max_matches=11, min_score=ms, boost=bst, want_all_scores=True, n_threads=8)
				got = c.query(Qb, q_normalize=False, locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=11, min_score=ms, boost=bst)
				assert_same_results(got.trimmed(), ref)
				np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-4)
	c.close()


def saturating(n, t, seed):
	"""a gap table that rises (not monotonically: nothing asks for that) up to k = t - 1 and is constant from k = t on"""
	rng = np.random.default_rng(seed)
	w = np.sort(rng.uniform(0.02, 0.9, size=n + 1)).astype(np.float32)
	w[1:t][::3] *= np.float32(0.5)
	w[0] = 0.0
	w[t:] = w[t]
	return ("table", w)


@pytest.mark.parametrize("tail", [1, 2, 3, 9, 34, 66, 126, 127, 400])
def test_documents_general_gaps_by_where_the_table_saturates(hip, oracle, tail):
	"""general gaps over documents: vk_doc_kernel's forms by the number of 8-candidate chunks a DPP row scans (table constant from
	k = 1, 2, 3, 9, 34, 66, 126 on), and the tables it leaves to vk_wide_kernel (127: beyond its ring of rows; 400: the matrix form).
	Scores of every slice and the winners' tracebacks against the oracle, every locality, gap_t a table of its own"""
	docs = ((0, 520), (3, 777), (9, 1500), (10, 64), (11, 65), (25, 129), (39, 1023))
	off = document_lengths(31, 40, docs)
	c, X, Xb = contextual(hip, off, 64, 32)
	n = int(np.diff(off).max())
	ws, wt = saturating(n, tail, 40 + tail), saturating(n, min(tail, 7), 41 + tail)
	# (17 .. 32 tokens: vk_docg_kernel -- two blocks of columns, the candidates dealt to two shares; its forms by chunks likewise)
	for qi, (sent, spread, len_t) in enumerate(((9, True, 7), (39, False, 16), (0, True, 11), (9, True, 20), (39, False, 32), (25, True, 17))):
		Qb = planted_query(X, off, sent, len_t, 60 + qi, spread)
		if len(Qb) < len_t:   # (a slice shorter than the query: random tokens behind the planted ones)
			extra = synth.normalize_rows(np.random.default_rng(600 + qi).standard_normal((len_t - len(Qb), 64)).astype(np.float32))
			Qb = np.concatenate((Qb, synth.to_bf16_bits(extra)))
		wt = saturating(n, min(tail, 7 if len_t <= 16 else 19), 41 + tail)
		for loc, ms in ((0, 0.0), (1, -1e9), (2, -1e9)):
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=64, sent_off=off, X=Xb, Q=Qb, locality=loc,
				gap_s=ws, gap_t=wt, max_matches=9, min_score=ms, want_all_scores=True, n_threads=8)
			got = c.query(Qb, q_normalize=False, locality=loc, gap_s=ws, gap_t=wt, max_matches=9, min_score=ms)
			assert_same_results(got.trimmed(), ref)
			np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-4)
	c.close()


@pytest.mark.parametrize("len_t", [9, 30])
def test_documents_static_layout_and_tag_weights(hip, oracle, len_t):
	"""token ids + vocabulary table (the reference's static layout), repeated words, sim[id(t_j)][j] = 1; with and without the
	tag-weighted modifier; submatch_weight (candidate rounds of tracebacks)"""
	corpus = synth.make_static_corpus(60, 1, 40, 900, 100, seed=5)
	lens = np.diff(corpus["sent_off"]).copy()
	for pos, ln in ((2, 600), (3, 2500), (30, 65), (59, 1100)):
		lens[pos] = ln
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	rng = np.random.default_rng(6)
	corpus["sent_off"] = off
	corpus["tok_id"] = rng.integers(0, 900, size=int(off[-1])).astype(np.int32)
	pos_s = rng.integers(0, 6, size=int(off[-1])).astype(np.int8)
	c, Eb = hip_static_corpus(hip, corpus)
	c.set_token_pos(pos_s)
	w = exp5(int(lens.max()))
	for qi, s in enumerate((3, 59, 20)):
		a = int(off[s])
		q_ids = corpus["tok_id"][a:a + 3 * len_t:3][:len_t].astype(np.int32)
		if len(q_ids) < len_t:
			q_ids = np.concatenate((q_ids, rng.integers(0, 900, size=len_t - len(q_ids)).astype(np.int32)))
		Qb = Eb[q_ids]
		tw = rng.uniform(0.3, 1.0, size=len_t).astype(np.float32)
		q_pos = rng.integers(0, 6, size=len_t).astype(np.int8)
		for loc, ms, gaps in ((0, 0.0, (0.1, 0.1)), (1, -1e9, (w, w)), (2, -1e9, (AFF, AFF))):
			for tagged in (False, True):
				okw = dict(tag_weights=tw, q_pos=q_pos, pos_s=pos_s, pos_mismatch_penalty=0.4, similarity_threshold=0.15) if tagged else {}
				hkw = dict(tag_weights=tw, q_pos=q_pos, pos_mismatch_penalty=0.4, similarity_threshold=0.15) if tagged else {}
				for sub in (0.0, 0.7) if loc == 0 else (0.0,):
					ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=100, sent_off=off, tok_id=corpus["tok_id"], E=Eb, Q=Qb, q_ids=q_ids,
						locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=8, min_score=ms, submatch_weight=sub, **okw)
					got = c.query(Qb, q_token_ids=q_ids, q_normalize=False, locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=8,
						min_score=ms, submatch_weight=sub, **hkw)
					assert_same_results(got.trimmed(), ref)
	c.close()


@pytest.mark.parametrize("len_t", [33, 64])
def test_wide_query_state_beyond_the_lds(hip, oracle, len_t):
	"""17 .. 64 query tokens with general gaps and traceback over slices of up to 512 tokens: H, the step lengths and the flags of
	513 x 65 cells take 233 KB -- the global-state form (round 2 / 3: VK_ERR_UNSUPPORTED, 'exceeds the LDS of a workgroup')"""
	off = document_lengths(3, 50, ((4, 512), (20, 400), (21, 66), (49, 300)))
	c, X, Xb = contextual(hip, off, 64, 4)
	w = exp5(512)
	for qi, sent in enumerate((4, 49)):
		Qb = planted_query(X, off, sent, len_t, 50 + qi)
		for loc, ms in ((0, 0.0), (1, -1e9), (2, -1e9)):
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=64, sent_off=off, X=Xb, Q=Qb, locality=loc,
				gap_s=w, gap_t=w, max_matches=10, min_score=ms, want_all_scores=True, n_threads=8)
			got = c.query(Qb, q_normalize=False, locality=loc, gap_s=w, gap_t=w, max_matches=10, min_score=ms)
			assert_same_results(got.trimmed(), ref)
			np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-4)
	c.close()


def test_document_windows_and_views(hip, oracle):
	"""overlapping windows of 800 tokens, step 300, over one token stream (set_slices), queried from a view of the corpus too (a view
	has workspaces of its own: gap table, scratch)"""
	T, d = 6000, 48
	start = np.arange(0, T - 800 + 1, 300, dtype=np.int64)
	end = start + 800
	X = np.random.default_rng(9).standard_normal((T, d)).astype(np.float32)
	Xb = synth.to_bf16_bits(synth.normalize_rows(X))
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=T, n_sentences=len(start))
	c.append_vectors(Xb, normalize=False)
	c.set_slices(start, end)
	c.finalize()
	v = c.view()
	w = exp5(800)
	rng = np.random.default_rng(10)
	idx = np.sort(rng.choice(np.arange(2000, 2600), size=14, replace=False))
	Qb = synth.to_bf16_bits(synth.normalize_rows(X[idx] + 0.05 * rng.standard_normal((14, d)).astype(np.float32)))
	for h in (c, v):
		for loc, ms, gaps in ((0, 0.0, (w, w)), (1, -1e9, (0.1, 0.1)), (2, -1e9, (AFF, w))):
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=start, sent_end=end, X=Xb, Q=Qb, locality=loc,
				gap_s=gaps[0], gap_t=gaps[1], max_matches=6, min_score=ms)
			got = h.query(Qb, q_normalize=False, locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=6, min_score=ms).trimmed()
			assert_same_results(got, ref)
	v.close()
	c.close()


RWMD_11 = {"nbow": (True, True, True), "bow/fast": (True, False, False), "nbow/fast": (True, False, True)}


@pytest.mark.parametrize("variant", list(RWMD_11))
def test_relaxed_wmd_over_slices_of_65_to_512_tokens(hip, oracle, variant):
	"""no slice beyond 512 tokens: the slices of 65 .. 512 take vk_doc_kernel's streaming arm (round 4; the fused kernel's long pass
	before), the others the fused kernel; contextual and static layout, with tag weights; winners restated on the host as everywhere"""
	docs = ((0, 512), (5, 65), (18, 300), (23, 131), (40, 64), (41, 511), (69, 90))
	off = document_lengths(33, 70, docs)
	flags = RWMD_11[variant]
	c, X, Xb = contextual(hip, off, 96, 34)
	rng = np.random.default_rng(35)
	pos_s = rng.integers(0, 6, size=int(off[-1])).astype(np.int8)
	for qi, (sent, len_t) in enumerate(((18, 9), (41, 16), (3, 5))):
		Qb = planted_query(X, off, sent, len_t, 80 + qi)
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=96, sent_off=off, X=Xb, Q=Qb, algorithm=oracle.ALG_RWMD, rwmd=flags,
			max_matches=12, min_score=0.0, want_all_scores=True)
		got = c.query(Qb, q_normalize=False, algorithm=hip.VK_ALG_RWMD, rwmd=flags, max_matches=12, min_score=0.0)
		assert_same_results(got.trimmed(), ref, check_mapping=False, exact=True)
		np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-5, rtol=0)
	c.set_token_pos(pos_s)
	kw = dict(tag_weights=rng.uniform(0.3, 1.0, size=len(Qb)).astype(np.float32), q_pos=rng.integers(0, 6, size=len(Qb)).astype(np.int8),
		pos_mismatch_penalty=0.4, similarity_threshold=0.15)
	ref_t = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=96, sent_off=off, X=Xb, Q=Qb, algorithm=oracle.ALG_RWMD, rwmd=flags,
		max_matches=12, min_score=0.0, want_all_scores=True, pos_s=pos_s, **kw)
	got_t = c.query(Qb, q_normalize=False, algorithm=hip.VK_ALG_RWMD, rwmd=flags, max_matches=12, min_score=0.0, **kw)
	assert_same_results(got_t.trimmed(), ref_t, check_mapping=False, exact=True)
	np.testing.assert_allclose(c.last_scores(), ref_t["all_scores"], atol=3e-5, rtol=0)
	c.close()
	# static layout: token ids + vocabulary
	corpus = synth.make_static_corpus(70, 1, 40, 700, 100, seed=36)
	corpus["sent_off"] = off
	corpus["tok_id"] = rng.integers(0, 700, size=int(off[-1])).astype(np.int32)
	cs, Eb = hip_static_corpus(hip, corpus)
	for s0, len_t in ((0, 8), (41, 13)):
		a = int(off[s0])
		q_ids = corpus["tok_id"][a:a + 2 * len_t:2][:len_t].astype(np.int32)
		ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=100, sent_off=off, tok_id=corpus["tok_id"], E=Eb, Q=Eb[q_ids], q_ids=q_ids,
			algorithm=oracle.ALG_RWMD, rwmd=flags, max_matches=9, min_score=0.0, want_all_scores=True)
		got = cs.query(Eb[q_ids], q_token_ids=q_ids, q_normalize=False, algorithm=hip.VK_ALG_RWMD, rwmd=flags, max_matches=9, min_score=0.0)
		assert_same_results(got.trimmed(), ref, check_mapping=False, exact=True)
		np.testing.assert_allclose(cs.last_scores(), ref["all_scores"], atol=1e-5, rtol=0)
	cs.close()


@pytest.mark.parametrize("variant", list(RWMD_11))
@pytest.mark.parametrize("len_t", [7, 16, 40])
def test_relaxed_wmd_over_documents(hip, oracle, variant, len_t):
	"""rwmd('nbow') and its 1:1 siblings (vectorian/alignment.py:232-237) over documents of up to 3,000 tokens: the scoring pass
	streams row / column minima (vk_doc_kernel<false, 4, .>; queries of more than 16 tokens: vk_wide_kernel), the winners' canonical similarity rows come back through global memory
	(vk_rows_kernel<., true>) and the host restates their scores in the reference's order of operations -- the oracle's floats"""
	docs = ((0, 513), (5, 1200), (18, 3000), (23, 131), (40, 2049), (41, 512), (69, 900))
	off = document_lengths(31, 70, docs)
	d = 64
	c, X, Xb = contextual(hip, off, d, 32)
	flags = RWMD_11[variant]
	for qi, sent in enumerate((18, 40, 3)):
		Qb = planted_query(X, off, sent, len_t, 40 + qi)
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=off, X=Xb, Q=Qb, algorithm=oracle.ALG_RWMD, rwmd=flags,
			max_matches=12, min_score=0.0, want_all_scores=True)
		got = c.query(Qb, q_normalize=False, algorithm=hip.VK_ALG_RWMD, rwmd=flags, max_matches=12, min_score=0.0)
		assert_same_results(got.trimmed(), ref, check_mapping=False, exact=True)
		np.testing.assert_allclose(got.raw_score[:got.n], ref["raw"], atol=1e-4, rtol=0)
		np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-5, rtol=0)
		if len_t == 16:   # the tag-weighted modifier (TagWeightedSlice wraps any slice, match/instantiate.cpp:173-189)
			rng = np.random.default_rng(70 + qi)
			pos_s = rng.integers(0, 6, size=int(off[-1])).astype(np.int8)
			c.set_token_pos(pos_s)
			kw = dict(tag_weights=rng.uniform(0.3, 1.0, size=len(Qb)).astype(np.float32), q_pos=rng.integers(0, 6, size=len(Qb)).astype(np.int8),
				pos_mismatch_penalty=0.4, similarity_threshold=0.15)
			ref_t = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=off, X=Xb, Q=Qb, algorithm=oracle.ALG_RWMD, rwmd=flags,
				max_matches=12, min_score=0.0, want_all_scores=True, pos_s=pos_s, **kw)
			got_t = c.query(Qb, q_normalize=False, algorithm=hip.VK_ALG_RWMD, rwmd=flags, max_matches=12, min_score=0.0, **kw)
			assert_same_results(got_t.trimmed(), ref_t, check_mapping=False, exact=True)
			# (the scores of the pass over ALL slices come from MFMA sums; under the modifier's similarity threshold one token of 3,000
			# whose cosine sits at the threshold may fall on the other side: 1.2e-5 seen.  The winners above are exact: restated canonically)
			np.testing.assert_allclose(c.last_scores(), ref_t["all_scores"], atol=3e-5, rtol=0)
		# the rows of the winners (what the host states a winner's flow from): the oracle's clipped cosines of the slice's tokens
		top = got.trimmed()
		s0 = int(top["sentence"][0])
		ls = int(off[s0 + 1] - off[s0])
		assert got.sim_rows.shape[1] >= 3000 and not got.sim_rows[0, ls:].any()
	c.close()


def test_relaxed_wmd_over_documents_static_layout(hip, oracle):
	"""token ids + vocabulary table: bags of words keyed by token id (repeated words are one entry)"""
	corpus = synth.make_static_corpus(60, 1, 40, 400, 100, seed=15)
	lens = np.diff(corpus["sent_off"]).copy()
	for pos, ln in ((2, 600), (3, 2500), (30, 65), (59, 1100)):
		lens[pos] = ln
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	rng = np.random.default_rng(16)
	corpus["sent_off"] = off
	corpus["tok_id"] = synth.zipf_ids(int(off[-1]), 400, rng).astype(np.int32)
	pos_s = rng.integers(0, 6, size=int(off[-1])).astype(np.int8)
	c, Eb = hip_static_corpus(hip, corpus)
	c.set_token_pos(pos_s)
	for qi, s in enumerate((3, 59, 20)):
		a = int(off[s])
		q_ids = corpus["tok_id"][a:a + 27:3].astype(np.int32)
		q_ids[4] = q_ids[1]
		Qb = Eb[q_ids]
		tw = rng.uniform(0.3, 1.0, size=len(q_ids)).astype(np.float32)
		q_pos = rng.integers(0, 6, size=len(q_ids)).astype(np.int8)
		for flags in RWMD_11.values():
			ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=100, sent_off=off, tok_id=corpus["tok_id"], E=Eb, Q=Qb, q_ids=q_ids,
				algorithm=oracle.ALG_RWMD, rwmd=flags, max_matches=8, min_score=-1.0, want_all_scores=True)
			got = c.query(Qb, q_token_ids=q_ids, q_normalize=False, algorithm=hip.VK_ALG_RWMD, rwmd=flags, max_matches=8, min_score=-1.0)
			assert_same_results(got.trimmed(), ref, check_mapping=False, exact=True)
			np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-5, rtol=0)
		# (tag weights over the static layout: upstream's distance matrix holds the cells of words that occur on both sides twice,
		# wmd.h:121-133, which the scoring pass resolves in LDS from the (id, tag) keys of both sides -- slices of at most 512 tokens;
		# with q_tags such a query is refused on this corpus)
		if qi == 0:
			tags = rng.integers(0, 9, size=int(off[-1])).astype(np.int8)
			c.set_token_tags(tags)
			with pytest.raises(hip.VkError) as e:
				c.query(Qb, q_token_ids=q_ids, q_normalize=False, algorithm=hip.VK_ALG_RWMD, max_matches=8, tag_weights=tw, q_pos=q_pos,
					q_tags=rng.integers(0, 9, size=len(q_ids)).astype(np.int8))
			assert e.value.status == hip.VK_ERR_UNSUPPORTED
	c.close()
