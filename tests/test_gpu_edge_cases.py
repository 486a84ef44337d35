"""-m gpu: edge cases of the path, HIP (through the C-ABI) against the oracle: empty slices, corpora
smaller than one wave group, k larger than the number of admissible slices, the k > 64 selection
path, exact ties (total order), boundary lengths (64-token sentences, 16-token queries, 1-token
everything), min_score admission."""

import numpy as np
import pytest

from vectorian_amd import synth

from helpers import assert_same_results, hip_contextual_corpus, prep_contextual, prep_query

pytestmark = pytest.mark.gpu

EXP5 = ("table", (1 - 2.0 ** (-np.arange(0, 65) / 5)).astype(np.float32))


def build(hip, Xb, off):
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=Xb.shape[1], n_tokens=Xb.shape[0], n_sentences=len(off) - 1)
	c.append_vectors(Xb, normalize=False)
	c.set_sentences(off)
	c.finalize()
	return c


def both(hip, oracle, Xb, off, Qb, **kw):
	c = build(hip, Xb, off)
	okw = dict(kw)
	gs, gt = okw.pop("gap_s", 0.0), okw.pop("gap_t", 0.0)
	rkw = dict(okw)
	rkw.pop("want_flow", None)
	ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=Xb.shape[1], sent_off=off, X=Xb, Q=Qb, gap_s=gs, gap_t=gt, **rkw)
	got = c.query(Qb, q_normalize=False, gap_s=gs, gap_t=gt, **okw).trimmed()
	c.close()
	return got, ref


def vectors(n, d, seed):
	return synth.to_bf16_bits(synth.normalize_rows(np.random.default_rng(seed).standard_normal((n, d)).astype(np.float32)))


@pytest.mark.parametrize("n_sent", [1, 2, 3, 4, 5, 7])
def test_fewer_sentences_than_a_wave_group(hip, oracle, n_sent):
	lens = np.arange(1, n_sent + 1) * 3
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	Xb, Qb = vectors(int(off[-1]), 48, 1), vectors(4, 48, 2)
	got, ref = both(hip, oracle, Xb, off, Qb, gap_s=0.05, gap_t=0.05, max_matches=10, locality=0)
	assert_same_results(got, ref)


def test_empty_slices_are_skipped(hip, oracle):
	# document.h:160: slices with len_s < 1 never reach the matcher
	lens = np.array([5, 0, 0, 7, 0, 3, 0, 0, 0, 9, 1, 0])
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	Xb, Qb = vectors(int(off[-1]), 64, 3), vectors(3, 64, 4)
	for loc, ms in ((0, 0.0), (1, -100.0), (2, -100.0)):
		got, ref = both(hip, oracle, Xb, off, Qb, gap_s=0.1, gap_t=0.1, max_matches=20, locality=loc, min_score=ms)
		assert_same_results(got, ref)
		assert set(got["sentence"]) <= {0, 3, 5, 9, 10}


def test_k_exceeds_admissible_and_min_score(hip, oracle):
	corpus = synth.make_contextual_corpus(300, 2, 20, 500, 64)
	Xb = synth.to_bf16_bits(synth.normalize_rows(corpus["X"]))
	Qb = prep_query(synth.make_queries(corpus, 1, 5)[0])
	got, ref = both(hip, oracle, Xb, corpus["sent_off"], Qb, gap_s=EXP5, gap_t=EXP5, max_matches=64, min_score=0.3)
	assert len(ref["score"]) < 64 and (ref["score"] > 0.3).all()
	assert_same_results(got, ref)
	got, ref = both(hip, oracle, Xb, corpus["sent_off"], Qb, gap_s=EXP5, gap_t=EXP5, max_matches=5, min_score=2.0)
	assert len(got["score"]) == 0 and len(ref["score"]) == 0


@pytest.mark.parametrize("k", [1, 56, 57, 64, 65, 128, 200, 512, 1024])
def test_selection_paths(hip, oracle, k):
	# k + 8 <= 64 (with traceback: k + 8 slices are selected): wave-streaming selection; beyond: staged bitonic sort (powers of two:
	# later stages merge sorted runs);
	# 9000 slices span several blocks and, for k = 512 / 1024, three stages
	corpus = synth.make_contextual_corpus(9000, 3, 12, 800, 32)
	Xb = synth.to_bf16_bits(synth.normalize_rows(corpus["X"]))
	Qb = prep_query(synth.make_queries(corpus, 1, 4)[0])
	got, ref = both(hip, oracle, Xb, corpus["sent_off"], Qb, gap_s=0.1, gap_t=0.1, max_matches=k, want_flow=(k <= 200))
	assert len(ref["score"]) == min(k, len(ref["score"]))
	assert_same_results(got, ref, check_mapping=(k <= 200))


def test_exact_ties_follow_the_total_order(hip, oracle):
	# identical sentences score identically: order = score desc, then sentence index desc
	base = np.random.default_rng(7).standard_normal((6, 40)).astype(np.float32)
	sent = np.concatenate([base, base, base[:3], base])          # slices: 6, 6, 3, 6 tokens
	X = np.concatenate([sent] * 3)                                 # three copies of that document
	lens = np.array([6, 6, 3, 6] * 3)
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	Xb = synth.to_bf16_bits(synth.normalize_rows(X))
	Qb = synth.to_bf16_bits(synth.normalize_rows(base[1:4]))
	got, ref = both(hip, oracle, Xb, off, Qb, gap_s=0.2, gap_t=0.2, max_matches=7)
	assert list(ref["sentence"]) == [11, 9, 8, 7, 5, 4, 3]
	assert list(got["sentence"]) == list(ref["sentence"])
	assert (got["score"] == ref["score"]).all()


def test_boundary_lengths(hip, oracle):
	# 64-token sentences with a 16-token query, and 1-token sentences with a 1-token query
	X = np.random.default_rng(8).standard_normal((64 * 9, 100)).astype(np.float32)
	off = (np.arange(10) * 64).astype(np.int64)
	Xb = synth.to_bf16_bits(synth.normalize_rows(X))
	Qb = synth.to_bf16_bits(synth.normalize_rows(X[70:86] + 0.1))
	for gaps in ((0.1, 0.1), (EXP5, EXP5), (("affine", 0.3, 0.02), ("affine", 0.1, 0.05))):
		for loc, ms in ((0, 0.0), (1, -100.0), (2, -100.0)):
			got, ref = both(hip, oracle, Xb, off, Qb, gap_s=gaps[0], gap_t=gaps[1], max_matches=9, locality=loc, min_score=ms)
			assert_same_results(got, ref)
	off1 = np.arange(0, 41, dtype=np.int64)
	got, ref = both(hip, oracle, Xb[:40], off1, Xb[5:6], gap_s=0.1, gap_t=0.1, max_matches=3)
	assert_same_results(got, ref)
	assert got["sentence"][0] == 5 and abs(got["score"][0] - 1.0) < 1e-3


def test_overlapping_slices(hip, oracle):
	# sliding windows (window_size 3, window_step 1) over 40 sentences: vk_corpus_set_slices
	rng = np.random.default_rng(12)
	lens = rng.integers(2, 15, size=40)
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	start = off[:-1]
	end = off[np.minimum(np.arange(40) + 3, 40)]
	Xb, Qb = vectors(int(off[-1]), 64, 13), vectors(6, 64, 14)
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=64, n_tokens=Xb.shape[0], n_sentences=40)
	c.append_vectors(Xb, normalize=False)
	c.set_slices(start, end)
	c.finalize()
	for loc, ms, gaps in ((0, 0.0, (0.1, 0.1)), (1, -100.0, (EXP5, EXP5)), (2, -100.0, (("affine", 0.2, 0.05),) * 2)):
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=64, sent_off=start, sent_end=end, X=Xb, Q=Qb, locality=loc,
			gap_s=gaps[0], gap_t=gaps[1], max_matches=12, min_score=ms)
		got = c.query(Qb, q_normalize=False, locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=12, min_score=ms).trimmed()
		assert_same_results(got, ref)
	ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=64, sent_off=start, sent_end=end, X=Xb, Q=Qb, algorithm=oracle.ALG_RWMD, max_matches=12)
	got = c.query(Qb, q_normalize=False, algorithm=hip.VK_ALG_RWMD, max_matches=12).trimmed()
	assert_same_results(got, ref, check_mapping=False, exact=True)
	c.close()


def test_views_serve_queries_from_two_threads(hip):
	"""vk_corpus_view: a second handle on the same resident corpus; two host threads, results as from one handle"""
	import threading
	corpus = synth.make_contextual_corpus(5000, 2, 40, 800, 64)
	X = corpus["X"]
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=64, n_tokens=X.shape[0], n_sentences=5000)
	c.append_vectors(X)
	c.set_sentences(corpus["sent_off"])
	c.finalize()
	v = c.view()
	qs = [q["vectors"] for q in synth.make_queries(corpus, 12, 6)]
	kw = dict(gap_s=EXP5, gap_t=EXP5, max_matches=7)
	ref = [c.query(q, **kw).trimmed() for q in qs]
	out = [None] * len(qs)

	def work(h, idx):
		for i in idx:
			out[i] = h.query(qs[i], **kw).trimmed()
	threads = [threading.Thread(target=work, args=(c, range(0, 12, 2))), threading.Thread(target=work, args=(v, range(1, 12, 2)))]
	for t in threads:
		t.start()
	for t in threads:
		t.join()
	for a, b in zip(ref, out):
		assert (a["sentence"] == b["sentence"]).all() and (a["score"] == b["score"]).all() and (a["mapping"] == b["mapping"]).all()
	with pytest.raises(hip.VkError):
		v.view()   # views are taken from the owning handle
	v.close()
	assert c.query(qs[0], **kw).n == 7   # the owner keeps working after its view is gone
	c.close()


@pytest.mark.parametrize("d,n", [(300, 16 * 4 * 4 * 3 + 5), (768, 4099), (384, 63), (1024, 16 * 4 * 4), (100, 1000), (300, 1)])
def test_span_kernel_every_slice(hip, oracle, d, n):
	"""one vector per slice against a one-token query (vk_span_kernel: runs of four tiles per wave, a last partial run, a last
	partial tile): every slice's score is the clipped cosine of the bf16-rounded unit rows, the result set their top k"""
	rng = np.random.default_rng(d + n)
	V = rng.standard_normal((n, d)).astype(np.float32)
	V[n // 2] = -V[0]                # a negative cosine: clipped to 0
	Vb, _ = oracle.normalize_rows_bf16(V)
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=n, n_sentences=n)
	c.append_vectors(Vb, normalize=False)
	c.set_sentences(np.arange(n + 1, dtype=np.int64))
	c.finalize()
	q = V[0:1] + 0.1 * rng.standard_normal((1, d)).astype(np.float32)
	Qb, _ = oracle.normalize_rows_bf16(q)
	got = c.query(Qb, q_normalize=False, max_matches=min(n, 20), min_score=-1.0)
	cos = oracle.sim_bf16(Vb, Qb)[:, 0]
	np.testing.assert_allclose(c.last_scores(), cos, atol=2e-6)
	order = np.argsort(-cos, kind="stable")[:got.n]
	np.testing.assert_allclose(got.score[:got.n], cos[order], atol=2e-6)
	assert got.n == min(n, 20) and int(got.sentence[0]) == 0
	# the aligner scores of the winners, with and without traceback, with and without a booster (the kernel writes its second
	# output array only in the last case)
	boost = (0.5 + rng.random(n)).astype(np.float32)
	for want_flow in (True, False):
		for b in (None, boost):
			r = c.query(Qb, q_normalize=False, max_matches=min(n, 20), min_score=-1.0, want_flow=want_flow, boost=b)
			sel = r.sentence[:r.n]
			np.testing.assert_allclose(r.raw_score[:r.n], cos[sel], atol=2e-6)
			np.testing.assert_allclose(r.score[:r.n], cos[sel] * (1.0 if b is None else b[sel]), atol=2e-6)
	c.close()


def test_different_corpora_from_different_threads(hip):
	"""the boundary's threading contract: concurrent calls on DIFFERENT corpus handles are safe -- three corpora of
	different layout / width / algorithm, one host thread each, every result as from the same handle used alone"""
	import threading
	ctx = synth.make_contextual_corpus(4000, 2, 40, 800, 96, noise=0.3, norm_sigma=0.25)
	a = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=96, n_tokens=ctx["X"].shape[0], n_sentences=4000, keep_magnitudes=True)
	a.append_vectors(ctx["X"], normalize=True)
	a.set_sentences(ctx["sent_off"])
	a.finalize()
	st = synth.make_static_corpus(6000, 1, 40, 500, 300, seed=5)
	b = hip.Corpus(layout=hip.VK_LAYOUT_STATIC, d=300, n_tokens=len(st["tok_id"]), n_sentences=6000, vocab_size=500)
	b.append_vectors(st["E"], normalize=True)
	b.set_token_ids(st["tok_id"])
	b.set_sentences(st["sent_off"])
	b.finalize()
	f32 = synth.make_contextual_corpus(3000, 8, 32, 800, 160)
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=160, n_tokens=f32["X"].shape[0], n_sentences=3000, precision="f32")
	c.append_vectors(f32["X"], normalize=True)
	c.set_sentences(f32["sent_off"])
	c.finalize()
	rng = np.random.default_rng(3)
	qa = [q["vectors"] for q in synth.make_queries(ctx, 10, 7)]
	qb = [rng.integers(0, 500, size=9).astype(np.int32) for _ in range(10)]
	qc = [q["vectors"] for q in synth.make_queries(f32, 10, 20)]    # the long-query kernel
	jobs = [
		(a, [lambda h, q=q: h.query(q, algorithm=hip.VK_ALG_WRD, q_normalize=True, max_matches=6).trimmed() for q in qa]),
		(b, [lambda h, q=q: h.query(st["E"][q], q_token_ids=q, q_normalize=True, gap_s=EXP5, gap_t=EXP5, max_matches=9).trimmed() for q in qb]),
		(c, [lambda h, q=q: h.query(q, q_normalize=True, gap_s=0.1, gap_t=0.1, locality=hip.Locality.SEMIGLOBAL, max_matches=5, min_score=-10.0).trimmed() for q in qc]),
	]
	ref = [[f(h) for f in fs] for h, fs in jobs]
	out = [[None] * len(fs) for _, fs in jobs]
	errors = []

	def work(k):
		h, fs = jobs[k]
		try:
			for rep in range(3):
				for i, f in enumerate(fs):
					out[k][i] = f(h)
		except Exception as e:   # surfaced below: a thread's exception would otherwise vanish
			errors.append(e)
	threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
	for t in threads:
		t.start()
	for t in threads:
		t.join()
	assert not errors, errors
	for rs, os_ in zip(ref, out):
		for r, o in zip(rs, os_):
			assert (r["sentence"] == o["sentence"]).all() and (r["score"] == o["score"]).all() and (r["mapping"] == o["mapping"]).all()
	for h in (a, b, c):
		h.close()


def test_degenerate_vectors(hip, oracle):
	"""Vectors.normalized (vectorian/embedding/vectors.py:71-80): rows / norm with NaN -> 0 -- an all-zero row, a row with a NaN and a
	row with an infinity become zero rows (similarity 0 to everything); a zero query token likewise.  Scores stay finite and equal
	the oracle's on the same rounded rows."""
	corpus = synth.make_contextual_corpus(300, 2, 30, 500, 64)
	X = corpus["X"].copy()
	off = corpus["sent_off"]
	X[int(off[5])] = 0.0
	X[int(off[9]) + 1, 3] = np.nan
	X[int(off[20]), 7] = np.inf
	Xb, _ = oracle.normalize_rows_bf16(X)
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=64, n_tokens=X.shape[0], n_sentences=300)
	c.append_vectors(X, normalize=True)
	c.set_sentences(off)
	c.finalize()
	q = synth.make_queries(corpus, 1, 6)[0]["vectors"].copy()
	q[2] = 0.0
	Qb, _ = oracle.normalize_rows_bf16(q)
	for kw, okw in ((dict(gap_s=EXP5, gap_t=EXP5), dict(gap_s=EXP5, gap_t=EXP5)), (dict(algorithm=hip.VK_ALG_RWMD), dict(algorithm=oracle.ALG_RWMD))):
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=64, sent_off=off, X=Xb, Q=Qb, max_matches=300, min_score=-1.0, want_all_scores=True, **okw)
		got = c.query(q, q_normalize=True, max_matches=300, min_score=-1.0, **kw)
		sc = c.last_scores()
		assert np.isfinite(sc).all()
		np.testing.assert_allclose(sc, ref["all_scores"], atol=1e-4)
		assert got.n == len(ref["score"])
	c.close()


@pytest.mark.parametrize("seed", range(6))
def test_near_ties_at_the_k_boundary_are_resolved_in_the_oracles_arithmetic(hip, oracle, seed):
	"""Thirteen near-duplicates of one slice -- each differs from the original in ONE bf16 ulp of one feature of one token, so their
	scores sit within ~1e-6 of each other, closer than the scoring pass's MFMA cosines can order them -- straddle the k-th place (k =
	10) of a corpus whose other slices score far below.  The selection hands k + 8 slices to the traceback kernel, which restates them
	in the oracle's arithmetic; the result set (ids, scores, order, tracebacks) must be the oracle's, bit for bit.  (With more than
	8 such slices beyond the k-th place the guarantee ends: DESIGN 7.1.)"""
	rng = np.random.default_rng(900 + seed)
	d, len_s, len_t, n = 300, 24, 8, 400
	X = rng.standard_normal((n * len_s, d)).astype(np.float32)
	base = int(rng.integers(0, n))
	Xb = synth.to_bf16_bits(synth.normalize_rows(X))
	dup = rng.choice(np.setdiff1d(np.arange(n), [base]), size=12, replace=False)
	for j, s in enumerate(dup):
		Xb[s * len_s:(s + 1) * len_s] = Xb[base * len_s:(base + 1) * len_s]
		tok, feat = int(rng.integers(3, 3 + len_t)), int(rng.integers(0, d))   # inside the aligned stretch
		v = int(Xb[s * len_s + tok, feat])
		if 0 < (v & 0x7fff) < 0x7f00:
			Xb[s * len_s + tok, feat] = np.uint16(v + 1 if j % 2 == 0 else v - 1)
	off = np.arange(n + 1, dtype=np.int64) * len_s
	Qb = Xb[base * len_s + 3:base * len_s + 3 + len_t].copy()
	for loc, ms, gaps in ((0, 0.0, (0.1, 0.1)), (0, 0.0, (EXP5, EXP5)), (1, -1e9, (0.05, 0.05))):
		got, ref = both(hip, oracle, Xb, off, Qb, gap_s=gaps[0], gap_t=gaps[1], max_matches=10, locality=loc, min_score=ms)
		assert set(int(x) for x in ref["sentence"]) <= set(int(x) for x in dup) | {base}
		assert np.ptp(ref["score"]) < 1e-4 and len(np.unique(ref["score"])) > 1      # near-ties, not exact ties
		assert_same_results(got, ref)


@pytest.mark.parametrize("layout", ["contextual", "static"])
def test_result_sets_beyond_1024_matches(hip, oracle, layout):
	"""max_matches > VK_MAX_MATCHES (refused until round 4): upstream's ResultSet is bounded by max_matches alone
	(vectorian/core/cpp/result_set.h:32-68).  Alignments: every score sorted on the device (hipcub), the k + 8 best retraced --
	slice ids, scores and tracebacks equal to the oracle's bit for bit, with and without flows; a result set larger than the corpus
	returns every admitted slice; the transports refuse"""
	n = 5000
	if layout == "static":
		corpus = synth.make_static_corpus(n, 0, 30, 500, 64)
		from helpers import hip_static_corpus
		c, Eb = hip_static_corpus(hip, corpus)
		q = synth.make_queries(corpus, 1, 6)[0]
		Qb = synth.to_bf16_bits(synth.normalize_rows(q["vectors"]))
		ckw, okw = dict(q_token_ids=q["ids"]), dict(layout=oracle.LAYOUT_STATIC, d=64, sent_off=corpus["sent_off"], tok_id=corpus["tok_id"], E=Eb, q_ids=q["ids"])
	else:
		corpus = synth.make_contextual_corpus(n, 0, 30, 500, 64)
		Xb = prep_contextual(corpus)
		c = hip_contextual_corpus(hip, corpus, Xb)
		Qb = prep_query(synth.make_queries(corpus, 1, 6)[0])
		ckw, okw = {}, dict(layout=oracle.LAYOUT_CONTEXTUAL, d=64, sent_off=corpus["sent_off"], X=Xb)
	for k, min_score in ((1500, 0.0), (3000, 0.15), (8000, -1.0)):
		ref = oracle.find(Q=Qb, locality=oracle.LOCAL, gap_s=0.1, gap_t=0.1, max_matches=k, min_score=min_score, **okw)
		got = c.query(Qb, locality=hip.Locality.LOCAL, gap_s=0.1, gap_t=0.1, q_normalize=False, max_matches=k, min_score=min_score, **ckw)
		assert got.n == len(ref["score"]) and (got.n > 1024 or min_score > 0.1)
		assert_same_results(got.trimmed(), ref)
		noflow = c.query(Qb, locality=hip.Locality.LOCAL, gap_s=0.1, gap_t=0.1, q_normalize=False, max_matches=k, min_score=min_score, want_flow=False, **ckw)
		assert_same_results(noflow.trimmed(), ref, check_mapping=False, score_tol=1e-4)
		np.testing.assert_allclose(noflow.raw_score[:noflow.n], ref["raw"], atol=1e-4)
	with pytest.raises(hip.VkError) as e:
		c.query(Qb, algorithm=hip.VK_ALG_RWMD, q_normalize=False, max_matches=2000, **ckw)
	assert e.value.status == hip.VK_ERR_UNSUPPORTED
	c.close()
