"""-m gpu: queries of 17 .. 64 tokens run on vk_wide_kernel (one wave per slice, lane = query column).
HIP (through the C-ABI) against the oracle: scores, order, mappings, edge similarities."""

import numpy as np
import pytest

from vectorian_amd import synth

from helpers import assert_same_results, hip_contextual_corpus, hip_static_corpus, prep_contextual, prep_query

pytestmark = pytest.mark.gpu

EXP5L = ("table", (1 - 2.0 ** (-np.arange(0, 513) / 5)).astype(np.float32))
AFF = ("affine", 0.2, 0.05)
CASES = ((0, 0.0, (0.1, 0.1)), (0, 0.0, (EXP5L, EXP5L)), (1, -1e9, (EXP5L, EXP5L)), (2, -1e9, (AFF, AFF)), (1, -1e9, (0.05, 0.2)),
	(2, -1e9, (EXP5L, 0.1)), (0, 0.0, (AFF, AFF)))


@pytest.mark.parametrize("d,len_t", [(64, 17), (300, 32), (300, 20), (768, 40), (128, 64)])
def test_wide_query_contextual(hip, oracle, d, len_t):
	corpus = synth.make_contextual_corpus(300, 1, 64, 1500, d)
	Xb = prep_contextual(corpus)
	c = hip_contextual_corpus(hip, corpus, Xb)
	qs = [prep_query(q) for q in synth.make_queries(corpus, 3, len_t)]
	boost = np.random.default_rng(3).uniform(0.5, 1.5, size=300).astype(np.float32)
	for qi, Qb in enumerate(qs):
		for ci, (loc, ms, gaps) in enumerate(CASES):
			bst = boost if (qi + ci) % 3 == 0 else None
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=corpus["sent_off"], X=Xb, Q=Qb, locality=loc,
				gap_s=gaps[0], gap_t=gaps[1], max_matches=12, min_score=ms, boost=bst, want_all_scores=True)
			got = c.query(Qb, q_normalize=False, locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=12, min_score=ms, boost=bst)
			assert_same_results(got.trimmed(), ref)
			np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-4)
			# edge similarities of the matched pairs
			off = corpus["sent_off"]
			for i in range(got.n):
				s = int(got.sentence[i])
				S = oracle.sim_bf16(Xb[off[s]:off[s + 1]], Qb)
				for j in range(len_t):
					m = int(got.mapping[i, j])
					if m >= 0:
						assert abs(got.edge_sim[i, j] - S[m, j]) < 2e-6
	Qb = qs[0]
	for flags in ((True, True, True), (True, False, True), (True, False, False)):
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=corpus["sent_off"], X=Xb, Q=Qb, algorithm=oracle.ALG_RWMD, rwmd=flags, max_matches=12)
		got = c.query(Qb, q_normalize=False, algorithm=hip.VK_ALG_RWMD, rwmd=flags, max_matches=12).trimmed()
		assert_same_results(got, ref, check_mapping=False, score_tol=2e-5, tie_tol=2e-5)
	c.close()


def test_wide_query_static_and_long_slices(hip, oracle):
	corpus = synth.make_static_corpus(150, 1, 40, 700, 100, seed=8)
	lens = np.diff(corpus["sent_off"]).copy()
	lens[[4, 77, 149]] = (150, 70, 256)
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	corpus["sent_off"] = off
	corpus["tok_id"] = np.random.default_rng(9).integers(0, 700, size=int(off[-1])).astype(np.int32)
	c, Eb = hip_static_corpus(hip, corpus)
	for s, len_t in ((4, 33), (149, 24), (20, 18)):
		q_ids = corpus["tok_id"][off[s]:off[s] + len_t].astype(np.int32)
		if len(q_ids) < len_t:
			q_ids = np.concatenate((q_ids, np.random.default_rng(s).integers(0, 700, size=len_t - len(q_ids)).astype(np.int32)))
		Qb = Eb[q_ids]
		for loc, ms, gaps in CASES[:4]:
			ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=100, sent_off=off, tok_id=corpus["tok_id"], E=Eb, Q=Qb, q_ids=q_ids,
				locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=10, min_score=ms)
			got = c.query(Qb, q_token_ids=q_ids, q_normalize=False, locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=10, min_score=ms)
			assert_same_results(got.trimmed(), ref)
	c.close()


@pytest.mark.parametrize("len_t", [33, 40, 44, 48, 64])
def test_four_block_kernel_general_gaps(hip, oracle, len_t):
	"""33..64 query tokens with general gaps: vk_score32_kernel<3 or 6, ., 4>; 33..48 tokens: the three-block balance of the far
	candidates (dp32_general<.., B3>: block 3 holds no column, its lanes and block 0's help blocks 1 and 2)"""
	d = 128
	corpus = synth.make_contextual_corpus(401, 1, 32 if len_t not in (44, 48) else 64, 900, d)   # 44, 48: slices up to 64 tokens (64-row history)
	Xb = prep_contextual(corpus)
	c = hip_contextual_corpus(hip, corpus, Xb)
	steep = ("table", np.minimum(0.02 * np.arange(0, 65) ** 2, 3.0).astype(np.float32))   # not subadditive: chains of gaps are cheaper (the closure of w_t)
	for Qb in [prep_query(q) for q in synth.make_queries(corpus, 2, len_t)]:
		for loc, ms, gap in ((0, 0.0, EXP5L), (1, -1e9, EXP5L), (2, -1e9, EXP5L), (0, 0.0, steep), (1, -1e9, steep)):
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=corpus["sent_off"], X=Xb, Q=Qb, locality=loc,
				gap_s=gap, gap_t=gap, max_matches=10, min_score=ms, want_all_scores=True)
			got = c.query(Qb, q_normalize=False, locality=loc, gap_s=gap, gap_t=gap, max_matches=10, min_score=ms)
			assert_same_results(got.trimmed(), ref)
			np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-4)
	c.close()


def test_wide_query_limits(hip):
	corpus = synth.make_contextual_corpus(20, 4, 20, 100, 32, norm_sigma=0.2)
	c = hip_contextual_corpus(hip, corpus, keep_magnitudes=True)
	Q65 = np.random.default_rng(1).standard_normal((65, 32)).astype(np.float32)
	assert c.query(Q65, max_matches=3).n == 3            # alignments: up to 512 query tokens since round 4 (vk_longq_kernel)
	with pytest.raises(hip.VkError):
		c.query(Q65, algorithm=hip.VK_ALG_WRD, max_matches=3)   # the transports stop at 64
	assert c.query(Q65[:20], algorithm=hip.VK_ALG_WRD, max_matches=3).n == 3   # exact transport runs up to 64 query tokens
	assert c.query(Q65[:20], algorithm=hip.VK_ALG_RWMD, rwmd=(False, False, True), max_matches=3, min_score=-1.0).n == 3   # ... and the 1:n form of RWMD
	c.close()


@pytest.mark.parametrize("len_t", [17, 19, 23, 28, 31, 32, 33, 40, 47, 49, 64])
def test_two_block_kernel_shapes(hip, oracle, len_t):
	"""vk_score32_kernel (17..32 query tokens, linear / affine gaps, two slices per wave): odd slice counts, empty and
	one-token slices, overlapping windows, every locality; scores of ALL slices against the oracle"""
	rng = np.random.default_rng(len_t)
	d = int(rng.choice([48, 300, 320]))
	n = 2 * int(rng.integers(120, 200)) + 1                       # odd: the last pair has one slice
	corpus = synth.make_contextual_corpus(n, 1, 64, 1200, d)
	Xb = prep_contextual(corpus)
	qs = [prep_query(q) for q in synth.make_queries(corpus, 2, len_t)]
	# (a) the sentence partition with some empty sentences, (b) windows of three sentences (capped at 64 tokens)
	at = np.arange(5, n, 16)
	at = at[:len(at) // 2 * 2]                                     # an even number of empty sentences: n stays odd
	off = np.insert(corpus["sent_off"], at, corpus["sent_off"][at])
	n = len(off) - 1
	assert n % 2 == 1 and (np.diff(off) == 0).sum() > 5
	start_w = off[:-1].copy()
	end_w = np.minimum(off[np.minimum(np.arange(n) + 3, n)], start_w + 64)
	for windows in (False, True):
		c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=Xb.shape[0], n_sentences=n)
		c.append_vectors(Xb, normalize=False)
		if windows:
			c.set_slices(start_w, end_w)
		else:
			c.set_sentences(off)
		c.finalize()
		geo = dict(sent_off=start_w, sent_end=end_w) if windows else dict(sent_off=off)
		for Qb in qs:
			for loc, ms, gaps in ((0, 0.0, (0.1, 0.1)), (1, -1e9, (0.05, 0.2)), (2, -1e9, (AFF, AFF)), (0, 0.0, (AFF, 0.15)), (1, -1e9, (0.3, AFF)),
					(0, 0.0, (EXP5L, EXP5L)), (1, -1e9, (EXP5L, EXP5L)), (2, -1e9, (0.1, EXP5L))):
				ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, X=Xb, Q=Qb, locality=loc, gap_s=gaps[0], gap_t=gaps[1],
					max_matches=9, min_score=ms, want_all_scores=True, **geo)
				got = c.query(Qb, q_normalize=False, locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=9, min_score=ms)
				assert_same_results(got.trimmed(), ref)
				lens = (end_w - start_w) if windows else np.diff(off)
				every = c.last_scores()
				np.testing.assert_allclose(every[lens > 0], ref["all_scores"][lens > 0], atol=1e-4)
				assert np.isneginf(every[lens == 0]).all()
			for flags in ((True, True, True), (True, False, False), (True, False, True)):
				ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, X=Xb, Q=Qb, algorithm=oracle.ALG_RWMD, rwmd=flags, max_matches=9,
					want_all_scores=True, **geo)
				got = c.query(Qb, q_normalize=False, algorithm=hip.VK_ALG_RWMD, rwmd=flags, max_matches=9)
				assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=2e-5, tie_tol=2e-5)
				np.testing.assert_allclose(c.last_scores()[lens > 0], ref["all_scores"][lens > 0], atol=2e-5)
		c.close()


@pytest.mark.parametrize("len_t", [17, 24, 32, 36, 48, 61])
def test_two_block_kernel_static_layout(hip, oracle, len_t):
	"""the static layout (token ids + two per-query tables) on vk_score32_kernel; general gaps keep to vk_wide_kernel"""
	corpus = synth.make_static_corpus(701, 1, 64, 900, 100, seed=40 + len_t)
	off, ids = corpus["sent_off"], corpus["tok_id"]
	c, Eb = hip_static_corpus(hip, corpus)
	rng = np.random.default_rng(len_t)
	for rep in range(2):
		s = int(rng.integers(0, 700))
		q_ids = ids[off[s]:off[s] + len_t].astype(np.int32)
		if len(q_ids) < len_t:
			q_ids = np.concatenate((q_ids, rng.integers(0, 900, size=len_t - len(q_ids)).astype(np.int32)))
		Qb = Eb[q_ids]
		for loc, ms, gaps in ((0, 0.0, (AFF, AFF)), (1, -1e9, (AFF, 0.1)), (2, -1e9, (0.2, AFF)), (0, 0.0, (EXP5L, EXP5L))):
			ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=100, sent_off=off, tok_id=ids, E=Eb, Q=Qb, q_ids=q_ids,
				locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=10, min_score=ms, want_all_scores=True)
			got = c.query(Qb, q_token_ids=q_ids, q_normalize=False, locality=loc, gap_s=gaps[0], gap_t=gaps[1], max_matches=10, min_score=ms)
			assert_same_results(got.trimmed(), ref)   # repeated words, co-optimal tracebacks: the winners are restated canonically
			np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=1e-4)
		for flags in ((True, True, True), (True, False, False)):
			ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=100, sent_off=off, tok_id=ids, E=Eb, Q=Qb, q_ids=q_ids, algorithm=oracle.ALG_RWMD,
				rwmd=flags, max_matches=10, want_all_scores=True)
			got = c.query(Qb, q_token_ids=q_ids, q_normalize=False, algorithm=hip.VK_ALG_RWMD, rwmd=flags, max_matches=10)
			assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=2e-5, tie_tol=2e-5)
			np.testing.assert_allclose(c.last_scores(), ref["all_scores"], atol=2e-5)
	c.close()


LONG_GAPS = {"linear": (0.1, 0.05), "affine": (AFF, ("affine", 0.1, 0.02)), "exp5": (EXP5L, EXP5L)}


@pytest.mark.parametrize("len_t", [65, 100, 200, 512])
@pytest.mark.parametrize("gap", ["linear", "affine", "exp5"])
def test_long_query_contextual(hip, oracle, len_t, gap):
	"""queries of 65 .. 512 tokens (round 4: VK_ERR_UNSUPPORTED until then; upstream's only bound is the int16 of a mapping,
	vectorian/core/cpp/metric/alignment.h:357-358): vk_longq_kernel -- one wave per slice, lane = slice token, anti-diagonal sweep over
	the query's tokens; the k + 8 best retraced in the canonical arithmetic with the oracle's order of candidates.  Slice ids, scores
	and tracebacks equal to the oracle's bit for bit, every locality; the score vector of all slices within 1e-4"""
	d = 64 if len_t >= 200 else 300
	n = 90 if (gap == "exp5" and len_t == 512) else 260
	corpus = synth.make_contextual_corpus(n, 0, 64, 600, d)
	Xb = prep_contextual(corpus)
	c = hip_contextual_corpus(hip, corpus, Xb)
	rng = np.random.default_rng(len_t)
	# a query that quotes a passage of the corpus (several consecutive slices) with noise, so that long alignments exist
	off = corpus["sent_off"]
	s0 = 40
	toks = np.arange(off[s0], off[s0] + len_t) % off[-1]
	Q = corpus["X"][toks] + 0.3 * rng.standard_normal((len_t, d)).astype(np.float32)
	Qb = synth.to_bf16_bits(synth.normalize_rows(Q))
	gs, gt = LONG_GAPS[gap]
	boost = rng.uniform(0.5, 1.5, size=n).astype(np.float32)
	for loc, ms in ((0, 0.0), (1, -1e9), (2, -1e9)):
		bst = boost if loc == 2 else None
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=off, X=Xb, Q=Qb, locality=loc, gap_s=gs, gap_t=gt, max_matches=9, min_score=ms,
			boost=bst, want_all_scores=True)
		got = c.query(Qb, q_normalize=False, locality=loc, gap_s=gs, gap_t=gt, max_matches=9, min_score=ms, boost=bst)
		assert_same_results(got.trimmed(), ref)
		full = np.diff(off) > 0      # (empty slices carry no score: -inf)
		np.testing.assert_allclose(c.last_scores()[full], np.asarray(ref["all_scores"])[full], atol=1e-4)
		assert np.isneginf(c.last_scores()[~full]).all()
		for i in range(got.n):   # edge similarities of the matched pairs
			s = int(got.sentence[i])
			S = oracle.sim_bf16(Xb[off[s]:off[s + 1]], Qb)
			for j in np.nonzero(got.mapping[i] >= 0)[0]:
				assert abs(got.edge_sim[i, j] - S[int(got.mapping[i, j]), j]) < 2e-6
		noflow = c.query(Qb, q_normalize=False, locality=loc, gap_s=gs, gap_t=gt, max_matches=9, min_score=ms, boost=bst, want_flow=False)
		assert_same_results(noflow.trimmed(), ref, check_mapping=False, score_tol=1e-4, tie_tol=1e-4)
	c.close()


@pytest.mark.parametrize("len_t,gap", [(80, "linear"), (130, "exp5"), (300, "affine")])
def test_long_query_static_with_tag_weights_and_views(hip, oracle, len_t, gap):
	"""the static layout (per-tile tables gathered by token id, sim[id(t_j)][j] = 1), the tag-weighted modifier, a view beside its owner,
	only_slices (the debug hook's walk), and the refusals: transports, long slices, a submatch weight"""
	corpus = synth.make_static_corpus(300, 1, 50, 400, 96, seed=5)
	off = corpus["sent_off"]
	n_tok = int(off[-1])
	Eb = synth.to_bf16_bits(synth.normalize_rows(corpus["E"]))
	rng = np.random.default_rng(17 + len_t)
	pos = rng.integers(1, 6, size=n_tok).astype(np.int8)
	c = hip.Corpus(layout=hip.VK_LAYOUT_STATIC, d=96, n_tokens=n_tok, n_sentences=300, vocab_size=400)
	c.append_vectors(Eb, normalize=False)
	c.set_token_ids(corpus["tok_id"])
	c.set_token_pos(pos)
	c.set_sentences(off)
	c.finalize()
	q_ids = np.concatenate([corpus["tok_id"][off[s]:off[s + 1]] for s in range(60, 75)])[:len_t].astype(np.int32)
	if len(q_ids) < len_t:
		q_ids = np.concatenate((q_ids, rng.integers(0, 400, size=len_t - len(q_ids)).astype(np.int32)))
	q_ids[5] = -1                                # a word the vocabulary does not hold
	Qb = Eb[np.where(q_ids >= 0, q_ids, 3)]
	gs, gt = LONG_GAPS[gap]
	tw = rng.uniform(0.5, 1.5, size=len_t).astype(np.float32)
	q_pos = rng.integers(1, 6, size=len_t).astype(np.int8)
	v = c.view()
	for loc, ms, tagged in ((0, 0.0, False), (1, -1e9, True), (2, -1e9, True), (0, 0.05, True)):
		okw = dict(tag_weights=tw, q_pos=q_pos, pos_s=pos, pos_mismatch_penalty=0.3, similarity_threshold=0.1) if tagged else {}
		ckw = dict(tag_weights=tw, q_pos=q_pos, pos_mismatch_penalty=0.3, similarity_threshold=0.1) if tagged else {}
		ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=96, sent_off=off, tok_id=corpus["tok_id"], E=Eb, Q=Qb, q_ids=q_ids, locality=loc,
			gap_s=gs, gap_t=gt, max_matches=7, min_score=ms, **okw)
		for h in (c, v):
			got = h.query(Qb, q_token_ids=q_ids, q_normalize=False, locality=loc, gap_s=gs, gap_t=gt, max_matches=7, min_score=ms, **ckw)
			assert_same_results(got.trimmed(), ref)
	# only_slices: exactly these slices, in this order, whatever their score
	ids = np.array([70, 3, 61, 299, 0], dtype=np.int64)
	got = c.query(Qb, q_token_ids=q_ids, q_normalize=False, locality=0, gap_s=gs, gap_t=gt, only_slices=ids)
	full = oracle.find(layout=oracle.LAYOUT_STATIC, d=96, sent_off=off, tok_id=corpus["tok_id"], E=Eb, Q=Qb, q_ids=q_ids, locality=0, gap_s=gs, gap_t=gt,
		max_matches=300, min_score=-1.0)
	by_id = {int(s): i for i, s in enumerate(full["sentence"])}
	assert got.n == 5 and (got.sentence[:5] == ids).all()
	for i, s in enumerate(ids):
		assert got.score[i] == np.float32(full["score"][by_id[int(s)]]) and (got.mapping[i] == full["mapping"][by_id[int(s)]]).all()
	for kw in (dict(algorithm=hip.VK_ALG_RWMD), dict(algorithm=hip.VK_ALG_WRD), dict(submatch_weight=0.5)):
		with pytest.raises(hip.VkError) as e:
			c.query(Qb, q_token_ids=q_ids, q_normalize=False, **kw)
		assert e.value.status in (hip.VK_ERR_UNSUPPORTED, hip.VK_ERR_STATE)
	with pytest.raises(hip.VkError) as e:
		c.query(np.zeros((513, 96), np.float32))
	assert e.value.status == hip.VK_ERR_UNSUPPORTED
	v.close()
	c.close()
