"""-m gpu: BASELINE.json configs[1] at full size -- 1,000,000 sentences x 32 tokens x 300-d, 10-token query -- checked
through properties that do not need the oracle to score a million sentences:
  * a sample of 3,000 sentences (their vectors are kept on the host while the shard is generated on the device) scores
    exactly as the oracle scores them;
  * the returned top-k equals the top-k of the full score vector under the result order (score desc, sentence desc) up to the
    rounding between the scoring pass and the winners' canonical restatement, and every winner is bit-exact against the oracle
    on its regenerated vectors (score and traceback);
  * planted copies of the query come out first with score ~ 1 and the identity traceback;
  * the same query twice gives the same bytes; a boost of c scales every score by c;
  * two half shards merged (vk_merge_topk with offsets) give the result of the whole;
  * queries of 24 and 40 tokens over the same shard: the sample against the oracle, the selection against the score vector."""

import numpy as np
import pytest

# torch generates the shard on the device (as bench.py does).  It must be imported before the HIP library is loaded:
# torch brings its own HIP runtime, and the process can only initialise one.
try:
	import torch
except Exception:   # pragma: no cover
	torch = None

from vectorian_amd import synth

pytestmark = pytest.mark.gpu

N_SENT, LEN_S, D, LEN_T, V = 1_000_000, 32, 300, 10, 50_000
EXP5 = ("table", (1 - 2.0 ** (-np.arange(0, 65) / 5)).astype(np.float32))


CHUNK_S = 1 << 15                                       # sentences per generator chunk


def chunk_vectors(torch, E_dev, ids, a, b, planted):
	"""fp32 vectors of sentences [a, b) (a a multiple of CHUNK_S): seeded per chunk, the same on every call"""
	device = E_dev.device
	gen = torch.Generator(device=device)
	gen.manual_seed(77 + a)                             # per chunk: the same vectors whatever the shard bounds (multiples of the chunk)
	idx = torch.from_numpy(ids[a * LEN_S:b * LEN_S].astype(np.int64)).to(device)
	x = E_dev[idx] + 0.1 * torch.randn(((b - a) * LEN_S, D), device=device, generator=gen, dtype=torch.float32)
	for s, qv in planted.items():                       # exact copies of the query inside chosen sentences
		if a <= s < b:
			x[(s - a) * LEN_S + 5:(s - a) * LEN_S + 5 + LEN_T] = torch.from_numpy(qv).to(device)
	return x.contiguous()


def regenerate(torch, E, ids, sentences, planted, n_total):
	"""fp32 vectors of the given sentences, from the generator (one chunk at a time)"""
	E_dev = torch.from_numpy(E).to(torch.device("cuda", 0))
	out = {}
	for a in sorted({int(s) // CHUNK_S * CHUNK_S for s in sentences}):
		x = chunk_vectors(torch, E_dev, ids, a, min(a + CHUNK_S, n_total), planted)
		for s in sentences:
			if a <= int(s) < a + CHUNK_S:
				out[int(s)] = x[(int(s) - a) * LEN_S:(int(s) - a + 1) * LEN_S].cpu().numpy()
		del x
	del E_dev
	torch.cuda.empty_cache()
	return out


def build(hip, torch, ids, E, lo, hi, sample, planted):
	"""sentences [lo, hi) of the synthetic corpus as one device corpus; returns it and the fp32 vectors of `sample`"""
	device = torch.device("cuda", 0)
	n_tok = (hi - lo) * LEN_S
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=D, n_tokens=n_tok, n_sentences=hi - lo)
	E_dev = torch.from_numpy(E).to(device)
	kept = {}
	for a in range(lo, hi, CHUNK_S):
		b = min(a + CHUNK_S, hi)
		x = chunk_vectors(torch, E_dev, ids, a, b, planted)
		for s in sample[(sample >= a) & (sample < b)]:
			kept[int(s)] = x[(s - a) * LEN_S:(s - a + 1) * LEN_S].cpu().numpy()
		torch.cuda.synchronize()
		c.append_vectors_device(x.data_ptr(), x.shape[0], hip.VK_F32, normalize=True)
		del x
	c.set_sentences(np.arange(hi - lo + 1, dtype=np.int64) * LEN_S)
	c.finalize()
	del E_dev
	torch.cuda.empty_cache()
	return c, kept


def check_selection(top, scores, k):
	"""The result set against the score vector of ALL slices.  The vector holds the scoring pass's values (MFMA cosines); the
	winners' scores are restated in the oracle's arithmetic (sim_canon) and a few runners-up with them, so the two differ by
	rounding (<= 4e-6): the winners are in result order, each within rounding of its entry in the vector, and no slice
	outside the result set scores above the k-th winner by more than rounding."""
	n = top.n
	sent, sc = top.sentence[:n].astype(np.int64), top.score[:n]
	assert len(set(sent.tolist())) == n
	np.testing.assert_allclose(sc, scores[sent], atol=4e-6, rtol=0)
	for i in range(n - 1):
		assert sc[i] > sc[i + 1] or (sc[i] == sc[i + 1] and sent[i] > sent[i + 1])
	if n == k:
		rest = scores.copy()
		rest[sent] = -np.inf
		assert rest.max() <= sc[n - 1] + 4e-6


def check_winners_exact(oracle, top, vectors, qv, **kw):
	"""every winner alone against the oracle on its regenerated vectors: score and traceback bit for bit"""
	Qb, _ = oracle.normalize_rows_bf16(qv)
	for i in range(top.n):
		xv = vectors[int(top.sentence[i])]
		xb, _ = oracle.normalize_rows_bf16(xv)
		one = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=xv.shape[1], sent_off=np.array([0, len(xv)], dtype=np.int64), X=xb, Q=Qb,
			max_matches=1, **kw)
		assert len(one["score"]) == 1
		assert np.float32(one["score"][0]).view(np.uint32) == np.float32(top.score[i]).view(np.uint32), (i, one["score"][0], top.score[i])
		assert list(one["mapping"][0]) == list(top.mapping[i])


def test_config2_full_size_properties(hip, oracle):
	if torch is None:
		pytest.skip("torch is needed to generate the shard on the device")
	rng = np.random.default_rng(11)
	E = synth.make_vocab(V, D)
	ids = synth.zipf_ids(N_SENT * LEN_S, V, rng)
	qv = np.ascontiguousarray(E[rng.integers(0, V, size=LEN_T)] + 0.05 * rng.standard_normal((LEN_T, D)).astype(np.float32))
	planted = {int(s): qv for s in (123_456, 999_999, 0)}
	sample = np.unique(np.concatenate((rng.integers(0, N_SENT, size=3000), list(planted))))
	c, kept = build(hip, torch, ids, E, 0, N_SENT, sample, planted)
	kw = dict(q_normalize=True, locality=0, gap_s=EXP5, gap_t=EXP5, max_matches=10, min_score=0.0)

	top = c.query(qv, **kw)
	scores = c.last_scores()
	assert scores.shape == (N_SENT,) and np.isfinite(scores).all()

	# (1) sampled sentences against the oracle
	Xs = np.concatenate([kept[int(s)] for s in sample])
	Xb, _ = oracle.normalize_rows_bf16(Xs)
	Qb, _ = oracle.normalize_rows_bf16(qv)
	ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=D, sent_off=np.arange(len(sample) + 1, dtype=np.int64) * LEN_S, X=Xb, Q=Qb,
		locality=0, gap_s=EXP5, gap_t=EXP5, max_matches=10, min_score=0.0, want_all_scores=True, n_threads=8)
	np.testing.assert_allclose(scores[sample], ref["all_scores"], atol=1e-4, rtol=0)

	# (2) selection: top-k of the full score vector under (score desc, sentence desc), up to the rounding between the scoring
	# pass and the canonical restatement of the winners; (2b) every winner bit-exact against the oracle on its regenerated vectors
	check_selection(top, scores, 10)
	check_winners_exact(oracle, top, regenerate(torch, E, ids, top.sentence[:top.n], planted, N_SENT), qv,
		locality=0, gap_s=EXP5, gap_t=EXP5, min_score=-1.0)

	# (3) planted copies first, score ~ 1, identity traceback at offset 5
	assert set(top.sentence[:3]) == set(planted)
	assert (top.score[:3] > 0.999).all()
	for i in range(3):
		assert list(top.mapping[i]) == list(range(5, 5 + LEN_T))

	# (4) idempotence; boost scales
	again = c.query(qv, **kw)
	np.testing.assert_array_equal(again.score, top.score)
	np.testing.assert_array_equal(again.mapping, top.mapping)
	boosted = c.query(qv, boost=np.full(N_SENT, 0.5, np.float32), **kw)
	np.testing.assert_allclose(boosted.score[:10], 0.5 * top.score[:10], rtol=1e-6)
	assert list(boosted.sentence[:10]) == list(top.sentence[:10])

	# (4b) the same shard under queries of 24 and 40 tokens (vk_score32_kernel, two and four column blocks): sampled
	# sentences against the oracle, and the selection against the full score vector
	for len_t, gaps in ((24, (0.1, 0.1)), (24, (EXP5, EXP5)), (40, (("affine", 0.2, 0.05), 0.1))):
		ql = np.ascontiguousarray(E[rng.integers(0, V, size=len_t)] + 0.05 * rng.standard_normal((len_t, D)).astype(np.float32))
		kwl = dict(q_normalize=True, locality=0, gap_s=gaps[0], gap_t=gaps[1], max_matches=10, min_score=0.0)
		topl = c.query(ql, **kwl)
		sl = c.last_scores()
		Ql, _ = oracle.normalize_rows_bf16(ql)
		refl = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=D, sent_off=np.arange(len(sample) + 1, dtype=np.int64) * LEN_S, X=Xb, Q=Ql,
			locality=0, gap_s=gaps[0], gap_t=gaps[1], max_matches=10, min_score=0.0, want_all_scores=True, n_threads=8)
		np.testing.assert_allclose(sl[sample], refl["all_scores"], atol=1e-4, rtol=0)
		check_selection(topl, sl, 10)
	c.close()

	# (5) two half shards merged == the whole
	half = N_SENT // 2 // (1 << 15) * (1 << 15)
	parts = []
	for lo, hi in ((0, half), (half, N_SENT)):
		ch, _ = build(hip, torch, ids, E, lo, hi, np.zeros(0, np.int64), planted)
		t = ch.query(qv, **kw)
		t.sentence[:t.n] += lo
		parts.append(t)
		ch.close()
	merged = hip.merge_topk(parts, LEN_T, 10)
	assert list(merged.sentence[:10]) == list(top.sentence[:10])
	np.testing.assert_array_equal(merged.score[:10], top.score[:10])
	np.testing.assert_array_equal(merged.mapping[:10], top.mapping[:10])


def test_config3_and_config4_full_size_properties(hip, oracle):
	"""BASELINE configs[2] at its per-GPU size (1,250,000 sentences x 32 x 300-d, global alignment, linear gap 0.1) and configs[3]
	(256 queries x the same kind of corpus, relaxed WMD as one GEMM-shaped pass), through properties:
	  * config 3: a sample against the oracle, the selection against the full score vector, a planted copy first;
	  * config 4: every query of the batch against the single-query path (same sentences, scores to 2e-6), a sample against the
	    oracle."""
	if torch is None:
		pytest.skip("torch is needed to generate the shard on the device")
	n3 = 1_250_000 // (1 << 15) * (1 << 15)            # whole chunks of the generator
	rng = np.random.default_rng(12)
	E = synth.make_vocab(V, D)
	ids = synth.zipf_ids(n3 * LEN_S, V, rng)
	qv = np.ascontiguousarray(E[rng.integers(0, V, size=LEN_T)] + 0.05 * rng.standard_normal((LEN_T, D)).astype(np.float32))
	planted = {int(s): qv for s in (77_777, n3 - 1)}
	sample = np.unique(np.concatenate((rng.integers(0, n3, size=2000), list(planted))))
	c, kept = build(hip, torch, ids, E, 0, n3, sample, planted)
	Xs = np.concatenate([kept[int(s)] for s in sample])
	Xb, _ = oracle.normalize_rows_bf16(Xs)
	soff = np.arange(len(sample) + 1, dtype=np.int64) * LEN_S

	# ---- config 3: global alignment (Needleman-Wunsch), linear gap 0.1 on both axes
	kw = dict(q_normalize=True, locality=1, gap_s=0.1, gap_t=0.1, max_matches=10, min_score=-1e9)
	top = c.query(qv, **kw)
	scores = c.last_scores()
	assert scores.shape == (n3,) and np.isfinite(scores).all()
	Qb, _ = oracle.normalize_rows_bf16(qv)
	ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=D, sent_off=soff, X=Xb, Q=Qb, locality=1, gap_s=0.1, gap_t=0.1,
		max_matches=10, min_score=-1e9, want_all_scores=True, n_threads=8)
	np.testing.assert_allclose(scores[sample], ref["all_scores"], atol=1e-4, rtol=0)
	check_selection(top, scores, 10)
	check_winners_exact(oracle, top, regenerate(torch, E, ids, top.sentence[:top.n], planted, n3), qv,
		locality=1, gap_s=0.1, gap_t=0.1, min_score=-1e9)
	assert set(top.sentence[:2]) == set(planted)
	for i in range(2):                                   # the copy sits at tokens 5..14: the ten matches, 22 skipped tokens
		assert list(top.mapping[i]) == list(range(5, 5 + LEN_T))
		assert abs(top.score[i] - (LEN_T - 0.1 * (LEN_S - LEN_T)) / LEN_T) < 2e-3

	# ---- config 4: 256 queries, rwmd('nbow') = relaxed, injective, symmetric, normalised
	B = 256
	qs = [np.ascontiguousarray(E[rng.integers(0, V, size=LEN_T)] + 0.05 * rng.standard_normal((LEN_T, D)).astype(np.float32)) for _ in range(B)]
	targets = rng.integers(0, n3, size=B)
	for i in range(0, B, 4):                             # every fourth query: a noisy copy of ten tokens of a sampled sentence
		s = int(sample[int(rng.integers(0, len(sample)))])
		qs[i] = np.ascontiguousarray(kept[s][3:3 + LEN_T] + 0.02 * rng.standard_normal((LEN_T, D)).astype(np.float32))
		targets[i] = s
	qs[5] = qs[5][:7]                                    # shorter queries in the same batch
	qs[6] = qs[6][:1]
	flags = (True, True, True)
	outs = c.query_batch(qs, algorithm=hip.VK_ALG_RWMD, rwmd=flags, q_normalize=True, max_matches=10, min_score=0.0)
	assert len(outs) == B
	# (a noisy copy of part of a sentence need not win under the symmetric relaxed distance: the sentence's other 22 tokens count too)
	for i in (0, 1, 4, 5, 6, 17, 100, 255):
		single = c.query(qs[i], algorithm=hip.VK_ALG_RWMD, rwmd=flags, q_normalize=True, max_matches=10, min_score=0.0)
		np.testing.assert_allclose(outs[i].score[:outs[i].n], single.score[:single.n], atol=2e-6, rtol=0)
		diff = outs[i].sentence[:outs[i].n] != single.sentence[:single.n]
		assert not diff.any() or np.abs(np.diff(single.score[:single.n]))[np.nonzero(diff)[0].clip(max=single.n - 2)].max() < 2e-6
		Qi, _ = oracle.normalize_rows_bf16(qs[i])
		r = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=D, sent_off=soff, X=Xb, Q=Qi, algorithm=oracle.ALG_RWMD, rwmd=flags,
			max_matches=10, min_score=0.0, want_all_scores=True, n_threads=8)
		np.testing.assert_allclose(c.last_scores()[sample], r["all_scores"], atol=2e-5, rtol=0)   # scores of the single-query pass
	c.close()


def test_config5_full_size_properties(hip, oracle):
	"""BASELINE configs[4] at its per-GPU size: 1,000,000 sentences of 8..64 tokens x 768-d with their magnitudes (55 GB resident), WSB
	local alignment and Word Rotator's Distance.  Alignment: a sample against the oracle, the selection against the score vector.
	WRD (bound pass + exact EMD of the survivors): the winners' scores against the oracle on their regenerated vectors, no sampled
	sentence above the k-th winner, a planted copy first."""
	if torch is None:
		pytest.skip("torch is needed to generate the shard on the device")
	device = torch.device("cuda", 0)
	n, d, len_t = 1_000_000, 768, 10
	rng = np.random.default_rng(13)
	lens = rng.integers(8, 65, size=n)
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	n_tok = int(off[-1])
	E = synth.make_vocab(V, d)
	E_dev = torch.from_numpy(E).to(device)
	chunk_s = 1 << 14                                   # sentences per generator chunk

	def chunk_vectors(a):
		"""fp32 vectors of sentences [a, a + chunk_s): the same on every call (seeded per chunk), magnitudes spread by LogNormal(0, 0.25)"""
		b = min(a + chunk_s, n)
		gen = torch.Generator(device=device)
		gen.manual_seed(500 + a)
		nt = int(off[b] - off[a])
		idx = torch.randint(0, V, (nt,), device=device, generator=gen)
		x = E_dev[idx] + 0.3 * torch.randn((nt, d), device=device, generator=gen, dtype=torch.float32)
		return (x * torch.exp(0.25 * torch.randn((nt, 1), device=device, generator=gen, dtype=torch.float32))).contiguous()

	def sentence_vectors(s):
		a = s // chunk_s * chunk_s
		x = chunk_vectors(a)
		return x[int(off[s] - off[a]):int(off[s + 1] - off[a])].cpu().numpy()

	sample = np.unique(rng.integers(0, n, size=1500))
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=n_tok, n_sentences=n, keep_magnitudes=True)
	kept = {}
	for a in range(0, n, chunk_s):
		x = chunk_vectors(a)
		for s in sample[(sample >= a) & (sample < a + chunk_s)]:
			kept[int(s)] = x[int(off[s] - off[a]):int(off[s + 1] - off[a])].cpu().numpy()
		torch.cuda.synchronize()
		c.append_vectors_device(x.data_ptr(), x.shape[0], hip.VK_F32, normalize=True)
		del x
	c.set_sentences(off)
	c.finalize()
	assert c.device_bytes > 55e9
	src = int(sample[len(sample) // 2])
	while lens[src] < len_t + 2:
		src += 1
	base = sentence_vectors(src)
	qv = np.ascontiguousarray(base[1:1 + len_t] + 0.02 * rng.standard_normal((len_t, d)).astype(np.float32) * np.linalg.norm(base[1:1 + len_t], axis=1, keepdims=True) / np.sqrt(d))
	Xs = np.concatenate([kept[int(s)] for s in sample])
	Xb, mag = oracle.normalize_rows_bf16(Xs)
	soff = np.concatenate(([0], np.cumsum(lens[sample]))).astype(np.int64)
	Qb, qmag = oracle.normalize_rows_bf16(qv)

	# ---- WSB local alignment
	kw = dict(q_normalize=True, locality=0, gap_s=EXP5, gap_t=EXP5, max_matches=10, min_score=0.0)
	top = c.query(qv, **kw)
	scores = c.last_scores()
	ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=soff, X=Xb, Q=Qb, locality=0, gap_s=EXP5, gap_t=EXP5,
		max_matches=10, min_score=0.0, want_all_scores=True, n_threads=8)
	np.testing.assert_allclose(scores[sample], ref["all_scores"], atol=1e-4, rtol=0)
	check_selection(top, scores, 10)
	check_winners_exact(oracle, top, {int(s_): sentence_vectors(int(s_)) for s_ in top.sentence[:top.n]}, qv,
		locality=0, gap_s=EXP5, gap_t=EXP5, min_score=-1.0)
	assert top.sentence[0] == src and list(top.mapping[0]) == list(range(1, 1 + len_t))

	# ---- Word Rotator's Distance
	for normalize in (True, False):
		wrd = c.query(qv, algorithm=hip.VK_ALG_WRD, q_normalize=True, wrd_normalize=normalize, max_matches=10, min_score=0.0)
		assert wrd.n == 10 and (np.diff(wrd.score[:10]) <= 0).all()
		r = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=soff, X=Xb, X_mag=mag, Q=Qb, Q_mag=qmag, algorithm=oracle.ALG_WRD,
			wrd_normalize=normalize, max_matches=10, min_score=0.0, want_all_scores=True, n_threads=8)
		# no sampled sentence beats the k-th winner unless it is a winner itself
		better = sample[r["all_scores"] > wrd.score[9] + 2e-5]
		assert set(int(s) for s in better) <= set(int(s) for s in wrd.sentence[:10])
		# the winners' scores, exactly: their vectors regenerated, the oracle solving each
		for i in (0, 1, 4, 9):
			xv = sentence_vectors(int(wrd.sentence[i]))
			xb, xm = oracle.normalize_rows_bf16(xv)
			one = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=np.array([0, len(xv)], dtype=np.int64), X=xb, X_mag=xm, Q=Qb, Q_mag=qmag,
				algorithm=oracle.ALG_WRD, wrd_normalize=normalize, max_matches=1, min_score=-1.0)
			assert abs(one["score"][0] - wrd.score[i]) < 2e-5
		if normalize:
			assert wrd.sentence[0] == src
	c.close()
