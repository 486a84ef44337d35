"""-m gpu: token filters (pos_filter / tag_filter, vectorian/core/cpp/query.h:8-28; FilteredSliceFactory,
slice/static.h:366-416).  vk_corpus_filter builds the filtered corpus on the device; the oracle sees the same corpus
compacted on the host, array by array: same slice ids, scores and tracebacks (positions among the passing tokens)."""

import numpy as np
import pytest

from vectorian_amd import synth

from helpers import assert_same_results, prep_query

pytestmark = pytest.mark.gpu

EXP5 = ("table", (1 - 2.0 ** (-np.arange(0, 65) / 5)).astype(np.float32))


def host_filter(pos, tags, pos_mask, tag_mask):
	"""TokenFilter::pass over code arrays; codes outside 0..63 have no bit"""
	drop = np.zeros(len(pos), dtype=bool)
	for mask, codes in ((pos_mask, pos), (tag_mask, tags)):
		cd = codes.astype(np.int64)
		bits = np.array([(int(mask) >> b) & 1 for b in range(64)], dtype=bool)
		drop |= (cd >= 0) & (cd < 64) & bits[np.clip(cd, 0, 63)]
	keep = ~drop
	return keep, np.concatenate(([0], np.cumsum(keep))).astype(np.int64)


def codes(rng, n):
	pos = rng.integers(0, 7, size=n).astype(np.int8)
	tags = rng.integers(0, 80, size=n).astype(np.int8)      # some codes beyond the 64 bits of a mask
	return pos, tags


MASKS = [((1 << 2) | (1 << 5), 0), (0, (1 << 7) | (1 << 63) | (1 << 0)), (1 << 1, sum(1 << b for b in range(10, 40)))]


@pytest.mark.parametrize("precision", ["bf16", "f32"])
@pytest.mark.parametrize("d,lo,hi,len_t", [(300, 1, 40, 10), (768, 8, 64, 5), (72, 1, 30, 16)])
def test_contextual_filter(hip, oracle, d, lo, hi, len_t, precision):
	n = 500
	corpus = synth.make_contextual_corpus(n, lo, hi, 1500, d, noise=0.3, norm_sigma=0.25)
	off = corpus["sent_off"]
	X = corpus["X"]
	if precision == "f32":
		Xr, mag = oracle.normalize_rows(X), oracle.magnitudes(X)
	else:
		Xr, mag = oracle.normalize_rows_bf16(X)
	rng = np.random.default_rng(5)
	pos, tags = codes(rng, X.shape[0])
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=X.shape[0], n_sentences=n, keep_magnitudes=True, precision=precision)
	c.append_vectors(X, normalize=True)
	c.set_sentences(off)
	c.finalize()
	c.set_token_pos(pos)
	c.set_token_tags(tags)
	for pos_mask, tag_mask in MASKS:
		keep, new_index = host_filter(pos, tags, pos_mask, tag_mask)
		assert 0 < keep.sum() < len(keep)
		f = c.filtered(pos_mask, tag_mask)
		f_off = new_index[off]
		base = dict(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=f_off, X=Xr[keep], X_mag=mag[keep], max_matches=12, want_all_scores=True)
		for q in synth.make_queries(corpus, 2, len_t):
			qv = q["vectors"].astype(np.float32)
			Qr, qmag = (oracle.normalize_rows(qv), oracle.magnitudes(qv)) if precision == "f32" else oracle.normalize_rows_bf16(qv)
			for kw in (dict(gap_s=0.1, gap_t=0.1), dict(gap_s=EXP5, gap_t=EXP5, locality=2), dict(algorithm=hip.VK_ALG_RWMD)):
				ref = oracle.find(Q=Qr, Q_mag=qmag, **base, **kw)
				got = f.query(qv, q_normalize=True, max_matches=12, **kw)
				assert_same_results(got.trimmed(), ref, check_mapping="algorithm" not in kw, exact=True)
				every, live = f.last_scores(), np.diff(f_off) > 0       # slices the filter emptied are skipped (document.h:160)
				np.testing.assert_allclose(every[live], ref["all_scores"][live], atol=1e-4, rtol=0)
				assert np.isneginf(every[~live]).all()
			if precision == "bf16" and len_t <= 16:
				ref = oracle.find(Q=Qr, Q_mag=qmag, algorithm=oracle.ALG_WRD, **base)
				got = f.query(qv, q_normalize=True, max_matches=12, algorithm=hip.VK_ALG_WRD)
				assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=1e-5)
		f.close()
	c.close()


def test_static_filter_views_and_batches(hip, oracle):
	corpus = synth.make_static_corpus(700, 1, 40, 900, 300, seed=31)
	Eb = synth.to_bf16_bits(synth.normalize_rows(corpus["E"]))
	off, ids = corpus["sent_off"], corpus["tok_id"]
	rng = np.random.default_rng(6)
	pos, tags = codes(rng, len(ids))
	c = hip.Corpus(layout=hip.VK_LAYOUT_STATIC, d=300, n_tokens=len(ids), n_sentences=len(off) - 1, vocab_size=Eb.shape[0])
	c.append_vectors(Eb, normalize=False)
	c.set_token_ids(ids)
	c.set_token_pos(pos)
	c.set_token_tags(tags)
	c.set_sentences(off)
	c.finalize()
	pos_mask, tag_mask = MASKS[2]
	keep, new_index = host_filter(pos, tags, pos_mask, tag_mask)
	f = c.filtered(pos_mask, tag_mask)
	v = f.view()
	base = dict(layout=oracle.LAYOUT_STATIC, d=300, sent_off=new_index[off], tok_id=ids[keep], E=Eb, max_matches=15)
	qs = synth.make_queries(corpus, 4, 7)
	for handle, q in zip((f, v, f, v), qs):
		Qb = prep_query(q)
		for kw in (dict(gap_s=EXP5, gap_t=EXP5), dict(gap_s=("affine", 0.2, 0.05), gap_t=0.1, locality=1, min_score=-100.0)):
			ref = oracle.find(Q=Qb, q_ids=q["ids"], **base, **kw)
			got = handle.query(Qb, q_token_ids=q["ids"], q_normalize=False, max_matches=15, **kw)
			assert_same_results(got.trimmed(), ref)
	# a batch of queries on the filtered corpus
	# (general gaps: with linear gaps and repeated tokens co-optimal alignments tie exactly, and the last bit of S decides)
	refs = [oracle.find(Q=prep_query(q), gap_s=EXP5, gap_t=EXP5, **base) for q in qs]
	outs = f.query_batch([prep_query(q) for q in qs], q_normalize=False, gap_s=EXP5, gap_t=EXP5, max_matches=15)
	for got, ref in zip(outs, refs):
		assert_same_results(got.trimmed(), ref)
	# the unfiltered corpus is untouched
	q = qs[0]
	ref = oracle.find(Q=prep_query(q), q_ids=q["ids"], layout=oracle.LAYOUT_STATIC, d=300, sent_off=off, tok_id=ids, E=Eb, max_matches=15, gap_s=EXP5, gap_t=EXP5)
	assert_same_results(c.query(prep_query(q), q_token_ids=q["ids"], q_normalize=False, gap_s=EXP5, gap_t=EXP5, max_matches=15).trimmed(), ref)
	v.close(); f.close(); c.close()


def test_filter_over_windows_and_long_slices(hip, oracle):
	"""overlapping windows, some longer than 64 tokens before the filter and shorter after it; a filter that empties
	slices; a filter that drops everything"""
	n, d = 240, 128
	corpus = synth.make_contextual_corpus(n, 5, 60, 800, d)
	off = corpus["sent_off"]
	Xb = synth.to_bf16_bits(synth.normalize_rows(corpus["X"]))
	start = off[:-1].copy()
	end = off[np.minimum(np.arange(n) + 3, n)]            # windows of 3 sentences, step 1
	rng = np.random.default_rng(7)
	pos = rng.integers(1, 4, size=Xb.shape[0]).astype(np.int8)
	pos[off[10]:off[14]] = 3                                # a stretch that one filter removes completely
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=Xb.shape[0], n_sentences=n)
	c.append_vectors(Xb, normalize=False)
	c.set_slices(start, end)
	c.set_token_pos(pos)
	c.finalize()
	assert (end - start).max() > 64
	q = synth.make_queries(corpus, 1, 8)[0]
	Qb = prep_query(q)
	long_exp5 = ("table", (1 - 2.0 ** (-np.arange(0, 256) / 5)).astype(np.float32))
	for pos_mask in (1 << 3, (1 << 1) | (1 << 2), 0b1110):
		keep, new_index = host_filter(pos, np.zeros_like(pos), pos_mask, 0)
		f = c.filtered(pos_mask, 0)
		for kw in (dict(gap_s=0.1, gap_t=0.1), dict(gap_s=long_exp5, gap_t=long_exp5)):
			got = f.query(Qb, q_normalize=False, max_matches=10, **kw).trimmed()
			if keep.sum() == 0:
				assert len(got["score"]) == 0
				continue
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=new_index[start], sent_end=new_index[end], X=Xb[keep], Q=Qb,
				max_matches=10, **kw)
			assert_same_results(got, ref)
		f.close()
	with pytest.raises(hip.VkError):
		c.filtered(0, 1 << 4)       # no tag codes on this corpus
	c.close()
