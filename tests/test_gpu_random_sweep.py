"""-m gpu: a seeded sweep over random small problems -- layout, dimension, slice lengths (empty and long ones
included), query length (1 .. 24), locality, gap family, k, boost, precision -- HIP through the C-ABI against the oracle."""

import numpy as np
import pytest

from vectorian_amd import synth

from helpers import assert_same_results

pytestmark = pytest.mark.gpu


def random_gap(rng, n):
	kind = rng.integers(0, 4)
	if kind == 0:
		return float(rng.uniform(0.0, 0.4))
	if kind == 1:
		return ("affine", float(rng.uniform(0.0, 0.4)), float(rng.uniform(0.0, 0.2)))
	if kind == 2:   # concave, strictly subadditive
		return ("table", (rng.uniform(0.3, 1.0) * (1 - 2.0 ** (-np.arange(0, n) / rng.uniform(1.0, 8.0)))).astype(np.float32))
	w = np.concatenate(([0.0], np.cumsum(rng.uniform(0.0, 0.3, size=n - 1)))).astype(np.float32)   # arbitrary increasing table
	return ("table", w)


# VK_SWEEP_SCALE=10 runs ten times the seeds (a one-off soak on the GPU box; the committed tier runs the default)
SCALE = int(__import__("os").environ.get("VK_SWEEP_SCALE", "1"))


@pytest.mark.parametrize("seed", range(200 * SCALE))
def test_random_problem(hip, oracle, seed):
	run_random_problem(hip, oracle, 1000 + seed, 1, 25)


@pytest.mark.parametrize("seed", range(100 * SCALE))
def test_random_problem_long_query(hip, oracle, seed):
	"""queries of 17..64 tokens: the multi-block kernel (vk_score32_kernel) and its fallbacks (long slices, fp32 tiles,
	gap costs that are not subadditive)"""
	run_random_problem(hip, oracle, 5000 + seed, 17, 65)


def run_random_problem(hip, oracle, seed, len_lo, len_hi):
	rng = np.random.default_rng(seed)
	static = bool(rng.integers(0, 2))
	d = int(rng.choice([16, 50, 64, 100, 300, 320]))
	n = int(rng.integers(1, 300))
	lens = rng.integers(0, 41, size=n)
	if rng.random() < 0.3:
		lens[rng.integers(0, n, size=max(1, n // 20))] = rng.integers(65, 200)
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	T = int(off[-1])
	if T == 0:
		pytest.skip("empty corpus")
	len_t = int(rng.integers(len_lo, len_hi))
	precision = "f32" if rng.random() < 0.25 else "bf16"
	loc = int(rng.integers(0, 3))
	gs, gt = random_gap(rng, 513), random_gap(rng, 65)
	k = int(rng.choice([1, 5, 10, 70]))
	boost = rng.uniform(0.5, 1.5, size=n).astype(np.float32) if rng.random() < 0.3 else None
	ms = 0.0 if loc == 0 else -1e9
	kw = dict(locality=loc, gap_s=gs, gap_t=gt, max_matches=k, min_score=ms, boost=boost)
	if static:
		V = int(rng.integers(5, 200))
		E = rng.standard_normal((V, d)).astype(np.float32)
		ids = rng.integers(0, V, size=T).astype(np.int32)
		q_ids = rng.integers(0, V, size=len_t).astype(np.int32)
		c = hip.Corpus(layout=hip.VK_LAYOUT_STATIC, d=d, n_tokens=T, n_sentences=n, vocab_size=V, precision=precision)
		c.append_vectors(E, normalize=True)
		c.set_token_ids(ids)
		c.set_sentences(off)
		c.finalize()
		En = oracle.normalize_rows(E) if precision == "f32" else oracle.normalize_rows_bf16(E)[0]
		ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=d, sent_off=off, tok_id=ids, E=En, Q=En[q_ids], q_ids=q_ids, **kw)
		got = c.query(E[q_ids], q_token_ids=q_ids, q_normalize=True, **kw)
	else:
		X = rng.standard_normal((T, d)).astype(np.float32)
		qv = rng.standard_normal((len_t, d)).astype(np.float32)
		if rng.random() < 0.5 and lens.max() >= 1:   # plant part of a slice
			s = int(np.argmax(lens))
			m = min(len_t, int(lens[s]))
			qv[:m] = X[off[s]:off[s] + m] + 0.1 * rng.standard_normal((m, d)).astype(np.float32)
		c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=T, n_sentences=n, precision=precision)
		c.append_vectors(X, normalize=True)
		c.set_sentences(off)
		c.finalize()
		if precision == "f32":
			Xn, Qn = oracle.normalize_rows(X), oracle.normalize_rows(qv)
		else:
			Xn, Qn = oracle.normalize_rows_bf16(X)[0], oracle.normalize_rows_bf16(qv)[0]
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=off, X=Xn, Q=Qn, **kw)
		got = c.query(qv, q_normalize=True, **kw)
	# winners are restated in the oracle's arithmetic (sim_canon): ids, scores and tracebacks bit for bit, in every layout --
	# a static slice that holds one word twice (two equal similarity rows, co-optimal alignments) included
	assert_same_results(got.trimmed(), ref)
	c.close()


@pytest.mark.parametrize("seed", range(80 * SCALE))
def test_random_transport_and_modifiers(hip, oracle, seed):
	"""the other strategies on random small problems: RWMD (all forms), full WMD, WRD (both mass conventions),
	submatch_weight, tag-weighted alignment"""
	rng = np.random.default_rng(5000 + seed)
	d = int(rng.choice([24, 64, 300]))
	n = int(rng.integers(2, 250))
	lens = rng.integers(1, 41, size=n)
	mode = int(np.random.default_rng(77000 + seed).integers(0, 5))   # (drawn apart: the other draws keep their sequence)
	if mode in (0, 3, 4) and np.random.default_rng(78000 + seed).random() < 0.35:
		# a few slices of 65 .. 400 tokens: the relaxed 1:1 WMD streams them (vk_doc_kernel<false, 4, .>), their alignments and the
		# tracebacks of submatch candidates / tag-weighted winners run on vk_doc_kernel's sweep
		r2 = np.random.default_rng(79000 + seed)
		lens[r2.integers(0, n, size=max(1, n // 15))] = r2.integers(65, 401)
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	T = int(off[-1])
	len_t = int(rng.integers(1, 17))
	X = (rng.standard_normal((T, d)) * rng.lognormal(0, 0.3, size=(T, 1))).astype(np.float32)
	qv = (rng.standard_normal((len_t, d)) * rng.lognormal(0, 0.3, size=(len_t, 1))).astype(np.float32)
	s = int(rng.integers(0, n))
	m = min(len_t, int(lens[s]))
	qv[:m] = X[off[s]:off[s] + m] * rng.uniform(0.5, 2.0) + 0.2 * rng.standard_normal((m, d)).astype(np.float32)
	Xn, mag = oracle.normalize_rows_bf16(X)
	Qn, qmag = oracle.normalize_rows_bf16(qv)
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=T, n_sentences=n, keep_magnitudes=True)
	c.append_vectors(X, normalize=True)
	c.set_sentences(off)
	pos = rng.integers(0, 5, size=T).astype(np.int8)
	c.set_token_pos(pos)
	c.finalize()
	k = int(rng.choice([1, 6, 20]))
	rng.integers(0, 5)   # (the draw the mode used to take)
	base = dict(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=off, X=Xn, Q=Qn, max_matches=k)
	if mode == 0:      # relaxed WMD, any legal form
		inj, sym, nbow = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
		if sym and not nbow:
			nbow = True
		kw = dict(rwmd=(inj, sym, nbow), min_score=-10.0)
		ref = oracle.find(algorithm=oracle.ALG_RWMD, **base, **kw)
		got = c.query(qv, q_normalize=True, algorithm=hip.VK_ALG_RWMD, max_matches=k, **kw)
		assert_same_results(got.trimmed(), ref, check_mapping=False, exact=True)   # restated on the host from canonical rows: the oracle's floats
	elif mode == 1:    # full WMD
		nbow = bool(rng.integers(0, 2))
		ref = oracle.find(algorithm=oracle.ALG_RWMD, rwmd=(False, False, nbow), wmd_full=True, min_score=0.0, **base)
		got = c.query(qv, q_normalize=True, algorithm=hip.VK_ALG_RWMD, rwmd=(False, False, nbow), wmd_full=True, max_matches=k, min_score=0.0)
		assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=2e-5, tie_tol=2e-5)
	elif mode == 2:    # WRD
		norm = bool(rng.integers(0, 2))
		ref = oracle.find(algorithm=oracle.ALG_WRD, X_mag=mag, Q_mag=qmag, wrd_normalize=norm, min_score=0.0, **base)
		got = c.query(qv, q_normalize=True, algorithm=hip.VK_ALG_WRD, wrd_normalize=norm, max_matches=k, min_score=0.0)
		assert_same_results(got.trimmed(), ref, check_mapping=False, score_tol=2e-5, tie_tol=2e-5)
	elif mode == 3:    # submatch weight
		kw = dict(locality=int(rng.integers(0, 3)), gap_s=float(rng.uniform(0, 0.3)), gap_t=float(rng.uniform(0, 0.3)),
			submatch_weight=float(rng.choice([0.5, 1.0, 2.0])), min_score=0.0)
		ref = oracle.find(**base, **kw)
		got = c.query(qv, q_normalize=True, max_matches=k, **kw)
		assert_same_results(got.trimmed(), ref, score_tol=2e-5)
	else:              # tag-weighted alignment
		tw = rng.uniform(0.2, 1.5, size=len_t).astype(np.float32)
		qp = rng.integers(0, 5, size=len_t).astype(np.int8)
		kw = dict(locality=0, gap_s=0.1, gap_t=0.1, min_score=0.0, tag_weights=tw, q_pos=qp,
			pos_mismatch_penalty=float(rng.uniform(0, 0.5)), similarity_threshold=float(rng.uniform(0, 0.3)))
		ref = oracle.find(pos_s=pos, **base, **kw)
		got = c.query(qv, q_normalize=True, max_matches=k, **kw)
		assert_same_results(got.trimmed(), ref, score_tol=2e-5)
	c.close()


@pytest.mark.parametrize("seed", range(30 * SCALE))
def test_random_documents(hip, oracle, seed):
	"""random corpora that mix sentences, paragraphs (65 .. 512 tokens) and documents (up to 1,500): query of 1 .. 64 tokens, every
	locality and gap family (tables that saturate early, late or never), both layouts and row types, boost -- the routes of round 4
	(vk_doc_kernel, vk_docw_kernel, vk_docg_kernel, the multi-block kernel beside them, vk_wide_kernel for the rest) against the oracle"""
	rng = np.random.default_rng(9000 + seed)
	static = bool(rng.integers(0, 2))
	d = int(rng.choice([32, 64, 100, 300]))
	n = int(rng.integers(5, 60))
	lens = rng.integers(0, 50, size=n)
	for _ in range(int(rng.integers(1, 5))):
		lens[rng.integers(0, n)] = rng.integers(65, 513)
	for _ in range(int(rng.integers(0, 3))):
		lens[rng.integers(0, n)] = rng.integers(513, 1501)
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	T = int(off[-1])
	len_t = int(rng.choice([1, 3, 9, 16, 17, 24, 32, 33, 48, 64]))
	precision = "f32" if rng.random() < 0.2 else "bf16"
	loc = int(rng.integers(0, 3))
	def gap(n_tab):
		kind = rng.integers(0, 5)
		if kind == 0:
			return float(rng.uniform(0.0, 0.4))
		if kind == 1:
			return ("affine", float(rng.uniform(0.0, 0.4)), float(rng.uniform(0.0, 0.2)))
		if kind == 2:   # saturates somewhere between 10 and 200 tokens
			return ("table", (rng.uniform(0.3, 1.0) * (1 - 2.0 ** (-np.arange(0, n_tab) / rng.uniform(0.4, 8.0)))).astype(np.float32))
		if kind == 3:   # constant from a random k on
			w = np.sort(rng.uniform(0.0, 1.0, size=n_tab)).astype(np.float32)
			w[0] = 0.0
			t = int(rng.integers(1, min(140, n_tab - 1)))
			w[t:] = w[t]
			return ("table", w)
		return ("table", np.concatenate(([0.0], np.cumsum(rng.uniform(0.0, 0.05, size=n_tab - 1)))).astype(np.float32))   # never
	gs, gt = gap(1502), gap(65)
	k = int(rng.choice([1, 4, 12]))
	boost = rng.uniform(0.5, 1.5, size=n).astype(np.float32) if rng.random() < 0.3 else None
	ms = 0.0 if loc == 0 else -1e9
	kw = dict(locality=loc, gap_s=gs, gap_t=gt, max_matches=k, min_score=ms, boost=boost)
	s0 = int(np.argmax(lens))
	if static:
		V = int(rng.integers(20, 300))
		E = rng.standard_normal((V, d)).astype(np.float32)
		ids = rng.integers(0, V, size=T).astype(np.int32)
		a = int(off[s0])
		q_ids = ids[a:a + 3 * len_t:3][:len_t].astype(np.int32)
		if len(q_ids) < len_t:
			q_ids = np.concatenate((q_ids, rng.integers(0, V, size=len_t - len(q_ids)).astype(np.int32)))
		c = hip.Corpus(layout=hip.VK_LAYOUT_STATIC, d=d, n_tokens=T, n_sentences=n, vocab_size=V, precision=precision)
		c.append_vectors(E, normalize=True)
		c.set_token_ids(ids)
		c.set_sentences(off)
		c.finalize()
		En = oracle.normalize_rows(E) if precision == "f32" else oracle.normalize_rows_bf16(E)[0]
		ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=d, sent_off=off, tok_id=ids, E=En, Q=En[q_ids], q_ids=q_ids, n_threads=8, **kw)
		got = c.query(E[q_ids], q_token_ids=q_ids, q_normalize=True, **kw)
	else:
		X = rng.standard_normal((T, d)).astype(np.float32)
		qv = rng.standard_normal((len_t, d)).astype(np.float32)
		m = min(len_t, int(lens[s0]))
		idx = np.sort(rng.choice(np.arange(int(off[s0]), int(off[s0 + 1])), size=m, replace=False))
		qv[:m] = X[idx] + 0.1 * rng.standard_normal((m, d)).astype(np.float32)
		c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=d, n_tokens=T, n_sentences=n, precision=precision)
		c.append_vectors(X, normalize=True)
		c.set_sentences(off)
		c.finalize()
		Xn = oracle.normalize_rows(X) if precision == "f32" else oracle.normalize_rows_bf16(X)[0]
		Qn = oracle.normalize_rows(qv) if precision == "f32" else oracle.normalize_rows_bf16(qv)[0]
		ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=d, sent_off=off, X=Xn, Q=Qn, n_threads=8, **kw)
		got = c.query(qv, q_normalize=True, **kw)
	assert_same_results(got.trimmed(), ref)
	c.close()

