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


@pytest.mark.parametrize("seed", range(200))
def test_random_problem(hip, oracle, seed):
	rng = np.random.default_rng(1000 + seed)
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
	len_t = int(rng.integers(1, 25))
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
	# tracebacks are compared where the scores are not within rounding of a neighbour's (helpers: tie_tol); a static
	# slice may hold one word twice, and which of the two equal cells the traceback takes can flip with the last bit
	assert_same_results(got.trimmed(), ref, score_tol=2e-5, tie_tol=2e-6, check_mapping=not static)
	c.close()
