"""CPU tier: pins the oracle (the checker of every GPU parity test).

 - golden vectors produced by the reference's own CosineSim / Vectors code
   (tests/golden/cosine_reference.npz, generator tools/make_goldens.py)
 - the one known answer the reference ships (mkdocs/docs/introduction.md:150-184)
 - an exhaustive enumeration of alignments for tiny inputs (independent formulation)
 - exact EMD against scipy's LP solver; RWMD / WRD bounds
"""

import itertools
import json
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from vectorian_amd import synth
from vectorian_amd.alignment import smooth_gap_cost

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
	return np.load(os.path.join(GOLD, "cosine_reference.npz"))


@pytest.mark.parametrize("case", ["small", "wide", "tiny"])
def test_vectors_and_cosine_match_reference(oracle, gold, case):
	a, b = gold[case + "_a"], gold[case + "_b"]
	np.testing.assert_allclose(oracle.magnitudes(a), gold[case + "_a_mag"], rtol=2e-7, atol=0)
	an, bn = oracle.normalize_rows(a), oracle.normalize_rows(b)
	np.testing.assert_allclose(an, gold[case + "_a_norm"], rtol=0, atol=2e-7)
	np.testing.assert_allclose(bn, gold[case + "_b_norm"], rtol=0, atol=2e-7)
	# cosine + clip of the reference, from the reference's own normalised rows
	S = oracle.sim_f32(gold[case + "_a_norm"], gold[case + "_b_norm"])
	np.testing.assert_allclose(S, gold[case + "_cos_clipped"], rtol=0, atol=1e-6)
	# the product's host mirror of Vectors
	from vectorian_amd.embedding import Vectors
	from vectorian_amd.sim import CosineSim
	out = np.zeros_like(gold[case + "_cos"])
	with np.errstate(all="ignore"):
		CosineSim()(Vectors(a), Vectors(b), out)
	np.testing.assert_allclose(out, gold[case + "_cos"], rtol=0, atol=1e-6)


def test_bf16_rounding(oracle):
	x = np.array([1.0, 1.00390625, 1.001953125, 1.005859375, -2.5, 3.3895314e38, np.nan, 0.0, 1e-40], dtype=np.float32)
	b = oracle.round_bf16(x)
	assert (b == synth.to_bf16_bits(x)).all()
	back = oracle.bf16_to_f32(b)
	assert back[0] == 1.0 and back[1] == 1.0            # 1 + 2^-8 is a tie -> even mantissa
	assert back[2] == 1.0 and back[3] == 1.0078125 and np.isnan(back[6])
	rng = np.random.default_rng(0)
	y = rng.standard_normal(10000).astype(np.float32)
	assert np.max(np.abs(oracle.bf16_to_f32(oracle.round_bf16(y)) - y) / np.abs(y)) <= 2.0 ** -8


def test_known_answer_from_reference_docs(oracle):
	ka = json.load(open(os.path.join(GOLD, "known_answer_introduction.json")))
	d = ka["edge_distance"]
	S = np.zeros((4, 3), dtype=np.float32)
	S[0, 0] = np.float32(1) - np.float32(d[0])
	S[1, 1] = np.float32(1) - np.float32(d[1])
	S[3, 2] = np.float32(1) - np.float32(d[2])
	S[2, 0], S[2, 2] = 0.05, 0.1
	gap = smooth_gap_cost(5)
	w = gap.costs(65)
	assert float(w[1]) == ka["gap_penalty_one_skipped_token"]
	raw, mapping = oracle.align(S, oracle.LOCAL, ("table", w), ("table", w))
	assert list(mapping) == ka["mapping"]
	assert float(np.float32(oracle.score(raw, 3, 3))) == ka["score"]   # bit exact


# ---------------------------------------------------------------------------
# exhaustive enumeration (independent of the DP): every monotone matching, gaps between
# consecutive matches paid once per axis -- valid for subadditive gap costs
# ---------------------------------------------------------------------------

def brute_force(S, locality, ws, wt):
	ls, lt = S.shape
	best = None
	cells = [(u, v) for u in range(ls) for v in range(lt)]
	for r in range(0, min(ls, lt) + 1):
		for combo in itertools.combinations(cells, r):
			if any(not (a[0] < b[0] and a[1] < b[1]) for a, b in zip(combo, combo[1:])):
				continue
			if r == 0:
				val = -(ws[ls] + wt[lt]) if locality == 1 else 0.0
			else:
				val = sum(float(S[u, v]) for u, v in combo)
				for a, b in zip(combo, combo[1:]):
					val -= ws[b[0] - a[0] - 1] + wt[b[1] - a[1] - 1]
				(u0, v0), (u1, v1) = combo[0], combo[-1]
				if locality == 1:
					val -= ws[u0] + wt[v0] + ws[ls - 1 - u1] + wt[lt - 1 - v1]
				elif locality == 2:
					val -= min(ws[u0], wt[v0]) + min(ws[ls - 1 - u1], wt[lt - 1 - v1])
			best = val if best is None else max(best, val)
	return best


@pytest.mark.parametrize("locality", [0, 1, 2])
@pytest.mark.parametrize("gapname", ["linear", "affine", "exp", "constant"])
def test_alignment_against_exhaustive_enumeration(oracle, locality, gapname):
	rng = np.random.default_rng(42 + locality)
	for trial in range(40):
		ls, lt = int(rng.integers(1, 5)), int(rng.integers(1, 5))
		S = rng.random((ls, lt)).astype(np.float32)
		if gapname == "linear":
			g, w = ("linear", 0.15), 0.15 * np.arange(8)
		elif gapname == "affine":
			g, w = ("affine", 0.2, 0.05), np.concatenate(([0], 0.2 + 0.05 * np.arange(1, 8)))
		elif gapname == "exp":
			w = 1 - 2.0 ** (-np.arange(8) / 3)
			g = ("table", w.astype(np.float32))
		else:
			w = np.concatenate(([0], np.full(7, 0.3)))
			g = ("table", w.astype(np.float32))
		raw, mapping = oracle.align(S, locality, g, g)
		ref = brute_force(S, locality, w, w)
		assert abs(raw - ref) < 1e-5, (trial, ls, lt, raw, ref)
		# the reported mapping is a valid alignment that achieves the score (matched part)
		m = [(int(mapping[j]), j) for j in range(lt) if mapping[j] >= 0]
		assert all(a[0] < b[0] for a, b in zip(m, m[1:]))
		# general solver agrees with the specialised ones
		raw2, _ = oracle.align(S, locality, ("table", w.astype(np.float32)), ("table", w.astype(np.float32)), general=True)
		assert abs(raw - raw2) < 1e-5


@pytest.mark.parametrize("gapname,len_long", [("linear", 30000), ("affine", 30000), ("exp5", 4000)])
def test_document_length_slices_embed_short_ones(oracle, gapname, len_long):
	"""Slices of document length (up to the mapping's int16, metric/alignment.h:357-358) through the same solvers: a short slice
	set into a long run of tokens that resemble nothing (similarity 0) keeps its local alignment -- the same aligner score, bit for
	bit, the same mapping shifted by its offset -- whatever the gap family; the short case is what the exhaustive enumeration and the
	reference's known answer pin"""
	rng = np.random.default_rng(7)
	g = {"linear": ("linear", 0.1), "affine": ("affine", 0.2, 0.05),
		"exp5": ("table", (1 - 2.0 ** (-np.arange(0, len_long + 1) / 5)).astype(np.float32))}[gapname]
	for trial in range(6):
		ls, lt = int(rng.integers(5, 60)), int(rng.integers(2, 14))
		S = np.clip(rng.normal(0.1, 0.35, size=(ls, lt)), 0, 1).astype(np.float32)
		raw, mapping = oracle.align(S, 0, g, g)
		off = int(rng.integers(0, len_long - ls))
		L = np.zeros((len_long, lt), dtype=np.float32)
		L[off:off + ls] = S
		raw_l, mapping_l = oracle.align(L, 0, g, g)
		assert np.float32(raw_l).view(np.uint32) == np.float32(raw).view(np.uint32), (trial, raw, raw_l)
		assert (np.asarray(mapping_l) == np.where(np.asarray(mapping) >= 0, np.asarray(mapping) + off, -1)).all(), (trial, off)
	# the longest slice there is, matched at its far end
	n_max, lt = 32767, 5
	L = np.zeros((n_max, lt), dtype=np.float32)
	L[n_max - lt + np.arange(lt), np.arange(lt)] = 0.5
	raw_l, mapping_l = oracle.align(L, 0, ("linear", 0.1), ("linear", 0.1))
	assert raw_l == 2.5 and list(mapping_l) == list(range(n_max - lt, n_max))


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 12), st.integers(1, 8), st.integers(0, 10 ** 6), st.floats(0.0, 0.5))
def test_locality_ordering_property(ls, lt, seed, g):
	from oracle import vk_oracle as vo
	S = np.random.default_rng(seed).random((ls, lt)).astype(np.float32)
	loc, _ = vo.align(S, vo.LOCAL, g, g)
	semi, _ = vo.align(S, vo.SEMIGLOBAL, g, g)
	glob, _ = vo.align(S, vo.GLOBAL, g, g)
	assert loc >= 0
	assert loc >= semi - 1e-6 >= glob - 2e-6
	# zero gap cost, local alignment = maximum-weight monotone matching >= best single cell
	z, _ = vo.align(S, vo.LOCAL, 0.0, 0.0)
	assert z >= S.max() - 1e-6


def test_score_formula(oracle):
	# reference_score (metric/alignment.h:84-106) + Score (match.h:302-307)
	assert oracle.score(2.4, 3, 3) == np.float32(2.4) / np.float32(3)
	assert oracle.score(2.4, 3, 1) == np.float32(2.4) / np.float32(3)      # submatch_weight 0: ref = len_t
	v = oracle.score(2.0, 4, 2, 1.0, 1.5)   # ref = 2 + (2/4)^1 * 2 = 3
	assert abs(v - 2.0 / 3.0 * 1.5) < 1e-6


def test_emd_matches_lp(oracle):
	from scipy.optimize import linprog
	rng = np.random.default_rng(3)
	for _ in range(10):
		n, m = int(rng.integers(1, 7)), int(rng.integers(1, 9))
		a = rng.random(n); a /= a.sum()
		b = rng.random(m); b /= b.sum()
		C = rng.random((n, m))
		cost, F = oracle.emd(a, b, C)
		A_eq, b_eq = [], []
		for i in range(n):
			r = np.zeros((n, m)); r[i] = 1; A_eq.append(r.ravel()); b_eq.append(a[i])
		for j in range(m):
			r = np.zeros((n, m)); r[:, j] = 1; A_eq.append(r.ravel()); b_eq.append(b[j])
		res = linprog(C.ravel(), A_eq=np.array(A_eq), b_eq=b_eq, bounds=(0, None))
		assert abs(cost - res.fun) < 1e-9
		np.testing.assert_allclose(F.sum(1), a, atol=1e-12)
		np.testing.assert_allclose(F.sum(0), b, atol=1e-12)


def test_rwmd_and_wrd_bounds(oracle):
	rng = np.random.default_rng(9)
	for _ in range(20):
		ls, lt = int(rng.integers(1, 20)), int(rng.integers(1, 10))
		S = rng.random((ls, lt)).astype(np.float32)
		# rwmd('nbow') on distinct tokens: closed form of SURVEY A.7
		D = np.maximum(1 - S, 0)
		acc0 = D.min(axis=0).mean()
		acc1 = D.min(axis=1).mean()
		want = 1 - max(acc0, acc1)
		got = oracle.rwmd(S, injective=True, symmetric=True, normalize_bow=True)
		assert abs(got - want) < 1e-5
		# uniform masses: WRD score = 1 - EMD <= RWMD score (RWMD cost is a lower bound of EMD)
		wrd = oracle.wrd(S, np.ones(ls), np.ones(lt))
		assert wrd <= got + 1e-5
		# one direction only is looser
		one = oracle.rwmd(S, injective=True, symmetric=False, normalize_bow=True)
		assert one >= got - 1e-6


def test_rwmd_static_bow_merges_repeated_tokens(oracle):
	# the same token twice in s: one vocabulary entry with weight 2/len_s (bow.h:204-275)
	S = np.array([[0.9, 0.1], [0.9, 0.1], [0.2, 0.7]], dtype=np.float32)
	ids_s, ids_t = [5, 5, 8], [1, 2]
	got = oracle.rwmd(S, ids_s=ids_s, ids_t=ids_t, injective=True, symmetric=True, normalize_bow=True)
	acc0 = ((1 - 0.9) + (1 - 0.7)) / 2                        # t -> s
	acc1 = (2 / 3) * (1 - 0.9) + (1 / 3) * (1 - 0.7)          # s -> t, weights 2/3 and 1/3
	assert abs(got - (1 - max(acc0, acc1))) < 1e-6


def test_find_topk_order_and_min_score(oracle):
	corpus = synth.make_contextual_corpus(300, 1, 20, 500, 32)
	Xb = synth.to_bf16_bits(synth.normalize_rows(corpus["X"]))
	q = synth.make_queries(corpus, 1, 4)[0]
	Qb = synth.to_bf16_bits(synth.normalize_rows(q["vectors"]))
	r = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=32, sent_off=corpus["sent_off"], X=Xb, Q=Qb, gap_s=0.1, gap_t=0.1,
		max_matches=25, min_score=0.05, want_all_scores=True)
	sc = r["all_scores"]
	order = sorted([i for i in range(300) if sc[i] > 0.05], key=lambda i: (-sc[i], -i))[:25]
	assert list(r["sentence"]) == order
	# threads do not change the result
	r4 = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=32, sent_off=corpus["sent_off"], X=Xb, Q=Qb, gap_s=0.1, gap_t=0.1,
		max_matches=25, min_score=0.05, n_threads=4)
	assert list(r4["sentence"]) == order
	assert (r4["mapping"] == r["mapping"]).all()
