"""-m gpu: a seeded sweep over the operator surface -- random sessions (static or contextual embedding, POS / tags, token masks),
partitions (sentence or token level, sliding windows), strategies (alignments of every locality and gap family, the relaxed /
full word mover's distances, word rotator's distance, each optionally tag-weighted), options (token filters, saliency boost,
submatch weight, n, min_score) -- through Session / Index.find on the HIP backend and on the oracle-backed double.
Alignments: documents, slices, scores, flows and the regions of the JSON report equal with `==` (the winners are restated in
the oracle's arithmetic on the device).  Transport strategies: scores at 2e-5, the same slices unless the double itself scores
them within that, flows of equal mass and cost.  VK_SWEEP_SCALE multiplies the seeds (soak runs)."""

import os

import numpy as np
import pytest

from fake_backend import OracleCorpus
from test_host_api import Corpus, Document, Session, StaticEmbedding
from vectorian_amd import alignment, synth
from vectorian_amd.embedding import ContextualEmbedding
from vectorian_amd.sim import CosineSim, EmbeddingTokenSim, OptimizedSpanSim

pytestmark = pytest.mark.gpu

SCALE = int(os.environ.get("VK_SWEEP_SCALE", "1"))
TAGS = ["NN", "VBZ", "DT", "JJ", ".", "NNS", "VBD"]
POS_OF = {"NN": "NOUN", "NNS": "NOUN", "VBZ": "VERB", "VBD": "VERB", "DT": "DET", "JJ": "ADJ", ".": "PUNCT"}


def random_gap(rng):
	kind = int(rng.integers(0, 5))
	if kind == 0:
		return alignment.ConstantGapCost(float(rng.uniform(0, 0.3)))
	if kind == 1:
		return alignment.LinearGapCost(float(rng.uniform(0, 0.4)))
	if kind == 2:
		return alignment.AffineGapCost(float(rng.uniform(0, 0.3)), float(rng.uniform(0, 0.2)))
	if kind == 3:
		return alignment.smooth_gap_cost(float(rng.uniform(1, 8)))
	return alignment.ExponentialGapCost(float(rng.uniform(1.5, 3.0)), float(rng.uniform(0.1, 0.9)))   # w(k) = 1 - base^(-k rate), base > 1


def random_strategy(rng):
	kind = int(rng.integers(0, 8))
	gap = random_gap(rng) if rng.random() < 0.6 else {"s": random_gap(rng), "t": random_gap(rng)}
	if kind <= 2:
		return [alignment.LocalAlignment, alignment.GlobalAlignment, alignment.SemiGlobalAlignment][kind](gap=gap), True
	if kind == 3:
		return alignment.LocalAlignment(gap=gap), True
	if kind == 4:
		return alignment.WordMoversDistance.rwmd(str(rng.choice(["nbow", "nbow/distributed", "bow/fast"]))), False
	if kind == 5:
		return alignment.WordMoversDistance.wmd(str(rng.choice(["nbow", "bow"]))), False
	return alignment.WordRotatorsDistance(normalize_magnitudes=bool(rng.integers(0, 2))), False


def build_session(rng, sentences=(8, 40)):
	V, d = int(rng.integers(30, 400)), int(rng.choice([32, 100, 300]))
	words = [f"w{i}" for i in range(V)]
	E = (synth.make_vocab(V, d) * rng.lognormal(0, 0.3, size=(V, 1))).astype(np.float32)
	tag_of = lambda w: TAGS[(int(w[1:]) * 7919) % len(TAGS)]
	contextual = bool(rng.integers(0, 2))
	docs = []
	for di in range(int(rng.integers(2, 6))):
		sents = []
		for _ in range(int(rng.integers(*sentences))):
			n = int(rng.integers(1, 31)) if rng.random() > 0.05 else int(rng.integers(65, 120))   # now and then a long sentence
			sents.append([words[int(i)] for i in rng.integers(0, V, size=n)])
		kw = dict(pos=[[POS_OF[tag_of(w)] for w in s] for s in sents], tags=[[tag_of(w) for w in s] for s in sents])
		n_raw = sum(len(s) for s in sents)
		if rng.random() < 0.25:
			kw["token_mask"] = rng.random(n_raw) > 0.1
		if contextual:
			ids = [int(w[1:]) for s in sents for w in s]
			X = (E[ids] + 0.1 * rng.standard_normal((n_raw, d))).astype(np.float32)
			kw["contextual_embeddings"] = {"ctx": X}
		docs.append(Document(sents, **kw))
	if contextual:
		emb = ContextualEmbedding("ctx", d, lambda tokens: E[[int(t[1:]) for t in tokens]])
		session = Session(docs, embeddings=[emb])
	else:
		emb = StaticEmbedding("toy", words, E)
		session = Session(Corpus(docs), embeddings=[emb])
	nlp = lambda text: [{"text": w, "pos": POS_OF[tag_of(w)], "tag": tag_of(w)} for w in text.split()]
	return session, emb, nlp, words


def flows_close(fx, fy, exact):
	if fy is None:
		return   # (no rows for this winner on the double)
	assert fx is not None and fx["type"] == fy["type"]
	if exact:
		for key in fx:
			if key != "type":
				assert (np.asarray(fx[key]) == np.asarray(fy[key])).all(), key
	elif fx["type"] == "dense":
		# an optimal plan need not be unique (repeated words give equal rows): same mass moved at the same cost
		assert fx["flow"].shape == fy["flow"].shape
		np.testing.assert_allclose(fx["dist"], fy["dist"], atol=2e-5)
		# (the plans themselves are not compared: two optimal plans may send a unit to different words, and a word that occurs
		# three times in the slice shows its flow at three positions -- the scores, compared above, are the plans' costs)
		assert np.isfinite(fx["flow"]).all() and (fx["flow"] >= -1e-6).all()


@pytest.mark.parametrize("seed", range(60 * SCALE))
def test_random_session_on_hip_equals_oracle_double(hip, seed):
	rng = np.random.default_rng(77000 + seed)
	session, emb, nlp, words = build_session(rng)
	strategy, is_align = random_strategy(rng)
	kw = {}
	if rng.random() < 0.3:
		kw = dict(tag_weights={t: float(rng.uniform(0.25, 2.5)) for t in rng.choice(TAGS, size=3, replace=False)},
			pos_mismatch_penalty=float(rng.uniform(0, 0.5)), similarity_threshold=float(rng.uniform(0, 0.2)))
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), strategy, **kw)
	if rng.random() < 0.7:
		part = session.partition("sentence", int(rng.integers(1, 4)), int(rng.integers(1, 3)))
	else:
		part = session.partition("token", int(rng.integers(4, 20)), int(rng.integers(1, 8)))
	n_slices = None
	index_kw = {}
	gpu = part.index(sim, nlp=nlp)
	n_slices = gpu.n_slices
	if rng.random() < 0.3:
		index_kw["saliency"] = rng.uniform(0.5, 1.5, size=n_slices).astype(np.float32)
	if seed >= 60 and seed % 5 == 0:   # (the first 60 seeds keep the sessions of the first soaks)
		index_kw["precision"] = "f32"
	if index_kw:
		gpu.close()
		gpu = part.index(sim, nlp=nlp, **index_kw)
	cpu = part.index(sim, nlp=nlp, corpus_factory=OracleCorpus, **index_kw)
	for _ in range(2):
		doc = session.documents[int(rng.integers(0, len(session.documents)))]
		len_t = int(rng.integers(1, 13)) if rng.random() < 0.8 else int(rng.integers(17, 40))
		if len(doc.tokens) > len_t and rng.random() < 0.7:
			a0 = int(rng.integers(0, len(doc.tokens) - len_t))
			text = " ".join(doc.tokens[a0:a0 + len_t])
		else:
			text = " ".join(words[int(i)] for i in rng.integers(0, len(words), size=len_t))
		options = {}
		if rng.random() < 0.25:
			options["pos_filter"] = [str(x) for x in rng.choice(["DET", "PUNCT", "ADJ"], size=int(rng.integers(1, 3)), replace=False)]
		if is_align and rng.random() < 0.2:
			options["submatch_weight"] = float(rng.choice([0.5, 1.0, 2.0]))
		if seed >= 60 and seed % 11 == 0:
			options["tag_filter"] = [str(x) for x in rng.choice(TAGS, size=int(rng.integers(1, 3)), replace=False)]
		n = int(rng.choice([1, 5, 12]))
		min_score = 0.0 if rng.random() < 0.7 else -100.0
		if seed >= 60 and seed % 7 == 0:
			n = int(rng.choice([60, 70, 150]))   # the margin of the canonical re-ranking across the selection's k <= 64 / k > 64 paths
		seen = {"hip": [], "double": []}
		exact_scores = is_align or bool((getattr(strategy, "_options", None) or {}).get("relaxed"))   # restated canonically: the oracle's floats
		hooked = seed >= 60 and seed % 6 == 0
		every = hooked and seed % 12 == 0 and "submatch_weight" not in options   # AllSlices: the hook for EVERY slice, stated chunk by chunk
		def hook_of(side):
			from vectorian_amd.index import AllSlices
			f = lambda name, data: seen[side].append((name, data))
			return AllSlices(f, chunk=int(rng.integers(1, 200))) if every else f
		a = gpu.find(text, n=n, min_score=min_score, options=dict(options, debug=hook_of("hip")) if hooked else options)
		b = cpu.find(text, n=n, min_score=min_score, options=dict(options, debug=hook_of("double")) if hooked else options)
		if hooked:
			# the debug hook (call_debug_hook, metric/alignment.h:145-173, 600-607), called for the winners: the same calls with the same data
			assert [x[0] for x in seen["hip"]] == [x[0] for x in seen["double"]], (seed, every)
			if exact_scores:
				assert len(seen["hip"]) >= len(a) if every else len(seen["hip"]) == len(a), (seed, every, len(seen["hip"]), len(a))
			for (_, x), (_, y) in zip(seen["hip"], seen["double"]):
				assert x.keys() == y.keys(), (seed, x.keys(), y.keys())
				if "slice" in x:
					assert x["slice"] == y["slice"]
				for key in ("score", "worst_score"):
					if key in x:
						assert x[key] == y[key] if exact_scores else abs(x[key] - y[key]) <= 2e-5, (seed, key, x[key], y[key])
				if "s" in x:   # the solvers' hooks of the exact transports: tokens, masses, distance matrix, plan, cost
					assert x["s"]["id"] == y["s"]["id"] and x["t"]["id"] == y["t"]["id"]
					for key in x:
						if isinstance(x[key], np.ndarray) and x[key].dtype.kind == "f" and key not in ("G", "flow_by_pos", "dist_by_pos"):   # (an optimal plan need not be unique)
							np.testing.assert_allclose(x[key], y[key], atol=2e-5, err_msg=str((seed, key)))
				if x.get("similarity") is not None and y.get("similarity") is not None:
					if not np.array_equal(x["similarity"], y["similarity"]):
						ne = x["similarity"] != y["similarity"]
						raise AssertionError("the hook's similarity matrix differs: %r" % ((seed, type(strategy).__name__, bool(kw), type(emb).__name__, index_kw.get("precision"), options,
							x["similarity"].shape, np.argwhere(ne).tolist()[:6], x["similarity"][ne][:6].tolist(), y["similarity"][ne][:6].tolist(), kw),))
		ctx = (seed, type(strategy).__name__, getattr(strategy, '_options', None), text, options, bool(kw), part.to_args(), type(emb).__name__, n, min_score, bool(index_kw))
		assert len(a) == len(b), ctx
		if is_align:
			assert [(m.doc_index, m.slice_id, m.score) for m in a] == [(m.doc_index, m.slice_id, m.score) for m in b], ctx
			for x, y in zip(a, b):
				flows_close(x.flow, y.flow, True)
			if len(a):
				assert a[0].to_json() == b[0].to_json(), ctx
		elif (getattr(strategy, "_options", None) or {}).get("relaxed"):
			# relaxed WMD: the winners' scores are restated on the host from their canonical similarity rows, in the reference's order
			# of operations (vk_transport_host.h): the oracle's floats, the oracle's result set
			assert [(m.doc_index, m.slice_id, m.score) for m in a] == [(m.doc_index, m.slice_id, m.score) for m in b], ctx
			for x, y in list(zip(a, b))[:3]:
				flows_close(x.flow, y.flow, False)
		else:
			sa, sb = np.array([m.score for m in a]), np.array([m.score for m in b])
			np.testing.assert_allclose(sa, sb, atol=2e-5, err_msg=str(ctx))
			for i, (x, y) in enumerate(zip(a, b)):
				if (x.doc_index, x.slice_id) != (y.doc_index, y.slice_id):
					assert (np.abs(sb - sb[i]) <= 2e-5).sum() > 1 or abs(sb[i] - sb[-1]) <= 2e-5, ctx   # a tie of the double's own
					continue
				if i < 3:
					flows_close(x.flow, y.flow, False)
	gpu.close(); cpu.close()


@pytest.mark.parametrize("seed", range(16 * SCALE))
def test_random_document_session_on_hip_equals_oracle_double(hip, seed):
	"""whole documents (600 .. 1,500 tokens) and windows of 30 .. 50 sentences as slices -- beyond VK_MAX_SENT_LEN: the
	one-wave-per-slice kernel with its state in global memory -- under alignments of every locality and gap family (tag weights,
	submatch weight, token filters, saliency) and the relaxed 1:1 word mover's distances: documents, slices, scores, flows and the
	JSON report of the best match equal the oracle double's with `==`"""
	rng = np.random.default_rng(55000 + seed)
	session, emb, nlp, words = build_session(rng, sentences=(40, 90))
	if rng.random() < 0.7:
		gap = random_gap(rng) if rng.random() < 0.6 else {"s": random_gap(rng), "t": random_gap(rng)}
		strategy, is_align = [alignment.LocalAlignment, alignment.GlobalAlignment, alignment.SemiGlobalAlignment][int(rng.integers(0, 3))](gap=gap), True
	else:
		strategy, is_align = alignment.WordMoversDistance.rwmd(str(rng.choice(["nbow", "bow/fast"]))), False
	kw = {}
	if is_align and rng.random() < 0.3:   # (tag-weighted vocabulary transports keyed by (id, tag) are refused over such slices)
		kw = dict(tag_weights={t: float(rng.uniform(0.25, 2.5)) for t in rng.choice(TAGS, size=3, replace=False)},
			pos_mismatch_penalty=float(rng.uniform(0, 0.5)), similarity_threshold=float(rng.uniform(0, 0.2)))
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), strategy, **kw)
	part = session.partition("document") if rng.random() < 0.6 else session.partition("sentence", int(rng.integers(30, 50)), int(rng.integers(10, 20)))
	index_kw = {}
	gpu = part.index(sim, nlp=nlp)
	if max(gpu._slice_end - gpu._slice_start) <= 512:   # (windows of short sentences)
		gpu.close()
		part = session.partition("document")
		gpu = part.index(sim, nlp=nlp)
	assert max(gpu._slice_end - gpu._slice_start) > 512
	if rng.random() < 0.3:
		index_kw["saliency"] = rng.uniform(0.5, 1.5, size=gpu.n_slices).astype(np.float32)
		gpu.close()
		gpu = part.index(sim, nlp=nlp, **index_kw)
	cpu = part.index(sim, nlp=nlp, corpus_factory=OracleCorpus, **index_kw)
	for _ in range(2):
		doc = session.documents[int(rng.integers(0, len(session.documents)))]
		len_t = int(rng.integers(1, 13)) if rng.random() < 0.7 else int(rng.integers(17, 40))
		a0 = int(rng.integers(0, len(doc.tokens) - 3 * len_t))
		text = " ".join(doc.tokens[a0:a0 + 3 * len_t:3]) if rng.random() < 0.5 else " ".join(doc.tokens[a0:a0 + len_t])
		options = {}
		if rng.random() < 0.25:
			options["pos_filter"] = [str(x) for x in rng.choice(["DET", "PUNCT", "ADJ"], size=int(rng.integers(1, 3)), replace=False)]
		if is_align and rng.random() < 0.25:
			options["submatch_weight"] = float(rng.choice([0.5, 1.0, 2.0]))
		n = int(rng.choice([1, 3, 8]))
		min_score = 0.0 if rng.random() < 0.6 else -100.0
		a = gpu.find(text, n=n, min_score=min_score, options=options)
		b = cpu.find(text, n=n, min_score=min_score, options=options)
		ctx = (seed, type(strategy).__name__, getattr(strategy, "_options", None), text, options, bool(kw), part.to_args(), type(emb).__name__, n, min_score, bool(index_kw))
		assert [(m.doc_index, m.slice_id, m.score) for m in a] == [(m.doc_index, m.slice_id, m.score) for m in b], ctx
		for x, y in zip(a, b):
			flows_close(x.flow, y.flow, is_align)
		if len(a) and is_align:
			assert a[0].to_json() == b[0].to_json(), ctx
	gpu.close(); cpu.close()


@pytest.mark.parametrize("seed", range(30 * SCALE))
def test_random_session_find_many_equals_find(hip, seed):
	"""Index.find_many -- queries in flight on handles of the resident corpus; over contextual embeddings alignments 16 per call on
	the shared pass, relaxed-WMD queries up to 256 per call on the GEMM kernels -- returns what Index.find returns, query by query:
	alignments with `==` (both restate their winners in the canonical arithmetic); transports at 2e-5 (the batched kernels sum
	their dot products in another order than the per-query kernel), the same slices unless scores tie within that."""
	rng = np.random.default_rng(99000 + seed)
	session, emb, nlp, words = build_session(rng)
	strategy, is_align = random_strategy(rng)
	kw = {}
	if rng.random() < 0.15:
		kw = dict(tag_weights={t: float(rng.uniform(0.25, 2.5)) for t in rng.choice(TAGS, size=3, replace=False)},
			pos_mismatch_penalty=float(rng.uniform(0, 0.5)), similarity_threshold=float(rng.uniform(0, 0.2)))
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), strategy, **kw)
	if rng.random() < 0.7:
		part = session.partition("sentence", int(rng.integers(1, 4)), int(rng.integers(1, 3)))
	else:
		part = session.partition("token", int(rng.integers(4, 20)), int(rng.integers(1, 8)))
	gpu = part.index(sim, nlp=nlp, **({"precision": "f32"} if seed >= 30 and seed % 5 == 0 else {}))
	texts = []
	wide = rng.random() < 0.2
	for _ in range(int(rng.integers(2, 24))):
		doc = session.documents[int(rng.integers(0, len(session.documents)))]
		len_t = int(rng.integers(17, 40)) if wide and rng.random() < 0.5 else int(rng.integers(1, 17))
		if len(doc.tokens) > len_t and rng.random() < 0.7:
			a0 = int(rng.integers(0, len(doc.tokens) - len_t))
			texts.append(" ".join(doc.tokens[a0:a0 + len_t]))
		else:
			texts.append(" ".join(words[int(i)] for i in rng.integers(0, len(words), size=len_t)))
	options = {}
	if rng.random() < 0.15:
		options["pos_filter"] = ["DET"]
	if is_align and rng.random() < 0.2:
		options["submatch_weight"] = float(rng.choice([0.5, 1.0, 2.0]))
	n = int(rng.choice([1, 5, 12]))
	min_score = 0.0 if rng.random() < 0.7 else -100.0
	if seed >= 30 and seed % 7 == 0:
		n = int(rng.choice([60, 70, 150]))
	many = gpu.find_many(texts, n=n, min_score=min_score, options=options, in_flight=int(rng.integers(1, 4)))
	assert len(many) == len(texts)
	for text, a in zip(texts, many):
		b = gpu.find(text, n=n, min_score=min_score, options=options)
		ctx = (seed, type(strategy).__name__, getattr(strategy, '_options', None), text, options, bool(kw), part.to_args(), type(emb).__name__, n, min_score)
		assert len(a) == len(b), ctx
		if is_align:
			assert [(m.doc_index, m.slice_id, m.score) for m in a] == [(m.doc_index, m.slice_id, m.score) for m in b], ctx
			for x, y in zip(a, b):
				flows_close(x.flow, y.flow, True)
		elif (getattr(strategy, "_options", None) or {}).get("relaxed"):
			assert [(m.doc_index, m.slice_id, m.score) for m in a] == [(m.doc_index, m.slice_id, m.score) for m in b], ctx   # canonical on both paths
			for x, y in zip(a, b):
				flows_close(x.flow, y.flow, True)
		else:
			sa, sb = np.array([m.score for m in a]), np.array([m.score for m in b])
			np.testing.assert_allclose(sa, sb, atol=2e-5, err_msg=str(ctx))
			for i, (x, y) in enumerate(zip(a, b)):
				if (x.doc_index, x.slice_id) != (y.doc_index, y.slice_id):
					assert (np.abs(sb - sb[i]) <= 2e-5).sum() > 1 or abs(sb[i] - sb[-1]) <= 2e-5, ctx
	gpu.close()
