"""CPU tier: the Python host surface (Session / Partition / Index.find / option dicts)
with the oracle-backed test double in place of the HIP corpus (tests/fake_backend.py).
BASELINE configs[0] -- the plumbing case: 5-token query over a 1k-sentence toy corpus,
static 300-d embedding, cosine + Waterman-Smith-Beyer."""

import numpy as np
import pytest

from fake_backend import OracleCorpus
from vectorian_amd import alignment, core, synth
from vectorian_amd.corpus import Corpus, Document
from vectorian_amd.embedding import ContextualEmbedding, StaticEmbedding, Vectors
from vectorian_amd.index import HipBruteForceIndex
from vectorian_amd.session import Session
from vectorian_amd.sim import CosineSim, EmbeddingTokenSim, OptimizedSpanSim


def test_option_dicts_match_reference_literals():
	# vectorian/alignment.py:31-38, 96-97, 129-130, 186-187
	g = alignment.smooth_gap_cost(5)
	for cls, loc in ((alignment.LocalAlignment, core.Locality.LOCAL), (alignment.GlobalAlignment, core.Locality.GLOBAL),
			(alignment.SemiGlobalAlignment, core.Locality.SEMIGLOBAL)):
		assert cls(gap=g).to_args(None) == {"algorithm": "pyalign", "options": {"locality": loc, "gap_cost": g}}
	with pytest.raises(ValueError):
		alignment.LocalAlignment(gap={"x": g})
	# vectorian/alignment.py:231-239, 275-283
	assert alignment.WordMoversDistance.rwmd("nbow").to_args(None) == {
		"algorithm": "word-movers-distance", "relaxed": True, "injective": True, "symmetric": True,
		"normalize_bow": True, "extra_mass_penalty": -1}
	assert alignment.WordMoversDistance.wmd("nbow").to_args(None)["relaxed"] is False
	assert alignment.WordMoversDistance.rwmd("bow/fast").to_args(None)["symmetric"] is False
	# vectorian/alignment.py:308-313
	assert alignment.WordRotatorsDistance().to_args(None) == {
		"algorithm": "word-rotators-distance", "normalize_magnitudes": True, "extra_mass_penalty": -1}


def test_gap_cost_families():
	assert alignment.ConstantGapCost(0).to_special_case() == {"linear": 0.0}
	assert alignment.ConstantGapCost(0.3).to_special_case() == {}
	assert list(alignment.ConstantGapCost(0.3).costs(4)) == [0, np.float32(0.3), np.float32(0.3), np.float32(0.3)]
	assert alignment.LinearGapCost(0.1).to_special_case() == {"linear": 0.1}
	assert alignment.AffineGapCost(0.2, 0.05).to_special_case() == {"affine": (0.2, 0.05)}
	assert alignment.AffineGapCost(0.2, 0.05).costs(3)[0] == 0
	e = alignment.smooth_gap_cost(5)
	assert e.to_tuple() == ("exponential", 2.0, 0.2)            # vectorian/interact.py:559-565
	assert float(e.costs(3)[1]) == 0.12944944202899933           # mkdocs/docs/introduction.md:174
	assert abs(e(5) - 0.5) < 1e-7
	assert alignment.smooth_gap_cost(0).to_special_case() == {"linear": 0.0}


def toy_session(n_docs=20, sents_per_doc=50, V=5000, d=300, seed=1):
	rng = np.random.default_rng(seed)
	E = synth.make_vocab(V, d)
	words = [f"w{i}" for i in range(V)]
	docs = []
	for di in range(n_docs):
		sents = []
		for _ in range(sents_per_doc):
			n = int(rng.integers(4, 41))
			sents.append([words[i] for i in synth.zipf_ids(n, V, rng)])
		docs.append(Document(sents, unique_id=f"doc{di}", metadata={"title": f"doc {di}"}))
	emb = StaticEmbedding("toy-300", words, E)
	return Session(Corpus(docs), embeddings=[emb]), emb, words, rng


def test_find_over_toy_corpus_static_wsb(oracle):
	session, emb, words, rng = toy_session()
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)))
	index = session.partition("sentence").index(sim, corpus_factory=OracleCorpus)
	assert isinstance(index, HipBruteForceIndex)
	assert index.n_slices == 1000
	# plant: 5 consecutive tokens of sentence 7 of document 3
	doc = session.documents[3]
	st, en = doc.spans["sentence"]["start"][7], doc.spans["sentence"]["end"][7]
	planted = doc.tokens[st + 1:st + 6]
	result = index.find(" ".join(planted), n=5)
	assert len(result) == 5
	top = result[0]
	assert top.doc_index == 3 and top.slice_id == 7
	assert abs(top.score - 1.0) < 1e-2
	assert list(top.flow["target"]) == [1, 2, 3, 4, 5]
	assert top.flow["type"] == "injective" and top.flow["target"].dtype == np.int16
	scores = [m.score for m in result]
	assert scores == sorted(scores, reverse=True)
	j = top.to_json()
	assert j["slice"] == 7 and j["metric"] == "toy-300-cosine" and j["level"] == "word"
	assert j["location"] == {"start": int(st), "end": int(en)}
	matched = [r for r in j["regions"] if "edges" in r]
	assert [r["s"] for r in matched] == planted
	assert all(abs(r["edges"][0]["distance"]) < 1e-2 for r in matched)
	assert result.duration >= 0

	# independent check of the whole pipeline: vocabulary table + gather + WSB in the oracle
	Eb = synth.to_bf16_bits(Vectors(emb.encode_tokens(session.vocab.tokens).unmodified).normalized)
	ids = np.concatenate([session.doc_token_ids(i) for i in range(len(session.documents))])
	qids = np.array([session.vocab.token_to_id(t) for t in planted], dtype=np.int32)
	Qb = synth.to_bf16_bits(emb.encode_tokens(planted).normalized)
	w = alignment.smooth_gap_cost(5).costs(65)
	ref = oracle.find(layout=oracle.LAYOUT_STATIC, d=300, sent_off=index._sent_off, tok_id=ids, E=Eb, Q=Qb, q_ids=qids,
		gap_s=("table", w), gap_t=("table", w), max_matches=5)
	assert [int(index._slice_doc[g]) for g in ref["sentence"]] == [m.doc_index for m in result]
	np.testing.assert_allclose([m.score for m in result], ref["score"], atol=1e-6)


def test_find_many_equals_find():
	session, emb, words, rng = toy_session(n_docs=4, sents_per_doc=30, V=400, d=48)
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)))
	index = session.index(sim, corpus_factory=OracleCorpus)
	texts = [" ".join(session.documents[i % 4].tokens[7 * i:7 * i + 4 + i % 3]) for i in range(7)] + [""]
	many = index.find_many(texts, n=5)
	assert len(many) == len(texts) and many[-1].matches == []
	for text, res in zip(texts, many):
		one = index.find(text, n=5)
		assert [(m.doc_index, m.slice_id, m.score) for m in res] == [(m.doc_index, m.slice_id, m.score) for m in one]
		assert all((a.flow["target"] == b.flow["target"]).all() for a, b in zip(res, one))
	index.close()


def test_regions_report_gap_penalties_like_the_reference_docs():
	# the shape of mkdocs/docs/introduction.md:150-184: one skipped document token between
	# the 2nd and 3rd match costs gap_cost_s(1) = 1 - 2^(-1/5)
	words = ["get", "our", "jewels", "and", "wealth", "together", "jewelry", "riches", "x"]
	rng = np.random.default_rng(0)
	base = rng.standard_normal((len(words), 64)).astype(np.float32)
	base[6] = base[2] + 0.3 * rng.standard_normal(64)     # jewelry ~ jewels
	base[7] = base[4] + 0.3 * rng.standard_normal(64)     # riches ~ wealth
	emb = StaticEmbedding("toy", words, base)
	doc = Document([["get", "our", "jewels", "and", "our", "wealth", "together", "x"]], metadata={"author": "ws"})
	session = Session([doc], embeddings=[emb])
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)))
	index = session.index(sim, corpus_factory=OracleCorpus)
	m = index.find("jewelry and riches", n=1)[0]
	assert list(m.flow["target"]) == [2, 3, 5]
	j = m.to_json(context_size=2)
	reg = j["regions"]
	assert reg[0] == {"s": "get our", "gap_penalty": 0.0}
	assert reg[1]["s"] == "jewels" and reg[1]["edges"][0]["t"]["text"] == "jewelry"
	assert reg[2]["s"] == "and" and reg[2]["edges"][0]["distance"] < 1e-6
	assert reg[3]["s"] == "our" and abs(reg[3]["gap_penalty"] - 0.12944944202899933) < 1e-9
	assert reg[4]["s"] == "wealth" and reg[4]["edges"][0]["t"]["index"] == 2
	assert reg[5] == {"s": "together", "gap_penalty": 0.0}
	d = [reg[i]["edges"][0]["distance"] for i in (1, 2, 4)]
	want = ((1 - d[0]) + (1 - d[1]) - 0.12944944202899933 + (1 - d[2])) / 3
	assert abs(m.score - want) < 1e-6
	assert j["omitted"] == []


def test_pos_filter_drops_document_tokens_for_one_query():
	# options pos_filter / tag_filter (Query::make_token_filter, vectorian/core/cpp/query.cpp:220-228): tokens with
	# these POS / tags are removed from every slice for this query; flow targets count the passing tokens, regions
	# are reported in document positions (Flow::py_regions index_map, match/flow.cpp:49-60,96-97)
	words = ["get", "our", "jewels", "and", "wealth", "together", "jewelry", "riches", "x"]
	rng = np.random.default_rng(0)
	base = rng.standard_normal((len(words), 64)).astype(np.float32)
	base[6] = base[2] + 0.3 * rng.standard_normal(64)
	base[7] = base[4] + 0.3 * rng.standard_normal(64)
	emb = StaticEmbedding("toy", words, base)
	sent = ["get", "our", "jewels", "and", "our", "wealth", "together", "x"]
	pos = ["VERB", "PRON", "NOUN", "CCONJ", "PRON", "NOUN", "ADV", "X"]
	tags = ["VB", "PRP$", "NNS", "CC", "PRP$", "NN", "RB", "XX"]
	session = Session([Document([sent, ["x", "x"]], pos=[pos, ["X", "X"]], tags=[tags, ["XX", "XX"]])], embeddings=[emb])
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)))
	index = session.index(sim, corpus_factory=OracleCorpus)
	plain = index.find("jewelry and riches", n=1)[0]
	assert list(plain.flow["target"]) == [2, 3, 5]
	# without the pronouns the slice reads get jewels and wealth together x: no gap left between "and" and "wealth"
	m = index.find("jewelry and riches", n=1, options={"pos_filter": ["PRON"]})[0]
	assert list(m.flow["target"]) == [1, 2, 3]
	assert abs(m.score - (plain.score + 0.12944944202899933 / 3)) < 1e-6
	reg = m.to_json(context_size=2)["regions"]
	assert [r["s"] for r in reg if "edges" in r] == ["jewels", "and", "wealth"]
	# tag filter, both at once, and the cache of filtered corpora
	m2 = index.find("jewelry and riches", n=1, options={"tag_filter": ["PRP$", "CC"]})[0]
	assert list(m2.flow["target"]) == [1, -1, 2] and m2.omitted == ["and"]
	m3 = index.find("jewelry and riches", n=1, options={"pos_filter": ["PRON"], "tag_filter": ["CC"]})[0]
	assert list(m3.flow["target"]) == [1, -1, 2]
	assert [r["s"] for r in m3.to_json(context_size=1)["regions"] if "edges" in r] == ["jewels", "wealth"]
	assert len(index._filtered) == 2
	again = index.find("jewelry and riches", n=1, options={"pos_filter": ["PRON"], "tag_filter": ["CC"]})[0]
	assert again.score == m3.score
	with pytest.raises(RuntimeError, match="illegal value"):
		index.find("jewelry", options={"pos_filter": ["NO_SUCH_POS"]})      # query.h:45-49
	index.close()


def test_contextual_embedding_and_boost():
	rng = np.random.default_rng(4)
	d = 48
	table = {}
	def vec(tok):
		if tok not in table:
			table[tok] = rng.standard_normal(d).astype(np.float32)
		return table[tok]
	docs = []
	for di in range(3):
		sents = [[f"t{int(rng.integers(0, 30))}" for _ in range(int(rng.integers(2, 9)))] for _ in range(10)]
		toks = [t for s in sents for t in s]
		X = np.stack([vec(t) + 0.05 * rng.standard_normal(d).astype(np.float32) for t in toks])
		docs.append(Document(sents, contextual_embeddings={"ctx": X}))
	emb = ContextualEmbedding("ctx", d, lambda tokens: np.stack([vec(t) for t in tokens]))
	session = Session(docs, embeddings=[emb])
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.GlobalAlignment(gap=alignment.LinearGapCost(0.1)))
	boost = np.ones(30, dtype=np.float32)
	boost[17] = 50.0
	index = session.index(sim, corpus_factory=OracleCorpus, saliency=boost)
	q = " ".join(docs[1].tokens[3:6])
	r = index.find(q, n=3, min_score=-100.0)
	assert r[0].doc_index == 1 and r[0].slice_id == 7          # slice 17 of the session = doc 1, sentence 7
	assert len(r) == 3


def contextual_toy(n_docs=4, sents=40, d=48, seed=4, uniform=None):
	rng = np.random.default_rng(seed)
	table = {}
	def vec(tok):
		if tok not in table:
			table[tok] = rng.standard_normal(d).astype(np.float32)
		return table[tok]
	docs = []
	for di in range(n_docs):
		ss = [[f"t{int(rng.integers(0, 60))}" for _ in range(uniform or int(rng.integers(2, 12)))] for _ in range(sents)]
		toks = [t for s_ in ss for t in s_]
		X = np.stack([vec(t) + 0.05 * rng.standard_normal(d).astype(np.float32) for t in toks])
		docs.append(Document(ss, contextual_embeddings={"ctx": X}))
	emb = ContextualEmbedding("ctx", d, lambda tokens: np.stack([vec(t) for t in tokens]))
	return Session(docs, embeddings=[emb]), emb, docs


@pytest.mark.parametrize("strategy", ["local", "rwmd"])
def test_find_many_shares_calls_over_contextual_embeddings(strategy):
	# queries with common options over a contextual embedding go to the backend several per call (query_batch); every Result as from find()
	session, emb, docs = contextual_toy()
	al = alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)) if strategy == "local" else alignment.WordMoversDistance.rwmd("nbow")
	index = session.index(OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), al), corpus_factory=OracleCorpus)
	texts = [" ".join(docs[i % 4].tokens[5 * i:5 * i + 3 + i % 4]) for i in range(21)] + [""]
	many = index.find_many(texts, n=5)
	assert index._corpus.batch_calls >= 1 and len(many) == len(texts) and many[-1].matches == []
	for text, res in zip(texts, many):
		one = index.find(text, n=5)
		assert [(m.doc_index, m.slice_id, m.score) for m in res] == [(m.doc_index, m.slice_id, m.score) for m in one]
		for a, b in zip(res, one):
			assert a.to_json()["regions"] == b.to_json()["regions"]
	calls = index._corpus.batch_calls
	assert len(index.find_many(texts, n=5, batch=False)) == len(texts) and index._corpus.batch_calls == calls   # never
	wrd = session.index(OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.WordRotatorsDistance()), corpus_factory=OracleCorpus)
	with pytest.raises(RuntimeError):
		wrd.find_many(texts[:3], n=5, batch=True)    # exact transport does not share calls
	assert len(wrd.find_many(texts[:3], n=5)) == 3
	index.close(); wrd.close()


def test_transport_strategies_over_static_embedding():
	"""WordRotatorsDistance and the named RWMD variants through Index.find on a static embedding (vocabulary magnitudes)."""
	session, emb, words, rng = toy_session(n_docs=2, sents_per_doc=40, V=120, d=24)
	ts = EmbeddingTokenSim(emb, CosineSim())
	doc = session.documents[1]
	st = doc.spans["sentence"]["start"][5]
	planted = " ".join(doc.tokens[st:st + 4])
	for strategy in (alignment.WordRotatorsDistance(), alignment.WordRotatorsDistance(normalize_magnitudes=False),
			alignment.WordMoversDistance.rwmd("nbow"), alignment.WordMoversDistance.rwmd("nbow/distributed"),
			alignment.WordMoversDistance.rwmd("bow/fast")):
		index = session.index(OptimizedSpanSim(ts, strategy), corpus_factory=OracleCorpus)
		result = index.find(planted, n=3, min_score=-10.0, options={"submatch_weight": 1.0})   # transport: reference stays len_t
		assert len(result) == 3
		scores = [m.score for m in result]
		assert scores == sorted(scores, reverse=True)
		if isinstance(strategy, alignment.WordRotatorsDistance):
			assert (1, 5) in [(m.doc_index, m.slice_id) for m in result]
		# flows as the reference states them (match/flow.cpp:226-290): dense for exact transport, sparse for the relaxed one
		flow = result[0].flow
		len_t = len(planted.split())
		if isinstance(strategy, alignment.WordRotatorsDistance):
			assert flow["type"] == "dense" and flow["flow"].shape[0] == len_t
			np.testing.assert_allclose(flow["flow"].sum(axis=1), 1.0, atol=1e-4)   # every query token ships all of its mass
		else:
			assert flow["type"] == "sparse" and set(flow["source"]) <= set(range(len_t))
			assert len(flow["source"]) == len(flow["target"]) == len(flow["flow"]) == len(flow["dist"]) >= 1
		j = result[0].to_json()
		assert any("edges" in r for r in j["regions"])


def test_f32_precision_index(oracle):
	"""precision="f32": the index keeps fp32 unit vectors (the reference's own precision); scores move by a few 1e-4 at most"""
	session, emb, words, rng = toy_session(n_docs=3, sents_per_doc=40, V=300, d=32)
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)))
	doc = session.documents[2]
	st = doc.spans["sentence"]["start"][9]
	query = " ".join(doc.tokens[st:st + 6])
	res = {}
	for prec in ("bf16", "f32"):
		index = session.partition("sentence").index(sim, corpus_factory=OracleCorpus, precision=prec)
		res[prec] = index.find(query, n=8)
	assert [(m.doc_index, m.slice_id) for m in res["f32"]][:1] == [(2, 9)]
	a, b = np.array([m.score for m in res["bf16"]]), np.array([m.score for m in res["f32"]])
	assert 0 < np.abs(a - b).max() < 3e-3


def test_unsupported_options_are_explicit():
	session, emb, words, rng = toy_session(n_docs=1, sents_per_doc=3, V=50, d=16)
	ts = EmbeddingTokenSim(emb, CosineSim())
	with pytest.raises(TypeError):
		OptimizedSpanSim("not a token sim")
	index = session.index(OptimizedSpanSim(ts, tag_weights={"NN": 2.0}), corpus_factory=OracleCorpus)
	with pytest.raises(RuntimeError):
		index.find("w1 w2")          # documents without pos / tags
	index = session.index(OptimizedSpanSim(ts, alignment.WordMoversDistance.wmd("nbow")), corpus_factory=OracleCorpus)
	assert len(index.find("w1 w2", n=2)) == 2          # full WMD runs (exact EMD)
	index = session.index(OptimizedSpanSim(ts), corpus_factory=OracleCorpus)
	with pytest.raises(RuntimeError):
		index.find("w1 w2", options={"no_such_option": 1})
	assert index.find("", n=3).matches == []


def test_sliding_windows_overlap():
	# Partition(level, window_size=3, window_step=1): Spans::iterate (vectorian/core/cpp/document.h:147-169)
	# yields overlapping slices [start(i), end(min(i+2, n-1)))
	session, emb, words, rng = toy_session(n_docs=2, sents_per_doc=6, V=200, d=32)
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.LocalAlignment(gap=alignment.LinearGapCost(0.05)))
	part = session.partition("sentence", 3, 1)
	index = part.index(sim, corpus_factory=OracleCorpus)
	assert index.n_slices == 12 and index._sent_off is None
	doc = session.documents[1]
	st, en = doc.spans["sentence"]["start"], doc.spans["sentence"]["end"]
	# a query spanning the end of sentence 2 and the start of sentence 3 of document 1
	q = doc.tokens[en[2] - 2:en[2] + 2]
	if len(doc.tokens[st[1]:en[3]]) <= 64:
		r = index.find(" ".join(q), n=4)
		assert r[0].doc_index == 1 and r[0].slice_id in (1, 2, 3)
		assert abs(r[0].score - 1.0) < 1e-2
		assert r[0].to_json()["location"]["start"] == st[r[0].slice_id]


def test_document_partition():
	"""session.partition("document") (mkdocs/docs/documents.md:37): the one span the importers give every document
	(vectorian/importers.py:30-36, 220-223) -- whole documents as the slices of an index, here ~ 700 tokens each (beyond
	VK_MAX_SENT_LEN: the gap tables and the winners' similarity rows are sized by the longest document)"""
	session, emb, words, rng = toy_session(n_docs=5, sents_per_doc=32, V=300, d=32)
	for doc in session.documents:
		assert list(doc.spans["document"]["start"]) == [0] and list(doc.spans["document"]["end"]) == [doc.n_tokens]
	assert session.max_len("document", 1) == max(d.n_tokens for d in session.documents) > core.VK_MAX_SENT_LEN
	doc = session.documents[3]
	st = doc.spans["sentence"]["start"][17]
	planted = " ".join(doc.tokens[st:st + 6])
	for strategy in (alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)), alignment.WordMoversDistance.rwmd("nbow")):
		index = session.partition("document").index(OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), strategy), corpus_factory=OracleCorpus)
		assert index.n_slices == 5 and index.corpus._longest() == session.max_len("document", 1)
		top = index.find(planted, n=3)[0]
		assert (top.doc_index, top.slice_id) == (3, 0)
		if isinstance(strategy, alignment.LocalAlignment):
			assert abs(top.score - 1.0) < 1e-2 and list(top.flow["target"]) == list(range(st, st + 6))   # token positions in the document
			j = top.to_json()
			assert j["location"] == {"start": 0, "end": doc.n_tokens} and [r["s"] for r in j["regions"] if "edges" in r] == doc.tokens[st:st + 6]
		else:
			assert top.flow["type"] == "sparse" and len(top.flow["target"]) >= 6   # (the symmetric form: edges of both directions)
	assert core.winner_rows(700) == 704 and core.winner_rows(40000) == core.VK_MAX_DOC_LEN + 1 and core.winner_rows(3) == core.VK_FAST_SENT_LEN
	with pytest.raises(ValueError):
		Document([["a"]], spans={"document": {"start": [0], "end": [1]}})   # built in, like "sentence" and "token"


def test_token_mask_and_span_levels():
	# PreparedDocument (vectorian/corpus/document.py:626-662): masked tokens disappear from the token table, every span
	# table is re-indexed with the cumulative sum of the mask; "token" is a partition level of its own (document.cpp:52-53)
	rng = np.random.default_rng(12)
	words = [f"w{i}" for i in range(40)]
	emb = StaticEmbedding("toy", words, rng.standard_normal((40, 32)).astype(np.float32))
	sents = [[words[int(i)] for i in rng.integers(0, 40, size=int(n))] for n in (6, 9, 4, 7)]
	n_raw = sum(len(s) for s in sents)
	mask = rng.random(n_raw) > 0.3
	ctx = rng.standard_normal((n_raw, 8)).astype(np.float32)
	masked = Document(sents, token_mask=mask, contextual_embeddings={"c": ctx},
		spans={"stanza": {"start": [0, 15], "end": [15, n_raw], "label": ["a", "b"]}})
	# the same document written down without the dropped tokens
	it = iter(mask)
	kept = [[t for t in s if next(it)] for s in sents]
	plain = Document(kept)
	assert masked.tokens == plain.tokens and masked.n_tokens == int(mask.sum())
	for k in ("start", "end"):
		assert (masked.spans["sentence"][k] == plain.spans["sentence"][k]).all()
	assert (masked.contextual_vectors("c") == ctx[mask]).all()
	assert masked.spans["stanza"]["end"][-1] == masked.n_tokens and list(masked.spans["stanza"]["label"]) == ["a", "b"]
	assert masked.spans["stanza"]["start"][1] == int(mask[:15].sum())
	assert (masked.spans["token"]["end"] - masked.spans["token"]["start"] == 1).all()
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.LocalAlignment(gap=alignment.LinearGapCost(0.1)))
	q = " ".join(plain.tokens[5:9])
	res = [Session([d], embeddings=[emb]).index(sim, corpus_factory=OracleCorpus).find(q, n=4) for d in (masked, plain)]
	assert [(m.slice_id, m.score) for m in res[0]] == [(m.slice_id, m.score) for m in res[1]]
	# windows of 5 tokens every 2 tokens, and the two stanzas as slices
	session = Session([masked], embeddings=[emb])
	index = session.partition("token", 5, 2).index(sim, corpus_factory=OracleCorpus)
	assert index.n_slices == (masked.n_tokens + 1) // 2
	top = index.find(q, n=1)[0]
	assert top.slice_id == 4 and abs(top.score - 1.0) < 1e-2       # tokens 5..8 lie inside the window [4, 9) only
	index = session.partition("stanza").index(sim, corpus_factory=OracleCorpus)
	assert index.n_slices == 2 and index.find(q, n=1)[0].slice_id == 0
	with pytest.raises(ValueError):
		Document(sents, spans={"token": {"start": [0], "end": [1]}})


def test_tag_weighted_metric():
	# 'alignment-tag-weighted' (vectorian/sim/span.py:63-71): noun matches count double, a POS mismatch is
	# penalised, and the score is divided by the sum of the query's tag weights
	words = ["dog", "cat", "runs", "sleeps", "the", "hound", "walks"]
	rng = np.random.default_rng(3)
	vec = rng.standard_normal((len(words), 48)).astype(np.float32)
	vec[5] = vec[0] + 0.2 * rng.standard_normal(48)       # hound ~ dog
	vec[6] = vec[2] + 0.2 * rng.standard_normal(48)       # walks ~ runs
	emb = StaticEmbedding("toy", words, vec)
	sents = [["the", "hound", "walks"], ["the", "cat", "sleeps"], ["runs", "dog", "the"]]
	pos = [["DET", "NOUN", "VERB"], ["DET", "NOUN", "VERB"], ["NOUN", "VERB", "DET"]]       # last sentence mis-tagged on purpose
	tags = [["DT", "NN", "VBZ"], ["DT", "NN", "VBZ"], ["NN", "VBZ", "DT"]]
	session = Session([Document(sents, pos=pos, tags=tags)], embeddings=[emb])
	nlp = lambda text: [{"text": w, "pos": p, "tag": t} for w, p, t in zip(text.split(), ["NOUN", "VERB"], ["NN", "VBZ"])]
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.LocalAlignment(gap=alignment.LinearGapCost(0.2)),
		tag_weights={"NN": 2.0}, pos_mismatch_penalty=0.5, similarity_threshold=0.1)
	assert sim.to_args(type("I", (), {"partition": None})())["metric"] == "alignment-tag-weighted"
	index = session.index(sim, nlp=nlp, corpus_factory=OracleCorpus)
	r = index.find("dog runs", n=3)
	assert r[0].slice_id == 0
	d = r[0].flow["dist"]                       # unmodified distances (ScoreComputer, metric/alignment.h:339)
	want = (2.0 * (1 - d[0]) + 1.0 * (1 - d[1])) / 3.0
	assert abs(r[0].score - want) < 1e-6
	# sentence 2 holds the exact words but in swapped order and with mismatching POS: one match at half weight
	s2 = [m for m in r if m.slice_id == 2][0]
	assert abs(s2.score - max(2.0 * 0.5, 1.0 * 0.5) / 3.0) < 1e-6


def test_span_embedding_index():
	# EmbeddedSpanSim -> one cosine per span + top-k (vectorian/index.py:679-731), on the alignment kernels
	from vectorian_amd.sim import EmbeddedSpanSim, SpanEmbedding
	rng = np.random.default_rng(8)
	words = [f"w{i}" for i in range(50)]
	wv = dict((w, rng.standard_normal(24).astype(np.float32)) for w in words)
	enc = lambda texts: np.stack([np.mean([wv[t] for t in x.split()], axis=0) for x in texts])
	docs = [Document([[words[int(i)] for i in rng.integers(0, 50, size=int(rng.integers(2, 9)))] for _ in range(12)]) for _ in range(3)]
	session = Session(docs, embeddings=[])
	index = session.partition("sentence").index(EmbeddedSpanSim(SpanEmbedding("mean", 24, enc)), corpus_factory=OracleCorpus)
	target = docs[2].span_tokens("sentence", 5)
	r = index.find(" ".join(target), n=3)
	assert r[0].prepared_doc is docs[2] and r[0].slice_id == 5 and abs(r[0].score - 1.0) < 1e-2
	assert r[0].level == "span" and r[0].to_json()["regions"][0]["s"] == " ".join(target)
	# brute-force reference
	allv = enc([" ".join(d.span_tokens("sentence", i)) for d in docs for i in range(12)])
	qv = enc([" ".join(target)])[0]
	cos = (allv @ qv) / (np.linalg.norm(allv, axis=1) * np.linalg.norm(qv))
	assert [int(np.argsort(-cos)[i]) for i in range(3)] == [12 * docs.index(m.prepared_doc) + m.slice_id for m in r]


def test_debug_hook_and_abort():
	"""options['debug'] (Index.find(debug=...), vectorian/index.py:468-469; call_debug_hook, metric/alignment.h:145-173): the hook
	is called with the reference's keys for the winners; Query.abort (module.cpp:120) ends a search before its device work"""
	session, emb, words, rng = toy_session()
	doc = session.documents[2]
	text = " ".join(doc.tokens[30:35])
	calls = []
	hook = lambda name, data: calls.append((name, data))
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.LocalAlignment(gap=alignment.LinearGapCost(0.2)))
	index = session.partition("sentence").index(sim, corpus_factory=OracleCorpus)
	r = index.find(text, n=4, debug=hook)
	assert [c[0] for c in calls] == ["alignment"] * 4
	for m, (_, data) in zip(r, calls):
		assert set(data) == {"slice", "similarity", "flow", "score"}
		assert data["slice"] == m.slice_id and abs(data["score"] - m.raw_score) < 1e-6
		S = data["similarity"]
		assert S.shape == (m._len_s, 5) and (data["flow"]["target"] == m.flow["target"]).all()
		for j, i in enumerate(m.flow["target"]):
			if i >= 0:
				assert abs((1.0 - S[i, j]) - m.flow["dist"][j]) < 1e-6
	with pytest.raises(TypeError):
		index.find(text, n=4, debug="not callable")
	calls.clear()
	wmd = session.partition("sentence").index(OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.WordMoversDistance.rwmd("nbow")),
		corpus_factory=OracleCorpus)
	r = wmd.find(text, n=3, debug=hook)
	assert [c[0] for c in calls] == ["alignment/word-movers-distance/make"] * 3
	assert abs(calls[0][1]["score"] - r[0].score) < 1e-7 and abs(calls[0][1]["worst_score"] - r[-1].score) < 1e-7
	# abort: raised before the search starts -> no matches; a fresh query is unaffected
	q = index.make_query(text, n=4)
	q.abort()
	assert q.aborted and index._find(q) == []
	assert len(index.find(text, n=4)) == 4


def test_debug_hook_for_every_slice_and_progress():
	"""debug = AllSlices(hook): the hook contract of the reference in full -- one call per scored slice, in slice order, with the
	keys of call_debug_hook (vectorian/core/cpp/metric/alignment.h:145-173); progress(done / total) per stage of a find and per
	query / chunk of find_many (vectorian/index.py:541-558)"""
	from vectorian_amd.index import AllSlices
	session, emb, words, rng = toy_session(n_docs=3, sents_per_doc=25, V=300, d=32)
	doc = session.documents[1]
	text = " ".join(doc.tokens[12:16])
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.LocalAlignment(gap=alignment.LinearGapCost(0.2)))
	index = session.partition("sentence").index(sim, corpus_factory=OracleCorpus)
	calls, seen = [], []
	hook = AllSlices(lambda name, data: calls.append((name, data)), chunk=20)
	matches = index._find(index.make_query(text, n=4, debug=hook), progress=seen.append)
	assert seen == [1.0, 1.0]                      # one GPU: all tokens scored, then the result set complete
	assert len(calls) == 75 and [c[0] for c in calls] == ["alignment"] * 75
	assert [c[1]["slice"] for c in calls] == [s for _ in range(3) for s in range(25)]   # every slice, document by document, in order
	by_slice = {}
	for k, (_, data) in enumerate(calls):
		assert set(data) == {"slice", "similarity", "flow", "score"}
		by_slice[(k // 25, data["slice"])] = data
	for m in matches:                              # the winners' data as the winners report it
		data = by_slice[(m.doc_index, m.slice_id)]
		assert abs(data["score"] - m.raw_score) < 1e-6 and (data["flow"]["target"] == m.flow["target"]).all()
		assert data["similarity"].shape == (m._len_s, 4)
	best = max(d["score"] for d in by_slice.values())
	assert abs(best - matches[0].raw_score) < 1e-6
	# relaxed WMD: score and the running worst score of a result set filled in slice order (metric/alignment.h:600-607)
	calls.clear()
	wmd = session.partition("sentence").index(OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.WordMoversDistance.rwmd("nbow")),
		corpus_factory=OracleCorpus)
	r = wmd.find(text, n=3, debug=hook)
	assert len(calls) == 75 and calls[0][0] == "alignment/word-movers-distance/make"
	assert calls[0][1]["worst_score"] == 0.0 and calls[-1][1]["worst_score"] <= r[-1].score + 1e-7
	admitted = sorted((c[1]["score"] for c in calls if c[1]["score"] > c[1]["worst_score"]), reverse=True)[:3]
	assert np.allclose(admitted, [m.score for m in r], atol=1e-7)
	# find_many: progress after every query (pipelined path) and after every chunk (batched calls)
	seen.clear()
	res = index.find_many([text] * 5, n=2, progress=seen.append, batch=False)
	assert len(res) == 5 and seen == [0.2, 0.4, 0.6, 0.8, 1.0]
	# static embeddings share calls too since round 4 (one table over the vocabulary per call): one chunk here, the same results
	seen.clear()
	shared = index.find_many([text] * 5, n=2, progress=seen.append)
	assert seen == [1.0] and [[(m.doc_index, m.slice_id, m.score) for m in r] for r in shared] == [[(m.doc_index, m.slice_id, m.score) for m in r] for r in res]


def test_matches_are_lazy():
	"""a match holds its place in the result set's arrays; flows, regions and index maps are stated on first access (CoreMatch,
	vectorian/index.py:295-379)"""
	session, emb, docs = contextual_toy()
	index = session.index(OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.WordMoversDistance.rwmd("nbow")), corpus_factory=OracleCorpus)
	stated = []
	orig = index._transport_flow
	index._transport_flow = lambda *a, **k: (stated.append(1), orig(*a, **k))[1]
	res = index.find_many([" ".join(docs[i % 4].tokens[4 * i:4 * i + 4]) for i in range(12)], n=5)
	assert sum(len(r) for r in res) == 60 and stated == []          # nothing stated yet
	scores = [m.score for r in res for m in r]                     # plain fields: still nothing
	assert stated == [] and len(scores) == 60
	f = res[3][0].flow
	assert f["type"] == "sparse" and len(stated) == 1
	assert res[3][0].flow is f and len(stated) == 1                 # cached
	assert res[3][0].to_json()["regions"] and len(stated) == 1
	index.close()


def test_solver_debug_hooks_for_the_winners():
	"""the exact solvers' debug hooks (WRD::call_debug_hook, vectorian/core/cpp/alignment/wrd.h:31-59; FullSolver, wmd.h:147-181) are
	called for the winners with the joint problem as upstream lays it out: query masses in the first len_t slots, slice masses behind
	them, D = 1 except D[t][len_t + s] = max(0, 1 - S[s][t]), the plan G and its cost"""
	session, emb, words, rng = toy_session(n_docs=2, sents_per_doc=30, V=150, d=24)
	ts = EmbeddingTokenSim(emb, CosineSim())
	doc = session.documents[1]
	st = doc.spans["sentence"]["start"][4]
	text = " ".join(doc.tokens[st:st + 4])
	calls = []
	hook = lambda name, data: calls.append((name, data))
	wrd = session.index(OptimizedSpanSim(ts, alignment.WordRotatorsDistance()), corpus_factory=OracleCorpus)
	r = wrd.find(text, n=3, debug=hook)
	assert [c[0] for c in calls] == ["alignment/word-rotators-distance/solver"] * 3
	for m, (_, d) in zip(r, calls):
		n = m._len_s + 4
		assert d["t"]["text"] == text.split() and len(d["s"]["text"]) == m._len_s and d["s"]["text"] == doc.tokens[m._token_at:m._token_at + m._len_s] or m.doc_index != 1
		assert d["D"].shape == (n, n) and d["solution"]["G"].shape == (n, n) and d["solution"]["type"] == "optimal"
		assert abs(d["mag_t"][:4].sum() - 1.0) < 1e-5 and not d["mag_t"][4:].any() and abs(d["mag_s"][4:].sum() - 1.0) < 1e-5 and not d["mag_s"][:4].any()
		assert (d["D"][4:] == 1.0).all() and (d["D"][:4, :4] == 1.0).all()
		G = d["solution"]["G"]
		np.testing.assert_allclose(G[:4, 4:].sum(axis=1), d["mag_t"][:4], atol=1e-5)      # the plan ships every query mass ...
		np.testing.assert_allclose(G[:4, 4:].sum(axis=0), d["mag_s"][4:], atol=1e-5)      # ... into the slice's masses
		assert abs((1.0 - d["solution"]["cost"]) - m.raw_score) < 1e-5                    # score = 1 - EMD for normalised masses
	calls.clear()
	wmd = session.index(OptimizedSpanSim(ts, alignment.WordMoversDistance.wmd("bow")), corpus_factory=OracleCorpus)   # (upstream's naming: "bow" normalises)
	r = wmd.find(text, n=2, debug=hook)
	assert [c[0] for c in calls] == ["alignment/word-movers-distance/solver", "alignment/word-movers-distance/make"] * 2
	d = calls[0][1]
	assert abs(d["score"] - r[0].raw_score) < 1e-6 and d["G"].shape == d["D"].shape and abs(d["bow_t"].sum() - 1.0) < 1e-6


def test_query_tokens_outside_the_corpus_get_ids_of_their_own():
	"""QueryVocabulary (vectorian/core/cpp/vocabulary.h:500-541) is an incremental lexicon over the session's: a query token the
	corpus does not hold gets a new id behind the session's, the same word the same id -- the bags of words of the transport
	strategies then merge repeated words and only those (one id -1 for every unknown word merged them all: the device, which
	works on positions, and the double, which keys its vocabulary by id, disagreed; found by the Index-level sweep)"""
	session, emb, words, rng = toy_session(n_docs=2, sents_per_doc=10, V=400, d=24)
	V = session.vocab.size
	unknown = [w for w in words if session.vocab.token_to_id(w) < 0][:3]
	assert len(unknown) == 3
	known = session.documents[0].tokens[0]
	index = session.index(OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.WordMoversDistance.rwmd("bow/fast")), corpus_factory=OracleCorpus)
	p = index.make_query(" ".join([unknown[0], known, unknown[1], unknown[0], unknown[2]]), n=3).prepare(None)
	ids = p.token_ids.tolist()
	assert ids[1] == session.vocab.token_to_id(known) and ids[0] == ids[3] == V and ids[2] == V + 1 and ids[4] == V + 2
	assert len(index.find(" ".join(unknown), n=3, min_score=-10.0)) == 3


def test_find_many_with_a_hook_for_every_slice_equals_find():
	"""find_many(debug = AllSlices(hook)): the walk over all slices of query i runs while the lane's thread is already inside query
	i + 1 on the same handle (two concurrent calls on one handle, and `last_scores` of the wrong query).  The handle's
	lock serialises the calls and the scores of every slice are read under the query's own hold of it: per query, the hook sees what
	`find` shows it.  Reference: the hook is called per slice inside the matcher, vectorian/core/cpp/metric/alignment.h:145-173."""
	from vectorian_amd.index import AllSlices
	session, emb, words, rng = toy_session(n_docs=3, sents_per_doc=20, V=200, d=32)
	texts = [" ".join(session.documents[i % 3].tokens[7 * i:7 * i + 4]) for i in range(7)]
	for optimizer in (alignment.LocalAlignment(gap=alignment.LinearGapCost(0.2)), alignment.WordMoversDistance.rwmd("nbow")):
		index = session.partition("sentence").index(OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), optimizer), corpus_factory=OracleCorpus)
		calls = []
		hook = AllSlices(lambda name, data: calls.append((name, data["slice"], float(data["score"]), data.get("worst_score"))), chunk=16)
		want = []
		for t in texts:
			calls.clear()
			r = index.find(t, n=3, debug=hook)
			want.append(([(m.doc_index, m.slice_id, m.score) for m in r], list(calls)))
		calls.clear()
		many = index.find_many(texts, n=3, in_flight=3, options={"debug": hook})
		per_query = len(calls) // len(texts)
		assert per_query * len(texts) == len(calls) == 60 * len(texts)
		for i, (r, (w_matches, w_calls)) in enumerate(zip(many, want)):
			assert [(m.doc_index, m.slice_id, m.score) for m in r] == w_matches
			assert calls[i * per_query:(i + 1) * per_query] == w_calls   # in query order, on the calling thread
