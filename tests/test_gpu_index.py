"""-m gpu: the Python operator surface (Session / Index.find) on the real HIP backend must
return what the same surface returns on the oracle-backed double."""

import numpy as np
import pytest

from helpers import assert_json_close

from fake_backend import OracleCorpus
from test_host_api import toy_session
from vectorian_amd import alignment
from vectorian_amd.sim import CosineSim, EmbeddingTokenSim, OptimizedSpanSim

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("optimizer", [
	alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)),
	alignment.LocalAlignment(),
	alignment.GlobalAlignment(gap=alignment.LinearGapCost(0.1)),
	alignment.SemiGlobalAlignment(gap={"s": alignment.AffineGapCost(0.2, 0.05), "t": alignment.LinearGapCost(0.1)}),
])
def test_index_find_on_hip_equals_oracle_double(hip, optimizer):
	session, emb, words, rng = toy_session()
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), optimizer)
	gpu = session.partition("sentence").index(sim)                       # core.Corpus: the HIP path
	cpu = session.partition("sentence").index(sim, corpus_factory=OracleCorpus)
	doc = session.documents[5]
	st = doc.spans["sentence"]["start"][11]
	long_query = " ".join(doc.tokens[st:st + 22])                      # a whole sentence as the query: vk_score32_kernel
	for text in (" ".join(doc.tokens[st:st + 5]), "w3 w17 w4 w900 w2", "w1", long_query):
		a = gpu.find(text, n=10, min_score=-100.0)
		b = cpu.find(text, n=10, min_score=-100.0)
		assert [(m.doc_index, m.slice_id) for m in a] == [(m.doc_index, m.slice_id) for m in b]
		# winners are restated in the oracle's arithmetic on the device (sim_canon): scores, tracebacks, per-edge distances bit for bit
		assert [m.score for m in a] == [m.score for m in b]
		for x, y in zip(a, b):
			assert (x.flow["target"] == y.flow["target"]).all()
			assert (x.flow["dist"] == y.flow["dist"]).all()
		assert a[0].to_json()["regions"] == b[0].to_json()["regions"]
	gpu.close()


@pytest.mark.parametrize("optimizer", [
	alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)),
	alignment.SemiGlobalAlignment(gap={"s": alignment.AffineGapCost(0.2, 0.05), "t": alignment.LinearGapCost(0.1)}),
])
def test_document_partition_on_hip_equals_oracle_double(hip, optimizer):
	"""session.partition("document"): whole documents as slices (the one span the importers give every document,
	vectorian/importers.py:30-36; mkdocs/docs/documents.md:37) -- documents of ~ 1,100 tokens, beyond VK_MAX_SENT_LEN: the
	one-wave-per-slice kernel with its state in global memory; also windows of 30 sentences, step 10"""
	session, emb, words, rng = toy_session(n_docs=12, sents_per_doc=50, V=2000, d=100)
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), optimizer)
	doc = session.documents[5]
	st = doc.spans["sentence"]["start"][31]
	for part in (("document",), ("sentence", 30, 10)):
		gpu = session.partition(*part).index(sim)
		cpu = session.partition(*part).index(sim, corpus_factory=OracleCorpus)
		assert gpu.n_slices == cpu.n_slices
		for text in (" ".join(doc.tokens[st:st + 6]), " ".join(doc.tokens[st:st + 40:2])):
			a = gpu.find(text, n=5, min_score=-100.0)
			b = cpu.find(text, n=5, min_score=-100.0)
			assert [(m.doc_index, m.slice_id) for m in a] == [(m.doc_index, m.slice_id) for m in b]
			assert [m.score for m in a] == [m.score for m in b]
			for x, y in zip(a, b):
				assert (x.flow["target"] == y.flow["target"]).all()
				assert (x.flow["dist"] == y.flow["dist"]).all()
			assert a[0].to_json()["regions"] == b[0].to_json()["regions"]
			if part == ("document",) and isinstance(optimizer, alignment.LocalAlignment):
				assert a[0].doc_index == 5 and a[0].slice_id == 0
		gpu.close()


def test_debug_hook_for_every_slice_on_hip_equals_oracle_double(hip):
	"""debug = AllSlices(hook): every slice stated by the traceback kernel (vk_query_desc.only_slices) -- the same calls, slice by
	slice, as on the oracle-backed double: aligner score, similarity matrix, flow"""
	from vectorian_amd.index import AllSlices
	session, emb, words, rng = toy_session(n_docs=4, sents_per_doc=60, V=500, d=100)
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)))
	doc = session.documents[2]
	st = doc.spans["sentence"]["start"][7]
	for text in (" ".join(doc.tokens[st:st + 6]), " ".join(doc.tokens[st:st + 19])):   # 19 tokens: the wide traceback kernel
		calls = {}
		for name, factory in (("gpu", None), ("cpu", OracleCorpus)):
			index = session.partition("sentence").index(sim, corpus_factory=factory)
			got = calls.setdefault(name, [])
			index.find(text, n=3, debug=AllSlices(lambda n_, d_, got=got: got.append((n_, d_)), chunk=100))
			index.close()
		assert len(calls["gpu"]) == len(calls["cpu"]) == 240
		for (na, a), (nb, b) in zip(calls["gpu"], calls["cpu"]):
			assert na == nb == "alignment" and a["slice"] == b["slice"]
			assert a["score"] == b["score"]
			assert (a["flow"]["target"] == b["flow"]["target"]).all() and (a["flow"]["dist"] == b["flow"]["dist"]).all()
			assert (a["similarity"] == b["similarity"]).all()


def test_debug_hook_for_every_slice_under_a_submatch_weight(hip):
	"""AllSlices(hook) with submatch_weight != 0 (round 3 fell back to the winners): the listed slices are stated from their own
	tracebacks, whatever the weight; the result set itself comes out of the candidate rounds as before"""
	from vectorian_amd.index import AllSlices
	session, emb, words, rng = toy_session(n_docs=3, sents_per_doc=40, V=500, d=100)
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)))
	doc = session.documents[1]
	st = doc.spans["sentence"]["start"][9]
	text = " ".join(doc.tokens[st:st + 7])
	calls, results = {}, {}
	for name, factory in (("gpu", None), ("cpu", OracleCorpus)):
		index = session.partition("sentence").index(sim, corpus_factory=factory)
		got = calls.setdefault(name, [])
		results[name] = index.find(text, n=4, options={"submatch_weight": 1.5, "debug": AllSlices(lambda n_, d_, got=got: got.append((n_, d_)), chunk=50)})
		index.close()
	assert len(calls["gpu"]) == len(calls["cpu"]) == 120
	for (na, a), (nb, b) in zip(calls["gpu"], calls["cpu"]):
		assert na == nb == "alignment" and a["slice"] == b["slice"] and a["score"] == b["score"]
		assert (a["flow"]["target"] == b["flow"]["target"]).all() and (a["similarity"] == b["similarity"]).all()
	assert [(m.doc_index, m.slice_id, m.score) for m in results["gpu"]] == [(m.doc_index, m.slice_id, m.score) for m in results["cpu"]]


def test_span_embedding_index_on_hip(hip):
	from test_host_api import Document, Session
	from vectorian_amd.sim import EmbeddedSpanSim, SpanEmbedding
	rng = np.random.default_rng(9)
	n, d = 3000, 384
	vectors = rng.standard_normal((n, d)).astype(np.float32)
	docs = [Document([[f"s{i}"] for i in range(1000)]) for _ in range(3)]
	session = Session(docs, embeddings=[])
	qv = vectors[1234] + 0.1 * rng.standard_normal(d).astype(np.float32)
	emb = SpanEmbedding("enc", d, lambda texts: np.stack([qv for _ in texts]))
	index = session.partition("sentence").index(EmbeddedSpanSim(emb), vectors=vectors)
	r = index.find("anything", n=5)
	cos = (vectors @ qv) / (np.linalg.norm(vectors, axis=1) * np.linalg.norm(qv))
	order = np.argsort(-cos)[:5]
	assert [docs.index(m.prepared_doc) * 1000 + m.slice_id for m in r] == [int(i) for i in order]
	np.testing.assert_allclose([m.score for m in r], cos[order], atol=2e-3)     # bf16 vectors
	index.close()


def test_filters_and_token_windows_on_hip_equal_oracle_double(hip):
	"""pos_filter / tag_filter (vk_corpus_filter on the device against the host-side compaction of the double), a
	token-level partition with overlapping windows, a masked document, alignment and transport strategies"""
	from test_host_api import Corpus, Document, Session, StaticEmbedding
	from vectorian_amd import synth
	rng = np.random.default_rng(31)
	V, d = 600, 96
	words = [f"w{i}" for i in range(V)]
	emb = StaticEmbedding("toy-96", words, synth.make_vocab(V, d))
	pos_names, tag_names = ["NOUN", "VERB", "DET", "ADJ", "PUNCT"], ["NN", "VBZ", "DT", "JJ", ".", "NNS", "VBD"]
	docs = []
	for di in range(6):
		sents, pos, tags = [], [], []
		for _ in range(40):
			n = int(rng.integers(3, 30))
			sents.append([words[i] for i in synth.zipf_ids(n, V, rng)])
			pos.append([pos_names[int(i)] for i in rng.integers(0, len(pos_names), size=n)])
			tags.append([tag_names[int(i)] for i in rng.integers(0, len(tag_names), size=n)])
		n_raw = sum(len(s) for s in sents)
		docs.append(Document(sents, pos=pos, tags=tags, token_mask=(rng.random(n_raw) > 0.1) if di % 2 else None))
	session = Session(Corpus(docs), embeddings=[emb])
	doc = session.documents[2]
	query = " ".join(doc.tokens[100:106])
	for strategy in (alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)), alignment.WordMoversDistance.rwmd("nbow"),
			alignment.WordRotatorsDistance()):
		sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), strategy)
		for part in (session.partition("sentence"), session.partition("token", 12, 5)):
			gpu, cpu = part.index(sim), part.index(sim, corpus_factory=OracleCorpus)
			for options in ({}, {"pos_filter": ["DET", "PUNCT"]}, {"tag_filter": ["NN", "."], "pos_filter": ["ADJ"]}):
				a, b = gpu.find(query, n=8, options=options), cpu.find(query, n=8, options=options)
				assert [(m.doc_index, m.slice_id) for m in a] == [(m.doc_index, m.slice_id) for m in b], (type(strategy).__name__, options)
				np.testing.assert_allclose([m.score for m in a], [m.score for m in b], atol=1e-4)
				if isinstance(strategy, alignment.LocalAlignment):
					for x, y in zip(a, b):
						assert (x.flow["target"] == y.flow["target"]).all()
					assert a[0].to_json()["regions"] == b[0].to_json()["regions"]
			gpu.close()


def test_find_many_on_hip_equals_find(hip):
	"""Index.find_many: three queries in flight on views of the resident corpus; every Result as from find()"""
	session, emb, words, rng = toy_session()
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)))
	index = session.partition("sentence").index(sim)
	texts = [" ".join(session.documents[i % 20].tokens[11 * i:11 * i + 3 + i % 20]) for i in range(40)]
	many = index.find_many(texts, n=10)
	for text, res in zip(texts, many):
		one = index.find(text, n=10)
		assert [(m.doc_index, m.slice_id) for m in res] == [(m.doc_index, m.slice_id) for m in one]
		np.testing.assert_array_equal([m.score for m in res], [m.score for m in one])
		assert all((a.flow["target"] == b.flow["target"]).all() for a, b in zip(res, one))
	index.close()


@pytest.mark.parametrize("strategy,uniform,d", [("local", None, 64), ("local", None, 300), ("rwmd", 32, 300), ("rwmd", None, 300), ("rwmd", None, 64), ("rwmd_bow", None, 300)])
def test_find_many_shares_calls_on_hip(hip, strategy, uniform, d):
	"""Index.find_many over a contextual embedding: alignment queries share passes over the corpus (vk_score_batch_kernel), relaxed-WMD
	queries go up to 256 per call through the GEMM kernels (uniform 32-token sentences: on the resident tiles; ragged: padded buckets),
	with the similarity rows of every query's winners for the flows; every Result as from find()"""
	from test_host_api import contextual_toy
	from vectorian_amd.alignment import WordMoversDistance
	session, emb, docs = contextual_toy(n_docs=6, sents=120, d=d, uniform=uniform)   # 300-d: the GEMM kernels; 64-d: the shared pass
	al = {"local": alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)), "rwmd": WordMoversDistance.rwmd("nbow"),
		"rwmd_bow": WordMoversDistance.rwmd("bow/fast")}[strategy]
	index = session.partition("sentence").index(OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), al))
	texts = [" ".join(docs[i % 6].tokens[9 * i:9 * i + 3 + i % 8]) for i in range(45)]
	many = index.find_many(texts, n=6, batch=True)
	for text, res in zip(texts, many):
		one = index.find(text, n=6)
		assert [(m.doc_index, m.slice_id) for m in res] == [(m.doc_index, m.slice_id) for m in one]
		np.testing.assert_allclose([m.score for m in res], [m.score for m in one], atol=2e-6)
		for a, b in zip(res, one):
			fa, fb = a.flow, b.flow
			assert fa["type"] == fb["type"]
			if fa["type"] == "injective":
				assert (fa["target"] == fb["target"]).all()
			else:
				assert set(fa) == set(fb)
				for key in fa:
					if key != "type":
						np.testing.assert_allclose(np.asarray(fa[key], dtype=float), np.asarray(fb[key], dtype=float), atol=1e-6)
	index.close()


def test_tag_weighted_transport_and_long_queries_on_hip_equal_oracle_double(hip):
	"""'alignment-tag-weighted' with the transport strategies (any matcher takes the modifier, match/instantiate.cpp:173-189) and
	transport queries of more than 16 tokens, through Session / Index.find: ids, scores and the stated flows as the double's"""
	from test_host_api import Corpus, Document, Session, StaticEmbedding
	from vectorian_amd import synth
	rng = np.random.default_rng(41)
	V, d = 500, 64
	words = [f"w{i}" for i in range(V)]
	emb = StaticEmbedding("toy-64", words, synth.make_vocab(V, d) * rng.lognormal(0, 0.3, size=(V, 1)).astype(np.float32))
	tag_names = ["NN", "VBZ", "DT", "JJ", ".", "NNS", "VBD"]
	pos_of_tag = {"NN": "NOUN", "NNS": "NOUN", "VBZ": "VERB", "VBD": "VERB", "DT": "DET", "JJ": "ADJ", ".": "PUNCT"}
	tag_of = lambda w: tag_names[(int(w[1:]) * 7919) % len(tag_names)]      # one tag per word, the universal POS a function of it
	docs = []
	for di in range(5):
		sents = [[words[i] for i in synth.zipf_ids(int(rng.integers(3, 40)), V, rng)] for _ in range(40)]
		docs.append(Document(sents, pos=[[pos_of_tag[tag_of(w)] for w in s] for s in sents], tags=[[tag_of(w) for w in s] for s in sents]))
	session = Session(Corpus(docs), embeddings=[emb])
	nlp = lambda text: [{"text": w, "pos": pos_of_tag[tag_of(w)], "tag": tag_of(w)} for w in text.split()]
	doc = session.documents[3]
	short, long_q = " ".join(doc.tokens[50:57]), " ".join(doc.tokens[200:227])
	cases = [
		(alignment.WordMoversDistance.rwmd("nbow"), True, short), (alignment.WordMoversDistance.wmd("nbow"), True, short),
		(alignment.WordRotatorsDistance(), True, short), (alignment.WordRotatorsDistance(), False, long_q),
		(alignment.WordMoversDistance.wmd("bow"), False, long_q), (alignment.WordRotatorsDistance(), True, long_q),
	]
	for strategy, tagged, text in cases:
		kw = dict(tag_weights={"NN": 2.0, "VBZ": 1.5, "DT": 0.25}, pos_mismatch_penalty=0.3, similarity_threshold=0.1) if tagged else {}
		sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), strategy, **kw)
		gpu = session.partition("sentence").index(sim, nlp=nlp)
		cpu = session.partition("sentence").index(sim, nlp=nlp, corpus_factory=OracleCorpus)
		a, b = gpu.find(text, n=6, min_score=0.0), cpu.find(text, n=6, min_score=0.0)
		assert len(a) == len(b) == 6
		assert [(m.doc_index, m.slice_id) for m in a] == [(m.doc_index, m.slice_id) for m in b], (type(strategy).__name__, tagged, len(text.split()))
		np.testing.assert_allclose([m.score for m in a], [m.score for m in b], atol=2e-5)
		for x, y in zip(a[:3], b[:3]):
			fx, fy = x.flow, y.flow
			assert fx["type"] == fy["type"]
			if fx["type"] == "dense":
				# an optimal plan need not be unique (repeated words give equal rows): same mass moved at the same cost
				assert fx["flow"].shape == fy["flow"].shape == fx["dist"].shape
				np.testing.assert_allclose(fx["dist"], fy["dist"], atol=2e-5)
				assert abs(float(np.sum(fx["flow"])) - float(np.sum(fy["flow"]))) < 1e-4
				assert abs(float(np.sum(fx["flow"] * fx["dist"])) - float(np.sum(fy["flow"] * fy["dist"]))) < 1e-4
			else:
				assert_json_close({k: np.asarray(v).tolist() for k, v in fx.items() if k != "type"},
					{k: np.asarray(v).tolist() for k, v in fy.items() if k != "type"}, 2e-5)
		gpu.close()


def test_debug_hook_and_abort_on_hip(hip):
	"""the debug hook on the real backend (similarity rows of the winners from vk_rows_kernel) and Query.abort / find_many(abort=)
	through vk_query_desc.abort"""
	session, emb, words, rng = toy_session()
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)))
	gpu = session.partition("sentence").index(sim)
	cpu = session.partition("sentence").index(sim, corpus_factory=OracleCorpus)
	doc = session.documents[7]
	text = " ".join(doc.tokens[40:46])
	got, want = [], []
	a = gpu.find(text, n=5, debug=lambda name, data: got.append((name, data)))
	b = cpu.find(text, n=5, debug=lambda name, data: want.append((name, data)))
	assert len(got) == len(want) == 5 and [m.slice_id for m in a] == [m.slice_id for m in b]
	for (n1, d1), (n2, d2) in zip(got, want):
		assert n1 == n2 == "alignment" and d1["slice"] == d2["slice"] and abs(d1["score"] - d2["score"]) < 1e-4
		np.testing.assert_allclose(d1["similarity"], d2["similarity"], atol=2e-6)
		assert (d1["flow"]["target"] == d2["flow"]["target"]).all()
	q = gpu.make_query(text, n=5)
	q.abort()
	assert gpu._find(q) == [] and len(gpu.find(text, n=5)) == 5
	flag = np.zeros(1, dtype=np.int32)
	texts = [" ".join(session.documents[i % 20].tokens[5 * i:5 * i + 4]) for i in range(12)]
	assert all(len(r) > 0 for r in gpu.find_many(texts, n=3, abort=flag))
	flag[0] = 1
	assert all(len(r) == 0 for r in gpu.find_many(texts, n=3, abort=flag))
	with pytest.raises(hip.VkError) as e:   # the C-ABI: VK_ERR_ABORTED from vk_query and vk_query_batch
		gpu.corpus.query(np.ones((3, emb.dimension), np.float32), q_token_ids=np.zeros(3, np.int32), abort_flag=flag)
	assert e.value.status == hip.VK_ERR_ABORTED
	gpu.close()


@pytest.mark.parametrize("variant", ["nbow", "nbow/distributed", "bow/fast"])
def test_debug_hook_for_every_slice_relaxed_wmd_on_hip_equals_oracle_double(hip, variant):
	"""debug = AllSlices(hook) under the relaxed word mover's distance: one 'alignment/word-movers-distance/make' call per slice with
	its score and the worst score of the result set so far (metric/alignment.h:600-607).  The scores are restated from canonical
	similarity rows on the host, chunk by chunk (vk_query_desc.only_slices): the oracle's floats for EVERY slice, not only the winners"""
	from vectorian_amd.index import AllSlices
	session, emb, words, rng = toy_session(n_docs=3, sents_per_doc=50, V=300, d=64)
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.WordMoversDistance.rwmd(variant))
	doc = session.documents[1]
	st = doc.spans["sentence"]["start"][11]
	for text in (" ".join(doc.tokens[st:st + 5]), " ".join(doc.tokens[st:st + 21])):
		calls = {}
		for name, factory in (("gpu", None), ("cpu", OracleCorpus)):
			index = session.partition("sentence", 2, 1).index(sim, corpus_factory=factory)   # sliding windows of two sentences
			got = calls.setdefault(name, [])
			index.find(text, n=4, min_score=-10.0, debug=AllSlices(lambda n_, d_, got=got: got.append((n_, d_)), chunk=64))
			index.close()
		assert len(calls["gpu"]) == len(calls["cpu"]) > 100
		for (na, a), (nb, b) in zip(calls["gpu"], calls["cpu"]):
			assert na == nb == "alignment/word-movers-distance/make" and a["slice"] == b["slice"]
			assert a["score"] == b["score"] and a["worst_score"] == b["worst_score"]


@pytest.mark.parametrize("strategy", ["wrd", "wmd"])
def test_debug_hook_for_every_slice_exact_transport_on_hip_equals_oracle_double(hip, strategy):
	"""debug = AllSlices(hook) under the exact transports: EVERY slice solved (vk_query_desc.only_slices: no bound pass, nothing pruned)
	and handed to the solver's hook -- tokens, masses, distance matrix, plan and cost (wrd.h:31-59, wmd.h:147-181); full WMD: 'make'
	with the worst score so far as well"""
	from vectorian_amd.index import AllSlices
	session, emb, words, rng = toy_session(n_docs=3, sents_per_doc=40, V=300, d=64)
	optimizer = alignment.WordRotatorsDistance() if strategy == "wrd" else alignment.WordMoversDistance.wmd("nbow")
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), optimizer)
	doc = session.documents[1]
	st = doc.spans["sentence"]["start"][9]
	text = " ".join(doc.tokens[st:st + 6])
	calls = {}
	for name, factory in (("gpu", None), ("cpu", OracleCorpus)):
		index = session.partition("sentence").index(sim, corpus_factory=factory)
		got = calls.setdefault(name, [])
		index.find(text, n=4, min_score=-10.0, debug=AllSlices(lambda n_, d_, got=got: got.append((n_, d_)), chunk=50))
		index.close()
	assert [x[0] for x in calls["gpu"]] == [x[0] for x in calls["cpu"]] and len(calls["gpu"]) >= 120
	for (na, a), (nb, b) in zip(calls["gpu"], calls["cpu"]):
		if na.endswith("/make"):
			assert a["slice"] == b["slice"] and abs(a["score"] - b["score"]) < 2e-5 and abs(a["worst_score"] - b["worst_score"]) < 2e-5
		else:
			assert a["s"]["id"] == b["s"]["id"] and a["t"]["id"] == b["t"]["id"]
			for key in a:
				if isinstance(a[key], np.ndarray) and a[key].dtype.kind == "f" and key != "G":   # (an optimal plan need not be unique)
					np.testing.assert_allclose(a[key], b[key], atol=2e-5, err_msg=key)
				elif isinstance(a[key], float):
					assert abs(a[key] - b[key]) < 2e-5, key


@pytest.mark.parametrize("strategy", ["wsb", "rwmd"])
def test_find_many_with_a_hook_for_every_slice_on_hip_equals_find(hip, strategy):
	"""find_many(debug = AllSlices(hook)) on the real backend, three handles in flight: per query the hook is handed what `find`
	hands it (the walk raced the lane's next query on the same handle; relaxed WMD read the wrong query's scores)"""
	from vectorian_amd.index import AllSlices
	session, emb, words, rng = toy_session(n_docs=6, sents_per_doc=40)
	optimizer = alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)) if strategy == "wsb" else alignment.WordMoversDistance.rwmd("nbow")
	gpu = session.partition("sentence").index(OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), optimizer))
	texts = [" ".join(session.documents[i % 6].tokens[9 * i:9 * i + 3 + i % 4]) for i in range(9)]
	calls = []
	hook = AllSlices(lambda name, data: calls.append((name, data["slice"], float(data["score"]), data.get("worst_score"))), chunk=64)
	want = []
	for t in texts:
		calls.clear()
		r = gpu.find(t, n=4, debug=hook)
		want.append(([(m.doc_index, m.slice_id, m.score) for m in r], list(calls)))
	calls.clear()
	many = gpu.find_many(texts, n=4, in_flight=3, options={"debug": hook})
	per_query = len(calls) // len(texts)
	assert per_query == 240 and per_query * len(texts) == len(calls)
	for i, (r, (w_matches, w_calls)) in enumerate(zip(many, want)):
		assert [(m.doc_index, m.slice_id, m.score) for m in r] == w_matches
		assert calls[i * per_query:(i + 1) * per_query] == w_calls
	gpu.close()


@pytest.mark.parametrize("strategy", ["rwmd", "rwmd_bow", "local"])
def test_find_many_shares_calls_over_static_embeddings_on_hip(hip, strategy):
	"""find_many(batch=True) over a STATIC embedding (refused until round 4): relaxed-WMD queries share one table over the vocabulary
	and one gather pass per call (vk_rwmd_static32_kernel), alignments are answered inside the call; every Result as from find() and
	as on the oracle double.  Reference path: metric/static.cpp:9-78, slice/static.h:71-75."""
	from vectorian_amd.alignment import WordMoversDistance
	session, emb, words, rng = toy_session(n_docs=8, sents_per_doc=60)
	al = {"local": alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)), "rwmd": WordMoversDistance.rwmd("nbow"),
		"rwmd_bow": WordMoversDistance.rwmd("bow/fast")}[strategy]
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), al)
	gpu = session.partition("sentence").index(sim)
	cpu = session.partition("sentence").index(sim, corpus_factory=OracleCorpus)
	texts = [" ".join(session.documents[i % 8].tokens[9 * i:9 * i + 3 + i % 8]) for i in range(40)] + ["w3 w17 zzz-unknown w4", "w1"]
	many = gpu.find_many(texts, n=6, batch=True)
	for text, res in zip(texts, many):
		one, ref = gpu.find(text, n=6), cpu.find(text, n=6)
		assert [(m.doc_index, m.slice_id) for m in res] == [(m.doc_index, m.slice_id) for m in one] == [(m.doc_index, m.slice_id) for m in ref]
		assert [m.score for m in res] == [m.score for m in one] == [m.score for m in ref]
		for a, b in zip(res, one):
			fa, fb = a.flow, b.flow
			assert fa["type"] == fb["type"] and set(fa) == set(fb)
			for key in fa:
				if key != "type":
					assert (np.asarray(fa[key]) == np.asarray(fb[key])).all()
	gpu.close()


def test_a_paragraph_as_the_query_on_hip_equals_oracle_double(hip):
	"""queries of more than 64 tokens through the operator surface (round 4; refused until then): a passage of 90 / 150 tokens as the
	query of Index.find and find_many -- matches, scores, flows and regions as on the oracle double; the debug hook is called for the
	winners without a similarity matrix (the rows of such queries are not returned)"""
	session, emb, words, rng = toy_session(n_docs=10, sents_per_doc=40)
	for optimizer in (alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)), alignment.SemiGlobalAlignment(gap=alignment.LinearGapCost(0.1))):
		sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), optimizer)
		gpu = session.partition("sentence").index(sim)
		cpu = session.partition("sentence").index(sim, corpus_factory=OracleCorpus)
		doc = session.documents[4]
		texts = [" ".join(doc.tokens[30:120]), " ".join(doc.tokens[200:350]), " ".join(doc.tokens[5:9])]
		many = gpu.find_many(texts, n=8, min_score=-100.0)
		for text, res in zip(texts, many):
			a, b = gpu.find(text, n=8, min_score=-100.0), cpu.find(text, n=8, min_score=-100.0)
			assert [(m.doc_index, m.slice_id) for m in a] == [(m.doc_index, m.slice_id) for m in b] == [(m.doc_index, m.slice_id) for m in res]
			assert [m.score for m in a] == [m.score for m in b] == [m.score for m in res]
			for x, y in zip(a, b):
				assert (x.flow["target"] == y.flow["target"]).all() and (x.flow["dist"] == y.flow["dist"]).all()
			assert a[0].to_json()["regions"] == b[0].to_json()["regions"]
		calls = []
		gpu.find(texts[0], n=3, debug=lambda name, data: calls.append(data))
		assert len(calls) == 3 and all(d["similarity"] is None for d in calls)
		gpu.close()
