"""Worker of tests/test_shards_gloo.py::test_sharded_index: every rank holds the same Session, keeps its
range of the slices in the (oracle-backed) backend and answers Index.find with the merged result."""

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(shard, strategy="align"):
	from fake_backend import OracleCorpus
	from test_host_api import toy_session
	from vectorian_amd import alignment
	from vectorian_amd.sim import CosineSim, EmbeddingTokenSim, OptimizedSpanSim
	session, emb, words, rng = toy_session(n_docs=6, sents_per_doc=30, V=400, d=32)
	optimizer = {"align": alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)), "wrd": alignment.WordRotatorsDistance(),
		"rwmd": alignment.WordMoversDistance.rwmd("nbow")}[strategy]
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), optimizer)
	index = session.partition("sentence", 2, 1).index(sim, corpus_factory=OracleCorpus, shard=shard)   # sliding windows of 2 sentences
	queries = []
	for di, si in ((1, 3), (4, 20), (5, 29)):
		doc = session.documents[di]
		st = doc.spans["sentence"]["start"][si]
		queries.append(" ".join(doc.tokens[st:st + 5]))
	return index, queries


def documents_answers(shard):
	"""session.partition("document"): whole documents of ~ 650 tokens as the slices of a (sharded) index -- alignments and the relaxed
	WMD, whose winners' similarity rows (sized by the longest document) travel in the rows exchange"""
	from fake_backend import OracleCorpus
	from test_host_api import toy_session
	from vectorian_amd import alignment
	from vectorian_amd.sim import CosineSim, EmbeddingTokenSim, OptimizedSpanSim
	session, emb, words, rng = toy_session(n_docs=7, sents_per_doc=30, V=400, d=32)
	out = {}
	for strategy in ("align", "rwmd"):
		optimizer = {"align": alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)), "rwmd": alignment.WordMoversDistance.rwmd("nbow")}[strategy]
		sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), optimizer)
		index = session.partition("document").index(sim, corpus_factory=OracleCorpus, shard=shard)
		queries = []
		for di, si in ((1, 3), (4, 20), (6, 29)):
			doc = session.documents[di]
			st = doc.spans["sentence"]["start"][si]
			queries.append(" ".join(doc.tokens[st:st + 6]))
		out[strategy] = [summary(r) for r in (index.find(q, n=4) for q in queries)]
	return out


def answers(index, queries):
	out = []
	for q in queries:
		r = index.find(q, n=6)
		rows = []
		for m in r:
			f = m.flow   # transport metrics: stated from the similarity rows / plans that travel with the records
			if f["type"] == "dense":
				shape = [float(round(float((f["flow"] * f["dist"]).sum()), 4)), float(round(float(f["flow"].sum()), 4))]
			else:
				shape = [int(x) for x in f["target"]]
			rows.append([m.doc_index, int(m.slice_id), float(m.score), f["type"], shape])
		out.append(rows)
	return out


def summary(r):
	rows = []
	for m in r:
		f = m.flow
		if f["type"] == "dense":
			shape = [float(round(float((f["flow"] * f["dist"]).sum()), 4)), float(round(float(f["flow"].sum()), 4))]
		elif f["type"] == "sparse":
			shape = [[int(x) for x in f["source"]], [int(x) for x in f["target"]], [float(round(float(x), 5)) for x in f["flow"]]]
		else:
			shape = [int(x) for x in f["target"]]
		rows.append([m.doc_index, int(m.slice_id), float(m.score), f["type"], shape])
	return rows


def many_queries(index, n=37):
	"""n queries cut from the session's own documents (contextual session: batched calls are possible)"""
	session = index.session
	out = []
	for i in range(n):
		doc = session.documents[i % len(session.documents)]
		st = int(doc.spans["sentence"]["start"][(7 * i) % len(doc.spans["sentence"]["start"])])
		# now and then a record of another size (32 columns); once a query of 70 tokens (round 4: the long-query path, records of 80 columns)
		out.append(" ".join(doc.tokens[st:st + (70 if i == 19 else 20 if i % 9 == 4 else 3 + i % 4)]))
	return out


def build_contextual(shard, strategy):
	"""a session with contextual embeddings (find_many shares calls over these: vk_query_batch)"""
	from fake_backend import OracleCorpus
	from test_host_api import contextual_toy
	from vectorian_amd import alignment
	from vectorian_amd.sim import CosineSim, EmbeddingTokenSim, OptimizedSpanSim
	session, emb, _ = contextual_toy(n_docs=5, sents=30)
	optimizer = {"align": alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)), "rwmd": alignment.WordMoversDistance.rwmd("nbow")}[strategy]
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), optimizer)
	return session.partition("sentence").index(sim, corpus_factory=OracleCorpus, shard=shard)


def find_many_answers(shard):
	"""find_many over a (sharded) index: alignments (16 per backend call) and relaxed WMD (256 per call), with progress"""
	out = {}
	for strategy in ("align", "rwmd"):
		index = build_contextual(shard, strategy)
		queries = many_queries(index)
		seen = []
		results = index.find_many(queries, n=5, batch=True, progress=seen.append)
		out[strategy] = [summary(r) for r in results]
		out[strategy + "_progress"] = seen
		out[strategy + "_batch_calls"] = index.corpus.batch_calls
		# one query at a time through the pipelined path as well (no shared calls)
		out[strategy + "_pipelined"] = [summary(r) for r in index.find_many(queries[:9], n=5, batch=False)]
	return out


def abort_answers(shard):
	"""Query.abort seen by ONE rank only: every rank still joins the exchange and the query yields no matches anywhere"""
	import numpy as np
	index = build_contextual(shard, "align")
	queries = many_queries(index, 6)
	flag = np.zeros(1, dtype=np.int32)
	if shard is not None and shard[0] == 1:
		flag[0] = 1
	res = {"batched": [len(r.matches) for r in index.find_many(queries, n=5, abort=flag)],
		"pipelined": [len(list(r)) for r in index.find_many(queries, n=5, abort=flag, batch=False)]}
	q = index.make_query(queries[0], n=5)
	if shard is not None and shard[0] == 1:
		q.abort()
	res["find"] = len(index._find(q))
	# a transport strategy: the rows of the merged winners follow in an all-reduce every rank must join, the aborted one too
	index = build_contextual(shard, "rwmd")
	res["rwmd"] = [len(list(r)) for r in index.find_many(queries, n=5, abort=flag, batch=False)] + [len(r.matches) for r in index.find_many(queries, n=5, abort=flag)]
	return res


def hook_answers(shard):
	"""find_many(debug = AllSlices(hook)) over a (sharded) index: every rank's hook walks the slices of ITS shard, per query, while the
	lanes' threads go on with the next queries on the same handles"""
	from vectorian_amd.index import AllSlices
	out = {}
	for strategy in ("align", "rwmd"):
		index = build_contextual(shard, strategy)
		queries = many_queries(index, 7)
		calls = []
		hook = AllSlices(lambda name, data: calls.append([int(data["slice"]), float(data["score"])]), chunk=32)
		results = index.find_many(queries, n=4, in_flight=3, batch=False, options={"debug": hook})
		per = len(calls) // len(queries)
		out[strategy] = {"matches": [summary(r) for r in results], "calls": [calls[i * per:(i + 1) * per] for i in range(len(queries))]}
	return out


def main(outdir):
	import torch.distributed as dist
	dist.init_process_group(backend="gloo")
	rank, world = dist.get_rank(), dist.get_world_size()
	res = {}
	for strategy in ("align", "wrd", "rwmd"):
		index, queries = build((rank, world), strategy)
		res[strategy] = answers(index, queries)
	res["find_many"] = find_many_answers((rank, world))
	res["abort"] = abort_answers((rank, world))
	res["documents"] = documents_answers((rank, world))
	res["hooks"] = hook_answers((rank, world))
	with open(os.path.join(outdir, f"index_rank{rank}.json"), "w") as f:
		json.dump(res, f)
	dist.barrier()
	dist.destroy_process_group()


if __name__ == "__main__":
	main(sys.argv[1])
