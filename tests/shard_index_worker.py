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


def main(outdir):
	import torch.distributed as dist
	dist.init_process_group(backend="gloo")
	rank, world = dist.get_rank(), dist.get_world_size()
	res = {}
	for strategy in ("align", "wrd", "rwmd"):
		index, queries = build((rank, world), strategy)
		res[strategy] = answers(index, queries)
	with open(os.path.join(outdir, f"index_rank{rank}.json"), "w") as f:
		json.dump(res, f)
	dist.barrier()
	dist.destroy_process_group()


if __name__ == "__main__":
	main(sys.argv[1])
