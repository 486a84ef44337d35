#!/usr/bin/env python3
"""Throughput through the operator surface (Session / Index.find_many), not the raw C-ABI: a static-embedding session of
`--sentences` synthetic sentences, `--queries` 10-token queries, WSB local alignment; prints one JSON line.

  python tools/index_throughput.py --sentences 1000000 --queries 60
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--sentences", type=int, default=1000000)
	ap.add_argument("--queries", type=int, default=60)
	ap.add_argument("--in-flight", type=int, default=3)
	args = ap.parse_args()
	from vectorian_amd import alignment, synth
	from vectorian_amd.corpus import Corpus, Document
	from vectorian_amd.embedding import StaticEmbedding
	from vectorian_amd.session import Session
	from vectorian_amd.sim import CosineSim, EmbeddingTokenSim, OptimizedSpanSim
	V, d, len_s = 50000, 300, 32
	rng = np.random.default_rng(5)
	words = [f"w{i}" for i in range(V)]
	emb = StaticEmbedding("synthetic-300", words, synth.make_vocab(V, d))
	per_doc = 10000
	docs = []
	for di in range((args.sentences + per_doc - 1) // per_doc):
		ids = synth.zipf_ids(per_doc * len_s, V, rng).reshape(per_doc, len_s)
		docs.append(Document([[words[j] for j in row] for row in ids]))
	session = Session(Corpus(docs), embeddings=[emb])
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)))
	t0 = time.perf_counter()
	index = session.partition("sentence").index(sim)
	build_s = time.perf_counter() - t0
	texts = [" ".join(docs[int(rng.integers(0, len(docs)))].tokens[a:a + 10]) for a in rng.integers(0, per_doc * len_s - 10, size=args.queries)]
	index.find_many(texts[:6], in_flight=args.in_flight)
	t0 = time.perf_counter()
	results = index.find_many(texts, in_flight=args.in_flight)
	el = time.perf_counter() - t0
	t0 = time.perf_counter()
	for t in texts[:10]:
		index.find(t)
	one = (time.perf_counter() - t0) / 10
	print(json.dumps({"surface": "Session / Index.find_many (static layout, WSB)", "sentences": index.n_slices, "queries": len(texts),
		"in_flight": args.in_flight, "alignments_per_s": index.n_slices * len(texts) / el, "ms_per_query": el / len(texts) * 1e3,
		"ms_per_find_one_at_a_time": one * 1e3, "index_build_s": build_s, "top_score": results[0][0].score if len(results[0]) else None}))
	index.close()


if __name__ == "__main__":
	main()
