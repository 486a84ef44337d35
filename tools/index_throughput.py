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


def build(n_sentences, contextual, strategy, n_queries=60):
	"""the synthetic session and its index; returns (index, texts, seconds the index took to build)"""
	from vectorian_amd import alignment, synth
	from vectorian_amd.corpus import Corpus, Document
	from vectorian_amd.embedding import ContextualEmbedding, StaticEmbedding
	from vectorian_amd.session import Session
	from vectorian_amd.sim import CosineSim, EmbeddingTokenSim, OptimizedSpanSim
	V, d, len_s = 50000, 300, 32
	rng = np.random.default_rng(5)
	words = [f"w{i}" for i in range(V)]
	E = synth.make_vocab(V, d)
	word_id = {w: i for i, w in enumerate(words)}
	if contextual:
		emb = ContextualEmbedding("ctx", d, lambda tokens: E[[word_id[t] for t in tokens]])
	else:
		emb = StaticEmbedding("synthetic-300", words, E)
	per_doc = 10000
	docs = []
	for di in range((n_sentences + per_doc - 1) // per_doc):
		ids = synth.zipf_ids(per_doc * len_s, V, rng).reshape(per_doc, len_s)
		sents = [[words[j] for j in row] for row in ids]
		if contextual:
			X = E[ids.reshape(-1)] + 0.1 * rng.standard_normal((per_doc * len_s, d)).astype(np.float32)
			docs.append(Document(sents, contextual_embeddings={"ctx": X}))
		else:
			docs.append(Document(sents))
	session = Session(Corpus(docs), embeddings=[emb])
	strategy = alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)) if strategy == "local" else alignment.WordMoversDistance.rwmd("nbow")
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), strategy)
	t0 = time.perf_counter()
	index = session.partition("sentence").index(sim)
	build_s = time.perf_counter() - t0
	texts = [" ".join(docs[int(rng.integers(0, len(docs)))].tokens[a:a + 10]) for a in rng.integers(0, per_doc * len_s - 10, size=n_queries)]
	return index, texts, build_s


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--sentences", type=int, default=1000000)
	ap.add_argument("--queries", type=int, default=60)
	ap.add_argument("--in-flight", type=int, default=3)
	ap.add_argument("--contextual", action="store_true", help="per-token vectors (contextual embedding) instead of the static layout: find_many then shares calls (vk_query_batch)")
	ap.add_argument("--strategy", choices=["local", "rwmd"], default="local")
	ap.add_argument("--no-batch", action="store_true", help="find_many(batch=False): one query per call")
	ap.add_argument("--profile", action="store_true", help="cProfile of the timed find_many on stderr")
	args = ap.parse_args()
	index, texts, build_s = build(args.sentences, args.contextual, args.strategy, args.queries)
	batch = False if args.no_batch else None
	index.find_many(texts[:max(6, min(len(texts), 32))], in_flight=args.in_flight, batch=batch)
	prof = None
	if args.profile:
		import cProfile
		prof = cProfile.Profile()
		prof.enable()
	t0 = time.perf_counter()
	results = index.find_many(texts, in_flight=args.in_flight, batch=batch)
	el = time.perf_counter() - t0
	if prof is not None:
		import pstats
		prof.disable()
		pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(18)
	t0 = time.perf_counter()
	for t in texts[:10]:
		index.find(t)
	one = (time.perf_counter() - t0) / 10
	print(json.dumps({"surface": f"Session / Index.find_many ({'contextual' if args.contextual else 'static'} layout, {args.strategy}, {'one query per call' if args.no_batch or not args.contextual else 'shared calls'})", "sentences": index.n_slices, "queries": len(texts),
		"in_flight": args.in_flight, "alignments_per_s": index.n_slices * len(texts) / el, "ms_per_query": el / len(texts) * 1e3,
		"ms_per_find_one_at_a_time": one * 1e3, "index_build_s": build_s, "top_score": results[0][0].score if len(results[0]) else None}))
	index.close()


if __name__ == "__main__":
	main()
