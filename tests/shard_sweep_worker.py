"""Worker of tests/test_gpu_shard_sweep.py: every rank (one process per rank, all on cuda:0, gloo between them) builds the same
random sessions as tests/test_gpu_index_sweep.py, keeps its range of the slices on the HIP backend (shard = (rank, world)) and
compares Index.find / find_many of the sharded index with the unsharded HIP index it builds beside it.
usage: torchrun ... shard_sweep_worker.py OUTDIR FIRST_SEED N_SEEDS"""

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def same(a, b, exact_flows):
	import numpy as np
	if [(m.doc_index, m.slice_id, m.score) for m in a] != [(m.doc_index, m.slice_id, m.score) for m in b]:
		return "result sets differ: %r | %r" % ([(m.doc_index, m.slice_id, m.score) for m in a][:4], [(m.doc_index, m.slice_id, m.score) for m in b][:4])
	for x, y in zip(a, b):
		fx, fy = x.flow, y.flow
		if (fx is None) != (fy is None):
			return "flow stated on one side only (sharded: %s, unsharded: %s; slice of %d tokens)" % (fx is not None, fy is not None, x._len_s)
		if fx is None:
			continue
		if fx["type"] != fy["type"]:
			return "flow types differ"
		for key in fx:
			if key != "type" and not (np.asarray(fx[key]) == np.asarray(fy[key])).all():
				return "flow['%s'] differs: %r | %r (slice of %d tokens)" % (key, np.asarray(fx[key]).tolist(), np.asarray(fy[key]).tolist(), x._len_s)
	return None


def one_seed(seed, rank, world):
	import numpy as np
	import test_gpu_index_sweep as T
	from vectorian_amd.sim import CosineSim, EmbeddingTokenSim, OptimizedSpanSim
	rng = np.random.default_rng(55000 + seed)
	session, emb, nlp, words = T.build_session(rng)
	strategy, is_align = T.random_strategy(rng)
	kw = {}
	if rng.random() < 0.2:
		kw = dict(tag_weights={t: float(rng.uniform(0.25, 2.5)) for t in rng.choice(T.TAGS, size=3, replace=False)},
			pos_mismatch_penalty=float(rng.uniform(0, 0.5)), similarity_threshold=float(rng.uniform(0, 0.2)))
	sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), strategy, **kw)
	if rng.random() < 0.7:
		part = session.partition("sentence", int(rng.integers(1, 4)), int(rng.integers(1, 3)))
	else:
		part = session.partition("token", int(rng.integers(4, 20)), int(rng.integers(1, 8)))
	whole = part.index(sim, nlp=nlp)
	index_kw = {}
	if rng.random() < 0.3:
		index_kw["saliency"] = rng.uniform(0.5, 1.5, size=whole.n_slices).astype(np.float32)
		whole.close()
		whole = part.index(sim, nlp=nlp, **index_kw)
	mine = part.index(sim, nlp=nlp, shard=(rank, world), **index_kw)
	texts = []
	for _ in range(int(rng.integers(3, 9))):
		doc = session.documents[int(rng.integers(0, len(session.documents)))]
		len_t = int(rng.integers(1, 13)) if rng.random() < 0.85 else int(rng.integers(17, 40))
		if len(doc.tokens) > len_t and rng.random() < 0.7:
			a0 = int(rng.integers(0, len(doc.tokens) - len_t))
			texts.append(" ".join(doc.tokens[a0:a0 + len_t]))
		else:
			texts.append(" ".join(words[int(i)] for i in rng.integers(0, len(words), size=len_t)))
	options = {}
	if rng.random() < 0.2:
		options["pos_filter"] = ["DET"]
	if is_align and rng.random() < 0.2:
		options["submatch_weight"] = float(rng.choice([0.5, 1.0, 2.0]))
	n = int(rng.choice([1, 5, 12]))
	min_score = 0.0 if rng.random() < 0.7 else -100.0
	ctx = "seed %d %s %r tags=%s %r %s n=%d min=%g saliency=%s opts=%r" % (seed, type(strategy).__name__, getattr(strategy, "_options", None), bool(kw),
		part.to_args(), type(emb).__name__, n, min_score, bool(index_kw), options)
	fails = []
	for t in texts[:2]:
		why = same(mine.find(t, n=n, min_score=min_score, options=options), whole.find(t, n=n, min_score=min_score, options=options), True)
		if why:
			fails.append("find: %s (%s) query %r" % (why, ctx, t))
	in_flight = int(rng.integers(1, 4))
	for a, b, t in zip(mine.find_many(texts, n=n, min_score=min_score, options=options, in_flight=in_flight),
			whole.find_many(texts, n=n, min_score=min_score, options=options, in_flight=in_flight), texts):
		why = same(a, b, True)
		if why:
			fails.append("find_many: %s (%s) query %r" % (why, ctx, t))
	mine.close(); whole.close()
	return fails


def main(outdir, first, count):
	import torch.distributed as dist
	from vectorian_amd import core
	core.init(0)
	dist.init_process_group(backend="gloo")
	rank, world = dist.get_rank(), dist.get_world_size()
	fails, done = [], 0
	for seed in range(first, first + count):
		fails += one_seed(seed, rank, world)
		done += 1
		if len(fails) > 20:
			break
	with open(os.path.join(outdir, f"sweep_rank{rank}.json"), "w") as f:
		json.dump({"done": done, "fails": fails}, f)
	dist.barrier()
	dist.destroy_process_group()


if __name__ == "__main__":
	main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
