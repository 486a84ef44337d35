"""replays one seed of tests/test_gpu_index_sweep.py::test_random_session_find_many_equals_find and prints the queries whose
find_many result differs from find: python tools/probe/find_many_case.py SEED"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_index_sweep as T
from vectorian_amd import core
from vectorian_amd.sim import CosineSim, EmbeddingTokenSim, OptimizedSpanSim
core.init(0)
seed = int(sys.argv[1])
rng = np.random.default_rng(99000 + seed)
session, emb, nlp, words = T.build_session(rng)
strategy, is_align = T.random_strategy(rng)
kw = {}
if rng.random() < 0.15:
	kw = dict(tag_weights={t: float(rng.uniform(0.25, 2.5)) for t in rng.choice(T.TAGS, size=3, replace=False)},
		pos_mismatch_penalty=float(rng.uniform(0, 0.5)), similarity_threshold=float(rng.uniform(0, 0.2)))
sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), strategy, **kw)
if rng.random() < 0.7:
	part = session.partition("sentence", int(rng.integers(1, 4)), int(rng.integers(1, 3)))
else:
	part = session.partition("token", int(rng.integers(4, 20)), int(rng.integers(1, 8)))
gpu = part.index(sim, nlp=nlp)
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
lens = gpu._slice_end - gpu._slice_start
print(type(strategy).__name__, getattr(strategy, "_options", None), "tags", bool(kw), part.to_args(), type(emb).__name__, "n", n, "min", min_score, options,
	"slices", gpu.n_slices, "longest", int(lens.max()), "queries", len(texts), "wide", wide)
calls = []
orig = gpu._corpus.query_batch
many = gpu.find_many(texts, n=n, min_score=min_score, options=options, in_flight=int(rng.integers(1, 4)))
print("batch calls", getattr(gpu._corpus, "batch_calls", None))
for text, a in zip(texts, many):
	b = gpu.find(text, n=n, min_score=min_score, options=options)
	ra, rb = [(m.doc_index, m.slice_id, m.score) for m in a], [(m.doc_index, m.slice_id, m.score) for m in b]
	if ra != rb:
		print("query", len(text.split()), "tokens:", text[:60])
		for x, y, m in zip(ra, rb, a):
			if x != y:
				print("   find_many", x, "find", y, "slice of", m._len_s, "tokens")
