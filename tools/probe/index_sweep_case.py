"""replays one seed of tests/test_gpu_index_sweep.py and prints where the HIP index and the oracle double disagree (score vectors
of all slices, the first differing slice): python tools/probe/index_sweep_case.py SEED"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_index_sweep as T
from fake_backend import OracleCorpus
from vectorian_amd import core
from vectorian_amd.sim import CosineSim, EmbeddingTokenSim, OptimizedSpanSim
core.init(0)
seed = int(sys.argv[1])
rng = np.random.default_rng(77000 + seed)
session, emb, nlp, words = T.build_session(rng)
strategy, is_align = T.random_strategy(rng)
kw = {}
if rng.random() < 0.3:
	kw = dict(tag_weights={t: float(rng.uniform(0.25, 2.5)) for t in rng.choice(T.TAGS, size=3, replace=False)},
		pos_mismatch_penalty=float(rng.uniform(0, 0.5)), similarity_threshold=float(rng.uniform(0, 0.2)))
sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), strategy, **kw)
if rng.random() < 0.7:
	part = session.partition("sentence", int(rng.integers(1, 4)), int(rng.integers(1, 3)))
else:
	part = session.partition("token", int(rng.integers(4, 20)), int(rng.integers(1, 8)))
gpu = part.index(sim, nlp=nlp)
index_kw = {}
if rng.random() < 0.3:
	index_kw["saliency"] = rng.uniform(0.5, 1.5, size=gpu.n_slices).astype(np.float32)
	gpu = part.index(sim, nlp=nlp, **index_kw)
cpu = part.index(sim, nlp=nlp, corpus_factory=OracleCorpus, **index_kw)
print("strategy", getattr(strategy, "_options", type(strategy).__name__), "tags", kw, "part", part.to_args(), type(emb).__name__, "slices", gpu.n_slices,
	"max len", int((gpu._slice_end - gpu._slice_start).max()), "V", session.vocab.size)
for _ in range(2):
	doc = session.documents[int(rng.integers(0, len(session.documents)))]
	len_t = int(rng.integers(1, 13)) if rng.random() < 0.8 else int(rng.integers(17, 40))
	if len(doc.tokens) > len_t and rng.random() < 0.7:
		a0 = int(rng.integers(0, len(doc.tokens) - len_t)); text = " ".join(doc.tokens[a0:a0 + len_t])
	else:
		text = " ".join(words[int(i)] for i in rng.integers(0, len(words), size=len_t))
	options = {}
	if rng.random() < 0.25:
		options["pos_filter"] = [str(x) for x in rng.choice(["DET", "PUNCT", "ADJ"], size=int(rng.integers(1, 3)), replace=False)]
	if is_align and rng.random() < 0.2:
		options["submatch_weight"] = float(rng.choice([0.5, 1.0, 2.0]))
	n = int(rng.choice([1, 5, 12])); min_score = 0.0 if rng.random() < 0.7 else -100.0
	a = gpu.find(text, n=n, min_score=min_score, options=options); b = cpu.find(text, n=n, min_score=min_score, options=options)
	q = gpu.make_query(text).prepare(nlp)
	print("query", text, "ids", q.token_ids.tolist(), "tags", q.tags, options, n, min_score)
	if not options:
		sg, sc = gpu._corpus.last_scores(), cpu._corpus.last_scores()
		bad = np.nonzero(np.abs(sg - sc) > 2e-5)[0]
		lens = (gpu._slice_end - gpu._slice_start)
		print("  slices whose scores differ:", len(bad), "of", len(sg), "max diff", float(np.nanmax(np.abs(sg - sc))))
		for g in bad[:3]:
			a0, b0 = int(gpu._slice_start[g]), int(gpu._slice_end[g])
			ids = gpu._token_ids[a0:b0] if gpu._token_ids is not None else None
			tg = gpu._tag_codes[a0:b0].tolist() if gpu._tag_codes is not None else None
			print("  slice", int(g), "len", b0 - a0, "ids", None if ids is None else ids.tolist(), "tags", tg, "hip", sg[g], "double", sc[g])
		if len(bad):
			np.set_printoptions(linewidth=250, precision=6, suppress=True)
			ma = {m.slice_id: m for m in a}; mb = {m.slice_id: m for m in b}
			for sid in sorted(set(ma) & set(mb)):
				x, y = ma[sid], mb[sid]
				if abs(x.score - y.score) < 2e-5: continue
				print("  slice", sid, "score", x.score, y.score)
				print("  target hip", x.flow["target"].tolist(), "\n  target dbl", y.flow["target"].tolist())
				print("  dist hip", x.flow["dist"], "\n  dist dbl", y.flow["dist"])
