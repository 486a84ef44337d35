"""replays one seed of tests/test_gpu_shard_sweep.py under torchrun (2 or 3 ranks on one GPU, gloo) and says which of find_many / find, sharded / unsharded
disagree, and whose similarity rows a sharded winner carries:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29512 tools/probe/shard_case.py SEED"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import shard_sweep_worker as W
import torch.distributed as dist
from vectorian_amd import core
core.init(0)
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
seed = int(sys.argv[1])
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
	whole.close(); whole = part.index(sim, nlp=nlp, **index_kw)
mine = part.index(sim, nlp=nlp, shard=(rank, world), **index_kw)
texts = []
for _ in range(int(rng.integers(3, 9))):
	doc = session.documents[int(rng.integers(0, len(session.documents)))]
	len_t = int(rng.integers(1, 13)) if rng.random() < 0.85 else int(rng.integers(17, 40))
	if len(doc.tokens) > len_t and rng.random() < 0.7:
		a0 = int(rng.integers(0, len(doc.tokens) - len_t)); texts.append(" ".join(doc.tokens[a0:a0 + len_t]))
	else:
		texts.append(" ".join(words[int(i)] for i in rng.integers(0, len(words), size=len_t)))
options = {}
if rng.random() < 0.2: options["pos_filter"] = ["DET"]
if is_align and rng.random() < 0.2: options["submatch_weight"] = float(rng.choice([0.5, 1.0, 2.0]))
n = int(rng.choice([1, 5, 12])); min_score = 0.0 if rng.random() < 0.7 else -100.0
fm_mine = mine.find_many(texts, n=n, min_score=min_score, options=options)
fm_whole = whole.find_many(texts, n=n, min_score=min_score, options=options)
out = []
for t, a, b in zip(texts, fm_mine, fm_whole):
	f_m = mine.find(t, n=n, min_score=min_score, options=options)
	f_w = whole.find(t, n=n, min_score=min_score, options=options)
	out.append(dict(q=len(t.split()), many_sharded_vs_find_whole=W.same(a, f_w, True), many_whole_vs_find_whole=W.same(b, f_w, True), find_sharded_vs_find_whole=W.same(f_m, f_w, True),
		lens=[m._len_s for m in f_w], n_local=mine._n_local, slices=whole.n_slices))
a, fw = fm_mine[0], whole.find(texts[0], n=n, min_score=min_score, options=options)
wa, ww = list(a)[0]._w, list(fw)[0]._w; fw = list(fw); a = list(a)
ra, rw = wa.top.sim_rows, ww.top.sim_rows
lt = len(texts[0].split())
for r in range(2):
	if rank == r:
		print("rank", rank, "slice_off", mine._slice_off, "n_local", mine._n_local, "rows shapes", ra.shape, rw.shape, flush=True)
		for j in range(min(len(a), len(fw))):
			ls = fw[j]._len_s
			hits = [jj for jj in range(len(fw)) if fw[jj]._len_s >= 1 and np.array_equal(ra[j][:min(ls, ra.shape[1]), :lt], rw[jj][:min(ls, ra.shape[1]), :lt])]
			print("  winner", j, "slice", fw[j].slice_id, "global", int(wa.sent[j]) if hasattr(wa, "sent") else None, "len", ls, "rows equal the unsharded rows of winner", hits, "nonzero rows", int((np.abs(ra[j]).sum(axis=1) > 0).sum()), flush=True)
	dist.barrier()
dist.barrier(); dist.destroy_process_group()
