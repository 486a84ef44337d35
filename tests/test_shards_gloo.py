"""CPU tier: the N > 1 path.  Two gloo ranks, each with half of the corpus; the all-gather +
merge must give every rank exactly the result set of the unsharded corpus."""

import os
import socket
import subprocess
import sys

import numpy as np

from vectorian_amd import shards, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges():
	assert shards.shard_ranges(10, 3) == [(0, 4), (4, 7), (7, 10)]
	assert shards.shard_ranges(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]


def test_shard_ranges_by_tokens():
	# equal slices: the cuts of shard_ranges; ragged slices: token counts within one slice of the ideal; always a partition
	assert shards.shard_ranges_by_tokens([4] * 12, 3) == [(0, 4), (4, 8), (8, 12)]
	assert shards.shard_ranges_by_tokens([], 3) == [(0, 0)] * 3
	r = shards.shard_ranges_by_tokens([5, 5], 4)      # more ranks than slices: some ranks hold nothing
	assert r[0][0] == 0 and r[-1][1] == 2 and all(r[i][1] == r[i + 1][0] for i in range(3)) and max(b - a for a, b in r) == 1
	rng = np.random.default_rng(0)
	for world in (2, 3, 8):
		lens = rng.integers(1, 65, size=5000)
		lens[:1000] = 64      # long slices first: equal slice counts would give rank 0 far more tokens
		r = shards.shard_ranges_by_tokens(lens, world)
		assert r[0][0] == 0 and r[-1][1] == len(lens) and all(r[i][1] == r[i + 1][0] for i in range(world - 1))
		tok = [int(lens[a:b].sum()) for a, b in r]
		assert max(tok) - min(tok) <= 2 * 64, tok


def test_pack_roundtrip():
	from vectorian_amd import core
	t = core.TopK(5, 3)
	t.n = 2
	t.score[:2] = [0.75, -0.5]; t.raw_score[:2] = [2.25, -1.5]; t.sentence[:2] = [2 ** 33 + 5, 7]
	t.mapping[:2] = [[4, -1, 9], [0, 1, 2]]; t.edge_sim[:2] = [[0.5, 0, 0.25], [1, 1, 1]]
	u = shards.unpack_topk(shards.pack_topk(t, 100, 5), 3)
	assert u.n == 2 and list(u.sentence[:2]) == [2 ** 33 + 105, 107]
	assert (u.mapping[:2] == t.mapping[:2]).all() and (u.edge_sim[:2] == t.edge_sim[:2]).all()
	assert (u.score[:2] == t.score[:2]).all() and (u.raw_score[:2] == t.raw_score[:2]).all()


def test_two_rank_gloo_matches_unsharded(tmp_path, oracle):
	with socket.socket() as s:
		s.bind(("127.0.0.1", 0))
		port = s.getsockname()[1]
	env = dict(os.environ, OMP_NUM_THREADS="1")
	cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
		"--master-addr", "127.0.0.1", "--master-port", str(port),
		os.path.join(ROOT, "tests", "shard_worker.py"), str(tmp_path)]
	r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
	assert r.returncode == 0, r.stderr[-3000:]
	got = [np.load(tmp_path / f"rank{k}.npz") for k in range(2)]

	corpus = synth.make_contextual_corpus(900, 2, 30, 800, 64)
	queries = synth.make_queries(corpus, 3, 6)
	Xb, _ = oracle.normalize_rows_bf16(corpus["X"])
	for qi, q in enumerate(queries):
		Qb, _ = oracle.normalize_rows_bf16(q["vectors"])
		for name, loc in (("local", 0), ("global", 1)):
			ref = oracle.find(layout=oracle.LAYOUT_CONTEXTUAL, d=64, sent_off=corpus["sent_off"], X=Xb, Q=Qb, locality=loc,
				gap_s=0.1, gap_t=0.1, max_matches=12, min_score=0.0 if loc == 0 else -100.0)
			for g in got:
				assert (g[f"{qi}_{name}_sentence"] == ref["sentence"]).all()
				assert (g[f"{qi}_{name}_score"] == ref["score"]).all()
				assert (g[f"{qi}_{name}_mapping"] == ref["mapping"]).all()
				if loc == 0:   # the same queries exchanged together in one all-gather
					assert (g[f"{qi}_batched_sentence"] == ref["sentence"]).all()
					assert (g[f"{qi}_batched_mapping"] == ref["mapping"]).all()


def test_sharded_index(tmp_path):
	"""HipBruteForceIndex(shard=(rank, world)): two gloo ranks, each with half of the slices in its backend; Index.find
	returns on both ranks exactly what the unsharded index returns"""
	import json
	sys.path.insert(0, os.path.join(ROOT, "tests"))
	import shard_index_worker
	with socket.socket() as s:
		s.bind(("127.0.0.1", 0))
		port = s.getsockname()[1]
	env = dict(os.environ, OMP_NUM_THREADS="1")
	cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
		"--master-addr", "127.0.0.1", "--master-port", str(port),
		os.path.join(ROOT, "tests", "shard_index_worker.py"), str(tmp_path)]
	r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
	assert r.returncode == 0, r.stderr[-3000:]
	ref = {}
	for strategy in ("align", "wrd", "rwmd"):   # the transport strategies: the flows of the merged winners are stated on every rank
		index, queries = shard_index_worker.build(None, strategy)
		ref[strategy] = shard_index_worker.answers(index, queries)
	assert ref["align"][0][0][:2] == [1, 3] and abs(ref["align"][0][0][2] - 1.0) < 1e-2
	assert ref["wrd"][0][0][3] == "dense" and ref["rwmd"][0][0][3] == "sparse"
	# find_many on the sharded index: the batched calls (16 alignment queries / 256 relaxed-WMD queries per backend call, one
	# all-gather per chunk) and the one-query-at-a-time pipeline (exchanges issued in query order) return what the unsharded
	# index returns, flows included, on both ranks; progress is reported per chunk
	ref["find_many"] = shard_index_worker.find_many_answers(None)
	for strategy in ("align", "rwmd"):
		assert any(len(rows) == 5 for rows in ref["find_many"][strategy])
		assert ref["find_many"][strategy + "_pipelined"] == ref["find_many"][strategy][:9]
		assert ref["find_many"][strategy + "_progress"][-1] == 1.0
	assert ref["find_many"]["rwmd"][0][0][3] == "sparse"
	# whole documents as slices (longer than VK_MAX_SENT_LEN): the planted document wins on every rank, rows of the merged winners included
	ref["documents"] = shard_index_worker.documents_answers(None)
	assert [rows[0][:2] for rows in ref["documents"]["align"]] == [[1, 0], [4, 0], [6, 0]] and ref["documents"]["rwmd"][0][0][3] == "sparse"
	# find_many(debug = AllSlices(hook)): every rank's hook walks the slices of its own shard per query (scores as on one GPU); the
	# two ranks' calls, rank 0 first, are the calls of the unsharded index; the merged matches are the same on both ranks
	ref_hooks = shard_index_worker.hook_answers(None)
	got_hooks = [json.load(open(tmp_path / f"index_rank{k}.json"))["hooks"] for k in range(2)]
	for strategy in ("align", "rwmd"):
		for k in range(2):
			assert got_hooks[k][strategy]["matches"] == ref_hooks[strategy]["matches"]
		for qi, want in enumerate(ref_hooks[strategy]["calls"]):
			assert got_hooks[0][strategy]["calls"][qi] + got_hooks[1][strategy]["calls"][qi] == want
	for k in range(2):
		got = json.load(open(tmp_path / f"index_rank{k}.json"))
		got.pop("hooks")
		# Query.abort raised on rank 1 only: no rank hangs in the collective, and the query yields no matches on either rank
		assert got.pop("abort") == {"batched": [0] * 6, "pipelined": [0] * 6, "find": 0, "rwmd": [0] * 12}
		assert got == ref
