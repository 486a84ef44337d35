"""-m gpu: the exchange step of the sharded path (vectorian_amd/shards.py) on the real backend -- RCCL (torch.distributed
backend "nccl") and the HIP library in ONE process, on result sets that come out of core.Corpus.  One rank: the box has one
GPU; the collective, its stream, the pinned staging and the merge are the code every rank of an 8-GPU job runs
(tests/test_shards_gloo.py covers world_size 2 on the CPU)."""

import os

import numpy as np
import pytest

from vectorian_amd import shards, synth

from helpers import hip_contextual_corpus, prep_query

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nccl_group(hip):
	import torch
	import torch.distributed as dist
	for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29541"), ("RANK", "0"), ("WORLD_SIZE", "1")):
		os.environ.setdefault(k, v)
	dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
	yield dist
	dist.destroy_process_group()


def test_allgather_of_hip_result_sets_over_rccl(hip, nccl_group):
	import torch
	corpus = synth.make_contextual_corpus(3000, 4, 40, 2000, 64)
	c = hip_contextual_corpus(hip, corpus)
	k = 7
	tops = []
	for q in synth.make_queries(corpus, 5, 6):
		tops.append(c.query(prep_query(q), q_normalize=False, locality=hip.Locality.LOCAL, gap_s=0.1, gap_t=0.1, max_matches=k, want_flow=True))
	offset = 100000
	# several result sets in one all-gather, in flight while another query is scored
	h = shards.allgather_start(tops[:4], offset, k, device=torch.device("cuda", 0))
	extra = c.query(prep_query(synth.make_queries(corpus, 1, 6, seed=9)[0]), q_normalize=False, gap_s=0.1, gap_t=0.1, max_matches=k)
	assert extra.n > 0
	merged = shards.allgather_finish(h)
	assert len(merged) == 4
	for t, m in zip(tops, merged):
		assert m.n == t.n
		np.testing.assert_array_equal(m.score[:m.n], t.score[:t.n])
		np.testing.assert_array_equal(m.sentence[:m.n], t.sentence[:t.n] + offset)
		np.testing.assert_array_equal(m.mapping[:m.n], t.mapping[:t.n])
		np.testing.assert_array_equal(m.edge_sim[:m.n], t.edge_sim[:t.n])
	# a single set, and the blocking form
	one = shards.allgather_finish(shards.allgather_start(tops[4], offset, k, device=torch.device("cuda", 0)))
	np.testing.assert_array_equal(one.sentence[:one.n], tops[4].sentence[:tops[4].n] + offset)
	blk = shards.allgather_merge(tops[4], offset, k)
	np.testing.assert_array_equal(blk.score[:blk.n], tops[4].score[:tops[4].n])
	c.close()


def test_allgather_merge_carries_rows_of_long_transport_winners(hip, nccl_group):
	"""the blocking exchange with the flows' payload: similarity rows and plans sized by the longest slice (here 150 tokens)"""
	rng = np.random.default_rng(3)
	lens = np.array([12, 150, 30, 7, 64, 90, 20])
	off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
	X = rng.standard_normal((int(off[-1]), 48)).astype(np.float32)
	c = hip.Corpus(layout=hip.VK_LAYOUT_CONTEXTUAL, d=48, n_tokens=X.shape[0], n_sentences=len(lens), keep_magnitudes=True)
	c.append_vectors(X, normalize=True)
	c.set_sentences(off)
	c.finalize()
	top = c.query(X[off[1] + 3:off[1] + 9], algorithm=hip.VK_ALG_WRD, q_normalize=True, max_matches=5, min_score=-1.0, want_flow=True)
	assert top.sim_rows.shape[1] == 192 and int(top.sentence[0]) == 1
	merged = shards.allgather_merge(top, 1000, 5)
	assert merged.n == top.n and merged.sim_rows.shape == top.sim_rows.shape
	np.testing.assert_array_equal(merged.sentence[:merged.n], top.sentence[:top.n] + 1000)
	np.testing.assert_array_equal(merged.sim_rows[:merged.n], top.sim_rows[:top.n])
	np.testing.assert_array_equal(merged.plan[:merged.n], top.plan[:top.n])
	assert abs(float(merged.plan[0].sum()) - 1.0) < 1e-5      # the plan of the 150-token winner moves all the mass
	c.close()


def test_sharded_index_on_hip_over_rccl_equals_the_unsharded_index(hip, nccl_group):
	"""HipBruteForceIndex(shard = (0, 1), group = the RCCL group) on the HIP backend: Index.find, the pipelined and the batched
	find_many (alignments: 16 per vk_query_batch call; relaxed WMD: one GEMM pass per chunk) go through the exchange -- all-gather
	of the records on the device, the rows of the merged transport winners in one all-reduce -- and return what the unsharded
	index returns, flows included"""
	from test_host_api import contextual_toy
	from vectorian_amd import alignment
	from vectorian_amd.sim import CosineSim, EmbeddingTokenSim, OptimizedSpanSim
	session, emb, docs = contextual_toy(n_docs=4, sents=60, d=64)
	texts = [" ".join(docs[i % 4].tokens[5 * i:5 * i + 3 + i % 4]) for i in range(23)]
	for strategy in (alignment.LocalAlignment(gap=alignment.smooth_gap_cost(5)), alignment.WordMoversDistance.rwmd("nbow"), alignment.WordRotatorsDistance()):
		sim = OptimizedSpanSim(EmbeddingTokenSim(emb, CosineSim()), strategy)
		whole = session.index(sim)
		shard = session.index(sim, shard=(0, 1), group=None)
		pairs = [(whole.find(texts[3], n=5), shard.find(texts[3], n=5))]
		pairs += list(zip(whole.find_many(texts, n=5), shard.find_many(texts, n=5)))
		pairs += list(zip(whole.find_many(texts[:7], n=5, batch=False), shard.find_many(texts[:7], n=5, batch=False)))
		for a, b in pairs:
			assert [(m.doc_index, m.slice_id, m.score) for m in a] == [(m.doc_index, m.slice_id, m.score) for m in b]
			for x, y in zip(a, b):
				fx, fy = x.flow, y.flow
				assert fx["type"] == fy["type"]
				for key in fx:
					if key != "type":
						np.testing.assert_array_equal(fx[key], fy[key])
		whole.close(); shard.close()
