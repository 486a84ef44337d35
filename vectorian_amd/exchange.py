"""The sharded index: ResultSet.extend across the ranks (vectorian/core/cpp/result_set.h:70-93) as one all-gather of k records per
query (vectorian_amd/shards.py), the similarity rows of merged transport winners in one all-reduce."""

from vectorian_amd import core


class ShardExchangeMixin:
	"""HipBruteForceIndex(shard = (rank, world)): merging the ranks' result sets"""

	def _merge_ranks(self, locals_):
		"""local results (dicts of _find_local / _find_batches: top, aborted, args) -> the result sets `find` goes on with, None
		for an aborted query.  One GPU: the local ones.  Sharded: ResultSet.extend across the ranks (result_set.h:70-93) -- ONE
		all-gather of the k-record result sets of all the queries handed over (local slice ids -> global), every rank ending
		with the same sets; the similarity rows / plans of transport winners follow in one all-reduce for the merged winners
		only (shards.rows_allreduce), so that flows are stated as on one GPU.  A rank whose query was aborted joins with an
		empty set and a flag: the query then yields no matches on any rank."""
		live = [x for x in locals_ if x is not None]
		if self._shard is None or not live:
			return [None if (x is None or x["aborted"]) else x["top"] for x in locals_]
		from vectorian_amd import shards
		args = live[0]["args"]
		k = args["max_matches"]
		# one exchange per record size (a record holds the query's columns rounded up to 16: queries of 5 and of 20 tokens
		# travel apart), in the order of the sizes -- the same on every rank
		by_size = {}
		for i, x in enumerate(live):
			by_size.setdefault(shards._layout(x["top"].len_t), []).append(i)
		done = [None] * len(live)
		for size in sorted(by_size):
			part = [live[i] for i in by_size[size]]
			tops = [x["top"] for x in part]
			h = shards.allgather_start(tops, self._slice_off, k, group=self._group, device=self._xdev,
				flags=[shards.FLAG_ABORTED if x["aborted"] else 0 for x in part])
			merged = shards.allgather_finish(h)
			# (every rank must take the same decision: by the algorithm, not by what this rank's sets happen to hold -- an aborted
			# or empty local set has no rows)
			transport = args.get("algorithm", core.VK_ALG_ALIGN) != core.VK_ALG_ALIGN
			with_rows = [i for i, x in enumerate(part) if transport or x.get("hook") is not None]   # (a debug hook asks for the rows of alignments too)
			if with_rows:
				lens = [self._slice_end[merged[i].sentence[:merged[i].n]] - self._slice_start[merged[i].sentence[:merged[i].n]] for i in with_rows]
				exact = args.get("algorithm") == core.VK_ALG_WRD or bool(args.get("wmd_full"))
				shards.rows_allreduce([tops[i] for i in with_rows], [merged[i] for i in with_rows], self._slice_off, self._n_local, lens,
					group=self._group, device=self._xdev, with_plan=exact)
			for i, m, f in zip(by_size[size], merged, h["flags_out"]):
				done[i] = None if (f & shards.FLAG_ABORTED) else m
		out = iter(done)
		return [None if x is None else next(out) for x in locals_]
