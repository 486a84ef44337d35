"""find_many of HipBruteForceIndex: several queries in flight on as many handles of the resident corpus (worker threads, one per
handle), batched calls (vk_query_batch), results finished in query order on the calling thread."""

import numpy as np

from vectorian_amd import core


class LanesMixin:
	"""the threads and handles behind HipBruteForceIndex.find_many"""

	def _handles(self, n):
		"""the resident corpus and n - 1 further handles on it (own stream and workspaces each)"""
		if not hasattr(self._corpus, "view"):
			return [self._corpus]
		while len(self._views) < n - 1:
			self._views.append(self._corpus.view())
		return [self._corpus] + self._views[:max(0, n - 1)]

	def _in_order(self, n_items, n_lanes, work, finish, group=1):
		"""work(item, lane) on n_lanes worker threads (lane l serves items l, l + n_lanes, .. in order: a handle is used by one
		thread); finish(first, outputs) on the CALLING thread for runs of `group` consecutive items, in item order -- the
		collectives of a sharded index must be issued in the same order on every rank, whatever the threads' timing."""
		from concurrent.futures import Future, ThreadPoolExecutor
		futs = [Future() for _ in range(n_items)]

		def lane(l):
			for i in range(l, n_items, n_lanes):
				try:
					futs[i].set_result(work(i, l))
				except BaseException as e:   # surfaces on the calling thread, in order
					futs[i].set_exception(e)
		with ThreadPoolExecutor(max_workers=max(1, n_lanes)) as pool:
			for l in range(n_lanes):
				pool.submit(lane, l)
			for a in range(0, n_items, group):
				finish(a, [futs[i].result() for i in range(a, min(a + group, n_items))])

	def _find_pipelined(self, queries, in_flight, progress):
		"""find_many, one vk_query per query: the local part on up to `in_flight` handles, the exchange (sharded) in query order"""
		filtered = any(self._filter_masks(q.options) is not None for q in queries)
		handles = self._handles(1 if (filtered or in_flight < 2) else in_flight)
		results = [None] * len(queries)
		done = [0]

		def work(i, l):
			return self._find_local(queries[i], corpus=handles[l] if not filtered else None)

		def finish(first, locals_):
			merged = self._merge_ranks(locals_)
			for j, (loc, top) in enumerate(zip(locals_, merged)):
				results[first + j] = self._finish_find(queries[first + j], loc, top)
			done[0] += len(locals_)
			if progress:
				progress(done[0] / len(queries))
		self._in_order(len(queries), len(handles), work, finish, group=4 if self._shard is not None else 1)
		return results

	def _batch_plan(self, queries, options):
		"""chunks of queries that can go to the backend in one call each, or None"""
		if (self._filter_masks(options) is not None or options.get("debug") is not None
				or not hasattr(self._corpus, "query_batch") or len(queries) < 2):
			return None
		# (static embeddings since round 4: relaxed-WMD queries share one table over the vocabulary and one gather pass over the token
		# ids per call -- vk_rwmd_static32_kernel; alignments are accepted and answered query by query inside the call)
		args, _ = self._backend_args(queries[0].options)
		if "tag_weighted" in args or args["submatch_weight"] != 0.0:
			return None
		alg = args["algorithm"]
		if alg == core.VK_ALG_ALIGN:
			per_call = 16
		elif alg == core.VK_ALG_RWMD and not args.get("wmd_full"):
			per_call = 256
		else:
			return None   # exact transport: per query (bound pass + solver rounds)
		return [range(a, min(a + per_call, len(queries))) for a in range(0, len(queries), per_call)]

	def _find_batches(self, queries, batches, in_flight, progress=None):
		"""find_many through vk_query_batch: the chunks of `batches` on up to two handles of the resident corpus; a sharded index
		exchanges the result sets of a chunk in one all-gather (config 4 sharded: each rank's GEMM, one all-gather of 256 x k records)"""
		emb = self._embedding
		args, gaps = self._backend_args(queries[0].options)
		prepared = [q.prepare(self._nlp) for q in queries]
		handles = self._handles(max(1, min(2, in_flight, len(batches))))
		results = [None] * len(queries)
		k = args["max_matches"]
		done = [0]

		def work(b, l):
			"""local result sets of chunk b: (indices of its non-empty queries, their TopKs, aborted)"""
			idx = [i for i in batches[b] if len(prepared[i]) > 0]
			if not idx and self._shard is None:
				return idx, [], False
			qvs = [emb.encode_tokens(prepared[i].tokens) for i in idx]
			ids = dict(token_ids=[prepared[i].token_ids for i in idx]) if emb.is_static else {}   # static layout: sim[id(t_j)][j] = 1 needs the words' ids
			try:
				tops = handles[l].query_batch([np.ascontiguousarray(qv.unmodified, dtype=np.float32) for qv in qvs], q_normalize=True,
					boost=self._dev_boost, want_flow=True, abort_flag=queries[idx[0]]._abort, **ids, **args) if idx else []
			except core.VkError as e:
				if e.status != core.VK_ERR_ABORTED:
					raise
				# Query.abort: no matches -- on a sharded index this rank still joins the exchange (with empty result sets and the
				# flag raised), or the ranks that did not see the flag in time would wait for it in the collective forever
				return idx, [self._empty_top(len(prepared[i]), args) for i in idx], True
			return idx, tops, False

		def finish(b, out):
			(idx, tops, aborted), = out
			merged = self._merge_ranks([dict(top=t, aborted=aborted, args=args) for t in tops])
			for i in batches[b]:
				results[i] = []
			for i, top in zip(idx, merged):
				if top is not None:
					results[i] = self._matches_from_topk(prepared[i], top, gaps, args, None, None, None)   # (magnitudes: WRD only, which does not share calls)
			done[0] += len(batches[b])
			if progress:
				progress(done[0] / len(queries))
		self._in_order(len(batches), len(handles), work, finish)
		return results

	def _empty_top(self, len_t, args):
		return core.TopK(max(1, args["max_matches"]), len_t)
