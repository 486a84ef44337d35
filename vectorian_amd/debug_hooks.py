"""The debug hook of a search (call_debug_hook, vectorian/core/cpp/metric/alignment.h:145-173; the solvers' hooks, alignment/wrd.h:
31-59, alignment/wmd.h:147-181): for the k winners by default, for EVERY slice with debug = AllSlices(hook)."""

import numpy as np

from vectorian_amd import core
from vectorian_amd.flows import _rows_room


class AllSlices:
	"""debug = AllSlices(hook): call the debug hook for EVERY slice the search scores, as the reference does (call_debug_hook,
	vectorian/core/cpp/metric/alignment.h:145-173; match/matcher_impl.h:137-170), instead of for the k winners only.  The fused
	scoring kernel keeps no per-slice matrices, so the slices are restated `chunk` at a time after the search: exact, opt-in, slow."""

	all_slices = True

	def __init__(self, hook, chunk=512):
		if not callable(hook):
			raise TypeError("debug must be callable: hook(name, data)")
		self.hook, self.chunk = hook, int(chunk)

	def __call__(self, name, data):
		return self.hook(name, data)



class DebugHookMixin:
	"""debug-hook methods of HipBruteForceIndex (they read the index's slice tables and call its corpus handles)"""

	def _call_debug_hook(self, hook, p_query, top, matches, args):
		"""the reference calls hook(name, data) for EVERY slice it scores (call_debug_hook, metric/alignment.h:145-173: slice,
		similarity [len_s x len_t], flow, score = the aligner's score; WMD: 'alignment/word-movers-distance/make' with score and
		worst_score, :600-607).  The scoring kernel keeps no per-slice matrices; by default the hook is called for the k winners, best
		first, with the same keys (`similarity` is None for winners longer than the rows the backend returned).  debug =
		AllSlices(hook) walks every slice instead (_call_debug_hook_all_slices)."""
		alg = args.get("algorithm", core.VK_ALG_ALIGN)
		worst = float(matches[-1].score) if len(matches) >= args["max_matches"] else float(args["min_score"])
		for i, m in enumerate(matches):
			if alg == core.VK_ALG_WRD or (alg == core.VK_ALG_RWMD and args.get("wmd_full")):
				data = self._solver_debug_data(p_query, top, i, m, args)
				if data is not None:
					hook("alignment/word-rotators-distance/solver" if alg == core.VK_ALG_WRD else "alignment/word-movers-distance/solver", data)
			if alg == core.VK_ALG_RWMD:
				hook("alignment/word-movers-distance/make", {"score": m.score, "worst_score": worst, "slice": m.slice_id, "flow": m.flow})
				continue
			if alg == core.VK_ALG_WRD:
				continue
			sim = None
			if getattr(top, "sim_rows", None) is not None and m._len_s <= _rows_room(top, i):
				sim = top.sim_rows[i][:m._len_s if m._index_map is None else len(m._index_map), :len(p_query)].copy()
			hook("alignment", {"slice": m.slice_id, "similarity": sim, "flow": m.flow, "score": m.raw_score})

	def _token_magnitudes(self, g):
		"""|x| of the tokens of slice g, as Slice::magnitude_s hands them to WRD (metric/contextual.cpp:49-54, metric/static.cpp:69-73)"""
		a, b = int(self._slice_start[g]), int(self._slice_end[g])
		emb = self._embedding
		if emb.is_static:
			if getattr(self, "_vocab_mag", None) is None:
				self._vocab_mag = np.asarray(emb.encode_tokens(self.session.vocab.tokens).magnitudes, dtype=np.float32)
			return self._vocab_mag[self._token_ids[a:b]]
		from vectorian_amd.embedding import Vectors
		doc_base = 0
		for doc in self.session.documents:
			if a < doc_base + doc.n_tokens:
				return np.asarray(Vectors(doc.contextual_vectors(emb.name)[a - doc_base:b - doc_base]).magnitudes, dtype=np.float32)
			doc_base += doc.n_tokens
		return None

	def _solver_debug_data(self, p_query, top, i, m, args):
		"""what the exact solvers hand the debug hook for a slice (WRD::call_debug_hook, vectorian/core/cpp/alignment/wrd.h:31-59;
		FullSolver::call_debug_hook, alignment/wmd.h:147-181), stated for a WINNER from the similarity rows and the optimal plan the
		backend returned: tokens of both sides, masses, the distance matrix over the joint problem, the plan G and its cost"""
		if getattr(top, "sim_rows", None) is None or getattr(top, "plan", None) is None or m._index_map is not None:
			return None
		len_s, len_t = m._len_s, len(p_query)
		if len_s > _rows_room(top, i):
			return None
		g = int(m._w.sent[m._i])
		a = int(self._slice_start[g])
		doc_tokens = m.prepared_doc.tokens
		S = top.sim_rows[i][:len_s, :len_t]
		G_ts = top.plan[i][:len_t, :len_s]
		ids_s = self._token_ids[a:a + len_s].tolist() if self._token_ids is not None else list(range(len_s))
		data = {
			"s": {"id": ids_s, "text": list(doc_tokens[m._token_at:m._token_at + len_s])},
			"t": {"id": [int(x) for x in p_query.token_ids], "text": list(p_query.tokens)}}
		n = len_s + len_t
		D = np.ones((n, n), dtype=np.float32)
		D[:len_t, len_t:] = np.maximum(0.0, 1.0 - S.T)
		G = np.zeros((n, n), dtype=np.float32)
		G[:len_t, len_t:] = G_ts
		if args.get("algorithm") == core.VK_ALG_WRD:
			mag_s, mag_t = np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.float32)
			ms = self._token_magnitudes(g)
			if ms is None:
				return None
			mag_t[:len_t] = np.asarray(self._embedding.encode_tokens(p_query.tokens).magnitudes, dtype=np.float32)
			mag_s[len_t:] = ms
			if args.get("wrd_normalize", True):
				mag_t /= mag_t.sum()
				mag_s /= mag_s.sum()
			data.update(mag_s=mag_s, mag_t=mag_t, D=D, elapsed_microseconds=0,
				solution={"G": G, "cost": float((D * G).sum()), "type": "optimal"})
			return data
		# full WMD over positions (every position its own vocabulary entry on the device; the host states flows over the joint vocabulary)
		nbow = args["rwmd"][2]
		data.update(bow_s=np.full(len_s, 1.0 / len_s if nbow else 1.0, dtype=np.float32), bow_t=np.full(len_t, 1.0 / len_t if nbow else 1.0, dtype=np.float32),
			D=D, G=G, flow_by_pos=m.flow["flow"], dist_by_pos=m.flow["dist"], score=m.raw_score)
		return data

	def _call_debug_hook_all_slices(self, hook, local, matches):
		"""debug = AllSlices(hook): the hook contract of the reference in full -- one call per slice this process scores, in slice
		order, with the reference's keys.  Alignments ('alignment': slice, similarity, flow, score; metric/alignment.h:145-173): the
		slices are stated `hook.chunk` at a time by the traceback kernel (vk_query_desc.only_slices: aligner score, mapping, edge
		similarities and the similarity rows, canonical arithmetic), whatever their score.  Relaxed WMD
		('alignment/word-movers-distance/make': score, worst_score; :600-607): the score of every slice restated from its canonical
		similarity rows (only_slices again: the reference's floats), with the worst score of a result set filled in slice order as
		upstream fills it.  Exact transports: every slice solved, the solver's hook per slice (tokens, masses, distance matrix, plan,
		cost).  Opt-in and slow (a Python call per slice); a sharded index walks its own slices on every rank.  A submatch weight does
		not change what the hook of an alignment is handed (the aligner's score, not Score::value)."""
		args, p_query, corpus = local["args"], local["p_query"], local["corpus"]
		alg = args.get("algorithm", core.VK_ALG_ALIGN)
		n_loc, off, len_t = self._n_local, self._slice_off, len(p_query)
		if alg == core.VK_ALG_RWMD and not args.get("wmd_full"):
			import heapq
			# the scores of all slices as the backend states winners: restated from canonical similarity rows in the reference's order
			# of operations (vk_query_desc.only_slices, `hook.chunk` slices per call) -- the floats upstream hands its hook; slices
			# longer than the rows a call returns keep the scoring pass's value
			# (read when the query ran, under the handle's lock: by now the lane's thread may have taken the handle to its next query)
			scores = local.get("all_scores")
			scores = np.array(corpus.last_scores(), dtype=np.float32) if scores is None else np.array(scores, dtype=np.float32)
			call, masks, lens_all = dict(local["call"]), local["masks"], self._slice_end[off:off + n_loc] - self._slice_start[off:off + n_loc]
			for a in range(0, n_loc, max(1, int(getattr(hook, "chunk", 512)))):
				ids = np.arange(a, min(n_loc, a + max(1, int(getattr(hook, "chunk", 512)))), dtype=np.int64)
				ids = ids[(lens_all[ids] > 0) & np.isfinite(scores[ids])]
				if len(ids) == 0:
					continue
				top = corpus.query(local["qv"].unmodified, q_normalize=True, boost=self._dev_boost, want_flow=True, only_slices=ids, **call)
				stated = np.isfinite(top.score[:top.n])
				scores[ids[:top.n][stated]] = top.score[:top.n][stated]
			heap, k, floor = [], args["max_matches"], float(args["min_score"])
			for g in range(n_loc):
				sc = float(scores[g])
				if not np.isfinite(sc):
					continue   # empty slice: Spans::iterate skips it (document.h:160-162)
				worst = heap[0] if len(heap) >= k else floor
				hook("alignment/word-movers-distance/make", {"score": sc, "worst_score": worst, "slice": self._slice_id[off + g]})
				if sc > worst:
					heapq.heappush(heap, sc) if len(heap) < k else heapq.heapreplace(heap, sc)
			return
		if alg == core.VK_ALG_WRD or (alg == core.VK_ALG_RWMD and args.get("wmd_full")):
			# exact transports: every slice solved (only_slices: no bound pass, nothing pruned), the solver's hook per slice
			# (WRD::call_debug_hook, wrd.h:31-59; FullSolver::call_debug_hook, wmd.h:147-181) and, full WMD, 'make' with the worst
			# score of a result set filled in slice order
			import heapq
			call, masks = dict(local["call"]), local["masks"]
			qmag = np.asarray(local["qv"].magnitudes, dtype=np.float32) if alg == core.VK_ALG_WRD else None
			lens_all = self._slice_end[off:off + n_loc] - self._slice_start[off:off + n_loc]
			heap, k, floor = [], args["max_matches"], float(args["min_score"])
			name = "alignment/word-rotators-distance/solver" if alg == core.VK_ALG_WRD else "alignment/word-movers-distance/solver"
			for a in range(0, n_loc, max(1, int(getattr(hook, "chunk", 512)))):
				ids = np.arange(a, min(n_loc, a + max(1, int(getattr(hook, "chunk", 512)))), dtype=np.int64)
				ids = ids[lens_all[ids] > 0]
				if len(ids) == 0:
					continue
				top = corpus.query(local["qv"].unmodified, q_normalize=True, boost=self._dev_boost, want_flow=True, only_slices=ids, **call)
				chunk = self._matches_from_topk(p_query, top, local["gaps"], args, qmag, masks, local["q_tag_codes"])
				for i, m in enumerate(chunk):
					if not np.isfinite(m.score):
						continue   # every token filtered out: the slice is not scored
					data = self._solver_debug_data(p_query, top, i, m, args)
					if data is not None:
						hook(name, data)
					if alg == core.VK_ALG_RWMD:
						worst = heap[0] if len(heap) >= k else floor
						hook("alignment/word-movers-distance/make", {"score": m.score, "worst_score": worst, "slice": m.slice_id, "flow": m.flow})
						if m.score > worst:
							heapq.heappush(heap, m.score) if len(heap) < k else heapq.heapreplace(heap, m.score)
			return
		if alg != core.VK_ALG_ALIGN:
			return self._call_debug_hook(hook, p_query, local["top"], matches, args)
		call = dict(local["call"], want_rows=len_t <= core.VK_MAX_QUERY_LEN)   # (queries of more than 64 tokens: no similarity rows)
		masks = local["masks"]
		for a in range(0, n_loc, max(1, int(getattr(hook, "chunk", 512)))):
			ids = np.arange(a, min(n_loc, a + max(1, int(getattr(hook, "chunk", 512)))), dtype=np.int64)
			lens = self._slice_end[off + ids] - self._slice_start[off + ids]
			ids = ids[lens > 0]   # Spans::iterate skips empty slices
			if len(ids) == 0:
				continue
			top = corpus.query(local["qv"].unmodified, q_normalize=True, boost=self._dev_boost, want_flow=True, only_slices=ids, **call)
			for i in range(top.n):
				g = off + int(ids[i])
				len_s = int(self._slice_end[g] - self._slice_start[g]) if not masks else len(self._index_map(g, masks))
				if len_s < 1:
					continue   # every token filtered out: the slice is not scored (FilteredSliceFactory, slice/static.h:366-416)
				target = top.mapping[i].astype(np.int16)
				matched = target >= 0
				flow = {"type": "injective", "target": target, "flow": matched.astype(np.float32),
					"dist": np.where(matched, 1.0 - top.edge_sim[i], 1.0).astype(np.float32)}
				sim = top.sim_rows[i][:len_s, :len_t].copy() if (top.sim_rows is not None and len_s <= top.sim_rows.shape[1]) else None
				hook("alignment", {"slice": self._slice_id[g], "similarity": sim, "flow": flow, "score": float(top.raw_score[i])})
