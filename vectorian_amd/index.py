"""Index family: `Index.find` and the MI355X brute-force index.

Mirrors vectorian/index.py: Query (:25), Match / PyMatch (:192, :382), Index (:434) with
`make_query` (:461-477) and `find` (:479-501), and BruteForceIndex (:509-560) -- whose `_find`
(one core.Document.find per document on a thread pool + ResultSet.extend) is what
HipBruteForceIndex replaces with one vk_query call against the corpus resident in HBM.

The parts live in modules of their own: query.py (Query, PreparedQuery), match.py (Match, HipMatch, PyMatch), options.py (option
dicts -> backend arguments), flows.py (transport flows of winners), lanes.py (find_many: handles, threads, batched calls),
exchange.py (the sharded index), debug_hooks.py (the debug hook, AllSlices); their names are re-exported here, where the reference
keeps them (vectorian/index.py).
"""

import collections
import contextlib
import logging
import time

import numpy as np

from vectorian_amd import core
from vectorian_amd.debug_hooks import AllSlices, DebugHookMixin
from vectorian_amd.exchange import ShardExchangeMixin
from vectorian_amd.flows import _rows_room, _vocab_entries, dense_flow, rwmd_sparse_flow, transport_flow
from vectorian_amd.lanes import LanesMixin
from vectorian_amd.match import HipMatch, Match, PyMatch, Region, TokenMatch, TokenMatchEdge, TokenMatchT, _Winners
from vectorian_amd.options import _QUERY_OPTION_WHITELIST, _split_gap, backend_args
from vectorian_amd.query import PreparedQuery, Query, default_tokenizer
from vectorian_amd.sim import CosineSim, EmbeddingTokenSim, OptimizedSpanSim

PartitionData = collections.namedtuple("PartitionData", ["level", "window_size", "window_step"])
_NO_LOCK = contextlib.nullcontext()   # backends without a per-handle lock (the test double)


class Index:
	def __init__(self, partition, sim):
		self._partition = partition
		self._sim = sim
		if not partition.contiguous:
			logging.warning("the used partition is non-contiguous, you will miss parts of the content.")

	@property
	def partition(self):
		return self._partition

	@property
	def session(self):
		return self._partition.session

	@property
	def sim(self):
		return self._sim

	def make_query(self, text, n=10, min_score=0.0, debug=None, options: dict = dict()):
		# vectorian/index.py:461-477
		options = options.copy()
		options["max_matches"] = n
		options["min_score"] = min_score
		if debug is not None:
			options["debug"] = debug
		options["partition"] = self._partition.to_args()
		if self._sim is not None:
			sim_args = self._sim.to_args(self)
			if sim_args:
				options["metric"] = sim_args
		return Query(self, self._partition.session.vocab, text, options)

	def find(self, text, n=10, min_score=0.0, debug=None, disable_progress=False,
			run_task=None, make_result=None, options: dict = dict()):
		# vectorian/index.py:479-501
		start_time = time.time()
		query = self.make_query(text, n=n, min_score=min_score, debug=debug, options=options)
		session = self._partition.session
		if make_result is None:
			make_result = session.make_result
		if run_task is None:
			run_task = lambda task: session.on_progress(task, disable_progress=disable_progress)
		matches = run_task(lambda progress: self._find(query, progress=progress))
		return make_result(self, matches, duration=time.time() - start_time)

	def _find(self, query, progress=None):
		raise NotImplementedError()



class HipBruteForceIndex(LanesMixin, ShardExchangeMixin, DebugHookMixin, Index):
	"""Brute-force search over every slice of the session, on one MI355X.

	Constructor signature of BruteForceIndex (vectorian/index.py:509-524):
	(partition, sim, *, nlp, saliency=None); `device` selects the GPU.  The whole corpus is
	uploaded once (token tiles in HBM); every `find` is one vk_query."""

	def __init__(self, partition, sim, *, nlp=None, saliency=None, device=0, corpus_factory=None, shard=None, group=None, precision="bf16"):
		"""shard = (rank, world): this process keeps only its contiguous range of the slices in HBM (one process per
		GPU, every process holds the same Session); `find` then merges the ranks' result sets with one all-gather
		of k records over `group` (torch.distributed; backend "nccl" = RCCL over xGMI) and returns the same matches
		on every rank (the reference merges per-document ResultSets, result_set.h:70-93).
		precision = "bf16": unit vectors rounded to bf16 in HBM (the fast path); "f32": the reference's own
		precision (CosineSim is an fp32 sgemm, vectorian/sim/vector.py:66-78), twice the bytes per token."""
		super().__init__(partition, sim)
		self._nlp = nlp
		self._shard, self._group = shard, group
		if not isinstance(sim, OptimizedSpanSim):
			raise TypeError(f"{type(sim).__name__}: the HIP index implements OptimizedSpanSim")
		token_sim = sim.token_sim
		if not isinstance(token_sim, EmbeddingTokenSim):
			raise NotImplementedError("token similarity modifiers are outside the HIP path (SURVEY 2.1)")
		if not isinstance(token_sim.similarity, CosineSim):
			raise NotImplementedError(f"{token_sim.similarity.name}: the HIP path computes CosineSim")
		self._embedding = token_sim.embedding
		self._metric_name = token_sim.to_args(self)["name"]
		session = self.session
		level, size = partition.level, partition.window_size
		# slices: Spans::iterate (vectorian/core/cpp/document.h:147-169) over every document
		starts, ends = [], []
		self._slice_doc, self._slice_id, self._slice_token_at = [], [], []
		base = 0
		for di, doc in enumerate(session.documents):
			st, en = doc.spans[level]["start"], doc.spans[level]["end"]
			n = len(st)
			if n and (st[0] != 0 or (n > 1 and (st[1:] != en[:-1]).any()) or en[-1] != doc.n_tokens):
				raise ValueError("spans must be contiguous and cover the document (document.h:151-168)")
			for sid in range(0, n, partition.window_step):
				j = min(sid + size - 1, n - 1)
				starts.append(base + int(st[sid]))
				ends.append(base + int(en[j]))
				self._slice_doc.append(di)
				self._slice_id.append(sid)
				self._slice_token_at.append(int(st[sid]))
			base += doc.n_tokens
		self._slice_start = np.array(starts, dtype=np.int64)
		self._slice_end = np.array(ends, dtype=np.int64)
		# contiguous partition (window_step == window_size): CSR offsets; otherwise general slices
		contiguous = len(starts) > 0 and starts[0] == 0 and ends[-1] == base and (self._slice_start[1:] == self._slice_end[:-1]).all()
		self._sent_off = np.concatenate(([0], self._slice_end)).astype(np.int64) if contiguous else None
		self._slice_doc = np.array(self._slice_doc, dtype=np.int64)
		n_tokens, n_slices = base, len(self._slice_doc)

		self._boost = None
		if saliency is not None:
			# Booster (vectorian/saliency.py:141-154 -> core.Booster): one multiplicative weight per slice
			self._boost = np.ascontiguousarray(saliency, dtype=np.float32)
			if self._boost.shape != (n_slices,):
				raise ValueError("saliency must hold one float per slice")

		# this process's part of the corpus: slices [sa, sb), tokens [t0, t1)
		self._slice_off, t0, t1 = 0, 0, n_tokens
		dev_start, dev_end, dev_off, dev_boost = self._slice_start, self._slice_end, self._sent_off, self._boost
		if shard is not None:
			from vectorian_amd import shards
			rank, world = shard
			sa, sb = shards.shard_ranges_by_tokens(self._slice_end - self._slice_start, world)[rank]   # equal token counts per rank
			self._slice_off = sa
			t0 = int(self._slice_start[sa]) if sb > sa else 0
			t1 = int(self._slice_end[sb - 1]) if sb > sa else 0
			dev_start, dev_end = self._slice_start[sa:sb] - t0, self._slice_end[sa:sb] - t0
			dev_off = np.concatenate(([0], dev_end)).astype(np.int64) if self._sent_off is not None else None
			dev_boost = None if self._boost is None else np.ascontiguousarray(self._boost[sa:sb])
			n_slices_dev, n_tokens_dev = sb - sa, t1 - t0
		else:
			n_slices_dev, n_tokens_dev = n_slices, n_tokens
		self._dev_boost = dev_boost
		self._n_local, self._n_local_tokens, self._n_tokens = n_slices_dev, n_tokens_dev, n_tokens
		self._xdev = None   # where exchanged records live: the process group's own device (RCCL: this GPU; gloo: the host)

		make = corpus_factory or core.Corpus
		emb = self._embedding
		self._token_ids = None   # static layout: token id per corpus token (joint vocabularies of the transport flows)
		if emb.is_static:
			vocab_vectors = emb.encode_tokens(session.vocab.tokens)
			# magnitudes are kept for WordRotatorsDistance (metric/static.cpp:69-73, 80-120); rows are normalised on upload
			self._corpus = make(layout=core.VK_LAYOUT_STATIC, d=emb.dimension, n_tokens=n_tokens_dev, n_sentences=n_slices_dev,
				vocab_size=max(1, session.vocab.size), keep_magnitudes=True, device=device, precision=precision)
			E = vocab_vectors.unmodified if session.vocab.size else np.zeros((1, emb.dimension), np.float32)
			self._corpus.append_vectors(E, normalize=True)
			ids = np.concatenate([session.doc_token_ids(i) for i in range(len(session.documents))]) if n_tokens else np.zeros(0, np.int32)
			self._corpus.set_token_ids(np.ascontiguousarray(ids[t0:t1]))
			self._token_ids = np.asarray(ids, dtype=np.int32)
		elif emb.is_contextual:
			from vectorian_amd.embedding import Vectors
			self._corpus = make(layout=core.VK_LAYOUT_CONTEXTUAL, d=emb.dimension, n_tokens=n_tokens_dev, n_sentences=n_slices_dev,
				keep_magnitudes=True, device=device, precision=precision)
			doc_base = 0
			for doc in session.documents:
				a, b = max(t0, doc_base), min(t1, doc_base + doc.n_tokens)   # this document's tokens inside the shard
				if b > a:
					self._corpus.append_vectors(Vectors(doc.contextual_vectors(emb.name)).unmodified[a - doc_base:b - doc_base], normalize=True)
				doc_base += doc.n_tokens
		else:
			raise TypeError(emb)
		self._has_pos = all(doc.pos is not None for doc in session.documents) and len(session.documents) > 0
		self._has_tags = all(doc.tags is not None for doc in session.documents) and len(session.documents) > 0
		self._pos_codes = self._tag_codes = None   # per corpus token, host copies (token filters: index maps of the winners)
		if self._has_pos:
			self._pos_codes = np.array([session.pos_code(x) for doc in session.documents for x in doc.pos], dtype=np.int8)
			if n_tokens_dev:
				self._corpus.set_token_pos(self._pos_codes[t0:t1])
		if self._has_tags:
			self._tag_codes = np.array([session.tag_code(x) for doc in session.documents for x in doc.tags], dtype=np.int8)
			if n_tokens_dev:
				self._corpus.set_token_tags(self._tag_codes[t0:t1])
		self._filtered = collections.OrderedDict()   # (pos_mask, tag_mask) -> filtered corpus; the last two filters stay resident
		self._views = []   # further handles on the resident corpus (find_many)
		if dev_off is not None:
			self._corpus.set_sentences(dev_off)
		else:
			self._corpus.set_slices(dev_start, dev_end)
		self._corpus.finalize()
		self._max_slice_len = int((np.asarray(dev_end) - np.asarray(dev_start)).max()) if len(dev_start) else 0   # of the resident part

	@property
	def metric_name(self):
		return self._metric_name

	@property
	def n_slices(self):
		return len(self._slice_doc)

	@property
	def corpus(self):
		return self._corpus

	def _backend_args(self, options):
		return backend_args(options)

	def find_many(self, texts, n=10, min_score=0.0, options: dict = dict(), in_flight=3, abort=None, batch=None, progress=None):
		"""Several queries, `in_flight` of them at a time on as many handles of the resident corpus (vk_corpus_view:
		shared arrays, own stream and workspaces) from as many host threads: the selection, traceback and host part of
		one query run beside the scoring kernel of the next (bench.py measures the path this way).  Returns one Result
		per text, in order; each equals what `find` returns.  Not part of the reference's Index (which has one
		ThreadPool task per document inside a single find, vectorian/index.py:544-558).
		batch: None (default) -- queries that can share a call do: over contextual embeddings, alignments go to the backend
		several per call (vk_query_batch: every token tile is read once per pair of queries) and relaxed-WMD queries up to 256
		per call (one MFMA-bound GEMM pass over the corpus per call, BASELINE config 4); False -- never; True -- or raise.
		A sharded index (shard = (rank, world)) takes the same paths: every rank scores its shard (batched calls included), the
		result sets of a chunk of queries travel in ONE all-gather, issued in query order from the calling thread while the
		worker threads score the next queries (the exchange bench.py times), and every rank returns the merged matches.
		progress: called with done / total after every query (every chunk of a batched call) is complete, as the reference
		reports done / total tokens after every document (vectorian/index.py:541-558)."""
		session = self.session
		queries = [self.make_query(t, n=n, min_score=min_score, options=options) for t in texts]
		if abort is not None:
			# one flag for the whole call (an int32 array of one element, as Query._abort): raising it from another thread drops
			# the queries that have not started yet -- they return no matches (Query::abort polled between the queries)
			if not (isinstance(abort, np.ndarray) and abort.dtype == np.int32 and abort.size >= 1):
				raise TypeError("abort must be an int32 numpy array")
			for q in queries:
				q._abort = abort
		batches = self._batch_plan(queries, options) if batch is not False else None
		if batch is True and batches is None:
			raise RuntimeError("find_many(batch=True): these queries cannot share a call (filters, tag weights, "
				"a debug hook, exact transport, or queries that differ in their options)")
		start = time.time()
		if batches is not None:
			results = self._find_batches(queries, batches, in_flight, progress)
		else:
			results = self._find_pipelined(queries, in_flight, progress)
		duration = (time.time() - start) / max(1, len(queries))
		return [session.make_result(self, m, duration=duration) for m in results]

	def _find(self, query, progress=None, corpus=None):
		local = self._find_local(query, corpus=corpus)
		if progress and local is not None:
			progress(self._n_local_tokens / max(1, self._n_tokens))   # this process's tokens are scored (one GPU: all of them)
		top, = self._merge_ranks([local])
		matches = self._finish_find(query, local, top)
		if progress:
			progress(1.0)
		return matches

	def _find_local(self, query, corpus=None):
		"""this process's part of a search: one vk_query against the resident corpus.  Returns None for an empty query, else a
		dict with the local result set (`aborted`: Query.abort was seen -- no matches)"""
		p_query = query.prepare(self._nlp)
		if len(p_query) == 0:
			return None
		args, gaps = self._backend_args(query.options)
		tw = args.pop("tag_weighted", None)
		if tw is not None:
			# parse_tag_weights (vectorian/core/cpp/match/instantiate.cpp:10-38): weight by Penn tag, default 1
			if not self._has_pos:
				raise RuntimeError("tag-weighted similarity needs documents with pos / tags")
			args["tag_weights"] = np.array([float(tw["tag_weights"].get(t, 1.0)) for t in p_query.tags], dtype=np.float32)
			args["q_pos"] = np.array([self.session.pos_code(x) if x is not None else 0 for x in p_query.pos], dtype=np.int8)
			args["pos_mismatch_penalty"] = tw["pos_mismatch_penalty"]
			args["similarity_threshold"] = tw["similarity_threshold"]
		emb = self._embedding
		qv = emb.encode_tokens(p_query.tokens)
		masks = self._filter_masks(query.options)
		if corpus is None:
			corpus = self._filtered_corpus(masks) if masks else self._corpus
		q_tag_codes = None
		if tw is not None and self._has_tags:
			# tag-weighted transport: the vocabulary of the bags of words is keyed by (token id, tag) (TaggedTokenFactory,
			# vectorian/core/cpp/alignment/bow.h:150-176).  The device works on positions, which is the same problem as long as
			# equal (id, tag) pairs have equal similarity rows, i.e. the universal POS is a function of the fine-grained tag (as
			# in spaCy's tag map); the host states the flows of the winners over that vocabulary.
			q_tag_codes = np.array([self.session.tag_code(t) if t is not None else 0 for t in p_query.tags], dtype=np.int8)
			if getattr(corpus, "takes_q_tags", False):
				args["q_tags"] = q_tag_codes
		hook = query.options.get("debug")
		if hook is not None and not callable(hook):
			raise TypeError("debug must be callable: hook(name, data)")   # query.cpp:73-75 casts to a py::object it later calls
		call = dict(args)
		if hook is not None and len(p_query) <= core.VK_MAX_QUERY_LEN:   # (queries of more than 64 tokens: the winners' similarity rows are not returned)
			call["want_rows"] = True
		call["abort_flag"] = query._abort
		if emb.is_static:
			call["q_token_ids"] = p_query.token_ids
		aborted = False
		all_scores = None
		# A handle serves one call at a time.  With find_many the hook's walk over all slices (debug = AllSlices) runs later, on the
		# calling thread, while this lane's thread may already be inside the next query on the same handle: every native call takes
		# the handle's lock, and what the walk needs from THIS query's device state -- the score of every slice, relaxed WMD -- is read
		# here, under the same hold of the lock as the query itself.
		with getattr(corpus, "lock", _NO_LOCK):
			try:
				top = corpus.query(qv.unmodified, q_normalize=True, boost=self._dev_boost, want_flow=True, **call)
			except core.VkError as e:
				if e.status != core.VK_ERR_ABORTED:   # Query.abort: the matcher loop ends (match/matcher_impl.h:105), nothing was added
					raise
				top, aborted = self._empty_top(len(p_query), args), True
			if (hook is not None and getattr(hook, "all_slices", False) and not aborted
					and args.get("algorithm") == core.VK_ALG_RWMD and not args.get("wmd_full")):
				all_scores = np.array(corpus.last_scores(), dtype=np.float32)
		call.pop("abort_flag")
		return dict(p_query=p_query, top=top, aborted=aborted, args=args, call=call, gaps=gaps, qv=qv, masks=masks, q_tag_codes=q_tag_codes,
			hook=hook, corpus=corpus, all_scores=all_scores)

	def _finish_find(self, query, local, top):
		"""matches of a (merged) result set; the debug hook"""
		if local is None or top is None:
			return []
		args, p_query = local["args"], local["p_query"]
		qmag = np.asarray(local["qv"].magnitudes, dtype=np.float32) if args.get("algorithm") == core.VK_ALG_WRD else None   # masses of the WRD flow
		matches = self._matches_from_topk(p_query, top, local["gaps"], args, qmag, local["masks"], local["q_tag_codes"])
		hook = local["hook"]
		if hook is not None:
			if getattr(hook, "all_slices", False):
				self._call_debug_hook_all_slices(hook, local, matches)
			else:
				self._call_debug_hook(hook, p_query, top, matches, args)
		return matches

	def _filter_masks(self, options):
		"""pos_filter / tag_filter: lists of POS / tag names whose tokens are dropped from every slice for this query
		(Query::make_token_filter, vectorian/core/cpp/query.cpp:220-228; parse_filter_mask, query.h:33-55).
		Returns (pos_mask, tag_mask) or None."""
		masks = []
		for name, lookup, have in (("pos_filter", self.session.pos_id, self._has_pos), ("tag_filter", self.session.tag_id, self._has_tags)):
			m = 0
			for x in options.get(name) or ():
				i = lookup(str(x)) if have else -1
				if i < 0 or i > 63:
					raise RuntimeError(f"illegal value {x} for {name}")   # query.h:45-49
				m |= 1 << i
			masks.append(m)
		return tuple(masks) if any(masks) else None

	def _filtered_corpus(self, masks):
		c = self._filtered.pop(masks, None)
		if c is None:
			c = self._corpus.filtered(*masks)
			while len(self._filtered) >= 2:
				self._filtered.popitem(last=False)[1].close()
		self._filtered[masks] = c
		return c

	def _index_map(self, g, masks):
		"""positions of the tokens of slice g that pass the filter (the index_map of Flow::py_regions, flow.cpp:49-60)"""
		a, b = int(self._slice_start[g]), int(self._slice_end[g])
		drop = np.zeros(b - a, dtype=bool)
		for mask, codes in zip(masks, (self._pos_codes, self._tag_codes)):
			if mask and codes is not None:
				cd = codes[a:b].astype(np.int64)
				bits = np.array([(mask >> b) & 1 for b in range(64)], dtype=bool)   # codes 0..63 have a bit (query.h:13-14)
				drop |= (cd >= 0) & (cd < 64) & bits[np.clip(cd, 0, 63)]
		return np.nonzero(~drop)[0]

	def _transport_flow(self, p_query, top, i, g, args, qmag, index_map=None, q_tag_codes=None, span=None):
		"""flow of winner i of a transport query (flows.transport_flow; a callable: HipMatch.flow evaluates it when asked)"""
		return transport_flow(self, p_query, top, i, g, args, qmag, index_map, q_tag_codes, span)

	def _matches_from_topk(self, p_query, top, gaps, args=None, qmag=None, masks=None, q_tag_codes=None):
		"""the matches of a result set: one shared `_Winners` and a two-word object per winner; everything else on access"""
		if top.n == 0:
			return []
		w = _Winners(self, p_query, top, gaps, args, qmag, masks, q_tag_codes)
		return [HipMatch(w, i) for i in range(w.n)]

	def close(self):
		"""frees the resident corpus: the further handles (find_many) and the filtered corpora first, then the corpus itself.  Call it
		(or use the index as a context manager) when the index is done: the garbage collector does not free GPU resources
		(core.Corpus.__del__ parks the handle and warns)."""
		for c in list(self._filtered.values()) + self._views:
			c.close()
		self._filtered.clear()
		self._views = []
		self._corpus.close()

	def __enter__(self):
		return self

	def __exit__(self, *exc):
		self.close()



class HipSpanEncoderIndex(Index):
	"""Brute-force cosine search over ONE embedding per span: what SpanEncoderIndex / FaissCosineIndex('Flat')
	do (vectorian/index.py:679-810), as a matrix-vector product + bounded result set on the GPU.  Realised on
	the same kernels: a corpus of one-token slices, a one-token query, score = clip(cosine)."""

	def __init__(self, partition, embedding, span_sim, nlp=None, vectors=None, device=0, corpus_factory=None):
		super().__init__(partition, span_sim)
		self._embedding = embedding
		self._nlp = nlp
		session = self.session
		step, level = partition.window_step, partition.level
		self._slice_doc, self._slice_id = [], []
		for di, doc in enumerate(session.documents):
			for sid in range(0, doc.n_spans(level), step):
				self._slice_doc.append(di)
				self._slice_id.append(sid)
		n = len(self._slice_doc)
		if vectors is None:
			size = partition.window_size
			texts = [" ".join(session.documents[di].span_tokens(level, sid, size)) for di, sid in zip(self._slice_doc, self._slice_id)]
			vectors = embedding.encode(texts)
		vectors = np.ascontiguousarray(vectors, dtype=np.float32)
		if vectors.shape != (n, embedding.dimension):
			raise ValueError(f"expected {(n, embedding.dimension)} span vectors, got {vectors.shape}")
		make = corpus_factory or core.Corpus
		self._corpus = make(layout=core.VK_LAYOUT_CONTEXTUAL, d=embedding.dimension, n_tokens=n, n_sentences=n, device=device)
		self._corpus.append_vectors(vectors, normalize=True)
		self._corpus.set_sentences(np.arange(n + 1, dtype=np.int64))
		self._corpus.finalize()

	def _find(self, query, progress=None):
		qv = self._embedding.encode([query.text])
		if qv.shape[0] != 1:
			raise RuntimeError("query produced more than one embedding")
		top = self._corpus.query(qv, q_normalize=True, locality=core.Locality.LOCAL, gap_s=0.0, gap_t=0.0,
			max_matches=int(query.options.get("max_matches", 100)), min_score=float(query.options.get("min_score", 0.0)),
			want_flow=False)
		if progress:
			progress(1.0)
		matches = []
		for i in range(top.n):
			g = int(top.sentence[i])
			doc = self.session.documents[self._slice_doc[g]]
			text = " ".join(doc.span_tokens(self.partition.level, self._slice_id[g], self.partition.window_size))
			matches.append(PyMatch(self, query, doc, self._slice_id[g], float(top.score[i]), self._sim.name,
				regions=[Region(s=text, match=None, gap_penalty=0)], level="span"))
		return matches

	def close(self):
		self._corpus.close()
