"""The corpus handle of the C-ABI (vk_corpus_t) as a Python object: lifetime (close / context manager / the parking of handles
whose objects the collector found), the query descriptor, vk_query / vk_query_batch, views and filtered corpora.  Split from
core.py (round 4); `core.Corpus`, `core.reap` stay the names everything else uses.  The library and its structures are looked up in
`core` at call time (tests put a fake library there)."""

import atexit
import collections
import ctypes as C
import threading
import warnings

import numpy as np

from . import core as _core

# Handles of Corpus objects that were garbage-collected without close().  The collector runs finalizers on whichever thread
# happens to allocate, possibly while other threads are inside vk_query: a finalizer therefore never frees GPU resources -- it
# parks the handle here, and the handles are freed at the next safe point of a calling thread (reap(): before a corpus is
# created, after one is closed, at exit).  The library itself tolerates any order of frees (the arrays handles share are
# reference-counted, vk_internal.h vk_devblock).
_graveyard = collections.deque()


def reap():
	"""frees the handles parked by finalizers; returns how many.  Called from the thread that creates / closes corpora."""
	n = 0
	while True:
		try:
			h = _graveyard.popleft()
		except IndexError:
			return n
		if _core._lib is not None:
			_core._lib.vk_corpus_free(C.c_void_p(h))
		n += 1


atexit.register(reap)


class Corpus:
	"""A corpus shard resident in HBM (opaque vk_corpus_t handle).

	Lifetime: close() it (Index.close() does, views and filtered corpora first); `with Corpus(...) as c:` works.  A Corpus that
	is garbage-collected unclosed warns (ResourceWarning) and its handle is parked for reap() -- the finalizer itself makes no GPU
	call.  A handle serves one call at a time: `lock` is held for the duration of every native call on it (and by close())."""

	takes_q_tags = True   # query(q_tags=...): tag codes of the query tokens (tag-weighted transport over (id, tag) vocabularies)

	def __init__(self, *, layout, d, n_tokens, n_sentences, vocab_size=0, keep_magnitudes=False, device=None, precision="bf16"):
		"""precision: "bf16" (unit rows rounded to bf16, the fast path) or "f32" (the reference's own precision, twice the bytes)"""
		if device is not None:
			_core.init(device)
		reap()
		self.lock = threading.RLock()
		self._h = C.c_void_p()
		desc = _core._CorpusDesc(layout, d, n_tokens, n_sentences, vocab_size, int(keep_magnitudes), {"bf16": _core.VK_PREC_BF16, "f32": _core.VK_PREC_F32}[precision])
		_core._check(_core.lib().vk_corpus_create(C.byref(desc), C.byref(self._h)))
		self.layout, self.d = layout, d
		self.n_tokens, self.n_sentences, self.vocab_size = n_tokens, n_sentences, vocab_size

	def append_vectors(self, rows, normalize=True):
		"""rows: numpy float32 / uint16(bf16) [n x d] on the host."""
		rows = np.ascontiguousarray(rows)
		if rows.dtype == np.float32:
			dt = _core.VK_F32
		elif rows.dtype == np.uint16:
			dt = _core.VK_BF16
		else:
			raise TypeError(f"vectors must be float32 or uint16 (bf16 bits), got {rows.dtype}")
		if rows.ndim != 2 or rows.shape[1] != self.d:
			raise ValueError(f"expected [n x {self.d}] vectors, got {rows.shape}")
		_core._check(_core.lib().vk_corpus_append_vectors(self._h, _core._np_ptr(rows), rows.shape[0], dt, _core.VK_MEM_HOST, int(normalize)))

	def append_vectors_device(self, ptr, n_rows, dtype, normalize=True):
		"""ptr: device pointer (e.g. torch.Tensor.data_ptr()) to [n_rows x d] row-major rows."""
		_core._check(_core.lib().vk_corpus_append_vectors(self._h, C.c_void_p(ptr), n_rows, dtype, _core.VK_MEM_DEVICE, int(normalize)))

	def set_token_ids(self, ids):
		ids = np.ascontiguousarray(ids, dtype=np.int32)
		_core._check(_core.lib().vk_corpus_set_token_ids(self._h, _core._np_ptr(ids), len(ids), _core.VK_MEM_HOST))

	def set_token_pos(self, pos):
		"""universal POS code per token occurrence (int8), for tag-weighted queries"""
		pos = np.ascontiguousarray(pos, dtype=np.int8)
		_core._check(_core.lib().vk_corpus_set_token_pos(self._h, _core._np_ptr(pos), len(pos), _core.VK_MEM_HOST))

	def set_token_tags(self, tags):
		"""fine-grained tag code per token occurrence (int8), for tag filters"""
		tags = np.ascontiguousarray(tags, dtype=np.int8)
		_core._check(_core.lib().vk_corpus_set_token_tags(self._h, _core._np_ptr(tags), len(tags), _core.VK_MEM_HOST))

	def set_sentences(self, sent_off):
		sent_off = np.ascontiguousarray(sent_off, dtype=np.int64)
		_core._check(_core.lib().vk_corpus_set_sentences(self._h, _core._np_ptr(sent_off), len(sent_off) - 1))
		self._max_len = int(np.diff(sent_off).max()) if len(sent_off) > 1 else 0

	def set_slices(self, start, end):
		"""general (possibly overlapping) slices: tokens [start[i], end[i])"""
		start = np.ascontiguousarray(start, dtype=np.int64)
		end = np.ascontiguousarray(end, dtype=np.int64)
		_core._check(_core.lib().vk_corpus_set_slices(self._h, _core._np_ptr(start), _core._np_ptr(end), len(start)))
		self._max_len = int((end - start).max()) if len(start) else 0

	def finalize(self):
		_core._check(_core.lib().vk_corpus_finalize(self._h))

	@property
	def device_bytes(self):
		b = C.c_int64(0)
		_core._check(_core.lib().vk_corpus_device_bytes(self._h, C.byref(b)))
		return b.value

	def _desc(self, q_vectors, keep, *, locality=_core.Locality.LOCAL, gap_s=0.0, gap_t=0.0, algorithm=_core.VK_ALG_ALIGN,
			q_token_ids=None, q_normalize=True, max_matches=10, min_score=0.0, boost=None, want_flow=True,
			submatch_weight=0.0, bidirectional=False, rwmd=(True, True, True), wrd_normalize=True,
			tag_weights=None, q_pos=None, q_tags=None, pos_mismatch_penalty=0.0, similarity_threshold=0.0, wmd_full=False,
			abort_flag=None, want_rows=False, only_slices=None):
		"""abort_flag: int32 array of one element another thread may set to 1 (Query.abort); want_rows: similarity rows of the
		winners of an alignment query too (the debug hook's 'similarity'); only_slices: state exactly these slices (in order)
		instead of searching -- the debug hook's walk over every slice (vk_query_desc.only_slices)"""
		q_vectors = np.ascontiguousarray(q_vectors)
		if q_vectors.dtype == np.uint16:
			qdt = _core.VK_BF16
		else:
			q_vectors = np.ascontiguousarray(q_vectors, dtype=np.float32)
			qdt = _core.VK_F32
		if q_vectors.ndim != 2 or q_vectors.shape[1] != self.d:
			raise ValueError(f"expected [len_t x {self.d}] query vectors, got {q_vectors.shape}")
		keep.append(q_vectors)
		len_t = q_vectors.shape[0]
		q = _core._QueryDesc()
		q.algorithm, q.len_t = int(algorithm), len_t
		q.q_vectors, q.q_dtype, q.q_normalize = _core._np_ptr(q_vectors), qdt, int(q_normalize)
		if q_token_ids is not None:
			ids = np.ascontiguousarray(q_token_ids, dtype=np.int32)
			keep.append(ids)
			q.q_token_ids = _core._np_ptr(ids)
		q.locality = int(locality)
		n_table = max(_core.VK_MAX_SENT_LEN, getattr(self, "_max_len", 0)) + 1   # a gap table covers the corpus's longest slice
		q.gap_s = _core.gap_to_struct(gap_s, keep, n_table)
		q.gap_t = _core.gap_to_struct(gap_t, keep, _core.VK_MAX_SENT_LEN + 1)
		q.submatch_weight, q.bidirectional = float(submatch_weight), int(bool(bidirectional))
		q.max_matches, q.min_score = int(max_matches), float(min_score)
		if boost is not None:
			b = boost if isinstance(boost, np.ndarray) and boost.dtype == np.float32 and boost.flags.c_contiguous \
				else np.ascontiguousarray(boost, dtype=np.float32)
			if len(b) != self.n_sentences:
				raise ValueError("boost must have one entry per sentence")
			keep.append(b)
			q.boost = _core._np_ptr(b)
		q.want_flow = int(bool(want_flow))
		q.rwmd_injective, q.rwmd_symmetric, q.rwmd_normalize_bow = [int(bool(x)) for x in rwmd]
		q.wrd_normalize_magnitudes = int(bool(wrd_normalize))
		q.wmd_full = int(bool(wmd_full))
		if abort_flag is not None:
			if not (isinstance(abort_flag, np.ndarray) and abort_flag.dtype == np.int32 and abort_flag.size >= 1):
				raise TypeError("abort_flag must be an int32 array")
			keep.append(abort_flag)
			q.abort = _core._np_ptr(abort_flag)
		if only_slices is not None:
			only = np.ascontiguousarray(only_slices, dtype=np.int64)
			keep.append(only)
			q.only_slices, q.n_only = _core._np_ptr(only), len(only)
			q.max_matches = max(1, len(only))
		if tag_weights is not None:
			tw = np.ascontiguousarray(tag_weights, dtype=np.float32)
			qp = np.ascontiguousarray(q_pos if q_pos is not None else np.zeros(len_t), dtype=np.int8)
			if len(tw) != len_t or len(qp) != len_t:
				raise ValueError("tag_weights / q_pos must have one entry per query token")
			keep.extend([tw, qp])
			q.tag_weights, q.q_pos = _core._np_ptr(tw), _core._np_ptr(qp)
			if q_tags is not None:
				qt = np.ascontiguousarray(q_tags, dtype=np.int8)
				if len(qt) != len_t:
					raise ValueError("q_tags must have one entry per query token")
				keep.append(qt)
				q.q_tags = _core._np_ptr(qt)
			q.pos_mismatch_penalty, q.similarity_threshold = float(pos_mismatch_penalty), float(similarity_threshold)
		return q, len_t

	def _winner_rows(self):
		"""similarity rows / plans of the winners: room for the longest slice of the corpus (a multiple of 64 tokens)"""
		return _core.winner_rows(getattr(self, "_max_len", 0))

	def query(self, q_vectors, **options):
		"""One query against the shard (vk_query).  Returns a _core.TopK."""
		keep = []
		q, len_t = self._desc(q_vectors, keep, **options)
		rows = self._winner_rows()
		out = _core.TopK(max(1, q.max_matches), len_t, transport=bool(q.want_flow) and (q.algorithm != _core.VK_ALG_ALIGN or bool(options.get("want_rows"))), rows=rows)
		so = out._struct()
		with self.lock:
			_core._check(_core.lib().vk_query(self._h, C.byref(q), C.byref(so)))
		out.n = so.n_out
		return out

	def query_batch(self, queries, token_ids=None, **options):
		"""A batch of queries with common options (vk_query_batch).  Returns a list of _core.TopK.
		token_ids: static layout -- the vocabulary ids of every query's tokens, a list parallel to `queries` (vk_query_desc.q_token_ids)"""
		keep = []
		n = len(queries)
		if token_ids is not None and len(token_ids) != n:
			raise ValueError("token_ids must hold one array per query")
		qs = (_core._QueryDesc * n)()
		sos = (_core._TopkOut * n)()
		outs = []
		if options.get("boost") is not None:
			options = dict(options, boost=np.ascontiguousarray(options["boost"], dtype=np.float32))
		# the options are common: build one descriptor and copy it, only the vectors differ (tag weights, token ids
		# and POS codes are per query and take the full path)
		per_query = any(options.get(k) is not None for k in ("q_token_ids", "tag_weights", "q_pos", "q_tags"))
		first = None
		fast = self._batch_fast(queries, per_query, options, qs, sos, keep, token_ids)
		if fast is not None:
			outs = fast
			queries = ()
		for i, qv in enumerate(queries):
			if first is None or per_query or token_ids is not None:
				q, len_t = self._desc(qv, keep, **(options if token_ids is None else dict(options, q_token_ids=token_ids[i])))
				first = first or q
			else:
				qv = np.ascontiguousarray(qv)
				if qv.dtype != np.uint16:
					qv = np.ascontiguousarray(qv, dtype=np.float32)
				if qv.ndim != 2 or qv.shape[1] != self.d:
					raise ValueError(f"expected [len_t x {self.d}] query vectors, got {qv.shape}")
				keep.append(qv)
				q = _core._QueryDesc.from_buffer_copy(first)
				q.q_vectors, q.q_dtype, q.len_t = _core._np_ptr(qv), (_core.VK_BF16 if qv.dtype == np.uint16 else _core.VK_F32), qv.shape[0]
				len_t = qv.shape[0]
			qs[i] = q
			# transport flows need the winners' similarity rows: always for the relaxed WMD (the batch path returns them for every
			# query), for exact transport (answered query by query) only in small batches -- 80 KB of rows and plans per query
			t = _core.TopK(max(1, q.max_matches), len_t, transport=bool(q.want_flow) and q.algorithm != _core.VK_ALG_ALIGN and
				(n <= 16 or (q.algorithm == _core.VK_ALG_RWMD and not q.wmd_full)), rows=self._winner_rows())
			outs.append(t)
			sos[i] = t._struct()
		with self.lock:
			_core._check(_core.lib().vk_query_batch(self._h, qs, n, sos))
		for t, so in zip(outs, sos):
			t.n = so.n_out
		return outs

	def _batch_fast(self, queries, per_query, options, qs, sos, keep, token_ids=None):
		"""large batches: the queries in one array (their lengths may differ), one allocation per result field, the
		descriptors filled by pointer arithmetic (256 queries: 6 ms of per-query numpy / ctypes work otherwise)"""
		n = len(queries)
		if per_query or n < 8:
			return None
		if getattr(self, "_max_len", 0) > _core.VK_FAST_SENT_LEN:
			# a corpus with slices of more than 64 tokens is answered query by query (vk_query_batch): per-query result sets sized by
			# that query -- one [n x k x R x 16] block with R = the longest slice would be gigabytes for documents (256 x 10 x 32768)
			return None
		arrs = [np.asarray(q) for q in queries]
		if any(a.ndim != 2 or a.shape[1] != self.d or a.shape[0] < 1 or a.dtype != arrs[0].dtype for a in arrs):
			return None
		if arrs[0].dtype != np.uint16:
			arrs = [np.asarray(a, dtype=np.float32) for a in arrs]
		lens = np.array([a.shape[0] for a in arrs], dtype=np.int64)
		Q = np.ascontiguousarray(np.concatenate(arrs))          # [sum of the lengths x d]
		keep.append(Q)
		first, _ = self._desc(arrs[0], keep, **options)
		if bool(first.want_flow) and first.algorithm != _core.VK_ALG_ALIGN and (n <= 16 or int(lens.max()) > _core.VK_FAST_QUERY_LEN):
			return None    # transport flows: per-query row / plan buffers (the general path; queries of more than 16 tokens: wider rows)
		k = max(1, first.max_matches)
		score, raw = np.zeros((n, k), np.float32), np.zeros((n, k), np.float32)
		sentence = np.zeros((n, k), np.int64)
		# mapping / edge_sim of query i: [k x len_t(i)], one after the other in a flat array
		row_off = np.concatenate(([0], np.cumsum(lens)))                     # in query tokens
		mapping, edge = np.full(k * int(row_off[-1]), -1, np.int16), np.zeros(k * int(row_off[-1]), np.float32)
		keep.extend((score, raw, sentence, mapping, edge))
		# the descriptor arrays as bytes: every row a copy of the first descriptor, the pointer fields patched in one go
		idx = np.arange(n, dtype=np.uint64)

		def patch(rows, field, arr, offsets=None):
			step = idx * np.uint64(arr.strides[0]) if offsets is None else offsets.astype(np.uint64) * np.uint64(arr.itemsize)
			rows[:, field.offset:field.offset + 8].view(np.uint64)[:, 0] = np.uint64(arr.ctypes.data) + step

		qrows = np.frombuffer(qs, dtype=np.uint8).reshape(n, C.sizeof(_core._QueryDesc))
		qrows[:] = np.frombuffer(first, dtype=np.uint8)
		patch(qrows, _core._QueryDesc.q_vectors, Q, offsets=row_off[:-1] * self.d)
		if token_ids is not None:
			T = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.int32) for t in token_ids]), dtype=np.int32)
			if len(T) != int(row_off[-1]) or any(len(t) != l for t, l in zip(token_ids, lens)):
				raise ValueError("token_ids must hold one id per query token")
			keep.append(T)
			patch(qrows, _core._QueryDesc.q_token_ids, T, offsets=row_off[:-1])
		qrows[:, _core._QueryDesc.len_t.offset:_core._QueryDesc.len_t.offset + 4].view(np.int32)[:, 0] = lens
		proto = _core._TopkOut()
		proto.capacity, proto.n_out = k, 0
		orows = np.frombuffer(sos, dtype=np.uint8).reshape(n, C.sizeof(_core._TopkOut))
		orows[:] = np.frombuffer(proto, dtype=np.uint8)
		for field, arr in ((_core._TopkOut.score, score), (_core._TopkOut.raw_score, raw), (_core._TopkOut.sentence, sentence)):
			patch(orows, field, arr)
		patch(orows, _core._TopkOut.mapping, mapping, offsets=row_off[:-1] * k)
		patch(orows, _core._TopkOut.edge_sim, edge, offsets=row_off[:-1] * k)
		rows = plan = None
		if bool(first.want_flow) and first.algorithm == _core.VK_ALG_RWMD and not bool(first.wmd_full):
			# relaxed WMD: the similarity rows of every query's winners (the host states their SparseFlow from them); no plans
			# (exact transport only): one zero array stands in for all of them
			# (a corpus with slices of more than 64 tokens is answered query by query: room for its longest slice, as `query`)
			R = self._winner_rows()
			rows = np.zeros((n, k, R, 16), np.float32)
			plan = np.zeros((k, 16, R), np.float32)
			keep.extend((rows, plan))
			patch(orows, _core._TopkOut.sim_rows, rows)
			orows[:, _core._TopkOut.rows_per_winner.offset:_core._TopkOut.rows_per_winner.offset + 4].view(np.int32)[:, 0] = R
		outs = []
		for i in range(n):
			a, b, lt = k * int(row_off[i]), k * int(row_off[i + 1]), int(lens[i])
			t = _core.TopK.over(k, lt, score[i], raw[i], sentence[i], mapping[a:b].reshape(k, lt), edge[a:b].reshape(k, lt))
			if rows is not None:
				t.sim_rows, t.plan = rows[i], plan
			outs.append(t)
		return outs

	def last_scores(self):
		s = np.empty(self.n_sentences, dtype=np.float32)
		with self.lock:
			_core._check(_core.lib().vk_last_scores(self._h, _core._np_ptr(s), len(s)))
		return s

	def last_timings(self):
		t = _core._Timings()
		with self.lock:
			_core._check(_core.lib().vk_last_timings(self._h, C.byref(t)))
		return {k: getattr(t, k) for k, _ in _core._Timings._fields_}

	def view(self):
		"""a second handle on the same resident corpus (vk_corpus_view): own stream and workspaces, shared arrays.
		Queries on different handles may run from different threads at the same time."""
		v = Corpus.__new__(Corpus)
		v.__dict__.update({k: val for k, val in self.__dict__.items() if k not in ("_h", "lock")})
		v._h = C.c_void_p()
		v.lock = threading.RLock()
		with self.lock:
			_core._check(_core.lib().vk_corpus_view(self._h, C.byref(v._h)))
		return v

	def filtered(self, pos_mask=0, tag_mask=0):
		"""the corpus without the tokens whose POS / tag code has its bit set in the masks (vk_corpus_filter;
		TokenFilter, vectorian/core/cpp/query.h:8-28): same slices, re-indexed; built once on the device"""
		f = Corpus.__new__(Corpus)
		f.__dict__.update({k: val for k, val in self.__dict__.items() if k not in ("_h", "lock")})
		f._h = C.c_void_p()
		f.lock = threading.RLock()
		reap()
		with self.lock:
			_core._check(_core.lib().vk_corpus_filter(self._h, C.c_uint64(int(pos_mask)), C.c_uint64(int(tag_mask)), C.byref(f._h)))
		return f

	def close(self):
		"""frees the handle (idempotent).  Waits for a call in progress on THIS handle; other handles of the corpus may be mid-call
		(the arrays they share are reference-counted in the library)."""
		with self.lock:
			h, self._h = self._h, C.c_void_p()
			if h:
				_core.lib().vk_corpus_free(h)
		reap()

	def __enter__(self):
		return self

	def __exit__(self, *exc):
		self.close()

	def __del__(self):
		# never a GPU call from a finalizer (module docstring of _graveyard): park the handle, warn
		h = self.__dict__.get("_h")
		if h:
			_graveyard.append(h.value)
			self.__dict__["_h"] = C.c_void_p()
			try:
				warnings.warn("vectorian_amd Corpus was garbage-collected without close(); its handle is queued for core.reap()", ResourceWarning, source=self)
			except Exception:
				pass   # interpreter shutdown
