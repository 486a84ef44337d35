"""ctypes binding of libvectorian_hip.so (C-ABI: include/vectorian_hip.h).

Takes the place of `vectorian.core` (vectorian/core/__init__.py:20-31, the pybind11
module `vectorian_core`) for the brute-force alignment search path.  There is no
CPU fallback: if the shared library is missing or no gfx950 device is present the
calls raise (RuntimeError), they never compute elsewhere.
"""

import atexit
import collections
import ctypes as C
import enum
import os
import threading
import warnings

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VECTORIAN_HIP_LIB", os.path.join(_HERE, "lib", "libvectorian_hip.so"))

VK_MAX_QUERY_LEN = 64
VK_MAX_LONG_QUERY_LEN = 512   # alignments over slices of at most 64 tokens (vk_longq_kernel)
VK_MAX_SENT_LEN = 512
VK_MAX_DOC_LEN = 32767
VK_FAST_SENT_LEN = 64
VK_PREC_BF16, VK_PREC_F32 = 0, 1
VK_FAST_QUERY_LEN = 16
VK_MAX_MATCHES = 1024
VK_MAX_MATCHES_SORTED = 1048576   # alignments: result sets beyond VK_MAX_MATCHES (every score sorted on the device)

VK_F32, VK_BF16 = 0, 1
VK_MEM_HOST, VK_MEM_DEVICE = 0, 1
VK_LAYOUT_CONTEXTUAL, VK_LAYOUT_STATIC = 0, 1
VK_ALG_ALIGN, VK_ALG_RWMD, VK_ALG_WRD = 0, 1, 2
VK_GAP_LINEAR, VK_GAP_AFFINE, VK_GAP_TABLE = 0, 1, 2
VK_ERR_INVALID, VK_ERR_UNSUPPORTED, VK_ERR_HIP, VK_ERR_NO_DEVICE, VK_ERR_STATE = 1, 2, 3, 4, 5
VK_ERR_ABORTED = 6


class Locality(enum.IntEnum):
	"""core.pyalign.Locality (vectorian/core/cpp/module.cpp:149-151, used at
	vectorian/alignment.py:97,130,187)."""
	LOCAL = 0
	GLOBAL = 1
	SEMIGLOBAL = 2


class pyalign:
	"""namespace stand-in so that `core.pyalign.Locality.GLOBAL` reads as in the reference"""
	Locality = Locality


class _CorpusDesc(C.Structure):
	_fields_ = [
		("layout", C.c_int32), ("d", C.c_int32),
		("n_tokens", C.c_int64), ("n_sentences", C.c_int64),
		("vocab_size", C.c_int32), ("keep_magnitudes", C.c_int32), ("precision", C.c_int32)]


class _Gap(C.Structure):
	_fields_ = [
		("kind", C.c_int32), ("u", C.c_float), ("v", C.c_float),
		("table", C.POINTER(C.c_float)), ("n_table", C.c_int32)]


class _QueryDesc(C.Structure):
	_fields_ = [
		("algorithm", C.c_int32), ("len_t", C.c_int32),
		("q_vectors", C.c_void_p), ("q_dtype", C.c_int32), ("q_normalize", C.c_int32),
		("q_token_ids", C.c_void_p),
		("locality", C.c_int32),
		("gap_s", _Gap), ("gap_t", _Gap),
		("submatch_weight", C.c_float), ("bidirectional", C.c_int32),
		("max_matches", C.c_int32), ("min_score", C.c_float),
		("boost", C.c_void_p), ("want_flow", C.c_int32),
		("rwmd_injective", C.c_int32), ("rwmd_symmetric", C.c_int32), ("rwmd_normalize_bow", C.c_int32),
		("wrd_normalize_magnitudes", C.c_int32),
		("tag_weights", C.c_void_p), ("q_pos", C.c_void_p), ("q_tags", C.c_void_p),
		("pos_mismatch_penalty", C.c_float), ("similarity_threshold", C.c_float), ("wmd_full", C.c_int32),
		("abort", C.c_void_p), ("only_slices", C.c_void_p), ("n_only", C.c_int32)]


class _TopkOut(C.Structure):
	_fields_ = [
		("capacity", C.c_int32), ("n_out", C.c_int32),
		("score", C.c_void_p), ("raw_score", C.c_void_p), ("sentence", C.c_void_p),
		("mapping", C.c_void_p), ("edge_sim", C.c_void_p),
		("sim_rows", C.c_void_p), ("plan", C.c_void_p), ("rows_per_winner", C.c_int32)]


class _Timings(C.Structure):
	_fields_ = [
		("prepare_ms", C.c_float), ("score_ms", C.c_float), ("topk_ms", C.c_float),
		("flow_ms", C.c_float), ("total_ms", C.c_float), ("queue_ms", C.c_float)]


EXPORTS = [
	"vk_abi_version", "vk_last_error", "vk_init", "vk_device_count", "vk_corpus_view",
	"vk_corpus_create", "vk_corpus_append_vectors", "vk_corpus_set_token_ids", "vk_corpus_set_token_pos", "vk_corpus_set_token_tags", "vk_corpus_filter",
	"vk_corpus_set_sentences", "vk_corpus_set_slices", "vk_corpus_finalize", "vk_corpus_free", "vk_corpus_device_bytes",
	"vk_query", "vk_query_batch", "vk_last_scores", "vk_last_timings", "vk_merge_topk",
	"vk_record_words", "vk_pack_records", "vk_merge_records", "vk_rwmd_from_rows"]

_lib = None


def _prefer_torch_hip_runtime():
	"""PyTorch-ROCm wheels bundle their own HIP runtime and a process can initialise only one.  If torch is installed
	but not imported yet, its libamdhip64 is loaded first, so that a later `import torch` (shards.py over RCCL, bench.py)
	still finds the GPU; this library then binds to the same runtime.  Without torch nothing happens."""
	import importlib.util
	import sys
	if "torch" in sys.modules or os.environ.get("VECTORIAN_HIP_NO_TORCH_PRELOAD"):
		return
	try:
		spec = importlib.util.find_spec("torch")
	except (ImportError, ValueError):
		return
	for d in (spec.submodule_search_locations or []) if spec else []:
		path = os.path.join(d, "lib", "libamdhip64.so")
		if os.path.exists(path):
			try:
				C.CDLL(path, mode=C.RTLD_GLOBAL)
			except OSError:
				pass
			return


def lib():
	"""Loads the shared library (no GPU needed to load it)."""
	global _lib
	if _lib is None:
		if not os.path.exists(LIB_PATH):
			raise RuntimeError(
				f"{LIB_PATH} is missing: build it with `make -C vectorian_amd/csrc` "
				"(or __graft_entry__.build()); there is no CPU fallback")
		_prefer_torch_hip_runtime()
		L = C.CDLL(LIB_PATH)
		L.vk_last_error.restype = C.c_char_p
		L.vk_corpus_create.argtypes = [C.POINTER(_CorpusDesc), C.POINTER(C.c_void_p)]
		L.vk_corpus_append_vectors.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32]
		L.vk_corpus_set_token_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
		L.vk_corpus_set_token_pos.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
		L.vk_corpus_set_token_tags.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
		L.vk_corpus_set_sentences.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
		L.vk_corpus_set_slices.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
		L.vk_corpus_finalize.argtypes = [C.c_void_p]
		L.vk_corpus_free.argtypes = [C.c_void_p]
		L.vk_corpus_view.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
		L.vk_corpus_filter.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p)]
		L.vk_corpus_device_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
		L.vk_query.argtypes = [C.c_void_p, C.POINTER(_QueryDesc), C.POINTER(_TopkOut)]
		L.vk_query_batch.argtypes = [C.c_void_p, C.POINTER(_QueryDesc), C.c_int32, C.POINTER(_TopkOut)]
		L.vk_last_scores.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
		L.vk_last_timings.argtypes = [C.c_void_p, C.POINTER(_Timings)]
		L.vk_merge_topk.argtypes = [C.POINTER(_TopkOut), C.c_int32, C.c_int32, C.c_int32, C.POINTER(_TopkOut)]
		L.vk_record_words.argtypes = [C.c_int32]
		L.vk_pack_records.argtypes = [C.POINTER(_TopkOut), C.c_int32, C.c_int32, C.c_int64, C.c_void_p]
		L.vk_merge_records.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_TopkOut)]
		L.vk_rwmd_from_rows.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float)]
		if L.vk_abi_version() != 12:
			raise RuntimeError("libvectorian_hip.so ABI version mismatch")
		_lib = L
	return _lib


class VkError(RuntimeError):
	"""C-ABI status != 0; the reference raises RuntimeError from C++ exceptions."""

	def __init__(self, status, message):
		super().__init__(f"vectorian_hip error {status}: {message}")
		self.status = status


def _check(status):
	if status != 0:
		raise VkError(status, lib().vk_last_error().decode("utf-8", "replace"))


def device_count():
	n = C.c_int(0)
	status = lib().vk_device_count(C.byref(n))
	return n.value if status == 0 else 0


def init(device=0):
	_check(lib().vk_init(int(device)))


def _np_ptr(a):
	return a.ctypes.data_as(C.c_void_p)


def gap_to_struct(gap, keep, n_table):
	"""gap: object with to_special_case()/costs(n) (vectorian_amd.alignment.GapCost),
	or a tuple ('linear', u) / ('affine', u, v) / ('table', array), or a float."""
	g = _Gap()
	if hasattr(gap, "to_special_case"):
		special = gap.to_special_case()
		if "linear" in special:
			gap = ("linear", special["linear"])
		elif "affine" in special:
			gap = ("affine",) + tuple(special["affine"])
		else:
			gap = ("table", gap.costs(n_table))
	if isinstance(gap, (int, float)):
		gap = ("linear", float(gap))
	if gap[0] == "linear":
		g.kind, g.u, g.v = VK_GAP_LINEAR, float(gap[1]), 0.0
	elif gap[0] == "affine":
		g.kind, g.u, g.v = VK_GAP_AFFINE, float(gap[1]), float(gap[2])
	elif gap[0] == "table":
		t = np.ascontiguousarray(gap[1], dtype=np.float32)
		keep.append(t)
		g.kind = VK_GAP_TABLE
		g.table = t.ctypes.data_as(C.POINTER(C.c_float))
		g.n_table = len(t)
	else:
		raise ValueError(gap)
	return g


def winner_rows(longest):
	"""rows of a winner's similarity matrix (vk_topk_out.rows_per_winner): the corpus's longest slice, a multiple of 64"""
	return min(VK_MAX_DOC_LEN + 1, max(VK_FAST_SENT_LEN, (int(longest) + 63) // 64 * 64))


class TopK:
	"""Bounded result set as plain arrays (ResultSet, vectorian/core/cpp/result_set.h:17-153)."""

	def __init__(self, k, len_t, transport=False, rows=VK_FAST_SENT_LEN):
		self.k, self.len_t = k, len_t
		self.score = np.zeros(k, dtype=np.float32)
		self.raw_score = np.zeros(k, dtype=np.float32)
		self.sentence = np.zeros(k, dtype=np.int64)
		self.mapping = np.full((k, len_t), -1, dtype=np.int16)
		self.edge_sim = np.zeros((k, len_t), dtype=np.float32)
		# transport algorithms: similarity rows S[i][j] and (exact transport) the plan G[j][i] of each winner
		w = (len_t + 15) // 16 * 16     # columns of a similarity row: the query length padded to a multiple of 16
		# rows: slice tokens per winner the two arrays hold (vk_topk_out.rows_per_winner; winners longer than that: no flow stated)
		self.sim_rows = np.zeros((k, rows, w), dtype=np.float32) if transport else None
		self.plan = np.zeros((k, w, rows), dtype=np.float32) if transport else None
		self.n = 0

	@classmethod
	def over(cls, k, len_t, score, raw_score, sentence, mapping, edge_sim):
		"""a result set over rows of arrays allocated for a whole batch"""
		t = cls.__new__(cls)
		t.k, t.len_t, t.n = k, len_t, 0
		t.score, t.raw_score, t.sentence, t.mapping, t.edge_sim = score, raw_score, sentence, mapping, edge_sim
		t.sim_rows = t.plan = None
		return t

	def _struct(self):
		s = _TopkOut()
		s.capacity, s.n_out = self.k, self.n
		s.score, s.raw_score, s.sentence = _np_ptr(self.score), _np_ptr(self.raw_score), _np_ptr(self.sentence)
		s.mapping, s.edge_sim = _np_ptr(self.mapping), _np_ptr(self.edge_sim)
		if self.sim_rows is not None:
			s.sim_rows, s.plan = _np_ptr(self.sim_rows), _np_ptr(self.plan)
			s.rows_per_winner = self.sim_rows.shape[1]
		return s

	def trimmed(self):
		n = self.n
		return dict(
			score=self.score[:n].copy(), raw_score=self.raw_score[:n].copy(), sentence=self.sentence[:n].copy(),
			mapping=self.mapping[:n].copy(), edge_sim=self.edge_sim[:n].copy())


def merge_topk(sets, len_t, max_matches):
	"""ResultSet.extend over several result sets (vk_merge_topk)."""
	arr = (_TopkOut * len(sets))(*[s._struct() for s in sets])
	out = TopK(max_matches, len_t)
	so = out._struct()
	_check(lib().vk_merge_topk(arr, len(sets), len_t, max_matches, C.byref(so)))
	out.n = so.n_out
	return out


def record_words(len_t):
	return int(lib().vk_record_words(len_t))


def pack_records(top, sentence_offset, k, out=None):
	"""the result set as k exchange records (vk_pack_records): int32 [k x record_words(len_t)]"""
	if out is None:
		out = np.empty((k, record_words(top.len_t)), dtype=np.int32)
	s = top._struct()
	_check(lib().vk_pack_records(C.byref(s), top.len_t, k, int(sentence_offset), _np_ptr(out)))
	return out


def rwmd_from_rows(S, key_s=None, key_t=None, injective=True, symmetric=True, normalize_bow=True):
	"""the relaxed word mover's distance of one slice from its similarity rows S [len_s x len_t], as vk_query states it for a winner
	(vk_rwmd_from_rows; host only)"""
	S = np.ascontiguousarray(S, dtype=np.float32)
	ks = None if key_s is None else np.ascontiguousarray(key_s, dtype=np.int32)
	kt = None if key_t is None else np.ascontiguousarray(key_t, dtype=np.int32)
	out = C.c_float()
	_check(lib().vk_rwmd_from_rows(_np_ptr(S), S.shape[1], S.shape[0], S.shape[1], None if ks is None else _np_ptr(ks), None if kt is None else _np_ptr(kt),
		int(bool(injective)), int(bool(symmetric)), int(bool(normalize_bow)), C.byref(out)))
	return np.float32(out.value)


def merge_records(records, n_sets, len_t, k):
	"""ResultSet.extend over the records of n_sets result sets (vk_merge_records); records: contiguous int32
	[n_sets x k x words]"""
	assert records.dtype == np.int32 and records.flags.c_contiguous and records.size == n_sets * k * record_words(len_t)
	out = TopK(k, len_t)
	so = out._struct()
	_check(lib().vk_merge_records(_np_ptr(records), n_sets, len_t, k, C.byref(so)))
	out.n = so.n_out
	return out


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
		if _lib is not None:
			_lib.vk_corpus_free(C.c_void_p(h))
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
			init(device)
		reap()
		self.lock = threading.RLock()
		self._h = C.c_void_p()
		desc = _CorpusDesc(layout, d, n_tokens, n_sentences, vocab_size, int(keep_magnitudes), {"bf16": VK_PREC_BF16, "f32": VK_PREC_F32}[precision])
		_check(lib().vk_corpus_create(C.byref(desc), C.byref(self._h)))
		self.layout, self.d = layout, d
		self.n_tokens, self.n_sentences, self.vocab_size = n_tokens, n_sentences, vocab_size

	def append_vectors(self, rows, normalize=True):
		"""rows: numpy float32 / uint16(bf16) [n x d] on the host."""
		rows = np.ascontiguousarray(rows)
		if rows.dtype == np.float32:
			dt = VK_F32
		elif rows.dtype == np.uint16:
			dt = VK_BF16
		else:
			raise TypeError(f"vectors must be float32 or uint16 (bf16 bits), got {rows.dtype}")
		if rows.ndim != 2 or rows.shape[1] != self.d:
			raise ValueError(f"expected [n x {self.d}] vectors, got {rows.shape}")
		_check(lib().vk_corpus_append_vectors(self._h, _np_ptr(rows), rows.shape[0], dt, VK_MEM_HOST, int(normalize)))

	def append_vectors_device(self, ptr, n_rows, dtype, normalize=True):
		"""ptr: device pointer (e.g. torch.Tensor.data_ptr()) to [n_rows x d] row-major rows."""
		_check(lib().vk_corpus_append_vectors(self._h, C.c_void_p(ptr), n_rows, dtype, VK_MEM_DEVICE, int(normalize)))

	def set_token_ids(self, ids):
		ids = np.ascontiguousarray(ids, dtype=np.int32)
		_check(lib().vk_corpus_set_token_ids(self._h, _np_ptr(ids), len(ids), VK_MEM_HOST))

	def set_token_pos(self, pos):
		"""universal POS code per token occurrence (int8), for tag-weighted queries"""
		pos = np.ascontiguousarray(pos, dtype=np.int8)
		_check(lib().vk_corpus_set_token_pos(self._h, _np_ptr(pos), len(pos), VK_MEM_HOST))

	def set_token_tags(self, tags):
		"""fine-grained tag code per token occurrence (int8), for tag filters"""
		tags = np.ascontiguousarray(tags, dtype=np.int8)
		_check(lib().vk_corpus_set_token_tags(self._h, _np_ptr(tags), len(tags), VK_MEM_HOST))

	def set_sentences(self, sent_off):
		sent_off = np.ascontiguousarray(sent_off, dtype=np.int64)
		_check(lib().vk_corpus_set_sentences(self._h, _np_ptr(sent_off), len(sent_off) - 1))
		self._max_len = int(np.diff(sent_off).max()) if len(sent_off) > 1 else 0

	def set_slices(self, start, end):
		"""general (possibly overlapping) slices: tokens [start[i], end[i])"""
		start = np.ascontiguousarray(start, dtype=np.int64)
		end = np.ascontiguousarray(end, dtype=np.int64)
		_check(lib().vk_corpus_set_slices(self._h, _np_ptr(start), _np_ptr(end), len(start)))
		self._max_len = int((end - start).max()) if len(start) else 0

	def finalize(self):
		_check(lib().vk_corpus_finalize(self._h))

	@property
	def device_bytes(self):
		b = C.c_int64(0)
		_check(lib().vk_corpus_device_bytes(self._h, C.byref(b)))
		return b.value

	def _desc(self, q_vectors, keep, *, locality=Locality.LOCAL, gap_s=0.0, gap_t=0.0, algorithm=VK_ALG_ALIGN,
			q_token_ids=None, q_normalize=True, max_matches=10, min_score=0.0, boost=None, want_flow=True,
			submatch_weight=0.0, bidirectional=False, rwmd=(True, True, True), wrd_normalize=True,
			tag_weights=None, q_pos=None, q_tags=None, pos_mismatch_penalty=0.0, similarity_threshold=0.0, wmd_full=False,
			abort_flag=None, want_rows=False, only_slices=None):
		"""abort_flag: int32 array of one element another thread may set to 1 (Query.abort); want_rows: similarity rows of the
		winners of an alignment query too (the debug hook's 'similarity'); only_slices: state exactly these slices (in order)
		instead of searching -- the debug hook's walk over every slice (vk_query_desc.only_slices)"""
		q_vectors = np.ascontiguousarray(q_vectors)
		if q_vectors.dtype == np.uint16:
			qdt = VK_BF16
		else:
			q_vectors = np.ascontiguousarray(q_vectors, dtype=np.float32)
			qdt = VK_F32
		if q_vectors.ndim != 2 or q_vectors.shape[1] != self.d:
			raise ValueError(f"expected [len_t x {self.d}] query vectors, got {q_vectors.shape}")
		keep.append(q_vectors)
		len_t = q_vectors.shape[0]
		q = _QueryDesc()
		q.algorithm, q.len_t = int(algorithm), len_t
		q.q_vectors, q.q_dtype, q.q_normalize = _np_ptr(q_vectors), qdt, int(q_normalize)
		if q_token_ids is not None:
			ids = np.ascontiguousarray(q_token_ids, dtype=np.int32)
			keep.append(ids)
			q.q_token_ids = _np_ptr(ids)
		q.locality = int(locality)
		n_table = max(VK_MAX_SENT_LEN, getattr(self, "_max_len", 0)) + 1   # a gap table covers the corpus's longest slice
		q.gap_s = gap_to_struct(gap_s, keep, n_table)
		q.gap_t = gap_to_struct(gap_t, keep, VK_MAX_SENT_LEN + 1)
		q.submatch_weight, q.bidirectional = float(submatch_weight), int(bool(bidirectional))
		q.max_matches, q.min_score = int(max_matches), float(min_score)
		if boost is not None:
			b = boost if isinstance(boost, np.ndarray) and boost.dtype == np.float32 and boost.flags.c_contiguous \
				else np.ascontiguousarray(boost, dtype=np.float32)
			if len(b) != self.n_sentences:
				raise ValueError("boost must have one entry per sentence")
			keep.append(b)
			q.boost = _np_ptr(b)
		q.want_flow = int(bool(want_flow))
		q.rwmd_injective, q.rwmd_symmetric, q.rwmd_normalize_bow = [int(bool(x)) for x in rwmd]
		q.wrd_normalize_magnitudes = int(bool(wrd_normalize))
		q.wmd_full = int(bool(wmd_full))
		if abort_flag is not None:
			if not (isinstance(abort_flag, np.ndarray) and abort_flag.dtype == np.int32 and abort_flag.size >= 1):
				raise TypeError("abort_flag must be an int32 array")
			keep.append(abort_flag)
			q.abort = _np_ptr(abort_flag)
		if only_slices is not None:
			only = np.ascontiguousarray(only_slices, dtype=np.int64)
			keep.append(only)
			q.only_slices, q.n_only = _np_ptr(only), len(only)
			q.max_matches = max(1, len(only))
		if tag_weights is not None:
			tw = np.ascontiguousarray(tag_weights, dtype=np.float32)
			qp = np.ascontiguousarray(q_pos if q_pos is not None else np.zeros(len_t), dtype=np.int8)
			if len(tw) != len_t or len(qp) != len_t:
				raise ValueError("tag_weights / q_pos must have one entry per query token")
			keep.extend([tw, qp])
			q.tag_weights, q.q_pos = _np_ptr(tw), _np_ptr(qp)
			if q_tags is not None:
				qt = np.ascontiguousarray(q_tags, dtype=np.int8)
				if len(qt) != len_t:
					raise ValueError("q_tags must have one entry per query token")
				keep.append(qt)
				q.q_tags = _np_ptr(qt)
			q.pos_mismatch_penalty, q.similarity_threshold = float(pos_mismatch_penalty), float(similarity_threshold)
		return q, len_t

	def _winner_rows(self):
		"""similarity rows / plans of the winners: room for the longest slice of the corpus (a multiple of 64 tokens)"""
		return winner_rows(getattr(self, "_max_len", 0))

	def query(self, q_vectors, **options):
		"""One query against the shard (vk_query).  Returns a TopK."""
		keep = []
		q, len_t = self._desc(q_vectors, keep, **options)
		rows = self._winner_rows()
		out = TopK(max(1, q.max_matches), len_t, transport=bool(q.want_flow) and (q.algorithm != VK_ALG_ALIGN or bool(options.get("want_rows"))), rows=rows)
		so = out._struct()
		with self.lock:
			_check(lib().vk_query(self._h, C.byref(q), C.byref(so)))
		out.n = so.n_out
		return out

	def query_batch(self, queries, token_ids=None, **options):
		"""A batch of queries with common options (vk_query_batch).  Returns a list of TopK.
		token_ids: static layout -- the vocabulary ids of every query's tokens, a list parallel to `queries` (vk_query_desc.q_token_ids)"""
		keep = []
		n = len(queries)
		if token_ids is not None and len(token_ids) != n:
			raise ValueError("token_ids must hold one array per query")
		qs = (_QueryDesc * n)()
		sos = (_TopkOut * n)()
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
				q = _QueryDesc.from_buffer_copy(first)
				q.q_vectors, q.q_dtype, q.len_t = _np_ptr(qv), (VK_BF16 if qv.dtype == np.uint16 else VK_F32), qv.shape[0]
				len_t = qv.shape[0]
			qs[i] = q
			# transport flows need the winners' similarity rows: always for the relaxed WMD (the batch path returns them for every
			# query), for exact transport (answered query by query) only in small batches -- 80 KB of rows and plans per query
			t = TopK(max(1, q.max_matches), len_t, transport=bool(q.want_flow) and q.algorithm != VK_ALG_ALIGN and
				(n <= 16 or (q.algorithm == VK_ALG_RWMD and not q.wmd_full)), rows=self._winner_rows())
			outs.append(t)
			sos[i] = t._struct()
		with self.lock:
			_check(lib().vk_query_batch(self._h, qs, n, sos))
		for t, so in zip(outs, sos):
			t.n = so.n_out
		return outs

	def _batch_fast(self, queries, per_query, options, qs, sos, keep, token_ids=None):
		"""large batches: the queries in one array (their lengths may differ), one allocation per result field, the
		descriptors filled by pointer arithmetic (256 queries: 6 ms of per-query numpy / ctypes work otherwise)"""
		n = len(queries)
		if per_query or n < 8:
			return None
		if getattr(self, "_max_len", 0) > VK_FAST_SENT_LEN:
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
		if bool(first.want_flow) and first.algorithm != VK_ALG_ALIGN and (n <= 16 or int(lens.max()) > VK_FAST_QUERY_LEN):
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

		qrows = np.frombuffer(qs, dtype=np.uint8).reshape(n, C.sizeof(_QueryDesc))
		qrows[:] = np.frombuffer(first, dtype=np.uint8)
		patch(qrows, _QueryDesc.q_vectors, Q, offsets=row_off[:-1] * self.d)
		if token_ids is not None:
			T = np.ascontiguousarray(np.concatenate([np.asarray(t, dtype=np.int32) for t in token_ids]), dtype=np.int32)
			if len(T) != int(row_off[-1]) or any(len(t) != l for t, l in zip(token_ids, lens)):
				raise ValueError("token_ids must hold one id per query token")
			keep.append(T)
			patch(qrows, _QueryDesc.q_token_ids, T, offsets=row_off[:-1])
		qrows[:, _QueryDesc.len_t.offset:_QueryDesc.len_t.offset + 4].view(np.int32)[:, 0] = lens
		proto = _TopkOut()
		proto.capacity, proto.n_out = k, 0
		orows = np.frombuffer(sos, dtype=np.uint8).reshape(n, C.sizeof(_TopkOut))
		orows[:] = np.frombuffer(proto, dtype=np.uint8)
		for field, arr in ((_TopkOut.score, score), (_TopkOut.raw_score, raw), (_TopkOut.sentence, sentence)):
			patch(orows, field, arr)
		patch(orows, _TopkOut.mapping, mapping, offsets=row_off[:-1] * k)
		patch(orows, _TopkOut.edge_sim, edge, offsets=row_off[:-1] * k)
		rows = plan = None
		if bool(first.want_flow) and first.algorithm == VK_ALG_RWMD and not bool(first.wmd_full):
			# relaxed WMD: the similarity rows of every query's winners (the host states their SparseFlow from them); no plans
			# (exact transport only): one zero array stands in for all of them
			# (a corpus with slices of more than 64 tokens is answered query by query: room for its longest slice, as `query`)
			R = self._winner_rows()
			rows = np.zeros((n, k, R, 16), np.float32)
			plan = np.zeros((k, 16, R), np.float32)
			keep.extend((rows, plan))
			patch(orows, _TopkOut.sim_rows, rows)
			orows[:, _TopkOut.rows_per_winner.offset:_TopkOut.rows_per_winner.offset + 4].view(np.int32)[:, 0] = R
		outs = []
		for i in range(n):
			a, b, lt = k * int(row_off[i]), k * int(row_off[i + 1]), int(lens[i])
			t = TopK.over(k, lt, score[i], raw[i], sentence[i], mapping[a:b].reshape(k, lt), edge[a:b].reshape(k, lt))
			if rows is not None:
				t.sim_rows, t.plan = rows[i], plan
			outs.append(t)
		return outs

	def last_scores(self):
		s = np.empty(self.n_sentences, dtype=np.float32)
		with self.lock:
			_check(lib().vk_last_scores(self._h, _np_ptr(s), len(s)))
		return s

	def last_timings(self):
		t = _Timings()
		with self.lock:
			_check(lib().vk_last_timings(self._h, C.byref(t)))
		return {k: getattr(t, k) for k, _ in _Timings._fields_}

	def view(self):
		"""a second handle on the same resident corpus (vk_corpus_view): own stream and workspaces, shared arrays.
		Queries on different handles may run from different threads at the same time."""
		v = Corpus.__new__(Corpus)
		v.__dict__.update({k: val for k, val in self.__dict__.items() if k not in ("_h", "lock")})
		v._h = C.c_void_p()
		v.lock = threading.RLock()
		with self.lock:
			_check(lib().vk_corpus_view(self._h, C.byref(v._h)))
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
			_check(lib().vk_corpus_filter(self._h, C.c_uint64(int(pos_mask)), C.c_uint64(int(tag_mask)), C.byref(f._h)))
		return f

	def close(self):
		"""frees the handle (idempotent).  Waits for a call in progress on THIS handle; other handles of the corpus may be mid-call
		(the arrays they share are reference-counted in the library)."""
		with self.lock:
			h, self._h = self._h, C.c_void_p()
			if h:
				lib().vk_corpus_free(h)
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
