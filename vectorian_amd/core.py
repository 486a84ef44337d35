"""ctypes binding of libvectorian_hip.so (C-ABI: include/vectorian_hip.h).

Takes the place of `vectorian.core` (vectorian/core/__init__.py:20-31, the pybind11
module `vectorian_core`) for the brute-force alignment search path.  There is no
CPU fallback: if the shared library is missing or no gfx950 device is present the
calls raise (RuntimeError), they never compute elsewhere.
"""

import ctypes as C
import enum
import os

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


# the corpus handle (lifetime, query descriptors, vk_query / vk_query_batch): vectorian_amd/handle.py
from .handle import Corpus, reap, _graveyard  # noqa: E402,F401
