"""ctypes front-end of the CPU oracle (oracle/vk_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package vectorian_amd.
"""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvk_oracle.so")

LOCAL, GLOBAL, SEMIGLOBAL = 0, 1, 2
GAP_LINEAR, GAP_AFFINE, GAP_TABLE = 0, 1, 2
LAYOUT_CONTEXTUAL, LAYOUT_STATIC = 0, 1
ALG_ALIGN, ALG_RWMD, ALG_WRD = 0, 1, 2


def build(force=False):
	src = [os.path.join(_HERE, f) for f in ("vk_oracle.c", "vk_oracle.h")]
	if (not force and os.path.exists(_LIB_PATH)
			and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
		return _LIB_PATH
	subprocess.run(["make", "-C", _HERE, "-B", "libvk_oracle.so"], check=True, capture_output=True)
	return _LIB_PATH


class Gap(C.Structure):
	_fields_ = [
		("kind", C.c_int32), ("u", C.c_float), ("v", C.c_float),
		("table", C.POINTER(C.c_float)), ("n_table", C.c_int32)]


class Corpus(C.Structure):
	_fields_ = [
		("layout", C.c_int32), ("d", C.c_int32),
		("n_tokens", C.c_int64), ("n_sentences", C.c_int64),
		("X", C.c_void_p), ("X_mag", C.c_void_p),
		("tok_id", C.c_void_p), ("E", C.c_void_p), ("V", C.c_int32),
		("sent_off", C.c_void_p), ("sent_end", C.c_void_p), ("pos_s", C.c_void_p),
		("X_f32", C.c_void_p), ("E_f32", C.c_void_p), ("tag_s", C.c_void_p)]


class Query(C.Structure):
	_fields_ = [
		("algorithm", C.c_int32), ("len_t", C.c_int32),
		("Q", C.c_void_p), ("Q_mag", C.c_void_p), ("q_ids", C.c_void_p),
		("locality", C.c_int32),
		("gap_s", Gap), ("gap_t", Gap),
		("submatch_weight", C.c_float),
		("max_matches", C.c_int32), ("min_score", C.c_float),
		("boost", C.c_void_p),
		("rwmd_injective", C.c_int32), ("rwmd_symmetric", C.c_int32), ("rwmd_normalize_bow", C.c_int32),
		("wrd_normalize_magnitudes", C.c_int32),
		("tag_weights", C.c_void_p), ("q_pos", C.c_void_p),
		("pos_mismatch_penalty", C.c_float), ("similarity_threshold", C.c_float), ("wmd_full", C.c_int32),
		("Q_f32", C.c_void_p), ("S_rows", C.c_void_p), ("q_tag", C.c_void_p)]


class Result(C.Structure):
	_fields_ = [
		("n_out", C.c_int32),
		("score", C.c_void_p), ("raw", C.c_void_p), ("sentence", C.c_void_p),
		("mapping", C.c_void_p), ("all_scores", C.c_void_p)]


_lib = None


def lib():
	global _lib
	if _lib is None:
		build()
		try:
			_lib = C.CDLL(_LIB_PATH)
		except OSError:
			build(force=True)
			_lib = C.CDLL(_LIB_PATH)
		L = _lib
		L.vko_f32_to_bf16.restype = C.c_uint16
		L.vko_f32_to_bf16.argtypes = [C.c_float]
		L.vko_gap_cost.restype = C.c_float
		L.vko_gap_cost.argtypes = [C.POINTER(Gap), C.c_int32]
		L.vko_score.restype = C.c_float
		L.vko_score.argtypes = [C.c_float, C.c_int32, C.c_int32, C.c_float, C.c_float]
		L.vko_rwmd.restype = C.c_float
		L.vko_wrd.restype = C.c_float
		L.vko_emd.restype = C.c_double
		for name in ("vko_align", "vko_align_general"):
			getattr(L, name).restype = C.c_int
			getattr(L, name).argtypes = [
				C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
				C.POINTER(Gap), C.POINTER(Gap), C.POINTER(C.c_float), C.c_void_p]
		L.vko_find.restype = C.c_int
		L.vko_find.argtypes = [C.POINTER(Corpus), C.POINTER(Query), C.POINTER(Result), C.c_int32]
	return _lib


def _ptr(a):
	return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
	return np.ascontiguousarray(a, dtype=np.float32)


def make_gap(spec, keep):
	"""spec: ('linear', u) | ('affine', u, v) | ('table', array) | float (linear)."""
	g = Gap()
	if isinstance(spec, (int, float)):
		spec = ("linear", float(spec))
	kind = spec[0]
	if kind == "linear":
		g.kind, g.u, g.v = GAP_LINEAR, float(spec[1]), 0.0
	elif kind == "affine":
		g.kind, g.u, g.v = GAP_AFFINE, float(spec[1]), float(spec[2])
	elif kind == "table":
		t = _f32(spec[1])
		keep.append(t)
		g.kind = GAP_TABLE
		g.table = t.ctypes.data_as(C.POINTER(C.c_float))
		g.n_table = len(t)
	else:
		raise ValueError(spec)
	return g


# ---- bf16 helpers ---------------------------------------------------------

def round_bf16(x):
	"""float32 array -> uint16 bf16 bit patterns (RNE)."""
	x = _f32(x)
	out = np.empty(x.shape, dtype=np.uint16)
	lib().vko_round_bf16(_ptr(x), _ptr(out), C.c_int64(x.size))
	return out


def bf16_to_f32(b):
	b = np.ascontiguousarray(b, dtype=np.uint16)
	return (b.astype(np.uint32) << 16).view(np.float32)


# ---- a1 -------------------------------------------------------------------

def magnitudes(x):
	x = _f32(x)
	out = np.empty(x.shape[0], dtype=np.float32)
	lib().vko_magnitudes(_ptr(x), C.c_int64(x.shape[0]), C.c_int32(x.shape[1]), _ptr(out))
	return out


def normalize_rows(x):
	x = _f32(x)
	out = np.empty_like(x)
	lib().vko_normalize_rows(_ptr(x), C.c_int64(x.shape[0]), C.c_int32(x.shape[1]), _ptr(out))
	return out


def normalize_rows_bf16(x):
	x = _f32(x)
	out = np.empty(x.shape, dtype=np.uint16)
	mag = np.empty(x.shape[0], dtype=np.float32)
	lib().vko_normalize_rows_bf16(_ptr(x), C.c_int64(x.shape[0]), C.c_int32(x.shape[1]), _ptr(out), _ptr(mag))
	return out, mag


# ---- similarity -----------------------------------------------------------

def sim_bf16(X, Q):
	X = np.ascontiguousarray(X, dtype=np.uint16)
	Q = np.ascontiguousarray(Q, dtype=np.uint16)
	S = np.empty((X.shape[0], Q.shape[0]), dtype=np.float32)
	lib().vko_sim_bf16(_ptr(X), C.c_int64(X.shape[0]), C.c_int32(X.shape[1]), _ptr(Q), C.c_int32(Q.shape[0]), _ptr(S))
	return S


def sim_f32(X, Q):
	X, Q = _f32(X), _f32(Q)
	S = np.empty((X.shape[0], Q.shape[0]), dtype=np.float32)
	lib().vko_sim_f32(_ptr(X), C.c_int64(X.shape[0]), C.c_int32(X.shape[1]), _ptr(Q), C.c_int32(Q.shape[0]), _ptr(S))
	return S


def sim_table_static_bf16(E, Q, q_ids=None):
	E = np.ascontiguousarray(E, dtype=np.uint16)
	Q = np.ascontiguousarray(Q, dtype=np.uint16)
	ids = None if q_ids is None else np.ascontiguousarray(q_ids, dtype=np.int32)
	T = np.empty((E.shape[0], Q.shape[0]), dtype=np.float32)
	lib().vko_sim_table_static_bf16(
		_ptr(E), C.c_int32(E.shape[0]), C.c_int32(E.shape[1]), _ptr(Q), C.c_int32(Q.shape[0]), _ptr(ids), _ptr(T))
	return T


# ---- alignment ------------------------------------------------------------

def align(S, locality, gap_s, gap_t, general=False):
	"""returns (raw_score, mapping[int16 len_t])."""
	S = _f32(S)
	len_s, len_t = S.shape
	keep = []
	gs, gt = make_gap(gap_s, keep), make_gap(gap_t, keep)
	raw = C.c_float()
	mapping = np.empty(len_t, dtype=np.int16)
	fn = lib().vko_align_general if general else lib().vko_align
	r = fn(_ptr(S), len_t, len_s, len_t, locality, C.byref(gs), C.byref(gt), C.byref(raw), _ptr(mapping))
	if r != 0:
		raise ValueError("vko_align failed: %d" % r)
	return raw.value, mapping


def gap_cost(gap, k):
	keep = []
	g = make_gap(gap, keep)
	return lib().vko_gap_cost(C.byref(g), k)


def score(raw, len_t, n_matched, submatch_weight=0.0, boost=1.0):
	return lib().vko_score(raw, len_t, n_matched, submatch_weight, boost)


def rwmd(S, ids_s=None, ids_t=None, injective=True, symmetric=True, normalize_bow=True):
	S = _f32(S)
	a = None if ids_s is None else np.ascontiguousarray(ids_s, dtype=np.int32)
	b = None if ids_t is None else np.ascontiguousarray(ids_t, dtype=np.int32)
	return lib().vko_rwmd(
		_ptr(S), C.c_int32(S.shape[1]), C.c_int32(S.shape[0]), C.c_int32(S.shape[1]),
		_ptr(a), _ptr(b), C.c_int32(int(injective)), C.c_int32(int(symmetric)), C.c_int32(int(normalize_bow)))


def wmd(S, ids_s=None, ids_t=None, relaxed=False, injective=False, symmetric=False, normalize_bow=False):
	S = _f32(S)
	a = None if ids_s is None else np.ascontiguousarray(ids_s, dtype=np.int32)
	b = None if ids_t is None else np.ascontiguousarray(ids_t, dtype=np.int32)
	L = lib()
	L.vko_wmd.restype = C.c_float
	return L.vko_wmd(
		_ptr(S), C.c_int32(S.shape[1]), C.c_int32(S.shape[0]), C.c_int32(S.shape[1]), _ptr(a), _ptr(b),
		C.c_int32(int(relaxed)), C.c_int32(int(injective)), C.c_int32(int(symmetric)), C.c_int32(int(normalize_bow)))


def wrd(S, mag_s, mag_t, normalize_magnitudes=True):
	S, mag_s, mag_t = _f32(S), _f32(mag_s), _f32(mag_t)
	return lib().vko_wrd(
		_ptr(S), C.c_int32(S.shape[1]), C.c_int32(S.shape[0]), C.c_int32(S.shape[1]),
		_ptr(mag_s), _ptr(mag_t), C.c_int32(int(normalize_magnitudes)))


def emd(a, b, Cm):
	a = np.ascontiguousarray(a, dtype=np.float64)
	b = np.ascontiguousarray(b, dtype=np.float64)
	Cm = np.ascontiguousarray(Cm, dtype=np.float64)
	F = np.empty_like(Cm)
	cost = lib().vko_emd(_ptr(a), C.c_int32(len(a)), _ptr(b), C.c_int32(len(b)), _ptr(Cm), _ptr(F))
	return cost, F


# ---- whole-corpus search --------------------------------------------------

def find_many(*, layout, d, sent_off, Qs, sent_end=None, X=None, X_mag=None, tok_id=None, E=None,
		algorithm=ALG_ALIGN, locality=LOCAL, gap_s=0.0, gap_t=0.0, q_ids=None, Q_mags=None,
		max_matches=10, min_score=0.0, boost=None, submatch_weight=0.0,
		rwmd=(True, True, True), wrd_normalize=True, n_threads=1, want_all_scores=False,
		pos_s=None, tag_weights=None, q_pos=None, pos_mismatch_penalty=0.0, similarity_threshold=0.0, wmd_full=False,
		S_rows=None, tag_s=None, q_tags=None):
	"""Runs vko_find_many over a batch of queries (Qs: list of uint16 bf16 [len_t x d]); q_ids / Q_mags
	are per-query lists or None.  S_rows: per-query list of float32 [n_tokens x len_t] similarity matrices the caller
	computed itself (contextual layout; the reference's one-sgemm-per-document form).  Returns a list of
	dict(score, raw, sentence, mapping[, all_scores])."""
	keep = []
	c = Corpus()
	sent_off = np.ascontiguousarray(sent_off, dtype=np.int64)
	n_sent = len(sent_off) - 1
	if sent_end is not None:
		sent_end = np.ascontiguousarray(sent_end, dtype=np.int64)
		n_sent = len(sent_end)
		c.sent_end = _ptr(sent_end)
	c.layout, c.d = layout, d
	c.n_tokens, c.n_sentences = int(max(sent_off[-1], sent_end[-1] if sent_end is not None and len(sent_end) else 0)), n_sent
	# rows of dtype float32 are taken as fp32 unit rows (the reference's own precision), uint16 as bf16 bits
	if X is not None and np.asarray(X).dtype == np.float32:
		X = np.ascontiguousarray(X, dtype=np.float32); c.X_f32 = _ptr(X)
	elif X is not None:
		X = np.ascontiguousarray(X, dtype=np.uint16); c.X = _ptr(X)
	if X_mag is not None:
		X_mag = _f32(X_mag); c.X_mag = _ptr(X_mag)
	if tok_id is not None:
		tok_id = np.ascontiguousarray(tok_id, dtype=np.int32); c.tok_id = _ptr(tok_id)
	if E is not None and np.asarray(E).dtype == np.float32:
		E = np.ascontiguousarray(E, dtype=np.float32); c.E_f32 = _ptr(E); c.V = E.shape[0]
	elif E is not None:
		E = np.ascontiguousarray(E, dtype=np.uint16); c.E = _ptr(E); c.V = E.shape[0]
	c.sent_off = _ptr(sent_off)
	if pos_s is not None:
		pos_s = np.ascontiguousarray(pos_s, dtype=np.int8); c.pos_s = _ptr(pos_s)
	if boost is not None:
		boost = _f32(boost)
	if tag_s is not None:
		tag_s = np.ascontiguousarray(tag_s, dtype=np.int8); c.tag_s = _ptr(tag_s)

	nq = len(Qs)
	qs = (Query * nq)()
	rs = (Result * nq)()
	bufs = []
	k = max_matches
	for i in range(nq):
		q = qs[i]
		if np.asarray(Qs[i]).dtype == np.float32:
			Q = np.ascontiguousarray(Qs[i], dtype=np.float32)
			q.Q_f32 = _ptr(Q)
		else:
			Q = np.ascontiguousarray(Qs[i], dtype=np.uint16)
			q.Q = _ptr(Q)
		keep.append(Q)
		q.algorithm, q.len_t = algorithm, Q.shape[0]
		if Q_mags is not None and Q_mags[i] is not None:
			m = _f32(Q_mags[i]); keep.append(m); q.Q_mag = _ptr(m)
		if q_ids is not None and q_ids[i] is not None:
			ids = np.ascontiguousarray(q_ids[i], dtype=np.int32); keep.append(ids); q.q_ids = _ptr(ids)
		q.locality = locality
		q.gap_s, q.gap_t = make_gap(gap_s, keep), make_gap(gap_t, keep)
		q.submatch_weight = submatch_weight
		q.max_matches, q.min_score = max_matches, min_score
		if boost is not None:
			q.boost = _ptr(boost)
		q.rwmd_injective, q.rwmd_symmetric, q.rwmd_normalize_bow = [int(x) for x in rwmd]
		q.wrd_normalize_magnitudes = int(wrd_normalize)
		q.wmd_full = int(bool(wmd_full))
		if q_tags is not None and q_tags[i] is not None:
			qt = np.ascontiguousarray(q_tags[i], dtype=np.int8); keep.append(qt); q.q_tag = _ptr(qt)
		if S_rows is not None and S_rows[i] is not None:
			sr = np.ascontiguousarray(S_rows[i], dtype=np.float32)
			if sr.shape != (c.n_tokens, Q.shape[0]):
				raise ValueError("S_rows must be [n_tokens x len_t]")
			keep.append(sr); q.S_rows = _ptr(sr)
		if tag_weights is not None and tag_weights[i] is not None:
			tw = _f32(tag_weights[i]); keep.append(tw); q.tag_weights = _ptr(tw)
			if q_pos is not None and q_pos[i] is not None:
				qp = np.ascontiguousarray(q_pos[i], dtype=np.int8); keep.append(qp); q.q_pos = _ptr(qp)
			q.pos_mismatch_penalty, q.similarity_threshold = float(pos_mismatch_penalty), float(similarity_threshold)
		score_ = np.zeros(k, dtype=np.float32)
		raw_ = np.zeros(k, dtype=np.float32)
		sent_ = np.zeros(k, dtype=np.int64)
		map_ = np.full((k, Q.shape[0]), -1, dtype=np.int16)
		all_ = np.zeros(n_sent, dtype=np.float32) if want_all_scores else None
		r = rs[i]
		r.score, r.raw, r.sentence, r.mapping, r.all_scores = _ptr(score_), _ptr(raw_), _ptr(sent_), _ptr(map_), _ptr(all_)
		bufs.append((score_, raw_, sent_, map_, all_))
	L = lib()
	L.vko_find_many.restype = C.c_int
	status = L.vko_find_many(C.byref(c), qs, C.c_int32(nq), rs, C.c_int32(n_threads))
	if status != 0:
		raise ValueError("vko_find failed: %d" % status)
	outs = []
	for i in range(nq):
		n = rs[i].n_out
		score_, raw_, sent_, map_, all_ = bufs[i]
		out = dict(score=score_[:n], raw=raw_[:n], sentence=sent_[:n], mapping=map_[:n])
		if want_all_scores:
			out["all_scores"] = all_
		outs.append(out)
	return outs


def find(*, Q, q_ids=None, Q_mag=None, len_t=None, tag_weights=None, q_pos=None, q_tag=None, **kw):
	"""one query (vko_find); see find_many"""
	return find_many(Qs=[Q], q_ids=None if q_ids is None else [q_ids], Q_mags=None if Q_mag is None else [Q_mag],
		tag_weights=None if tag_weights is None else [tag_weights], q_pos=None if q_pos is None else [q_pos],
		q_tags=None if q_tag is None else [q_tag], **kw)[0]
