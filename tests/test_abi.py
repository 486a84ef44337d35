"""CPU tier: the C-ABI library loads without a GPU, exports every symbol the header
declares, its pure-host entry point works, and compute entry points fail loudly."""

import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
	text = open(os.path.join(ROOT, "include", "vectorian_hip.h")).read()
	text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
	return sorted(set(re.findall(r"\b(vk_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
	from vectorian_amd import core
	lib = core.lib()
	syms = declared_symbols()
	assert len(syms) >= 15
	missing = [s for s in syms if not hasattr(lib, s)]
	assert not missing, missing
	assert sorted(core.EXPORTS) == syms
	assert lib.vk_abi_version() == 12


def test_document_pass_sizing():
	"""the host-side sizing of the one-wave-per-slice pass (internal helpers of the library, pure host code): the LDS ring of the
	column history covers the window a saturated gap table leaves (a power of two above ws_tail), the launch takes one scratch region
	per workgroup within its caps, and the launcher's own figures are what the query code sizes the scratch by"""
	import ctypes as C
	from vectorian_amd import core
	lib = core.lib()
	lib.vk_wide_ring_rows.restype = C.c_int32
	lib.vk_wide_ring_rows.argtypes = [C.c_int32] * 3
	lib.vk_wide_gs_blocks.restype = C.c_int32
	lib.vk_wide_gs_blocks.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32]
	lib.vk_wide_scratch_bytes.restype = C.c_size_t
	lib.vk_wide_scratch_bytes.argtypes = [C.c_int32] * 5
	lib.vk_wide_lds_demand.restype = C.c_size_t
	lib.vk_wide_lds_demand.argtypes = [C.c_int32] * 5
	for nq in (1, 2, 3, 4):
		for tail in (1, 5, 126, 127, 128, 300):
			ring = lib.vk_wide_ring_rows(nq, 2, tail)
			assert ring == 0 or (ring > tail and ring & (ring - 1) == 0 and ring * (16 * nq + 2) * 4 <= 40 * 1024)
		assert lib.vk_wide_ring_rows(nq, 2, 126) == 128                     # 1 - 2^(-k/5) saturates at k = 126
		assert lib.vk_wide_ring_rows(nq, 2, 0) == 0 and lib.vk_wide_ring_rows(nq, 0, 126) == 0 and lib.vk_wide_ring_rows(nq, 1, 126) == 0
	doc = core.VK_MAX_DOC_LEN
	# scoring: no state in the scratch without general gaps or with the ring; the whole history of a document otherwise
	assert lib.vk_wide_scratch_bytes(doc, 1, 0, 0, 0) == 16 and lib.vk_wide_scratch_bytes(doc, 1, 2, 0, 128) == 16
	assert lib.vk_wide_scratch_bytes(doc, 1, 2, 0, 0) >= (doc + 1) * 17 * 4
	# tracebacks: 3 bytes per cell (step lengths + flags), plus the history unless it is the ring
	assert (doc + 1) * 17 * 3 <= lib.vk_wide_scratch_bytes(doc, 1, 0, 1, 0) <= (doc + 1) * 17 * 3 + 64
	assert lib.vk_wide_scratch_bytes(doc, 4, 2, 1, 0) >= (doc + 1) * 65 * 7
	for gap, ring in ((0, 0), (2, 0), (2, 128)):
		for n in (1, 100, 10 ** 6):
			b = lib.vk_wide_gs_blocks(doc, 4, gap, 0, n, ring)
			assert 1 <= b <= min(2048, n) and b * lib.vk_wide_scratch_bytes(doc, 4, gap, 0, ring) <= 4 << 30
		assert lib.vk_wide_gs_blocks(doc, 4, gap, 18, 10 ** 6, ring) == 18          # tracebacks: one workgroup per winner
	# 64 query tokens over 512-token slices with general gaps and traceback: beyond the LDS (the global-state form takes them)
	assert lib.vk_wide_lds_demand(512, 4, 2, 0, 1) > 160 * 1024 >= lib.vk_wide_lds_demand(512, 1, 2, 0, 1)
	# the skewed sweeps (round 4): a winner's records -- a byte per cell, general gaps the cell's value -- for every row 0 .. len and
	# every column of the query's blocks, in whole 256-byte lines
	for fn, argtypes in (("vk_doc_scratch_bytes", [C.c_int32] * 2), ("vk_docw_scratch_bytes", [C.c_int32] * 2), ("vk_docg_scratch_bytes", [C.c_int32])):
		getattr(lib, fn).restype = C.c_size_t
		getattr(lib, fn).argtypes = argtypes
	for n in (1, 65, 512, 5000, doc):
		for gap, cell in ((0, 1), (1, 1), (2, 4)):
			b = lib.vk_doc_scratch_bytes(n, gap)
			assert b % 256 == 0 and (n + 1) * 16 * cell <= b <= (n + 2) * 16 * cell + 255
		for nq in (2, 3, 4):
			b = lib.vk_docw_scratch_bytes(n, nq)
			assert b % 256 == 0 and (n + 1) * 16 * nq <= b <= (n + 2) * 16 * nq + 255
		b = lib.vk_docg_scratch_bytes(n)
		assert b % 256 == 0 and (n + 1) * 32 * 4 <= b <= (n + 2) * 32 * 4 + 255


def test_no_gpu_means_loud_failure():
	import torch
	if torch.cuda.is_available():
		pytest.skip("a GPU is present")
	from vectorian_amd import core
	assert core.device_count() == 0
	with pytest.raises(core.VkError):
		core.init(0)
	with pytest.raises(core.VkError):
		core.Corpus(layout=core.VK_LAYOUT_CONTEXTUAL, d=8, n_tokens=4, n_sentences=1)


def test_merge_topk_is_resultset_extend():
	# ResultSet::extend (result_set.h:70-93): bounded merge, order score desc then sentence desc
	from vectorian_amd import core
	a, b = core.TopK(4, 3), core.TopK(4, 3)
	a.n = 3
	a.score[:3] = [0.9, 0.5, 0.5]; a.sentence[:3] = [7, 12, 3]
	a.mapping[:3] = [[0, 1, 2], [1, -1, 2], [-1, -1, 0]]
	b.n = 2
	b.score[:2] = [0.7, 0.5]; b.sentence[:2] = [100, 101]
	b.mapping[:2] = [[5, 6, 7], [9, -1, -1]]
	m = core.merge_topk([a, b], 3, 4)
	assert m.n == 4
	assert list(m.sentence[:4]) == [7, 100, 101, 12]
	assert list(m.mapping[1]) == [5, 6, 7] and list(m.mapping[3]) == [1, -1, 2]
	np.testing.assert_array_equal(m.score[:4], np.array([0.9, 0.7, 0.5, 0.5], np.float32))


@pytest.mark.parametrize("len_t", [1, 10, 16, 17, 40, 64])
def test_exchange_records_native_against_numpy(len_t):
	# vk_pack_records writes the layout shards.pack_topk_numpy spells out; vk_merge_records over the records of several
	# sets is vk_merge_topk over the sets
	from vectorian_amd import core, shards
	rng = np.random.default_rng(len_t)
	k, world = 6, 5
	sets, recs = [], []
	for r in range(world):
		t = core.TopK(k, len_t)
		t.n = int(rng.integers(0, k + 1))
		t.score[:t.n] = np.sort(rng.integers(0, 8, size=t.n).astype(np.float32) / 8)[::-1]   # ties across sets
		t.raw_score[:t.n] = rng.random(t.n)
		t.sentence[:t.n] = rng.integers(0, 1 << 40, size=t.n)
		t.mapping[:t.n] = rng.integers(-1, 64, size=(t.n, len_t))
		t.edge_sim[:t.n] = rng.random((t.n, len_t))
		rec = shards.pack_topk(t, 1000 * r, k)
		assert rec.shape == (k, core.record_words(len_t))
		np.testing.assert_array_equal(rec, shards.pack_topk_numpy(t, 1000 * r, k))
		back = shards.unpack_topk(rec, len_t)
		assert back.n == t.n
		np.testing.assert_array_equal(back.sentence[:t.n], t.sentence[:t.n] + 1000 * r)
		np.testing.assert_array_equal(back.mapping[:t.n], t.mapping[:t.n])
		sets.append(back)
		recs.append(rec)
	ref = core.merge_topk(sets, len_t, k)
	got = core.merge_records(np.stack(recs), world, len_t, k)
	assert got.n == ref.n
	for f in ("score", "raw_score", "sentence", "mapping", "edge_sim"):
		np.testing.assert_array_equal(getattr(got, f)[:got.n], getattr(ref, f)[:ref.n])


def test_merge_survives_nan_scores_and_carries_flags():
	# records arrive from other ranks unchecked: a NaN score must not break the sort (it ranks last); the per-query flag word
	# (shards.allgather_start: the spare last word of a query's first record) is not part of a record's payload
	from vectorian_amd import core, shards
	k, len_t = 6, 10
	recs = []
	for r, scores in enumerate(([0.9, np.nan, 0.3], [0.8, 0.5], [np.nan])):
		t = core.TopK(k, len_t)
		t.n = len(scores)
		t.score[:t.n] = scores
		t.sentence[:t.n] = np.arange(t.n) + 10 * r
		rec = shards.pack_topk(t, 0, k)
		rec[0, -1] = 5            # a flag word
		recs.append(rec)
	for _ in range(20):            # std::sort over a broken order is undefined: run it a few times
		got = core.merge_records(np.stack(recs), 3, len_t, k)
	assert got.n == 6
	assert list(got.score[:4]) == [np.float32(0.9), np.float32(0.8), np.float32(0.5), np.float32(0.3)]
	assert got.score[4] == -np.inf and got.score[5] == -np.inf and sorted(int(x) for x in got.sentence[4:6]) == [1, 20]


def test_struct_fields_agree_header_shim_and_integration_stub():
	# the ctypes structures of the shim and of INTEGRATION.md's stub list the fields of the header's structs, in order
	from vectorian_amd import core
	header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "vectorian_hip.h")).read(), flags=re.S)
	doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()

	def header_fields(name):
		end = header.index("} " + name + ";")
		body = header[header.rindex("typedef struct {", 0, end) + len("typedef struct {"):end]
		out = []
		for decl in body.split(";"):
			for part in decl.split(","):
				m = re.search(r"(\w+)\s*(\[\w*\])?\s*$", part.strip())
				if m and part.strip():
					out.append(m.group(1))
		return out

	def stub_fields(name):
		i = doc.index(f"class {name}(C.Structure)")
		j = min(x for x in (doc.find("\nclass ", i + 10), doc.find("\ndef check", i)) if x > 0)
		return re.findall(r'\("(\w+)"', doc[i:j])
	for cname, shim, stub in (("vk_query_desc", core._QueryDesc, "QueryDesc"), ("vk_topk_out", core._TopkOut, "TopkOut"), ("vk_corpus_desc", core._CorpusDesc, "CorpusDesc")):
		names = [f[0] for f in shim._fields_]
		assert header_fields(cname) == names, (cname, header_fields(cname), names)
		assert stub_fields(stub) == names, (stub, stub_fields(stub), names)


def test_rwmd_from_rows_is_the_oracles_relaxed_solver(oracle):
	# the host restatement of the winners' relaxed-WMD scores (vk_transport_host.h) against the oracle's vko_wmd on random rows: every
	# form, position vocabularies and token vocabularies with repeated ids on both sides, ties among the distances -- bit for bit
	from vectorian_amd import core
	rng = np.random.default_rng(5)
	n = 0
	for trial in range(400):
		len_s, len_t = int(rng.integers(1, 70)), int(rng.integers(1, 40))
		S = rng.random((len_s, len_t)).astype(np.float32)
		S[rng.random(S.shape) < 0.3] = 0.0                       # thresholded cells: ties at distance 1
		S[rng.random(S.shape) < 0.05] = np.float32(1.7)          # tag-weighted similarities exceed 1: distance clamped at 0
		if trial % 2:
			ids_s, ids_t = rng.integers(0, 12, size=len_s).astype(np.int32), rng.integers(0, 12, size=len_t).astype(np.int32)
			for i in range(len_s):                                  # equal tokens have equal rows / columns, as a similarity matrix has
				S[i] = S[int(np.nonzero(ids_s == ids_s[i])[0][0])]
			for j in range(len_t):
				S[:, j] = S[:, int(np.nonzero(ids_t == ids_t[j])[0][0])]
		else:
			ids_s = ids_t = None
		for inj, sym, nbow in ((True, True, True), (True, False, True), (True, False, False), (False, True, True), (False, False, True), (False, False, False)):
			ref = np.float32(oracle.rwmd(S, ids_s, ids_t, injective=inj, symmetric=sym, normalize_bow=nbow))
			got = core.rwmd_from_rows(S, ids_s, ids_t, injective=inj, symmetric=sym, normalize_bow=nbow)
			assert got.view(np.uint32) == ref.view(np.uint32), (trial, inj, sym, nbow, got, ref)
			n += 1
	assert n == 2400
	with pytest.raises(core.VkError):
		core.rwmd_from_rows(S, None, None, injective=True, symmetric=True, normalize_bow=False)
