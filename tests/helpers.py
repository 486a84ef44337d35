"""Shared helpers of the parity tests: build the same corpus for the oracle (CPU
checker) and for the HIP path (C-ABI), compare result sets."""

import numpy as np

from vectorian_amd import synth


def prep_contextual(corpus):
	"""unit rows rounded to bf16 -- the values both sides consume"""
	Xb = synth.to_bf16_bits(synth.normalize_rows(corpus["X"]))
	return Xb


def prep_query(q):
	return synth.to_bf16_bits(synth.normalize_rows(q["vectors"]))


def hip_contextual_corpus(core, corpus, Xb=None, keep_magnitudes=False):
	Xb = prep_contextual(corpus) if Xb is None else Xb
	off = corpus["sent_off"]
	c = core.Corpus(layout=core.VK_LAYOUT_CONTEXTUAL, d=Xb.shape[1], n_tokens=Xb.shape[0],
		n_sentences=len(off) - 1, keep_magnitudes=keep_magnitudes)
	c.append_vectors(Xb, normalize=False)
	c.set_sentences(off)
	c.finalize()
	return c


def hip_static_corpus(core, corpus):
	Eb = synth.to_bf16_bits(synth.normalize_rows(corpus["E"]))
	off = corpus["sent_off"]
	c = core.Corpus(layout=core.VK_LAYOUT_STATIC, d=Eb.shape[1], n_tokens=len(corpus["tok_id"]),
		n_sentences=len(off) - 1, vocab_size=Eb.shape[0])
	c.append_vectors(Eb, normalize=False)
	c.set_token_ids(corpus["tok_id"])
	c.set_sentences(off)
	c.finalize()
	return c, Eb


def assert_same_results(got, ref, *, score_tol=1e-4, tie_tol=2e-6, check_mapping=True, exact=None):
	"""got: core.TopK.trimmed(); ref: oracle.find() dict.

	exact (default: whenever the mappings are compared, i.e. alignments with traceback): the library restates its
	winners and a few runners-up in the oracle's own arithmetic (sim_canon), so slice ids, scores and
	tracebacks must equal the oracle's BIT FOR BIT -- no tolerance, no tie rule, repeated words or not.

	Otherwise (transport scores, alignments without traceback: MFMA cosines, fp32 accumulation in another order):
	scores within score_tol; a slice id may differ from the oracle's only if it is one of the slices the oracle
	itself scores within tie_tol of that place, or if the place ties with the last one (the k boundary: the
	oracle's next-best slice is not in `ref`)."""
	n = len(ref["score"])
	assert len(got["score"]) == n, (len(got["score"]), n)
	if exact is None:
		exact = check_mapping
	if exact:
		assert (np.asarray(got["sentence"]) == np.asarray(ref["sentence"])).all(), (got["sentence"], ref["sentence"], got["score"], ref["score"])
		a = np.asarray(got["score"], dtype=np.float32).view(np.uint32)
		b = np.asarray(ref["score"], dtype=np.float32).view(np.uint32)
		assert (a == b).all(), (got["score"], ref["score"], np.asarray(got["score"], dtype=np.float64) - np.asarray(ref["score"], dtype=np.float64))
		if check_mapping:
			assert (np.asarray(got["mapping"]) == np.asarray(ref["mapping"])).all(), (got["mapping"], ref["mapping"])
		return
	np.testing.assert_allclose(got["score"], ref["score"], atol=score_tol, rtol=0)
	rs = np.asarray(ref["score"], dtype=np.float64)
	for i in range(n):
		if got["sentence"][i] != ref["sentence"][i]:
			tied = {int(ref["sentence"][j]) for j in range(n) if abs(rs[j] - rs[i]) <= tie_tol}
			at_boundary = abs(rs[i] - rs[n - 1]) <= tie_tol
			assert int(got["sentence"][i]) in tied or at_boundary, (i, got["sentence"][i], ref["sentence"][i], ref["score"][:n])
			continue
		if check_mapping:
			assert (got["mapping"][i] == ref["mapping"][i]).all(), (i, got["mapping"][i], ref["mapping"][i])


def assert_json_close(a, b, tol, path="$"):
	"""two JSON-like structures: same shape, keys and non-float leaves; floats within tol"""
	if isinstance(a, dict):
		assert isinstance(b, dict) and set(a) == set(b), (path, a, b)
		for k in a:
			assert_json_close(a[k], b[k], tol, f"{path}.{k}")
	elif isinstance(a, (list, tuple)):
		assert isinstance(b, (list, tuple)) and len(a) == len(b), (path, a, b)
		for i, (x, y) in enumerate(zip(a, b)):
			assert_json_close(x, y, tol, f"{path}[{i}]")
	elif isinstance(a, (float, np.floating)):
		assert abs(float(a) - float(b)) <= tol, (path, a, b)
	else:
		assert a == b, (path, a, b)
