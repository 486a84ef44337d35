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


def assert_same_results(got, ref, *, score_tol=1e-4, tie_tol=2e-6, check_mapping=True):
	"""got: core.TopK.trimmed(); ref: oracle.find() dict.
	Scores within score_tol; identical sentence ids and mappings, except where the
	oracle's own scores are closer than tie_tol (fp32 accumulation-order ties)."""
	n = len(ref["score"])
	assert len(got["score"]) == n, (len(got["score"]), n)
	np.testing.assert_allclose(got["score"], ref["score"], atol=score_tol, rtol=0)
	for i in range(n):
		if got["sentence"][i] != ref["sentence"][i]:
			near = np.abs(ref["score"] - ref["score"][i]) <= tie_tol
			assert near.sum() > 1, (i, got["sentence"][i], ref["sentence"][i], ref["score"][:n])
			continue
		if check_mapping:
			assert (got["mapping"][i] == ref["mapping"][i]).all(), (i, got["mapping"][i], ref["mapping"][i])
