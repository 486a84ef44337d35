"""Seeded synthetic corpora of the shapes named in BASELINE.md / SURVEY.md 8(d).

The reference's pretrained embeddings (GloVe, fastText, spaCy transformers) have
to be downloaded and are unavailable offline; these generators produce vectors of
the same dimensionality with a cosine spread that exercises the clip and the DP.
"""

import numpy as np

SEED_VOCAB, SEED_CORPUS, SEED_QUERY = 1234, 2345, 3456


def to_bf16_bits(x):
	"""float32 -> bf16 bit patterns (uint16), round to nearest even, NaN kept NaN."""
	x = np.ascontiguousarray(x, dtype=np.float32)
	u = x.view(np.uint32)
	nan = (u & 0x7FFFFFFF) > 0x7F800000
	r = ((u + (0x7FFF + ((u >> 16) & 1))) >> 16).astype(np.uint16)
	r[nan] = ((u[nan] >> 16) | 0x0040).astype(np.uint16)
	return r


def bf16_bits_to_f32(b):
	return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def normalize_rows(x):
	"""Vectors.normalized (vectorian/embedding/vectors.py:71-80): x / |x|, NaN -> 0."""
	x = np.ascontiguousarray(x, dtype=np.float32)
	mag = np.sqrt(np.sum(x.astype(np.float64) ** 2, axis=1)).astype(np.float32)
	with np.errstate(divide="ignore", invalid="ignore"):
		out = x / mag[:, None]
	np.nan_to_num(out, copy=False, nan=0.0)
	return out


def make_vocab(V, d, seed=SEED_VOCAB, n_centres=512, noise=0.6):
	"""clustered Gaussian vocabulary (unnormalised): cosines spread over about [-0.2, 0.9]"""
	rng = np.random.default_rng(seed)
	centres = rng.standard_normal((n_centres, d)).astype(np.float32)
	z = rng.integers(0, n_centres, size=V)
	return centres[z] + noise * rng.standard_normal((V, d)).astype(np.float32)


def zipf_ids(n, V, rng, a=1.1):
	p = 1.0 / np.arange(1, V + 1) ** a
	cdf = np.cumsum(p)
	cdf /= cdf[-1]
	return np.searchsorted(cdf, rng.random(n)).astype(np.int32).clip(0, V - 1)


def make_sentences(n_sent, min_len, max_len, seed=SEED_CORPUS):
	rng = np.random.default_rng(seed)
	lens = rng.integers(min_len, max_len + 1, size=n_sent) if max_len > min_len else np.full(n_sent, min_len)
	off = np.zeros(n_sent + 1, dtype=np.int64)
	np.cumsum(lens, out=off[1:])
	return off


def make_static_corpus(n_sent, min_len, max_len, V, d, seed=SEED_CORPUS):
	"""token ids + vocabulary table (the reference's static layout)."""
	E = make_vocab(V, d)
	off = make_sentences(n_sent, min_len, max_len, seed)
	rng = np.random.default_rng(seed + 1)
	ids = zipf_ids(int(off[-1]), V, rng)
	return dict(E=E, tok_id=ids, sent_off=off)


def make_contextual_corpus(n_sent, min_len, max_len, V, d, seed=SEED_CORPUS, noise=0.1, norm_sigma=0.0):
	"""one vector per token occurrence: vocabulary vector + per-token noise."""
	st = make_static_corpus(n_sent, min_len, max_len, V, d, seed)
	rng = np.random.default_rng(seed + 2)
	X = st["E"][st["tok_id"]] + noise * rng.standard_normal((len(st["tok_id"]), d)).astype(np.float32)
	if norm_sigma > 0:
		X = X * rng.lognormal(0.0, norm_sigma, size=(X.shape[0], 1)).astype(np.float32)
	st["X"] = np.ascontiguousarray(X, dtype=np.float32)
	return st


def make_queries(corpus, n_queries, len_t, seed=SEED_QUERY, noise=0.05):
	"""half of the queries are noisy copies of len_t consecutive tokens of a corpus sentence.
	Vectors are unnormalised (the library / Vectors.normalized normalises them)."""
	rng = np.random.default_rng(seed)
	E, ids, off = corpus["E"], corpus["tok_id"], corpus["sent_off"]
	d = E.shape[1]
	out = []
	for qi in range(n_queries):
		if qi % 2 == 0:
			for _ in range(100):
				s = int(rng.integers(0, len(off) - 1))
				if off[s + 1] - off[s] >= len_t:
					break
			start = int(off[s]) + int(rng.integers(0, off[s + 1] - off[s] - len_t + 1)) if off[s + 1] - off[s] >= len_t else int(off[s])
			qids = ids[start:start + len_t].copy()
			if len(qids) < len_t:
				qids = np.concatenate([qids, zipf_ids(len_t - len(qids), E.shape[0], rng)])
		else:
			qids = zipf_ids(len_t, E.shape[0], rng)
		qv = E[qids] + noise * rng.standard_normal((len_t, d)).astype(np.float32)
		out.append(dict(ids=qids.astype(np.int32), vectors=np.ascontiguousarray(qv, dtype=np.float32)))
	return out
