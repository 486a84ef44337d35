"""Vector containers and array-backed token embeddings.

`Vectors` mirrors vectorian/embedding/vectors.py:60-86 (the only numerics of that
module on the hot path: magnitudes, normalized).  The loaders of the reference
(gensim / fastText / spaCy / model zoo downloads) are out of scope;
embeddings are supplied as arrays.
"""

import numpy as np


class AbstractVectors:
	@property
	def unmodified(self):
		raise NotImplementedError()

	@property
	def normalized(self):
		raise NotImplementedError()

	@property
	def magnitudes(self):
		raise NotImplementedError()


class Vectors(AbstractVectors):
	def __init__(self, unmodified):
		self._unmodified = np.ascontiguousarray(unmodified, dtype=np.float32)
		self._normalized = None
		self._magnitudes = None

	@property
	def shape(self):
		return self._unmodified.shape

	@property
	def unmodified(self):
		return self._unmodified

	@property
	def magnitudes(self):
		# np.linalg.norm(axis=1) + nan_to_num (vectors.py:82-86)
		if self._magnitudes is None:
			data = np.linalg.norm(self._unmodified, axis=1)
			np.nan_to_num(data, copy=False, nan=0)
			self._magnitudes = data
		return self._magnitudes

	@property
	def normalized(self):
		# unmodified / magnitudes, NaN -> 0 (vectors.py:71-80; the zeroing of "vanishing"
		# rows there acts on a copy and has no effect)
		if self._normalized is None:
			with np.errstate(divide="ignore", invalid="ignore"):
				data = self._unmodified / self.magnitudes[:, np.newaxis]
			np.nan_to_num(data, copy=False, nan=0)
			self._normalized = data
		return self._normalized


class TokenEmbedding:
	@property
	def is_static(self):
		return False

	@property
	def is_contextual(self):
		return False

	@property
	def name(self):
		raise NotImplementedError()

	@property
	def dimension(self):
		raise NotImplementedError()


class StaticEmbedding(TokenEmbedding):
	"""word -> vector table (what the reference's keyed / fastText embeddings resolve to,
	vectorian/embedding/token/keyed.py:72-109).  Out-of-vocabulary words get a zero vector."""

	def __init__(self, name, words, vectors):
		vectors = np.ascontiguousarray(vectors, dtype=np.float32)
		if len(words) != vectors.shape[0]:
			raise ValueError("one vector per word expected")
		self._name = name
		self._token2id = dict((t, i) for i, t in enumerate(words))
		self._vectors = vectors

	@property
	def is_static(self):
		return True

	@property
	def name(self):
		return self._name

	@property
	def dimension(self):
		return self._vectors.shape[1]

	def word_vec(self, t):
		k = self._token2id.get(t)
		if k is not None:
			return self._vectors[k]
		return np.zeros((self.dimension,), dtype=np.float32)

	def encode_tokens(self, tokens):
		indices = np.array([self._token2id.get(t, -1) for t in tokens], dtype=np.int64)
		data = self._vectors[np.maximum(indices, 0)].copy()
		data[indices < 0, :] = 0
		return Vectors(data.reshape(len(tokens), self.dimension))


class ContextualEmbedding(TokenEmbedding):
	"""one vector per token occurrence (vectorian/embedding/token/contextual.py).  Document
	vectors are attached to the documents; `encoder(tokens) -> [len x d]` embeds a query."""

	def __init__(self, name, dimension, encoder):
		self._name = name
		self._dimension = int(dimension)
		self._encoder = encoder

	@property
	def is_contextual(self):
		return True

	@property
	def name(self):
		return self._name

	@property
	def dimension(self):
		return self._dimension

	def encode_tokens(self, tokens):
		v = np.ascontiguousarray(self._encoder(tokens), dtype=np.float32)
		if v.shape != (len(tokens), self._dimension):
			raise ValueError(f"encoder returned {v.shape}, expected {(len(tokens), self._dimension)}")
		return Vectors(v)
