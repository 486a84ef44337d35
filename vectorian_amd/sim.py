"""Similarity operators: the surface of `vectorian.sim` on the hot path.

VectorSim / CosineSim   vectorian/sim/vector.py:17-78
TokenSim / EmbeddingTokenSim   vectorian/sim/token.py:4-44
SpanSim / OptimizedSpanSim     vectorian/sim/span.py:9-71
The other VectorSims and the modifier combinators of the reference are out of scope
(SURVEY 2.1); the tag-weighted variant is a "next" row and is rejected
explicitly rather than ignored.
"""

import numpy as np

from vectorian_amd.alignment import ConstantGapCost, LocalAlignment, Optimizer
from vectorian_amd.embedding import AbstractVectors


class VectorSim:
	"""a scalar similarity in [0, 1] for pairs of vectors (vectorian/sim/vector.py:17-44)"""

	def __call__(self, a: AbstractVectors, b: AbstractVectors, out: np.ndarray):
		self.compute(a, b, out)

	def compute(self, a: AbstractVectors, b: AbstractVectors, out: np.ndarray):
		raise NotImplementedError()

	@property
	def name(self) -> str:
		raise NotImplementedError()


class CosineSim(VectorSim):
	"""cosine of the angle between two vectors (vectorian/sim/vector.py:63-82); negative
	values are clipped to 0 downstream (SimilarityMatrix::clip, metric/metric.h:28-30)"""

	def compute(self, a, b, out):
		np.linalg.multi_dot([a.normalized, b.normalized.T], out=out)

	@property
	def name(self):
		return "cosine"


class TokenSim:
	@property
	def is_modifier(self):
		return False

	@property
	def name(self):
		raise NotImplementedError()

	@property
	def embeddings(self):
		raise NotImplementedError()


class EmbeddingTokenSim(TokenSim):
	def __init__(self, embedding, sim: VectorSim):
		self._embedding = embedding
		self._sim = sim

	@property
	def name(self):
		return f"{self._sim.name}[{self._embedding.name}]"

	@property
	def embeddings(self):
		return [self._embedding]

	@property
	def embedding(self):
		return self._embedding

	@property
	def similarity(self):
		return self._sim

	def to_args(self, index):
		return {
			"name": self._embedding.name + "-" + self._sim.name,
			"embedding": self._embedding.name,
			"metric": self._sim
		}


class SpanSim:
	def create_index(self, partition, **kwargs):
		raise NotImplementedError()

	def to_args(self, index):
		raise NotImplementedError()


class OptimizedSpanSim(SpanSim):
	def __init__(self, token_sim: TokenSim, optimizer: Optimizer = None, tag_weights: dict = None, **kwargs):
		if not isinstance(token_sim, TokenSim):
			raise TypeError(token_sim)
		if optimizer is None:
			# vectorian/sim/span.py:28-32
			optimizer = LocalAlignment(gap={"s": ConstantGapCost(0), "t": ConstantGapCost(0)})
		if not isinstance(optimizer, Optimizer):
			raise TypeError(optimizer)
		self._token_sim = token_sim
		self._optimizer = optimizer
		self._tag_weights = tag_weights
		self._options = kwargs

	@property
	def token_sim(self):
		return self._token_sim

	@property
	def optimizer(self):
		return self._optimizer

	def create_index(self, partition, **kwargs):
		# the plugin seam: SpanSim.create_index -> Index (vectorian/sim/span.py:50-51).
		# The reference returns BruteForceIndex here; this package returns its MI355X twin.
		from vectorian_amd.index import HipBruteForceIndex
		return HipBruteForceIndex(partition, self, **kwargs)

	def to_args(self, index):
		if not self._tag_weights:
			if self._options:
				raise ValueError(f"illegal option(s): {', '.join(self._options.keys())}")
			return {
				"metric": "alignment-isolated",
				"token_metric": self._token_sim,
				"alignment": self._optimizer.to_args(index.partition)
			}
		else:
			return {
				"metric": "alignment-tag-weighted",
				"token_metric": self._token_sim,
				"alignment": self._optimizer.to_args(index.partition),
				"pos_mismatch_penalty": self._options.get("pos_mismatch_penalty", 0),
				"similarity_threshold": self._options.get("similarity_threshold", 0),
				"tag_weights": self._tag_weights
			}


class SpanEmbedding:
	"""one vector per span (vectorian/embedding/span.py).  The sentence encoders of the reference are out of
	scope; vectors come precomputed, `encode(texts) -> [n x d]` embeds queries."""

	def __init__(self, name, dimension, encode):
		self._name, self._dimension, self._encode = name, int(dimension), encode

	@property
	def name(self):
		return self._name

	@property
	def dimension(self):
		return self._dimension

	def encode(self, texts):
		v = np.ascontiguousarray(self._encode(texts), dtype=np.float32)
		return v.reshape(len(texts), self._dimension)


class EmbeddedSpanSim(SpanSim):
	"""span similarity by ONE embedding per span (vectorian/sim/span.py:74-95)"""

	def __init__(self, embedding: SpanEmbedding, sim: VectorSim = None):
		if sim is None:
			sim = CosineSim()
		if not isinstance(sim, VectorSim):
			raise TypeError(f"{sim} is expected to be a VectorSim")
		self._embedding = embedding
		self._vector_sim = sim

	@property
	def embedding(self):
		return self._embedding

	def create_index(self, partition, **kwargs):
		# the reference picks FaissCosineIndex / SpanEncoderIndex here (vectorian/sim/span.py:83-88)
		from vectorian_amd.index import HipSpanEncoderIndex
		if not isinstance(self._vector_sim, CosineSim):
			raise NotImplementedError(f"{self._vector_sim.name}: the HIP path computes CosineSim")
		return HipSpanEncoderIndex(partition, self._embedding, self, **kwargs)

	def to_args(self, index):
		return None

	@property
	def name(self):
		return "EmbeddedSpanSim"
