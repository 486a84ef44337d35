"""Session / Partition / Result: the orchestration surface of vectorian/session.py.

Session (session.py:165)  holds the documents, the token embeddings and the shared
vocabulary; Partition (session.py:85) selects the slicing of documents into spans
(level, window_size, window_step) and builds an Index through `SpanSim.create_index`
(session.py:134-142); Result (session.py:24) carries the matches of one `find`.
"""

import collections

import numpy as np

from vectorian_amd.corpus import Corpus, Document
from vectorian_amd.embedding import TokenEmbedding
from vectorian_amd.sim import CosineSim, EmbeddingTokenSim, OptimizedSpanSim, SpanSim


class Result:
	def __init__(self, index, matches, duration):
		self._index = index
		self._matches = matches
		self._duration = duration

	@property
	def index(self):
		return self._index

	@property
	def matches(self):
		return self._matches

	def __len__(self):
		return len(self._matches)

	def __iter__(self):
		return iter(self._matches)

	def __getitem__(self, i):
		return self._matches[i]

	def to_json(self, context_size=10):
		return [m.to_json(context_size) for m in self._matches]

	def limit_to(self, n):
		return type(self)(self._index, self._matches[:n], self._duration)

	@property
	def duration(self):
		return self._duration


Slice = collections.namedtuple("Slice", ["level", "start", "end"])


class Vocabulary:
	"""token string <-> id over the session's documents (the role of core.Vocabulary,
	vectorian/core/cpp/vocabulary.h:183, for the search path only)"""

	def __init__(self):
		self._token2id = {}
		self._tokens = []

	def add(self, token):
		i = self._token2id.get(token)
		if i is None:
			i = len(self._tokens)
			self._token2id[token] = i
			self._tokens.append(token)
		return i

	def token_to_id(self, token):
		return self._token2id.get(token, -1)

	def id_to_token(self, i):
		return self._tokens[i]

	@property
	def size(self):
		return len(self._tokens)

	@property
	def tokens(self):
		return self._tokens


class Partition:
	def __init__(self, session, level, window_size, window_step):
		self._session = session
		self._level = level
		self._window_size = window_size
		self._window_step = window_step

	@property
	def contiguous(self):
		return self._window_step <= self._window_size

	@property
	def session(self):
		return self._session

	@property
	def level(self):
		return self._level

	@property
	def window_size(self):
		return self._window_size

	@property
	def window_step(self):
		return self._window_step

	def to_args(self):
		return {
			"level": self._level,
			"window_size": self._window_size,
			"window_step": self._window_step
		}

	@property
	def cache_key(self):
		return self._level, self._window_size, self._window_step

	def max_len(self):
		return self._session.max_len(self._level, self._window_size)

	def index(self, metric, nlp=None, **kwargs):
		if not isinstance(metric, SpanSim):
			raise TypeError(metric)
		if nlp:
			kwargs = kwargs.copy()
			kwargs["nlp"] = nlp
		return metric.create_index(self, **kwargs)

	def slice_id_to_slice(self, slice_id):
		return Slice(self._level, self._window_step * slice_id, self._window_size)


class Session:
	def __init__(self, corpus, embeddings=None, normalization=None):
		if not isinstance(corpus, Corpus):
			corpus = Corpus(corpus)
		self._corpus = corpus
		self._normalization = normalization
		embeddings = embeddings or []
		self._token_embeddings = tuple(e for e in embeddings if isinstance(e, TokenEmbedding))
		for embedding in self._token_embeddings:
			if embedding.is_contextual:
				for i, doc in enumerate(corpus):
					if not doc.has_contextual_embedding(embedding.name):
						raise RuntimeError(f"doc {doc.unique_id or i} misses contextual embedding {embedding.name}")
		self._embedding_encoders = collections.OrderedDict((e.name, e) for e in self._token_embeddings)
		self._pos_codes = {}
		self._tag_codes = {}
		self._vocab = Vocabulary()
		self._doc_token_ids = []
		for doc in corpus:
			self._doc_token_ids.append(np.array([self._vocab.add(t) for t in doc.tokens], dtype=np.int32))

	@property
	def corpus(self):
		return self._corpus

	@property
	def vocab(self):
		return self._vocab

	@property
	def normalization(self):
		return self._normalization

	@property
	def documents(self):
		return self._corpus.docs

	def doc_token_ids(self, doc_index):
		return self._doc_token_ids[doc_index]

	def pos_code(self, pos):
		"""small integer per universal POS string (the reference stores Token.pos as int8, common.h:34-42)"""
		c = self._pos_codes.get(pos)
		if c is None:
			c = len(self._pos_codes) + 1
			if c > 127:
				raise ValueError("more than 127 distinct POS values")
			self._pos_codes[pos] = c
		return c

	def tag_code(self, tag):
		"""small integer per fine-grained tag string (Token.tag, int8, common.h:34-42)"""
		c = self._tag_codes.get(tag)
		if c is None:
			c = len(self._tag_codes) + 1
			if c > 127:
				raise ValueError("more than 127 distinct tag values")
			self._tag_codes[tag] = c
		return c

	def pos_id(self, pos):
		"""code of a POS string seen in the documents, -1 otherwise (Vocabulary::unsafe_pos_id, vocabulary.h:392-394)"""
		return self._pos_codes.get(pos, -1)

	def tag_id(self, tag):
		return self._tag_codes.get(tag, -1)

	@property
	def encoders(self):
		return self._embedding_encoders

	def to_encoder(self, embedding):
		return self._embedding_encoders[embedding.name]

	def default_metric(self):
		return OptimizedSpanSim(EmbeddingTokenSim(self._token_embeddings[0], CosineSim()))

	def max_len(self, level, window_size):
		return max([doc.max_len(level, window_size) for doc in self.documents] or [0])

	def make_result(self, *args, **kwargs):
		return Result(*args, **kwargs)

	def on_progress(self, task, disable_progress=False):
		return task(None)

	def partition(self, level, window_size=1, window_step=None):
		if window_step is None:
			window_step = window_size
		return Partition(self, level, window_size, window_step)

	def index(self, *args, **kwargs):
		return self.partition("sentence").index(*args, **kwargs)
