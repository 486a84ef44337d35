"""Array-backed documents: the data contract `PreparedDocument` hands to the C++ core
(vectorian/corpus/document.py:626-662 -> core.Document, vectorian/core/cpp/document.cpp:30-58):
a token table and sentence spans in token units, optionally one contextual vector per
token.  Importers, on-disk storage (h5 / sqlite) and text normalisation of the reference
are out of scope; documents arrive tokenised.
"""

import numpy as np


class Document:
	def __init__(self, sentences, unique_id=None, metadata=None, contextual_embeddings=None, pos=None, tags=None,
			spans=None, token_mask=None):
		"""sentences: list of sentences, each a list of token strings.
		contextual_embeddings: {embedding name: float32 [n_tokens x d]}
		pos / tags: optional universal POS / Penn treebank tags, same nesting as `sentences`
		(Token.pos, Token.tag of the reference, vectorian/core/cpp/common.h:34-42)
		spans: further partition levels, {level: {"start": int[], "end": int[]}} in token units of the unmasked text
		(the span tables of the stored document, vectorian/corpus/document.py:641-649); the levels "sentence" (from
		`sentences`) and "token" (one span per token, document.cpp:52-53) always exist.
		token_mask: bool per token; False drops the token as the session's text normalisation does
		(`flavor_record.token_mask`, document.py:636-649): tokens, tags and contextual vectors are masked and every
		span table is re-indexed with the cumulative sum of the mask."""
		self._sentences = [list(s) for s in sentences]
		tokens = [t for s in self._sentences for t in s]
		n_raw = len(tokens)
		flat_pos = [t for s in pos for t in s] if pos is not None else None
		flat_tags = [t for s in tags for t in s] if tags is not None else None
		for name, seq in (("pos", flat_pos), ("tags", flat_tags)):
			if seq is not None and len(seq) != n_raw:
				raise ValueError(f"{name}: one entry per token expected")
		lens = np.array([len(s) for s in self._sentences], dtype=np.int64)
		tables = {"sentence": {
			"start": np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int32) if len(lens) else np.zeros(0, np.int32),
			"end": np.cumsum(lens).astype(np.int32)}}
		# the importers give every document one span over all of its tokens (vectorian/importers.py:30-36, 220-223):
		# session.partition("document") makes whole documents the slices
		tables["document"] = {"start": np.zeros(1, dtype=np.int32), "end": np.array([n_raw], dtype=np.int32)}
		for level, table in (spans or {}).items():
			if level in ("sentence", "token", "document"):
				raise ValueError(f"span level {level} is built in")
			st, en = np.asarray(table["start"], dtype=np.int32), np.asarray(table["end"], dtype=np.int32)
			if st.shape != en.shape or (len(st) and (st.min() < 0 or en.max() > n_raw or (en < st).any())):
				raise ValueError(f"span level {level}: spans outside the document")
			tables[level] = dict((k, np.asarray(v)) for k, v in table.items())
			tables[level]["start"], tables[level]["end"] = st, en
		contextual = {}
		for name, v in (contextual_embeddings or {}).items():
			v = np.ascontiguousarray(v, dtype=np.float32)
			if v.shape[0] != n_raw:
				raise ValueError(f"contextual embedding {name}: {v.shape[0]} vectors for {n_raw} tokens")
			contextual[name] = v
		if token_mask is not None:
			mask = np.asarray(token_mask, dtype=bool)
			if mask.shape != (n_raw,):
				raise ValueError("token_mask: one entry per token expected")
			reindex = np.cumsum(np.concatenate(([False], mask)), dtype=np.int32)       # document.py:641
			for table in tables.values():
				table["start"], table["end"] = reindex[table["start"]], reindex[table["end"]]
			tokens = [t for t, m in zip(tokens, mask) if m]
			flat_pos = None if flat_pos is None else [t for t, m in zip(flat_pos, mask) if m]
			flat_tags = None if flat_tags is None else [t for t, m in zip(flat_tags, mask) if m]
			contextual = dict((k, np.ascontiguousarray(v[mask])) for k, v in contextual.items())      # MaskedVectorsRef
		n = len(tokens)
		tables["token"] = {"start": np.arange(n, dtype=np.int32), "end": np.arange(1, n + 1, dtype=np.int32)}
		self._tokens, self._pos, self._tags = tokens, flat_pos, flat_tags
		self._spans = tables
		self._unique_id = unique_id
		self._metadata = metadata or {}
		self._contextual = contextual

	@property
	def unique_id(self):
		return self._unique_id

	@property
	def metadata(self):
		return self._metadata

	@property
	def tokens(self):
		return self._tokens

	@property
	def n_tokens(self):
		return len(self._tokens)

	@property
	def pos(self):
		return self._pos

	@property
	def tags(self):
		return self._tags

	@property
	def spans(self):
		return self._spans

	def n_spans(self, level="sentence"):
		return len(self._spans[level]["start"])

	def has_contextual_embedding(self, name):
		return name in self._contextual

	def contextual_vectors(self, name):
		return self._contextual[name]

	def max_len(self, level, window_size):
		# Document::max_len (vectorian/core/cpp/document.cpp): longest window in tokens
		st, en = self._spans[level]["start"], self._spans[level]["end"]
		n = len(st)
		best = 0
		for i in range(n):
			j = min(i + window_size - 1, n - 1)
			best = max(best, int(en[j] - st[i]))
		return best

	def span_tokens(self, level, slice_id, window_size=1):
		st, en = self._spans[level]["start"], self._spans[level]["end"]
		i = slice_id
		j = min(i + window_size - 1, len(st) - 1)
		return self._tokens[int(st[i]):int(en[j])]

	def span_info(self, partition_args, slice_id):
		"""PreparedDocument.span_info (vectorian/corpus/document.py:764-775): slice_id is already the
		index of the window's first span; the row of the span table is reported"""
		level = partition_args["level"] if isinstance(partition_args, dict) else partition_args.level
		table = self._spans[level]
		return dict((k, int(v[slice_id])) for k, v in table.items())


class Corpus:
	"""an ordered collection of documents (the in-memory part of vectorian/corpus/corpus.py:245)"""

	def __init__(self, docs=()):
		self._docs = list(docs)

	def add_doc(self, doc):
		self._docs.append(doc)

	@property
	def docs(self):
		return self._docs

	def __len__(self):
		return len(self._docs)

	def __iter__(self):
		return iter(self._docs)

	def __getitem__(self, i):
		return self._docs[i]

	def get_doc_index(self, doc):
		return self._docs.index(doc)
