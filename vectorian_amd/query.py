"""Query and PreparedQuery of the operator surface (vectorian/index.py:25-106): the text of a search with its options, and its
tokens / token ids as the index hands them to the backend."""

import numpy as np


def default_tokenizer(text):
	"""stand-in for the spaCy pipeline `nlp` of the reference (tokenisation is out of scope)"""
	return text.split()


class Query:
	def __init__(self, index, vocab, text, options):
		self._index = index
		self._vocab = vocab
		self._text = text
		self._options = options
		self._abort = np.zeros(1, dtype=np.int32)   # the flag the native side polls (vk_query_desc.abort)

	def abort(self):
		"""core.Query.abort (vectorian/core/cpp/module.cpp:120, query.h:183-189): may be called from another thread; a search that
		has not started its device work yet (or the rest of a batch) is dropped and returns no matches"""
		self._abort[0] = 1

	@property
	def aborted(self):
		return bool(self._abort[0])

	@property
	def index(self):
		return self._index

	@property
	def text(self):
		return self._text

	@property
	def options(self):
		return self._options

	def prepare(self, nlp):
		return PreparedQuery(self, self._vocab, nlp)


class PreparedQuery:
	"""tokenised query + its vectors (vectorian/index.py:56-106; the spaCy specifics dropped)"""

	def __init__(self, query, vocab, nlp):
		self._query = query
		raw = list((nlp or default_tokenizer)(query.text))
		# nlp may return plain strings, or dicts with 'text' / 'pos' / 'tag' (as spaCy's doc.to_json()["tokens"])
		tokens, self._pos, self._tags = [], [], []
		for t in raw:
			if isinstance(t, dict):
				tokens.append(t["text"]); self._pos.append(t.get("pos")); self._tags.append(t.get("tag"))
			else:
				tokens.append(t); self._pos.append(None); self._tags.append(None)
		self._tokens = tokens
		# QueryVocabulary (vectorian/core/cpp/vocabulary.h:500-541) is an incremental lexicon over the session's: a query token the
		# corpus does not hold gets a NEW id behind the session's, the same word the same id -- so that the bags of words of the
		# transport strategies merge repeated words, and only those (all unknown words as one id -1 would merge them all)
		ids, fresh = [], {}
		for t in tokens:
			i = vocab.token_to_id(t)
			if i < 0:
				i = fresh.setdefault(t, vocab.size + len(fresh))
			ids.append(i)
		self._token_ids = np.array(ids, dtype=np.int32)

	@property
	def index(self):
		return self._query.index

	@property
	def text_str(self):
		return self._query.text

	@property
	def options(self):
		return self._query.options

	@property
	def tokens(self):
		return self._tokens

	@property
	def token_ids(self):
		return self._token_ids

	@property
	def pos(self):
		return self._pos

	@property
	def tags(self):
		return self._tags

	@property
	def n_tokens(self):
		return len(self._tokens)

	def __len__(self):
		return len(self._tokens)

