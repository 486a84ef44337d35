"""Matches of a search: the abstract Match interface and PyMatch of the reference (vectorian/index.py:192-292, 382-431), and
HipMatch -- what CoreMatch (vectorian/index.py:295-379) exposes, over the arrays the HIP backend returned."""

import collections

import numpy as np

from vectorian_amd import core
from vectorian_amd.flows import _rows_room

Region = collections.namedtuple("Region", ["s", "match", "gap_penalty"])
TokenMatch = collections.namedtuple("TokenMatch", ["pos_s", "edges"])
TokenMatchEdge = collections.namedtuple("TokenMatchEdge", ["t", "flow", "distance", "metric"])
TokenMatchT = collections.namedtuple("TokenMatchT", ["text", "index", "pos"])


class Match:
	"""abstract match interface (vectorian/index.py:192-292)"""

	@property
	def index(self):
		raise NotImplementedError()

	@property
	def partition(self):
		return self.index.partition

	@property
	def query(self):
		raise NotImplementedError()

	@property
	def doc(self):
		return self.prepared_doc

	@property
	def prepared_doc(self):
		raise NotImplementedError()

	@property
	def slice_id(self):
		raise NotImplementedError()

	@property
	def slice(self):
		return self.partition.slice_id_to_slice(self.slice_id)

	@property
	def score(self):
		raise NotImplementedError()

	@property
	def metric(self):
		raise NotImplementedError()

	@property
	def omitted(self):
		raise NotImplementedError()

	def regions(self, context_size=10):
		raise NotImplementedError()

	@property
	def level(self):
		raise NotImplementedError()

	@property
	def flow(self):
		return None

	def to_json(self, context_size=10):
		regions = []
		partition = self.query.options["partition"]
		span_info = self.prepared_doc.span_info(partition, self.slice_id)
		for region in self.regions(context_size):
			s = region.s
			if region.match:
				edges = []
				for e in region.match.edges:
					edges.append({
						"t": {"text": e.t.text, "index": e.t.index, "pos": e.t.pos},
						"flow": e.flow,
						"distance": e.distance,
						"metric": e.metric
					})
				regions.append(dict(s=s, pos_s=region.match.pos_s, edges=edges))
			else:
				regions.append(dict(s=s, gap_penalty=region.gap_penalty))
		return dict(
			slice=self.slice_id,
			location=span_info,
			score=self.score,
			metric=self.metric,
			regions=regions,
			omitted=self.omitted,
			level=self.level)



class _Winners:
	"""the result set of one query as the arrays the backend returned; the HipMatch objects of the query index into it"""
	__slots__ = ("index", "query", "top", "n", "sent", "docs", "starts", "ends", "gaps", "args", "qmag", "masks", "q_tag_codes", "transport")

	def __init__(self, index, query, top, gaps, args, qmag, masks, q_tag_codes):
		self.index, self.query, self.top, self.gaps, self.args = index, query, top, gaps, args
		self.qmag, self.masks, self.q_tag_codes = qmag, masks, q_tag_codes
		self.n = n = top.n
		self.sent = np.asarray(top.sentence[:n], dtype=np.int64)
		self.docs = index._slice_doc[self.sent]
		self.starts, self.ends = index._slice_start[self.sent], index._slice_end[self.sent]
		self.transport = args is not None and args.get("algorithm", core.VK_ALG_ALIGN) != core.VK_ALG_ALIGN


class HipMatch(Match):
	"""one winner of a search; what CoreMatch (vectorian/index.py:295-379) exposes.  As CoreMatch, it holds a reference to the
	native result (here: the result set's arrays, `_Winners`, and its place in them) and materialises flow, regions, omitted
	tokens and the flows of transport metrics when they are asked for -- a batch of 256 queries returns 2,560 matches, and
	building every flow eagerly cost four fifths of config 4's throughput at the operator level."""
	__slots__ = ("_w", "_i", "_flow_cache", "_index_map_cache")

	def __init__(self, winners, i):
		self._w, self._i = winners, i
		self._flow_cache = None
		self._index_map_cache = False   # False: not computed yet (None is a value: no filter)

	@property
	def _index(self):
		return self._w.index

	@property
	def _query(self):
		return self._w.query

	@property
	def _doc_index(self):
		return int(self._w.docs[self._i])

	@property
	def _slice_id(self):
		return self._w.index._slice_id[int(self._w.sent[self._i])]

	@property
	def _token_at(self):
		return self._w.index._slice_token_at[int(self._w.sent[self._i])]

	@property
	def _len_s(self):
		return int(self._w.ends[self._i] - self._w.starts[self._i])

	@property
	def _mapping(self):
		return self._w.top.mapping[self._i]

	@property
	def _edge_sim(self):
		return self._w.top.edge_sim[self._i]

	@property
	def _gaps(self):
		return self._w.gaps

	@property
	def _index_map(self):
		"""token filter: position among the slice's passing tokens -> position in the slice (None: no filter)"""
		if self._index_map_cache is False:
			w = self._w
			self._index_map_cache = w.index._index_map(int(w.sent[self._i]), w.masks) if w.masks else None
		return self._index_map_cache

	@property
	def _transport_flow(self):
		"""flow dict of a transport metric (sparse / dense), stated on first access; None for alignments"""
		w = self._w
		if not w.transport:
			return None
		if self._flow_cache is None:
			i = self._i
			state = w.index._transport_flow(w.query, w.top, i, int(w.sent[i]), w.args, w.qmag, self._index_map, w.q_tag_codes,
				span=(int(w.starts[i]), int(w.ends[i])))
			self._flow_cache = state() if state is not None else False
		return self._flow_cache if self._flow_cache is not False else None

	@property
	def _score(self):
		return float(self._w.top.score[self._i])

	@property
	def _raw_score(self):
		return float(self._w.top.raw_score[self._i])

	@property
	def index(self):
		return self._index

	@property
	def query(self):
		return self._query

	@property
	def prepared_doc(self):
		return self._index.session.documents[self._doc_index]

	@property
	def doc_index(self):
		return self._doc_index

	@property
	def slice_id(self):
		return self._slice_id

	@property
	def score(self):
		return self._score

	@property
	def score_max(self):
		"""reference_score (metric/alignment.h:84-106): matched weight + unmatched weight scaled by the share of
		unmatched weight to the power submatch_weight; len(query) for submatch_weight 0 and for transport metrics"""
		w = float(self._query.options.get("submatch_weight", 0.0))
		metric = self._query.options.get("metric", {})
		weights = np.ones(len(self._query), dtype=np.float32)
		if isinstance(metric, dict) and metric.get("metric") == "alignment-tag-weighted":
			weights = np.array([float(metric["tag_weights"].get(t, 1.0)) for t in self._query.tags], dtype=np.float32)
		total = float(weights.sum())
		if self._w.transport or w == 0.0 or total <= 0.0:
			return total
		matched = float(weights[np.asarray(self._mapping[:len(weights)]) >= 0].sum())
		return matched + ((total - matched) / total) ** w * (total - matched)

	@property
	def raw_score(self):
		return self._raw_score

	@property
	def metric(self):
		return self._index.metric_name

	@property
	def level(self):
		return "word"

	@property
	def flow(self):
		"""InjectiveFlow::to_py (vectorian/core/cpp/match/flow.cpp:190-216); per-edge values as
		ScoreComputer fills them (metric/alignment.h:335-345)"""
		if self._w.transport:
			return self._transport_flow
		target = self._mapping.astype(np.int16)
		matched = target >= 0
		return {
			"type": "injective",
			"target": target,
			"flow": matched.astype(np.float32),
			"dist": np.where(matched, 1.0 - self._edge_sim, 1.0).astype(np.float32)}

	def _edges(self):
		"""(target s, source t, flow, distance) of the flow, as Flow::to_edges (match/match.h:61-73,151-153,202-218)"""
		flow = self.flow
		if flow["type"] == "injective":
			return [(int(flow["target"][j]), j, float(flow["flow"][j]), float(flow["dist"][j]))
				for j in range(len(self._query)) if flow["target"][j] >= 0]
		if flow["type"] == "sparse":
			return [(int(s_), int(t), float(f), float(d)) for t, s_, f, d in zip(flow["source"], flow["target"], flow["flow"], flow["dist"])]
		ts, ss = np.nonzero(flow["flow"] > 0.0)
		return [(int(s_), int(t), float(flow["flow"][t, s_]), float(flow["dist"][t, s_])) for t, s_ in zip(ts, ss)]

	@property
	def omitted(self):
		# Flow::py_omitted (match/flow.cpp:170-188) over to_injective(): query tokens without a partner
		have = {t for _, t, f, _ in self._edges() if f > 0.0}
		return [self._query.tokens[j] for j in range(len(self._query)) if j not in have]

	def regions(self, context_size=10):
		"""Flow::py_regions (vectorian/core/cpp/match/flow.cpp:9-167) in token units: unmatched
		document stretches carry the gap penalty gap_cost_s(skipped), matched tokens their edges"""
		doc_tokens = self.prepared_doc.tokens
		gap_s, gap_t = self._gaps
		token_at = self._token_at
		all_edges = self._edges()
		if self._index_map is not None:
			# targets count the tokens that pass the query's token filter; back to slice positions (flow.cpp:49-60,96-97)
			all_edges = [(int(self._index_map[e[0]]),) + tuple(e[1:]) for e in all_edges]
		all_edges.sort(key=lambda e: (e[0], -e[2]))      # by target, biggest flow first (flow.cpp:32-41)
		edges = [(e[0], e[1]) for e in all_edges]
		weight = {(e[0], e[1]): (e[2], e[3]) for e in all_edges}
		text = lambda a, b: " ".join(doc_tokens[a:b])
		regions = []
		if not edges:
			regions.append(Region(s=text(token_at, token_at + self._len_s), match=None, gap_penalty=0.0))
			return regions
		last_anchor = max(0, token_at + edges[0][0] - context_size)
		last_matched = False
		last_source = -1
		k = 0
		while k < len(edges):
			target = edges[k][0]
			pos = token_at + target
			if pos > last_anchor:
				p = float(gap_s(pos - last_anchor)) if last_matched else 0.0
				regions.append(Region(s=text(last_anchor, pos), match=None, gap_penalty=p))
			region_edges = []
			while k < len(edges) and edges[k][0] == target:
				source = edges[k][1]
				if last_source >= 0:
					p = float(gap_t(source - last_source - 1))
					if p > 0.0:
						regions.append(Region(s="", match=None, gap_penalty=p))
				last_source = source
				region_edges.append(TokenMatchEdge(
					t=TokenMatchT(text=self._query.tokens[source], index=source, pos=None),
					flow=weight[(target, source)][0],
					distance=weight[(target, source)][1],
					metric=self.metric))
				k += 1
			regions.append(Region(s=doc_tokens[pos], match=TokenMatch(pos_s=None, edges=region_edges), gap_penalty=0.0))
			last_anchor = pos + 1
			last_matched = True
		up_to = min(last_anchor + context_size, len(doc_tokens) - 1)
		if up_to > last_anchor:
			regions.append(Region(s=text(last_anchor, up_to), match=None, gap_penalty=0.0))
		return regions


class PyMatch(Match):
	"""plain-data match (vectorian/index.py:382-431)"""

	def __init__(self, index, query, document, slice_id, score, metric=None, omitted=None, regions=None, level="word"):
		self._index = index
		self._query = query
		self._document = document
		self._slice_id = slice_id
		self._score = score
		self._metric = metric or ""
		self._omitted = omitted or []
		self._regions = regions or []
		self._level = level

	@property
	def index(self):
		return self._index

	@property
	def query(self):
		return self._query

	@property
	def prepared_doc(self):
		return self._document

	@property
	def slice_id(self):
		return self._slice_id

	@property
	def score(self):
		return self._score

	@property
	def score_max(self):
		return 1

	@property
	def metric(self):
		return self._metric

	@property
	def omitted(self):
		return self._omitted

	def regions(self, context_size=None):
		return self._regions

	@property
	def level(self):
		return self._level
